"""CPU tier: the C-ABI library loads and exports every symbol include/ryolo.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ryolo.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ryolo_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    if not os.path.exists(g.LIB):
        g.build()
    lib = ctypes.CDLL(g.LIB)
    names = _declared()
    assert "ryolo_rnms" in names and "ryolo_conv2d_bn_act" in names
    for n in names:
        assert hasattr(lib, n), "include/ryolo.h declares %s but libryolo_hip.so does not export it" % n
    lib.ryolo_abi_version.restype = ctypes.c_int
    assert lib.ryolo_abi_version() >= 1
    lib.ryolo_strerror.restype = ctypes.c_char_p
    assert lib.ryolo_strerror(-1) == b"invalid argument"


def test_argument_validation_without_gpu():
    # pure host-side checks: they return before anything is enqueued
    import __graft_entry__ as g
    lib = ctypes.CDLL(g.LIB)
    lib.ryolo_rnms.restype = ctypes.c_int
    vp = ctypes.c_void_p
    lib.ryolo_rnms.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, vp, vp, vp, ctypes.c_size_t, vp]
    assert lib.ryolo_rnms(None, -1, 6, 0.5, None, None, None, 0, None) == -1
    assert lib.ryolo_rnms(None, 5, 6, 0.5, None, ctypes.c_void_p(16), None, 0, None) == -1
    lib.ryolo_conv_packed_weight_bytes.restype = ctypes.c_size_t
    assert lib.ryolo_conv_packed_weight_bytes(504, 1024, 1) == (512 * 1024 + 128) * 2
    assert lib.ryolo_conv_packed_weight_bytes(32, 8, 3) == (128 * 128 + 128) * 2
    assert lib.ryolo_conv_packed_weight_bytes(32, 8, 5) == 0


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "rotate-yolov3_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "oracle/" not in txt or f.endswith(".hip"), f   # .hip files only CITE the oracle in comments


def test_r_nms_rejects_cpu_tensors_like_the_reference():
    import pytest
    import torch
    import rotate_yolov3_amd  # noqa: F401
    from rotate_yolov3_amd.utils.nms.r_nms import r_nms
    with pytest.raises(RuntimeError):
        r_nms(torch.zeros(4, 6), 0.5)

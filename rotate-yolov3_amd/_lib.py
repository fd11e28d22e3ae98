"""ctypes binding of include/ryolo.h (libryolo_hip.so).

The library is built in-tree by __graft_entry__.build() (hipcc --offload-arch=gfx950).  There is NO fallback:
if the shared object is missing, or a call returns an error code, a RuntimeError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RYOLO_HIP_LIB: another build of the same sources (the ablation build of tools/mp_ablate.py); never a different implementation
LIB_PATH = os.environ.get("RYOLO_HIP_LIB") or os.path.join(_HERE, "libryolo_hip.so")

_lib = None

_vp = C.c_void_p
_sigs = {
    "ryolo_strerror": (C.c_char_p, [C.c_int]),
    "ryolo_abi_version": (C.c_int, []),
    "ryolo_build_id": (C.c_char_p, []),
    "ryolo_set_tuning": (C.c_int, [C.c_char_p, C.c_char_p]),
    "ryolo_rnms_workspace_bytes": (C.c_size_t, [C.c_int]),
    "ryolo_rnms": (C.c_int, [_vp, C.c_int, C.c_int, C.c_float, _vp, _vp, _vp, C.c_size_t, _vp]),
    "ryolo_rnms_segmented_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "ryolo_rnms_segmented": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_float, _vp, _vp, C.c_size_t, _vp]),
    "ryolo_riou_pairs": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp]),
    "ryolo_riou_matrix": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp]),
    "ryolo_rnms_count_pairs": (None, [_vp]),
    "ryolo_skew_iou_pairs": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp]),
    "ryolo_skew_iou_matrix": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp]),
}


def declare(name, restype, argtypes):
    """Other modules (conv, decode) register their entry points here so all signatures live in one table."""
    _sigs[name] = (restype, argtypes)
    if _lib is not None:
        fn = getattr(getattr(_lib, "_real", _lib), name)
        fn.restype, fn.argtypes = restype, argtypes


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "rotate-yolov3_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the HIP hot path." % LIB_PATH)
        # torch first: its bundled HIP runtime (libamdhip64) must be the one already in the process when the library's own
        # dependency on libamdhip64 is resolved -- loading this library BEFORE torch pulls in the system ROCm runtime and the
        # process ends up with two, streams and memory from one, kernels registered with the other: every launch then fails
        # (seen as `build(); smoke()` in one process: "HIP launch failed")
        import torch  # noqa: F401
        _lib = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in _sigs.items():
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = restype, argtypes
    return _lib


TUNING_SWITCHES = ("RYOLO_CONV3X3", "RYOLO_CONV1X1", "RYOLO_RNMS_TILES", "RYOLO_MQ_KORDER", "RYOLO_BN_REDUCE_TILES", "RYOLO_STEM_DGRAD")


def set_tuning(name, value):
    """Set (value: str) or clear (None) one of the library's tuning switches in this process (include/ryolo.h: ryolo_set_tuning).  The
    library reads the environment only once, so tests and in-process A/Bs go through here."""
    rc = getattr(lib(), "_real", lib()).ryolo_set_tuning(name.encode(), None if value is None else str(value).encode())
    if rc != 0:
        raise RuntimeError("ryolo_set_tuning(%s): unknown switch" % name)


class _CallTracer(object):
    """Stands in for the CDLL while trace_calls() is active: every entry point called through lib() is bracketed by two events on
    torch's current stream and logged as (name, args, start, end).  Measurement plumbing for bench.py's in-run kernel tables; the
    product path never sees it."""

    def __init__(self, real, log):
        self._real, self._log = real, log

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if not name.startswith("ryolo_") or not callable(fn):
            return fn
        log = self._log

        def traced(*args):
            import torch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            log.append((name, args, e0, e1))
            return rc
        return traced


class trace_calls(object):
    """with trace_calls() as log: ... -> log = [(entry point, ctypes args, start event, end event)] of every library call made inside
    (launches must be eager: a hipGraph replay makes no calls)."""

    def __enter__(self):
        global _lib
        lib()
        self._saved = _lib
        self.log = []
        _lib = _CallTracer(self._saved, self.log)
        return self.log

    def __exit__(self, *exc):
        global _lib
        _lib = self._saved
        return False


def check(rc, what):
    if rc != 0:
        msg = lib().ryolo_strerror(rc)
        raise RuntimeError("%s failed: %s (code %d)" % (what, msg.decode() if msg else "?", rc))


def stream_ptr(device=None):
    import torch
    return torch.cuda.current_stream(device).cuda_stream

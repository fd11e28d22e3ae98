"""CPU tier: the C restatement of the reference's rotated IoU / NMS (oracle/riou_oracle.c) under AddressSanitizer and
UndefinedBehaviorSanitizer.  The reference's device function collects up to 24 intersection points into
8-element buffers (rotate_polygon_nms_kernel.cu:163-201: undefined behaviour); the restatement documents 24-point buffers as one
of its three definitions -- this test holds it to that on random, clustered and degenerate boxes, and checks that the sanitized
build returns the same keep lists and IoU values as the library the other tests use."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import riou

ODIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


def _boxes(kind, n, seed):
    d = riou.random_boxes(n, seed=seed, extent=120.0 if kind != "sparse" else 2000.0)
    rng = np.random.default_rng(seed)
    if kind == "degenerate":
        d[::7, 2] = 0.0                       # zero width: in_rect degenerates, the IoU is inf / nan in the reference arithmetic
        d[3::11, 2:4] = 0.0                   # zero size
        d[5::13, 2] = 1e-4                    # needles
        d[1::17] = d[0]                       # exact duplicates (up to 24 candidate points per pair)
        d[2::19, 4] = 0.0                     # axis-aligned, many coincident edges
    if kind == "clustered":
        d[:, :2] = 50.0 + rng.normal(0, 3.0, (n, 2)).astype(np.float32)
    return np.ascontiguousarray(d, dtype=np.float32)


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not on PATH")
@pytest.mark.parametrize("kind,n,thr", [("random", 1500, 0.5), ("clustered", 600, 0.3), ("degenerate", 800, 0.5), ("sparse", 300, 0.1),
                                        ("degenerate", 300, -0.5)])
def test_restatement_is_clean_under_asan_and_ubsan(tmp_path, kind, n, thr):
    subprocess.run(["make", "-C", ODIR, "-s", "san_driver"], check=True)
    d = _boxes(kind, n, seed=len(kind) * 31 + n)
    src, out = str(tmp_path / "boxes.bin"), str(tmp_path / "out.bin")
    d.tofile(src)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([os.path.join(ODIR, "san_driver"), src, str(n), repr(float(thr)), out], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-3000:]
    raw = np.fromfile(out, dtype=np.uint8)
    k = int(raw[:8].view(np.int64)[0])
    keep = raw[8:8 + 8 * k].view(np.int64)
    off = 8 + 8 * k
    k2 = int(raw[off:off + 8].view(np.int64)[0])
    keep2 = raw[off + 8:off + 8 + 8 * max(k2, 0)].view(np.int64)
    off += 8 + 8 * max(k2, 0)
    m = min(n, 256)
    iou = raw[off:off + 4 * m * m].view(np.float32).reshape(m, m)
    assert np.array_equal(keep, riou.rnms(d, thr))                       # same results as the optimized library build
    assert k2 == k and np.array_equal(keep2, keep)                       # lazy scan == the reference-shaped bit-matrix scan
    assert np.array_equal(iou.view(np.uint32), riou.riou_matrix(d[:m], d[:m]).view(np.uint32))

"""Generate golden vectors for the rotated-IoU / rotated-NMS path FROM THE REFERENCE's own arithmetic.

Runs only in the build container (needs /root/reference): oracle/Makefile compiles kernel.cu:19-260 of
/root/reference/utils/nms/src/rotate_polygon_nms_kernel.cu as host C++ into oracle/_ref/libref_riou.so;
this script calls it and writes small .npz fixtures (inputs + expected outputs, no reference source).

    python tests/golden/gen_riou_golden.py

Fixtures:
  riou_known4.npz      the 4 boxes of utils/nms/nms_wrapper_test.py:35-38 (first 6 columns), their IoU
                       matrix from the reference arithmetic, and r_nms(dets, 0.1) = [0, 3]
  riou_matrix256.npz   256 seeded boxes (SURVEY 8(d) config-3 distribution, 160 px extent so that many
                       pairs overlap) and their 256x256 IoU matrix
  rnms_keep_n{64,1000,8192}.npz  inputs + keep lists at thr 0.5
  rnms_keep_n50000.npz seed + keep list (inputs regenerated from the seed by oracle.riou.random_boxes)
                       + SHA-256 of the int64 little-endian keep list
"""
import hashlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import riou  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)
    assert riou.have_ref(), "needs /root/reference to build oracle/_ref"

    known = np.array([[150, 150, 100, 100, 0, 0.99],
                      [160, 160, 100, 100, 0, 0.88],
                      [150, 150, 100, 100, -0.7854, 0.66],
                      [300, 300, 100, 100, 0., 0.77]], dtype=np.float32)
    np.savez(os.path.join(OUT, "riou_known4.npz"), dets=known,
             iou=riou.riou_matrix(known, known, use_ref=True),
             keep_thr0p1=riou.rnms(known, 0.1, use_ref=True))

    b = riou.random_boxes(256, seed=7, extent=160.0)
    np.savez_compressed(os.path.join(OUT, "riou_matrix256.npz"), boxes=b,
                        iou=riou.riou_matrix(b, b, use_ref=True))

    for n, seed, extent in ((64, 11, 120.0), (1000, 12, 608.0), (8192, 13, 608.0)):
        d = riou.random_boxes(n, seed=seed, extent=extent)
        keep = riou.rnms(d, 0.5, use_ref=True)
        np.savez_compressed(os.path.join(OUT, "rnms_keep_n%d.npz" % n), dets=d, thr=np.float32(0.5), keep=keep)
        print(n, "kept", len(keep))

    n, seed = 50000, 0
    d = riou.random_boxes(n, seed=seed)
    keep = riou.rnms(d, 0.5, use_ref=True)
    sha = hashlib.sha256(keep.astype("<i8").tobytes()).hexdigest()
    dsha = hashlib.sha256(d.astype("<f4").tobytes()).hexdigest()
    np.savez_compressed(os.path.join(OUT, "rnms_keep_n50000.npz"), n=n, seed=seed, thr=np.float32(0.5),
                        keep=keep.astype(np.int32), keep_sha256=sha, dets_sha256=dsha)
    print(n, "kept", len(keep), sha)


if __name__ == "__main__":
    main()

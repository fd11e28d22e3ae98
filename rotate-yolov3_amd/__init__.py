"""rotate-yolov3_amd -- the MI355X (gfx950) hot path of rotated-YOLOv3 behind the reference's Python surface.

Layout (only what the hot path needs):
  csrc/            hand-written HIP kernels + the C ABI (include/ryolo.h) -> libryolo_hip.so
  _lib.py          ctypes binding of that C ABI (device pointers + stream from torch; no CPU fallback)
  utils/nms/       r_nms (mirror of the reference's pybind module) and non_max_suppression
  utils/           parse_config (cfg / data / hyp), geometry helpers
  model/           Darknet (cfg -> layer plan -> HIP conv stack), YOLO decode, loss
"""
__version__ = "0.1.0"

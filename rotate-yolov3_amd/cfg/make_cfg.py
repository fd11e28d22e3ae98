"""Generate Darknet .cfg text for the rotated-YOLOv3 topologies this build benchmarks.

The reference ships its cfgs as hand-edited copies of stock darknet files; the two BASELINE.json names
(cfg/yolov3.cfg, cfg/yolov3-tiny.cfg) do not even load in its own parser.  Here the same
topologies are produced by code so that width/height/anchors/classes are parameters:

  darknet53(...)   Darknet-53 trunk + 3-scale FPN head, 75 convs / 23 shortcuts / 4 routes / 2 upsamples / 3 yolo
                   (layer indices as SURVEY.md Appendix A: yolo at 82, 94, 106; routes 83, 86, 95, 98)
  tiny(...)        yolov3-tiny: 13 convs, 6 maxpools, 2 yolo heads

    python -m rotate_yolov3_amd.cfg.make_cfg darknet53 > yolov3.cfg
"""
import sys

ANCHORS_ARA = ("792, 2061, 3870, 6353, 9623, 15803 / 4.18, 6.48, 8.71  / "
               "-75, -60, -45, -30, -15 ,0,15, 30,45, 60,75, 90")   # the values of the reference's cfg/yolov3.cfg:609
TINY_PAIRS = "10,14,  23,27,  37,58,  81,82,  135,169,  344,319"


def _net(width, height):
    return ["[net]", "batch=16", "subdivisions=1", "width=%d" % width, "height=%d" % height, "channels=3", ""]


def _conv(filters, size, stride, bn=1, act="leaky"):
    out = ["[convolutional]"]
    if bn:
        out.append("batch_normalize=1")
    out += ["filters=%d" % filters, "size=%d" % size, "stride=%d" % stride, "pad=1", "activation=%s" % act, ""]
    return out


def _yolo(mask, anchors, classes):
    return ["[yolo]", "mask = %s" % mask, "anchors = %s" % anchors, "classes=%d" % classes, "num=9", ""]


def darknet53(width=608, height=608, anchors="ara " + ANCHORS_ARA, classes=1, na_per_head=72, masks=None):
    no = na_per_head * (classes + 6)
    masks = masks or ["%d-%d" % (2 * na_per_head, 3 * na_per_head - 1), "%d-%d" % (na_per_head, 2 * na_per_head - 1),
                      "0-%d" % (na_per_head - 1)]
    L = _net(width, height)
    L += _conv(32, 3, 1)
    for filters, nblocks in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
        L += _conv(filters, 3, 2)
        for _ in range(nblocks):
            L += _conv(filters // 2, 1, 1) + _conv(filters, 3, 1) + ["[shortcut]", "from=-3", "activation=linear", ""]
    # head 0 @ stride 32
    for _ in range(3):
        L += _conv(512, 1, 1) + _conv(1024, 3, 1)
    L += _conv(no, 1, 1, bn=0, act="linear") + _yolo(masks[0], anchors, classes)
    # head 1 @ stride 16
    L += ["[route]", "layers = -4", ""] + _conv(256, 1, 1) + ["[upsample]", "stride=2", ""]
    L += ["[route]", "layers = -1, 61", ""]
    for _ in range(3):
        L += _conv(256, 1, 1) + _conv(512, 3, 1)
    L += _conv(no, 1, 1, bn=0, act="linear") + _yolo(masks[1], anchors, classes)
    # head 2 @ stride 8
    L += ["[route]", "layers = -4", ""] + _conv(128, 1, 1) + ["[upsample]", "stride=2", ""]
    L += ["[route]", "layers = -1, 36", ""]
    for _ in range(3):
        L += _conv(128, 1, 1) + _conv(256, 3, 1)
    L += _conv(no, 1, 1, bn=0, act="linear") + _yolo(masks[2], anchors, classes)
    return "\n".join(L) + "\n"


def tiny(width=608, height=608, anchors=TINY_PAIRS, classes=80):
    """Stock yolov3-tiny topology with the rotated head width: stock (w,h) pairs x 12 angles, comma masks index
    pairs (3,4,5 -> anchors 36..71; 1,2,3 -> 12..47), 36 anchors per head, filters = 36*(classes+6)."""
    no = 36 * (classes + 6)
    L = _net(width, height)
    for f in (16, 32, 64, 128, 256):
        L += _conv(f, 3, 1) + ["[maxpool]", "size=2", "stride=2", ""]
    L += _conv(512, 3, 1) + ["[maxpool]", "size=2", "stride=1", ""]
    L += _conv(1024, 3, 1) + _conv(256, 1, 1) + _conv(512, 3, 1)
    L += _conv(no, 1, 1, bn=0, act="linear") + _yolo("3,4,5", anchors, classes)
    L += ["[route]", "layers = -4", ""] + _conv(128, 1, 1) + ["[upsample]", "stride=2", ""]
    L += ["[route]", "layers = -1, 8", ""] + _conv(256, 3, 1)
    L += _conv(no, 1, 1, bn=0, act="linear") + _yolo("1,2,3", anchors, classes)
    return "\n".join(L) + "\n"


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "darknet53"
    sys.stdout.write({"darknet53": darknet53, "tiny": tiny}[which]())

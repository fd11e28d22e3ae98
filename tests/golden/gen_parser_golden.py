"""Parser goldens from the REFERENCE's own utils/parse_config.py (parse_model_cfg, cfg2anchors) and utils/utils.py:hyp_parse
run in the build container on every cfg / hyp file the reference ships.  Stored: the parse OUTPUT only (per block: type and the
key/value strings; anchors as float64 arrays; the numeric content of the k-means anchor files the cfgs point to; hyp dicts),
plus which files the reference itself cannot load and why.  No reference file text is written into the repo; the test
re-serialises the stored blocks into equivalent cfg text to feed the product parser (tests/test_oracle_model.py).

    python tests/golden/gen_parser_golden.py        (needs /root/reference)
"""
import glob
import io
import json
import os
import sys
import types
import contextlib

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def main():
    assert os.path.isdir(REF)
    sys.path.insert(0, REF)
    os.chdir(REF)                      # anchor txt paths in the cfgs are relative to the repo root
    cv2 = types.ModuleType("cv2")
    cv2.setNumThreads = lambda n: None
    sys.modules["cv2"] = cv2
    sh = types.ModuleType("shapely")
    shg = types.ModuleType("shapely.geometry")
    shg.Polygon = shg.MultiPoint = object
    sh.geometry = shg
    sys.modules["shapely"] = sh
    sys.modules["shapely.geometry"] = shg
    import matplotlib
    matplotlib.use("Agg")
    from utils import parse_config as rpc
    from utils import utils as ru

    cfgs = sorted(glob.glob("cfg/*.cfg") + glob.glob("cfg/*/*.cfg"))
    out = {"cfgs": {}, "anchor_files": {}, "hyps": {}}
    arrays = {}
    for path in cfgs:
        try:
            defs = rpc.parse_model_cfg(path)
        except Exception as e:          # the reference cannot load some of its own files
            out["cfgs"][path] = {"loadable": False, "error": "%s: %s" % (type(e).__name__, str(e)[:120])}
            continue
        blocks = []
        for bi, d in enumerate(defs):
            b = {"type": d["type"], "kv": []}
            for k, v in d.items():
                if k == "type":
                    continue
                if isinstance(v, np.ndarray):
                    key = "%s|%d|%s" % (path, bi, k)
                    arrays[key] = v.astype(np.float64)
                    b["kv"].append([k, {"array": key}])
                else:
                    b["kv"].append([k, v if isinstance(v, str) else {"int": int(v)}])
            blocks.append(b)
        # the raw anchors value of every yolo block (the INPUT the product must accept): read from the file's text
        raw = [ln.split("=", 1)[1] for ln in open(path).read().split("\n") if ln.strip().startswith("anchors")]
        out["cfgs"][path] = {"loadable": True, "blocks": blocks, "anchors_raw": raw}
        for r in raw:
            r = r.strip()
            if "ara" not in r and os.path.isfile(r) and r not in out["anchor_files"]:
                arr = np.loadtxt(r)
                arrays["file|" + r] = arr.astype(np.float64)
                out["anchor_files"][r] = "file|" + r
    for path in sorted(glob.glob("cfg/*.py") + glob.glob("cfg/*/*.py")):
        with contextlib.redirect_stdout(io.StringIO()):
            h = ru.hyp_parse(path)
        lines = []
        for ln in open(path):
            if ln.startswith("#") or len(ln.strip()) == 0:
                continue
            v = ln.strip().split(":")
            lines.append([v[0], v[1].strip().split(" ")[0]])         # key and the value token hyp_parse evaluates
        out["hyps"][path] = {"parsed": {k: float(v) for k, v in h.items()}, "tokens": lines}
    np.savez_compressed(os.path.join(OUT, "parser_ref_arrays.npz"), **arrays)
    json.dump(out, open(os.path.join(OUT, "parser_ref_cfgs.json"), "w"), indent=0, sort_keys=True)
    for p, v in out["cfgs"].items():
        print(p, "loadable" if v["loadable"] else "NOT loadable: " + v["error"], len(v.get("blocks", [])))
    print("hyps:", {p: len(v["parsed"]) for p, v in out["hyps"].items()})


if __name__ == "__main__":
    main()

"""Summarise the FETCH_SIZE / WRITE_SIZE passes of tools/traffic_pmc.sh into profiles-style JSON (per-launch averages of the conv
kernel).  FETCH_SIZE on gfx950 reports half the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM section): doubled here;
FETCH_SIZE / WRITE_SIZE are in KiB-units of 1024 B as rocprofv3 reports them... they are reported in KB (1 unit = 1024 bytes)."""
import glob
import json
import os
import sqlite3
import sys

d = sys.argv[1]
wgrad = "--wgrad" in sys.argv
args = [a for a in sys.argv[2:] if a != "--wgrad"]
k, s, cin, cout, ho = [int(v) for v in args[:5]]
bs = 64 if wgrad else 32
pat = "%wgrad_wide%" if wgrad else "%conv%"          # (the split-K reduce is a launch of its own: not part of the tile kernel's traffic)
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    dbs = glob.glob(os.path.join(d, c, "**", "*.db"), recursive=True)
    if not dbs:
        res[c] = None
        continue
    con = sqlite3.connect(dbs[0])
    rows = con.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name = ? and kernel_name like ? "
                       "group by kernel_name", (c, pat)).fetchall()
    res[c] = [(r[0][:90], r[1], r[2]) for r in rows]
print(json.dumps(res, indent=1))
alg = 2.0 * bs * ((ho * s) ** 2 * cin + ho * ho * cout) + (4.0 if wgrad else 2.0) * k * k * cin * cout      # inputs + outputs once (wgrad: x, dz, fp32 dW)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
kname = None
try:          # the name bench.py's per-kernel table gives this layer (dry run of the library's dispatch; needs the GPU box)
    import rotate_yolov3_amd  # noqa: F401
    from rotate_yolov3_amd.model import hip_ops as _ops
    if wgrad:
        import torch
        from rotate_yolov3_amd import _lib
        from rotate_yolov3_amd.model import hip_train_ops as _tr
        import ctypes
        dd = _tr.make_desc(torch.empty(bs, ho * s, ho * s, cin, dtype=torch.bfloat16, device="cuda:0"), cout, k, s, (k - 1) // 2)
        code = _lib.lib().ryolo_conv_wgrad_kernel_choice(ctypes.byref(dd))
        kname = {256: "wgrad_wide<256,128>", 257: "wgrad_wide<128,256>", 258: "wgrad_wide<128,3x64>", 259: "wgrad_wide<128,128>", 260: "wgrad_wide<64,128>"}.get(code, "wgrad<%d>" % code)
    else:
        kname = _ops.conv_kernel_name(bs, ho * s, ho * s, cin, cout, k, s)
except Exception as e:
    kname = None
out = {"shape": "k%d s%d %d->%d @%d bs%d" % (k, s, cin, cout, ho, bs), "kernel": kname, "algorithmic_bytes_per_launch": alg, "raw": res,
       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/%s, tools/traffic_pmc.sh" % ("one_wgrad.py" if wgrad else "one_layer.py")}
try:
    f = max(res["FETCH_SIZE"], key=lambda r: r[2])[2] * 1024.0 * 2.0      # KB units; x2: gfx950 correction for 16-B/lane reads
    w = max(res["WRITE_SIZE"], key=lambda r: r[2])[2] * 1024.0
    out.update({"fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "hbm_bytes_per_launch": f + w,
                "ratio_to_algorithmic": (f + w) / alg})
except Exception as e:
    out["error"] = str(e)
json.dump(out, open(os.path.join(d, "traffic.json"), "w"), indent=1)
print(json.dumps({k2: v for k2, v in out.items() if k2 != "raw"}, indent=1))

"""CPU tier: the model-level oracle and the product's host logic against goldens captured from the reference's own
Python (tests/golden/gen_model_golden.py)."""
import os

import numpy as np
import torch

import rotate_yolov3_amd  # noqa: F401
from oracle import darknet_oracle as do
from rotate_yolov3_amd.cfg import make_cfg
from rotate_yolov3_amd.model.models import Darknet
from rotate_yolov3_amd.utils import parse_config as pc
from tests.procedural import fill_procedural

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_parser_matches_reference():
    z = np.load(os.path.join(G, "parser_ara.npz"))
    defs = pc.parse_model_cfg_text(make_cfg.darknet53())
    assert len(defs) == int(z["n_blocks"]) and [d["type"] for d in defs] == list(z["types"])
    y = [d for d in defs if d["type"] == "yolo"]
    assert np.array_equal(y[0]["anchors"], z["anchors"])                     # product parser, bit-exact fp64
    assert np.array_equal(do.anchors_of(make_cfg.ANCHORS_ARA), z["anchors"])   # oracle restatement
    # form 3 (no `ara` prefix, cfg/yolov3.cfg:609) parses to the same anchors
    assert np.array_equal(pc.cfg2anchors(make_cfg.ANCHORS_ARA), z["anchors"])
    assert pc.yolo_mask(y[0]) == list(range(144, 216))
    # stock tiny spelling: pairs x 12 angles, comma masks index pairs
    t = pc.parse_model_cfg_text(make_cfg.tiny())
    ty = [d for d in t if d["type"] == "yolo"]
    assert ty[0]["anchors"].shape == (72, 3) and pc.yolo_mask(ty[0]) == list(range(36, 72))
    assert np.array_equal(ty[0]["anchors"], do.anchors_of(make_cfg.TINY_PAIRS))


def test_hyp_parse_safe_arith(tmp_path):
    f = tmp_path / "hyp.py"
    f.write_text("giou: 0.1  # gain\nang_t: 3.1415926/12\n# c\n\nmultiplier:10\nlrf: -4.\n")
    h = pc.hyp_parse(str(f))
    assert h["giou"] == 0.1 and abs(h["ang_t"] - 3.1415926 / 12) < 1e-15 and h["multiplier"] == 10 and h["lrf"] == -4.0
    try:
        pc.safe_arith("__import__('os').system('true')")
        assert False
    except ValueError:
        pass


def test_decode_matches_reference():
    z = np.load(os.path.join(G, "decode_head0.npz"))
    io, p5 = do.decode(torch.from_numpy(z["head"]), z["anchors"], (128, 128))
    assert np.allclose(io.numpy(), z["io"], rtol=1e-6, atol=1e-6)
    assert np.array_equal(p5.numpy(), z["p"])


def _check_forward(fix, cfg_text):
    z = np.load(os.path.join(G, fix))
    m = fill_procedural(Darknet(cfg_text, {"context_factor": 1.0}).eval())
    x = torch.from_numpy(z["x"])
    # oracle restatement (functional) vs the reference
    io_o, p_o = do.forward(cfg_text, m.state_dict(), x)
    assert np.allclose(io_o.numpy(), z["io"], rtol=2e-4, atol=2e-4), np.abs(io_o.numpy() - z["io"]).max()
    # product's ATen chain (Darknet on a CPU tensor) vs the reference
    with torch.no_grad():
        io_p, p_p = m(x)
    assert np.allclose(io_p.numpy(), z["io"], rtol=2e-4, atol=2e-4)
    return z, m, io_o, p_o, p_p


def test_forward_darknet53_matches_reference():
    z, m, io_o, p_o, p_p = _check_forward("forward_d53_64.npz", make_cfg.darknet53())
    for k in range(3):
        assert np.allclose(p_o[k].numpy(), z["p%d" % k], rtol=2e-4, atol=2e-4)
        assert np.allclose(p_p[k].numpy(), z["p%d" % k], rtol=2e-4, atol=2e-4)
    # per-conv statistics of the oracle walk vs the hooks on the reference's modules
    _, _, outs = do.forward(make_cfg.darknet53(), m.state_dict(), torch.from_numpy(z["x"]), return_layers=True)
    for i, (mean, amean) in zip(z["conv_idx"], z["conv_stats"]):
        o = outs[int(i)]
        assert abs(float(o.mean()) - mean) < 1e-4 + 1e-3 * abs(mean) and abs(float(o.abs().mean()) - amean) < 1e-3 * amean + 1e-5
    assert len(m.state_dict()) == 510 and sum(p.numel() for p in m.parameters()) == 62396176


def test_forward_tiny_matches_reference():
    _check_forward("forward_tiny_64.npz", make_cfg.tiny())


def test_nms_wrapper_oracle_matches_reference():
    z = np.load(os.path.join(G, "nms_wrapper.npz"))
    pred = torch.from_numpy(z["pred"].copy())
    out = do.non_max_suppression(pred, 0.3, 0.5)
    assert np.array_equal(out[0].numpy(), z["det0"]) and np.array_equal(out[1].numpy(), z["det1"])
    assert np.array_equal(pred.numpy(), z["pred_after"], equal_nan=True)    # the in-place score update


def test_bf16_contract_mode_is_close_to_fp32():
    cfg = make_cfg.darknet53()
    m = fill_procedural(Darknet(cfg, None).eval())
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    io32, _ = do.forward(cfg, m.state_dict(), x)
    io16, _ = do.forward(cfg, m.state_dict(), x, bf16=True)
    rel = (io16 - io32).abs() / (io32.abs() + 1.0)
    assert float(rel.max()) < 0.15 and float(rel.mean()) < 0.01


def test_weights_roundtrip(tmp_path):
    from rotate_yolov3_amd.model.model_utils import load_darknet_weights, save_weights
    cfg = make_cfg.tiny()
    a = fill_procedural(Darknet(cfg, None))
    path = str(tmp_path / "t.weights")
    save_weights(a, path)
    b = Darknet(cfg, None)
    load_darknet_weights(b, path)
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        if "num_batches" in k or "activation" in k:
            continue       # PReLU slopes are not part of the darknet format
        assert torch.equal(sa[k], sb[k]), k
    assert os.path.getsize(path) == 20 + 4 * sum(v.numel() for k, v in sa.items() if "num_batches" not in k and "activation" not in k)


# ---------------------------------------------------------------- every cfg / hyp file the reference ships
def _ref_parser_fixture():
    import json
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return json.load(open(os.path.join(g, "parser_ref_cfgs.json"))), np.load(os.path.join(g, "parser_ref_arrays.npz"))


def _serialise(blocks, anchors_raw):
    """cfg text equivalent to what the reference parsed: one [type] line per block, key=value lines in the stored order; the
    anchors lines carry the ORIGINAL right-hand side (a path or an `ara` expression), i.e. the parser's input."""
    out, ai = [], 0
    for b in blocks:
        out.append("[%s]" % b["type"])
        for k, v in b["kv"]:
            if isinstance(v, dict) and "array" in v:
                out.append("%s=%s" % (k, anchors_raw[ai]))
                ai += 1
            elif isinstance(v, dict):
                continue                                   # the pre-seeded int batch_normalize=0 of blocks without the key
            else:
                out.append("%s=%s" % (k, v))
        out.append("")
    return "\n".join(out)


def _same_defs(defs, blocks, arrays):
    assert len(defs) == len(blocks)
    for d, b in zip(defs, blocks):
        assert d["type"] == b["type"]
        assert set(d) == {"type"} | {k for k, _ in b["kv"]}, (b["type"], sorted(d), [k for k, _ in b["kv"]])
        for k, v in b["kv"]:
            if isinstance(v, dict) and "array" in v:
                assert isinstance(d[k], np.ndarray) and d[k].dtype == np.float64 and np.array_equal(d[k], arrays[v["array"]]), k
            elif isinstance(v, dict):
                assert d[k] == v["int"] and isinstance(d[k], int)       # int 0, not the string "0" (load_darknet_weights tests truthiness)
            else:
                assert d[k] == v, (k, d[k], v)


def test_product_parser_reproduces_the_reference_parser_on_every_loadable_reference_cfg(tmp_path):
    from rotate_yolov3_amd.utils.parse_config import parse_model_cfg
    fx, arrays = _ref_parser_fixture()
    loadable = {p: v for p, v in fx["cfgs"].items() if v["loadable"]}
    assert len(loadable) >= 5 and any("txt" in r for v in loadable.values() for r in v["anchors_raw"])
    # the k-means anchor files the cfgs point to, re-created from their numeric content (form 2 of the grammar)
    for rel, key in fx["anchor_files"].items():
        dst = tmp_path / rel
        dst.parent.mkdir(parents=True, exist_ok=True)
        np.savetxt(str(dst), arrays[key], fmt="%.18g")
        np.savetxt(str(tmp_path / os.path.basename(rel)), arrays[key], fmt="%.18g")      # for the basename fallback (form 5)
    for path, v in loadable.items():
        cfg = tmp_path / ("c_" + path.replace("/", "_"))
        cfg.write_text(_serialise(v["blocks"], v["anchors_raw"]))
        _same_defs(parse_model_cfg(str(cfg)), v["blocks"], arrays)
        # form 5: the same file referenced by an absolute path that does not exist (cfg/HRSC/yolov3-416.cfg does that)
        if any("txt" in r for r in v["anchors_raw"]):
            raw5 = [" /py/rotated-yolo/" + r.strip() if "txt" in r else r for r in v["anchors_raw"]]
            cfg.write_text(_serialise(v["blocks"], raw5))
            _same_defs(parse_model_cfg(str(cfg)), v["blocks"], arrays)
    # the reference's own files, where they exist (build container only)
    if os.path.isdir("/root/reference"):
        cwd = os.getcwd()
        os.chdir("/root/reference")
        try:
            for path, v in loadable.items():
                _same_defs(parse_model_cfg(path), v["blocks"], arrays)
        finally:
            os.chdir(cwd)


def test_product_hyp_parse_equals_the_reference_on_every_shipped_hyp_file(tmp_path):
    from rotate_yolov3_amd.utils.parse_config import hyp_parse
    fx, _ = _ref_parser_fixture()
    assert len(fx["hyps"]) >= 4
    for path, v in fx["hyps"].items():
        f = tmp_path / "hyp.py"
        f.write_text("# comment\n\n" + "\n".join("%s: %s  # note" % (k, tok) if i % 2 else "%s:%s" % (k, tok)
                                                   for i, (k, tok) in enumerate(v["tokens"])) + "\n")
        got = hyp_parse(str(f))
        assert set(got) == set(v["parsed"])
        for k, want in v["parsed"].items():
            assert float(got[k]) == want, (path, k, got[k], want)
    if os.path.isdir("/root/reference"):
        for path, v in fx["hyps"].items():
            got = hyp_parse(os.path.join("/root/reference", path))
            assert {k: float(x) for k, x in got.items()} == v["parsed"]


def test_product_loads_weights_and_checkpoint_written_by_the_reference(tmp_path):
    """a .weights file written by the reference's save_weights (model_utils.py:95-118) and a checkpoint
    dict written by torch.save with the reference's layout (train.py:323-363), both produced in the build container by
    tests/golden/gen_weights_golden.py, go through the product's readers; the product's writer reproduces the file byte for byte."""
    import torch
    from rotate_yolov3_amd.model.model_utils import load_darknet_weights, save_weights
    from rotate_yolov3_amd.model.models import Darknet
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(g, "ref_saved_mini_state.npz"))
    cfg = str(z["cfg"])
    m = Darknet(cfg, {"context_factor": 1.0})
    assert set(m.state_dict()) == set(k for k in z.files if k != "cfg")          # same parameter names as the reference's module tree
    cutoff = load_darknet_weights(m, os.path.join(g, "ref_saved_mini.weights"))
    assert cutoff == -1 and int(m.seen[0]) == 12345 and list(m.version) == [0, 2, 5]
    sd = m.state_dict()
    for k in sd:
        if "num_batches_tracked" in k or "activation" in k:
            continue                 # not part of the darknet format (the reference's writer drops the PReLU slopes too)
        assert np.array_equal(sd[k].numpy(), z[k]), k
    out = tmp_path / "again.weights"
    save_weights(m, str(out))
    assert out.read_bytes() == open(os.path.join(g, "ref_saved_mini.weights"), "rb").read()
    # the .pt checkpoint
    ck = torch.load(os.path.join(g, "ref_saved_mini.pt"))
    assert set(ck) == {"epoch", "best_fitness", "training_results", "model", "optimizer"} and ck["epoch"] == 3
    m2 = Darknet(cfg, {"context_factor": 1.0})
    m2.load_state_dict(ck["model"])
    for k, v in m2.state_dict().items():
        assert np.array_equal(v.numpy(), z[k]), k
    # and both loaded models compute the same thing once the PReLU slopes (absent from the darknet format) are aligned
    with torch.no_grad():
        for k, v in m.state_dict().items():
            if "activation" in k:
                v.copy_(torch.from_numpy(z[k]))
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(0))
    m.eval(), m2.eval()
    with torch.no_grad():
        assert torch.equal(m(x)[0], m2(x)[0])

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

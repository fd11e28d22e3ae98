"""Darknet .cfg / .data / hyper-parameter parsers -- mirror of the reference's utils/parse_config.py
(parse_model_cfg :37-59, cfg2anchors :6-31, parse_data_cfg :62-75) and utils/utils.py:hyp_parse (:33-47).

Same return types as the reference (list of dicts with string values, anchors as float64 ndarray [n,3] of
(w_px, h_px, angle_rad)); the grammar is a SUPERSET so that the two cfg files the
reference ships but cannot load itself (cfg/yolov3.cfg, cfg/yolov3-tiny.cfg) parse:
  1. `ara <areas> / <ratios> / <angles_deg>`            reference form 1 (parse_config.py:7-24)
  2. `<path to a "w h" text file>`                      reference form 2: each row x 12 angles k*pi/12, k=-6..5
  3. `<areas> / <ratios> / <angles_deg>` without `ara`  (cfg/yolov3.cfg:609)            -> as 1
  4. stock darknet `w,h, w,h, ...` pairs                (cfg/yolov3-tiny.cfg:134)       -> as an inline file of 2
  5. a path that does not exist                         -> basename looked up beside the cfg / in ./utils/kmeans
`hyp_parse` evaluates `3.1415926/12`-style values with a small arithmetic evaluator instead of eval().
"""
import ast
import math
import operator
import os

import numpy as np

_ANGLES12 = np.array([i for i in range(-6, 6)], dtype=np.float64) * math.pi / 12


def _expand_wh(rows):
    rows = np.asarray(rows, dtype=np.float64).reshape(-1, 2)
    out = []
    for wh in rows:
        for a in _ANGLES12:
            out.append([wh[0], wh[1], a])
    return np.array(out, dtype=np.float64)


def _ara(parts):
    areas = [float(i) for i in parts[0].split(',') if i.strip()]
    ratios = [float(i) for i in parts[1].split(',') if i.strip()]
    angles = [float(i) for i in parts[2].split(',') if i.strip()]
    anchors = []
    for area in areas:              # area-major, then ratio, then angle (parse_config.py:13-20)
        for ratio in ratios:
            for angle in angles:
                anchors.append([math.sqrt(area * ratio), math.sqrt(area / ratio), angle * math.pi / 180])
    return np.array(anchors, dtype=np.float64)


def _find_anchor_file(path, cfg_dir):
    cands = [path, os.path.join(cfg_dir or '.', path), os.path.join(cfg_dir or '.', os.path.basename(path)),
             os.path.join('utils', 'kmeans', os.path.basename(path)),
             os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'cfg', os.path.basename(path))]
    for c in cands:
        if os.path.isfile(c):
            return c
    raise FileNotFoundError('anchor file %r not found (tried %s)' % (path, cands))


def cfg2anchors(val, cfg_dir=None):
    val = val.strip()
    if 'ara' in val:
        val = val[val.index('ara') + 3:]
        return _ara([i for i in val.split('/') if len(i.strip()) != 0])
    if '/' in val and not any(ch.isalpha() for ch in val):      # form 3: a / r / d without the prefix
        parts = [i for i in val.split('/') if len(i.strip()) != 0]
        if len(parts) == 3:
            return _ara(parts)
    if ',' in val and not any(ch.isalpha() for ch in val):      # form 4: stock darknet pairs
        nums = [float(i) for i in val.split(',') if i.strip()]
        if len(nums) % 2:
            raise ValueError('odd number of anchor values: %r' % val)
        return _expand_wh(nums)
    return _expand_wh(np.loadtxt(_find_anchor_file(val, cfg_dir)))   # forms 2 and 5


def parse_model_cfg(path):
    """Parses the yolo-v3 layer configuration file and returns module definitions (list of dicts; [0] = [net])."""
    cfg_dir = os.path.dirname(os.path.abspath(path))
    with open(path, 'r') as f:
        text = f.read()
    return parse_model_cfg_text(text, cfg_dir)


def parse_model_cfg_text(text, cfg_dir=None):
    lines = [x for x in text.split('\n') if x and not x.startswith('#')]
    lines = [x.rstrip().lstrip() for x in lines]
    mdefs = []
    for line in lines:
        if not line or line.startswith('#'):
            continue
        if line.startswith('['):
            mdefs.append({})
            mdefs[-1]['type'] = line[1:-1].rstrip()
            if mdefs[-1]['type'] == 'convolutional':
                mdefs[-1]['batch_normalize'] = 0   # int 0 while parsed values are strings, as in the reference
        else:
            key, val = line.split('=', 1)
            key = key.rstrip()
            if 'anchors' in key:
                mdefs[-1][key] = cfg2anchors(val, cfg_dir)
            else:
                mdefs[-1][key] = val.strip()
    return mdefs


def yolo_mask(mdef):
    """Anchor row indices selected by a [yolo] block.  Reference form 'lo-hi' (inclusive, models.py:123-124);
    also stock comma lists whose entries index (w,h) PAIRS: pair m -> anchors 12m .. 12m+11 (Appendix B.1)."""
    m = mdef['mask']
    if '-' in m:
        lo, hi = [int(i) for i in m.split('-')]
        return list(range(lo, hi + 1))
    out = []
    for p in [int(i) for i in m.split(',') if i.strip()]:
        out.extend(range(12 * p, 12 * p + 12))
    return out


def parse_data_cfg(path):
    options = dict()
    with open(path, 'r') as fp:
        lines = fp.readlines()
    for line in lines:
        line = line.strip()
        if line == '' or line.startswith('#'):
            continue
        key, val = line.split('=', 1)
        options[key.strip()] = val.strip()
    return options


_BIN = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: operator.truediv,
        ast.Pow: operator.pow, ast.Mod: operator.mod, ast.FloorDiv: operator.floordiv}
_UN = {ast.UAdd: operator.pos, ast.USub: operator.neg}


def safe_arith(expr):
    """Evaluate a numeric expression made of literals, + - * / ** % // and parentheses (no names, no calls)."""
    def ev(n):
        if isinstance(n, ast.Expression):
            return ev(n.body)
        if isinstance(n, ast.Constant) and isinstance(n.value, (int, float)):
            return n.value
        if isinstance(n, ast.BinOp) and type(n.op) in _BIN:
            return _BIN[type(n.op)](ev(n.left), ev(n.right))
        if isinstance(n, ast.UnaryOp) and type(n.op) in _UN:
            return _UN[type(n.op)](ev(n.operand))
        raise ValueError('unsupported expression in hyper-parameter file: %r' % expr)
    return ev(ast.parse(expr.strip(), mode='eval'))


def hyp_parse(hyp_path, verbose=False):
    """`key: value  # comment` lines -> dict; value float() else arithmetic (utils/utils.py:33-47)."""
    hyp = {}
    with open(hyp_path, 'r') as f:
        for line in f:
            if line.startswith('#') or len(line.strip()) == 0:
                continue
            v = line.strip().split(':')
            tok = v[1].strip().split(' ')[0]
            try:
                hyp[v[0]] = float(tok)
            except ValueError:
                hyp[v[0]] = safe_arith(tok)
    if verbose:
        print(hyp)
    return hyp

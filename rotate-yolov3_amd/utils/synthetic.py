"""Synthetic HRSC-shaped data: the benchmarks and entry points run without the
reference's OpenCV/imgaug input pipeline (utils/datasets.py, utils/augment.py -- out of scope, CPU side)."""
import math

import torch


def synthetic_targets(bs, seed=1, device="cpu"):
    """rows (img, cls, cx, cy, w, h, a): k~U{1..4} boxes per image; cx,cy~U(.15,.85); w~U(.10,.50); h=w/r, r~U(3,9);
    a~U(-pi/2, pi/2) open; class 0.  Satisfies the loader's assertions (utils/datasets.py:340-343)."""
    g = torch.Generator().manual_seed(seed)
    rows = []
    for i in range(bs):
        k = int(torch.randint(1, 5, (1,), generator=g))
        for _ in range(k):
            cx, cy = (0.15 + 0.70 * torch.rand(2, generator=g)).tolist()
            w = float(0.10 + 0.40 * torch.rand(1, generator=g))
            r = float(3.0 + 6.0 * torch.rand(1, generator=g))
            a = float((torch.rand(1, generator=g) - 0.5) * (math.pi - 1e-4))
            rows.append([i, 0, cx, cy, w, w / r, a])
    return torch.tensor(rows, dtype=torch.float32, device=device)


def random_boxes(n, seed=0, extent=608.0):
    """n rotated detections (cx, cy, w, h, angle, score) as a float32 numpy array, the BASELINE configs[2] distribution of
    SURVEY.md section 8(d): cx, cy ~ U(0, extent); w, h = 8 * 16^U(0,1); angle ~ U(-pi/2, pi/2); scores = a random
    permutation of (i + 0.5) / n (all distinct).  (tests/ keep their own copy next to the oracle; a CPU test holds the two equal.)"""
    import numpy as np
    rng = np.random.default_rng(seed)
    d = np.empty((n, 6), dtype=np.float32)
    d[:, 0] = rng.uniform(0, extent, n)
    d[:, 1] = rng.uniform(0, extent, n)
    d[:, 2] = 8.0 * 16.0 ** rng.uniform(0, 1, n)
    d[:, 3] = 8.0 * 16.0 ** rng.uniform(0, 1, n)
    d[:, 4] = rng.uniform(-np.pi / 2, np.pi / 2, n)
    d[:, 5] = (rng.permutation(n) + 0.5) / n
    return d


class SyntheticLoader(object):
    """Yields (imgs[bs,3,S,S] in [0,1], targets[nt,7], paths, shapes) like LoadImagesAndLabels' collate_fn."""

    def __init__(self, n_images, batch_size, img_size, seed=0, device="cpu"):
        self.n, self.bs, self.size, self.seed, self.device = n_images, batch_size, img_size, seed, device

    def __len__(self):
        return (self.n + self.bs - 1) // self.bs

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for b in range(len(self)):
            bs = min(self.bs, self.n - b * self.bs)
            imgs = torch.rand(bs, 3, self.size, self.size, generator=g).to(self.device)
            yield imgs, synthetic_targets(bs, seed=self.seed * 7919 + b, device=self.device), None, None

#!/usr/bin/env python3
"""A/B of the input-gradient bodies (matrix cores vs vector FMA) of the stride-1 kernels per layer shape: microseconds by
HIP events and the largest relative difference of dx.  SPECTRUM_DGRAD_VALU=1 selects the vector body at plan time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectrum_vae_amd import ops  # noqa: E402
from tools.kbench import LAYERS, timeit  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for name, (h, w, cin, cout, up, pooled, act, first) in LAYERS.items():
    if first:
        continue
    H, W = h * up, w * up
    g = ops.conv_geom(n, H, W, cin, 3, 3, 1, 1, "same", cout, (up, up))
    oh, ow = (H // 2, W // 2) if pooled else (H, W)
    x = torch.randn(n, h, w, cin, device="cuda")
    y = torch.randn(n, oh, ow, cout, device="cuda")
    dy = torch.randn(n, oh, ow, cout, device="cuda")
    wt = torch.randn(3, 3, cin, cout, device="cuda") * 0.1
    idx = torch.randint(0, 4, (n, oh, ow, cout), dtype=torch.uint8, device="cuda") if pooled else None
    res = {}
    for mode in ("mfma", "valu"):
        if mode == "valu":
            os.environ["SPECTRUM_DGRAD_VALU"] = "1"
        else:
            os.environ.pop("SPECTRUM_DGRAD_VALU", None)
        dx = torch.zeros_like(x)
        fn = lambda: ops.conv2d_bwd(g, x, wt, dy, y, idx, dx, None, None, act, 1.0, None)  # noqa: E731
        us = timeit(fn, 20)
        res[mode] = (us, dx.clone())
    d = (res["mfma"][1] - res["valu"][1]).abs().max().item() / max(res["valu"][1].abs().max().item(), 1e-30)
    print(f"{name}: mfma {res['mfma'][0]:7.1f} us   valu {res['valu'][0]:7.1f} us   rel diff dx {d:.2e}")

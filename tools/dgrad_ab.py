#!/usr/bin/env python3
"""Input-gradient kernels of the default stack per layer shape: microseconds by HIP events for the matrix-core body, the
vector body (plan-time option dgrad_valu) and a sweep of the workgroup count (option dgrad_grid), with the largest
relative difference of dx between the bodies.

usage: python tools/dgrad_ab.py [batch] [grid,grid,...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectrum_vae_amd import ops  # noqa: E402
from tools.kbench import LAYERS, timeit  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
grids = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else []
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
        with ops.option("dgrad_valu", 1 if mode == "valu" else -1):
            dx = torch.zeros_like(x)
            fn = lambda: ops.conv2d_bwd(g, x, wt, dy, y, idx, dx, None, None, act, 1.0, None)  # noqa: E731
            us = timeit(fn, 50)
            res[mode] = (us, dx.clone())
    d = (res["mfma"][1] - res["valu"][1]).abs().max().item() / max(res["valu"][1].abs().max().item(), 1e-30)
    line = f"{name}: mfma {res['mfma'][0]:7.1f} us   valu {res['valu'][0]:7.1f} us   rel diff dx {d:.2e}"
    for gr in grids:
        with ops.option("dgrad_grid", gr):
            dx = torch.zeros_like(x)
            us = timeit(lambda: ops.conv2d_bwd(g, x, wt, dy, y, idx, dx, None, None, act, 1.0, None), 50)
        line += f"   grid {gr}: {us:6.1f}"
    print(line)

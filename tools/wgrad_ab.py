#!/usr/bin/env python3
"""A/B of the kernel-gradient bodies (matrix cores vs vector FMA) per layer shape: microseconds by HIP events and the
largest difference between the two results.  SPECTRUM_WGRAD_VALU=1 selects the vector body at plan time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectrum_vae_amd import ops  # noqa: E402
from tools.kbench import LAYERS, timeit  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 256
for name, (h, w, cin, cout, up, pooled, act, first) in LAYERS.items():
    H, W = h * up, w * up
    g = ops.conv_geom(n, H, W, cin, 3, 3, 1, 1, "same", cout, (up, up))
    oh, ow = (H // 2, W // 2) if pooled else (H, W)
    x = torch.randn(n, h, w, cin, device="cuda")
    y = torch.randn(n, oh, ow, cout, device="cuda")
    dy = torch.randn(n, oh, ow, cout, device="cuda")
    idx = torch.randint(0, 4, (n, oh, ow, cout), dtype=torch.uint8, device="cuda") if pooled else None
    res = {}
    for mode in ("mfma", "valu"):
        if mode == "valu":
            os.environ["SPECTRUM_WGRAD_VALU"] = "1"
        else:
            os.environ.pop("SPECTRUM_WGRAD_VALU", None)
        ws = torch.empty(max(ops.conv2d_wgrad_workspace(g), 16), dtype=torch.uint8, device="cuda")
        job = dict(g=g, x=x, dy=dy, y=y, idx=idx, act=act, alpha=1.0, dw=torch.zeros(3, 3, cin, cout, device="cuda"),
                   db=torch.zeros(cout, device="cuda"), ws=ws)
        us = timeit(lambda: ops.conv2d_wgrad_batch([job], reduce=False, blocks=blocks), 20)
        job["dw"].zero_(), job["db"].zero_()
        ops.conv2d_wgrad_batch([job], reduce=True, blocks=blocks)
        res[mode] = (us, job["dw"].clone(), job["db"].clone())
    d = (res["mfma"][1] - res["valu"][1]).abs().max().item() / max(res["valu"][1].abs().max().item(), 1e-30)
    db = (res["mfma"][2] - res["valu"][2]).abs().max().item() / max(res["valu"][2].abs().max().item(), 1e-30)
    print(f"{name}: mfma {res['mfma'][0]:7.1f} us   valu {res['valu'][0]:7.1f} us   rel diff dw {d:.2e} db {db:.2e}")

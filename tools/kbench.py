#!/usr/bin/env python3
"""Micro-benchmark of single conv-layer entry points at the bench shapes (batch 1024).

  python tools/kbench.py [--layer e1|e2|e3|d1|d2|d3|all] [--op fwd|bwd|all] [--reps 50] [--batch 1024]

Times each C-ABI call with HIP events on the launch stream and prints microseconds, TFLOP/s (2 x MAC of the
GEMM view; bwd = kernel gradient + input gradient) and the fraction of the 157.3 TFLOP/s f32 peak.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectrum_vae_amd import ops  # noqa: E402

LAYERS = {  # name: (stored H, W, Cin, Cout, up, pooled, act, first)
    "e1": (32, 32, 3, 8, 1, True, "elu", True), "e2": (16, 16, 8, 16, 1, True, "elu", False),
    "e3": (8, 8, 16, 32, 1, True, "elu", False), "d1": (4, 4, 32, 16, 2, False, "elu", False),
    "d2": (8, 8, 16, 8, 2, False, "elu", False), "d3": (16, 16, 8, 3, 2, False, "tanh", False),
}


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", default="all")
    ap.add_argument("--op", default="all")
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--k", type=int, default=3)
    a = ap.parse_args()
    n, K = a.batch, a.k
    tot = 0.0
    for name, (h, w, cin, cout, up, pooled, act, first) in LAYERS.items():
        if a.layer not in ("all", name):
            continue
        H, W = h * up, w * up
        g = ops.conv_geom(n, H, W, cin, K, K, 1, 1, "same", cout, (up, up))
        x = torch.randn(n, h, w, cin, device="cuda")
        wt = torch.randn(K, K, cin, cout, device="cuda") * 0.1
        b = torch.zeros(cout, device="cuda")
        oh, ow = (H // 2, W // 2) if pooled else (H, W)
        y = torch.empty(n, oh, ow, cout, device="cuda")
        idx = torch.zeros(n, oh, ow, cout, dtype=torch.uint8, device="cuda") if pooled else None
        dy = torch.randn(n, oh, ow, cout, device="cuda")
        dx = None if first else torch.empty_like(x)
        dw, db = torch.zeros_like(wt), torch.zeros_like(b)
        ws = torch.empty(max(ops.conv2d_wgrad_workspace(g), 16), dtype=torch.uint8, device="cuda")
        macs = n * H * W * cout * K * K * cin
        if a.op in ("all", "fwd"):
            f = (lambda: ops.conv2d_pool_fwd(g, x, wt, b, y, idx, act, 1.0)) if pooled else (
                lambda: ops.conv2d_fwd(g, x, wt, b, y, act, 1.0))
            us = timeit(f, a.reps)
            tf = 2 * macs / us / 1e6
            tot += us
            print(f"{name} fwd  {us:8.1f} us  {tf:6.2f} TF  {tf / 157.3:6.1%}")
        if a.op in ("all", "bwd"):
            us = timeit(lambda: ops.conv2d_bwd(g, x, wt, dy, y, idx, dx, dw, db, act, 1.0, ws), a.reps)
            fl = 2 * macs * (1 if first else 2)
            tf = fl / us / 1e6
            tot += us
            print(f"{name} bwd  {us:8.1f} us  {tf:6.2f} TF  {tf / 157.3:6.1%}")
    print(f"total {tot:.1f} us")


if __name__ == "__main__":
    main()

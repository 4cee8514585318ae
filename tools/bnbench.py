#!/usr/bin/env python3
"""Per-call time of the BatchNorm entry points (training forward, backward) at the shapes of the GAN models, HIP events
around a HIP-graph replay of 20 calls.   python tools/bnbench.py [--dtype f32|bf16]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from spectrum_vae_amd import ops  # noqa: E402


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (5 * reps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=INT", help="plan-time option of the library")
    a = ap.parse_args()
    for kv in a.opt:
        ops.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    print(f"{'rows':>8} {'C':>5} {'grp':>3} {'fwd us':>8} {'bwd us':>8} {'bwd+act us':>10}  MB")
    for rows, c, g in [(16384, 64, 1), (4096, 64, 1), (1024, 128, 1), (8192, 64, 1), (2048, 128, 1), (512, 256, 1),
                       (32768, 64, 2), (8192, 64, 2), (2048, 128, 2), (1048576 // 4, 64, 1), (65536, 128, 1)]:
        x = torch.randn(rows, c, device="cuda").to(dt)
        dy = torch.randn(rows, c, device="cuda").to(dt)
        y, dx = torch.empty_like(x), torch.empty_like(x)
        gamma, beta = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
        mean, inv = torch.empty(g, c, device="cuda"), torch.empty(g, c, device="cuda")
        mm, mv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
        dg, db, dbias = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
        ws = torch.empty(ops.batchnorm_workspace(rows, c, g), dtype=torch.uint8, device="cuda")
        f = timed(lambda: ops.batchnorm_act_fwd_train(x, gamma, beta, y, mean, inv, mm, mv, rows, c, 1e-3, 0.99, True, g,
                                                      "lrelu", 0.2, ws))
        b = timed(lambda: ops.batchnorm_act_bwd(x, dy, None, None, 0.0, gamma, mean, inv, dx, dg, db, True, None, 0.0, None,
                                                True, rows, c, g, ws))
        b2 = timed(lambda: ops.batchnorm_act_bwd(x, dy, y, "lrelu", 0.2, gamma, mean, inv, dx, dg, db, True, "lrelu", 0.2,
                                                 dbias, True, rows, c, g, ws))
        print(f"{rows:8d} {c:5d} {g:3d} {f:8.1f} {b:8.1f} {b2:10.1f}  {rows * c * x.element_size() / 1e6:.2f}")


if __name__ == "__main__":
    main()

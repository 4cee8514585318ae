#!/usr/bin/env python3
"""Micro-benchmark of the generic conv entry points (the implicit-GEMM MFMA kernels) at the geometries of the U-Net and
context-encoder stacks: microseconds by HIP events, TFLOP/s of the GEMM view as a fraction of the matrix peak of the
dtype (f32 157.3 TFLOP/s, bf16 2500 TFLOP/s dense), and the algorithmic bytes (each operand and the result once) per
second as a fraction of the 8 TB/s HBM peak; the larger of the two fractions names the roofline that binds the layer.

  python tools/gemmbench.py [--batch 64] [--reps 30] [--only NAME] [--dtype f32|bf16]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectrum_vae_amd import ops  # noqa: E402

PEAKS = {"f32": 157.3e12, "bf16": 2500e12}
HBM = 8e12
# name: (kind, H, W, Cin, k, stride, padding, Cout)   conv: input (H, W, Cin);  convT: input (H, W, Cin) -> output Cout
GEOMS = {
    "unet.enc2": ("conv", 16, 16, 64, 2, 2, "same", 64), "unet.enc3": ("conv", 8, 8, 64, 2, 2, "same", 64),
    "unet.dec2": ("convT", 4, 4, 128, 2, 2, "same", 64), "unet.dec3": ("convT", 8, 8, 128, 2, 2, "same", 64),
    "unet.out": ("convT", 16, 16, 128, 2, 2, "same", 3),
    "ce.enc2": ("conv", 16, 16, 64, 4, 2, "same", 128), "ce.enc3": ("conv", 8, 8, 128, 4, 2, "same", 256),
    "ce.enc4": ("conv", 4, 4, 256, 4, 2, "same", 512), "ce.bottleneck": ("conv", 4, 4, 128, 4, 1, "valid", 1000),
    "ce.dec1": ("convT", 1, 1, 1000, 4, 1, "valid", 512), "ce.dec2": ("convT", 4, 4, 512, 4, 2, "same", 256),
    "ce.dec3": ("convT", 8, 8, 256, 4, 2, "same", 128), "ce.dec4": ("convT", 16, 16, 128, 4, 2, "same", 64),
    "ce.out": ("convT", 32, 32, 64, 4, 1, "same", 3), "ce.out_s2": ("convT", 16, 16, 64, 4, 2, "same", 3),
    "disc.c2": ("conv", 16, 16, 64, 4, 2, "same", 128), "disc.head": ("conv", 4, 4, 256, 4, 1, "valid", 1),
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
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--only", default="")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=INT", help="plan-time option of the library")
    a = ap.parse_args()
    for kv in a.opt:
        ops.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    n = a.batch
    dev = "cuda"
    PEAK = PEAKS[a.dtype]
    adt = torch.float32 if a.dtype == "f32" else torch.bfloat16
    eb = 4 if a.dtype == "f32" else 2
    print(f"# batch {n}, {a.dtype} storage, matrix peak {PEAK / 1e12:.1f} TFLOP/s, HBM peak {HBM / 1e12:.0f} TB/s")
    print(f"{'layer':14s} {'op':6s} {'us':>9s} {'TFLOP/s':>9s} {'of mfma':>8s} {'GB/s':>8s} {'of hbm':>7s} {'bound':>6s}")
    for name, (kind, H, W, cin, k, s, pad, cout) in GEOMS.items():
        if a.only and a.only not in name:
            continue
        if kind == "conv":
            g = ops.conv_geom(n, H, W, cin, k, k, s, s, pad, cout)
            x = torch.randn(n, H, W, cin, device=dev).to(adt)
            w = (torch.randn(k, k, cin, cout, device=dev) * 0.05).to(adt)
            y = torch.empty(n, g.Ho, g.Wo, cout, device=dev, dtype=adt)
            fl = 2.0 * n * g.Ho * g.Wo * k * k * cin * cout
            sc = {o: ops.conv2d_scratch(g, o) for o in (ops.OP_FWD, ops.OP_DGRAD)}
            wsf = torch.empty(max(sc[ops.OP_FWD], 16), dtype=torch.uint8, device=dev) if sc[ops.OP_FWD] else None
            wsd = torch.empty(max(sc[ops.OP_DGRAD], 16), dtype=torch.uint8, device=dev) if sc[ops.OP_DGRAD] else None
            wsw = torch.empty(max(ops.conv2d_wgrad_workspace(g), 16), dtype=torch.uint8, device=dev)
            dy, dx, dw = torch.randn_like(y), torch.empty_like(x), torch.empty(w.shape, device=dev)
            runs = [("fwd", lambda: ops.conv2d_fwd(g, x, w, None, y, "lrelu", 0.2, wsf)),
                    ("dgrad", lambda: ops.conv2d_dgrad(g, dy, w, dx, wsd)),
                    ("wgrad", lambda: ops.conv2d_wgrad(g, x, dy, dw, False, wsw))]
        else:
            g = ops.deconv_geom(n, H, W, cin, k, k, s, s, pad, cout)
            x = torch.randn(n, H, W, cin, device=dev).to(adt)
            w = (torch.randn(k, k, cout, cin, device=dev) * 0.05).to(adt)
            y = torch.empty(n, g.H, g.W, cout, device=dev, dtype=adt)
            fl = 2.0 * n * H * W * k * k * cin * cout
            sc = {o: ops.conv2d_scratch(g, o) for o in (ops.OP_TRANSPOSE_FWD, ops.OP_TRANSPOSE_DGRAD)}
            wsf = torch.empty(sc[ops.OP_TRANSPOSE_FWD], dtype=torch.uint8, device=dev) if sc[ops.OP_TRANSPOSE_FWD] else None
            wsd = torch.empty(sc[ops.OP_TRANSPOSE_DGRAD], dtype=torch.uint8, device=dev) if sc[ops.OP_TRANSPOSE_DGRAD] else None
            wsw = torch.empty(max(ops.conv2d_wgrad_workspace(g), 16), dtype=torch.uint8, device=dev)
            dy, dx, dw = torch.randn_like(y), torch.empty_like(x), torch.empty(w.shape, device=dev)
            runs = [("fwd", lambda: ops.conv2d_transpose_fwd(g, x, w, None, y, "relu", 0.0, wsf)),
                    ("dgrad", lambda: ops.conv2d_transpose_dgrad(g, dy, w, dx, wsd)),
                    ("wgrad", lambda: ops.conv2d_transpose_wgrad(g, x, dy, dw, False, wsw))]
        # algorithmic bytes: the two operands and the result, each once (kernel gradients leave as float32)
        nb = {"fwd": (x.numel() + w.numel() + y.numel()) * eb, "dgrad": (y.numel() + w.numel() + x.numel()) * eb,
              "wgrad": (x.numel() + y.numel()) * eb + w.numel() * 4}
        for op, fn in runs:
            us = timeit(fn, a.reps)
            fm, fh = fl / us * 1e6 / PEAK, nb[op] / us * 1e6 / HBM
            print(f"{name:14s} {op:6s} {us:9.1f} {fl / us / 1e6:9.2f} {fm:8.3f} {nb[op] / us / 1e3:8.0f} {fh:7.3f} "
                  f"{'mfma' if fm >= fh else 'hbm':>6s}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The channel-last matrix-core kernel for stride-1 SAME convs (option s1_mfma = 1) against the planar kernels (= 0) on the
layer shapes of the run_model.py hyper-parameter grid: largest relative difference and time per call of the forward, the
pooled forward and the input gradient (from an explicit dz), batch 1024 unless told otherwise.

  python tools/s1bench.py [--batch 1024] [--k 3,5]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from spectrum_vae_amd import ops  # noqa: E402

PEAK = 157.3e12
# (Cin, Cout, logical width, upsampled input)
SHAPES = [(16, 16, 32, 1), (16, 16, 32, 2), (16, 32, 16, 1), (32, 32, 16, 1), (32, 32, 16, 2), (32, 16, 16, 1),
          (32, 64, 8, 1), (64, 64, 8, 1), (64, 64, 8, 2), (64, 32, 8, 1), (64, 128, 4, 1), (128, 128, 4, 1),
          (128, 128, 4, 2), (128, 64, 4, 1), (3, 16, 32, 1), (16, 3, 32, 1), (8, 8, 32, 1), (8, 8, 16, 2), (3, 8, 32, 1),
          (8, 16, 16, 1), (16, 8, 16, 2), (8, 3, 32, 2)]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--k", default="3,5")
    ap.add_argument("--shapes", default="", help="cin,cout,w,up[;cin,cout,w,up...]: only these shapes")
    ap.add_argument("--only-s1", action="store_true", help="run the matrix-core kernels only (profiling)")
    a = ap.parse_args()
    shapes = [tuple(int(v) for v in t.split(",")) for t in a.shapes.split(";") if t] or SHAPES
    variants = (1, 1) if a.only_s1 else (0, 1)
    n = a.batch
    torch.manual_seed(1)
    print(f"# batch {n}; us per call: planar | s1_mfma (fraction of the {PEAK / 1e12:.1f} TF f32 peak) | max rel diff")
    for k in (int(v) for v in a.k.split(",")):
        for cin, cout, w_, up in shapes:
            h = w_
            g = ops.conv_geom(n, h, w_, cin, k, k, 1, 1, "same", cout, (up, up))
            x = torch.randn(n, h // up, w_ // up, cin, device="cuda")
            wt = torch.randn(k, k, cin, cout, device="cuda") * 0.1
            b = torch.randn(cout, device="cuda")
            fl = 2.0 * n * h * w_ * k * k * cin * cout
            row = f"k{k} {cin:3d}->{cout:3d} @{w_:2d} up{up}:"
            # forward (+ pooled forward)
            for pooled in (False, True):
                if pooled and (up == 2 or cout % 4):
                    continue
                res = {}
                for o in variants:
                    with ops.option("s1_mfma", o):
                        if pooled:
                            y = torch.zeros(n, h // 2, w_ // 2, cout, device="cuda")
                            idx = torch.zeros(n, h // 2, w_ // 2, cout, dtype=torch.uint8, device="cuda")
                            if not ops.conv2d_fused_supported(g, True):
                                res = None
                                break
                            f = lambda: ops.conv2d_pool_fwd(g, x, wt, b, y, idx, "elu", 1.0)  # noqa: E731
                        else:
                            y = torch.zeros(n, h, w_, cout, device="cuda")
                            idx = None
                            f = lambda: ops.conv2d_fwd(g, x, wt, b, y, "elu", 1.0)  # noqa: E731
                        t = timed(f)
                        res[o] = (t, y.clone(), None if idx is None else idx.clone())
                if res is None:
                    continue
                res.setdefault(0, res[1])
                d = float((res[0][1] - res[1][1]).abs().max() / res[0][1].abs().max())
                same_idx = "" if not pooled else f" idx_diff={int((res[0][2] != res[1][2]).sum())}"
                print(f"{row} {'pool' if pooled else 'fwd '} {res[0][0]:8.1f} | {res[1][0]:8.1f} ({fl / res[1][0] / 1e-6 / PEAK:.3f}) | {d:.2e}{same_idx}")
            # whole backward of the layer: dz formed from (dy, y[, idx]), input + kernel + bias gradients
            for pooled in (False, True):
                if pooled and (up == 2 or cout % 4):
                    continue
                oh, ow = (h // 2, w_ // 2) if pooled else (h, w_)
                yb = torch.randn(n, oh, ow, cout, device="cuda")
                dyb = torch.randn(n, oh, ow, cout, device="cuda")
                idxb = torch.randint(0, 4, (n, oh, ow, cout), dtype=torch.uint8, device="cuda") if pooled else None
                if not ops.conv2d_fused_supported(g, pooled):
                    continue
                res = {}
                for o in variants:
                    with ops.option("s1_mfma", o):
                        ws = torch.empty(max(ops.conv2d_wgrad_workspace(g), 16), dtype=torch.uint8, device="cuda")
                        dx = torch.zeros(n, h // up, w_ // up, cin, device="cuda")
                        dw, db = torch.zeros_like(wt), torch.zeros(cout, device="cuda")

                        def f():
                            dw.zero_(), db.zero_()
                            ops.conv2d_bwd(g, x, wt, dyb, yb, idxb, dx, dw, db, "elu", 1.0, ws)
                        t = timed(f)
                        res[o] = (t, dx.clone(), dw.clone(), db.clone())
                res.setdefault(0, res[1])
                d = max(float((res[0][i] - res[1][i]).abs().max() / res[0][i].abs().max()) for i in (1, 2, 3))
                print(f"{row} {'bwdp' if pooled else 'bwd '} {res[0][0]:8.1f} | {res[1][0]:8.1f} ({2 * fl / res[1][0] / 1e-6 / PEAK:.3f}) | {d:.2e}")
            # input gradient from an explicit dz
            dz = torch.randn(n, h, w_, cout, device="cuda")
            res = {}
            for o in variants:
                with ops.option("s1_mfma", o):
                    dx = torch.zeros(n, h // up, w_ // up, cin, device="cuda")
                    t = timed(lambda: ops.conv2d_dgrad(g, dz, wt, dx))
                    res[o] = (t, dx.clone())
            res.setdefault(0, res[1])
            d = float((res[0][1] - res[1][1]).abs().max() / res[0][1].abs().max())
            print(f"{row} dgrad {res[0][0]:8.1f} | {res[1][0]:8.1f} ({fl / res[1][0] / 1e-6 / PEAK:.3f}) | {d:.2e}")


if __name__ == "__main__":
    main()

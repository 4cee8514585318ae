#!/usr/bin/env python3
"""In-kernel phase timeline of the tiled conv kernels (diagnostic; needs `make -C spectrum_vae_amd/csrc TRACE=1`).

  python tools/ktrace.py [--layer e2] [--batch 1024]

For every layer and op (fwd, wgrad half, dgrad half of the fused backward) prints, over all workgroups and
relative to the first workgroup's start, the min / median / max time in microseconds of each phase mark:
0 start, 1 LDS zero + filter staging issued, 2 first input staged, 3 compute done, 4 epilogue done, 5 exit.
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SPECTRUM_HIP_LIB"] = os.path.join(ROOT, "spectrum_vae_amd", "libspectrum_hip_trace.so")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from spectrum_vae_amd import _lib, ops  # noqa: E402
from tools.kbench import LAYERS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", default="all")
    ap.add_argument("--batch", type=int, default=1024)
    a = ap.parse_args()
    lib = C.CDLL(_lib.LIB_PATH)
    lib.sv_debug_set_trace.argtypes = [C.c_void_p]
    NB = 8192
    tr = torch.zeros(NB * 8, dtype=torch.int64, device="cuda")
    assert lib.sv_debug_set_trace(tr.data_ptr()) == 0
    n, K = a.batch, 3
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
        runs = {"fwd": (lambda: ops.conv2d_pool_fwd(g, x, wt, b, y, idx, act, 1.0)) if pooled else (
            lambda: ops.conv2d_fwd(g, x, wt, b, y, act, 1.0)),
            "wgrad": lambda: ops.conv2d_bwd(g, x, wt, dy, y, idx, None, dw, db, act, 1.0, ws)}
        if not first:
            runs["dgrad"] = lambda: ops.conv2d_bwd(g, None, wt, dy, y, idx, dx, None, None, act, 1.0, None)
        for op, fn in runs.items():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            tr.zero_()
            torch.cuda.synchronize()
            fn()
            torch.cuda.synchronize()
            t = tr.cpu().numpy().reshape(NB, 8)
            live = t[:, 0] != 0
            t = t[live]
            t0 = t[:, 0].min()
            hw = t[:, 7]
            cu = (hw >> 8) & 0xF
            se = (hw >> 13) & 0x7
            xcc = (hw >> 32) & 0xF
            cuid = (xcc * 8 + se) * 16 + cu
            per_cu = np.bincount(np.unique(cuid, return_inverse=True)[1])
            print(f"{name} {op}: {t.shape[0]} workgroups on {per_cu.size} CUs (max {per_cu.max()} per CU)")
            for s in range(6):
                v = t[:, s]
                ok = v != 0
                if not ok.any():
                    continue
                d = (v[ok] - t0) / 100.0
                print(f"   mark {s}: min {d.min():7.2f}  med {np.median(d):7.2f}  max {d.max():7.2f} us")
            dur = (t[:, 5] - t[:, 0]) / 100.0
            print(f"   workgroup lifetime: min {dur.min():.2f} med {np.median(dur):.2f} max {dur.max():.2f} us; "
                  f"phases med: zero+w {np.median(t[:,1]-t[:,0])/100:.2f}, stage {np.median(t[:,2]-t[:,1])/100:.2f}, "
                  f"compute {np.median(t[:,3]-t[:,2])/100:.2f}, tail {np.median(t[:,5]-t[:,3])/100:.2f}")


if __name__ == "__main__":
    main()

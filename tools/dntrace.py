#!/usr/bin/env python3
"""Phase timeline of the channel-last input-gradient kernels per layer of the default stack (diagnostic; needs
`make -C spectrum_vae_amd/csrc TRACE=1`): over all workgroups and relative to the first workgroup's start, the
min / median / max time in microseconds of each mark: start, filters in registers, LDS cleared and first dz loaded,
then per pass 'dz staged' and 'tiles done', and exit with every store retired."""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lib = C.CDLL(_lib.LIB_PATH)
lib.sv_debug_set_dn_trace.argtypes = [C.c_void_p]
tr = torch.zeros(2048 * 16, dtype=torch.int64, device="cuda")
assert lib.sv_debug_set_dn_trace(tr.data_ptr()) == 0
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
    dx = torch.zeros_like(x)
    for _ in range(3):
        ops.conv2d_bwd(g, x, wt, dy, y, idx, dx, None, None, act, 1.0, None)
    torch.cuda.synchronize()
    tr.zero_()
    torch.cuda.synchronize()
    ops.conv2d_bwd(g, x, wt, dy, y, idx, dx, None, None, act, 1.0, None)
    torch.cuda.synchronize()
    t = tr.cpu().numpy().reshape(-1, 16)
    t = t[t[:, 0] != 0]
    t0 = t[:, 0].min()
    print(f"{name}: {len(t)} workgroups")
    labels = ["start", "filters", "cleared+dz0"] + [f"pass {p} {w}" for p in range(6) for w in ("staged", "done")] + ["exit"]
    prev = None
    for s, nm in enumerate(labels):
        if not t[:, s].any():
            continue
        v = (t[:, s] - t0) / 100.0
        d = "" if prev is None else f"   (+{np.median(v - prev):6.2f})"
        print(f"  {nm:14s} min {v.min():7.2f} med {np.median(v):7.2f} max {v.max():7.2f}{d}")
        prev = v

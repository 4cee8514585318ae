#!/usr/bin/env python3
"""Phase timeline of the kernel-gradient launch of the default stack (diagnostic; needs
`make -C spectrum_vae_amd/csrc TRACE=1`): per layer (one launch of 128 workgroups) or for the six layers in one launch,
median over the layer's workgroups of: start -> LDS cleared, then per pass the wait for its loads, the staging and the
matrix instructions, and the epilogue.

usage: python tools/wntrace.py [batch] [all]"""
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
together = len(sys.argv) > 2 and sys.argv[2] == "all"
lib = C.CDLL(_lib.LIB_PATH)
lib.sv_debug_set_wn_trace.argtypes = [C.c_void_p]
tr = torch.zeros(2048 * 32, dtype=torch.int64, device="cuda")
assert lib.sv_debug_set_wn_trace(tr.data_ptr()) == 0
jobs = {}
for name, (h, w, cin, cout, up, pooled, act, first) in LAYERS.items():
    H, W = h * up, w * up
    g = ops.conv_geom(n, H, W, cin, 3, 3, 1, 1, "same", cout, (up, up))
    oh, ow = (H // 2, W // 2) if pooled else (H, W)
    ws = torch.empty(max(ops.conv2d_wgrad_workspace(g), 16), dtype=torch.uint8, device="cuda")
    jobs[name] = dict(g=g, x=torch.randn(n, h, w, cin, device="cuda"), dy=torch.randn(n, oh, ow, cout, device="cuda"),
                      y=torch.randn(n, oh, ow, cout, device="cuda"),
                      idx=torch.randint(0, 4, (n, oh, ow, cout), dtype=torch.uint8, device="cuda") if pooled else None,
                      act=act, alpha=1.0, dw=torch.zeros(3, 3, cin, cout, device="cuda"),
                      db=torch.zeros(cout, device="cuda"), ws=ws)


def show(label, t):
    t = t[t[:, 0] != 0]
    if not len(t):
        return
    t0 = t[:, 0].min()
    med = lambda a: float(np.median(a)) / 100.0  # noqa: E731
    line = f"{label}: {len(t)} workgroups, start {med(t[:, 0] - t0):5.2f}, cleared +{med(t[:, 1] - t[:, 0]):5.2f}"
    waits, stages, comps = [], [], []
    for p in range(7):
        if not t[:, 5 + 4 * p].all():
            break
        waits.append(med(t[:, 3 + 4 * p] - t[:, 2 + 4 * p]))
        stages.append(med(t[:, 4 + 4 * p] - t[:, 3 + 4 * p]))
        comps.append(med(t[:, 5 + 4 * p] - t[:, 4 + 4 * p]))
    line += f", {len(waits)} passes traced: load wait {np.mean(waits):5.2f}  stage {np.mean(stages):5.2f}  matrix {np.mean(comps):5.2f}"
    line += f" (first pass {waits[0]:4.2f} / {stages[0]:4.2f} / {comps[0]:4.2f}), exit at {med(t[:, 31] - t0):6.2f} (max {float((t[:, 31] - t0).max()) / 100:6.2f})"
    print(line)


sets = [list(jobs.items())] if together else [[kv] for kv in jobs.items()]
for sel in sets:
    js = [j for _, j in sel]
    for _ in range(3):
        ops.conv2d_wgrad_batch_slabs(js)
    torch.cuda.synchronize()
    tr.zero_()
    torch.cuda.synchronize()
    ops.conv2d_wgrad_batch_slabs(js)
    torch.cuda.synchronize()
    t = tr.cpu().numpy().reshape(-1, 32)
    if together:
        # workgroups are laid out job after job, longest-lived jobs first: 128 each
        for k in range(len(js)):
            show(f"job slot {k}", t[128 * k:128 * (k + 1)])
    else:
        show(sel[0][0], t)

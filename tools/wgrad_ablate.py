#!/usr/bin/env python3
"""Ablation of the kernel-gradient launch per layer of the default stack (diagnostic; needs
`make -C spectrum_vae_amd/csrc TRACE=1`): microseconds of a 128-workgroup launch with parts of the kernel switched off
(SPECTRUM_WGRAD_DBG bits: 1 no matrix-core loop, 2 no LDS stores of the staging, 4 no global loads of later passes).
The results of an ablated launch are wrong by construction; only the times mean something."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SPECTRUM_HIP_LIB"] = os.path.join(ROOT, "spectrum_vae_amd", "libspectrum_hip_trace.so")
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from spectrum_vae_amd import ops  # noqa: E402
from tools.kbench import LAYERS  # noqa: E402
from tools.wgrad_ab import timeit  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for name, (h, w, cin, cout, up, pooled, act, first) in LAYERS.items():
    H, W = h * up, w * up
    g = ops.conv_geom(n, H, W, cin, 3, 3, 1, 1, "same", cout, (up, up))
    oh, ow = (H // 2, W // 2) if pooled else (H, W)
    ws = torch.empty(4 * max(ops.conv2d_wgrad_workspace(g), 16), dtype=torch.uint8, device="cuda")
    job = dict(g=g, x=torch.randn(n, h, w, cin, device="cuda"), dy=torch.randn(n, oh, ow, cout, device="cuda"),
               y=torch.randn(n, oh, ow, cout, device="cuda"),
               idx=torch.randint(0, 4, (n, oh, ow, cout), dtype=torch.uint8, device="cuda") if pooled else None,
               act=act, alpha=1.0, dw=torch.zeros(3, 3, cin, cout, device="cuda"), db=torch.zeros(cout, device="cuda"), ws=ws)
    line = f"{name}:"
    for dbg, label in ((0, "full"), (1, "no matrix loop"), (2, "no LDS stores"), (6, "no loads, no stores"), (7, "empty")):
        os.environ["SPECTRUM_WGRAD_DBG"] = str(dbg)
        line += f"  {label} {timeit(lambda: ops.conv2d_wgrad_batch_slabs([job]), 20):6.1f}"
    print(line)

#!/usr/bin/env python3
"""Kernel-gradient launch of the default stack: per layer and for the six layers in one launch, microseconds by HIP
events for the stored-pixel form of the upsampled layers against the form over the upsampled pixels (plan-time option
wgrad_stored = 0), and the largest relative difference between the two results.

usage: python tools/wgrad_ab.py [batch] [workgroups per layer] [layer=workgroups,...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spectrum_vae_amd import ops  # noqa: E402
from tools.kbench import LAYERS  # noqa: E402


def timeit(fn, reps):
    """Microseconds per call with the calls recorded into a graph (the host's launch path is not timed)."""
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(reps):
                fn()
        gr.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        gr.replay()
        e.record()
        torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    over = dict(kv.split("=") for kv in sys.argv[3].split(",")) if len(sys.argv) > 3 else {}
    torch.manual_seed(0)
    jobs = {}
    for name, (h, w, cin, cout, up, pooled, act, first) in LAYERS.items():
        H, W = h * up, w * up
        g = ops.conv_geom(n, H, W, cin, 3, 3, 1, 1, "same", cout, (up, up))
        oh, ow = (H // 2, W // 2) if pooled else (H, W)
        ws = torch.empty(4 * max(ops.conv2d_wgrad_workspace(g), 16), dtype=torch.uint8, device="cuda")
        jobs[name] = dict(g=g, x=torch.randn(n, h, w, cin, device="cuda"), dy=torch.randn(n, oh, ow, cout, device="cuda"),
                          y=torch.randn(n, oh, ow, cout, device="cuda"),
                          idx=torch.randint(0, 4, (n, oh, ow, cout), dtype=torch.uint8, device="cuda") if pooled else None,
                          act=act, alpha=1.0, dw=torch.zeros(3, 3, cin, cout, device="cuda"),
                          db=torch.zeros(cout, device="cuda"), ws=ws)
        if name in over:
            jobs[name]["blocks"] = int(over[name])


    def run(sel, stored):
        with ops.option("wgrad_stored", -1 if stored else 0):
            us = timeit(lambda: ops.conv2d_wgrad_batch_slabs(sel, blocks=blocks), 30)
            for j in sel:
                j["dw"].zero_(), j["db"].zero_()
            ops.conv2d_wgrad_batch(sel, reduce=True, blocks=blocks)
            torch.cuda.synchronize()
        return us, [(j["dw"].clone(), j["db"].clone()) for j in sel]


    def rel(a, b):
        return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


    for name, job in list(jobs.items()) + [("all", None)]:
        sel = list(jobs.values()) if job is None else [job]
        us1, r1 = run(sel, True)
        us0, r0 = run(sel, False)
        dw = max(rel(a[0], b[0]) for a, b in zip(r1, r0))
        db = max(rel(a[1], b[1]) for a, b in zip(r1, r0))
        print(f"{name:4s}: stored-pixel form {us1:7.1f} us   upsampled form {us0:7.1f} us   rel diff dw {dw:.2e} db {db:.2e}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Diagnostic: the backward pass of the headline model at batch N as (a) the five input-gradient launches followed by the
batched kernel-gradient launch (what the step does) and (b) every layer's kernel gradient on a second stream beside the
input-gradient launch of the same layer (its dy exists by then), the first layer's at the end.  HIP events, eager."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from spectrum_vae_amd import ops  # noqa: E402
from test_gpu_wgrad_nhwc import LAYERS, make_jobs  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    rng = np.random.default_rng(0)
    jobs, _ = make_jobs(n, rng)
    for j in jobs:
        j["blocks"] = 128
    ws = [torch.randn(3, 3, s[2], s[3], device="cuda") * 0.1 for s in LAYERS]
    dxs = [torch.empty_like(j["x"]) for j in jobs]

    def dgrad(l):
        j = jobs[l]
        ops.conv2d_bwd(j["g"], None, ws[l], j["dy"], j["y"], j["idx"], dxs[l], None, None, j["act"], 1.0, None)

    def serial():
        for l in range(5, 0, -1):
            dgrad(l)
        ops.conv2d_wgrad_batch(jobs, reduce=True)

    side = torch.cuda.Stream()

    def overlapped():
        main_s = torch.cuda.current_stream()
        for l in range(5, 0, -1):
            ev = torch.cuda.Event()
            ev.record(main_s)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                ops.conv2d_wgrad_batch([jobs[l]], reduce=True)
            dgrad(l)
        done = torch.cuda.Event()
        done.record(side)
        main_s.wait_event(done)
        ops.conv2d_wgrad_batch([jobs[0]], reduce=True)

    for name, fn in (("serial", serial), ("overlapped", overlapped)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name}: {e0.elapsed_time(e1) / 50 * 1000:.1f} us per backward pass (graph replay, batch {n})")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Times sv_conv2d_wgrad_batch (reduce = 1) on the six headline layers at batch N with HIP events (diagnostic)."""
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
    big = int(os.environ.get("WN_BLOCKS_LARGE", "128"))
    small = int(os.environ.get("WN_BLOCKS", "256"))
    only = os.environ.get("WN_ONLY")
    layers = LAYERS if only is None else [LAYERS[int(i)] for i in only.split(",")]
    rng = np.random.default_rng(0)
    jobs, _ = make_jobs(n, rng, layers=layers)
    for j in jobs:
        j["blocks"] = big if j["g"].H * j["g"].W >= 256 else small
    for _ in range(5):
        ops.conv2d_wgrad_batch(jobs, reduce=True)
    torch.cuda.synchronize()
    reps = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv2d_wgrad_batch(jobs, reduce=True)
    e1.record()
    torch.cuda.synchronize()
    print(f"n={n} layers={only or 'all'} blocks={big}/{small} dbg={os.environ.get('SPECTRUM_WGRAD_DBG', '0')} "
          f"nhwc={os.environ.get('SPECTRUM_WGRAD_NHWC', '1')}: {e0.elapsed_time(e1) / reps * 1000:.1f} us per call (both launches)")


if __name__ == "__main__":
    main()

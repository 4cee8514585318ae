#!/usr/bin/env python3
"""Runs the whole-network forward (and optionally one full train step) a few times: a target for rocprofv3 passes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from spectrum_vae_amd import autoencoder  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
m = autoencoder.CAEPadded(**bench.RUN_MODEL_CFG)
m.build((None, 32, 32, 3))
m._materialize(torch.device("cuda"))
x = torch.rand(n, 32, 32, 3, device="cuda")
for _ in range(reps):
    m.forward_raw(x, False)
torch.cuda.synchronize()
print("done")

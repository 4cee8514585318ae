#!/usr/bin/env python3
"""The context-encoder GAN step alone (batch 64, float32), for kernel traces."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from spectrum_vae_amd import context_encoder, run_context_encoder  # noqa: E402
from spectrum_vae_amd.data import synth_training_tensor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
X = torch.from_numpy(synth_training_tensor(B, 7)).to(dev)
mask = torch.from_numpy(run_context_encoder.batch_random_mask(X.cpu().numpy(), 0.25, np.random.default_rng(1))).to(dev)
ce, disc = context_encoder.ContextEncoder(), context_encoder.Discriminator()
o1, o2 = run_context_encoder.build_optimizer(2e-3), run_context_encoder.build_optimizer(2e-4)
for _ in range(steps):
    run_context_encoder.train_step(ce, disc, o1, o2, X, mask)
torch.cuda.synchronize()

#!/usr/bin/env python3
"""Training-step rates of the secondary configurations (not the bench line): the context-encoder + discriminator
step of run_context_encoder.py at batch 64 and the U-Net GAN step of run_unet.py at batch 32, device-resident inputs,
HIP-graph replay, under the float32 policy or the bf16 policy (bf16 activations / filters on the bf16 matrix cores,
float32 sums, master weights and Adam).  These models run on the implicit-GEMM conv kernels.

  python tools/bench_models.py [--steps 50] [--batch B] [--dtype f32|bf16]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# before the HIP runtime is loaded (it reads the variable then): the steps' two streams on hardware queues of their own
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from spectrum_vae_amd import context_encoder, mixed_precision, run_context_encoder, run_unet, unet  # noqa: E402
from spectrum_vae_amd.data import synth_training_tensor  # noqa: E402


def timed(fn, steps):
    """Mean step time of the best of five timed runs of ``steps`` steps (after 10 untimed ones): these steps are a few
    hundred microseconds of GPU work behind a host-side graph launch, so a single run also measures the host's mood."""
    for _ in range(10):
        fn()
    best = None
    for _ in range(5):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / steps
        best = dt if best is None or dt < best else best
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=0, help="override both batch sizes (default: the drivers' 64 and 32)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    mixed_precision.set_global_policy("mixed_bfloat16" if a.dtype == "bf16" else "float32")
    out = {"dtype": a.dtype}
    # context encoder + discriminator, batch 64 (run_context_encoder.py:380)
    B = a.batch or 64
    X = torch.from_numpy(synth_training_tensor(B, 7)).to(dev)
    mask = torch.from_numpy(run_context_encoder.batch_random_mask(X.cpu().numpy(), 0.25, np.random.default_rng(1))).to(dev)
    ce, disc = context_encoder.ContextEncoder(), context_encoder.Discriminator()
    o1, o2 = run_context_encoder.build_optimizer(2e-3), run_context_encoder.build_optimizer(2e-4)
    dt = timed(lambda: run_context_encoder.train_step(ce, disc, o1, o2, X, mask), a.steps)
    # algorithmic FLOP per sample of one step: generator fwd + bwd, discriminator fwd + bwd on real and fake for its own
    # update, plus its fwd + input-gradient pass for the generator's adversarial term (2 MAC, convs and dense only)
    flop = 107_937_792 + 2 * 40_906_752 + 14_159_872 * 2
    out[f"context_encoder_step_b{B}"] = {"ms_per_step": round(dt * 1e3, 3), "samples_per_s": round(B / dt, 1),
                                         "model_tflops": round(flop * B / dt / 1e12, 2)}
    # U-Net generator + discriminator, batch 32 (run_unet.py:296)
    B = a.batch or 32
    Y = torch.from_numpy(synth_training_tensor(B, 9)).to(dev)
    Xu = Y[..., :1].contiguous()
    gen, d2 = unet.Generator(), unet.Discriminator()
    g1, g2 = run_unet.build_optimizer(2e-4), run_unet.build_optimizer(2e-4)
    dt = timed(lambda: run_unet.train_step(gen, d2, g1, g2, Xu, Y), a.steps)
    # generator fwd + bwd (27 000 832), discriminator (4->64->128->256 4x4 s2, 4x4 head: 35 659 776 fwd FLOP) fwd + bwd on
    # real and fake, plus its forward + input-gradient pass for the generator's adversarial term
    flop_u = 27_000_832 + 2 * 3 * 35_659_776 + 2 * 35_659_776
    out[f"unet_gan_step_b{B}"] = {"ms_per_step": round(dt * 1e3, 3), "samples_per_s": round(B / dt, 1),
                                  "model_tflops": round(flop_u * B / dt / 1e12, 2)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

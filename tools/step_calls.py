#!/usr/bin/env python3
"""HIP-event time of every C-ABI call of one eager training step of a model (bench.py's OpTimer), largest first.

  python tools/step_calls.py vae|cae|sweep:<init>,<layers>,<mult>,<ops>,<k>,<bn> [--batch 1024]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from spectrum_vae_amd import autoencoder, losses, ops  # noqa: E402
from spectrum_vae_amd.run_model import build_optimizer  # noqa: E402


def build(spec: str):
    if spec == "vae":
        return autoencoder.VAEPadded(conv_layers=3, conv_layer_multiplier=2, initial_filters=8, activation="elu",
                                     kernel_size=3, conv_operations=1, fc_layers=2, fc_layer_size=64, latent_dimensions=8)
    if spec == "cae":
        return autoencoder.CAEPadded(**bench.RUN_MODEL_CFG)
    init, layers, mult, nops, k, bn = (int(v) for v in spec.split(":")[1].split(","))
    cfg = dict(bench.RUN_MODEL_CFG, initial_filters=init, conv_layers=layers, conv_layer_multiplier=mult,
               conv_operations=nops, kernel_size=k, batch_norm=bool(bn))
    return autoencoder.CAEPadded(**cfg)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    model = build(a.model)
    model.build((None, 32, 32, 3))
    model._materialize(dev)
    model.compile(optimizer=build_optimizer(learning_rate=5e-4), loss=losses.MeanSquaredError())
    X = torch.from_numpy(bench.synth_data(a.batch, 3)).to(dev)
    b = model._step_buffers(a.batch, (32, 32, 3))
    ops.fill(b["acc"], 0.0)
    b["x"].copy_(X)
    dt = bench_timed(lambda: model.train_on_device_batch(b, True, float(a.batch)), a.steps)
    fl = bench.conv_flops_per_sample(model)["fwd_bwd"]
    print(f"{a.model} batch {a.batch}: {dt * 1e3:.4f} ms/step (graph replay), {a.batch / dt:.0f} samples/s, "
          f"{fl * a.batch / dt / 1e12:.1f} TFLOP/s = {fl * a.batch / dt / 1e12 / bench.FP32_MATRIX_PEAK_TFLOPS:.3f} of the f32 peak "
          f"({fl / 1e6:.2f} MFLOP/sample)")
    model.use_graphs = False
    agg = {}
    reps = 5
    for _ in range(reps):
        with bench.OpTimer() as ot:
            model.train_on_device_batch(b, True, float(a.batch))
        for name, g, us in ot.table:
            key = (name, json.dumps({k: g[k] for k in ("H", "W", "Cin", "Cout", "KH", "up_h", "half") if k in g}) if g else "")
            r = agg.setdefault(key, [0.0, 0])
            r[0] += us
            r[1] += 1
    tot = 0.0
    for (name, g), (us, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"{name:34s} {us / n:8.1f} us x{n / reps:g} {g}")
        tot += us / reps
    print(f"sum of call times {tot:.1f} us")


def bench_timed(fn, steps):
    import time
    for _ in range(8):
        fn()
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / steps
        best = dt if best is None or dt < best else best
    return best


if __name__ == "__main__":
    main()

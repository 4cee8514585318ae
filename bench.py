#!/usr/bin/env python3
"""Throughput bench of the hot path: training steps of the run_model.py auto-encoder on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One step = one optimizer step of ``CAEPadded`` at the run_model.py hyper-parameters (3 conv blocks, 8 initial
filters, x2 multiplier, elu, 3x3, one conv per block, no bottleneck, no BatchNorm; 12 035 weights), float32,
per-GPU batch 1024: batch gather from the HBM-resident training tensor -> forward -> MSE (+dL/dprediction) ->
backward -> [gradient all-reduce over RCCL when N > 1] -> Keras-form Adam with its decayed learning rate.
Inputs are synthetic tensors in the reference's training format, resident in HBM before the clock starts.

Prints ONE JSON line (rank 0): train samples/s over all GPUs, plus
  roofline     - the step's dominant kernel timed with HIP events on the launch stream, its algorithmic FLOPs
                 (or bytes) per launch, and the fraction of the matching peak;
  cpu_baseline - the torch-CPU restatement of the same step timed on this box's host cores (rank 0, N = 1).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# Before the HIP runtime is loaded (it reads the variable then): the secondary GAN steps fork two streams inside their
# recorded graphs, which only run side by side on hardware queues of their own (HIP's default is 4 per process, where
# the branch usually shares the launch stream's queue).  The headline step is one stream and does not depend on it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PER_GPU_BATCH = 1024
DATASET = 8192  # samples resident in HBM per rank
RUN_MODEL_CFG = dict(conv_layers=3, conv_layer_multiplier=2, initial_filters=8, activation="elu", kernel_size=3,
                     conv_operations=1, fc_layers=0, fc_layer_size=256, dropout=0.2, latent_dimensions=0,
                     batch_norm=False)
FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X f32-input MFMA == f32 vector peak
HBM_PEAK_GBS = 8000.0


def synth_data(n: int, seed: int) -> np.ndarray:
    """(n, 32, 32, 3) float32 in the reference's training format; a pool of 512 distinct regions is generated
    through the data-prep path and tiled with per-sample row permutations (permuting haplotypes keeps the
    format) so start-up stays short."""
    from spectrum_vae_amd.data import synth_training_tensor
    base = synth_training_tensor(min(n, 512), seed)
    if n <= base.shape[0]:
        return base[:n]
    rng = np.random.default_rng(seed + 1)
    out = np.empty((n,) + base.shape[1:], np.float32)
    for i in range(n):
        out[i] = base[i % base.shape[0]][rng.permutation(32)]
    return out


def conv_flops_per_sample(model) -> dict:
    """Algorithmic FLOPs (2 x MAC) of the conv GEMMs: forward, input-gradient (none for the first layer),
    kernel-gradient."""
    from spectrum_vae_amd.nn import Conv2D, Conv2DTranspose, Dense
    layers = [l for l in model._all_layers() if isinstance(l, (Conv2D, Conv2DTranspose, Dense))]
    order = {id(l): i for i, l in enumerate(model.encoder.layers + model.decoder.layers)}
    layers.sort(key=lambda l: order.get(id(l), 0))
    fwd = sum(2 * l.macs_per_sample() for l in layers)
    dgrad = sum(2 * l.macs_per_sample() for l in layers[1:])
    return {"fwd": fwd, "fwd_bwd": 2 * fwd + dgrad, "per_layer_macs": [l.macs_per_sample() for l in layers]}


class OpTimer:
    """Times every C-ABI call of one eager step with HIP events recorded on the launch stream."""

    def __init__(self):
        self.records = []
        self.calls = []  # (name, info, unwrapped function, arguments) of every call, for re-issuing one of them

    def __enter__(self):
        from spectrum_vae_amd import _lib
        self._lib = _lib.lib
        self._orig = {}
        for name in _lib.SIGNATURES:
            fn = getattr(self._lib, name)
            if fn.restype is not None and name.startswith("sv_") and "workspace" not in name and "supported" not in name \
                    and name not in ("sv_last_error", "sv_version"):
                self._orig[name] = fn
                setattr(self._lib, name, self._wrap(name, fn))
        return self

    def _wrap(self, name, fn):
        from spectrum_vae_amd._lib import ConvGeom

        def call(*args):
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            rc = fn(*args)
            e.record()
            info = None
            g = getattr(args[0], "_obj", None) if args else None
            if isinstance(g, ConvGeom):
                info = {f: getattr(g, f) for f, _ in g._fields_}
                if name == "sv_conv2d_bwd":  # which half ran: dx = args[6], dw = args[7]
                    info["half"] = "dgrad" if not args[7] else ("wgrad" if not args[6] else "both")
            elif name in ("sv_conv2d_wgrad_batch", "sv_conv2d_wgrad_batch_slabs"):
                arr, nj = args[0], args[1]
                fl = 0
                for i in range(nj):
                    q = arr[i].g
                    fl += 2 * q.N * q.Ho * q.Wo * q.Cout * q.KH * q.KW * q.Cin
                shapes = sorted((arr[i].g.Cin, arr[i].g.Cout, arr[i].g.W) for i in range(nj))
                nhwc = all(sh in ((3, 8, 32), (8, 16, 16), (16, 32, 8), (32, 16, 8), (16, 8, 16), (8, 3, 32)) for sh in shapes)
                slabs = name.endswith("_slabs")  # one launch: the slabs stay for the optimizer launch to sum
                info = {"jobs": nj, "flops": fl, "single_launch": slabs or not args[2], "nhwc": nhwc, "slabs": slabs}
            elif name == "sv_conv2d_dgrad_chain":
                arr, nj = args[0], args[1]
                info = {"jobs": nj, "flops": sum(2 * arr[i].g.N * arr[i].g.Ho * arr[i].g.Wo * arr[i].g.Cout * arr[i].g.KH *
                                                 arr[i].g.KW * arr[i].g.Cin for i in range(nj))}
            elif name in ("sv_convnet_fwd", "sv_convnet_fwd_mse", "sv_convnet_fwd_mse_rows", "sv_convnet_fwd_mse_dgrad"):
                arr, nl, n, h, w = args[0], args[1], args[2], args[3], args[4]
                fl = 0
                for i in range(nl):
                    L = arr[i]
                    ho, wo = h * L.up, w * L.up
                    fl += 2 * n * ho * wo * L.K * L.K * L.Cin * L.Cout
                    h, w = (ho // 2, wo // 2) if L.pool else (ho, wo)
                chain = [(arr[i].Cin, arr[i].Cout, arr[i].up, arr[i].pool) for i in range(nl)]
                info = {"N": n, "layers": nl, "flops": fl, "default_chain": chain == DEFAULT_CHAIN and args[3] == 32 and args[4] == 32}
                if name == "sv_convnet_fwd_mse_dgrad":  # the input-gradient jobs the launch also runs: args[13], args[14]
                    jb, nj = args[13], args[14]
                    info["dgrad_jobs"] = nj
                    info["flops"] += sum(2 * jb[i].g.N * jb[i].g.Ho * jb[i].g.Wo * jb[i].g.Cout * jb[i].g.KH * jb[i].g.KW *
                                         jb[i].g.Cin for i in range(nj))
            self.records.append((name, info, s, e))
            self.calls.append((name, info, fn, args))
            return rc
        return call

    def __exit__(self, *exc):
        for name, fn in self._orig.items():
            setattr(self._lib, name, fn)
        torch.cuda.synchronize()
        self.table = [(n, g, s.elapsed_time(e) * 1e3) for n, g, s, e in self.records]  # microseconds


# (Cin, Cout, upsampling, pooled) of the conv stack of CAEPadded at the run_model.py defaults
DEFAULT_CHAIN = [(3, 8, 1, 1), (8, 16, 1, 1), (16, 32, 1, 1), (32, 16, 2, 0), (16, 8, 2, 0), (8, 3, 2, 0)]


def _opt(name: str) -> int:
    from spectrum_vae_amd import ops
    return ops.get_option(name)


def kernel_of_call(name: str, g):
    """(rocprof kernel name, algorithmic FLOPs) of a C-ABI call that is exactly ONE kernel launch, else None."""
    if g is None:
        return None
    if name in ("sv_convnet_fwd", "sv_convnet_fwd_mse", "sv_convnet_fwd_mse_rows"):
        # the layer table of the run_model.py defaults has a kernel built for it; other tables take the run-time-table one
        known = g.get("default_chain") and _opt("netfwd_generic") != 1
        return ("net_fwd_cae" if known else "net_fwd_s1<3>"), float(g["flops"])
    if name == "sv_convnet_fwd_mse_dgrad":
        return "net_fwd_dgrad_cae", float(g["flops"])  # forward + the input-gradient chain, one launch
    if name == "sv_conv2d_dgrad_chain":
        return "dgrad_chain_k", float(g["flops"])
    if name == "sv_conv2d_wgrad_batch_slabs":
        return "wgrad_nhwc_multi", float(g["flops"])  # exactly that launch; the optimizer launch sums its slabs
    if name == "sv_conv2d_wgrad_batch":
        if g["single_launch"]:
            return "wgrad_s1_multi<3>", float(g["flops"])
        return None  # two launches (partial sums, then their sum): no single kernel to hold the call's time against
    if "KH" not in g:
        return None
    fl = 2.0 * g["N"] * g["Ho"] * g["Wo"] * g["Cout"] * g["KH"] * g["KW"] * g["Cin"]
    k = g["KH"]
    if name == "sv_conv2d_bwd_partial":
        return f"wgrad_s1_lds<{k}, {8 if (k == 3 and g['Cout'] > 4) else 4}>", fl
    if name == "sv_conv2d_bwd" and g.get("half") == "dgrad":
        if k == 3 and (g["Cout"], g["Cin"], g["W"]) in ((3, 8, 32), (8, 16, 16), (16, 32, 8), (32, 16, 8), (16, 8, 16)) \
                and _opt("dgrad_nhwc") != 0:
            # last template argument: 1 = the stored-pixel form of a conv behind an upsampling, 2 = pixel pairs as result
            # columns for an 8-channel result (where the kernel has them), 0 = the direct form
            shape = (g["Cout"], g["Cin"], g["W"])
            mode = 0
            if g["up_h"] == 2 and shape in ((3, 8, 32), (8, 16, 16)) and _opt("dgrad_phase") != 0:
                mode = 1
            elif g["up_h"] == 1 and shape == (16, 8, 16) and _opt("dgrad_pair") != 0:
                mode = 2
            return f"dgrad_nhwc<{shape[0]}, {shape[1]}, {shape[2]}, {mode}>", fl
        epi = 2 if g["up_h"] == 2 else 0
        return f"conv_s1_lds<{k}, 1, {8 if g['Cin'] > 4 else 4}, {epi}>", fl
    if name == "sv_conv2d_pool_fwd":
        return f"conv_s1_lds<{k}, {g['up_h']}, {8 if g['Cout'] > 4 else 4}, 1>", fl
    if name == "sv_conv2d_fwd":
        return f"conv_s1_lds<{k}, {g['up_h']}, {8 if g['Cout'] > 4 else 4}, 0>", fl
    return None


BF16_MATRIX_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak


def _dominant_call(table, dtype: str) -> dict:
    """Of the conv / dense calls of one eager step, the one holding the most time, against the roof that bounds it:
    algorithmic FLOPs (2 x MAC) against the matrix peak of the storage type, or the bytes of its operands and result
    against HBM, whichever leaves the smaller fraction unexplained (the larger of the two fractions is reported)."""
    esz = 2 if dtype == "bf16" else 4
    peak_tf = BF16_MATRIX_PEAK_TFLOPS if dtype == "bf16" else FP32_MATRIX_PEAK_TFLOPS
    best = None
    for row in table:
        g = row["geom"]
        if not g or "KH" not in g:
            continue
        us = row["avg_us"] * row["calls_per_step"]
        if best is None or us > best[0]:
            best = (us, row)
    if best is None:
        return {}
    row = best[1]
    g = row["geom"]
    fl = 2.0 * g["N"] * g["Ho"] * g["Wo"] * g["Cout"] * g["KH"] * g["KW"] * g["Cin"]
    by = esz * (g["N"] * (g["H"] // g["up_h"]) * (g["W"] // g["up_w"]) * g["Cin"] + g["N"] * g["Ho"] * g["Wo"] * g["Cout"]
                + g["KH"] * g["KW"] * g["Cin"] * g["Cout"])
    t = row["avg_us"] * 1e-6
    f_m, f_h = fl / t / 1e12 / peak_tf, by / t / 1e9 / HBM_PEAK_GBS
    return {"call": row["call"], "geom": {k: g[k] for k in ("N", "H", "W", "Cin", "Cout", "KH", "SH", "up_h")},
            "avg_call_us": round(row["avg_us"], 2), "calls_per_step": row["calls_per_step"],
            "bound": "mfma" if f_m >= f_h else "hbm", "frac": round(max(f_m, f_h), 4),
            "frac_of_matrix_peak": round(f_m, 4), "frac_of_hbm_peak": round(f_h, 4)}


def run_secondary(device, steps: int = 40) -> dict:
    """The other single-GPU configurations under the same clock as the headline, after its timed region: the U-Net GAN
    step of run_unet.py:164-203 under the bf16 policy at batch 32 and 1024, the context-encoder + discriminator step of
    run_context_encoder.py:225-260 in f32 at batch 64, and one VAEPadded (MSE + KL) step at batch 1024.  Each: best of
    five runs of `steps` HIP-graph replays, and the dominant conv call of one eager step against its roof."""
    from spectrum_vae_amd import autoencoder, context_encoder, losses, mixed_precision, run_context_encoder, run_unet, unet
    from spectrum_vae_amd.data import synth_training_tensor
    from spectrum_vae_amd.run_model import build_optimizer

    def timed(fn):  # the same procedure as tools/bench_models.py: 10 untimed steps, best of five runs
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

    def per_call(fn_eager, dtype):
        agg = {}
        reps = 3
        for _ in range(reps):
            with OpTimer() as ot:
                fn_eager()
            for name, g, us in ot.table:
                key = (name, json.dumps(g, sort_keys=True) if g else "")
                a = agg.setdefault(key, {"us": 0.0, "calls": 0, "geom": g})
                a["us"] += us
                a["calls"] += 1
        table = [{"call": k[0], "geom": v["geom"], "avg_us": v["us"] / v["calls"], "calls_per_step": v["calls"] / reps}
                 for k, v in agg.items()]
        return _dominant_call(table, dtype)

    out = {}
    try:
        # ---- U-Net generator + discriminator, bf16 policy
        mixed_precision.set_global_policy("mixed_bfloat16")
        for B in (32, 1024):
            Y = torch.from_numpy(synth_training_tensor(min(B, 256), 9)).to(device)
            if B > Y.shape[0]:
                Y = Y.repeat((B + Y.shape[0] - 1) // Y.shape[0], 1, 1, 1)[:B].contiguous()
            Xu = Y[..., :1].contiguous()
            gen, d2 = unet.Generator(), unet.Discriminator()
            g1, g2 = run_unet.build_optimizer(2e-4), run_unet.build_optimizer(2e-4)
            dt = timed(lambda: run_unet.train_step(gen, d2, g1, g2, Xu, Y))
            dom = per_call(lambda: run_unet.train_step(gen, d2, g1, g2, Xu, Y, use_graph=False), "bf16")
            out[f"unet_gan_step_bf16_b{B}"] = {"ms_per_step": round(dt * 1e3, 4), "samples_per_s": round(B / dt, 1),
                                                "dtype": "bf16", "dominant": dom}
            del gen, d2, g1, g2
        # ---- context encoder + discriminator, f32
        mixed_precision.set_global_policy("float32")
        B = 64
        X = torch.from_numpy(synth_training_tensor(B, 7)).to(device)
        mask = torch.from_numpy(run_context_encoder.batch_random_mask(X.cpu().numpy(), 0.25, np.random.default_rng(1))).to(device)
        ce, disc = context_encoder.ContextEncoder(), context_encoder.Discriminator()
        o1, o2 = run_context_encoder.build_optimizer(2e-3), run_context_encoder.build_optimizer(2e-4)
        dt = timed(lambda: run_context_encoder.train_step(ce, disc, o1, o2, X, mask))
        dom = per_call(lambda: run_context_encoder.train_step(ce, disc, o1, o2, X, mask, use_graph=False), "f32")
        out[f"context_encoder_step_f32_b{B}"] = {"ms_per_step": round(dt * 1e3, 4), "samples_per_s": round(B / dt, 1),
                                                 "dtype": "f32", "dominant": dom}
        del ce, disc, o1, o2
        # ---- VAEPadded (no reference model uses Sampling; this build's variational form of CAEPadded), f32
        B = 1024
        vae = autoencoder.VAEPadded(conv_layers=3, conv_layer_multiplier=2, initial_filters=8, activation="elu",
                                    kernel_size=3, conv_operations=1, fc_layers=2, fc_layer_size=64, latent_dimensions=8)
        vae.build((None, 32, 32, 3))
        vae._materialize(device)
        vae.compile(optimizer=build_optimizer(learning_rate=5e-4), loss=losses.MeanSquaredError())
        Xv = torch.from_numpy(synth_data(B, 77)).to(device)
        bv = vae._step_buffers(B, (32, 32, 3))
        ops_fill = __import__("spectrum_vae_amd.ops", fromlist=["fill"]).fill
        ops_fill(bv["acc"], 0.0)
        bv["x"].copy_(Xv)
        dt = timed(lambda: vae.train_on_device_batch(bv, True, float(B)))
        vae.use_graphs = False
        dom = per_call(lambda: vae.train_on_device_batch(bv, True, float(B)), "f32")
        out[f"vae_padded_step_f32_b{B}"] = {"ms_per_step": round(dt * 1e3, 4), "samples_per_s": round(B / dt, 1),
                                            "dtype": "f32", "dominant": dom}
    except Exception as e:  # noqa: BLE001 - the headline line must still be printed
        out["error"] = f"{type(e).__name__}: {e}"
    finally:
        mixed_precision.set_global_policy("float32")
    return out


def run_cpu_baseline(budget_s: float = 14.0, batch: int = PER_GPU_BATCH) -> dict:
    """The torch-CPU restatement of the same training step (oracle/torch_ref.py), timed on the host cores: at the
    bench's batch (the reported value) and at the reference loop's own batch of 32 (run_model.py:234,289)."""
    from oracle import models as omodels
    from oracle import torch_ref
    host_cpus = os.cpu_count() or 1
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(host_cpus, 16)
    torch.set_num_threads(threads)

    def timed(bsz: int, budget: float):
        cfg = {k: v for k, v in RUN_MODEL_CFG.items()}
        enc, dec = omodels.caepadded(**cfg)
        ref = torch_ref.RefSequential(enc + dec, (32, 32, 3))
        opt = torch_ref.KerasAdam(ref.params, 5e-4, decay_steps=10, decay_rate=0.96)
        x = torch.from_numpy(synth_data(bsz, 1833))
        torch_ref.train_step_mse(ref, opt, x, x)  # warm-up
        t0 = time.perf_counter()
        steps = 0
        while True:
            torch_ref.train_step_mse(ref, opt, x, x)
            steps += 1
            el = time.perf_counter() - t0
            if el >= budget or steps >= 2000:
                break
        return steps, el

    steps, el = timed(batch, budget_s)
    steps32, el32 = timed(32, 6.0)
    return {"value": round(steps * batch / el, 1), "unit": "samples/s", "cores": threads, "kind": "port",
            "host_cpu_count": host_cpus,
            "value_batch32": round(steps32 * 32 / el32, 1),
            "sample": f"{steps} training steps at batch {batch} in {el:.1f} s (value) and {steps32} steps at the "
                      f"reference loop's batch 32 in {el32:.1f} s (value_batch32) of the torch-CPU restatement of the "
                      f"reference step (TensorFlow is not installable here), {threads} threads of {host_cpus} host CPUs"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the U-Net / context-encoder / VAE steps")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=INT",
                    help="plan-time option of the library (sv_set_option) for this run; measurements only")
    ap.add_argument("--profile-out", default=None, help="write the per-call HIP-event table to this JSON file")
    args = ap.parse_args()
    if os.environ.get("SV_BENCH_WATCHDOG"):  # diagnostic: dump every thread's stack if the run takes longer than this
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["SV_BENCH_WATCHDOG"]), exit=True)

    import torch.distributed as dist

    from spectrum_vae_amd import autoencoder, losses, ops, parallel
    from spectrum_vae_amd.run_model import build_optimizer

    for kv in args.opt:
        name, _, val = kv.partition("=")
        ops.set_option(name, int(val))
    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    model = autoencoder.CAEPadded(**RUN_MODEL_CFG)
    model.build((None, 32, 32, 3))
    model._materialize(device)
    model.use_graphs = not args.no_graph
    model.compile(optimizer=build_optimizer(learning_rate=5e-4), loss=losses.MeanSquaredError())
    if world > 1:
        red = parallel.GradReducer()
        red.broadcast_(model._params)
        model.distribute(red)

    X = torch.from_numpy(synth_data(DATASET, 1833 + rank)).to(device)
    bs = args.batch
    b = model._step_buffers(bs, (32, 32, 3))
    ops.fill(b["acc"], 0.0)
    gen = torch.Generator(device=device)
    gen.manual_seed(1833 + rank)
    n_batches = DATASET // bs
    perm = torch.randperm(DATASET, device=device, generator=gen)

    # the forward launch gathers its own batch where the model takes that form (one GPU, default stack); else a
    # gather launch in front of the step
    rows = model.step_rows(X, perm, bs) if model.rows_step_supported(X, bs) else None

    def step(i):
        if rows is not None:
            if i % n_batches == 0:
                model.restart_rows(rows)  # 8-byte device copy once per pass over the data
            b["rows"] = rows
        else:
            idx = perm[(i % n_batches) * bs:(i % n_batches + 1) * bs]
            ops.gather_rows(X, idx, b["x"])
        model.train_on_device_batch(b, True, float(bs))

    # set-up, before the W warm-up steps: step 0 runs eagerly (allocates every buffer), step 1 records the HIP graphs, and
    # a further 100 replays (about 30 ms) bring the clocks to their steady state - with the driver's W = 5 the timed region
    # would otherwise start on a GPU that has been busy for a millisecond
    setup_steps = 2 + (0 if args.no_graph else 100)
    for i in range(setup_steps):
        step(i)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(b["loss"].item())

    # ---- per-kernel timing of one eager step with HIP events (rank 0)
    roofline, table = None, None
    if rank == 0:
        model.use_graphs = False
        saved_red = model._reducer
        model._reducer = None
        agg = {}
        reps = 20
        for r in range(reps):
            with OpTimer() as ot:
                step(r)
            for name, g, us in ot.table:
                key = (name, json.dumps(g, sort_keys=True) if g else "")
                a = agg.setdefault(key, {"us": 0.0, "calls": 0, "geom": g})
                a["us"] += us
                a["calls"] += 1
        model._reducer = saved_red
        table = sorted(({"call": k[0], "geom": v["geom"], "avg_us": v["us"] / v["calls"],
                         "calls_per_step": v["calls"] / reps} for k, v in agg.items()), key=lambda d: -d["avg_us"])
        # group the single-launch calls by the kernel they launch (the name rocprofv3 reports); the dominant kernel is
        # the one with the most time per step
        groups = {}
        for row in table:
            kf = kernel_of_call(row["call"], row["geom"])
            if kf is None:
                continue
            gname, fl = kf
            gr = groups.setdefault(gname, {"us_per_step": 0.0, "launches_per_step": 0.0, "flops_per_step": 0.0})
            gr["us_per_step"] += row["avg_us"] * row["calls_per_step"]
            gr["launches_per_step"] += row["calls_per_step"]
            gr["flops_per_step"] += fl * row["calls_per_step"]
        if groups:
            kname, gr = max(groups.items(), key=lambda kv: kv[1]["us_per_step"])
            avg_us_in_step = gr["us_per_step"] / gr["launches_per_step"]
            fl = gr["flops_per_step"] / gr["launches_per_step"]
            # The dominant kernel's launch duration: its calls of the last eager step are issued again, 20 times back to
            # back between two HIP events on the launch stream, so the queue stays full and the host's launch path (5-7 us
            # of an eager call's event-to-event time) is not in the figure; this is the number to hold against AverageNs
            # of the same kernel in profiles/*_kernel_stats.csv.
            again = [(fn, a) for (n_, info, fn, a) in ot.calls
                     if (kernel_of_call(n_, info) or (None,))[0] == kname]
            reps_k = 20
            for fn, a in again:
                fn(*a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps_k):
                for fn, a in again:
                    fn(*a)
            e1.record()
            torch.cuda.synchronize()
            avg_us = e0.elapsed_time(e1) * 1e3 / (reps_k * max(len(again), 1)) if again else avg_us_in_step
            ach = fl / (avg_us * 1e-6) / 1e12
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
            if os.path.exists(tpath):
                with open(tpath) as f:
                    traffic = json.load(f).get("hbm_bytes_per_launch", {}).get(kname)
            roofline = {"bound": "mfma", "kernel": kname, "achieved": round(ach, 3), "peak": FP32_MATRIX_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / FP32_MATRIX_PEAK_TFLOPS, 4), "traffic": traffic,
                        "avg_launch_us": round(avg_us, 2), "avg_call_us_in_eager_step": round(avg_us_in_step, 2),
                        "launches_per_step": gr["launches_per_step"],
                        "flops_per_launch": fl, "time_share_of_step": round(avg_us * gr["launches_per_step"] / (elapsed / args.steps * 1e6), 3),
                        "note": "f32 vector FMA peak == f32 MFMA peak on MI355X; avg_launch_us: this kernel's calls of "
                                "one step issued 20 times back to back between two HIP events on the launch stream; "
                                "avg_call_us_in_eager_step: event-to-event time of the same calls inside an eager step "
                                "(includes the host launch path)"}
            roofline["kernels_ranked_by_eager_call_us"] = {k: round(v["us_per_step"], 1) for k, v in
                                                           sorted(groups.items(), key=lambda kv: -kv[1]["us_per_step"])[:4]}
            # calls that are more than one launch cannot be put against a single kernel's row; when such a call is longer
            # than the dominant kernel, say so beside it
            multi = [r_ for r_ in table if kernel_of_call(r_["call"], r_["geom"]) is None and r_["geom"] and "flops" in r_["geom"]]
            if multi and multi[0]["avg_us"] * multi[0]["calls_per_step"] > avg_us_in_step * gr["launches_per_step"]:
                m = multi[0]
                roofline["longer_multi_launch_call"] = {
                    "call": m["call"], "avg_call_us_in_eager_step": round(m["avg_us"], 2),
                    "frac_of_peak_over_the_call": round(m["geom"]["flops"] / (m["avg_us"] * 1e-6) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4),
                    "note": "a call of two or more launches (see profiles/*_kernel_stats.csv)"}
        else:
            top = table[0]
            roofline = {"bound": "hbm", "kernel": top["call"], "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": None, "traffic": None, "avg_launch_us": round(top["avg_us"], 2)}
        if args.profile_out:
            with open(args.profile_out, "w") as f:
                json.dump({"batch": bs, "per_call": table}, f, indent=1)

    secondary = None
    if rank == 0 and world == 1 and not args.no_secondary:
        import gc
        gc.collect()
        torch.cuda.empty_cache()  # the secondary models record their graphs into pools of their own, as in a fresh process
        secondary = run_secondary(device)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = run_cpu_baseline()

    if rank == 0:
        fl = conv_flops_per_sample(model)
        samples = args.steps * bs * world
        out = {
            "metric": "train samples/sec (fwd+bwd+loss+optimizer)",
            "value": round(samples / elapsed, 1),
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "CAEPadded run_model.py defaults (12035 weights), MSE + Keras Adam, "
                                   f"per-GPU batch {bs}, 32x32x3 inputs", "global_batch": bs * world,
                       "parallelism": f"dp{world}", "hip_graph": bool(model.use_graphs or not args.no_graph),
                       "setup_steps": setup_steps,
                       "conv_flops_per_sample_fwd_bwd": fl["fwd_bwd"]},
            "model_tflops": round(fl["fwd_bwd"] * samples / elapsed / 1e12, 3),
            "final_loss": final_loss,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "secondary": secondary,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()  # rank 0 is the last to arrive (per-kernel timing above); tear the group down together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

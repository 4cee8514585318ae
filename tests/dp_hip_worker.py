"""Child process of tests/test_gpu_dp.py: one data-parallel rank running the HIP training step.

  python dp_hip_worker.py RANK WORLD PORT OUTDIR BATCH_PER_RANK STEPS [KIND]

Every rank builds the run_model.py auto-encoder from the same seed, takes its contiguous shard of one common
training tensor, and runs STEPS optimizer steps through ``Model.train_on_device_batch`` with a ``GradReducer``
attached (two recorded graphs with the all-reduce between them; gloo when the ranks share one GPU).  WORLD = 1 runs
the same steps on the whole batch without a reducer.  Weights and per-step losses go to OUTDIR/rank<r>.npz.
KIND: "cae" (default), "cae_bn" = the auto-encoder with BatchNormalization and "disc" = context_encoder.Discriminator
(MSE against a constant), both with the reducer's sync_batchnorm on: statistics over the rows of both ranks, steps run
eagerly (the exchange inside the passes goes through host memory with gloo).  "ce_overlap" / "ce_plain" =
context_encoder.ContextEncoder (18 MB of gradients) with 1 MB buckets, eager steps, the buckets sent during the backward
pass or after it.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CFG = dict(conv_layers=3, conv_layer_multiplier=2, initial_filters=8, activation="elu", kernel_size=3,
           conv_operations=1, fc_layers=0, fc_layer_size=256, dropout=0.2, latent_dimensions=0, batch_norm=False)


def main():
    rank, world, port, out, bpr, steps = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4],
                                          int(sys.argv[5]), int(sys.argv[6]))
    kind = sys.argv[7] if len(sys.argv) > 7 else "cae"
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                      SPECTRUM_DIST_BACKEND="gloo")
    from spectrum_vae_amd import autoencoder, context_encoder, losses, ops, parallel
    from spectrum_vae_amd.data import synth_training_tensor
    from spectrum_vae_amd.run_model import build_optimizer
    r, w, local = parallel.init_from_env()
    assert (r, w) == (rank, world)
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    if kind.startswith("ce_"):
        model = context_encoder.ContextEncoder()
        model.use_graphs = False
    elif kind == "disc":
        model = context_encoder.Discriminator()
    else:
        model = autoencoder.CAEPadded(**dict(CFG, batch_norm=kind == "cae_bn"))
    model.build((None, 32, 32, 3))
    model._materialize(device)
    model.compile(optimizer=build_optimizer(learning_rate=5e-4), loss=losses.MeanSquaredError())
    red = None
    if world > 1:
        red = parallel.GradReducer(sync_batchnorm=kind in ("cae_bn", "disc"))
        assert red.world == world
        if kind.startswith("ce_"):
            red.bucket_bytes = 1 << 20
            red.overlap = kind == "ce_overlap"
        if rank != 0:  # replicas start different on purpose: the broadcast must make them equal
            model._params.mul_(1.5)
        red.broadcast_(model._params)
        model.distribute(red)
    n_groups = 3
    X = synth_training_tensor(n_groups * bpr * world, 77)
    Xd = torch.from_numpy(X).to(device)
    bs = bpr
    same_xy = kind != "disc"
    b = model._step_buffers(bs, (32, 32, 3) if same_xy else (1,))
    if not same_xy:
        b["y"].fill_(0.3)
    step_losses = []
    for step in range(steps):
        g0 = (step % n_groups) * bpr * world  # the global batch of this step; rank r owns rows [r * bpr, (r + 1) * bpr)
        ops.gather_rows(Xd[g0 + rank * bpr:g0 + (rank + 1) * bpr], None, b["x"])
        model.train_on_device_batch(b, same_xy, float(bs))
        step_losses.append(float(b["loss"].item()))
    if kind == "cae" or (world == 1 and not kind.startswith("ce_")):
        g = next(iter(model._graphs.values()))
        assert not isinstance(g, str) and len(g) == (2 if world > 1 else 1), "the step was not replayed from graphs"
    if red is not None:
        assert red.calls == steps
    state = [v.numpy().ravel() for v in model.non_trainable_variables]
    order, early = red.last_pipeline if red is not None and red.last_pipeline else ((), 0)
    np.savez(os.path.join(out, f"rank{rank}.npz"), params=model._params.cpu().numpy(), losses=np.array(step_losses),
             state=np.concatenate(state) if state else np.zeros(0, np.float32), order=np.array(order, np.int64),
             early=np.int64(early))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

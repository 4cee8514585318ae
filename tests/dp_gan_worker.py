"""Child process of tests/test_gpu_dp.py: data-parallel forms of the training steps on the one GPU of the test box.

  python dp_gan_worker.py RANK WORLD PORT OUTDIR BACKEND SHARDS

BACKEND "none": no process group, no reducers (the single-process answer).  "nccl": RCCL with WORLD = 1 and the
collective forced, which exercises the all-reduce recorded INSIDE the step's HIP graph.  "gloo": WORLD ranks sharing the
GPU.  SHARDS "same": every rank trains on the same rows (the reduced gradient is then exactly the local one);
"split": rank r takes its own rows.

Each run does 5 steps of (a) the run_model.py auto-encoder through ``Model.train_on_device_batch``, (b) the context
encoder + discriminator step, (c) the U-Net generator + discriminator step, and writes the final weights, per-step
losses and which graph form was used to OUTDIR/rank<r>.npz.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CFG = dict(conv_layers=3, conv_layer_multiplier=2, initial_filters=8, activation="elu", kernel_size=3,
           conv_operations=1, fc_layers=0, fc_layer_size=256, dropout=0.2, latent_dimensions=0, batch_norm=False)
STEPS, B = 5, 16


def main():
    rank, world, port, out, backend, shards = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5],
                                               sys.argv[6])
    dp = backend != "none"
    if dp:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=port, SPECTRUM_DIST_BACKEND=backend)
        if world == 1:
            os.environ["SPECTRUM_DP_FORCE_COLLECTIVE"] = "1"
    import torch.distributed as dist
    from spectrum_vae_amd import autoencoder, context_encoder, losses, ops, parallel, run_context_encoder, run_unet, unet
    from spectrum_vae_amd.data import synth_training_tensor
    from spectrum_vae_amd.run_model import build_optimizer
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    if dp:
        if world == 1:  # init_from_env only sets a group up for WORLD_SIZE > 1
            dist.init_process_group(backend=backend, rank=0, world_size=1, init_method=f"tcp://127.0.0.1:{port}")
        else:
            parallel.init_from_env()
    row0 = rank * B if shards == "split" else 0
    X = synth_training_tensor(B * max(world, 2), 91)[row0:row0 + B]
    res = {}

    def attach(*models):
        for m in models:
            if dp:
                red = parallel.GradReducer()
                red.broadcast_(m._params)
                m.distribute(red)

    # (a) auto-encoder through Model.train_on_device_batch
    model = autoencoder.CAEPadded(**CFG)
    model.build((None, 32, 32, 3))
    model._materialize(device)
    model.compile(optimizer=build_optimizer(learning_rate=5e-4), loss=losses.MeanSquaredError())
    attach(model)
    b = model._step_buffers(B, (32, 32, 3))
    ops.gather_rows(torch.from_numpy(X).to(device), None, b["x"])
    la = []
    for _ in range(STEPS):
        model.train_on_device_batch(b, True, float(B))
        la.append(float(b["loss"].item()))
    g = next(iter(model._graphs.values()))
    res["ae_params"], res["ae_losses"], res["ae_graph_len"] = model._params.cpu().numpy(), np.array(la), len(g)
    if dp:
        assert model._reducer.calls == STEPS, model._reducer.calls
        res["capturable"] = int(model._reducer.capturable(device))

    # (b) context encoder + discriminator
    rng = np.random.default_rng(5)
    mask = run_context_encoder.batch_random_mask(X, 0.25, rng).astype(np.float32)
    ce, disc = context_encoder.ContextEncoder(), context_encoder.Discriminator()
    ce.build((None, 32, 32, 3)); disc.build((None, 32, 32, 3))
    ce._materialize(device); disc._materialize(device)
    attach(ce, disc)
    co, do = run_context_encoder.build_optimizer(learning_rate=2e-3), run_context_encoder.build_optimizer(learning_rate=2e-4)
    lb = []
    for _ in range(STEPS):
        lb.append([float(v) for v in run_context_encoder.train_step(ce, disc, co, do, X, mask, use_graph=dp)])
    res["ce_params"], res["ced_params"], res["ce_losses"] = ce._params.cpu().numpy(), disc._params.cpu().numpy(), np.array(lb)
    if dp:
        st = next(iter(ce._gan_steps.values()))
        res["ce_graph_len"] = len(st.dp_graphs)
        assert ce._reducer.calls == STEPS and disc._reducer.calls == STEPS

    # (c) U-Net generator + discriminator
    gen, ud = unet.Generator(), unet.Discriminator()
    Xg, yg = X[..., :1].copy(), X
    gen.build((None, 32, 32, 1)); ud.build((None, 32, 32, 4))
    gen._materialize(device); ud._materialize(device)
    attach(gen, ud)
    go, uo = run_unet.build_optimizer(2e-4), run_unet.build_optimizer(2e-4)
    lc = []
    for _ in range(STEPS):
        lc.append([float(v) for v in run_unet.train_step(gen, ud, go, uo, Xg, yg, use_graph=dp)])
    res["un_params"], res["und_params"], res["un_losses"] = gen._params.cpu().numpy(), ud._params.cpu().numpy(), np.array(lc)
    np.savez(os.path.join(out, f"rank{rank}.npz"), **res)
    if dp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

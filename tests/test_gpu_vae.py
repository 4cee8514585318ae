"""VAEPadded on the GPU against its float64 CPU twin (oracle.models.RefVAEPadded).

The model has no counterpart in the reference (which ships only the Sampling layer, autoencoder.py:264-272): the
twin restates the same construction from the reference's pieces, so these tests pin the HIP path to that definition.
Bars: loss, KL and every gradient within 1e-4 (relative L2); ELBO trajectory within 1e-3 over 100 fixed-seed steps with
replayed epsilon.
"""
import numpy as np
import pytest
import torch

from oracle import models as omodels
from oracle.torch_ref import KerasAdam

pytestmark = pytest.mark.gpu

CFG = dict(initial_filters=8, conv_layer_multiplier=2, conv_layers=3, fc_layers=2, fc_layer_size=64, activation="elu",
           kernel_size=3, conv_operations=1, latent_dimensions=8, beta=0.5)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def data(n, seed=1833):
    from spectrum_vae_amd.data import synth_training_tensor
    return synth_training_tensor(n, seed)


def build_pair(**over):
    from spectrum_vae_amd import autoencoder
    cfg = dict(CFG, **over)
    model = autoencoder.VAEPadded(**cfg)
    model.build((None, 32, 32, 3))
    model._materialize()
    ref = omodels.RefVAEPadded(**cfg)
    ref.load_params([v.numpy() for v in model.trainable_variables])
    return model, ref


def test_structure_matches_the_twin():
    model, ref = build_pair()
    tv = model.trainable_variables
    assert len(tv) == len(ref.params)
    for v, p in zip(tv, ref.params):
        assert tuple(v.shape) == tuple(p.shape), v.name
    assert model.count_params() == sum(p.numel() for p in ref.params)
    assert model.output_shape_ == (32, 32, 3)


@pytest.mark.parametrize("n", [1, 7, 32])
def test_loss_kl_and_gradients_match_the_twin(n):
    from spectrum_vae_amd import losses
    from spectrum_vae_amd.losses import to_device_f32
    model, ref = build_pair()
    x = data(n, seed=41 + n)
    eps = np.random.default_rng(n).standard_normal((n, 8))
    model.sampling.replay(eps)
    xt = to_device_f32(x)
    model.zero_grad()
    pred = model.forward_raw(xt, True)
    ctx = model._calls[-1]
    (kl, w), = ctx.extra_losses
    assert w == 0.5
    loss, dpred = torch.zeros(1, device="cuda"), torch.empty_like(pred)
    losses.MeanSquaredError().value_and_grad(xt, pred, loss, dpred, 1.0)
    model.backward_raw(dpred, -1, need_dx=False)
    xr, er = torch.from_numpy(x).double(), torch.from_numpy(eps)
    dec, kl_ref = ref(xr, er, training=True)
    total = torch.mean((dec - xr) ** 2) + ref.beta * kl_ref
    total.backward()
    assert rel(pred.cpu().numpy(), dec.detach().numpy()) < 1e-4
    assert abs(float(kl) - float(kl_ref)) <= 1e-4 * abs(float(kl_ref))
    assert abs(float(loss) + 0.5 * float(kl) - float(total)) <= 1e-4 * abs(float(total))
    for v, p in zip(model.trainable_variables, ref.params):
        assert rel(v.grad.cpu().numpy(), p.grad.numpy()) < 1e-4, v.name


def test_elbo_trajectory_100_steps_with_replayed_epsilon():
    """100 Adam steps (lr 1e-3, batch 32, fixed batch order, epsilon drawn on the host and replayed on both sides): the
    ELBO of every step within 1e-3 of the float64 twin, final weights within 1e-3.  Steps 3.. replay one HIP graph."""
    from spectrum_vae_amd import losses, ops
    from spectrum_vae_amd.optimizers import Adam
    model, ref = build_pair()
    X = data(32 * 10)
    model.compile(optimizer=Adam(1e-3), loss=losses.MeanSquaredError())
    opt_ref = KerasAdam(ref.params, 1e-3)
    Xd = torch.from_numpy(X).cuda()
    b = model._step_buffers(32, (32, 32, 3))
    b["acc"] = torch.zeros(1, device="cuda")
    rng = np.random.default_rng(7)
    hip, cpu = [], []
    for step in range(100):
        sl = slice((step % 10) * 32, (step % 10 + 1) * 32)
        eps = rng.standard_normal((32, 8))
        model.sampling.replay(eps)
        ops.gather_rows(Xd[sl], None, b["x"])
        model.train_on_device_batch(b, True, 32.0)
        hip.append(float(b["loss"].item()))
        for p in ref.params:
            p.grad = None
        loss = ref.loss(torch.from_numpy(X[sl]).double(), torch.from_numpy(eps))
        loss.backward()
        opt_ref.apply_gradients([p.grad for p in ref.params])
        cpu.append(float(loss.detach()))
    hip, cpu = np.array(hip), np.array(cpu)
    assert len(model._graphs) == 1 and not isinstance(next(iter(model._graphs.values())), str), "graph not captured"
    assert np.max(np.abs(hip - cpu) / np.abs(cpu)) <= 1e-3, (hip[-5:], cpu[-5:])
    assert cpu[-1] < cpu[0]
    assert abs(float(b["acc"]) - 32.0 * hip.sum()) <= 1e-4 * 32.0 * hip.sum()  # the epoch's running sum includes the KL
    for v, p in zip(model.trainable_variables, ref.params):
        assert rel(v.numpy(), p.detach().numpy()) <= 1e-3, v.name


def test_generator_draws_fresh_standard_normals_in_recorded_steps():
    """Without replayed epsilon the kernel draws from its counter-based generator; the stream position lives on the
    device, so every replay of the recorded step draws new values."""
    from spectrum_vae_amd import losses, ops
    from spectrum_vae_amd.optimizers import Adam
    model, _ = build_pair(latent_dimensions=64)
    model.compile(optimizer=Adam(1e-4), loss=losses.MeanSquaredError())
    X = torch.from_numpy(data(256)).cuda()
    b = model._step_buffers(256, (32, 32, 3))
    b["acc"] = torch.zeros(1, device="cuda")
    ops.gather_rows(X, None, b["x"])
    draws = []
    for step in range(5):
        model.train_on_device_batch(b, True, 256.0)
        eps = [t for k, t in model._pool.bufs.items() if k[0][0] == id(model.sampling) and k[0][1] == "eps"]
        assert len(eps) == 1
        draws.append(eps[0].cpu().numpy().copy())
        assert int(model.sampling._counter.item()) == (step + 1) * 256 * 64
    for i in range(5):
        for j in range(i):
            assert not np.array_equal(draws[i], draws[j])
    allv = np.concatenate([d.ravel() for d in draws])
    assert abs(allv.mean()) < 0.02 and abs(allv.std() - 1.0) < 0.02
    assert abs(np.mean(allv ** 3)) < 0.05 and abs(np.mean(allv ** 4) - 3.0) < 0.1


def test_reference_style_sampling_call_and_evaluate():
    """``Sampling()([z_mean, z_log_var])`` as the reference layer is called (autoencoder.py:267-272), and ``evaluate``
    reporting reconstruction + beta * KL."""
    from spectrum_vae_amd import autoencoder, losses
    from spectrum_vae_amd.optimizers import Adam
    s = autoencoder.Sampling()
    mu, lv = np.zeros((512, 16), np.float32), np.full((512, 16), np.log(4.0), np.float32)
    z1 = s([mu, lv]).numpy()
    z2 = s([mu, lv]).numpy()
    assert z1.shape == (512, 16) and not np.array_equal(z1, z2)
    assert abs(z1.std() - 2.0) < 0.1 and abs(z1.mean()) < 0.1
    want_kl = -0.5 * 16 * (1 + np.log(4.0) - 4.0)
    assert abs(float(s.last_kl) - want_kl) < 1e-4 * want_kl
    model, ref = build_pair()
    model.compile(optimizer=Adam(1e-3), loss=losses.MeanSquaredError())
    x = data(48, seed=3)
    eps = np.random.default_rng(1).standard_normal((48, 8))
    model.sampling.replay(eps)
    got = model.evaluate(x, batch_size=48)
    want = float(ref.loss(torch.from_numpy(x).double(), torch.from_numpy(eps), training=False))
    assert abs(got - want) <= 1e-4 * abs(want)
    mean, logvar = model.encode(x)
    assert tuple(mean.shape) == (48, 8) and tuple(logvar.shape) == (48, 8)

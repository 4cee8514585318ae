"""Points of the reference's hyper-parameter grid (run_model.py:164-219: initial filters {8, 16}, blocks {3, 4},
multiplier {1, 2}, convs per block {1, 2, 3}, kernel {3, 5}, BatchNorm {on, off}, relu / elu) through the training step
the drivers use (``train_on_device_batch``: eager, recorded, replayed) against the torch-CPU restatement with identical
weights: (1) one forward / backward pass - prediction, loss and every gradient at 1e-4; (2) 10 optimizer steps at batch
256 - every step's loss and the final weights at 1e-3.  The layers with 16..128 channels run on the matrix-core kernels of
conv_s1_mfma.hip / wgrad_s1_mfma.hip; the test also checks that they do."""
import numpy as np
import pytest
import torch

from oracle import models as omodels
from oracle import torch_ref
from oracle.torch_ref import KerasAdam, RefSequential

from test_gpu_models import _ref_pass, check_fwd_bwd, data, load_into_ref, rel

pytestmark = pytest.mark.gpu

BASE = dict(fc_layers=0, fc_layer_size=256, dropout=0.2, latent_dimensions=0)
POINTS = {
    "heaviest_i16_l4_m2_o3_k5": dict(initial_filters=16, conv_layers=4, conv_layer_multiplier=2, conv_operations=3,
                                     kernel_size=5, batch_norm=False, activation="elu"),
    "i16_l3_m2_o2_k3": dict(initial_filters=16, conv_layers=3, conv_layer_multiplier=2, conv_operations=2, kernel_size=3,
                            batch_norm=False, activation="elu"),
    "i8_l3_m1_o2_k3_bn": dict(initial_filters=8, conv_layers=3, conv_layer_multiplier=1, conv_operations=2, kernel_size=3,
                              batch_norm=True, activation="relu"),
    "i16_l3_m1_o1_k5_relu": dict(initial_filters=16, conv_layers=3, conv_layer_multiplier=1, conv_operations=1,
                                 kernel_size=5, batch_norm=False, activation="relu"),
    "i8_l4_m2_o1_k3": dict(initial_filters=8, conv_layers=4, conv_layer_multiplier=2, conv_operations=1, kernel_size=3,
                           batch_norm=False, activation="elu"),
    # thirteen few-channel layers: the 8 -> 8 input gradients of the channel-last family, issued as two chain launches
    "i8_l3_m1_o2_k3": dict(initial_filters=8, conv_layers=3, conv_layer_multiplier=1, conv_operations=2, kernel_size=3,
                           batch_norm=False, activation="elu"),
    "i16_l4_m2_o2_k3_bn": dict(initial_filters=16, conv_layers=4, conv_layer_multiplier=2, conv_operations=2,
                               kernel_size=3, batch_norm=True, activation="elu"),
}


def build(cfg):
    from spectrum_vae_amd import autoencoder
    full = dict(BASE, **cfg)
    model = autoencoder.CAEPadded(**full)
    model.build((None, 32, 32, 3))
    model._materialize()
    enc, dec = omodels.caepadded(**full)
    return model, RefSequential(enc + dec, (32, 32, 3)), full


@pytest.mark.parametrize("name", list(POINTS))
def test_forward_backward_matches_oracle(name):
    model, ref, full = build(POINTS[name])
    enc, dec = omodels.caepadded(**full)
    ref64 = RefSequential(enc + dec, (32, 32, 3), dtype=torch.float64) if full["batch_norm"] else None
    check_fwd_bwd(model, ref, data(6, seed=7), ref64=ref64)


@pytest.mark.parametrize("name", list(POINTS))
def test_ten_steps_at_batch_256_match_oracle(name):
    from spectrum_vae_amd import losses, ops
    from spectrum_vae_amd.optimizers import Adam, ExponentialDecay
    n = 256
    model, ref, full = build(POINTS[name])
    load_into_ref(ref, model)
    # the sweep's optimizer: lr 1e-4, decay 0.92 every 10 steps (run_model.py:77-89,211-219)
    model.compile(optimizer=Adam(ExponentialDecay(1e-4, decay_steps=10, decay_rate=0.92, staircase=True)),
                  loss=losses.MeanSquaredError())
    opt_ref = KerasAdam(ref.params, 1e-4, decay_steps=10, decay_rate=0.92, staircase=True)
    X = data(2 * n, seed=13)
    Xd = torch.from_numpy(X).cuda()
    b = model._step_buffers(n, (32, 32, 3))
    ops.fill(b["acc"], 0.0)
    hip, cpu = [], []
    for step in range(10):
        sl = slice((step % 2) * n, (step % 2 + 1) * n)
        ops.gather_rows(Xd[sl], None, b["x"])
        model.train_on_device_batch(b, True, float(n))
        hip.append(float(b["loss"].item()))
        xb = torch.from_numpy(X[sl])
        cpu.append(torch_ref.train_step_mse(ref, opt_ref, xb, xb))
    hip, cpu = np.array(hip), np.array(cpu)
    g = next(iter(model._graphs.values()))
    assert not isinstance(g, str), "the steps were not replayed from a recorded graph"
    assert np.max(np.abs(hip - cpu) / np.abs(cpu)) <= 1e-3, (hip, cpu)
    # Final weights: 1e-3 of each tensor's largest entry.  With BatchNorm some gradients are sums that cancel to round-off
    # (a bias in front of a normalisation), and Adam turns the SIGN of such a gradient into a step of the full learning
    # rate: there the two float32 evaluations may part by a fraction of the 10 * lr an entry can travel at all.
    travel = 10 * 1e-4
    for v, p in zip(model.trainable_variables, ref.params):
        want = p.detach().numpy()
        slack = 0.1 * travel if full["batch_norm"] else 0.0
        assert np.max(np.abs(v.numpy() - want)) <= 1e-3 * max(np.abs(want).max(), 1e-3) + slack, v.name


def test_wide_layers_take_the_matrix_core_kernels():
    """Which kernels the step of a grid point runs is a plan-time decision of the library; this pins it: every conv with
    16..128 channels on both sides reports the matrix-core backward, the 3- and 8-channel layers do not."""
    from spectrum_vae_amd import ops
    for cin, cout, w, k, up, want in [(16, 16, 32, 5, 1, True), (32, 64, 8, 3, 1, True), (128, 128, 4, 5, 2, True),
                                      (64, 32, 8, 5, 1, True), (3, 16, 32, 5, 1, False), (8, 8, 16, 3, 1, False),
                                      (16, 3, 32, 3, 1, False)]:
        g = ops.conv_geom(64, w, w, cin, k, k, 1, 1, "same", cout, (up, up))
        assert ops.conv2d_s1_mfma_bwd(g) == want, (cin, cout, w, k, up)
    with ops.option("s1_mfma", 0):
        g = ops.conv_geom(64, 8, 8, 32, 3, 3, 1, 1, "same", 64, (1, 1))
        assert not ops.conv2d_s1_mfma_bwd(g)

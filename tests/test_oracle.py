"""CPU: the oracle against itself (two independent restatements), against analytic known answers, and the
data-preparation restatements against golden vectors minted from the reference's own ``util`` module."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import dataprep, models, np_ops, torch_ref
from oracle.torch_ref import KerasAdam, RefSequential

GOLD = os.path.join(os.path.dirname(__file__), "golden", "dataprep_golden.npz")

CONV_CASES = [
    (3, 3, 1, 1, "same", 8, 8), (5, 5, 1, 1, "same", 8, 8), (1, 3, 1, 1, "valid", 4, 9), (2, 2, 2, 2, "same", 8, 8),
    (4, 4, 2, 2, "same", 8, 8), (4, 4, 1, 1, "valid", 4, 4), (1, 4, 1, 2, "same", 4, 8), (4, 4, 1, 1, "same", 6, 6),
    (3, 3, 2, 2, "same", 7, 9),
]


def t64(a):
    return torch.tensor(np.asarray(a), dtype=torch.float64)


# ------------------------------------------------------------------------------------------------ geometry (R1, R2, R4)
def test_same_padding_rules():
    assert np_ops.same_pads(32, 4, 1) == (32, 1, 2)  # 4x4 s1: 1 before, 2 after
    assert np_ops.same_pads(32, 4, 2) == (16, 1, 1)
    assert np_ops.same_pads(32, 2, 2) == (16, 0, 0)
    assert np_ops.same_pads(32, 3, 1) == (32, 1, 1)
    assert np_ops.same_pads(7, 3, 2) == (4, 1, 1)
    assert np_ops.valid_out(4, 4, 1) == 1
    assert np_ops.deconv_out_size(1, 4, 1, "valid") == 4
    assert np_ops.deconv_out_size(16, 4, 2, "same") == 32
    assert np_ops.deconv_out_size(6, 3, 1, "valid") == 8


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_numpy_vs_torch_and_autograd(case):
    kh, kw, sh, sw, pad, H, W = case
    rng = np.random.default_rng(sum(case[:4]) + H)
    x, w, b = rng.standard_normal((2, H, W, 3)), rng.standard_normal((kh, kw, 3, 5)), rng.standard_normal(5)
    y = np_ops.conv2d(x, w, b, (sh, sw), pad)
    xt, wt, bt = t64(x).requires_grad_(), t64(w).requires_grad_(), t64(b).requires_grad_()
    yt = torch_ref.conv2d_nchw(xt.permute(0, 3, 1, 2), wt, bt, (sh, sw), pad).permute(0, 2, 3, 1)
    assert np.abs(y - yt.detach().numpy()).max() < 1e-10
    dz = rng.standard_normal(y.shape)
    (yt * t64(dz)).sum().backward()
    dx, dw, db = np_ops.conv2d_grads(x, w, dz, (sh, sw), pad)
    assert np.abs(dx - xt.grad.numpy()).max() < 1e-10
    assert np.abs(dw - wt.grad.numpy()).max() < 1e-10
    assert np.abs(db - bt.grad.numpy()).max() < 1e-10


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_transpose_numpy_vs_torch_and_adjoint(case):
    kh, kw, sh, sw, pad, H, W = case
    rng = np.random.default_rng(sum(case[:4]) + W)
    x = rng.standard_normal((2, H, W, 3))
    w = rng.standard_normal((kh, kw, 5, 3))  # (kh, kw, Cout, Cin)
    b = rng.standard_normal(5)
    y = np_ops.conv2d_transpose(x, w, b, (sh, sw), pad)
    xt, wt = t64(x).requires_grad_(), t64(w).requires_grad_()
    yt = torch_ref.conv2d_transpose_nchw(xt.permute(0, 3, 1, 2), wt, t64(b), (sh, sw), pad).permute(0, 2, 3, 1)
    assert y.shape == tuple(yt.shape)
    assert np.abs(y - yt.detach().numpy()).max() < 1e-10
    dz = rng.standard_normal(y.shape)
    (yt * t64(dz)).sum().backward()
    dx, dw, _ = np_ops.conv2d_transpose_grads(x, w, dz, (sh, sw), pad)
    assert np.abs(dx - xt.grad.numpy()).max() < 1e-10
    assert np.abs(dw - wt.grad.numpy()).max() < 1e-10
    # transposed conv == input-gradient of the forward conv on the output extent (same kernel memory)
    gx, _, _ = np_ops.conv2d_grads(np.zeros(y.shape[:3] + (5,)), w, x, (sh, sw), pad)
    assert np.abs(gx - (y - b)).max() < 1e-10


def test_conv_known_answers():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 6, 6, 4))
    ident = np.zeros((3, 3, 4, 4))
    ident[1, 1] = np.eye(4)
    assert np.allclose(np_ops.conv2d(x, ident, None, (1, 1), "same"), x)
    # 2x2 stride 2 == space-to-depth GEMM
    w = rng.standard_normal((2, 2, 4, 3))
    y = np_ops.conv2d(x, w, None, (2, 2), "same")
    s2d = x.reshape(1, 3, 2, 3, 2, 4).transpose(0, 1, 3, 2, 4, 5).reshape(1, 3, 3, 16)
    assert np.allclose(y, s2d @ w.reshape(16, 3))
    # 4x4 valid on a 4x4 map == dense over the NHWC-flattened input
    x4 = rng.standard_normal((2, 4, 4, 4))
    w4 = rng.standard_normal((4, 4, 4, 7))
    assert np.allclose(np_ops.conv2d(x4, w4, None, (1, 1), "valid")[:, 0, 0], x4.reshape(2, -1) @ w4.reshape(-1, 7))
    # numerical gradient of one weight
    wv = rng.standard_normal((3, 3, 4, 2))
    dz = rng.standard_normal((1, 6, 6, 2))
    _, dw, _ = np_ops.conv2d_grads(x, wv, dz, (1, 1), "same")
    e = np.zeros_like(wv)
    e[2, 0, 1, 1] = 1e-6
    num = ((np_ops.conv2d(x, wv + e, None, (1, 1), "same") - np_ops.conv2d(x, wv - e, None, (1, 1), "same")) * dz).sum() / 2e-6
    assert abs(num - dw[2, 0, 1, 1]) < 1e-6


def test_pool_upsample_bn_activations():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 6, 8, 3))
    for pool, pad in (((2, 2), "same"), ((2, 2), "valid"), ((1, 2), "valid")):
        y, arg = np_ops.maxpool2d(x, pool, pad)
        xt = t64(x).requires_grad_()
        yt = torch_ref.maxpool_nchw(xt.permute(0, 3, 1, 2), pool, pad).permute(0, 2, 3, 1)
        assert np.array_equal(y, yt.detach().numpy())
        dy = rng.standard_normal(y.shape)
        (yt * t64(dy)).sum().backward()
        assert np.allclose(np_ops.maxpool2d_bwd(dy, arg, x.shape, pool, pad), xt.grad.numpy())
    # odd size with SAME pads with -inf (pooled value comes from real data only)
    y, _ = np_ops.maxpool2d(x[:, :5, :5], (2, 2), "same")
    assert y.shape == (2, 3, 3, 3) and np.isfinite(y).all()
    u = np_ops.upsample2d(x, (2, 2))
    assert np.array_equal(u, torch_ref.upsample_nchw(t64(x).permute(0, 3, 1, 2), (2, 2)).permute(0, 2, 3, 1).numpy())
    assert np.allclose(np_ops.upsample2d_bwd(u, (2, 2)), 4 * x)
    # BN of a constant tensor: output == beta, var == 0; moving var uses the unbiased estimate on 4-D inputs
    c = np.full((2, 3, 3, 2), 5.0)
    y, mean, var = np_ops.batchnorm_train(c, np.array([2.0, 3.0]), np.array([0.5, -1.0]))
    assert np.allclose(y, [0.5, -1.0]) and np.allclose(mean, 5.0) and np.allclose(var, 0)
    y, mean, var = np_ops.batchnorm_train(x, np.ones(3), np.zeros(3))
    mm, mv = np_ops.batchnorm_moving_update(np.zeros(3), np.ones(3), mean, var, 2 * 6 * 8)
    assert np.allclose(mm, 0.01 * mean) and np.allclose(mv, 0.99 + 0.01 * var * 96 / 95)
    # BN backward against autograd
    g, bt = rng.standard_normal(3), rng.standard_normal(3)
    xt = t64(x).requires_grad_()
    m = xt.mean(dim=(0, 1, 2))
    v = xt.var(dim=(0, 1, 2), unbiased=False)
    yt = (xt - m) / torch.sqrt(v + 1e-3) * t64(g) + t64(bt)
    dy = rng.standard_normal(x.shape)
    (yt * t64(dy)).sum().backward()
    dx, dg, db = np_ops.batchnorm_train_bwd(x, g, dy)
    assert np.allclose(dx, xt.grad.numpy()) and np.allclose(dg, (dy * ((x - x.mean((0, 1, 2))) / np.sqrt(x.var((0, 1, 2)) + 1e-3))).sum((0, 1, 2)))
    for act in ("relu", "elu", "lrelu", "tanh", "sigmoid", None):
        z = rng.standard_normal(100)
        zt = t64(z).requires_grad_()
        yt = torch_ref._act(zt, act, 0.2)
        yt.sum().backward()
        assert np.allclose(np_ops.act_fwd(z, act, 0.2), yt.detach().numpy())
        assert np.allclose(np_ops.act_bwd_from_output(yt.detach().numpy(), act, 0.2), zt.grad.numpy())


def test_losses_known_answers():
    rng = np.random.default_rng(2)
    t, p = rng.standard_normal((3, 4, 4, 3)), rng.standard_normal((3, 4, 4, 3))
    ones = np.ones((3, 4, 4))
    assert abs(np_ops.mse_weighted(t, p, ones * 10) - 10 * np_ops.mse(t, p)) < 1e-12
    m = np.zeros((3, 4, 4))
    m[:, 1:3, 1:3] = 1
    # denominator is the pixel count, not sum(mask)
    manual = (np.mean((p - t) ** 2, axis=-1) * m * 10).sum() / (3 * 4 * 4)
    assert abs(np_ops.mse_weighted(t, p, m * 10) - manual) < 1e-12
    for f, g, ft in ((np_ops.mse, np_ops.mse_grad, torch_ref.mse), (np_ops.mae, np_ops.mae_grad, torch_ref.mae)):
        pt = t64(p).requires_grad_()
        l = ft(t64(t), pt)
        l.backward()
        assert abs(f(t, p) - float(l)) < 1e-12 and np.allclose(g(t, p), pt.grad.numpy())
    pt = t64(p).requires_grad_()
    l = torch_ref.mse_weighted(t64(t), pt, t64(m * 10))
    l.backward()
    assert np.allclose(np_ops.mse_weighted_grad(t, p, m * 10), pt.grad.numpy())
    z = rng.standard_normal((16, 1)) * 3
    pr = 1 / (1 + np.exp(-z))
    for label in (0.0, 1.0):
        lab = np.full_like(z, label)
        zt, prt = t64(z).requires_grad_(), t64(pr).requires_grad_()
        l1, l2 = torch_ref.bce_logits(t64(lab), zt), torch_ref.bce_probs(t64(lab), prt)
        l1.backward(); l2.backward()
        assert abs(np_ops.bce_logits(lab, z) - float(l1)) < 1e-12 and abs(np_ops.bce_probs(lab, pr) - float(l2)) < 1e-12
        assert np.allclose(np_ops.bce_logits_grad(lab, z), zt.grad.numpy())
        assert np.allclose(np_ops.bce_probs_grad(lab, pr), prt.grad.numpy())
        # the two BCE forms agree up to the 1e-7 clip
        assert abs(np_ops.bce_logits(lab, z) - np_ops.bce_probs(lab, pr)) < 1e-4
    mu, lv = rng.standard_normal((4, 8)), rng.standard_normal((4, 8))
    assert abs(np_ops.kl_standard_normal(mu, lv) - float(torch_ref.kl_standard_normal(t64(mu), t64(lv)))) < 1e-12
    assert np_ops.kl_standard_normal(np.zeros((4, 8)), np.zeros((4, 8))) == 0


def test_keras_adam_and_schedule():
    # staircase schedule: constant inside each window of decay_steps updates
    lrs = [np_ops.exponential_decay(5e-4, s, 10, 0.96, True) for s in range(25)]
    assert lrs[0] == lrs[9] == 5e-4 and math.isclose(lrs[10], 5e-4 * 0.96) and math.isclose(lrs[20], 5e-4 * 0.96 ** 2)
    # first step closed form: delta = -lr * g / (|g| + eps / sqrt(1 - b2))
    g = np.array([1.0, -2.0, 1e-4])
    th, m, v = np_ops.adam_keras_step(np.zeros(3), g, np.zeros(3), np.zeros(3), 0, 1e-3)
    assert np.allclose(th, -1e-3 * g / (np.abs(g) + 1e-7 / math.sqrt(1 - 0.999)), rtol=1e-9)
    # torch-side KerasAdam == numpy-side step over several updates
    rng = np.random.default_rng(3)
    p = torch.tensor(rng.standard_normal(7))
    opt = KerasAdam([p], 1e-2, beta1=0.5, decay_steps=3, decay_rate=0.9)
    th, m, v = p.numpy().copy(), np.zeros(7), np.zeros(7)
    for it in range(8):
        gr = rng.standard_normal(7)
        opt.apply_gradients([torch.tensor(gr)])
        th, m, v = np_ops.adam_keras_step(th, gr, m, v, it, np_ops.exponential_decay(1e-2, it, 3, 0.9), beta1=0.5)
    assert np.allclose(th, p.numpy(), rtol=1e-10)
    # not torch's Adam: epsilon sits outside the bias correction
    q = torch.nn.Parameter(torch.zeros(3, dtype=torch.float64))
    q.grad = torch.tensor(g)
    torch.optim.Adam([q], lr=1e-3, eps=1e-7).step()
    assert not np.allclose(q.detach().numpy()[2], th[0] * 0 + (-1e-3 * g / (np.abs(g) + 1e-7 / math.sqrt(1 - 0.999)))[2], rtol=1e-6)


def test_model_shape_tables_from_the_reference():
    """Layer tables derivable from the reference files (they stand in for .summary() output)."""
    cfg = dict(conv_layers=3, conv_layer_multiplier=2, initial_filters=8, activation="elu", kernel_size=3,
               conv_operations=1, fc_layers=0, latent_dimensions=0)
    enc, dec = models.caepadded(**cfg)
    m = RefSequential(enc + dec, (32, 32, 3))
    assert m.num_params() == 12_035
    convs = [s for s, l in zip(m.out_shapes, m.layers) if l["type"] == "conv2d"]
    assert convs == [(32, 32, 8), (16, 16, 16), (8, 8, 32), (8, 8, 16), (16, 16, 8), (32, 32, 3)]
    assert [p.numel() for p in m.params] == [216, 8, 1152, 16, 4608, 32, 4608, 16, 1152, 8, 216, 3]
    assert m.layers[-1]["activation"] == "tanh"
    heavy = models.caepadded(conv_layers=4, conv_layer_multiplier=2, initial_filters=16, kernel_size=5,
                             conv_operations=3, fc_layers=0, latent_dimensions=0)
    assert RefSequential(heavy[0] + heavy[1], (32, 32, 3)).num_params() == 2_717_315
    assert models.FILTER_SIZES == [64, 64, 128]
    e, d = models.context_encoder()
    ce = RefSequential(e + d, (32, 32, 3))
    assert ce.num_params() == 4_497_899
    assert [s for s, l in zip(ce.out_shapes, ce.layers) if l["type"].startswith("conv")] == [
        (16, 16, 64), (8, 8, 64), (4, 4, 128), (1, 1, 1000), (4, 4, 128), (8, 8, 64), (16, 16, 64), (32, 32, 3)]
    assert RefSequential(models.discriminator(), (32, 32, 3)).num_params() == 202_497
    u = models.RefUNet()
    assert u.num_params() == 133_379 and u.output_shape == (32, 32, 3)
    y = u(torch.zeros(2, 32, 32, 1))
    assert tuple(y.shape) == (2, 32, 32, 3)


def test_oracle_training_reduces_loss():
    enc, dec = models.caepadded(conv_layers=3, conv_layer_multiplier=2, initial_filters=8, kernel_size=3,
                                conv_operations=1, fc_layers=0, latent_dimensions=0)
    m = RefSequential(enc + dec, (32, 32, 3))
    opt = KerasAdam(m.params, 5e-3, decay_steps=10, decay_rate=0.96)
    x = torch.from_numpy(dataprep.synth_training_tensor(16))
    losses = [torch_ref.train_step_mse(m, opt, x, x) for _ in range(15)]
    assert losses[-1] < losses[0] and opt.it == 15


# ------------------------------------------------------------------------------------------------ golden vectors
def test_dataprep_oracle_matches_reference_golden_vectors():
    g = np.load(GOLD)
    for ci in range(int(g["n_cases"])):
        X, pos = g[f"X_{ci}"], g[f"pos_{ci}"]
        region = dataprep.process_region(X.copy(), pos.copy())
        assert region.dtype == np.float32 and np.array_equal(region, g[f"region_{ci}"]), ci
        assert np.array_equal(dataprep.inter_snp_distances(pos), g[f"dist_{ci}"]), ci
        hap = np.sum(np.transpose(X, (1, 0, 2)), axis=2)
        assert np.array_equal(dataprep.major_minor(hap), g[f"majmin_{ci}"]), ci
        # sort_min_diff: same anchor row, same distance sequence (ties are ordered by sklearn internals there)
        ref_perm = g[f"sort_{ci}"]
        mine = dataprep.sort_min_diff(g[f"region_{ci}"][:, :, 0])
        D = dataprep.manhattan_rows(g[f"region_{ci}"][:, :, 0])
        anchor = int(np.argmin(D.sum(axis=1)))
        assert D[anchor, ref_perm[0]] == 0 and D[anchor, mine[0]] == 0
        assert sorted(mine.tolist()) == list(range(32))
        assert np.array_equal(D[anchor][mine], D[anchor][ref_perm]), ci


def test_training_tensor_format():
    x = dataprep.synth_training_tensor(8)
    assert x.shape == (8, 32, 32, 3) and x.dtype == np.float32
    assert set(np.unique(x[..., 0])) <= {-1.0, 0.0, 1.0}
    assert (x[..., 1] == 0).all()
    assert x[..., 2].min() >= -1 and x[..., 2].max() <= 1
    assert (x[..., 2] == x[:, :1, :, 2]).all()  # distances tiled down the haplotypes
    assert ((x[..., 0] == 1).sum(axis=1) <= 16).all()  # +1 is the minor allele


def test_vae_twin_closed_forms():
    """The variational twin (no reference model exists for it): its KL against the closed form for constant heads, the
    Sampling formula of autoencoder.py:272, and dELBO/d(z_mean bias) against a central difference."""
    from oracle.models import RefVAEPadded
    torch.manual_seed(0)
    vae = RefVAEPadded(conv_operations=1, latent_dimensions=4, beta=0.7)
    x = torch.rand(3, 32, 32, 3, dtype=torch.float64)
    eps = torch.randn(3, 4, dtype=torch.float64)
    with torch.no_grad():  # heads that ignore their input: mean = 0.3, logvar = -0.4
        for part, c in ((vae.z_mean, 0.3), (vae.z_log_var, -0.4)):
            part.params[0].zero_()
            part.params[1].fill_(c)
    dec, kl = vae(x, eps)
    want_kl = -0.5 * 4 * (1 - 0.4 - 0.09 - np.exp(-0.4))
    assert abs(float(kl.detach()) - want_kl) < 1e-12
    z = 0.3 + np.exp(-0.2) * eps
    assert torch.allclose(dec, vae.decoder(z), atol=1e-14)
    loss = vae.loss(x, eps)
    loss.backward()
    b = vae.z_mean.params[1]
    g = b.grad[1].item()
    with torch.no_grad():
        b[1] += 1e-6
        up = float(vae.loss(x, eps))
        b[1] -= 2e-6
        dn = float(vae.loss(x, eps))
    assert abs((up - dn) / 2e-6 - g) < 1e-6 * max(1.0, abs(g))

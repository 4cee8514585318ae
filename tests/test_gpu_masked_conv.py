"""The masked input of the context encoder folded into its first convolution (run_context_encoder.py:242-243:
X_masked = X * (1 - mask); pred = context_encoder(X_masked)): sv_conv2d_fwd_masked / sv_conv2d_wgrad_masked through the
C ABI against the unfolded pair (sv_mask_input, then the plain conv entry points) - bit for bit, the arithmetic is the
same - and against float64; then the model-level switch."""
import numpy as np
import pytest
import torch

from oracle import np_ops

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


# (N, H, W, Cin, K, stride, padding, Cout): the context encoder's first layer, a 4-channel variant (16-byte gathers), the
# U-Net's 2x2 stride-2 block on 8 channels, a VALID conv, a ragged batch
CASES = [(5, 32, 32, 3, 4, 2, "same", 64), (3, 16, 16, 4, 4, 2, "same", 32), (7, 16, 16, 8, 2, 2, "same", 16),
         (2, 12, 12, 3, 4, 1, "valid", 8), (64, 32, 32, 3, 4, 2, "same", 64)]


@pytest.mark.parametrize("case", CASES)
def test_masked_conv_equals_mask_then_conv(case):
    from spectrum_vae_amd import ops
    n, h, w, cin, k, s, pad, cout = case
    rng = np.random.default_rng(n * 100 + cin)
    g = ops.conv_geom(n, h, w, cin, k, k, s, s, pad, cout)
    assert ops.conv2d_masked_supported(g)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    mask = np.repeat((rng.random((n, h, w, 1)) < 0.3).astype(np.float32), cin, axis=-1)
    wt = (rng.standard_normal((k, k, cin, cout)) * 0.2).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    xd, md, wd, bd = dev(x), dev(mask), dev(wt), dev(b)
    y = torch.empty((n, g.Ho, g.Wo, cout), device="cuda")
    y2 = torch.empty_like(y)
    xm = torch.empty_like(xd)
    ops.mask_input(xd, md, xm)
    ops.conv2d_fwd(g, xm, wd, bd, y2, "lrelu", 0.2)
    ops.conv2d_fwd_masked(g, xd, md, wd, bd, y, "lrelu", 0.2)
    assert torch.equal(y, y2)
    ref = np_ops.conv2d(x.astype(np.float64) * (1 - mask), wt.astype(np.float64), b.astype(np.float64), (s, s), pad)
    ref = np.where(ref >= 0, ref, 0.2 * ref)
    assert float(np.abs(y.cpu().numpy() - ref).max()) <= 1e-4 * max(float(np.abs(ref).max()), 1.0)
    dz = dev(rng.standard_normal((n, g.Ho, g.Wo, cout)))
    ws = torch.empty(max(ops.conv2d_wgrad_workspace(g), 16), dtype=torch.uint8, device="cuda")
    dw, dw2 = torch.zeros_like(wd), torch.zeros_like(wd)
    ops.conv2d_wgrad(g, xm, dz, dw2, True, ws)
    ops.conv2d_wgrad_masked(g, xd, md, dz, dw, True, ws)
    assert torch.equal(dw, dw2)


def test_masked_entry_points_refuse_other_geometries():
    from spectrum_vae_amd import _lib, ops
    g = ops.conv_geom(2, 8, 8, 3, 3, 3, 1, 1, "same", 8)  # the LDS-tiled 3x3 family
    assert not ops.conv2d_masked_supported(g)
    x = torch.zeros((2, 8, 8, 3), device="cuda")
    with pytest.raises(_lib.SpectrumHipError):
        ops.conv2d_fwd_masked(g, x, x, torch.zeros((3, 3, 3, 8), device="cuda"), None, torch.zeros((2, 8, 8, 8), device="cuda"),
                              "relu", 0.0)


def test_model_forward_and_gradients_with_a_folded_mask_equal_the_explicit_product():
    """context_encoder.ContextEncoder on (X, mask): forward_raw(input_mask=...) against forward_raw(X * (1 - mask));
    outputs and every parameter gradient agree bit for bit (the first conv is the only layer that differs)."""
    from spectrum_vae_amd import context_encoder, ops
    rng = np.random.default_rng(9)
    n = 6
    x = dev(rng.standard_normal((n, 32, 32, 3)))
    mask = dev(np.repeat((rng.random((n, 32, 32, 1)) < 0.25).astype(np.float32), 3, axis=-1))
    dy = dev(rng.standard_normal((n, 32, 32, 3)))
    m = context_encoder.ContextEncoder()
    m.build_graph((1, 32, 32, 3))
    m._materialize(torch.device("cuda"))
    xm = torch.empty_like(x)
    ops.mask_input(x, mask, xm)
    m.zero_grad()
    ref = m.forward_raw(xm, True).clone()
    m.backward_raw(dy.clone())
    gref = m._grads.clone()
    m.zero_grad()
    got = m.forward_raw(x, True, input_mask=mask).clone()
    m.backward_raw(dy.clone())
    assert torch.equal(got, ref)
    assert torch.equal(m._grads, gref)
    first = m.encoder.layers[0] if type(m.encoder.layers[0]).__name__ == "Conv2D" else m.encoder.layers[1]
    assert first.masked_ok(first.geom(n)) and m._calls[-1].saved[id(first)]["mask"] is mask  # the folded form ran


def test_a_model_whose_first_layer_is_not_a_conv_takes_the_elementwise_pass():
    from spectrum_vae_amd import ops
    from spectrum_vae_amd.model import SequentialModel
    from spectrum_vae_amd.nn import Dense, Flatten
    rng = np.random.default_rng(2)
    x = dev(rng.standard_normal((4, 8, 8, 3)))
    mask = dev(np.repeat((rng.random((4, 8, 8, 1)) < 0.5).astype(np.float32), 3, axis=-1))
    m = SequentialModel([Flatten(), Dense(5)])
    m.build((None, 8, 8, 3))
    m._materialize(torch.device("cuda"))
    xm = torch.empty_like(x)
    ops.mask_input(x, mask, xm)
    ref = m.forward_raw(xm, False).clone()
    assert torch.equal(m.forward_raw(x, False, input_mask=mask), ref)


def test_the_context_encoder_step_launches_no_masking_pass(monkeypatch):
    """run_context_encoder.train_step under the float32 policy: X * (1 - mask) lives inside the first conv's gathers, so
    the elementwise masking entry point is never called while the step is built and recorded."""
    from spectrum_vae_amd import context_encoder, ops, run_context_encoder
    rng = np.random.default_rng(4)
    n = 16
    ce = context_encoder.ContextEncoder()
    ce.build_graph((1, 32, 32, 3))
    disc = context_encoder.Discriminator()
    disc.build_graph((1, 32, 32, 3))
    ce_opt, disc_opt = run_context_encoder.build_optimizer(2e-3), run_context_encoder.build_optimizer(2e-4)
    x = dev(rng.standard_normal((n, 32, 32, 3)))
    mask = run_context_encoder.batch_random_mask_device((n, 32, 32, 3), 0.25, seed=7)
    calls = []
    real = ops.mask_input
    monkeypatch.setattr(ops, "mask_input", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    for _ in range(3):  # eager, recording, replay
        d, l2, tot = run_context_encoder.train_step(ce, disc, ce_opt, disc_opt, x, mask)
    assert np.isfinite([float(d), float(l2), float(tot)]).all()
    assert calls == []

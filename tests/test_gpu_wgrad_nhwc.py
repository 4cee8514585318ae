"""The channel-last matrix-core kernel-gradient launch (sv_conv2d_wgrad_batch with reduce = 1 on the six layer shapes of
the CAEPadded run_model.py defaults, autoencoder.py:437-526) against a float64 evaluation of
dw[kh, kw, ci, co] = sum x[p + (kh - 1, kw - 1), ci] * dz[p, co], db[co] = sum dz[p, co] through the C ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# (stored h, stored w, Cin, Cout, upsampling of the input, pooled output, activation)
LAYERS = [(32, 32, 3, 8, 1, True, "elu"), (16, 16, 8, 16, 1, True, "elu"), (8, 8, 16, 32, 1, True, "elu"),
          (4, 4, 32, 16, 2, False, "elu"), (8, 8, 16, 8, 2, False, "elu"), (16, 16, 8, 3, 2, False, "tanh")]


def act_grad(y, act):
    if act == "elu":
        return np.where(y > 0, 1.0, y + 1.0)
    if act == "relu":
        return (y > 0).astype(np.float64)
    if act == "tanh":
        return 1.0 - y * y
    return np.ones_like(y)


def reference(x, dy, y, idx, up, act):
    """float64 dw, db from the stored tensors."""
    x = x.astype(np.float64)
    if up == 2:
        x = x.repeat(2, axis=1).repeat(2, axis=2)
    g = dy.astype(np.float64) * act_grad(y.astype(np.float64), act)
    if idx is not None:
        n, ph, pw, c = g.shape
        dz = np.zeros((n, 2 * ph, 2 * pw, c))
        for s in range(4):
            dz[:, (s >> 1)::2, (s & 1)::2, :] = np.where(idx == s, g, 0.0)
    else:
        dz = g
    n, H, W, cin = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    dw = np.empty((3, 3, cin, dz.shape[-1]))
    for kh in range(3):
        for kw in range(3):
            dw[kh, kw] = np.tensordot(xp[:, kh:kh + H, kw:kw + W, :], dz, axes=([0, 1, 2], [0, 1, 2]))
    return dw, dz.sum(axis=(0, 1, 2))


def make_jobs(n, rng, layers=LAYERS, blocks=0):
    from spectrum_vae_amd import ops
    jobs, refs = [], []
    for h, w, cin, cout, up, pooled, act in layers:
        H, W = h * up, w * up
        g = ops.conv_geom(n, H, W, cin, 3, 3, 1, 1, "same", cout, (up, up))
        oh, ow = (H // 2, W // 2) if pooled else (H, W)
        x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
        y = rng.standard_normal((n, oh, ow, cout)).astype(np.float32)
        if act == "tanh":
            y = np.tanh(y)
        dy = rng.standard_normal((n, oh, ow, cout)).astype(np.float32)
        idx = rng.integers(0, 4, (n, oh, ow, cout)).astype(np.uint8) if pooled else None
        ws = torch.empty(max(ops.conv2d_wgrad_workspace(g), 16), dtype=torch.uint8, device="cuda")
        jobs.append(dict(g=g, x=torch.from_numpy(x).cuda(), dy=torch.from_numpy(dy).cuda(), y=torch.from_numpy(y).cuda(),
                         idx=torch.from_numpy(idx).cuda() if pooled else None, act=act, alpha=1.0,
                         dw=torch.zeros(3, 3, cin, cout, device="cuda"), db=torch.zeros(cout, device="cuda"), ws=ws,
                         blocks=blocks))
        refs.append(reference(x, dy, y, idx, up, act))
    return jobs, refs


def rel_err(got, want):
    return float(np.abs(got.cpu().numpy().astype(np.float64) - want).max() / max(np.abs(want).max(), 1e-30))


@pytest.mark.parametrize("n,blocks", [(1, 0), (3, 0), (7, 2), (130, 0), (130, 16), (1024, 128), (1024, 256), (1027, 64)])
def test_matches_float64(n, blocks, monkeypatch):
    from spectrum_vae_amd import ops
    monkeypatch.delenv("SPECTRUM_WGRAD_NHWC", raising=False)
    rng = np.random.default_rng(n * 31 + blocks)
    jobs, refs = make_jobs(n, rng, blocks=blocks)
    ops.conv2d_wgrad_batch(jobs, reduce=True)
    for j, (dw, db), spec in zip(jobs, refs, LAYERS):
        assert rel_err(j["dw"], dw) < 2e-5, (spec, rel_err(j["dw"], dw))
        assert rel_err(j["db"], db) < 2e-5, (spec, rel_err(j["db"], db))
    # a second call adds the same sums to what is there; from zero it repeats the first call bit for bit
    first = [(j["dw"].clone(), j["db"].clone()) for j in jobs]
    ops.conv2d_wgrad_batch(jobs, reduce=True)
    for j, (dw1, db1) in zip(jobs, first):
        assert float((j["dw"] - 2 * dw1).abs().max()) <= 1e-6 * float(dw1.abs().max())
        assert float((j["db"] - 2 * db1).abs().max()) <= 1e-6 * float(db1.abs().max())
        j["dw"].zero_(), j["db"].zero_()
    ops.conv2d_wgrad_batch(jobs, reduce=True)
    for j, (dw1, db1) in zip(jobs, first):
        assert torch.equal(j["dw"], dw1) and torch.equal(j["db"], db1)


def test_agrees_with_the_planar_kernels(monkeypatch):
    """SPECTRUM_WGRAD_NHWC=0 selects the planar (vector / matrix-core) bodies for the same call."""
    from spectrum_vae_amd import ops
    rng = np.random.default_rng(77)
    jobs, _ = make_jobs(37, rng)
    monkeypatch.delenv("SPECTRUM_WGRAD_NHWC", raising=False)
    ops.conv2d_wgrad_batch(jobs, reduce=True)
    new = [(j["dw"].clone(), j["db"].clone()) for j in jobs]
    for j in jobs:
        j["dw"].zero_(), j["db"].zero_()
    monkeypatch.setenv("SPECTRUM_WGRAD_NHWC", "0")
    ops.conv2d_wgrad_batch(jobs, reduce=True)
    for j, (dw1, db1) in zip(jobs, new):
        assert float((j["dw"] - dw1).abs().max()) <= 1e-5 * float(dw1.abs().max())
        assert float((j["db"] - db1).abs().max()) <= 1e-5 * float(db1.abs().max())


def test_other_shapes_take_the_planar_kernels():
    """A batch holding a layer outside the family (here 8 -> 8 channels) still gives the right sums."""
    from spectrum_vae_amd import ops
    rng = np.random.default_rng(3)
    layers = [(16, 16, 8, 8, 1, True, "relu"), (16, 16, 8, 16, 1, True, "elu")]
    jobs, refs = make_jobs(9, rng, layers=layers)
    ops.conv2d_wgrad_batch(jobs, reduce=True)
    for j, (dw, db) in zip(jobs, refs):
        assert rel_err(j["dw"], dw) < 2e-5 and rel_err(j["db"], db) < 2e-5

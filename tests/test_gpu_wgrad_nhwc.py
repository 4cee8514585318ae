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
def test_matches_float64(n, blocks):
    from spectrum_vae_amd import ops
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


def test_agrees_with_the_planar_kernels(sv_opts):
    """Option wgrad_nhwc = 0 selects the planar (vector / matrix-core) bodies for the same call."""
    from spectrum_vae_amd import ops
    rng = np.random.default_rng(77)
    jobs, _ = make_jobs(37, rng)
    ops.conv2d_wgrad_batch(jobs, reduce=True)
    new = [(j["dw"].clone(), j["db"].clone()) for j in jobs]
    for j in jobs:
        j["dw"].zero_(), j["db"].zero_()
    sv_opts("wgrad_nhwc", 0)
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


@pytest.mark.parametrize("n,blocks,extra", [(5, 0, 0), (130, 16, 24), (1024, 128, 0)])
def test_optimizer_launch_that_sums_the_slabs_equals_the_two_launch_route(n, blocks, extra):
    """sv_conv2d_wgrad_batch_slabs leaves the partial sums in the workspaces and sv_adam_keras_step_slabs forms each
    gradient as grad + sum of slabs on its way into the Adam update; sv_conv2d_wgrad_batch(reduce = 1) followed by
    sv_adam_keras_step is the same arithmetic in the same order: weights, moments, the cleared gradient buffer and the
    update count agree bit for bit, over two steps, with a non-zero gradient already in the buffer and with `extra`
    elements that no slab covers (other variables, alignment gaps)."""
    from spectrum_vae_amd import ops
    rng = np.random.default_rng(n + blocks)
    jobs, _ = make_jobs(n, rng, blocks=blocks)
    # lay the variables out in one flat buffer the way a model does (8-float alignment), `extra` uncovered floats in front
    offs, at = [], extra
    for j in jobs:
        nw, nb = j["dw"].numel(), j["db"].numel()
        offs.append((at, at + -(-nw // 8) * 8))
        at += -(-nw // 8) * 8 + -(-nb // 8) * 8
    total = at

    def run(slabs_route):
        torch.manual_seed(5)
        theta = torch.randn(total, device="cuda")
        grad = torch.randn(total, device="cuda") * 0.01  # something is already there: both routes add to it
        m, v = torch.zeros(total, device="cuda"), torch.zeros(total, device="cuda")
        state = torch.zeros(3, dtype=torch.int64, device="cuda")
        js = []
        for j, (wo, bo) in zip(jobs, offs):
            jj = dict(j)
            jj["dw"] = grad[wo:wo + j["dw"].numel()].view(j["dw"].shape)
            jj["db"] = grad[bo:bo + j["db"].numel()]
            js.append(jj)
        for _ in range(2):
            if slabs_route:
                assert ops.conv2d_wgrad_batch_slabs_supported(js)
                table = ops.conv2d_wgrad_batch_slabs(js)
                ops.adam_step(theta, grad, m, v, state, 1e-3, 0.9, 0.999, 1e-7, 10, 0.96, True, 1.0, True, None, None, table)
            else:
                ops.conv2d_wgrad_batch(js, reduce=True)
                ops.adam_step(theta, grad, m, v, state, 1e-3, 0.9, 0.999, 1e-7, 10, 0.96, True, 1.0, True, None, None)
        torch.cuda.synchronize()
        return theta, grad, m, v, state

    a, b = run(True), run(False)
    for x, y, name in zip(a, b, ("theta", "grad", "m", "v", "state")):
        assert torch.equal(x, y), name
    assert int(a[4][0]) == 2 and float(a[1].abs().max()) == 0.0 and float(a[2].abs().max()) > 0


def test_slabs_form_refuses_shapes_outside_the_family():
    from spectrum_vae_amd import _lib, ops
    rng = np.random.default_rng(4)
    jobs, _ = make_jobs(9, rng, layers=[(16, 16, 8, 8, 1, True, "relu")])
    assert not ops.conv2d_wgrad_batch_slabs_supported(jobs)
    with pytest.raises(_lib.SpectrumHipError, match="outside"):
        ops.conv2d_wgrad_batch_slabs(jobs)

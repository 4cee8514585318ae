"""The channel-last matrix-core input-gradient kernel (the dx half of sv_conv2d_bwd on the layer shapes of the CAEPadded
run_model.py defaults, autoencoder.py:437-526) against a float64 evaluation of
dx[p, ci] = sum dz[p - (kh - 1, kw - 1), co] * w[kh, kw, ci, co] (summed over 2x2 windows where the conv read its input
through nearest upsampling), through the C ABI."""
import numpy as np
import pytest
import torch

from test_gpu_wgrad_nhwc import act_grad

pytestmark = pytest.mark.gpu

# (stored h, stored w, Cin, Cout, upsampling of the input, pooled output, activation): every layer but the first
LAYERS = [(16, 16, 8, 16, 1, True, "elu"), (8, 8, 16, 32, 1, True, "elu"), (4, 4, 32, 16, 2, False, "elu"),
          (8, 8, 16, 8, 2, False, "elu"), (16, 16, 8, 3, 2, False, "tanh")]


def reference(w, dy, y, idx, up, act):
    g = dy.astype(np.float64) * act_grad(y.astype(np.float64), act)
    if idx is not None:
        n, ph, pw, c = g.shape
        dz = np.zeros((n, 2 * ph, 2 * pw, c))
        for s in range(4):
            dz[:, (s >> 1)::2, (s & 1)::2, :] = np.where(idx == s, g, 0.0)
    else:
        dz = g
    n, H, W, _ = dz.shape
    zp = np.pad(dz, ((0, 0), (1, 1), (1, 1), (0, 0)))
    dx = np.zeros((n, H, W, w.shape[2]))
    for kh in range(3):
        for kw in range(3):
            # x[p + (kh - 1, kw - 1)] feeds z[p]: dx[q] += dz[q - (kh - 1, kw - 1)] . w[kh, kw]
            dx += zp[:, 2 - kh:2 - kh + H, 2 - kw:2 - kw + W, :] @ w[kh, kw].astype(np.float64).T
    if up == 2:
        dx = dx.reshape(n, H // 2, 2, W // 2, 2, -1).sum(axis=(2, 4))
    return dx


# the 8 -> 8 channel layers of the run_model.py:164-219 grid points with conv_layer_multiplier 1 / conv_operations > 1, at the
# three image widths, plain / pooled / behind an upsampling
LAYERS_8TO8 = [(32, 32, 8, 8, 1, True, "elu"), (32, 32, 8, 8, 1, False, "elu"), (16, 16, 8, 8, 2, False, "elu"),
               (16, 16, 8, 8, 1, True, "relu"), (16, 16, 8, 8, 1, False, "elu"), (8, 8, 8, 8, 2, False, "elu"),
               (8, 8, 8, 8, 1, True, "elu"), (8, 8, 8, 8, 1, False, "elu"), (4, 4, 8, 8, 2, False, "elu")]


def _ids(specs):
    return [f"{s[2]}to{s[3]}at{s[0] * s[4]}{'p' if s[5] else ''}{'u' if s[4] == 2 else ''}" for s in specs]


@pytest.mark.parametrize("n", [1, 3, 7, 130, 1024, 1027])
@pytest.mark.parametrize("spec", LAYERS + LAYERS_8TO8, ids=_ids(LAYERS + LAYERS_8TO8))
def test_matches_float64(spec, n, sv_opts):
    from spectrum_vae_amd import ops
    h, w_, cin, cout, up, pooled, act = spec
    H, W = h * up, w_ * up
    rng = np.random.default_rng(n + cin * 7)
    g = ops.conv_geom(n, H, W, cin, 3, 3, 1, 1, "same", cout, (up, up))
    oh, ow = (H // 2, W // 2) if pooled else (H, W)
    wt = rng.standard_normal((3, 3, cin, cout)).astype(np.float32)
    y = rng.standard_normal((n, oh, ow, cout)).astype(np.float32)
    if act == "tanh":
        y = np.tanh(y)
    dy = rng.standard_normal((n, oh, ow, cout)).astype(np.float32)
    idx = rng.integers(0, 4, (n, oh, ow, cout)).astype(np.uint8) if pooled else None
    dx = torch.full((n, h, w_, cin), 7.0, device="cuda")
    args = (g, None, torch.from_numpy(wt).cuda(), torch.from_numpy(dy).cuda(), torch.from_numpy(y).cuda(),
            torch.from_numpy(idx).cuda() if pooled else None, dx, None, None, act, 1.0, None)
    job = dict(g=g, w=args[2], dy=args[3], y=args[4], idx=args[5], dx=dx, act=act, alpha=1.0)
    assert ops.conv2d_dgrad_chain_supported([job, job]), "the layer is in the channel-last matrix-core family"
    ops.conv2d_bwd(*args)
    want = reference(wt, dy, y, idx, up, act)
    err = float(np.abs(dx.cpu().numpy() - want).max() / np.abs(want).max())
    assert err < 2e-6, err
    first = dx.clone()
    dx.fill_(-3.0)
    ops.conv2d_bwd(*args)
    assert torch.equal(dx, first)  # overwrites, bit-reproducible
    sv_opts("dgrad_nhwc", 0)  # the planar kernels on the same call
    ops.conv2d_bwd(*args)
    assert float((dx - first).abs().max()) <= 4e-6 * float(first.abs().max())


@pytest.mark.parametrize("n", [1, 5, 130, 1027])
@pytest.mark.parametrize("spec", [LAYERS[3], LAYERS[4]], ids=["16to8", "8to3"])
def test_stored_pixel_form_of_upsampled_layers_agrees_with_the_direct_form(spec, n, sv_opts):
    """A conv behind a 2x upsampling: the kernel sums a 4 x 4 window of dz per STORED pixel with tap sums as weights
    (default) or computes the upsampled image's gradient and adds 2 x 2 windows (option dgrad_phase = 0).  The same
    mathematics in a different association: agreement to rounding, and the switch does select two bodies."""
    from spectrum_vae_amd import ops
    h, w_, cin, cout, up, pooled, act = spec
    rng = np.random.default_rng(n + cout)
    g = ops.conv_geom(n, h * up, w_ * up, cin, 3, 3, 1, 1, "same", cout, (up, up))
    wt = torch.from_numpy(rng.standard_normal((3, 3, cin, cout)).astype(np.float32)).cuda()
    y = torch.from_numpy(np.tanh(rng.standard_normal((n, h * up, w_ * up, cout))).astype(np.float32)).cuda()
    dy = torch.from_numpy(rng.standard_normal((n, h * up, w_ * up, cout)).astype(np.float32)).cuda()
    got = {}
    for phase in ("1", "0"):
        sv_opts("dgrad_phase", int(phase))
        dx = torch.full((n, h, w_, cin), 7.0, device="cuda")
        ops.conv2d_bwd(g, None, wt, dy, y, None, dx, None, None, act, 1.0, None)
        got[phase] = dx
    assert float((got["1"] - got["0"]).abs().max()) <= 4e-6 * float(got["0"].abs().max())
    assert not torch.equal(got["1"], got["0"])


@pytest.mark.parametrize("n", [1, 5, 130, 1027])
def test_pixel_pair_form_of_the_8_channel_layer_equals_the_direct_form(n, sv_opts):
    """16 -> 8 channels back through a pooled 16 x 16 layer: rows 8..15 of the filter operand hold the filters shifted by
    one column, a result column is a pair of adjacent pixels (default), or every pixel is its own column
    (option dgrad_pair = 0).  The extra products have zero weights: the sums are the same, bit for bit."""
    from spectrum_vae_amd import ops
    h, w_, cin, cout, up, pooled, act = LAYERS[0]
    rng = np.random.default_rng(n + 17)
    g = ops.conv_geom(n, h, w_, cin, 3, 3, 1, 1, "same", cout, (1, 1))
    wt = torch.from_numpy(rng.standard_normal((3, 3, cin, cout)).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.standard_normal((n, h // 2, w_ // 2, cout)).astype(np.float32)).cuda()
    dy = torch.from_numpy(rng.standard_normal((n, h // 2, w_ // 2, cout)).astype(np.float32)).cuda()
    idx = torch.from_numpy(rng.integers(0, 4, (n, h // 2, w_ // 2, cout)).astype(np.uint8)).cuda()
    got = {}
    for pair in ("1", "0"):
        sv_opts("dgrad_pair", int(pair))
        dx = torch.full((n, h, w_, cin), 7.0, device="cuda")
        ops.conv2d_bwd(g, None, wt, dy, y, idx, dx, None, None, act, 1.0, None)
        got[pair] = dx
    assert torch.equal(got["1"], got["0"])


def chain_jobs(n, seed):
    """The five input-gradient jobs of the default stack's backward pass, decoder first, each reading the previous
    job's dx as its dy."""
    from spectrum_vae_amd import ops
    rng = np.random.default_rng(seed)
    jobs, dy = [], None
    for h, w_, cin, cout, up, pooled, act in (LAYERS[4], LAYERS[3], LAYERS[2], LAYERS[1], LAYERS[0]):
        H, W = h * up, w_ * up
        oh, ow = (H // 2, W // 2) if pooled else (H, W)
        g = ops.conv_geom(n, H, W, cin, 3, 3, 1, 1, "same", cout, (up, up))
        y = rng.standard_normal((n, oh, ow, cout)).astype(np.float32)
        if dy is None:
            dy = torch.from_numpy(rng.standard_normal((n, oh, ow, cout)).astype(np.float32)).cuda()
        assert tuple(dy.shape) == (n, oh, ow, cout)
        idx = torch.from_numpy(rng.integers(0, 4, (n, oh, ow, cout)).astype(np.uint8)).cuda() if pooled else None
        dx = torch.full((n, h, w_, cin), 7.0, device="cuda")
        jobs.append(dict(g=g, w=torch.from_numpy((0.3 * rng.standard_normal((3, 3, cin, cout))).astype(np.float32)).cuda(),
                         dy=dy, y=torch.from_numpy(np.tanh(y) if act == "tanh" else y).cuda(), idx=idx, dx=dx, act=act,
                         alpha=1.0))
        dy = dx
    return jobs


@pytest.mark.parametrize("grid", [-1, 2, 256, 1024])
@pytest.mark.parametrize("n", [1, 5, 130, 1024, 1027])
def test_chain_of_layers_in_one_launch_equals_one_launch_per_layer(n, grid, sv_opts):
    from spectrum_vae_amd import ops
    sv_opts("dgrad_chain", grid)
    jobs = chain_jobs(n, 11 + n)
    assert ops.conv2d_dgrad_chain_supported(jobs)
    ops.conv2d_dgrad_chain(jobs)
    got = [j["dx"].clone() for j in jobs]
    for j in jobs:
        j["dx"].fill_(-3.0)
    for j in jobs:
        ops.conv2d_bwd(j["g"], None, j["w"], j["dy"], j["y"], j["idx"], j["dx"], None, None, j["act"], j["alpha"], None)
    for a, j in zip(got, jobs):
        assert torch.equal(a, j["dx"])  # the same products summed in the same order, whoever owns the sample


def test_chain_declines_what_it_does_not_cover(sv_opts):
    from spectrum_vae_amd import ops
    jobs = chain_jobs(8, 3)
    sv_opts("dgrad_chain", 0)
    assert not ops.conv2d_dgrad_chain_supported(jobs)
    sv_opts("dgrad_chain", -1)
    other = chain_jobs(4, 3)
    assert not ops.conv2d_dgrad_chain_supported(jobs[:2] + other[2:])  # batch sizes differ
    g5 = ops.conv_geom(8, 16, 16, 8, 5, 5, 1, 1, "same", 16)
    bad = dict(jobs[0], g=g5)
    assert not ops.conv2d_dgrad_chain_supported([bad] + jobs[1:])
    with pytest.raises(RuntimeError, match="outside the few-channel 3x3 family"):
        ops.conv2d_dgrad_chain([bad] + jobs[1:])


@pytest.mark.parametrize("n", [1, 5, 130, 1027])
@pytest.mark.parametrize("spec", [s for s in LAYERS_8TO8 if s[0] * s[4] >= 16 and not s[5]], ids=lambda s: f"at{s[0] * s[4]}up{s[4]}")
def test_forms_of_the_8_to_8_layers_agree_with_the_direct_form(spec, n, sv_opts):
    """Pixel pairs as result columns (plain layers) and the window over the stored pixels (upsampled layers) against the
    direct form of the same kernel: the same products, summed in another order."""
    from spectrum_vae_amd import ops
    h, w_, cin, cout, up, pooled, act = spec
    H, W = h * up, w_ * up
    rng = np.random.default_rng(3 * n + H)
    g = ops.conv_geom(n, H, W, cin, 3, 3, 1, 1, "same", cout, (up, up))
    wt = torch.from_numpy(rng.standard_normal((3, 3, cin, cout)).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.standard_normal((n, H, W, cout)).astype(np.float32)).cuda()
    dy = torch.from_numpy(rng.standard_normal((n, H, W, cout)).astype(np.float32)).cuda()
    out = {}
    for form in (1, 0):
        sv_opts("dgrad_phase", form)
        sv_opts("dgrad_pair", form)
        dx = torch.full((n, h, w_, cin), 7.0, device="cuda")
        ops.conv2d_bwd(g, None, wt, dy, y, None, dx, None, None, act, 1.0, None)
        out[form] = dx
    err = float((out[1] - out[0]).abs().max() / out[0].abs().max())
    assert err < 2e-6, err

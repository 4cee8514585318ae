"""GPU parity: every C-ABI primitive against the float64 NumPy oracle on the same seeded inputs.

Tolerance: the north star's fp32 bar, relative error <= 1e-4 measured as
max|hip - oracle| / max(1, max|oracle|) (float32 arithmetic vs a float64 oracle).
"""
import zlib

import numpy as np
import pytest
import torch

from oracle import np_ops

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device="cuda", dtype=dtype).contiguous()


def close(got: torch.Tensor, want: np.ndarray, tol=RTOL, what=""):
    g = got.detach().cpu().numpy().astype(np.float64)
    assert g.shape == want.shape, (what, g.shape, want.shape)
    err = np.max(np.abs(g - want)) / max(1.0, np.max(np.abs(want)))
    assert np.isfinite(g).all(), what
    assert err <= tol, f"{what}: rel err {err:.3e} > {tol}"


# every (kh, kw, sh, sw, padding, H, W) variant the reference's models instantiate, plus odd sizes
CONV_CASES = [
    (3, 3, 1, 1, "same", 32, 32), (5, 5, 1, 1, "same", 16, 16), (1, 3, 1, 1, "valid", 8, 32),
    (1, 5, 1, 1, "valid", 4, 32), (2, 2, 2, 2, "same", 32, 32), (4, 4, 2, 2, "same", 32, 32),
    (4, 4, 1, 1, "valid", 4, 4), (1, 4, 1, 2, "same", 4, 32), (4, 4, 1, 1, "same", 6, 6),
    (3, 3, 2, 2, "same", 7, 9), (3, 3, 1, 1, "valid", 5, 7),
]
CHANNELS = [(3, 8), (8, 3), (1, 64), (16, 32), (5, 7)]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("ch", CHANNELS)
def test_conv2d_fwd_dgrad_wgrad(case, ch):
    from spectrum_vae_amd import ops
    kh, kw, sh, sw, pad, H, W = case
    cin, cout = ch
    rng = np.random.default_rng(zlib.crc32(repr((case, ch)).encode()))
    n = 3
    x = rng.standard_normal((n, H, W, cin))
    w = rng.standard_normal((kh, kw, cin, cout)) * 0.3
    b = rng.standard_normal(cout)
    g = ops.conv_geom(n, H, W, cin, kh, kw, sh, sw, pad, cout)
    for act in ("elu", None, "tanh"):
        y = torch.empty((n, g.Ho, g.Wo, cout), device="cuda")
        ops.conv2d_fwd(g, dev(x), dev(w), dev(b), y, act, 1.0)
        close(y, np_ops.act_fwd(np_ops.conv2d(x, w, b, (sh, sw), pad), act), what=f"fwd {act}")
    dz = rng.standard_normal((n, g.Ho, g.Wo, cout))
    dx_o, dw_o, _ = np_ops.conv2d_grads(x, w, dz, (sh, sw), pad)
    dx = torch.empty((n, H, W, cin), device="cuda")
    ops.conv2d_dgrad(g, dev(dz), dev(w), dx)
    close(dx, dx_o, what="dgrad")
    ws = torch.empty(max(ops.conv2d_wgrad_workspace(g), 16), dtype=torch.uint8, device="cuda")
    dw = torch.full((kh, kw, cin, cout), 7.0, device="cuda")
    ops.conv2d_wgrad(g, dev(x), dev(dz), dw, False, ws)
    close(dw, dw_o, what="wgrad")
    ops.conv2d_wgrad(g, dev(x), dev(dz), dw, True, ws)
    close(dw, 2 * dw_o, what="wgrad accumulate")


@pytest.mark.parametrize("case", [c for c in CONV_CASES if not (c[4] == "same" and c[2] == 1 and c[0] == 4)])
@pytest.mark.parametrize("ch", [(3, 8), (64, 3), (1000 // 8, 16)])
def test_conv2d_transpose(case, ch):
    from spectrum_vae_amd import ops
    kh, kw, sh, sw, pad, H, W = case
    H, W = max(H // 4, 1), max(W // 4, 1)
    cin, cout = ch
    rng = np.random.default_rng(zlib.crc32(repr((case, ch, "T")).encode()))
    n = 2
    x = rng.standard_normal((n, H, W, cin))
    w = rng.standard_normal((kh, kw, cout, cin)) * 0.3
    b = rng.standard_normal(cout)
    g = ops.deconv_geom(n, H, W, cin, kh, kw, sh, sw, pad, cout)
    y = torch.empty((n, g.H, g.W, cout), device="cuda")
    ops.conv2d_transpose_fwd(g, dev(x), dev(w), dev(b), y, "relu", 1.0)
    y_o = np_ops.conv2d_transpose(x, w, b, (sh, sw), pad)
    close(y, np_ops.act_fwd(y_o, "relu"), what="convT fwd")
    dz = rng.standard_normal(y_o.shape)
    dx_o, dw_o, _ = np_ops.conv2d_transpose_grads(x, w, dz, (sh, sw), pad)
    dx = torch.empty((n, H, W, cin), device="cuda")
    ops.conv2d_transpose_dgrad(g, dev(dz), dev(w), dx)
    close(dx, dx_o, what="convT dgrad")
    ws = torch.empty(max(ops.conv2d_wgrad_workspace(g), 16), dtype=torch.uint8, device="cuda")
    dw = torch.empty((kh, kw, cout, cin), device="cuda")
    ops.conv2d_transpose_wgrad(g, dev(x), dev(dz), dw, False, ws)
    close(dw, dw_o, what="convT wgrad")


@pytest.mark.parametrize("k,s", [(4, 2), (2, 2), (4, 1)])
@pytest.mark.parametrize("cimg,crich,hs,ws", [(3, 64, 16, 16), (1, 16, 5, 40), (4, 40, 24, 9), (2, 128, 33, 17)])
def test_small_image_channel_input_gradient(k, s, cimg, crich, hs, ws):
    """The LDS-tiled kernel for an image side of <= 4 channels (output layers of the context encoder and the U-Net):
    as the input gradient of the mirrored conv, and as a Conv2DTranspose forward with bias and tanh; extents that leave
    tiles partly empty, channel counts that leave the last staging pass short."""
    from spectrum_vae_amd import ops
    rng = np.random.default_rng(zlib.crc32(repr((k, s, cimg, crich, hs, ws)).encode()))
    n = 2
    H, W = hs * s, ws * s
    g = ops.conv_geom(n, H, W, cimg, k, k, s, s, "same", crich)
    assert (g.Ho, g.Wo) == (hs, ws)
    x = rng.standard_normal((n, H, W, cimg))
    w = rng.standard_normal((k, k, cimg, crich)) * 0.2
    dz = rng.standard_normal((n, hs, ws, crich))
    dx_o, _, _ = np_ops.conv2d_grads(x, w, dz, (s, s), "same")
    dx = torch.full((n, H, W, cimg), 9.0, device="cuda")
    ops.conv2d_dgrad(g, dev(dz), dev(w), dx)
    close(dx, dx_o, what="input gradient")
    if s > 1:  # the same launch as a transposed conv's forward, with its epilogue
        b = rng.standard_normal(cimg)
        gt = ops.deconv_geom(n, hs, ws, crich, k, k, s, s, "same", cimg)
        y = torch.empty((n, H, W, cimg), device="cuda")
        ops.conv2d_transpose_fwd(gt, dev(dz), dev(w), dev(b), y, "tanh", 1.0)
        close(y, np.tanh(np_ops.conv2d_transpose(dz, w, b, (s, s), "same")), what="transposed forward")


@pytest.mark.parametrize("k,s", [(4, 2), (2, 2)])
def test_small_image_channel_kernel_forms_agree(k, s):
    """Launches with fewer 16 x 16 tiles than CUs take the form that splits the channels over the four waves of 8 x 8
    tiles; larger ones the one-pixel-per-thread form.  Same data through both (the first images of a batch of 300 and a
    batch of those images alone), and the oracle."""
    from spectrum_vae_amd import ops
    rng = np.random.default_rng(5 + k)
    n, hs, ws, crich, cimg = 300, 16, 16, 64, 3
    dz = rng.standard_normal((n, hs, ws, crich)).astype(np.float32)
    w = (rng.standard_normal((k, k, cimg, crich)) * 0.2).astype(np.float32)
    b = rng.standard_normal(cimg).astype(np.float32)
    outs = []
    for m in (n, 3):
        gt = ops.deconv_geom(m, hs, ws, crich, k, k, s, s, "same", cimg)
        y = torch.empty((m, hs * s, ws * s, cimg), device="cuda")
        ops.conv2d_transpose_fwd(gt, dev(dz[:m]), dev(w), dev(b), y, "tanh", 1.0)
        outs.append(y.cpu().numpy())
    np.testing.assert_allclose(outs[0][:3], outs[1], rtol=0, atol=1e-5)  # different summation orders
    close(torch.from_numpy(outs[1]), np.tanh(np_ops.conv2d_transpose(dz[:3], w, b, (s, s), "same")), what="split form")


@pytest.mark.parametrize("cfg", [
    # (kh, kw, s, padding, H, W, cin, cout, n): few result tiles, reductions of hundreds to thousands of terms
    (4, 4, 2, "same", 8, 8, 128, 96, 2), (4, 4, 1, "valid", 4, 4, 128, 200, 5), (4, 4, 2, "same", 4, 4, 256, 64, 3),
    (3, 3, 2, "same", 7, 9, 70, 33, 1),
])
def test_conv2d_with_scratch_cuts_the_reduction(cfg):
    """The scratch-taking entry points (reduction cut over workgroups, fixed-order sum + bias + activation) against the
    oracle, against the plain entry points, and bit-identical from run to run; plain and transposed convs."""
    from spectrum_vae_amd import ops
    kh, kw, st, pad, H, W, cin, cout, n = cfg
    rng = np.random.default_rng(zlib.crc32(repr(cfg).encode()))

    def scratch(g, op, must=False):
        nb = ops.conv2d_scratch(g, op)
        assert nb > 0 or not must, "this geometry was chosen to need scratch"
        return torch.empty(nb, dtype=torch.uint8, device="cuda") if nb else None

    x = rng.standard_normal((n, H, W, cin))
    w = rng.standard_normal((kh, kw, cin, cout)) * 0.1
    b = rng.standard_normal(cout)
    g = ops.conv_geom(n, H, W, cin, kh, kw, st, st, pad, cout)
    y_o = np_ops.act_fwd(np_ops.conv2d(x, w, b, (st, st), pad), "elu")
    ys = []
    for ws in (scratch(g, ops.OP_FWD, True), scratch(g, ops.OP_FWD), None):
        y = torch.full((n, g.Ho, g.Wo, cout), 9.0, device="cuda")
        ops.conv2d_fwd(g, dev(x), dev(w), dev(b), y, "elu", 1.0, ws)
        close(y, y_o, what="fwd")
        ys.append(y)
    assert torch.equal(ys[0], ys[1])
    dz = rng.standard_normal((n, g.Ho, g.Wo, cout))
    dx_o, _, _ = np_ops.conv2d_grads(x, w, dz, (st, st), pad)
    dxs = []
    for ws in (scratch(g, ops.OP_DGRAD), scratch(g, ops.OP_DGRAD), None):
        dx = torch.full((n, H, W, cin), 9.0, device="cuda")
        ops.conv2d_dgrad(g, dev(dz), dev(w), dx, ws)
        close(dx, dx_o, what="dgrad")
        dxs.append(dx)
    assert torch.equal(dxs[0], dxs[1])
    # the transposed conv with the same geometry struct: input on the (Ho, Wo, cout) side, output on the (H, W, cin) side
    if (H, W) == (ops.deconv_out_size(g.Ho, kh, st, pad), ops.deconv_out_size(g.Wo, kw, st, pad)):
        gt = ops.deconv_geom(n, g.Ho, g.Wo, cout, kh, kw, st, st, pad, cin)
        xt = rng.standard_normal((n, g.Ho, g.Wo, cout))
        bt = rng.standard_normal(cin)
        yt_o = np_ops.act_fwd(np_ops.conv2d_transpose(xt, w, bt, (st, st), pad), "tanh")
        for ws in (scratch(gt, ops.OP_TRANSPOSE_FWD), None):
            yt = torch.full((n, H, W, cin), 9.0, device="cuda")
            ops.conv2d_transpose_fwd(gt, dev(xt), dev(w), dev(bt), yt, "tanh", 1.0, ws)
            close(yt, yt_o, what="convT fwd")
        dzt = rng.standard_normal(yt_o.shape)
        dxt_o, _, _ = np_ops.conv2d_transpose_grads(xt, w, dzt, (st, st), pad)
        for ws in (scratch(gt, ops.OP_TRANSPOSE_DGRAD, True), None):
            dxt = torch.full((n, g.Ho, g.Wo, cout), 9.0, device="cuda")
            ops.conv2d_transpose_dgrad(gt, dev(dzt), dev(w), dxt, ws)
            close(dxt, dxt_o, what="convT dgrad")
    # a scratch that is too small is ignored, not an error
    y = torch.empty((n, g.Ho, g.Wo, cout), device="cuda")
    ops.conv2d_fwd(g, dev(x), dev(w), dev(b), y, "elu", 1.0, torch.empty(16, dtype=torch.uint8, device="cuda"))
    close(y, y_o, what="fwd, short scratch")


def test_conv2d_wgrad_of_a_wide_kernel_goes_straight_into_dw():
    """A kernel gradient with >= 512 result tiles (the 4x4x128x1000 bottleneck conv, context_encoder.py:77) is not cut
    into partial sums: the tiles are written / added to dw by the GEMM itself."""
    from spectrum_vae_amd import ops
    rng = np.random.default_rng(77)
    n, cin, cout = 3, 128, 1000
    x = rng.standard_normal((n, 4, 4, cin))
    dz = rng.standard_normal((n, 1, 1, cout))
    w = np.zeros((4, 4, cin, cout))
    g = ops.conv_geom(n, 4, 4, cin, 4, 4, 1, 1, "valid", cout)
    _, dw_o, _ = np_ops.conv2d_grads(x, w, dz, (1, 1), "valid")
    ws = torch.empty(max(ops.conv2d_wgrad_workspace(g), 16), dtype=torch.uint8, device="cuda")
    dw = torch.full((4, 4, cin, cout), 7.0, device="cuda")
    ops.conv2d_wgrad(g, dev(x), dev(dz), dw, False, ws)
    close(dw, dw_o, what="wgrad")
    ops.conv2d_wgrad(g, dev(x), dev(dz), dw, True, ws)
    close(dw, 2 * dw_o, what="wgrad accumulate")


@pytest.mark.parametrize("up", [(2, 2), (1, 2)])
def test_conv2d_folded_upsampling(up):
    """UpSampling2D folded into the conv gather == explicit upsample then conv, forward and both gradients."""
    from spectrum_vae_amd import ops
    rng = np.random.default_rng(5)
    n, h, w, cin, cout = 2, 4, 8, 16, 8
    x = rng.standard_normal((n, h, w, cin))
    k = rng.standard_normal((3, 3, cin, cout)) * 0.2
    b = rng.standard_normal(cout)
    xu = np_ops.upsample2d(x, up)
    g = ops.conv_geom(n, h * up[0], w * up[1], cin, 3, 3, 1, 1, "same", cout, up)
    y = torch.empty((n, g.Ho, g.Wo, cout), device="cuda")
    ops.conv2d_fwd(g, dev(x), dev(k), dev(b), y, "elu", 1.0)
    close(y, np_ops.act_fwd(np_ops.conv2d(xu, k, b, (1, 1), "same"), "elu"), what="fwd")
    dz = rng.standard_normal((n, g.Ho, g.Wo, cout))
    dxu, dw_o, _ = np_ops.conv2d_grads(xu, k, dz, (1, 1), "same")
    dx = torch.empty((n, h, w, cin), device="cuda")
    ops.conv2d_dgrad(g, dev(dz), dev(k), dx)
    close(dx, np_ops.upsample2d_bwd(dxu, up), what="dgrad")
    ws = torch.empty(max(ops.conv2d_wgrad_workspace(g), 16), dtype=torch.uint8, device="cuda")
    dw = torch.empty((3, 3, cin, cout), device="cuda")
    ops.conv2d_wgrad(g, dev(x), dev(dz), dw, False, ws)
    close(dw, dw_o, what="wgrad")


@pytest.mark.parametrize("cfg", [
    # (K, H, W, Cin, Cout, up, pooled, act, N)
    (3, 32, 32, 3, 8, 1, True, "elu", 5), (3, 16, 16, 8, 16, 1, True, "elu", 9), (3, 8, 8, 16, 32, 1, True, "relu", 7),
    (3, 4, 4, 32, 16, 2, False, "elu", 11), (3, 8, 8, 16, 8, 2, False, "elu", 6), (3, 16, 16, 8, 3, 2, False, "tanh", 3),
    (5, 16, 16, 16, 16, 1, True, "relu", 4), (5, 8, 8, 32, 16, 2, False, "elu", 5), (3, 32, 32, 16, 16, 1, False, None, 2),
    (3, 4, 4, 128, 128, 1, True, "elu", 3), (5, 2, 2, 64, 128, 2, False, "relu", 2), (3, 6, 12, 5, 7, 1, True, "sigmoid", 3),
])
def test_conv2d_fused_pool_fwd_and_fused_bwd(cfg):
    """conv+act+maxpool in one pass and the whole layer backward in the fused kernels, against the oracle chain
    conv -> act -> maxpool (forward) and maxpool_bwd -> act' -> conv grads (backward)."""
    from spectrum_vae_amd import ops
    K, h, w, cin, cout, up, pooled, act, n = cfg
    rng = np.random.default_rng(zlib.crc32(repr(cfg).encode()))
    x = rng.standard_normal((n, h, w, cin))
    k = rng.standard_normal((K, K, cin, cout)) * (0.5 / np.sqrt(K * K * cin))
    b = rng.standard_normal(cout) * 0.1
    H, W = h * up, w * up
    g = ops.conv_geom(n, H, W, cin, K, K, 1, 1, "same", cout, (up, up))
    assert ops.conv2d_fused_supported(g, pooled)
    xu = np_ops.upsample2d(x, (up, up)) if up > 1 else x
    z = np_ops.conv2d(xu, k, b, (1, 1), "same")
    a = np_ops.act_fwd(z, act)
    xd, kd, bd = dev(x), dev(k), dev(b)
    if pooled:
        y_o, arg_o = np_ops.maxpool2d(a, (2, 2), "valid")
        y = torch.empty(y_o.shape, device="cuda")
        idx = torch.empty(y_o.shape, dtype=torch.uint8, device="cuda")
        ops.conv2d_pool_fwd(g, xd, kd, bd, y, idx, act, 1.0)
        close(y, y_o, what="pooled fwd")
        # argmax may legitimately differ only where two window entries tie to float32 precision
        same = idx.cpu().numpy().astype(np.int64) == arg_o
        assert same.mean() > 0.999
        arg = idx.cpu().numpy().astype(np.int64)
    else:
        y_o, arg = a, None
        y = torch.empty(y_o.shape, device="cuda")
        ops.conv2d_fwd(g, xd, kd, bd, y, act, 1.0)
        close(y, y_o, what="fwd")
        idx = None
    dy = rng.standard_normal(y_o.shape)
    g_act = dy * np_ops.act_bwd_from_output(y_o, act)
    dz = np_ops.maxpool2d_bwd(g_act, arg, a.shape, (2, 2), "valid") if pooled else g_act
    dxu, dw_o, db_o = np_ops.conv2d_grads(xu, k, dz, (1, 1), "same")
    dx_o = np_ops.upsample2d_bwd(dxu, (up, up)) if up > 1 else dxu
    ws = torch.empty(max(ops.conv2d_wgrad_workspace(g), 16), dtype=torch.uint8, device="cuda")
    dx = torch.empty(x.shape, device="cuda")
    dw = torch.ones((K, K, cin, cout), device="cuda")
    db = torch.ones(cout, device="cuda")
    ops.conv2d_bwd(g, xd, kd, dev(dy), dev(y_o), idx, dx, dw, db, act, 1.0, ws)
    close(dx, dx_o, what="fused dx")
    close(dw, 1.0 + dw_o, what="fused dw (accumulating)")
    close(db, 1.0 + db_o, what="fused db (accumulating)")
    # without dx and without bias
    dw2 = torch.zeros((K, K, cin, cout), device="cuda")
    ops.conv2d_bwd(g, xd, kd, dev(dy), dev(y_o), idx, None, dw2, None, act, 1.0, ws)
    close(dw2, dw_o, what="fused dw only")
    # the two halves on their own, and the kernel gradient in its deferred form (partial sums now, one reduction later)
    dx3 = torch.empty(x.shape, device="cuda")
    ops.conv2d_bwd(g, None, kd, dev(dy), dev(y_o), idx, dx3, None, None, act, 1.0, None)
    close(dx3, dx_o, what="input-gradient half")
    ws2 = torch.empty_like(ws)
    dw3, db3 = torch.full((K, K, cin, cout), 2.0, device="cuda"), torch.full((cout,), 2.0, device="cuda")
    dw4 = torch.zeros((K, K, cin, cout), device="cuda")
    ops.conv2d_bwd_partial(g, xd, dev(dy), dev(y_o), idx, act, 1.0, True, ws)
    ops.conv2d_bwd_partial(g, xd, dev(dy), dev(y_o), idx, act, 1.0, False, ws2)
    ops.conv2d_wgrad_finish([(g, ws, dw3, db3), (g, ws2, dw4, None)])
    close(dw3, 2.0 + dw_o, what="deferred dw (accumulating)")
    close(db3, 2.0 + db_o, what="deferred db (accumulating)")
    close(dw4, dw_o, what="deferred dw, second job")


@pytest.mark.parametrize("act", ["relu", "elu", "lrelu", "tanh", "sigmoid"])
@pytest.mark.parametrize("shape", [(1000, 8), (37, 3), (5, 300), (4096, 64)])
def test_act_fwd_bwd_bias(act, shape):
    from spectrum_vae_amd import ops
    rng = np.random.default_rng(3)
    rows, c = shape
    z = rng.standard_normal(shape) * 2
    y = torch.empty(shape, device="cuda")
    ops.act_fwd(dev(z), y, act, 0.2)
    y_o = np_ops.act_fwd(z, act, 0.2)
    close(y, y_o, what="act fwd")
    dy = rng.standard_normal(shape)
    dz_o = dy * np_ops.act_bwd_from_output(y_o, act, 0.2)
    d = dev(dy)
    db = torch.ones(c, device="cuda")
    ws = torch.empty(ops.colreduce_workspace(rows, c), dtype=torch.uint8, device="cuda")
    ops.act_bwd(d, dev(y_o), d, db, True, rows, c, act, 0.2, ws)  # in place, accumulating into ones
    close(d, dz_o, what="act bwd")
    close(db, 1.0 + dz_o.sum(axis=0), what="bias grad")
    db2 = torch.empty(c, device="cuda")
    ops.bias_grad(dev(dz_o), db2, False, rows, c, ws)
    close(db2, dz_o.sum(axis=0), what="bias_grad")


@pytest.mark.parametrize("cfg", [((2, 2), "same", 32, 32, 8), ((2, 2), "valid", 7, 9, 5), ((1, 2), "valid", 4, 30, 16),
                                 ((2, 2), "same", 5, 5, 3)])
def test_maxpool(cfg):
    from spectrum_vae_amd import ops
    pool, pad, H, W, C = cfg
    rng = np.random.default_rng(11)
    n = 3
    x = rng.standard_normal((n, H, W, C)).astype(np.float32).astype(np.float64)
    y_o, arg_o = np_ops.maxpool2d(x, pool, pad)
    ho, wo = y_o.shape[1:3]
    pt = np_ops.same_pads(H, pool[0], pool[0])[1] if pad == "same" else 0
    pl = np_ops.same_pads(W, pool[1], pool[1])[1] if pad == "same" else 0
    y = torch.empty(y_o.shape, device="cuda")
    idx = torch.empty(y_o.shape, dtype=torch.uint8, device="cuda")
    ops.maxpool2d_fwd(dev(x), y, idx, n, H, W, C, pool[0], pool[1], pt, pl, ho, wo)
    close(y, y_o, tol=0, what="maxpool fwd")
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), arg_o)
    dy = rng.standard_normal(y_o.shape)
    dx = torch.empty(x.shape, device="cuda")
    ops.maxpool2d_bwd(dev(dy), idx, dx, n, H, W, C, pool[0], pool[1], pt, pl, ho, wo)
    close(dx, np_ops.maxpool2d_bwd(dy, arg_o, x.shape, pool, pad), tol=1e-7, what="maxpool bwd")


@pytest.mark.parametrize("size", [(2, 2), (1, 2)])
def test_upsample(size):
    from spectrum_vae_amd import ops
    rng = np.random.default_rng(12)
    n, h, w, c = 2, 4, 5, 7
    x = rng.standard_normal((n, h, w, c)).astype(np.float32).astype(np.float64)
    y = torch.empty((n, h * size[0], w * size[1], c), device="cuda")
    ops.upsample2d_fwd(dev(x), y, n, h, w, c, size[0], size[1])
    close(y, np_ops.upsample2d(x, size), tol=0)
    dy = rng.standard_normal(tuple(y.shape))
    dx = torch.empty(x.shape, device="cuda")
    ops.upsample2d_bwd(dev(dy), dx, n, h, w, c, size[0], size[1])
    close(dx, np_ops.upsample2d_bwd(dy, size), tol=1e-6)


@pytest.mark.parametrize("shape", [(4, 16, 16, 64), (64, 100), (2, 3, 5, 7), (1024, 4, 4, 128), (3, 2, 2, 1000)])
def test_batchnorm(shape):
    from spectrum_vae_amd import ops
    rng = np.random.default_rng(13)
    c = shape[-1]
    rows = int(np.prod(shape[:-1]))
    x = rng.standard_normal(shape) * 3 + 5  # offset mean stresses the one-pass variance
    gamma, beta = rng.standard_normal(c), rng.standard_normal(c)
    mm, mv = rng.standard_normal(c), rng.random(c) + 0.5
    y_o, mean_o, var_o = np_ops.batchnorm_train(x, gamma, beta)
    unb = len(shape) == 4
    mm_o, mv_o = np_ops.batchnorm_moving_update(mm, mv, mean_o, var_o, rows, unbiased_var=unb)
    ws = torch.empty(ops.colreduce_workspace(rows, c), dtype=torch.uint8, device="cuda")
    y = torch.empty(shape, device="cuda")
    sm, si = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    dmm, dmv = dev(mm), dev(mv)
    ops.batchnorm_fwd_train(dev(x), dev(gamma), dev(beta), y, sm, si, dmm, dmv, rows, c, 1e-3, 0.99, unb, ws)
    close(y, y_o, what="bn train")
    close(sm, mean_o, what="bn mean")
    close(si, 1 / np.sqrt(var_o + 1e-3), what="bn invstd")
    close(dmm, mm_o, what="moving mean")
    close(dmv, mv_o, what="moving var")
    yi = torch.empty(shape, device="cuda")
    ops.batchnorm_fwd_infer(dev(x), dev(gamma), dev(beta), dev(mm), dev(mv), yi, rows, c, 1e-3)
    close(yi, np_ops.batchnorm_infer(x, gamma, beta, mm, mv), what="bn infer")
    dy = rng.standard_normal(shape)
    dx_o, dg_o, db_o = np_ops.batchnorm_train_bwd(x, gamma, dy)
    dx = torch.empty(shape, device="cuda")
    dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
    ops.batchnorm_bwd(dev(x), dev(dy), dev(gamma), sm, si, dx, dg, db, True, rows, c, ws)
    close(dx, dx_o, what="bn dx")
    close(dg, dg_o, tol=2e-4, what="bn dgamma")
    close(db, db_o, what="bn dbeta")


@pytest.mark.parametrize("shape,groups", [((6, 8, 8, 64), 2), ((4, 4, 4, 300), 2), ((12, 5, 3, 7), 3), ((8, 16, 16, 64), 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batchnorm_groups_and_neighbouring_activations(shape, groups, dtype):
    """BatchNorm over ``groups`` independent parts of the batch with (a) the LeakyReLU that follows it and (b) the
    activation + bias gradient of the conv that feeds it folded into its launches, against the numpy restatement run
    group by group with the activations as separate steps.  bf16: the same arithmetic on bf16-rounded inputs, compared
    at bf16 resolution."""
    from spectrum_vae_amd import ops
    rng = np.random.default_rng(131)
    c = shape[-1]
    rows = int(np.prod(shape[:-1]))
    n = shape[0]
    lp = dtype == torch.bfloat16
    tol = 2.0 ** -7 if lp else RTOL

    def q(a):  # what the kernels read
        return dev(a, dtype).float().cpu().numpy().astype(np.float64) if lp else a

    pre = rng.standard_normal(shape) * 2 + 1     # the producing conv's pre-activation
    x = q(np_ops.act_fwd(pre, "lrelu", 0.2))            # ... its output, the layer's input
    gamma, beta = rng.standard_normal(c), rng.standard_normal(c)
    mm, mv = rng.standard_normal(c), rng.random(c) + 0.5
    dy = q(rng.standard_normal(shape))
    per = n // groups
    unb = True
    y_o, dx_o = np.empty(shape), np.empty(shape)
    mean_o, inv_o = np.empty((groups, c)), np.empty((groups, c))
    dg_o, db_o, dbias_o = np.zeros(c), np.zeros(c), np.zeros(c)
    mm_o, mv_o = mm.copy(), mv.copy()
    for g in range(groups):
        sl = slice(g * per, (g + 1) * per)
        bn, mean_o[g], var = np_ops.batchnorm_train(x[sl], gamma, beta)
        inv_o[g] = 1 / np.sqrt(var + 1e-3)
        mm_o, mv_o = np_ops.batchnorm_moving_update(mm_o, mv_o, mean_o[g], var, rows // groups, unbiased_var=unb)
        y_o[sl] = np_ops.act_fwd(bn, "lrelu", 0.2)
        gin = dy[sl] * np.where(y_o[sl] > 0, 1.0, 0.2)          # the following LeakyReLU's backward
        dxg, dgg, dbg = np_ops.batchnorm_train_bwd(x[sl], gamma, gin)
        dz = dxg * np.where(x[sl] > 0, 1.0, 0.2)                # the producing conv's activation backward
        dx_o[sl] = dz
        dg_o += dgg
        db_o += dbg
        dbias_o += dz.reshape(-1, c).sum(0)
    ws = torch.empty(ops.batchnorm_workspace(rows, c, groups), dtype=torch.uint8, device="cuda")
    xd, y = dev(x, dtype), torch.empty(shape, device="cuda", dtype=dtype)
    sm, si = torch.empty(groups, c, device="cuda"), torch.empty(groups, c, device="cuda")
    dmm, dmv = dev(mm), dev(mv)
    ops.batchnorm_act_fwd_train(xd, dev(gamma), dev(beta), y, sm, si, dmm, dmv, rows, c, 1e-3, 0.99, unb, groups, "lrelu",
                                0.2, ws)
    close(y.float(), y_o, tol=tol, what="bn+act train")
    close(sm, mean_o, what="bn group means")
    close(si, inv_o, tol=2e-4, what="bn group invstd")
    close(dmm, mm_o, what="moving mean after all groups")
    close(dmv, mv_o, what="moving var after all groups")
    dx = torch.empty(shape, device="cuda", dtype=dtype)
    dg, db, dbias = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda"), torch.full((c,), 2.0, device="cuda")
    # the oracle's sign tests use its own y; where the device's rounded y disagrees in sign the element is a tie
    ops.batchnorm_act_bwd(xd, dev(dy, dtype), y, "lrelu", 0.2, dev(gamma), sm, si, dx, dg, db, True, "lrelu", 0.2, dbias,
                          True, rows, c, groups, ws)
    gtol = 2.0 ** -5 if lp else 2e-4
    close(dx.float(), dx_o, tol=gtol, what="bn+act dx")
    close(dg, dg_o, tol=gtol, what="bn+act dgamma")
    close(db, db_o, tol=gtol, what="bn+act dbeta")
    close(dbias, 2.0 + dbias_o, tol=gtol, what="producer's bias gradient")
    yi = torch.empty(shape, device="cuda", dtype=dtype)
    ops.batchnorm_act_fwd_infer(xd, dev(gamma), dev(beta), dev(mm), dev(mv), yi, rows, c, 1e-3, "lrelu", 0.2)
    close(yi.float(), np_ops.act_fwd(np_ops.batchnorm_infer(x, gamma, beta, mm, mv), "lrelu", 0.2), tol=tol,
          what="bn+act infer")


def test_deferred_reductions_equal_immediate_ones():
    """Kernel / bias gradient slab sums queued between begin and flush and run as one launch give the per-call
    reductions' values; a dropped queue (flush(False)) launches nothing and later calls reduce at once again."""
    from spectrum_vae_amd import ops
    rng = np.random.default_rng(132)
    jobs = []
    for (n, h, w, cin, k, s, pad, cout) in [(16, 16, 16, 64, 4, 2, "same", 64), (16, 4, 4, 128, 4, 1, "valid", 200),
                                             (16, 8, 8, 64, 4, 2, "same", 128)]:
        g = ops.conv_geom(n, h, w, cin, k, k, s, s, pad, cout)
        x = dev(rng.standard_normal((n, h, w, cin)))
        dz = dev(rng.standard_normal((n, g.Ho, g.Wo, cout)))
        jobs.append((g, x, dz, (k, k, cin, cout)))
    def run(deferred):
        outs = []
        if deferred:
            ops.deferred_reduce_begin()
        for g, x, dz, wshape in jobs:
            dw = torch.full(wshape, 0.5, device="cuda")
            db = torch.full((wshape[-1],), 0.25, device="cuda")
            ws = torch.empty(max(ops.conv2d_wgrad_workspace(g), 256), dtype=torch.uint8, device="cuda")  # one per job
            ops.conv2d_wgrad(g, x, dz, dw, True, ws)
            rows = dz.numel() // wshape[-1]
            ws2 = torch.empty(ops.colreduce_workspace(rows, wshape[-1]), dtype=torch.uint8, device="cuda")
            ops.bias_grad(dz, db, True, rows, wshape[-1], ws2)
            outs.append((dw, db, ws, ws2))
        if deferred:
            before = [o[0].clone() for o in outs]
            ops.deferred_reduce_flush(True)
            assert any(not torch.equal(b, o[0]) for b, o in zip(before, outs))  # the sums ran in the flush, not before
        return outs
    imm, dfr = run(False), run(True)
    for (a, b, _, _), (c2, d, _, _) in zip(imm, dfr):
        assert torch.equal(a, c2)      # the same summation order as the per-call kernel-gradient reduction
        assert rel_close(d, b, 1e-6)   # bias sums: the same terms, associated differently from the per-call fold
    ops.deferred_reduce_begin()
    g, x, dz, wshape = jobs[0]
    dw = torch.zeros(wshape, device="cuda")
    ws = torch.empty(max(ops.conv2d_wgrad_workspace(g), 256), dtype=torch.uint8, device="cuda")
    ops.conv2d_wgrad(g, x, dz, dw, False, ws)
    ops.deferred_reduce_flush(False)
    torch.cuda.synchronize()
    dw2 = torch.zeros(wshape, device="cuda")
    ops.conv2d_wgrad(g, x, dz, dw2, False, ws)
    assert torch.equal(dw2, imm[0][0] - 0.5) or rel_close(dw2, imm[0][0] - 0.5)


def rel_close(a, b, tol=1e-5):
    return float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))


def test_concat_and_misc():
    from spectrum_vae_amd import ops
    rng = np.random.default_rng(14)
    a = rng.standard_normal((2, 4, 4, 64)).astype(np.float32).astype(np.float64)
    b = rng.standard_normal((2, 4, 4, 3)).astype(np.float32).astype(np.float64)
    y = torch.empty((2, 4, 4, 67), device="cuda")
    ops.concat_channels_fwd(dev(a), dev(b), y, 32, 64, 3)
    close(y, np.concatenate([a, b], axis=-1), tol=0)
    da, db = torch.empty(a.shape, device="cuda"), torch.empty(b.shape, device="cuda")
    ops.concat_channels_bwd(y, da, db, 32, 64, 3)
    close(da, a, tol=0)
    close(db, b, tol=0)
    # gather with cast from float64 and an index vector
    src = rng.standard_normal((10, 6))
    idx = torch.tensor([9, 0, 3], device="cuda")
    dst = torch.empty((3, 6), device="cuda")
    ops.gather_rows(dev(src, torch.float64), idx, dst)
    close(dst, src[[9, 0, 3]].astype(np.float32).astype(np.float64), tol=0)
    m = (rng.random((3, 6)) > 0.5).astype(np.float64)
    out = torch.empty((3, 6), device="cuda")
    ops.mask_input(dst, dev(m), out)
    close(out, dst.cpu().numpy().astype(np.float64) * (1 - m), tol=0)
    # reduce_sum over H
    x = rng.standard_normal((2, 5, 3, 4))
    r = torch.empty((2, 3, 4), device="cuda")
    ops.reduce_sum_h_fwd(dev(x), r, 2, 5, 3, 4)
    close(r, x.sum(axis=1), tol=1e-6)
    dx = torch.empty(x.shape, device="cuda")
    ops.reduce_sum_h_bwd(r, dx, 2, 5, 3, 4)
    close(dx, np.repeat(r.cpu().numpy().astype(np.float64)[:, None], 5, axis=1), tol=0)


@pytest.mark.parametrize("mkn", [(32, 2048, 1000), (7, 33, 5), (64, 512, 64), (130, 70, 129), (1024, 512, 64),
                                 (1027, 64, 8), (300, 8, 64)])
def test_dense(mkn):
    from spectrum_vae_amd import ops
    m, k, n = mkn
    rng = np.random.default_rng(15)
    x, w, b = rng.standard_normal((m, k)), rng.standard_normal((k, n)) / np.sqrt(k), rng.standard_normal(n)
    y = torch.empty((m, n), device="cuda")
    ops.dense_fwd(dev(x), dev(w), dev(b), y, m, k, n, "elu", 1.0)
    close(y, np_ops.act_fwd(np_ops.dense(x, w, b), "elu"), what="dense fwd")
    dz = rng.standard_normal((m, n))
    dx_o, dw_o, _ = np_ops.dense_grads(x, w, dz)
    dx = torch.empty((m, k), device="cuda")
    ops.dense_dgrad(dev(dz), dev(w), dx, m, k, n)
    close(dx, dx_o, what="dense dgrad")
    dw = torch.zeros((k, n), device="cuda")
    ops.dense_wgrad(dev(x), dev(dz), dw, True, m, k, n)
    close(dw, dw_o, what="dense wgrad")
    # with scratch the reduction over the batch is cut over many workgroups; the sums are added to what is there
    nbytes = ops.dense_wgrad_workspace(m, k, n)
    assert (nbytes > 0) == (m >= 64 and k * n <= 64 * 64 * 128)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda")
    dw2 = torch.ones((k, n), device="cuda")
    ops.dense_wgrad(dev(x), dev(dz), dw2, True, m, k, n, ws)
    close(dw2, dw_o + 1.0, what="dense wgrad (cut reduction)")
    dw3 = torch.ones((k, n), device="cuda")
    ops.dense_wgrad(dev(x), dev(dz), dw3, True, m, k, n, ws)
    assert torch.equal(dw2, dw3)  # fixed order of the partial sums


def test_losses():
    from spectrum_vae_amd import ops
    rng = np.random.default_rng(16)
    shape = (5, 8, 8, 3)
    t, p = rng.standard_normal(shape), rng.standard_normal(shape)
    part = torch.zeros(ops.loss_workspace_floats(), device="cuda")
    out = torch.empty(1, device="cuda")
    dp = torch.empty(shape, device="cuda")
    ops.loss_mse(dev(t), dev(p), out, part, dp)
    close(out, np.array([np_ops.mse(t, p)]), what="mse")
    close(dp, np_ops.mse_grad(t, p), tol=1e-6, what="mse grad")
    # masked, x10, weight read with a stride from channel 0 of an NHWC mask (run_context_encoder.py:198)
    mask = np.zeros(shape)
    mask[:, 1:5, 2:6, :] = 1
    ops.loss_mse(dev(t), dev(p), out, part, dp, 1.0, dev(mask), 3, 10.0)
    close(out, np.array([np_ops.mse_weighted(t, p, mask[..., 0] * 10)]), what="masked mse")
    close(dp, np_ops.mse_weighted_grad(t, p, mask[..., 0] * 10), tol=1e-6, what="masked mse grad")
    # all-ones mask == 10 * MSE
    ops.loss_mse(dev(t), dev(p), out, part, None, 1.0, dev(np.ones(shape)), 3, 10.0)
    close(out, np.array([10 * np_ops.mse(t, p)]), what="ones mask")
    mp = torch.empty(shape[:-1], device="cuda")
    ops.loss_mse_map(dev(t), dev(p), mp, dev(mask), 3, 10.0)
    close(mp, np_ops.mse_per_pixel(t, p, mask[..., 0] * 10), what="mse map")
    ops.loss_mae(dev(t), dev(p), out, part, dp, 100.0)
    close(out, np.array([np_ops.mae(t, p)]), what="mae")
    close(dp, 100 * np_ops.mae_grad(t, p), tol=1e-6, what="mae grad")
    z = rng.standard_normal((64, 1)) * 3
    pr = 1 / (1 + np.exp(-z))
    pr[0], pr[1] = 0.0, 1.0  # saturated predictions hit the clip
    d1 = torch.empty((64, 1), device="cuda")
    for label in (0.0, 1.0):
        ops.loss_bce(dev(pr), label, False, out, part, d1)
        # float32 evaluation of the clipped logs (what TensorFlow does) for the saturated entries
        close(out, np.array([np_ops.bce_probs(np.full_like(pr, label), pr, np.float32)]), what="bce probs")
        inner = slice(2, None)
        ops.loss_bce(dev(pr[inner]), label, False, out, part, d1[inner])
        close(out, np.array([np_ops.bce_probs(np.full_like(pr[inner], label), pr[inner])]), what="bce probs f64")
        close(d1[inner], np_ops.bce_probs_grad(np.full_like(pr[inner], label), pr[inner]), tol=1e-3,
              what="bce probs grad")
        ops.loss_bce(dev(z), label, True, out, part, d1)
        close(out, np.array([np_ops.bce_logits(np.full_like(z, label), z)]), what="bce logits")
        close(d1, np_ops.bce_logits_grad(np.full_like(z, label), z), tol=1e-6, what="bce logits grad")
    # accumulate
    ops.loss_mse(dev(t), dev(p), out, part)
    ops.loss_mse(dev(t), dev(p), out, part, accumulate=True)
    close(out, np.array([2 * np_ops.mse(t, p)]), what="accumulate")
    # running sum in the same launch: running += weight * loss
    run = torch.full((1,), 3.0, device="cuda")
    ops.loss_mse(dev(t), dev(p), out, part, running=run, running_weight=7.0)
    close(run, np.array([3.0 + 7.0 * np_ops.mse(t, p)]), what="running sum")
    # a launch of one workgroup finishes the loss itself (no fold launch): accumulate and the running sum as above
    t1, p1 = t[:1], p[:1]
    ops.loss_mse(dev(t1), dev(p1), out, part)
    ops.loss_mse(dev(t1), dev(p1), out, part, accumulate=True, running=run, running_weight=2.0)
    close(out, np.array([2 * np_ops.mse(t1, p1)]), what="one-workgroup accumulate")
    close(run, np.array([3.0 + 7.0 * np_ops.mse(t, p) + 4.0 * np_ops.mse(t1, p1)]), what="one-workgroup running sum")


def test_scalar_program_equals_fill_and_axpy():
    """The GAN steps' loss bookkeeping (run_context_encoder.py:246-251) as one launch: the same bits as a fill and an
    axpy per term, later statements reading earlier results."""
    from spectrum_vae_amd import ops
    rng = np.random.default_rng(3)
    lt, lf, l2 = (dev(rng.standard_normal(1)) for _ in range(3))
    w = 0.999
    d_ref, t_ref = torch.empty(1, device="cuda"), torch.empty(1, device="cuda")
    ops.fill(d_ref, 0.0); ops.axpy(d_ref, lt, 0.5); ops.axpy(d_ref, lf, 0.5)
    ops.fill(t_ref, 0.0); ops.axpy(t_ref, l2, w); ops.axpy(t_ref, d_ref, 1.0 - w)
    d, tot = torch.full((1,), 9.0, device="cuda"), torch.full((1,), 9.0, device="cuda")
    ops.scalar_program([(d, False, [(lt, 0.5), (lf, 0.5)]), (tot, False, [(l2, w), (d, 1.0 - w)])])
    assert torch.equal(d, d_ref) and torch.equal(tot, t_ref)
    ops.scalar_program([(tot, True, [(lt, 2.0)])])
    ops.axpy(t_ref, lt, 2.0)
    assert torch.equal(tot, t_ref)
    with pytest.raises(Exception):
        ops.scalar_program([(d, False, [(lt, 1.0)] * 4)])


def test_adam_keras_and_schedule():
    from spectrum_vae_amd import ops
    rng = np.random.default_rng(17)
    n = 1003
    theta = rng.standard_normal(n)
    th = dev(theta)
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    state = torch.zeros(2, dtype=torch.int64, device="cuda")
    th_o, m_o, v_o = theta.copy(), np.zeros(n), np.zeros(n)
    for it in range(25):
        g = rng.standard_normal(n)
        ops.adam_prepare(state, 5e-4, 0.9, 0.999, 10, 0.96, True)
        ops.adam_apply(th, dev(g), m, v, state, 0.9, 0.999, 1e-7, 1.0)
        lr = np_ops.exponential_decay(5e-4, it, 10, 0.96, True)
        th_o, m_o, v_o = np_ops.adam_keras_step(th_o, g.astype(np.float32).astype(np.float64), m_o, v_o, it, lr)
    assert int(state[0].item()) == 25
    close(th, th_o, tol=1e-6, what="adam theta")
    close(m, m_o, tol=1e-6, what="adam m")
    # closed form of the first step: delta = -lr * g / (|g| + eps / sqrt(1 - b2))
    th1 = torch.zeros(8, device="cuda")
    g1 = np.array([1.0, -2.0, 0.5, 1e-3, -1e-3, 3.0, -0.1, 10.0])
    st = torch.zeros(2, dtype=torch.int64, device="cuda")
    ops.adam_prepare(st, 1e-3, 0.9, 0.999, 0, 1.0, False)
    ops.adam_apply(th1, dev(g1), torch.zeros(8, device="cuda"), torch.zeros(8, device="cuda"), st, 0.9, 0.999, 1e-7)
    close(th1, -1e-3 * g1 / (np.abs(g1) + 1e-7 / np.sqrt(1 - 0.999)), tol=1e-7, what="adam closed form")
    # zero_grad: same update, and the gradient buffer is cleared by the launch (1003 elements: vector body + scalar tail)
    gz = dev(rng.standard_normal(n))
    th_a, th_b = th.clone(), th.clone()
    m_a, v_a, m_b, v_b = m.clone(), v.clone(), m.clone(), v.clone()
    ga, gb = gz.clone(), gz.clone()
    ops.adam_apply(th_a, ga, m_a, v_a, state, 0.9, 0.999, 1e-7, 0.5)
    ops.adam_apply(th_b, gb, m_b, v_b, state, 0.9, 0.999, 1e-7, 0.5, zero_grad=True)
    assert torch.equal(th_a, th_b) and torch.equal(m_a, m_b) and torch.equal(v_a, v_b)
    assert torch.equal(ga, gz) and not gb.any()


@pytest.mark.parametrize("n", [1003, 70000, 2_000_003])
def test_adam_single_launch_step_equals_two_launch_form(n):
    """Schedule + update (+ the bf16 copy) in one launch: bit-equal to prepare + apply, count advanced exactly once per
    launch whatever the number of workgroups, arrival counter back at zero."""
    from spectrum_vae_amd import ops
    rng = np.random.default_rng(171)
    th_a = dev(rng.standard_normal(n))
    th_b = th_a.clone()
    m_a, v_a, m_b, v_b = (torch.zeros(n, device="cuda") for _ in range(4))
    st_a = torch.zeros(2, dtype=torch.int64, device="cuda")
    st_b = torch.zeros(3, dtype=torch.int64, device="cuda")
    lp = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    for it in range(12):
        g = dev(rng.standard_normal(n))
        ga, gb = g.clone(), g.clone()
        ops.adam_prepare(st_a, 5e-4, 0.9, 0.999, 5, 0.9, it % 2 == 0)
        ops.adam_apply(th_a, ga, m_a, v_a, st_a, 0.9, 0.999, 1e-7, 0.5, zero_grad=True)
        ops.adam_step(th_b, gb, m_b, v_b, st_b, 5e-4, 0.9, 0.999, 1e-7, 5, 0.9, it % 2 == 0, 0.5, zero_grad=True,
                      theta_lp=lp)
        assert not gb.any()
    assert torch.equal(th_a, th_b) and torch.equal(m_a, m_b) and torch.equal(v_a, v_b)
    assert torch.equal(st_a, st_b[:2]) and int(st_b[0].item()) == 12 and int(st_b[2].item()) == 0
    assert torch.equal(lp, th_b.to(torch.bfloat16))


def test_reparam_kl():
    from spectrum_vae_amd import ops
    rng = np.random.default_rng(18)
    b, d = 64, 8
    mu, lv, eps = rng.standard_normal((b, d)), rng.standard_normal((b, d)) * 0.5, rng.standard_normal((b, d))
    z, kl = torch.empty((b, d), device="cuda"), torch.empty(1, device="cuda")
    part = torch.zeros(ops.loss_workspace_floats(), device="cuda")
    ops.reparam_kl_fwd(dev(mu), dev(lv), dev(eps), z, None, kl, part)
    close(z, np_ops.sampling(mu, lv, eps), what="sampling")
    close(kl, np.array([np_ops.kl_standard_normal(mu, lv)]), what="kl")
    dz = rng.standard_normal((b, d))
    dmu, dlv = torch.empty((b, d), device="cuda"), torch.empty((b, d), device="cuda")
    ops.reparam_kl_bwd(dev(mu), dev(lv), dev(eps), dev(dz), dmu, dlv, 1.0)
    close(dmu, dz + mu / b, what="dmu")
    close(dlv, dz * 0.5 * np.exp(0.5 * lv) * eps + 0.5 * (np.exp(lv) - 1) / b, what="dlogvar")
    # generated epsilon: standard normal, replayable from (seed, offset)
    big = torch.zeros((4096, 64), device="cuda")
    e1, e2 = torch.empty_like(big), torch.empty_like(big)
    ops.reparam_kl_fwd(big, big, None, torch.empty_like(big), e1, kl, part, 1833, 0)
    ops.reparam_kl_fwd(big, big, None, torch.empty_like(big), e2, kl, part, 1833, 0)
    assert torch.equal(e1, e2)
    assert abs(float(e1.mean())) < 0.01 and abs(float(e1.std()) - 1.0) < 0.01


def test_dropout():
    from spectrum_vae_amd import ops
    x = torch.ones(1 << 18, device="cuda")
    y = torch.empty_like(x)
    ops.dropout_fwd(x, y, 0.25, 7, 0)
    kept = (y != 0).float().mean().item()
    assert abs(kept - 0.75) < 0.01
    assert torch.allclose(y[y != 0], torch.tensor(1 / 0.75, device="cuda"))
    dx = torch.empty_like(x)
    ops.dropout_bwd(x, dx, 0.25, 7, 0)
    assert torch.equal(dx, y)


def test_errors_are_loud():
    from spectrum_vae_amd import _lib, ops
    with pytest.raises(_lib.SpectrumHipError):
        ops.act_fwd(torch.zeros(4), torch.zeros(4), "relu", 0.0)  # CPU tensors: refused, no fallback
    g = ops.conv_geom(1, 8, 8, 3, 3, 3, 1, 1, "same", 4)
    x = torch.zeros((1, 8, 8, 3), device="cuda")
    with pytest.raises(_lib.SpectrumHipError, match="workspace"):
        ops.conv2d_wgrad(g, x, torch.zeros((1, 8, 8, 4), device="cuda"), torch.zeros((3, 3, 3, 4), device="cuda"),
                         False, torch.empty(8, dtype=torch.uint8, device="cuda"))


def test_convnet_entry_points_reject_what_they_do_not_cover():
    """Chains outside the fused family are reported by the support query and refused with an error text."""
    from spectrum_vae_amd import _lib, ops
    ok = dict(Cin=3, Cout=8, K=3, up=1, pool=1, act="elu", alpha=1.0, w_off=0, b_off=216)
    assert ops.convnet_supported(ops.net_layers([dict(ok, y=torch.empty(1, device="cuda"))]), 4, 32, 32)
    bad_k = ops.net_layers([dict(ok, K=5)])
    assert not ops.convnet_supported(bad_k, 4, 32, 32)
    odd = ops.net_layers([dict(ok)])
    assert not ops.convnet_supported(odd, 4, 31, 32)  # odd extent: no 2x2 tiling
    chain = ops.net_layers([dict(ok), dict(ok, Cin=4)])  # channel mismatch between layers
    assert not ops.convnet_supported(chain, 4, 32, 32)
    x = torch.zeros(4, 32, 32, 3, device="cuda")
    with pytest.raises(_lib.SpectrumHipError):
        ops.convnet_fwd(bad_k, 4, 32, 32, x, x)
    with pytest.raises(_lib.SpectrumHipError):  # supported shape but no output buffer
        ops.convnet_fwd(ops.net_layers([dict(ok)]), 4, 32, 32, x, x)


def test_conv2d_wgrad_batch_matches_per_layer_calls():
    """Several layers' kernel gradients from one launch equal the per-layer fused backward: bit for bit where both
    split the samples over workgroups alike, to rounding (2e-6 of the largest entry) where the splits differ; the
    batched call itself is bit-reproducible, with and without the deferred reduction."""
    from spectrum_vae_amd import ops

    def same(a, b):
        return torch.equal(a, b) or float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())

    rng = np.random.default_rng(5)
    specs = [(16, 16, 8, 16, 1, True, "elu"), (4, 4, 32, 16, 2, False, "elu"), (32, 32, 3, 8, 1, True, "relu"),
             (16, 16, 8, 3, 2, False, "tanh")]
    n = 6
    jobs, want = [], []
    for h, w, cin, cout, up, pooled, act in specs:
        H, W = h * up, w * up
        g = ops.conv_geom(n, H, W, cin, 3, 3, 1, 1, "same", cout, (up, up))
        oh, ow = (H // 2, W // 2) if pooled else (H, W)
        x = dev(rng.standard_normal((n, h, w, cin)))
        y = dev(rng.standard_normal((n, oh, ow, cout)))
        dy = dev(rng.standard_normal((n, oh, ow, cout)))
        idx = torch.from_numpy(rng.integers(0, 4, (n, oh, ow, cout)).astype(np.uint8)).cuda() if pooled else None
        wt = dev(rng.standard_normal((3, 3, cin, cout)))
        ws = torch.empty(max(ops.conv2d_wgrad_workspace(g), 16), dtype=torch.uint8, device="cuda")
        dw_ref, db_ref = torch.zeros(3, 3, cin, cout, device="cuda"), torch.zeros(cout, device="cuda")
        ops.conv2d_bwd(g, x, wt, dy, y, idx, None, dw_ref, db_ref, act, 1.0, ws)
        want.append((dw_ref, db_ref))
        jobs.append(dict(g=g, x=x, dy=dy, y=y, idx=idx, act=act, alpha=1.0, dw=torch.zeros_like(dw_ref),
                         db=torch.zeros_like(db_ref), ws=torch.empty_like(ws)))
    ops.conv2d_wgrad_batch(jobs, reduce=True)
    for j, (dw_ref, db_ref) in zip(jobs, want):
        assert same(j["dw"], dw_ref) and same(j["db"], db_ref)
    first = [(j["dw"].clone(), j["db"].clone()) for j in jobs]
    for j in jobs:
        j["dw"].zero_(), j["db"].zero_()
    ops.conv2d_wgrad_batch(jobs, reduce=False)
    ops.conv2d_wgrad_finish([(j["g"], j["ws"], j["dw"], j["db"]) for j in jobs])
    for j, (dw1, db1) in zip(jobs, first):  # reduce=True may run the channel-last kernel: equal to rounding
        assert same(j["dw"], dw1) and same(j["db"], db1)
    again = [(j["dw"].clone(), j["db"].clone()) for j in jobs]
    for j in jobs:
        j["dw"].zero_(), j["db"].zero_()
    ops.conv2d_wgrad_batch(jobs, reduce=False)
    ops.conv2d_wgrad_finish([(j["g"], j["ws"], j["dw"], j["db"]) for j in jobs])
    for j, (dw1, db1) in zip(jobs, again):
        assert torch.equal(j["dw"], dw1) and torch.equal(j["db"], db1)
    # fewer, longer-lived workgroups per layer: another summation order, same sums to rounding
    for j in jobs:
        j["dw"].zero_(), j["db"].zero_()
    ops.conv2d_wgrad_batch(jobs, reduce=False, blocks=2)
    ops.conv2d_wgrad_finish([(j["g"], j["ws"], j["dw"], j["db"], 2) for j in jobs])
    for j, (dw_ref, db_ref) in zip(jobs, want):
        close(j["dw"], dw_ref.cpu().numpy(), what="batched dw, 2 workgroups")
        close(j["db"], db_ref.cpu().numpy(), what="batched db, 2 workgroups")


@pytest.mark.parametrize("shape,pct", [((5, 32, 32, 3), 0.25), ((3, 32, 32, 1), 0.5), ((2, 64, 64, 3), 0.25),
                                       ((4, 24, 32, 2), 0.3), ((1, 32, 32, 3), 0.0), ((2, 40, 40, 3), 0.85)])
def test_random_block_mask_equals_host_procedure(shape, pct):
    """Masks generated in HBM: bit-equal to the reference's block procedure (run_context_encoder.py:24-58) fed with the
    same counter-hash draws; coverage reached; a different offset gives different masks."""
    from spectrum_vae_amd import ops
    n, h, w, c = shape
    m = torch.full(shape, 7.0, device="cuda")
    ops.random_block_mask(m, pct, seed=1833, offset=12345)
    want = np_ops.random_block_mask(n, h, w, c, pct, seed=1833, offset=12345)
    got = m.cpu().numpy()
    assert np.array_equal(got, want)
    assert (got[..., 0].sum(axis=(1, 2)) / w ** 2 >= pct).all()
    seg = w // 8
    assert (got[:, seg:3 * seg, seg:3 * seg] == 1).all()
    if pct > 0.2:
        m2 = torch.empty(shape, device="cuda")
        ops.random_block_mask(m2, pct, seed=1833, offset=12346)
        assert not torch.equal(m, m2)
        assert len({got[i].tobytes() for i in range(n)}) == n  # images of a batch get different masks


def test_random_block_mask_rejects_unreachable_coverage():
    from spectrum_vae_amd import ops
    with pytest.raises(Exception):
        ops.random_block_mask(torch.empty((1, 4, 32, 3), device="cuda"), 0.25, 0)  # the reference would never return
    with pytest.raises(Exception):
        ops.random_block_mask(torch.empty((1, 32, 32, 3), device="cuda"), 0.95, 0)

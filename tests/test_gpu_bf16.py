"""GPU parity of the bf16 path: activations / filters stored as bfloat16, sums in float32.

Every comparison feeds the float64 oracle the SAME bf16-rounded inputs the kernel reads, so what is left is the
kernel's own error: float32 accumulation plus one rounding of the result to bf16 (2^-9 relative).  Bar for a bf16
result: |hip - oracle| <= 2^-8 * |oracle| + 2^-8 * rms(oracle) element-wise... stated per test below as BF16_RTOL
relative to the largest oracle magnitude; float32 results of bf16 inputs (kernel gradients) keep the float32 bar."""
import zlib

import numpy as np
import pytest
import torch

from oracle import np_ops

pytestmark = pytest.mark.gpu

BF16_RTOL = 2.0 ** -8   # one bf16 rounding of the result is 2^-9 relative; sums of up to 4096 float32 products add less
F32_RTOL = 1e-4


def bf(a):
    """Round to bf16 and back (float64 array of exactly representable values)."""
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).to(torch.float64).numpy()


def dev16(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device="cuda", dtype=torch.bfloat16).contiguous()


def dev32(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device="cuda", dtype=torch.float32).contiguous()


def close(got, want, tol, what=""):
    g = got.detach().to(torch.float64).cpu().numpy()
    assert g.shape == want.shape, (what, g.shape, want.shape)
    assert np.isfinite(g).all(), what
    err = np.max(np.abs(g - want)) / max(1.0, np.max(np.abs(want)))
    assert err <= tol, f"{what}: rel err {err:.3e} > {tol:.3e}"


# (kh, kw, s, padding, H, W, cin, cout, n): the GAN models' layer shapes at small batch, plus ragged ones
CASES = [
    (2, 2, 2, "same", 16, 16, 64, 64, 3),     # U-Net down block
    (2, 2, 2, "same", 32, 32, 1, 64, 2),      # U-Net first block: reduction of 4 -> direct kernel
    (4, 4, 2, "same", 32, 32, 4, 64, 2),      # U-Net discriminator, first block (4 channels: element gathers)
    (4, 4, 2, "same", 32, 32, 3, 64, 2),      # context-encoder first block (3 channels)
    (4, 4, 2, "same", 16, 16, 64, 128, 5),    # 1024-term reductions
    (4, 4, 1, "valid", 4, 4, 128, 200, 5),    # bottleneck: one output position, cut reduction
    (4, 4, 1, "valid", 4, 4, 256, 1, 7),      # discriminator head: one logit
    (3, 3, 2, "same", 7, 9, 72, 40, 1),       # ragged extents, tiles partly empty
    (4, 4, 2, "same", 8, 8, 128, 96, 2),
    # enough result tiles for the 128-row tiles (128 x 128 and 128 x 64), with ragged row counts
    (4, 4, 2, "same", 16, 16, 64, 128, 771),
    (2, 2, 2, "same", 16, 16, 64, 64, 773),
    (4, 4, 2, "same", 8, 8, 64, 136, 260),
]


@pytest.mark.parametrize("cfg", CASES)
def test_conv2d_bf16(cfg):
    from spectrum_vae_amd import ops
    kh, kw, st, pad, H, W, cin, cout, n = cfg
    rng = np.random.default_rng(zlib.crc32(repr(cfg).encode()))
    x = bf(rng.standard_normal((n, H, W, cin)))
    w = bf(rng.standard_normal((kh, kw, cin, cout)) * 0.2)
    b = rng.standard_normal(cout).astype(np.float32).astype(np.float64)
    g = ops.conv_geom(n, H, W, cin, kh, kw, st, st, pad, cout)

    def scratch(op):
        nb = ops.conv2d_scratch(g, op)
        return torch.empty(nb, dtype=torch.uint8, device="cuda") if nb else None

    for act in ("lrelu", None):
        y_o = np_ops.act_fwd(np_ops.conv2d(x, w, b, (st, st), pad), act, 0.2)
        for ws in (scratch(ops.OP_FWD), None):
            y = torch.full((n, g.Ho, g.Wo, cout), 9.0, device="cuda", dtype=torch.bfloat16)
            ops.conv2d_fwd(g, dev16(x), dev16(w), dev32(b), y, act, 0.2, ws)
            close(y, y_o, BF16_RTOL, f"fwd {act}")
    dz = bf(rng.standard_normal((n, g.Ho, g.Wo, cout)))
    dx_o, dw_o, _ = np_ops.conv2d_grads(x, w, dz, (st, st), pad)
    for ws in (scratch(ops.OP_DGRAD), None):
        dx = torch.full((n, H, W, cin), 9.0, device="cuda", dtype=torch.bfloat16)
        ops.conv2d_dgrad(g, dev16(dz), dev16(w), dx, ws)
        close(dx, dx_o, BF16_RTOL, "dgrad")
    wsb = torch.empty(max(ops.conv2d_wgrad_workspace(g), 16), dtype=torch.uint8, device="cuda")
    dw = torch.full((kh, kw, cin, cout), 7.0, device="cuda")
    ops.conv2d_wgrad(g, dev16(x), dev16(dz), dw, False, wsb)
    close(dw, dw_o, F32_RTOL, "wgrad")
    ops.conv2d_wgrad(g, dev16(x), dev16(dz), dw, True, wsb)
    close(dw, 2 * dw_o, F32_RTOL, "wgrad accumulate")


# transposed convs: (kh, kw, s, padding, H, W, cin, cout, n) with (H, W, cin) the layer INPUT
T_CASES = [
    (2, 2, 2, "same", 4, 4, 64, 64, 3),       # U-Net up block
    (2, 2, 2, "same", 16, 16, 128, 3, 2),     # U-Net output layer: 3 channels
    (4, 4, 1, "valid", 1, 1, 200, 128, 5),    # context-encoder decoder head: 1x1 -> 4x4
    (4, 4, 2, "same", 4, 4, 128, 64, 3),
    (4, 4, 2, "same", 16, 16, 64, 3, 2),      # context-encoder output layer
    (3, 3, 2, "same", 5, 6, 24, 10, 1),
    (4, 4, 2, "same", 8, 8, 128, 128, 195),   # 4 row classes x 128-row tiles
    (2, 2, 2, "same", 8, 8, 128, 64, 770),
]


@pytest.mark.parametrize("cfg", T_CASES)
def test_conv2d_transpose_bf16(cfg):
    from spectrum_vae_amd import ops
    kh, kw, st, pad, H, W, cin, cout, n = cfg
    rng = np.random.default_rng(zlib.crc32(repr((cfg, "T")).encode()))
    x = bf(rng.standard_normal((n, H, W, cin)))
    w = bf(rng.standard_normal((kh, kw, cout, cin)) * 0.2)
    b = rng.standard_normal(cout).astype(np.float32).astype(np.float64)
    g = ops.deconv_geom(n, H, W, cin, kh, kw, st, st, pad, cout)

    def scratch(op):
        nb = ops.conv2d_scratch(g, op)
        return torch.empty(nb, dtype=torch.uint8, device="cuda") if nb else None

    y_o = np_ops.conv2d_transpose(x, w, b, (st, st), pad)
    for ws in (scratch(ops.OP_TRANSPOSE_FWD), None):
        y = torch.full((n, g.H, g.W, cout), 9.0, device="cuda", dtype=torch.bfloat16)
        ops.conv2d_transpose_fwd(g, dev16(x), dev16(w), dev32(b), y, "tanh", 1.0, ws)
        close(y, np_ops.act_fwd(y_o, "tanh"), BF16_RTOL, "convT fwd")
    dz = bf(rng.standard_normal(y_o.shape))
    dx_o, dw_o, _ = np_ops.conv2d_transpose_grads(x, w, dz, (st, st), pad)
    for ws in (scratch(ops.OP_TRANSPOSE_DGRAD), None):
        dx = torch.full((n, H, W, cin), 9.0, device="cuda", dtype=torch.bfloat16)
        ops.conv2d_transpose_dgrad(g, dev16(dz), dev16(w), dx, ws)
        close(dx, dx_o, BF16_RTOL, "convT dgrad")
    wsb = torch.empty(max(ops.conv2d_wgrad_workspace(g), 16), dtype=torch.uint8, device="cuda")
    dw = torch.empty((kh, kw, cout, cin), device="cuda")
    ops.conv2d_transpose_wgrad(g, dev16(x), dev16(dz), dw, False, wsb)
    close(dw, dw_o, F32_RTOL, "convT wgrad")

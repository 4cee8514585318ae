"""sv_conv2d_wgrad_static: the kernel gradient with its partial sums in the library's built-in device array (no
workspace argument, no allocation) against sv_conv2d_wgrad with a caller workspace, through the C ABI.

Bar: bit-identical (the same kernels run, the partial sums are folded in the same fixed order), on two calls in a row of
which the second has the smaller input - it must not see what the first left in the shared array."""
import gc

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(geom_args, seed):
    from spectrum_vae_amd import ops
    g = ops.conv_geom(*geom_args)
    rng = np.random.default_rng(seed)
    up = (g.up_h, g.up_w)
    x = torch.from_numpy(rng.standard_normal((g.N, g.H // up[0], g.W // up[1], g.Cin)).astype(np.float32)).cuda()
    dz = torch.from_numpy(rng.standard_normal((g.N, g.Ho, g.Wo, g.Cout)).astype(np.float32)).cuda()
    return g, x, dz


def _both(g, x, dz, accumulate):
    from spectrum_vae_amd import ops
    shape = (g.KH, g.KW, g.Cin, g.Cout)
    start = torch.linspace(-1, 1, int(np.prod(shape)), device="cuda").reshape(shape).contiguous()
    need = ops.conv2d_wgrad_workspace(g)
    assert 0 < need <= ops.static_scratch_bytes()
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    want = start.clone()
    ops.conv2d_wgrad(g, x, dz, want, accumulate, ws)
    got = start.clone()
    ops.static_scratch_bytes()  # the once-per-GPU address lookup, outside the measurement below
    # earlier tests' models (reference cycles) are collected now, not by a collection that happens to run inside the
    # measured window and returns their buffers to the allocator
    gc.collect()
    gc.disable()
    try:
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated()
        free_before = torch.cuda.mem_get_info()[0]
        ops.conv2d_wgrad_static(g, x, dz, got, accumulate)
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() == before  # nothing came from torch's allocator ...
        assert torch.cuda.mem_get_info()[0] == free_before  # ... and nothing from hipMalloc inside the library
    finally:
        gc.enable()
    return got, want


# (n, h, w, cin, kh, kw, sh, sw, padding, cout): first the larger call, then a smaller one of each kernel family
CASES = [
    ((64, 16, 16, 16, 3, 3, 1, 1, "same", 32), (5, 8, 8, 8, 3, 3, 1, 1, "same", 16)),        # LDS-tiled, MFMA + vector body
    ((48, 32, 32, 64, 4, 4, 2, 2, "same", 64), (3, 8, 8, 64, 4, 4, 2, 2, "same", 128)),      # implicit GEMM, split reduction
    ((32, 32, 32, 3, 5, 5, 1, 1, "same", 8), (2, 6, 10, 2, 1, 3, 1, 1, "valid", 3)),          # tiled 5x5, direct kernel
]


@pytest.mark.parametrize("big,small", CASES)
def test_static_scratch_kernel_gradient_equals_the_workspace_form(big, small):
    from spectrum_vae_amd import ops
    g1, x1, dz1 = _pair(big, 1)
    g2, x2, dz2 = _pair(small, 2)
    assert x2.numel() < x1.numel() and ops.conv2d_wgrad_workspace(g2) <= ops.conv2d_wgrad_workspace(g1)
    got1, want1 = _both(g1, x1, dz1, accumulate=False)
    got2, want2 = _both(g2, x2, dz2, accumulate=True)   # right after the larger call, adding to existing contents
    assert torch.equal(got1, want1)
    assert torch.equal(got2, want2)
    assert float(want1.abs().max()) > 0 and float(want2.abs().max()) > 0
    # and against the CPU restatement (float64), to the float32 bar of the other op tests
    from oracle import np_ops
    w0 = np.zeros((g2.KH, g2.KW, g2.Cin, g2.Cout))
    _, ref, _ = np_ops.conv2d_grads(x2.cpu().numpy(), w0, dz2.cpu().numpy(), (g2.SH, g2.SW), small[8])
    start = np.linspace(-1, 1, ref.size).reshape(ref.shape)
    err = np.abs(got2.cpu().numpy() - (start + ref)).max() / max(np.abs(ref).max(), 1e-3)
    assert err <= 1e-4


def test_static_scratch_refuses_what_does_not_fit():
    from spectrum_vae_amd import _lib, ops
    g = ops.conv_geom(8, 8, 8, 1024, 3, 3, 1, 1, "same", 1024)  # 36 MiB of kernel gradient alone
    need = ops.conv2d_wgrad_workspace(g)
    assert need > ops.static_scratch_bytes()
    x = torch.zeros(1, device="cuda")
    with pytest.raises(_lib.SpectrumHipError, match="built-in scratch"):
        ops.conv2d_wgrad_static(g, x, x, x, False)


def test_static_scratch_inside_a_deferred_reduce_bracket():
    """Between sv_deferred_reduce_begin and _flush the slab sums of ordinary calls are queued; a queued sum whose source
    is the shared array would be overwritten by the next static call.  The static entry point therefore reduces at once:
    two calls of different geometry inside one bracket both give the workspace form's bits."""
    from spectrum_vae_amd import ops
    (big, small) = CASES[1]  # implicit GEMM with a split reduction: the path that defers
    g1, x1, dz1 = _pair(big, 11)
    g2, x2, dz2 = _pair(small, 12)
    want = []
    for g, x, dz in ((g1, x1, dz1), (g2, x2, dz2)):
        w = torch.zeros((g.KH, g.KW, g.Cin, g.Cout), device="cuda")
        ops.conv2d_wgrad(g, x, dz, w, False, torch.empty(ops.conv2d_wgrad_workspace(g), dtype=torch.uint8, device="cuda"))
        want.append(w)
    got = [torch.zeros_like(w) for w in want]
    ops.deferred_reduce_begin()
    try:
        ops.conv2d_wgrad_static(g1, x1, dz1, got[0], False)
        ops.conv2d_wgrad_static(g2, x2, dz2, got[1], False)
    finally:
        ops.deferred_reduce_flush(True)
    torch.cuda.synchronize()
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    assert float(want[0].abs().max()) > 0 and float(want[1].abs().max()) > 0


def test_static_scratch_call_recorded_into_a_graph():
    """The use case the entry point names: recorded into a HIP graph (no allocation may happen there) and replayed."""
    from spectrum_vae_amd import ops
    from spectrum_vae_amd.nn import recording
    g, x, dz = _pair(CASES[0][0], 21)
    ops.static_scratch_bytes()  # address lookup before the capture, as the header asks
    ws = torch.empty(ops.conv2d_wgrad_workspace(g), dtype=torch.uint8, device="cuda")
    want = torch.zeros((g.KH, g.KW, g.Cin, g.Cout), device="cuda")
    ops.conv2d_wgrad(g, x, dz, want, False, ws)
    got = torch.full_like(want, 5.0)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with recording(graph):
        ops.conv2d_wgrad_static(g, x, dz, got, False)
    for _ in range(2):
        got.fill_(5.0)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(got, want)

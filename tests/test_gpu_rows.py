"""The forward launch that gathers its own batch (sv_convnet_fwd_mse_rows) against a gather launch followed by the plain
forward launch: same kernels on the same values, so every result is compared bit for bit."""
import numpy as np
import pytest
import torch

from tests.test_gpu_models import RUN_MODEL_DEFAULTS, data


def _model():
    from spectrum_vae_amd import autoencoder
    model = autoencoder.CAEPadded(**RUN_MODEL_DEFAULTS)
    model.build((None, 32, 32, 3))
    model._materialize()
    return model


def _chain(model, n):
    """(layer table, output tensors) of the model's conv chain at batch size n, on buffers of this test's own."""
    from spectrum_vae_amd import ops
    specs, outs = [], []
    for l in model._conv_chain():
        ho, wo, cout = l.output_shape_
        shp = (n, ho // 2, wo // 2, cout) if l.pool is not None else (n, ho, wo, cout)
        y = torch.full(shp, float("nan"), device="cuda")
        idx = torch.zeros(shp, dtype=torch.uint8, device="cuda") if l.pool is not None else None
        specs.append(dict(Cin=l.input_shape_[2], Cout=cout, K=3, up=l.up[0], pool=int(l.pool is not None), act=l.act,
                          alpha=l.alpha, w_off=l.kernel.offset, b_off=l.bias.offset if l.bias is not None else -1,
                          y=y, idx=idx))
        outs.append((y, idx))
    return ops.net_layers(specs), outs


@pytest.mark.gpu
@pytest.mark.parametrize("n,step", [(1, 0), (5, 3), (64, 1), (1024, 2), (1027, 0)])
def test_forward_that_gathers_its_batch_equals_gather_then_forward(n, step):
    from spectrum_vae_amd import ops
    model = _model()
    rows_total = 4 * n + 7
    X = torch.from_numpy(data(rows_total, seed=77 + n)).cuda()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(n)
    perm = torch.randperm(rows_total, device="cuda", generator=gen)
    arr_a, outs_a = _chain(model, n)
    arr_b, outs_b = _chain(model, n)
    part_a = torch.full((ops.loss_workspace_floats(),), float("nan"), device="cuda")
    part_b = part_a.clone()
    if not ops.convnet_rows_supported(arr_a, n, 32, 32):
        # a batch the kernel built for the default stack does not take (one sample): the call says so
        assert n == 1
        xb = torch.empty(n, 32, 32, 3, device="cuda")
        with pytest.raises(RuntimeError, match="does not gather"):
            ops.convnet_fwd_mse_rows(arr_b, n, 32, 32, ops.batch_rows(X, perm, xb), model._params, xb,
                                     torch.empty_like(xb), 1.0, part_b)
        return
    # reference: gather launch, then the forward launch with the loss
    xa = torch.empty(n, 32, 32, 3, device="cuda")
    ops.gather_rows(X, perm[step * n:(step + 1) * n], xa)
    da = torch.empty_like(xa)
    cnt_a = ops.convnet_fwd_mse(arr_a, n, 32, 32, xa, model._params, xa, da, 1.0, part_a)
    # the launch that gathers: counter 40 + step with base 40 walks `step` batches into the index
    xb = torch.full((n, 32, 32, 3), float("nan"), device="cuda")
    db = torch.empty_like(xb)
    counter = torch.tensor([40 + step, 0, 0], dtype=torch.int64, device="cuda")
    base = torch.tensor([40], dtype=torch.int64, device="cuda")
    rows = ops.batch_rows(X, perm, xb, counter, base, n)
    cnt_b = ops.convnet_fwd_mse_rows(arr_b, n, 32, 32, rows, model._params, xb, db, 1.0, part_b)
    torch.cuda.synchronize()
    assert cnt_a == cnt_b
    assert torch.equal(xa, xb), "the gathered batch"
    assert torch.equal(da, db) and torch.equal(part_a[:cnt_a], part_b[:cnt_b])
    for (ya, ia), (yb, ib) in zip(outs_a, outs_b):
        assert torch.equal(ya, yb)
        assert ia is None or torch.equal(ia, ib)


@pytest.mark.gpu
def test_positions_and_rows_outside_their_ranges_are_clamped():
    """A counter past the end of the index reads its last position, a row number past the tensor its last row: the
    launch never reads outside the two buffers."""
    from spectrum_vae_amd import ops
    model = _model()
    n = 4
    X = torch.from_numpy(data(10, seed=5)).cuda()
    index = torch.tensor([3, 99, -2, 7, 1, 0, 2, 5], dtype=torch.int64, device="cuda")
    arr, _ = _chain(model, n)
    part = torch.zeros(ops.loss_workspace_floats(), device="cuda")
    xb = torch.empty(n, 32, 32, 3, device="cuda")
    d = torch.empty_like(xb)
    ops.convnet_fwd_mse_rows(arr, n, 32, 32, ops.batch_rows(X, index, xb), model._params, xb, d, 1.0, part)
    assert torch.equal(xb, X[[3, 9, 0, 7]])
    counter = torch.tensor([9], dtype=torch.int64, device="cuda")
    base = torch.tensor([6], dtype=torch.int64, device="cuda")  # position 12 .. 15 of 8
    ops.convnet_fwd_mse_rows(arr, n, 32, 32, ops.batch_rows(X, index, xb, counter, base, n), model._params, xb, d, 1.0, part)
    assert torch.equal(xb, X[[5, 5, 5, 5]])


@pytest.mark.gpu
def test_rows_form_is_refused_for_other_layer_tables():
    from spectrum_vae_amd import autoencoder, ops
    model = autoencoder.CAEPadded(**dict(RUN_MODEL_DEFAULTS, initial_filters=16))
    model.build((None, 32, 32, 3))
    model._materialize()
    arr, _ = _chain(model, 8)
    assert not ops.convnet_rows_supported(arr, 8, 32, 32)
    X = torch.zeros(8, 32, 32, 3, device="cuda")
    assert not model.rows_step_supported(X, 8)


@pytest.mark.gpu
def test_fit_with_the_gather_inside_the_forward_equals_fit_with_gather_launches():
    """Three shuffled epochs with a ragged last batch, recorded steps included: identical parameters and losses."""
    from spectrum_vae_amd import losses
    from spectrum_vae_amd.optimizers import Adam
    x = data(200, seed=41)
    res = {}
    for inside in (True, False):
        model = _model()
        model.gather_in_forward = inside
        model.compile(optimizer=Adam(1e-3), loss=losses.MeanSquaredError())
        assert model.rows_step_supported(torch.from_numpy(x).cuda(), 64) == inside
        h = model.fit(x, x, epochs=3, batch_size=64, shuffle=True, verbose=0, seed=9)
        res[inside] = (np.array(h.history["loss"]), model._params.cpu().numpy().copy(), model.optimizer.iterations)
    assert res[True][2] == res[False][2] == 12
    assert np.array_equal(res[True][0], res[False][0])
    assert np.array_equal(res[True][1], res[False][1])

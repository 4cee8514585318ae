"""The launch that runs the input-gradient chain of the backward pass behind the forward pass
(sv_convnet_fwd_mse_dgrad, the first half of the training step of run_model.py:281-289) against the forward launch
followed by the chain launch: the same kernels' code on the same values, so every result is compared bit for bit."""
import numpy as np
import pytest
import torch

from tests.test_gpu_models import RUN_MODEL_DEFAULTS, data
from tests.test_gpu_rows import _chain, _model

pytestmark = pytest.mark.gpu


def _jobs(model, n, outs, dpred):
    """The input-gradient jobs of the conv chain's backward pass (last layer first; the first layer has none) on the
    forward buffers `outs`, the loss gradient `dpred` and dx buffers of this test's own."""
    from spectrum_vae_amd import ops
    chain = model._conv_chain()
    jobs, dy = [], dpred
    for li in range(len(chain) - 1, 0, -1):
        l = chain[li]
        h, w, cin = l.input_shape_
        H, W = h * l.up[0], w * l.up[1]
        g = ops.conv_geom(n, H, W, cin, 3, 3, 1, 1, "same", l.filters, l.up)
        y, idx = outs[li]
        dx = torch.full((n, h, w, cin), float("nan"), device="cuda")
        jobs.append(dict(g=g, w=l.kernel.data, dy=dy, y=y, idx=idx, dx=dx, act=l.act, alpha=l.alpha))
        dy = dx
    return jobs


@pytest.mark.parametrize("rows", [False, True])
@pytest.mark.parametrize("n", [2, 5, 64, 1024, 1027])
def test_forward_with_the_chain_behind_it_equals_forward_then_chain(n, rows):
    from spectrum_vae_amd import ops
    model = _model()
    X = torch.from_numpy(data(2 * n + 3, seed=31 + n)).cuda()
    perm = torch.randperm(2 * n + 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(n))
    res = []
    for fused in (False, True):
        arr, outs = _chain(model, n)
        part = torch.full((ops.loss_workspace_floats(),), float("nan"), device="cuda")
        xb = torch.full((n, 32, 32, 3), float("nan"), device="cuda")
        if not rows:
            ops.gather_rows(X, perm[:n], xb)
        d = torch.full_like(xb, float("nan"))
        jobs = _jobs(model, n, outs, d)
        table = ops.batch_rows(X, perm, xb) if rows else None
        if fused:
            assert ops.convnet_fwd_mse_dgrad_supported(arr, n, 32, 32, jobs)
            cnt = ops.convnet_fwd_mse_dgrad(arr, n, 32, 32, None if rows else xb, table, model._params, xb, d, 1.0, part, jobs)
        else:
            if rows:
                cnt = ops.convnet_fwd_mse_rows(arr, n, 32, 32, table, model._params, xb, d, 1.0, part)
            else:
                cnt = ops.convnet_fwd_mse(arr, n, 32, 32, xb, model._params, xb, d, 1.0, part)
            ops.conv2d_dgrad_chain(jobs)
        torch.cuda.synchronize()
        res.append((cnt, xb, d, part, outs, [j["dx"] for j in jobs]))
    (ca, xa, da, pa, oa, ga), (cb, xb_, db, pb, ob, gb) = res
    assert ca == cb and torch.equal(xa, xb_) and torch.equal(da, db) and torch.equal(pa[:ca], pb[:cb])
    for (ya, ia), (yb, ib) in zip(oa, ob):
        assert torch.equal(ya, yb) and (ia is None or torch.equal(ia, ib))
    for a, b in zip(ga, gb):
        assert not torch.isnan(a).any()
        assert torch.equal(a, b)


def test_jobs_the_forward_launch_cannot_carry_are_refused():
    from spectrum_vae_amd import autoencoder, ops
    model = _model()
    arr, outs = _chain(model, 8)
    d = torch.zeros(8, 32, 32, 3, device="cuda")
    jobs = _jobs(model, 8, outs, d)
    other = _jobs(model, 4, _chain(model, 4)[1], d[:4])
    assert not ops.convnet_fwd_mse_dgrad_supported(arr, 8, 32, 32, other)          # another batch size
    wide = autoencoder.CAEPadded(**dict(RUN_MODEL_DEFAULTS, initial_filters=16))     # not the default stack's kernel
    wide.build((None, 32, 32, 3))
    wide._materialize()
    arr16, _ = _chain(wide, 8)
    assert not ops.convnet_fwd_mse_dgrad_supported(arr16, 8, 32, 32, jobs)
    part = torch.zeros(ops.loss_workspace_floats(), device="cuda")
    with pytest.raises(RuntimeError, match="do not run behind this forward launch"):
        ops.convnet_fwd_mse_dgrad(arr, 8, 32, 32, d, None, model._params, d, torch.empty_like(d), 1.0, part, other)


@pytest.mark.parametrize("graphs", [True, False])
def test_fit_with_the_chain_in_the_forward_launch_equals_fit_with_separate_launches(graphs):
    """Shuffled epochs with a ragged last batch: identical losses and parameters whether the forward launch carries the
    input-gradient chain, the chain is a launch of its own, or every layer's input gradient is."""
    from spectrum_vae_amd import losses
    from spectrum_vae_amd.optimizers import Adam
    x = data(200, seed=43)
    res = {}
    for mode in ("fused", "chain", "layers"):
        model = _model()
        model.use_graphs = graphs
        model.fuse_fwd_dgrad = mode == "fused"
        model.chain_dgrad = mode != "layers"
        model.compile(optimizer=Adam(1e-3), loss=losses.MeanSquaredError())
        h = model.fit(x, x, epochs=4, batch_size=64, shuffle=True, verbose=0, seed=9)
        res[mode] = (np.array(h.history["loss"]), model._params.cpu().numpy().copy(), bool(model._dgrad_plans))
    assert res["fused"][2] and res["chain"][2] and not res["layers"][2]
    for mode in ("chain", "layers"):
        assert np.array_equal(res["fused"][0], res[mode][0])
        assert np.array_equal(res["fused"][1], res[mode][1])


def test_conv_runs_around_a_dense_bottleneck_go_as_one_forward_launch_each():
    """CAEPadded with latent_dimensions > 0 (the constructor default of autoencoder.py:410-428) is not one conv chain: its
    encoder and decoder convs are runs inside the Sequentials, each issued as one forward launch; predictions, losses and
    weights equal the layer-by-layer form's to float32 round-off (other kernels, another summation order)."""
    from spectrum_vae_amd import autoencoder, losses
    from spectrum_vae_amd.optimizers import Adam
    x = data(96, seed=47)
    res = {}
    for fused in (True, False):
        model = autoencoder.CAEPadded(**dict(RUN_MODEL_DEFAULTS, latent_dimensions=8, fc_layers=1, fc_layer_size=32, dropout=0.0))
        model.build((None, 32, 32, 3))
        model._materialize()
        model.use_net_fused = fused
        assert model._conv_chain() is None
        runs = model.encoder._conv_runs()
        assert len(runs) == 1 and len(next(iter(runs.values()))[1]) == 3
        model.compile(optimizer=Adam(1e-3), loss=losses.MeanSquaredError())
        h = model.fit(x, x, epochs=2, batch_size=32, shuffle=False, verbose=0)
        res[fused] = (np.array(h.history["loss"]), model._params.cpu().numpy().copy(), model.predict(x[:8]))
    assert bool(model.encoder.__dict__.get("_run_plans")) is False and res[True][0].shape == res[False][0].shape
    assert np.allclose(res[True][0], res[False][0], rtol=1e-5)
    assert np.allclose(res[True][2], res[False][2], rtol=1e-4, atol=1e-5)
    assert np.abs(res[True][1] - res[False][1]).max() < 1e-4

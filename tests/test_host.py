"""CPU-only: the C-ABI library loads and exports every symbol the header declares, the host-side mirror of the
reference interface (shape tables, schedules, callbacks, data preparation, planner) behaves as specified, and
the data-parallel exchange is exact at world_size 2 over gloo.  No kernel is launched here."""
import ctypes
import math
import os
import re
import socket

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "spectrum_hip.h")
GOLD = os.path.join(os.path.dirname(__file__), "golden", "dataprep_golden.npz")

RUN_MODEL_DEFAULTS = dict(conv_layers=3, conv_layer_multiplier=2, initial_filters=8, activation="elu", kernel_size=3,
                          conv_operations=1, fc_layers=0, fc_layer_size=256, dropout=0.2, latent_dimensions=0,
                          batch_norm=False)


# ------------------------------------------------------------------------------------------------ the boundary
def _header_decls():
    """name -> number of parameters, for every function the header declares."""
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    out = {}
    for m in re.finditer(r"\b(sv_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[name] = n
    return out


def test_header_and_binding_table_agree():
    from spectrum_vae_amd import _lib
    decls = _header_decls()
    assert len(decls) >= 40
    assert set(decls) == set(_lib.SIGNATURES), (set(decls) ^ set(_lib.SIGNATURES))
    for name, n in decls.items():
        assert len(_lib.SIGNATURES[name][1]) == n, name


def test_library_exports_every_declared_symbol():
    from spectrum_vae_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _header_decls():
        assert hasattr(lib, name), name
    lib.sv_version.restype = ctypes.c_int
    assert lib.sv_version() >= 1
    lib.sv_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.sv_last_error(), bytes)


def test_every_entry_point_cites_the_reference_or_says_it_has_no_counterpart():
    src = open(HEADER).read()
    assert src.count(".py:") >= 20  # file:line citations of the reference interface
    assert 'extern "C"' in src


def test_argument_errors_are_reported_without_a_gpu():
    """Null / malformed geometry is rejected on the host before any launch."""
    from spectrum_vae_amd import _lib
    g = _lib.ConvGeom(N=1, H=4, W=4, Cin=1, KH=3, KW=3, SH=0, SW=1, pad_t=1, pad_l=1, Ho=4, Wo=4, Cout=1, up_h=1, up_w=1)
    rc = _lib.lib.sv_conv2d_fwd(ctypes.byref(g), None, None, None, None, 0, 0.0, 0, None)
    assert rc != 0
    assert len(_lib.last_error()) > 0
    with pytest.raises(_lib.SpectrumHipError):
        _lib.check(rc, "sv_conv2d_fwd")


def test_fused_support_query_is_host_only():
    from spectrum_vae_amd import ops
    g = ops.conv_geom(1024, 32, 32, 3, 3, 3, 1, 1, "same", 8)
    assert ops.conv2d_fused_supported(g, True)
    g2 = ops.conv_geom(4, 32, 32, 3, 4, 4, 2, 2, "same", 8)  # strided 4x4: outside the fused family
    assert not ops.conv2d_fused_supported(g2, False)


# ------------------------------------------------------------------------------------------------ geometry / tables
def test_same_padding_matches_oracle():
    from oracle import np_ops
    from spectrum_vae_amd import ops
    for n in (1, 4, 7, 32, 33):
        for k in (1, 2, 3, 4, 5):
            for s in (1, 2):
                assert tuple(ops.same_pads(n, k, s)) == tuple(np_ops.same_pads(n, k, s)), (n, k, s)


@pytest.mark.parametrize("cfg,want", [(RUN_MODEL_DEFAULTS, 12035), ({}, None)])
def test_caepadded_parameter_table(cfg, want):
    from oracle import models as omodels
    from oracle.torch_ref import RefSequential
    from spectrum_vae_amd import autoencoder
    m = autoencoder.CAEPadded(**cfg)
    m.build_graph((1, 32, 32, 3))
    enc, dec = omodels.caepadded(**cfg)
    ref = RefSequential(enc + dec, (32, 32, 3))
    n_train = sum(v.size for v in m.trainable_variables)
    assert n_train == ref.num_params()
    if want is not None:
        assert n_train == want
    assert m.output_shape_ == (32, 32, 3)
    lines = []
    m.summary(print_fn=lines.append)
    assert any("Total params" in l for l in lines)


def test_other_model_tables():
    from spectrum_vae_amd import autoencoder, context_encoder, unet
    u = unet.ContextEncoder()
    u.build_graph((1, 32, 32, 1))  # one input channel: the masked genotype image (run_unet.py)
    assert sum(v.size for v in u.trainable_variables) == 133379
    ce = context_encoder.ContextEncoder()
    ce.build_graph((1, 32, 32, 3))
    assert sum(v.size for v in ce.trainable_variables) == 4497899
    d = context_encoder.Discriminator()
    d.build_graph((1, 32, 32, 3))
    assert sum(v.size for v in d.trainable_variables) == 202497
    c1 = autoencoder.CAE1D(conv_layers=2, conv_operations=1, latent_dimensions=0)
    c1.build_graph((1, 32, 32, 3))
    assert len(c1.output_shape_) == 3
    with pytest.raises(ValueError):  # default depth pools a 32-wide row below one column
        autoencoder.CAE1D().build_graph((1, 32, 1, 3))


def test_model_refuses_to_run_unbuilt_and_wrong_shapes():
    from spectrum_vae_amd import autoencoder
    m = autoencoder.CAEPadded(**RUN_MODEL_DEFAULTS)
    with pytest.raises(ValueError):
        m.summary()
    with pytest.raises(ValueError):
        m._materialize()
    m.build_graph((1, 32, 32, 3))
    with pytest.raises(RuntimeError):
        m.fit(np.zeros((4, 32, 32, 3), np.float32))  # not compiled


# ------------------------------------------------------------------------------------------------ schedules / callbacks
def test_exponential_decay_matches_oracle():
    from oracle import np_ops
    from spectrum_vae_amd.optimizers import ExponentialDecay
    for stair in (False, True):
        s = ExponentialDecay(5e-4, 10, 0.96, staircase=stair)
        for step in (0, 1, 9, 10, 11, 99, 100, 1234):
            assert math.isclose(s(step), np_ops.exponential_decay(5e-4, step, 10, 0.96, stair), rel_tol=1e-12)
    with pytest.raises(ValueError):
        ExponentialDecay(1e-3, 0, 0.9)


def test_early_stopping_patience():
    from spectrum_vae_amd.model import EarlyStopping
    es = EarlyStopping(monitor="val_loss", patience=2)
    es.on_train_begin()
    seq = [1.0, 0.9, 0.95, 0.97, 0.5]
    stops = [es.on_epoch_end(i, {"val_loss": v}) for i, v in enumerate(seq)]
    assert stops == [False, False, False, True, False]
    assert es.stopped_epoch == 3
    assert es.on_epoch_end(0, {"loss": 1.0}) is False  # monitored key absent


def test_run_model_host_helpers():
    from spectrum_vae_amd import run_model
    rng = np.random.default_rng(3)
    cfg = run_model.sample_config(rng)
    assert isinstance(cfg, dict) and len(cfg) > 0
    neg, pos = np.array([0.1, 0.2, 0.3, 0.4]), np.array([0.35, 0.5, 0.6, 0.7])
    auc = run_model.roc_auc(neg, pos)
    # label 1 = the first (normal) set, score = reconstruction loss (run_model.py:390-398): one of the 16
    # (normal, mutator) pairs has the normal sample scoring higher
    assert math.isclose(auc, 1 / 16)
    assert math.isclose(run_model.roc_auc(pos, neg), 15 / 16)
    x = np.zeros((2, 30, 32, 3), np.float32)
    assert run_model.pad_height(x, 3).shape[1] % 8 == 0


def test_roc_auc_equals_sklearn_roc_curve_auc():
    """run_model.py:390-398 calls sklearn.metrics.roc_curve + auc (label 1 = the normal set, score = loss); the rank-sum
    form must give the same area, ties included."""
    sklearn_metrics = pytest.importorskip("sklearn.metrics")
    from spectrum_vae_amd import run_model
    rng = np.random.default_rng(11)
    for n1, n0, levels in ((1, 1, 3), (7, 5, 4), (200, 333, 50), (64, 64, 0)):
        a = rng.random(n1) if levels == 0 else rng.integers(0, levels, n1).astype(np.float64) / levels
        b = rng.random(n0) if levels == 0 else rng.integers(0, levels, n0).astype(np.float64) / levels
        y_true = np.concatenate((np.ones(n1), np.zeros(n0)))
        if len(np.unique(y_true)) < 2:
            continue
        fpr, tpr, _ = sklearn_metrics.roc_curve(y_true, np.concatenate((a, b)))
        assert abs(run_model.roc_auc(a, b) - sklearn_metrics.auc(fpr, tpr)) < 1e-12, (n1, n0, levels)


# ------------------------------------------------------------------------------------------------ data preparation
def test_package_dataprep_matches_reference_golden_vectors():
    from spectrum_vae_amd import data
    g = np.load(GOLD)
    for ci in range(int(g["n_cases"])):
        X, pos = g[f"X_{ci}"], g[f"pos_{ci}"]
        region = data.process_region(X.copy(), pos.copy())
        assert region.dtype == np.float32 and np.array_equal(region, g[f"region_{ci}"]), ci
        assert np.array_equal(data.inter_snp_distances(pos), g[f"dist_{ci}"]), ci
        hap = np.sum(np.transpose(X, (1, 0, 2)), axis=2)
        assert np.array_equal(data.major_minor(hap), g[f"majmin_{ci}"]), ci
        perm = data.sort_min_diff(g[f"region_{ci}"][:, :, 0])
        assert sorted(np.asarray(perm).tolist()) == list(range(32))


def test_pack_regions_lays_ragged_regions_back_to_back():
    """Host packing for the device data-preparation call: offsets, dtypes, and the errors the reference raises."""
    from spectrum_vae_amd import data
    g = np.load(GOLD)
    regions = [(g[f"X_{ci}"], g[f"pos_{ci}"]) for ci in range(int(g["n_cases"]))]
    x, pos, off, n_haps, n_mut = data.pack_regions(regions)
    assert (n_haps, n_mut) == (32, 6) and x.dtype == np.float32 and pos.dtype == np.int64 and off.dtype == np.int64
    sizes = [r[0].shape[0] for r in regions]
    assert off.tolist() == np.concatenate(([0], np.cumsum(sizes))).tolist()
    assert x.size == sum(sizes) * 32 * 6 and pos.size == sum(sizes)
    k = 3
    assert np.array_equal(x[off[k] * 192:off[k + 1] * 192].reshape(sizes[k], 32, 6), regions[k][0])
    assert np.array_equal(pos[off[k]:off[k + 1]], regions[k][1])
    with pytest.raises(ValueError):
        data.pack_regions([])
    with pytest.raises(ValueError):
        data.pack_regions([(np.zeros((5, 32, 6)), np.arange(4))])
    with pytest.raises(ValueError):
        data.pack_regions([(np.zeros((0, 32, 6)), np.arange(0))])
    with pytest.raises(ValueError):
        data.pack_regions([(np.zeros((5, 32, 6)), np.arange(5)), (np.zeros((5, 16, 6)), np.arange(5))])


def test_process_region_rejects_ragged_input():
    from spectrum_vae_amd import data
    with pytest.raises(ValueError):
        data.process_region(np.zeros((5, 32, 4)), np.arange(4))


def test_synthetic_training_tensor_format():
    from spectrum_vae_amd import data
    x = data.synth_training_tensor(6, seed=7)
    assert x.shape == (6, 32, 32, 3) and x.dtype == np.float32
    assert set(np.unique(x[..., 0])) <= {-1.0, 0.0, 1.0}
    assert (x[..., 1] == 0).all()
    assert np.array_equal(x, data.synth_training_tensor(6, seed=7))


# ------------------------------------------------------------------------------------------------ sharding + exchange
def test_shard_bounds_cover_equal_shards():
    from spectrum_vae_amd.parallel import shard_bounds
    for n, w in ((8192, 8), (1000, 3), (7, 2), (1, 2)):
        spans = [shard_bounds(n, r, w) for r in range(w)]
        assert all(b - a == n // w for a, b in spans)
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, out):
    """Each rank: oracle gradients on its shard -> GradReducer.all_reduce -> Keras-Adam step with the returned scale."""
    import torch.distributed as dist
    from oracle import models as omodels
    from oracle.torch_ref import KerasAdam, RefSequential
    from spectrum_vae_amd.parallel import GradReducer, shard_bounds
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)  # replicas start different on purpose; broadcast_ must fix that
        enc, dec = omodels.caepadded(**RUN_MODEL_DEFAULTS)
        ref = RefSequential(enc + dec, (32, 32, 3), dtype=torch.float64, seed=100 + rank)
        red = GradReducer()
        assert red.world == world
        flat = torch.cat([p.detach().reshape(-1) for p in ref.params])
        red.broadcast_(flat)
        off = 0
        for p in ref.params:
            p.data.copy_(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        X = torch.tensor(np.random.default_rng(5).standard_normal((8, 32, 32, 3)), dtype=torch.float64)
        a, b = shard_bounds(X.shape[0], rank, world)
        opt = KerasAdam(ref.params, 1e-3)
        for _ in range(3):
            for p in ref.params:
                p.grad = None
            pred = ref(X[a:b], training=True)
            ((pred - X[a:b]) ** 2).mean().backward()
            g = torch.cat([p.grad.reshape(-1) for p in ref.params])
            scale = red.all_reduce(g)
            g *= scale
            off = 0
            for p in ref.params:
                p.grad.copy_(g[off:off + p.numel()].view_as(p))
                off += p.numel()
            opt.apply_gradients([p.grad for p in ref.params])
        assert red.calls == 3 and red.bytes == 3 * g.numel() * 8
        torch.save(torch.cat([p.detach().reshape(-1) for p in ref.params]), os.path.join(out, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_data_parallel_world2_equals_single_process_full_batch(tmp_path):
    """Two gloo ranks at batch B each == one process at batch 2B (mean-type loss, summed gradients / world)."""
    import torch.multiprocessing as mp
    from oracle import models as omodels
    from oracle.torch_ref import KerasAdam, RefSequential
    port = _free_port()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"), weights_only=True)
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"), weights_only=True)
    assert torch.equal(r0, r1)  # replicas stay bit-identical
    enc, dec = omodels.caepadded(**RUN_MODEL_DEFAULTS)
    ref = RefSequential(enc + dec, (32, 32, 3), dtype=torch.float64, seed=100)
    X = torch.tensor(np.random.default_rng(5).standard_normal((8, 32, 32, 3)), dtype=torch.float64)
    opt = KerasAdam(ref.params, 1e-3)
    for _ in range(3):
        for p in ref.params:
            p.grad = None
        ((ref(X, training=True) - X) ** 2).mean().backward()
        opt.apply_gradients([p.grad for p in ref.params])
    single = torch.cat([p.detach().reshape(-1) for p in ref.params])
    assert (r0 - single).abs().max().item() < 1e-9


def _syncbn_worker(rank, world, port, out):
    """Each rank: the moments of its shard -> GradReducer.combine_moments; a small sum through all_reduce_small."""
    import torch.distributed as dist
    from spectrum_vae_amd.parallel import GradReducer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X = torch.tensor(np.random.default_rng(11).standard_normal((2, 12, 5)) * 3 + 100.0, dtype=torch.float64)
        shard = X[:, rank * 6:(rank + 1) * 6]  # [groups, rows, C]: every group's rows split over the ranks
        mom = torch.stack([torch.full((2, 5), 6.0, dtype=torch.float64), shard.mean(1),
                           ((shard - shard.mean(1, keepdim=True)) ** 2).sum(1)], dim=1)  # [groups, 3, C]
        red = GradReducer(sync_batchnorm=True)
        mean, invstd, var, total = red.combine_moments(mom, 1e-3)
        s = red.all_reduce_small(torch.full((3,), float(rank + 1)))
        torch.save({"mean": mean, "invstd": invstd, "var": var, "total": total, "sum": s}, os.path.join(out, f"bn{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_batchnorm_moments_of_two_ranks_combine_to_the_statistics_of_all_rows(tmp_path):
    """The exchange step of synchronised BatchNorm at world_size 2 over gloo: count-weighted means and deviations plus
    the between-rank term reproduce mean / biased variance of the undivided rows (data with mean 100, deviation 3: the
    form sum(x^2) / n - mean^2 would lose five digits in float32; this one is exact to float64 round-off)."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_syncbn_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "bn0.pt"), weights_only=True)
    r1 = torch.load(os.path.join(tmp_path, "bn1.pt"), weights_only=True)
    X = torch.tensor(np.random.default_rng(11).standard_normal((2, 12, 5)) * 3 + 100.0, dtype=torch.float64)
    for r in (r0, r1):
        assert torch.equal(r["total"], torch.full((2, 5), 12.0, dtype=torch.float64))
        assert (r["var"] - X.var(1, unbiased=False)).abs().max() < 1e-12
        assert (r["mean"].double() - X.mean(1)).abs().max() < 1e-5 and r["mean"].dtype == torch.float32
        assert (r["invstd"].double() - (X.var(1, unbiased=False) + 1e-3).rsqrt()).abs().max() < 1e-6
        assert torch.equal(r["sum"], torch.full((3,), 3.0))


def test_init_from_env_single_process_is_a_no_op(monkeypatch):
    from spectrum_vae_amd import parallel
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert parallel.init_from_env() == (0, 1, 0)
    red = parallel.GradReducer()
    g = torch.ones(4)
    assert red.all_reduce(g) == 1.0 and torch.equal(g, torch.ones(4))

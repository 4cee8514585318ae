"""GPU parity of the data-preparation kernels, through the C ABI.

This part of the path is PINNED: the expected outputs in tests/golden/dataprep_golden.npz were produced by the
reference's own util module (tests/golden/make_dataprep_golden.py).  The bar is bit-exact: float32 images, float64
distance vectors, int64 row orders.  Beyond the golden cases the kernels are compared, bit for bit, with the NumPy
oracle (itself pinned by the same golden vectors) on ragged random batches that cover the reference's branches:
regions shorter than / equal to / longer than 32 sites, odd and even site counts, one- and two-site regions (the
0/0 of the reference's normalisation), sites where the derived allele is the majority, sites with no derived allele.
"""
import os

import numpy as np
import pytest
import torch

from oracle import dataprep as oracle_prep

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "dataprep_golden.npz")


def _device_batch(regions, **kw):
    from spectrum_vae_amd import data
    x, pos, off, n_haps, n_mut = data.pack_regions(regions)
    to = lambda a: torch.from_numpy(a).cuda()
    return data.process_regions_device(to(x), to(pos), to(off), n_haps, n_mut, **kw)


def _same_bits(a: np.ndarray, b: np.ndarray):
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b, equal_nan=True) and \
        np.array_equal(np.signbit(a), np.signbit(b))


def test_process_regions_matches_the_reference_golden_vectors():
    g = np.load(GOLD)
    n = int(g["n_cases"])
    regions = [(g[f"X_{ci}"], g[f"pos_{ci}"]) for ci in range(n)]
    out, dist = _device_batch(regions, return_distances=True)
    out, dist = out.cpu().numpy(), dist.cpu().numpy()
    o = 0
    for ci in range(n):
        assert _same_bits(out[ci], g[f"region_{ci}"]), ci
        k = g[f"pos_{ci}"].shape[0]
        assert _same_bits(dist[o:o + k], g[f"dist_{ci}"]), ci
        o += k
        # major_minor of the whole matrix, where the window shows it (regions shorter than 32 sites are shown whole)
        mm = g[f"majmin_{ci}"]
        if k < 32:
            i = 16 - k // 2
            assert np.array_equal(out[ci][:, i:i + k, 0], mm.astype(np.float32)), ci


def test_sort_min_diff_matches_the_reference_golden_vectors():
    """The reference's tie order comes from scikit-learn's partial sort and is not specified; what is specified - the
    start row and the sequence of distances along the returned order - must agree with the golden orders, and the
    order itself must equal the index-stable order of the oracle."""
    from spectrum_vae_amd import data
    g = np.load(GOLD)
    n = int(g["n_cases"])
    imgs = np.stack([g[f"region_{ci}"] for ci in range(n)])
    order = data.sort_min_diff_device(torch.from_numpy(imgs).cuda()).cpu().numpy()
    assert order.dtype == np.int64
    for ci in range(n):
        D = oracle_prep.manhattan_rows(imgs[ci][:, :, 0])
        start = int(np.argmin(D.sum(axis=1)))
        ref = g[f"sort_{ci}"]
        assert sorted(order[ci].tolist()) == list(range(32))
        assert np.array_equal(D[start][order[ci]], D[start][ref]), ci
        assert np.array_equal(order[ci], oracle_prep.sort_min_diff(imgs[ci][:, :, 0])), ci


def _random_regions(rng, sizes, n_haps=32, n_mut=6):
    regions = []
    for n_sites in sizes:
        X, pos = oracle_prep.synth_genotypes(rng, int(n_sites), n_haps, n_mut)
        # force the rarely drawn branches: a majority-derived site and a site without derived alleles
        if n_sites >= 3:
            s = int(rng.integers(n_sites))
            X[s] = 0
            X[s, : n_haps // 2 + 1 + int(rng.integers(n_haps // 2 - 1)), int(rng.integers(n_mut))] = 1
            if n_sites >= 40:
                X[n_sites // 2 + 1] = 0
                X[0, 0, :] = 0
                X[0, 0, 0] = 1  # keeps the matrix maximum at 1 whatever the draw
        regions.append((X, pos))
    return regions


def test_process_regions_ragged_batch_matches_the_oracle_bit_for_bit():
    rng = np.random.default_rng(20240611)
    sizes = [1, 2, 3, 30, 31, 32, 33, 34, 63, 64, 65, 199, 200, 517] + list(rng.integers(1, 260, size=180))
    regions = _random_regions(rng, sizes)
    out, dist = _device_batch(regions, return_distances=True)
    out, dist = out.cpu().numpy(), dist.cpu().numpy()
    o = 0
    with np.errstate(all="ignore"):
        for r, (X, pos) in enumerate(regions):
            want = oracle_prep.process_region(X.copy(), pos.copy())
            assert _same_bits(out[r], want), (r, X.shape[0])
            assert _same_bits(dist[o:o + len(pos)], oracle_prep.inter_snp_distances(pos)), (r, X.shape[0])
            o += len(pos)


def test_process_regions_other_shapes():
    """16 haplotypes x 4 mutation types (the kernels take the shape from the arguments, not from constants)."""
    rng = np.random.default_rng(5)
    regions = _random_regions(rng, [5, 40, 77, 32], n_haps=16, n_mut=4)
    out = _device_batch(regions).cpu().numpy()
    assert out.shape == (4, 16, 32, 3)
    for r, (X, pos) in enumerate(regions):
        hap = X.sum(axis=2).T
        n = X.shape[0]
        if n >= 32:
            i = n // 2 - 16
            assert np.array_equal(out[r][:, :, 0], oracle_prep.major_minor(hap[:, i:i + 32]).astype(np.float32))
            assert np.array_equal(out[r][0, :, 2], oracle_prep.inter_snp_distances(pos)[i:i + 32].astype(np.float32))
        else:
            i = 16 - n // 2
            assert np.array_equal(out[r][:, i:i + n, 0], oracle_prep.major_minor(hap).astype(np.float32))
            assert (out[r][:, :i] == 0).all() and (out[r][:, i + n:] == 0).all()
        assert (out[r][:, :, 1] == 0).all()


def test_process_regions_raises_where_the_reference_asserts():
    from spectrum_vae_amd import _lib, data
    rng = np.random.default_rng(9)
    good = _random_regions(rng, [40, 50])
    two_types = _random_regions(rng, [45])[0]
    two_types[0][7, 3, :] = 0
    two_types[0][7, 3, 1] = 1
    two_types[0][7, 3, 4] = 1  # the summed matrix has a 2 (util.py:198)
    with pytest.raises(ValueError, match="at most one mutation type"):
        _device_batch(good + [two_types])
    empty = (np.zeros((40, 32, 6), np.float32), np.arange(40, dtype=np.int64) * 10)  # maximum 0, not 1
    with pytest.raises(ValueError, match="at most one mutation type"):
        _device_batch([empty])
    with pytest.raises(ValueError):
        data.pack_regions([(np.zeros((5, 32, 6)), np.arange(4))])  # util.py:176
    with pytest.raises(ValueError):
        data.pack_regions([(np.zeros((0, 32, 6)), np.arange(0))])
    with pytest.raises(_lib.SpectrumHipError):  # 128 haplotypes x 1024 sites do not fit the workgroup's LDS tile
        from spectrum_vae_amd import ops
        z = torch.zeros(8, device="cuda")
        ops.process_regions(z, z.long(), z.long(), 1, 128, 6, 1024, 3, 100, z, None, z.int())


def test_sort_batch_on_device_matches_the_host_path():
    from spectrum_vae_amd import data, run_context_encoder, run_model
    x = oracle_prep.synth_training_tensor(40, seed=3)
    x[5, 7] = x[5, 2]  # duplicate haplotypes: ties at distance 0
    x[6] = 0           # a constant image: every distance ties
    xd = torch.from_numpy(x).cuda()
    order = data.sort_min_diff_device(xd).cpu().numpy()
    for r in range(x.shape[0]):
        assert np.array_equal(order[r], oracle_prep.sort_min_diff(x[r][:, :, 0])), r
        assert np.array_equal(order[r], data.sort_min_diff(x[r][:, :, 0])), r
    got = data.sort_batch_device(xd).cpu().numpy()
    assert np.array_equal(got.astype(np.float64), run_model.sort_batch(x))
    assert np.array_equal(got.astype(np.float64), run_context_encoder.batch_sort(x))
    # other row counts / widths
    y = np.random.default_rng(1).integers(-1, 2, size=(7, 13, 20, 2)).astype(np.float32)
    o2 = data.sort_min_diff_device(torch.from_numpy(y).cuda(), channel=1).cpu().numpy()
    for r in range(7):
        assert np.array_equal(o2[r], oracle_prep.sort_min_diff(y[r][:, :, 1])), r

"""NumPy restatement of the reference's data-preparation path (util.py).

Test infrastructure (see ``oracle/__init__.py``).  PINNED: checked against
golden vectors minted from the reference's own ``util`` module by
``tests/golden/make_dataprep_golden.py``.
"""
from __future__ import annotations

import numpy as np

NUM_SNPS = 32  # global_vars.py:3
L = 100_000  # global_vars.py:4
NUM_CHANNELS = 3  # global_vars.py:6


def inter_snp_distances(positions: np.ndarray, norm_len: int = L) -> np.ndarray:
    """util.py:34-45: [0, gaps/norm_len], then min-max scaled to [-1, 1]."""
    positions = np.asarray(positions)
    n = positions.shape[0]
    d = np.zeros(n, dtype=np.float64)
    if n > 1:
        d[1:] = (positions[1:] - positions[:-1]) / norm_len
    lo, hi = d.min(), d.max()
    return 2.0 * (d - lo) / (hi - lo) - 1.0


def major_minor(matrix: np.ndarray) -> np.ndarray:
    """util.py:228-256 on an (n_haps, n_sites) 0/1 matrix: a site with no derived
    allele stays all zero; a site with more than n_haps/2 derived alleles is
    flipped; then 0 -> -1."""
    m = np.asarray(matrix, dtype=np.float64).copy()
    n_haps = m.shape[0]
    cnt = np.count_nonzero(m, axis=0)
    flip = cnt > (n_haps / 2)
    m[:, flip] = 1 - m[:, flip]
    live = cnt > 0
    sub = m[:, live]
    sub[sub == 0] = -1
    m[:, live] = sub
    return m


def process_region(X: np.ndarray, positions: np.ndarray) -> np.ndarray:
    """util.py:152-225: (n_sites, n_haps, 6) -> (n_haps, 32, 3) float32; channel 0
    = major/minor-polarised haplotypes of the central 32 sites (or all sites,
    centred, zero-padded), channel 1 untouched (zeros), channel 2 = inter-SNP
    distances tiled down the haplotypes."""
    n_sites, n_haps, _ = X.shape
    assert n_sites == positions.shape[0]
    mid = n_sites // 2
    half = NUM_SNPS // 2
    region = np.zeros((n_haps, NUM_SNPS, NUM_CHANNELS), dtype=np.float32)
    dist = inter_snp_distances(positions, L)
    hap = np.transpose(X, (1, 0, 2)).sum(axis=2)
    assert hap.max() == 1
    if mid >= half:
        i, j = mid - half, mid + half
        region[:, :, 0] = major_minor(hap[:, i:j])
        region[:, :, -1] = np.tile(dist[i:j], (n_haps, 1))
    else:
        other = half + 1 if n_sites % 2 == 1 else half
        i, j = half - mid, mid + other
        region[:, i:j, 0] = major_minor(hap)
        region[:, i:j, -1] = np.tile(dist, (n_haps, 1))
    return region


def manhattan_rows(X: np.ndarray) -> np.ndarray:
    Xc = np.asarray(X, dtype=np.float64).copy()
    Xc[Xc < 0] = 0
    return np.abs(Xc[:, None, :] - Xc[None, :, :]).sum(axis=2)


def sort_min_diff(X: np.ndarray) -> np.ndarray:
    """util.py:48-58: clamp negatives to 0, all-pairs Manhattan distance, pick the
    row whose distances to all rows sum to the minimum (first on ties) and return
    the row indices ordered by distance from it.  Equal distances are ordered by
    row index here; the reference inherits sklearn's unspecified tie order, so
    parity is defined on the distance sequence, not on the raw permutation."""
    D = manhattan_rows(X)
    smallest = int(np.argmin(D.sum(axis=1)))
    return np.lexsort((np.arange(D.shape[0]), D[smallest]))


def synth_genotypes(rng: np.random.Generator, n_sites: int, n_haps: int = 32, n_mut: int = 6):
    """Surrogate for generator.prep_simulated_region (msprime is unavailable):
    one-hot-by-mutation-type genotype array with a ~1/k site-frequency spectrum
    and sorted positions with exponential gaps inside [0, L)."""
    X = np.zeros((n_sites, n_haps, n_mut), dtype=np.float32)
    ks = np.arange(1, n_haps)
    p = (1.0 / ks) / np.sum(1.0 / ks)
    for s in range(n_sites):
        k = int(rng.choice(ks, p=p))
        X[s, rng.choice(n_haps, size=k, replace=False), int(rng.integers(n_mut))] = 1
    c = np.cumsum(rng.exponential(1.0, size=n_sites))
    pos = np.floor(c / c[-1] * (L - n_sites - 1)).astype(np.int64) + np.arange(n_sites)  # strictly increasing, < L
    return X, pos


def synth_training_tensor(n: int, seed: int = 1833, sort_rows: bool = False) -> np.ndarray:
    """(n, 32, 32, 3) float32 in the reference's training format
    (make_training_data.py:86 via util.process_region)."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, 32, NUM_SNPS, NUM_CHANNELS), dtype=np.float32)
    for i in range(n):
        X, pos = synth_genotypes(rng, int(rng.integers(40, 201)))
        r = process_region(X, pos)
        if sort_rows:
            r = r[sort_min_diff(r[:, :, 0])]
        out[i] = r
    return out

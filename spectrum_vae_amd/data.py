"""Host-side data preparation in the reference's format, and the synthetic surrogate for its simulator.

``process_region`` / ``major_minor`` / ``inter_snp_distances`` / ``sort_min_diff`` restate
util.py:34-58,152-256 (vectorised NumPy; checked against golden vectors minted from the reference's own
``util`` module).  ``synth_training_tensor`` replaces ``make_training_data.py`` + ``generator.py`` +
``simulation.py``, which need msprime/tskit: it draws genotype arrays with a neutral-like 1/k site
frequency spectrum and exponential inter-SNP gaps and pushes them through ``process_region``, so the
tensors have the reference's layout and value ranges: (N, 32, 32, 3) float32, channel 0 in {-1, 0, +1}
(minor allele = +1), channel 1 all zero, channel 2 inter-SNP distances in [-1, 1] tiled down the rows.
"""
from __future__ import annotations

import numpy as np

from . import global_vars


def inter_snp_distances(positions: np.ndarray, norm_len: int = global_vars.L) -> np.ndarray:
    """util.py:34-45."""
    positions = np.asarray(positions)
    d = np.concatenate(([0.0], np.diff(positions) / norm_len)) if positions.shape[0] > 0 else np.array([])
    return (2 * (d - np.min(d)) / (np.max(d) - np.min(d))) - 1


def major_minor(matrix: np.ndarray) -> np.ndarray:
    """util.py:228-256: per site, flip 0/1 when the derived allele is the majority; then 0 -> -1, except at
    sites without any derived allele, which stay all zero."""
    m = np.array(matrix, dtype=np.float64)
    counts = np.count_nonzero(m, axis=0)
    flip = counts > m.shape[0] / 2
    m[:, flip] = 1 - m[:, flip]
    seg = counts > 0
    m[:, seg] = np.where(m[:, seg] == 0, -1.0, m[:, seg])
    return m


def process_region(X: np.ndarray, positions: np.ndarray) -> np.ndarray:
    """util.py:152-225."""
    n_sites, n_haps, _ = X.shape
    if n_sites != positions.shape[0]:
        raise ValueError("one position per site is required")
    S = global_vars.NUM_SNPS
    mid, half = n_sites // 2, S // 2
    region = np.zeros((n_haps, S, global_vars.NUM_CHANNELS), dtype=np.float32)
    dist = inter_snp_distances(positions, global_vars.L)
    hap = X.sum(axis=2).T  # (n_haps, n_sites)
    if hap.max() != 1:
        raise ValueError("each (site, haplotype) must carry at most one mutation type")
    if mid >= half:
        i, j = mid - half, mid + half
        region[:, :, 0] = major_minor(hap[:, i:j])
        region[:, :, -1] = dist[i:j][None, :]
    else:
        i = half - mid
        j = mid + (half + 1 if n_sites % 2 == 1 else half)
        region[:, i:j, 0] = major_minor(hap)
        region[:, i:j, -1] = dist[None, :]
    return region


def sort_min_diff(X: np.ndarray) -> np.ndarray:
    """util.py:48-58: indices that order the rows by Manhattan distance (negatives clamped to 0) from the row
    whose summed distance to all rows is smallest.  Ties are broken by row index (the reference inherits an
    unspecified tie order from sklearn)."""
    Xc = np.array(X, dtype=np.float64)
    if Xc.ndim != 2:
        raise ValueError("sort_min_diff expects a 2-D (rows, sites) matrix")
    Xc[Xc < 0] = 0
    D = np.abs(Xc[:, None, :] - Xc[None, :, :]).sum(axis=2)
    start = int(np.argmin(D.sum(axis=1)))
    return np.lexsort((np.arange(D.shape[0]), D[start]))


# --------------------------------------------------------------------------- the same functions on the device
def pack_regions(regions):
    """Host-side packing of ragged regions for ``process_regions_device``: ``regions`` is a sequence of
    (X (n_sites, n_haps, n_mut), positions (n_sites,)) pairs.  Returns float32 genotypes back to back, int64
    positions, int64 offsets (len + 1).  Raises what the reference raises: ragged X / positions (util.py:176) and
    empty regions (``np.min`` of an empty distance vector, util.py:44)."""
    if len(regions) == 0:
        raise ValueError("no regions")
    n_haps, n_mut = regions[0][0].shape[1:]
    xs, ps, off = [], [], [0]
    for X, pos in regions:
        X, pos = np.asarray(X), np.asarray(pos)
        if X.ndim != 3 or X.shape[1:] != (n_haps, n_mut):
            raise ValueError("every region must be (n_sites, n_haps, n_mut) with the same n_haps and n_mut")
        if X.shape[0] != pos.shape[0]:
            raise ValueError("one position per site is required")
        if X.shape[0] == 0:
            raise ValueError("a region needs at least one site")
        xs.append(np.ascontiguousarray(X, dtype=np.float32).reshape(-1))
        ps.append(np.ascontiguousarray(pos, dtype=np.int64))
        off.append(off[-1] + X.shape[0])
    return np.concatenate(xs), np.concatenate(ps), np.asarray(off, dtype=np.int64), n_haps, n_mut


def process_regions_device(x, positions, offsets, n_haps: int, n_mut: int, return_distances: bool = False, check: bool = True):
    """util.process_region (util.py:152-225) for a packed batch of regions on the GPU: ``x`` / ``positions`` /
    ``offsets`` are device tensors laid out as ``pack_regions`` returns them.  Returns the (R, n_haps, NUM_SNPS,
    NUM_CHANNELS) float32 device tensor (and the float64 distance vectors of all sites when asked).  ``check``
    reads the status words back (one host sync) and raises where the reference's assertion (util.py:198) fails."""
    import torch
    from . import ops
    n_regions = offsets.numel() - 1
    out = torch.empty((n_regions, n_haps, global_vars.NUM_SNPS, global_vars.NUM_CHANNELS), dtype=torch.float32,
                      device=x.device)
    status = torch.empty(n_regions, dtype=torch.int32, device=x.device)
    dist = torch.empty(positions.numel(), dtype=torch.float64, device=x.device) if return_distances else None
    ops.process_regions(x, positions, offsets, n_regions, n_haps, n_mut, global_vars.NUM_SNPS, global_vars.NUM_CHANNELS,
                        global_vars.L, out, dist, status)
    if check and bool(status.any().item()):
        bad = torch.nonzero(status).flatten().tolist()
        raise ValueError(f"each (site, haplotype) must carry at most one mutation type (regions {bad[:8]})")
    return (out, dist) if return_distances else out


def sort_min_diff_device(images, channel: int = 0):
    """util.sort_min_diff (util.py:48-58) of every image of a (R, n_haps, n_snps, C) float32 device tensor:
    (R, n_haps) int64 row orders; ties keep row order, as in ``sort_min_diff`` above."""
    import torch
    from . import ops
    order = torch.empty(images.shape[:2], dtype=torch.int64, device=images.device)
    ops.sort_min_diff(images, channel, order)
    return order


def sort_batch_device(images, channel: int = 0):
    """run_model.py:22-29 / run_context_encoder.py:101-112 on the device: every image with its haplotype rows
    reordered by ``sort_min_diff_device``."""
    import torch
    from . import ops
    out = torch.empty_like(images)
    ops.reorder_rows(images, sort_min_diff_device(images, channel), out)
    return out


def synth_genotypes(rng: np.random.Generator, n_sites: int, n_haps: int = global_vars.NUM_HAPLOTYPES,
                    n_mut: int = 6):
    """A (n_sites, n_haps, 6) one-hot-by-mutation-type array and sorted positions in [0, L)."""
    ks = np.arange(1, n_haps)
    p = (1.0 / ks) / np.sum(1.0 / ks)
    k = rng.choice(ks, size=n_sites, p=p)
    order = np.argsort(rng.random((n_sites, n_haps)), axis=1)  # a random subset of size k per site
    derived = order < k[:, None]
    mut = rng.integers(n_mut, size=n_sites)
    X = np.zeros((n_sites, n_haps, n_mut), dtype=np.float32)
    s_idx, h_idx = np.nonzero(derived)
    X[s_idx, h_idx, mut[s_idx]] = 1
    c = np.cumsum(rng.exponential(1.0, size=n_sites))
    pos = np.floor(c / c[-1] * (global_vars.L - n_sites - 1)).astype(np.int64) + np.arange(n_sites)
    return X, pos


def synth_training_tensor(n: int, seed: int = global_vars.DEFAULT_SEED, sort_rows: bool = False) -> np.ndarray:
    """(n, 32, 32, 3) float32 training tensor in the reference's format (make_training_data.py:86)."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, global_vars.NUM_HAPLOTYPES, global_vars.NUM_SNPS, global_vars.NUM_CHANNELS), np.float32)
    for i in range(n):
        X, pos = synth_genotypes(rng, int(rng.integers(40, 201)))
        r = process_region(X, pos)
        if sort_rows:
            r = r[sort_min_diff(r[:, :, 0])]
        out[i] = r
    return out

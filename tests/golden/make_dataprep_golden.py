"""Mint golden vectors for the data-preparation path from the REFERENCE itself.

Run in the build container only (``/root/reference`` does not exist on the GPU
box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_dataprep_golden.py

It imports the reference's ``util`` / ``global_vars`` modules (pure NumPy +
sklearn), feeds them seeded synthetic genotype arrays of the shape
``generator.prep_simulated_region`` produces - (n_sites, n_haps, 6) one-hot by
mutation type, plus sorted integer positions - and stores inputs and outputs
in ``dataprep_golden.npz`` next to this file.  Only data is stored.
"""
import os
import sys

import numpy as np

REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import global_vars  # noqa: E402
import util  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dataprep_golden.npz")


def synth_region(rng, n_sites, n_haps=32, n_mut=6):
    """(n_sites, n_haps, 6) with, per site, derived alleles on a random subset of
    haplotypes, all in one random mutation-type channel; positions sorted, < L."""
    X = np.zeros((n_sites, n_haps, n_mut), dtype=np.float32)
    for s in range(n_sites):
        k = int(rng.integers(1, n_haps))  # 1 .. n_haps-1 derived copies
        haps = rng.choice(n_haps, size=k, replace=False)
        X[s, haps, int(rng.integers(n_mut))] = 1
    pos = np.sort(rng.choice(global_vars.L, size=n_sites, replace=False)).astype(np.int64)
    return X, pos


def main():
    rng = np.random.default_rng(global_vars.DEFAULT_SEED)
    out = {}
    cases = [50, 200, 33, 32, 31, 20, 7, 64, 41]
    out["n_cases"] = np.array(len(cases))
    for ci, n_sites in enumerate(cases):
        X, pos = synth_region(rng, n_sites)
        out[f"X_{ci}"] = X.copy()
        out[f"pos_{ci}"] = pos.copy()
        region = util.process_region(X.copy(), pos.copy())
        out[f"region_{ci}"] = region
        out[f"dist_{ci}"] = util.inter_snp_distances(pos.copy(), global_vars.L)
        hap = np.sum(np.transpose(X, (1, 0, 2)), axis=2)
        out[f"majmin_{ci}"] = util.major_minor(hap.copy())
        out[f"sort_{ci}"] = util.sort_min_diff(region[:, :, 0].copy())
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items() if k.endswith("_0")})


if __name__ == "__main__":
    main()

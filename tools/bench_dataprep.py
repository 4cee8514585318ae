#!/usr/bin/env python3
"""Rates of the device data-preparation kernels against the HBM roofline, with the NumPy path timed beside them.

  python tools/bench_dataprep.py [--regions 16384] [--reps 20]

Workload: ragged regions in the reference's simulator format, (n_sites, 32, 6) float32 one-hot genotypes + int64
positions, n_sites uniform in [40, 200] (make_training_data.py draws regions of this order), packed back to back
and resident in HBM before the timed region.  Algorithmic bytes of process_regions = genotypes + positions read once
+ images written once; of sort_min_diff = the image's haplotype channel read once + the order written; of
reorder_rows = images read + written once.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from spectrum_vae_amd import data, ops  # noqa: E402

HBM_PEAK = 8.0e12


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) * 1e-3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--regions", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cpu-regions", type=int, default=512)
    a = ap.parse_args()
    rng = np.random.default_rng(1833)
    R, H, M = a.regions, 32, 6
    sizes = rng.integers(40, 201, size=R)
    off = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
    total = int(off[-1])
    dev = torch.device("cuda", 0)
    # synthetic genotypes straight on the device: one mutation type per site, each haplotype derived with p = 0.3
    g = torch.Generator(device=dev).manual_seed(1833)
    derived = torch.rand((total, H), device=dev, generator=g) < 0.3
    derived[:, 0] = True  # every site segregates, the matrix maximum is 1
    mut = torch.randint(M, (total,), device=dev, generator=g)
    x = torch.zeros((total, H, M), device=dev)
    x.scatter_(2, mut[:, None, None].expand(total, H, 1), derived[:, :, None].float())
    gaps = torch.randint(1, 600, (total,), device=dev, generator=g)
    pos = torch.cumsum(gaps, 0)
    offd = torch.from_numpy(off).to(dev)
    x = x.reshape(-1).contiguous()

    out = {}
    img = data.process_regions_device(x, pos, offd, H, M)  # also checks the status words once
    t = timed(lambda: data.process_regions_device(x, pos, offd, H, M, check=False), a.reps)
    nbytes = x.numel() * 4 + pos.numel() * 8 + img.numel() * 4
    out["process_regions"] = {"regions_per_s": round(R / t, 1), "us": round(t * 1e6, 1), "GB_per_s": round(nbytes / t / 1e9, 1),
                              "hbm_frac": round(nbytes / t / HBM_PEAK, 3), "bytes": nbytes}
    order = data.sort_min_diff_device(img)
    t = timed(lambda: data.sort_min_diff_device(img), a.reps)
    nb = img.numel() // 3 * 4 + order.numel() * 8
    out["sort_min_diff"] = {"images_per_s": round(R / t, 1), "us": round(t * 1e6, 1), "GB_per_s": round(nb / t / 1e9, 1),
                            "hbm_frac": round(nb / t / HBM_PEAK, 3)}
    dst = torch.empty_like(img)
    t = timed(lambda: ops.reorder_rows(img, order, dst), a.reps)
    nb = img.numel() * 8 + order.numel() * 8
    out["reorder_rows"] = {"images_per_s": round(R / t, 1), "us": round(t * 1e6, 1), "GB_per_s": round(nb / t / 1e9, 1),
                           "hbm_frac": round(nb / t / HBM_PEAK, 3)}

    # what the memory system gives a plain streaming kernel on this box: a device-to-device copy of the genotype buffer
    # (read + write) and a read-only reduction of it
    cp = torch.empty_like(x)
    t = timed(lambda: cp.copy_(x), a.reps)
    out["reference_copy"] = {"GB_per_s": round(2 * x.numel() * 4 / t / 1e9, 1), "us": round(t * 1e6, 1)}
    del cp
    t = timed(lambda: x.max(), a.reps)
    out["reference_read_only_max"] = {"GB_per_s": round(x.numel() * 4 / t / 1e9, 1), "us": round(t * 1e6, 1)}

    # the NumPy path on the host cores (one thread: it is a per-region Python loop in the reference too)
    n_cpu = min(a.cpu_regions, R)
    xs = x.reshape(total, H, M)[: int(off[n_cpu])].cpu().numpy()
    ps = pos[: int(off[n_cpu])].cpu().numpy()
    t0 = time.perf_counter()
    imgs = [data.process_region(xs[off[r]:off[r + 1]], ps[off[r]:off[r + 1]]) for r in range(n_cpu)]
    t1 = time.perf_counter()
    for r in range(min(n_cpu, 128)):
        data.sort_min_diff(imgs[r][:, :, 0])
    t2 = time.perf_counter()
    out["cpu_numpy"] = {"process_regions_per_s": round(n_cpu / (t1 - t0), 1),
                        "sort_min_diff_per_s": round(min(n_cpu, 128) / (t2 - t1), 1), "cores": 1,
                        "sample": f"{n_cpu} regions / {min(n_cpu, 128)} sorts"}
    same = all(np.array_equal(imgs[r], img[r].cpu().numpy()) for r in range(n_cpu))
    out["bit_identical_to_numpy_on_sample"] = bool(same)
    out["config"] = {"regions": R, "sites": total, "n_haps": H, "n_mut": M}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

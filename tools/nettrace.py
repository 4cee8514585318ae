#!/usr/bin/env python3
"""Phase timeline of the whole-network forward kernel (diagnostic; needs `make -C spectrum_vae_amd/csrc TRACE=1`)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("SPECTRUM_HIP_LIB", os.path.join(ROOT, "spectrum_vae_amd", "libspectrum_hip_trace.so"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from spectrum_vae_amd import _lib, autoencoder  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lib = C.CDLL(_lib.LIB_PATH)
lib.sv_debug_set_net_trace.argtypes = [C.c_void_p]
tr = torch.zeros(1024 * 32, dtype=torch.int64, device="cuda")
assert lib.sv_debug_set_net_trace(tr.data_ptr()) == 0
m = autoencoder.CAEPadded(**bench.RUN_MODEL_CFG)
m.build((None, 32, 32, 3))
m._materialize(torch.device("cuda"))
x = torch.rand(n, 32, 32, 3, device="cuda")
from spectrum_vae_amd import losses  # noqa: E402
m.compile(optimizer=None, loss=losses.MeanSquaredError())


def run():
    m.zero_grad()
    m.forward_raw(x, False)


for _ in range(3):
    run()
torch.cuda.synchronize()
tr.zero_()
torch.cuda.synchronize()
run()
torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(-1, 32)
t = t[t[:, 0] != 0]
t0 = t[:, 0].min()
names = ["start", "x staged"] + [f"L{l} {w}" for l in range(6) for w in ("ready", "computed", "copied")]
prev = None
for s, nm in enumerate(names):
    v = (t[:, s] - t0) / 100.0
    d = "" if prev is None else f"   (+{np.median(v - prev):6.2f})"
    print(f"{nm:14s} min {v.min():7.2f} med {np.median(v):7.2f} max {v.max():7.2f}{d}")
    prev = v
if t[:, 20].any():
    for s_, nm in ((20, "L1: next layer set up"), (21, "L1: copy-out issued")):
        v = (t[:, s_] - t0) / 100.0
        print(f"{nm:26s} med {np.median(v):7.2f}   after 'L1 computed': {np.median(v - (t[:, 6] - t0) / 100.0):6.2f}")
    for s_, nm in ((23, "filter loads issued"), (24, "arenas zeroed"), (25, "input in LDS"), (26, "filters in LDS")):
        v = (t[:, s_] - t[:, 0]) / 100.0
        print(f"{nm:26s} med {np.median(v):7.2f} after the workgroup's start")
print("workgroups:", t.shape[0])

"""Times the CAEPadded training step under different stream layouts (diagnostic, GPU only)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from spectrum_vae_amd import autoencoder, optimizers, ops, data

def run(tag, n_side, skip=None, batch=1024, steps=60):
    real = ops.conv2d_bwd
    def patched(g, x, w, dy, y, idx, dx, dw, db, act, alpha, ws):
        if skip == "wgrad" and dx is None: return
        if skip == "dgrad" and dw is None: return
        return real(g, x, w, dy, y, idx, dx, dw, db, act, alpha, ws)
    import spectrum_vae_amd.nn as nn
    nn.ops.conv2d_bwd = patched
    import bench
    m = autoencoder.CAEPadded(**bench.RUN_MODEL_CFG)
    m.build_graph((1, 32, 32, 3))
    m.compile(optimizer=optimizers.Adam(1e-3), loss="mse")
    m.use_side_stream = n_side > 0
    m.n_side_streams = max(n_side, 1)
    X = torch.rand(batch, 32, 32, 3, device="cuda")
    m._materialize(X.device)
    b = m._step_buffers(batch, (32, 32, 3)); b["x"].copy_(X)
    for _ in range(5): m.train_on_device_batch(b, True, 1.0)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps): m.train_on_device_batch(b, True, 1.0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / steps
    print(f"{tag:40s} {dt*1e6:8.1f} us/step", flush=True)
    nn.ops.conv2d_bwd = real

run("no side stream", 0)
run("1 side stream", 1)
run("3 side streams", 3)
run("1 side, wgrads skipped", 1, "wgrad")
run("1 side, dgrads skipped", 1, "dgrad")
run("0 side, wgrads skipped (both halves in one call not affected)", 0, "wgrad")

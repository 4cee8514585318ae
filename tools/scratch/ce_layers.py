import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import models as omodels
from oracle.torch_ref import RefSequential
from spectrum_vae_amd import context_encoder, mixed_precision, nn
from spectrum_vae_amd.data import synth_training_tensor
from spectrum_vae_amd.losses import to_device_f32
n = 16
x = synth_training_tensor(n, 1833)
mixed_precision.set_global_policy("mixed_bfloat16")
m = context_encoder.ContextEncoder()
mixed_precision.set_global_policy("float32")
m.build((None, 32, 32, 3)); m._materialize()
e, d = omodels.context_encoder()
ref = RefSequential(e + d, (32, 32, 3), torch.float64, lowp=True)
ref.load_params([v.numpy() for v in m.trainable_variables])
pred = m.forward_raw(to_device_f32(x), True)
ctx = m._calls[-1]
taps = {}
ref(torch.from_numpy(x).double(), training=True, taps=taps)
hip_layers = [l for l in m.encoder.layers + m.decoder.layers if not isinstance(l, nn.InputLayer)]
ref_layers = [(i, l) for i, l in enumerate(ref.layers)]
print(len(hip_layers), len(ref_layers))
for (i, rl), hl in zip(ref_layers, hip_layers):
    s = ctx.saved.get(id(hl))
    if s is None or "y" not in s:
        # BN saves x only; its output is the next layer's saved x
        print(i, rl["type"], type(hl).__name__, "no y saved"); continue
    got = s["y"].float().cpu().numpy().astype(np.float64)
    want = taps[i].detach().numpy()
    if got.shape != want.shape: got = got.reshape(want.shape)
    print(i, rl["type"], type(hl).__name__, got.shape, "relL2 %.5f max %.5f" % (np.linalg.norm(got-want)/np.linalg.norm(want), np.abs(got-want).max()/np.abs(want).max()))
    if "x" in s:
        gx = s["x"].float().cpu().numpy().astype(np.float64)
        if i > 0:
            wx = taps[i-1].detach().numpy()
            if gx.shape == wx.shape: print("      input relL2 %.5f" % (np.linalg.norm(gx-wx)/np.linalg.norm(wx)))

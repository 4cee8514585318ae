"""Measures what the bf16 policy costs in accuracy: U-Net, context encoder and discriminator forward / gradients under
``mixed_bfloat16`` against the float64 CPU restatement with the same float32 master weights, beside the float32 policy's
error on the same inputs.  Prints one JSON line per model (the numbers behind the tolerances of tests/test_gpu_bf16.py)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel(got, want, floor=1e-6):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.max(np.abs(got - want)) / max(np.max(np.abs(want)), floor))


def rel_l2(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))


def main():
    from oracle import models as omodels
    from oracle import torch_ref
    from oracle.torch_ref import RefSequential
    from spectrum_vae_amd import context_encoder, losses, mixed_precision, unet
    from spectrum_vae_amd.data import synth_training_tensor
    from spectrum_vae_amd.losses import to_device_f32
    n = int(os.environ.get("N", "16"))
    full = synth_training_tensor(n, 1833)

    def run(make_model, make_ref, x, tgt, rdt=torch.float64, make_ref_lowp=None):
        out = {}
        for policy in ("float32", "mixed_bfloat16", "mixed_bfloat16_vs_emulation"):
            emu = policy.endswith("emulation")
            if emu and make_ref_lowp is None:
                continue
            mixed_precision.set_global_policy("mixed_bfloat16" if emu else policy)
            model = make_model()
            mixed_precision.set_global_policy("float32")
            model.build((None,) + tuple(x.shape[1:]))
            model._materialize()
            ref = make_ref_lowp() if emu else make_ref()
            ref.load_params([v.numpy() for v in model.trainable_variables])
            xd = to_device_f32(x)
            model.zero_grad()
            pred = model.forward_raw(xd, True)
            loss = torch.empty(1, device="cuda")
            dpred = torch.empty_like(pred)
            losses.MeanSquaredError().value_and_grad(to_device_f32(tgt), pred, loss, dpred)
            model.backward_raw(dpred, -1, need_dx=False)
            for p in ref.params:
                p.grad = None
            pr = ref(torch.from_numpy(x).to(rdt), training=True)
            l64 = torch_ref.mse(torch.from_numpy(tgt).to(rdt), pr)
            l64.backward()
            g = [v.grad.detach().cpu().numpy() for v in model.trainable_variables]
            g64 = [p.grad.numpy() for p in ref.params]
            out[policy] = {
                "pred_max": rel(pred.float().cpu().numpy(), pr.detach().numpy()),
                "pred_l2": rel_l2(pred.float().cpu().numpy(), pr.detach().numpy()),
                "loss": abs(float(loss) - float(l64)) / abs(float(l64)),
                "grad_max_worst": max(rel(a, b, 1e-6 + 1e-3 * float(np.abs(b).max())) for a, b in zip(g, g64)),
                "grad_l2_worst": max(rel_l2(a, b) for a, b in zip(g, g64)),
                "grad_l2_all": rel_l2(np.concatenate([a.ravel() for a in g]), np.concatenate([b.ravel() for b in g64])),
            }
            if os.environ.get("PER_VAR"):
                out[policy]["per_var"] = {v.name + str(v.shape): round(rel_l2(a, b), 5)
                                          for v, a, b in zip(model.trainable_variables, g, g64)}
            model._calls.clear()
        return out

    def show(name, v):
        print(json.dumps({"model": name, "batch": n, **v}), flush=True)

    show("unet", run(unet.ContextEncoder, lambda: omodels.RefUNet(dtype=torch.float64), full[..., :1].copy(), full,
                     make_ref_lowp=lambda: omodels.RefUNet(dtype=torch.float64, lowp=True)))
    e, d = omodels.context_encoder()
    show("context_encoder", run(context_encoder.ContextEncoder, lambda: RefSequential(e + d, (32, 32, 3), torch.float64),
                                full, full, make_ref_lowp=lambda: RefSequential(e + d, (32, 32, 3), torch.float64, lowp=True)))
    # torch's float64 CPU conv backward refuses the one-filter head of the discriminator; its float32 restatement is
    # three orders of magnitude closer to the truth than bf16 is, which is all this comparison needs
    show("discriminator", run(context_encoder.Discriminator,
                              lambda: RefSequential(omodels.discriminator(), (32, 32, 3), torch.float32), full,
                              np.full((n, 1), 0.3, np.float32), torch.float32,
                              make_ref_lowp=lambda: RefSequential(omodels.discriminator(), (32, 32, 3), torch.float32,
                                                                  lowp=True)))


if __name__ == "__main__":
    main()

"""Counterpart of the reference's ``run_unet.py`` (pix2pix-style GAN driver) on the HIP engine.

Function names and argument meaning follow run_unet.py:25-214.  ``train_step`` (:164-203) takes the two
gradients the reference's two tapes take: d(disc_loss)/d(discriminator weights) and
d(gen_total_loss)/d(generator weights), the latter flowing back THROUGH the discriminator's second call
(with the discriminator weights of the forward pass) without touching the discriminator's gradients.
CIFAR-10 (run_unet.py:221) needs a download; ``main`` uses the haplotype channel of the synthetic
training tensor as the 1-channel input and the full tensor as the 3-channel target instead.
"""
from __future__ import annotations

import argparse
import os
from typing import Tuple

import numpy as np
import torch

from . import global_vars, ops, parallel, unet
from .losses import BinaryCrossentropy, MeanAbsoluteError, MeanSquaredError, to_device_f32
from .nn import as_dtensor, default_device, recording
from .optimizers import Adam, ExponentialDecay

# A/B switch: fork the two models' updates onto two streams inside the recorded step (default) or keep one chain
_OVERLAP = os.environ.get("SPECTRUM_GAN_OVERLAP", "1") != "0"


def build_model(input_shape: Tuple[int] = (1, global_vars.NUM_HAPLOTYPES, global_vars.NUM_SNPS,
                                           global_vars.INPUT_CHANNELS)):
    """run_unet.py:25-45."""
    model = unet.Generator(input_shape=input_shape)
    model.build_graph(tuple(input_shape))
    model.summary()
    return model


def build_optimizer(learning_rate: float = 1e-4, beta_1: float = 0.5, beta_2: float = 0.999,
                    decay_rate: float = 0.96, decay_steps: int = 10) -> Adam:
    """run_unet.py:48-66."""
    return Adam(ExponentialDecay(learning_rate, decay_steps=decay_steps, decay_rate=decay_rate, staircase=True),
                beta_1=beta_1, beta_2=beta_2)


def _t(x):
    return x.as_subclass(torch.Tensor)


def measure_disc_loss(disc_real_output, disc_generated_output, reduction: str = "auto"):
    """run_unet.py:69-94: BCE_logits(1, D(real)) + BCE_logits(0, D(generated))."""
    bce = BinaryCrossentropy(from_logits=True, reduction=reduction)
    return as_dtensor(_t(bce(1.0, disc_real_output)) + _t(bce(0.0, disc_generated_output)))


def measure_recon_loss(y_true, y_pred, reduction: str = "auto"):
    """run_unet.py:97-103."""
    return MeanSquaredError(reduction=reduction)(y_true, y_pred)


def measure_generator_loss(disc_generated_output, gen_output, y, lambda_weight: int = 100, reduction: str = "auto"):
    """run_unet.py:128-161: BCE_logits(1, D(generated)) + lambda * MAE(y, generated)."""
    gan_loss = BinaryCrossentropy(from_logits=True, reduction=reduction)(1.0, disc_generated_output)
    l1_loss = MeanAbsoluteError(reduction=reduction)(y, gen_output)
    total = as_dtensor(_t(gan_loss) + lambda_weight * _t(l1_loss))
    return total, gan_loss, l1_loss


class _StepState:
    def __init__(self, gen, disc, bs, device):
        if gen._act_dtype != disc._act_dtype:
            raise ValueError("generator and discriminator must be built under the same mixed-precision policy")
        adt = gen._act_dtype
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=device)  # noqa: E731
        a = lambda *s: torch.empty(s, dtype=adt, device=device)  # noqa: E731  activations / their gradients
        h, w, cx = gen.input_shape_
        cy = gen.output_shape_[-1]
        self.X, self.y = f(bs, h, w, cx), f(bs, h, w, cy)  # float32: what the caller hands over, and the loss targets
        self.Xa, self.ya = (self.X, self.y) if adt == torch.float32 else (a(bs, h, w, cx), a(bs, h, w, cy))
        # the discriminator sees [real pair; generated pair] as ONE batch of 2 bs (two BatchNorm statistics groups)
        self.cat_both = a(2 * bs, h, w, cx + cy)
        self.cat_real, self.cat_fake = self.cat_both[:bs], self.cat_both[bs:]
        self.d_both = a(2 * bs, 1)
        self.d_real, self.d_fake, self.d_gan = self.d_both[:bs], self.d_both[bs:], a(bs, 1)
        self.dgen = a(bs, h, w, cy)       # lambda * dMAE/dgen
        self.dgen_gan = a(bs, h, w, cy)   # gradient of the GAN term w.r.t. the generated image
        self.dX_unused = a(bs, h, w, cx)
        self.l_real, self.l_fake, self.l_gan, self.l_l1 = f(1), f(1), f(1), f(1)
        self.disc_loss, self.gen_total = f(1), f(1)
        self.graph = None
        self.warm = False
        self.side = torch.cuda.Stream(device=device)  # the generator update runs beside the discriminator's


def _step_state(gen, disc, gen_opt, disc_opt, bs: int, weight: float, device) -> _StepState:
    """The step state of (generator, discriminator, optimizers, batch size, lambda).  It lives ON the generator and
    holds references to the other three participants, so a recorded graph can never outlive - or be looked up for
    look-alikes of - the objects whose buffers it is bound to."""
    cache = gen.__dict__.setdefault("_gan_steps", {})
    key = (id(disc), id(gen_opt), id(disc_opt), int(bs), float(weight))
    st = cache.get(key)
    if st is None:
        st = cache[key] = _StepState(gen, disc, bs, device)
        st.partners = (disc, gen_opt, disc_opt)  # keeps the ids in the key from being reused while the entry exists
    return st


def _step_body(gen, disc, gen_opt, disc_opt, st: _StepState, lambda_weight: float, overlap: bool = False,
               defer_updates: bool = False):
    """One step of run_unet.py:164-203.  The discriminator runs ONCE on [[X, y]; [X, G(X)]] (2 B samples, two BatchNorm
    statistics groups: the moving statistics get the real batch's update first, as with two calls); the GAN term goes
    back to the generated image through the second group alone, the discriminator's own gradients come from one
    backward pass over both groups.
    ``defer_updates``: stop after the backward passes (the data-parallel step reduces the gradients and applies the
    updates itself).  ``overlap``: once the GAN term's gradient has come back through the discriminator, the generator
    backward + update and the discriminator backward + update touch disjoint buffers (each model owns its buffer pool,
    workspace and optimizer state), so the former is forked onto a second stream; at batch 32 neither fills the chip."""
    bce, mae = BinaryCrossentropy(from_logits=True), MeanAbsoluteError()
    n, h, w, cx = st.X.shape
    cy = st.y.shape[-1]
    rows = n * h * w
    gen.zero_grad_raw()
    disc.zero_grad_raw()
    if st.Xa is not st.X:  # bf16 policy: the models read bf16 copies of the inputs
        ops.cast(st.X, st.Xa)
        ops.cast(st.y, st.ya)
    ops.concat_channels_fwd(st.Xa, st.ya, st.cat_real, rows, cx, cy)
    gen_out = gen.forward_raw(st.Xa, True)
    ops.concat_channels_fwd(st.Xa, gen_out, st.cat_fake, rows, cx, cy)
    z = disc.forward_raw(st.cat_both, True, bn_groups=2)   # call 0
    z_real, z_fake = z[:n], z[n:]
    # generator loss: BCE(1, D(fake)) + lambda * MAE(y, gen_out)
    bce.value_and_grad(1.0, z_fake, st.l_gan, st.d_gan, 1.0)
    mae.value_and_grad(st.y, gen_out, st.l_l1, st.dgen, float(lambda_weight))
    # discriminator loss: BCE(1, D(real)) + BCE(0, D(fake))
    bce.value_and_grad(1.0, z_real, st.l_real, st.d_real, 1.0)
    bce.value_and_grad(0.0, z_fake, st.l_fake, st.d_fake, 1.0)
    ops.scalar_program([(st.gen_total, False, [(st.l_gan, 1.0), (st.l_l1, float(lambda_weight))]),
                        (st.disc_loss, False, [(st.l_real, 1.0), (st.l_fake, 1.0)])])
    # (1) GAN term back through the generated half of the discriminator's call to the generated image; the
    #     discriminator's own gradients are not touched (param_grads off), its weights are still those of the forward
    ctx_fake = disc._calls[0].group_view(1)
    ctx_fake.param_grads = False
    d_cat = disc.backward(ctx_fake, st.d_gan, need_dx=True)
    ops.concat_channels_bwd(d_cat, st.dX_unused, st.dgen_gan, rows, cx, cy)

    def generator_update():  # (3) dL/dgen = lambda * dMAE + dGAN
        ops.axpy(st.dgen, st.dgen_gan, 1.0)
        gen.backward_raw(st.dgen, 0, need_dx=False)
        if not defer_updates:
            gen_opt.apply_model(gen, 1.0)

    main, joined = torch.cuda.current_stream(), None
    if overlap and not defer_updates:
        forked = torch.cuda.Event()
        forked.record(main)
        with torch.cuda.stream(st.side):
            st.side.wait_event(forked)
            generator_update()
            joined = torch.cuda.Event()
            joined.record(st.side)
    # (2) discriminator update: one backward pass over both groups
    disc.backward_raw(st.d_both, 0, need_dx=False)
    if not defer_updates:
        disc_opt.apply_model(disc, 1.0)
    if joined is not None:
        main.wait_event(joined)
    else:
        generator_update()
    gen._calls.clear()
    disc._calls.clear()


def train_step(generator, discriminator, ce_opt, disc_opt, X, y, lambda_weight: int = 100, use_graph: bool = True):
    """run_unet.py:164-203.  Returns (disc_loss, gen_total_loss, gen_gan_loss, gen_l1_loss)."""
    gen, disc = generator, discriminator
    device = default_device()
    Xd, yd = to_device_f32(X, device), to_device_f32(y, device)
    if not gen.built:
        gen.build(tuple(Xd.shape))
    if not disc.built:
        disc.build((Xd.shape[0], Xd.shape[1], Xd.shape[2], Xd.shape[3] + yd.shape[3]))
    gen._materialize(device)
    disc._materialize(device)
    st = _step_state(gen, disc, ce_opt, disc_opt, Xd.shape[0], lambda_weight, device)
    st.X.copy_(Xd)
    st.y.copy_(yd)
    if gen._reducer is not None or disc._reducer is not None:  # data-parallel: reduce both gradient buffers, then update
        parallel.run_dp_step(st, use_graph,
                             lambda: _step_body(gen, disc, ce_opt, disc_opt, st, lambda_weight, defer_updates=True),
                             [(disc, disc_opt), (gen, ce_opt)], device)
    elif not use_graph or not st.warm:
        _step_body(gen, disc, ce_opt, disc_opt, st, lambda_weight)
        st.warm = True
    else:
        if st.graph is None:
            torch.cuda.synchronize()
            st.graph = torch.cuda.CUDAGraph()
            with recording(st.graph):
                _step_body(gen, disc, ce_opt, disc_opt, st, lambda_weight, overlap=_OVERLAP)
        st.graph.replay()
    c = lambda t: as_dtensor(t.view(()).clone())  # noqa: E731
    return c(st.disc_loss), c(st.gen_total), c(st.l_gan), c(st.l_l1)


def format_cifar(data: np.ndarray) -> np.ndarray:
    """run_unet.py:206-214: per-image min-max scaling to [-1, 1]."""
    n_examples, _, _, n_channels = data.shape
    new_data = np.zeros((n_examples, 32, 32, n_channels))
    for i in np.arange(n_examples):
        img = data[i]
        new_data[i] = 2 * ((img - np.amin(img)) / (np.amax(img) - np.amin(img))) - 1
    return new_data


def main(n_train: int = 2048, epochs: int = 2, batch_size: int = 32, seed: int = global_vars.DEFAULT_SEED):
    """run_unet.py:217-332 without the plots and without the CIFAR download."""
    from .data import synth_training_tensor
    data = synth_training_tensor(n_train, seed)
    X_all, y_all = data[..., :1], data
    split = int(0.8 * n_train)
    X_train, y_train, X_test, y_test = X_all[:split], y_all[:split], X_all[split:], y_all[split:]
    model, disc = unet.Generator(), unet.Discriminator()
    LR = 2e-4
    ce_optimizer, disc_optimizer = build_optimizer(learning_rate=LR), build_optimizer(learning_rate=LR)
    res = []
    for epoch in range(epochs):
        for step in range(X_train.shape[0] // batch_size):
            sl = slice(batch_size * step, batch_size * (step + 1))
            disc_total_loss, gen_total_loss, gan_loss, l1_loss = train_step(model, disc, ce_optimizer,
                                                                            disc_optimizer, X_train[sl], y_train[sl])
        predictions = model.predict(X_test)
        disc_real = disc([X_test, y_test])
        disc_fake = disc([X_test, predictions])
        disc_loss_test = measure_disc_loss(disc_real, disc_fake)
        gen_loss_test, gan_loss_test, l1_loss_test = measure_generator_loss(disc_fake, predictions, y_test)
        print(f"epoch {epoch}: DISC LOSS {float(disc_total_loss):.5f} gen {float(gen_total_loss):.5f} "
              f"GAN LOSS {float(gan_loss):.5f} l1 {float(l1_loss):.5f} | val disc {float(disc_loss_test):.5f} "
              f"val gen {float(gen_loss_test):.5f}")
        res.append({"epoch": epoch, "disc": float(disc_total_loss), "gen": float(gen_total_loss),
                    "val_disc": float(disc_loss_test), "val_gen": float(gen_loss_test)})
    return res


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("-run_sweep", action="store_true")
    p.add_argument("-search_size", type=int, default=20)
    args = p.parse_args()
    main()

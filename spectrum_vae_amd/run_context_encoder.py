"""Counterpart of the reference's ``run_context_encoder.py`` (inpainting GAN driver) on the HIP engine.

Same function names and argument meaning as the reference (run_context_encoder.py:24-273); the
TensorFlow ``GradientTape`` of ``train_step`` (:225-260) is replaced by the engine's explicit
forward/backward calls, which compute exactly the two gradients the reference takes:
d(disc_loss)/d(discriminator weights) and d(l2_loss)/d(context-encoder weights).  ``total_loss`` is
computed and returned but, as in the reference, never differentiated.
"""
from __future__ import annotations

import argparse
import os
from typing import Tuple

import numpy as np
import torch

from . import context_encoder, global_vars, ops, parallel
from .data import sort_min_diff
from .losses import BinaryCrossentropy, MeanSquaredError, to_device_f32
from .nn import as_dtensor, default_device, recording
from .optimizers import Adam, ExponentialDecay

# A/B switch: fork the two models' updates onto two streams inside the recorded step (default) or keep one chain
_OVERLAP = os.environ.get("SPECTRUM_GAN_OVERLAP", "1") != "0"


# ------------------------------------------------------------------------------------------------ masks
def create_random_image_mask(X: np.ndarray, pct_mask: float = 0.25, rng=None) -> np.ndarray:
    """run_context_encoder.py:39-58: a fixed square [w/8 : 3w/8]^2 plus random (w/8)-sized squares until at
    least ``pct_mask`` of the pixels are masked.  ``rng`` (a numpy Generator) makes it reproducible; the
    reference draws from the global ``np.random`` state."""
    n_haps, n_snps, n_channels = X.shape
    mask = np.zeros(X.shape)
    seg = n_snps // 8
    mask[seg:seg * 3, seg:seg * 3, :] = 1
    draw = (lambda hi: int(rng.integers(hi))) if rng is not None else (lambda hi: np.random.randint(hi))
    while (np.sum(mask[:, :, 0]) / (n_snps ** 2)) < pct_mask:
        rx, ry = draw(n_snps - seg), draw(n_snps - seg)
        mask[rx:rx + seg, ry:ry + seg, :] = 1
    return mask


def batch_random_mask(X: np.ndarray, pct_mask: float = 0.25, rng=None) -> np.ndarray:
    """run_context_encoder.py:24-36."""
    mask_arr = np.zeros(X.shape)
    for i in np.arange(X.shape[0]):
        mask_arr[i] = create_random_image_mask(X[i], pct_mask=pct_mask, rng=rng)
    return mask_arr


def batch_random_mask_device(shape, pct_mask: float = 0.25, seed: int = global_vars.DEFAULT_SEED, offset: int = 0, device=None):
    """``batch_random_mask`` generated in HBM (one workgroup per image, the same block procedure with counter-hash
    draws): returns a float32 device tensor of ``shape`` (N, H, W, C).  The training loop slices it per batch, so the
    masks never cross PCIe."""
    t = torch.empty(tuple(int(v) for v in shape), dtype=torch.float32, device=device or default_device())
    ops.random_block_mask(t, pct_mask, seed, offset)
    return t


def create_image_mask(X: np.ndarray, pct_mask: float = 0.25):
    """run_context_encoder.py:61-79: zero the central square, return (masked image, removed square)."""
    n_haps = X.shape[0]
    i = int(n_haps * pct_mask)
    j = i * 3
    X_ = X.copy()
    y_ = X[i:j, i:j, :]
    X_[i:j, i:j, :] = 0
    return X_, y_


def batch_mask(X: np.ndarray, pct_mask: float = 0.25):
    """run_context_encoder.py:82-99."""
    batch_size, n_haps, n_snps, n_channels = X.shape
    assert n_haps == n_snps
    mask_dim = int(n_haps * pct_mask) * 2
    X_masked_batch = np.zeros(X.shape)
    X_true_batch = np.zeros((batch_size, mask_dim, mask_dim, n_channels))
    for i in np.arange(batch_size):
        X_masked_batch[i], X_true_batch[i] = create_image_mask(X[i])
    return X_masked_batch, X_true_batch


def batch_sort(X: np.ndarray) -> np.ndarray:
    """run_context_encoder.py:101-112: reorder each sample's haplotypes by genetic similarity (channel 0)."""
    X_new = np.zeros(X.shape)
    for b in np.arange(X.shape[0]):
        X_new[b] = X[b][sort_min_diff(X[b][:, :, 0]), :, :]
    return X_new


def pad_training_data(train_data: np.ndarray) -> np.ndarray:
    """run_context_encoder.py:263-273: zero-pad the height up to the width."""
    n, h, w, c = train_data.shape
    padding = max(w - h, 0)
    return np.concatenate((train_data, np.zeros((n, padding, w, c))), axis=1)


def format_cifar(data: np.ndarray) -> np.ndarray:
    """run_context_encoder.py:276-284: per-image min-max scaling to [-1, 1]."""
    new_data = np.zeros((data.shape[0], 32, 32, 3))
    for i in np.arange(data.shape[0]):
        img = data[i]
        new_data[i] = 2 * ((img - np.amin(img)) / (np.amax(img) - np.amin(img))) - 1
    return new_data


# ------------------------------------------------------------------------------------------------ model / optimizer
def build_model(input_shape: Tuple[int] = (4, global_vars.NUM_SNPS, global_vars.NUM_CHANNELS),
                activation: str = "elu", latent_dimensions: int = 8):
    """run_context_encoder.py:115-141."""
    model = context_encoder.ContextEncoder(input_shape=input_shape, latent_dimensions=latent_dimensions)
    model.build_graph((1, input_shape[0], input_shape[1], input_shape[2]))
    model.encoder.summary()
    model.decoder.summary()
    return model


def build_optimizer(learning_rate: float = 1e-4, beta_1: float = 0.5, beta_2: float = 0.999,
                    decay_rate: float = 0.96, decay_steps: int = 50) -> Adam:
    """run_context_encoder.py:144-162."""
    return Adam(ExponentialDecay(learning_rate, decay_steps=decay_steps, decay_rate=decay_rate, staircase=True),
                beta_1=beta_1, beta_2=beta_2)


# ------------------------------------------------------------------------------------------------ losses
def measure_disc_loss(discriminator, y_true, y_pred, reduction: str = "auto"):
    """run_context_encoder.py:165-187: mean of BCE(1, D(real)) and BCE(0, D(fake)); both discriminator calls
    run with ``training=True`` (batch statistics, moving-average updates), as in the reference."""
    bce = BinaryCrossentropy(from_logits=False, reduction=reduction)
    p_true = discriminator(y_true, training=True)
    p_fake = discriminator(y_pred, training=True)
    discriminator._calls.clear()  # evaluation only: nothing will be back-propagated through these calls
    lt, lf = bce(1.0, p_true), bce(0.0, p_fake)
    return as_dtensor((lt.as_subclass(torch.Tensor) + lf.as_subclass(torch.Tensor)) / 2.0)


def measure_recon_loss(y_true, y_pred, mask, reduction: str = "auto"):
    """run_context_encoder.py:190-198: MSE with ``sample_weight = mask[:, :, :, 0] * 10``."""
    m = to_device_f32(mask)
    w = (m[:, :, :, 0] * 10).contiguous()
    return MeanSquaredError(reduction=reduction)(y_true, y_pred, sample_weight=w)


def measure_total_loss(discriminator, y_true, y_pred, mask, l2_weight: float = 0.999, reduction: str = "auto"):
    """run_context_encoder.py:201-222 (the discriminator sees (y_pred, y_true) in that order there)."""
    d = measure_disc_loss(discriminator, y_pred, y_true, reduction=reduction)
    r = measure_recon_loss(y_true, y_pred, mask, reduction=reduction)
    return as_dtensor(r.as_subclass(torch.Tensor) * l2_weight + d.as_subclass(torch.Tensor) * (1 - l2_weight))


# ------------------------------------------------------------------------------------------------ training step
class _StepState:
    """Static device buffers and the captured HIP graph of one (generator, discriminator, batch size)."""

    def __init__(self, ce, disc, bs: int, device):
        if ce._act_dtype != disc._act_dtype:
            raise ValueError("context encoder and discriminator must be built under the same mixed-precision policy")
        adt = ce._act_dtype
        shp = (bs,) + ce.input_shape_
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=device)  # noqa: E731
        a = lambda *s: torch.empty(s, dtype=adt, device=device)  # noqa: E731  activations / their gradients
        # the discriminator sees [real; generated] as ONE batch of 2 bs (two BatchNorm statistics groups)
        self.both = a(2 * bs, *ce.input_shape_)
        self.X = self.both[:bs] if adt == torch.float32 else f(*shp)  # float32: what the caller hands over
        self.mask = f(*shp)
        self.Xm = a(*shp)
        self.dpred = a(*((bs,) + ce.output_shape_))
        self.d_both = a(2 * bs, 1)
        self.d_real, self.d_fake = self.d_both[:bs], self.d_both[bs:]
        self.l2, self.lt, self.lf, self.disc_loss, self.total = f(1), f(1), f(1), f(1), f(1)
        self.graph = None
        self.warm = False
        self.side = torch.cuda.Stream(device=device)  # the context-encoder update runs beside the discriminator's


def _step_state(ce, disc, ce_opt, disc_opt, bs: int, weight: float, device) -> _StepState:
    """The step state of (ce, disc, optimizers, batch size, weight).  It lives ON the context encoder and holds
    references to the other three participants, so a recorded graph can never outlive - or be looked up for
    look-alikes of - the objects whose buffers it is bound to."""
    cache = ce.__dict__.setdefault("_gan_steps", {})
    key = (id(disc), id(ce_opt), id(disc_opt), int(bs), float(weight))
    st = cache.get(key)
    if st is None:
        st = cache[key] = _StepState(ce, disc, bs, device)
        st.partners = (disc, ce_opt, disc_opt)  # keeps the ids in the key from being reused while the entry exists
    return st


def _step_body(ce, disc, ce_opt, disc_opt, st: _StepState, l2_weight: float, overlap: bool = False,
               defer_updates: bool = False):
    """One step of run_context_encoder.py:225-260.  The discriminator runs ONCE on [X; prediction] (2 B samples, two
    BatchNorm statistics groups: the moving statistics get the real batch's update first, as with two calls) and its
    backward pass once on [dL/dD(real); dL/dD(fake)].
    ``defer_updates``: stop after the backward passes (the data-parallel step reduces the gradients and applies the
    updates itself).  ``overlap``: once the prediction and the reconstruction gradient exist, the context-encoder
    backward + update and the discriminator forward / backward / update touch disjoint buffers (each model owns its
    buffer pool, workspace and optimizer state), so the former is forked onto a second stream; at batch 64 neither
    fills the chip on its own."""
    mse, bce = MeanSquaredError(), BinaryCrossentropy(from_logits=False)
    bs = st.mask.shape[0]
    ce.zero_grad_raw()
    disc.zero_grad_raw()
    main = torch.cuda.current_stream()
    if st.X is not st.both and st.X.data_ptr() != st.both.data_ptr():  # bf16 policy: the real half in the models' type
        ops.cast(st.X, st.both[:bs])
    if ce._act_dtype == torch.float32:
        pred = ce.forward_raw(st.X, True, input_mask=st.mask)  # X * (1 - mask), formed inside the first conv's gathers
    else:
        ops.mask_input(st.X, st.mask, st.Xm)
        pred = ce.forward_raw(st.Xm, True)
    # l2 = MSE(X, pred, sample_weight = mask[..., 0] * 10): channel 0 of the NHWC mask is read with stride C
    mse.value_and_grad(st.X, pred, st.l2, st.dpred, 1.0, sample_weight=st.mask, weight_scale=10.0,
                       weight_stride=st.mask.shape[-1])
    ops.copy(pred, st.both[bs:])
    joined = None
    if overlap and not defer_updates:
        forked = torch.cuda.Event()
        forked.record(main)
        with torch.cuda.stream(st.side):
            st.side.wait_event(forked)
            ce.backward_raw(st.dpred, 0, need_dx=False)
            ce_opt.apply_model(ce, 1.0)
            joined = torch.cuda.Event()
            joined.record(st.side)
    p = disc.forward_raw(st.both, True, bn_groups=2)
    p_real, p_fake = p[:bs], p[bs:]
    bce.value_and_grad(1.0, p_real, st.lt, st.d_real, 0.5)
    bce.value_and_grad(0.0, p_fake, st.lf, st.d_fake, 0.5)
    ops.scalar_program([(st.disc_loss, False, [(st.lt, 0.5), (st.lf, 0.5)]),
                        (st.total, False, [(st.l2, l2_weight), (st.disc_loss, 1.0 - l2_weight)])])
    # discriminator update: gradient of disc_loss w.r.t. the discriminator's weights only
    disc.backward_raw(st.d_both, 0, need_dx=False)
    if not defer_updates:
        disc_opt.apply_model(disc, 1.0)
    # context-encoder update: gradient of the masked reconstruction loss only
    if joined is not None:
        main.wait_event(joined)
    else:
        ce.backward_raw(st.dpred, 0, need_dx=False)
        if not defer_updates:
            ce_opt.apply_model(ce, 1.0)
    ce._calls.clear()
    disc._calls.clear()


def train_step(context_encoder, discriminator, ce_opt, disc_opt, X, mask, l2_weight: float = 0.999,
               use_graph: bool = True):
    """run_context_encoder.py:225-260.  ``X`` and ``mask`` are NumPy arrays or device tensors of shape
    (B, H, W, C).  Returns (disc_loss, l2_loss, total_loss) as scalar device tensors."""
    ce, disc = context_encoder, discriminator
    device = default_device()
    Xd, Md = to_device_f32(X, device), to_device_f32(mask, device)
    if not ce.built:
        ce.build(tuple(Xd.shape))
    if not disc.built:
        disc.build(tuple(Xd.shape))
    ce._materialize(device)
    disc._materialize(device)
    st = _step_state(ce, disc, ce_opt, disc_opt, Xd.shape[0], l2_weight, device)
    st.X.copy_(Xd)
    st.mask.copy_(Md)
    if ce._reducer is not None or disc._reducer is not None:  # data-parallel: reduce both gradient buffers, then update
        parallel.run_dp_step(st, use_graph,
                             lambda: _step_body(ce, disc, ce_opt, disc_opt, st, l2_weight, defer_updates=True),
                             [(disc, disc_opt), (ce, ce_opt)], device)
    elif not use_graph or not st.warm:
        _step_body(ce, disc, ce_opt, disc_opt, st, l2_weight)
        st.warm = True
    else:
        if st.graph is None:
            torch.cuda.synchronize()
            st.graph = torch.cuda.CUDAGraph()
            with recording(st.graph):
                _step_body(ce, disc, ce_opt, disc_opt, st, l2_weight, overlap=_OVERLAP)
        st.graph.replay()
    return (as_dtensor(st.disc_loss.view(()).clone()), as_dtensor(st.l2.view(()).clone()),
            as_dtensor(st.total.view(()).clone()))


# ------------------------------------------------------------------------------------------------ driver
def main(n_train: int = 2048, epochs: int = 2, batch_size: int = 64, seed: int = global_vars.DEFAULT_SEED):
    """run_context_encoder.py:287-527 without the plotting: synthetic data in the reference's format stands in
    for ``data/data.npz`` (produced there by an msprime pipeline that is not available offline)."""
    from .data import synth_training_tensor
    rng = np.random.default_rng(seed)
    data = batch_sort(synth_training_tensor(n_train, seed))
    split = int(0.8 * n_train)
    perm = rng.permutation(n_train)
    X_train, X_test = data[perm[:split]], data[perm[split:]]
    # the training set and its masks live in HBM for the whole run (the masks are generated there); a step slices both
    device = default_device()
    X_train, X_test = to_device_f32(X_train, device), to_device_f32(X_test, device)
    X_train_mask = batch_random_mask_device(X_train.shape, 0.25, seed, 0, device)
    X_test_mask = batch_random_mask_device(X_test.shape, 0.25, seed, X_train.shape[0] << 20, device)
    X_test_masked = torch.empty_like(X_test)
    ops.mask_input(X_test, X_test_mask, X_test_masked)
    model = build_model(input_shape=tuple(X_train.shape[1:]), activation="relu", latent_dimensions=1_000)
    disc = context_encoder.Discriminator()
    disc.build_graph((1, global_vars.NUM_HAPLOTYPES, global_vars.NUM_SNPS, global_vars.NUM_CHANNELS))
    disc.discriminator.summary()
    LR = 2e-4
    ce_optimizer, disc_optimizer = build_optimizer(learning_rate=LR * 10), build_optimizer(learning_rate=LR)
    res = []
    for epoch in range(epochs):
        for step in range(X_train.shape[0] // batch_size):
            sl = slice(batch_size * step, batch_size * (step + 1))
            disc_loss, recon_loss, total_loss = train_step(model, disc, ce_optimizer, disc_optimizer, X_train[sl],
                                                           X_train_mask[sl])
        predictions = model.predict(X_test_masked)
        disc_loss_test = measure_disc_loss(disc, X_test, predictions)
        recon_loss_test = measure_recon_loss(X_test, predictions, X_test_mask)
        print(f"epoch {epoch}: disc {float(disc_loss):.5f} l2 {float(recon_loss):.5f} | "
              f"val disc {float(disc_loss_test):.5f} val l2 {float(recon_loss_test):.5f}")
        res.append({"epoch": epoch, "disc": float(disc_loss), "recon": float(recon_loss),
                    "val_disc": float(disc_loss_test), "val_recon": float(recon_loss_test)})
    predictions = model.predict(X_test_masked)
    loss_dist = measure_recon_loss(X_test, predictions, X_test_mask, reduction="none")
    per_sample = loss_dist.numpy().mean(axis=(1, 2))
    return res, per_sample


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("-run_sweep", action="store_true")
    p.add_argument("-search_size", type=int, default=20)
    args = p.parse_args()
    main()

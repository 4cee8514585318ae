"""The reference's ``unet.py`` on the HIP layer engine: ``convolution`` / ``upconvolution`` blocks and the
skip-connected U-Net generator the reference names ``ContextEncoder`` (unet.py:29-148).

``Generator`` and a two-input ``Discriminator`` are what ``run_unet.py:33,276,279`` asks this module for;
the reference file defines neither (its ``Discriminator`` cannot be constructed: ``filter_size`` is
undefined at unet.py:177).  They are provided here with the structure the driver implies - the U-Net
as generator, a conv stack over concat([X, y]) emitting logits - and are not pinned by any reference
definition.
"""
from __future__ import annotations

from typing import List, Tuple

import torch

from . import global_vars, ops
from .losses import to_device_f32
from .model import Model
from .nn import (BatchNormalization, CallCtx, Conv2D, Conv2DTranspose, Flatten, Layer, LeakyReLU, Sequential,
                 as_dtensor, default_device)

# unet.py:13-26
final_width = 4
n_filters = 0
while global_vars.NUM_SNPS / (2 ** n_filters) > final_width:
    n_filters += 1
FILTER_SIZES = [64]
for _fi in range(1, n_filters):
    FILTER_SIZES.append(FILTER_SIZES[0] if _fi == 1 else FILTER_SIZES[_fi - 1] * 2)


def convolution(n_filters: int, kernel: Tuple[int, int], stride: Tuple[int, int], apply_batchnorm: bool = True,
                padding: str = "same") -> Sequential:
    """unet.py:29-55: Conv2D(no bias, N(0, 0.02) init) -> BatchNorm -> LeakyReLU(0.2)."""
    block = Sequential(name=None)
    block.add(Conv2D(n_filters, kernel, strides=stride, padding=padding, kernel_initializer="normal0.02",
                     use_bias=False))
    if apply_batchnorm:
        block.add(BatchNormalization())
    block.add(LeakyReLU(0.2))
    return block


def upconvolution(n_filters: int, kernel: Tuple[int, int], stride: Tuple[int, int], apply_batchnorm: bool = True,
                  padding: str = "same") -> Sequential:
    """unet.py:58-81: Conv2DTranspose(no bias, N(0, 0.02) init) -> LeakyReLU(0.2); the batch-norm flag is
    accepted and ignored, as in the reference."""
    block = Sequential(name=None)
    block.add(Conv2DTranspose(n_filters, kernel, strides=stride, padding=padding, kernel_initializer="normal0.02",
                              use_bias=False))
    block.add(LeakyReLU(0.2))
    return block


class ContextEncoder(Model):
    """unet.py:84-148: four stride-2 down blocks; three up blocks, each followed by
    concat([up, skip]) with skips = reversed(first three down outputs); final Conv2DTranspose to
    OUTPUT_CHANNELS with tanh (and bias).  The reference constructs that last layer inside ``call``
    (so it is re-initialised on every eager call); here it is constructed once."""

    def __init__(self, kernel: Tuple[int, int] = (2, 2), stride: Tuple[int, int] = (2, 2),
                 latent_dimensions: int = 128, input_shape=None):
        super().__init__(name="unet")
        self.kernel, self.stride, self.latent_dimensions = kernel, stride, latent_dimensions
        self.declared_input_shape = input_shape
        self.encoder_stack: List[Sequential] = [convolution(64, kernel, stride) for _ in range(4)]
        self.decoder_stack: List[Sequential] = [upconvolution(64, kernel, stride) for _ in range(3)]
        self.final_conv = Conv2DTranspose(global_vars.OUTPUT_CHANNELS, kernel, strides=stride, padding="same",
                                          kernel_initializer="normal0.02", activation="tanh")

    def _build(self, in_shape):
        shape = in_shape
        downs = []
        for blk in self.encoder_stack:
            shape = blk.build(shape)
            downs.append(shape)
        self._skip_shapes = downs[:-1][::-1]
        for blk, skip in zip(self.decoder_stack, self._skip_shapes):
            up = blk.build(shape)
            if up[:2] != skip[:2]:
                raise ValueError(f"skip {skip} does not line up with upsampled {up}")
            shape = (up[0], up[1], up[2] + skip[2])
        return self.final_conv.build(shape)

    def forward(self, ctx, x):
        outs = []
        for blk in self.encoder_stack:
            x = blk.forward(ctx, x)
            outs.append(x)
        skips = outs[:-1][::-1]
        for i, (blk, skip) in enumerate(zip(self.decoder_stack, skips)):
            up = blk.forward(ctx, x)
            n, h, w, ca = up.shape
            cb = skip.shape[-1]
            x = ctx.buf(self, ("cat", i), (n, h, w, ca + cb))
            ops.concat_channels_fwd(up, skip, x, n * h * w, ca, cb)
        return self.final_conv.forward(ctx, x)

    def backward(self, ctx, dy, need_dx=True):
        d = self.final_conv.backward(ctx, dy, need_dx=True)
        # gradient arriving at each encoder output through its skip connection
        skip_grads = [None] * len(self.encoder_stack)
        n_dec = len(self.decoder_stack)
        for i in range(n_dec - 1, -1, -1):
            n, h, w, c = d.shape
            cb = self._skip_shapes[i][2]
            ca = c - cb
            d_up = ctx.buf(self, ("d_up", i), (n, h, w, ca))
            d_skip = ctx.buf(self, ("d_skip", i), (n, h, w, cb))
            ops.concat_channels_bwd(d, d_up, d_skip, n * h * w, ca, cb)
            skip_grads[len(self.encoder_stack) - 2 - i] = d_skip
            d = self.decoder_stack[i].backward(ctx, d_up, need_dx=True)
        for j in range(len(self.encoder_stack) - 1, -1, -1):
            if skip_grads[j] is not None:
                ops.axpy(d, skip_grads[j], 1.0)
            d = self.encoder_stack[j].backward(ctx, d, need_dx=(need_dx or j > 0))
        return d


class Generator(ContextEncoder):
    """What ``run_unet.py:33,276`` calls ``unet.Generator``: the skip-connected U-Net above
    (INPUT_CHANNELS in, OUTPUT_CHANNELS out).  No definition exists in the reference."""

    def __init__(self, input_shape=None, kernel=(2, 2), stride=(2, 2)):
        super().__init__(kernel=kernel, stride=stride, input_shape=input_shape)
        self.name = "generator"


class Discriminator(Model):
    """What ``run_unet.py:279,176-177`` calls ``unet.Discriminator()`` with a list input ``[X, y]``: the two
    images are concatenated on the channel axis, three stride-2 ``convolution`` blocks reduce 32x32 to 4x4
    and a 4x4 valid conv emits ONE LOGIT per sample (the driver's BCE uses ``from_logits=True``).
    The reference's class of this name cannot be constructed (unet.py:177), so this structure is the
    build's own (a pix2pix-style patch discriminator collapsed to one logit)."""

    def __init__(self, kernel_size: int = 4):
        super().__init__(name="unet_discriminator")
        k = (kernel_size, kernel_size)
        self.blocks: List[Sequential] = [
            convolution(64, k, (2, 2), apply_batchnorm=False),
            convolution(128, k, (2, 2)),
            convolution(256, k, (2, 2)),
        ]
        self.head = Sequential([Conv2D(1, k, strides=(1, 1), padding="valid", kernel_initializer="normal0.02"),
                                Flatten()], name="head")

    def _build(self, in_shape):
        shape = in_shape
        for b in self.blocks:
            shape = b.build(shape)
        return self.head.build(shape)

    def forward(self, ctx, x):
        for b in self.blocks:
            x = b.forward(ctx, x)
        return self.head.forward(ctx, x)

    def backward(self, ctx, dy, need_dx=True):
        d = self.head.backward(ctx, dy, need_dx=True)
        for j in range(len(self.blocks) - 1, -1, -1):
            d = self.blocks[j].backward(ctx, d, need_dx=(need_dx or j > 0))
        return d

    # ---- list input [X, y] -> channel concat
    def concat_inputs(self, xs, slot_key) -> torch.Tensor:
        a = to_device_f32(xs[0], self._pool.device or default_device())
        b = to_device_f32(xs[1], a.device)
        if a.shape[:3] != b.shape[:3]:
            raise ValueError(f"Discriminator inputs {tuple(a.shape)} and {tuple(b.shape)} differ in batch/space")
        n, h, w, ca = a.shape
        cb = b.shape[-1]
        if self._pool.device is None:
            self._pool.device = a.device
        cat = self._pool.get(("cat_in", slot_key), (n, h, w, ca + cb))
        ops.concat_channels_fwd(a, b, cat, n * h * w, ca, cb)
        return cat

    def __call__(self, x, training: bool = False):
        if isinstance(x, (list, tuple)):
            x = self.concat_inputs(x, len(self._calls) if training else "infer")
        return super().__call__(x, training=training)


if __name__ == "__main__":
    # unet.py:208-217
    generator = ContextEncoder()
    generator.build_graph((1, global_vars.NUM_HAPLOTYPES, global_vars.NUM_SNPS, global_vars.INPUT_CHANNELS))
    generator.summary()

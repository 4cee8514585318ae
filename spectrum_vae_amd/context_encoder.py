"""The reference's ``context_encoder.py`` on the HIP layer engine: the inpainting generator
(``ContextEncoder``), its ``Discriminator`` and the (1 x k)-kernel "one-hot" variants
(context_encoder.py:29-372).  Same class names, constructor arguments and defaults.
"""
from __future__ import annotations

from typing import List

import numpy as np

from . import global_vars
from .autoencoder import _EncoderDecoder
from .model import Model
from .nn import BatchNormalization, Conv2D, Conv2DTranspose, Flatten, Input, Layer, LeakyReLU, Sequential

# context_encoder.py:13-26: filter counts of the stride-2 stages that take NUM_SNPS down to a width of 4
final_width = 4
n_filters = 0
while global_vars.NUM_SNPS / (2 ** n_filters) > final_width:
    n_filters += 1
FILTER_SIZES = [64]
for _fi in range(1, n_filters):
    FILTER_SIZES.append(FILTER_SIZES[0] if _fi == 1 else FILTER_SIZES[_fi - 1] * 2)

_DEFAULT_IN = (global_vars.NUM_HAPLOTYPES, global_vars.NUM_SNPS, global_vars.NUM_CHANNELS)


def _down_stages(filter_sizes, kernel, strides) -> List[Layer]:
    """Conv(stride 2, same, LeakyReLU 0.2) -> BatchNorm, once per entry of ``filter_sizes``."""
    layers: List[Layer] = []
    for f in filter_sizes:
        layers.append(Conv2D(f, kernel, strides=strides, activation=LeakyReLU(0.2), padding="same"))
        layers.append(BatchNormalization())
    return layers


class ContextEncoder(_EncoderDecoder):
    """context_encoder.py:212-304.  Encoder: three 4x4 stride-2 stages (64, 64, 128 filters), then a 4x4
    valid conv collapsing 4x4 to 1x1 x ``latent_dimensions``.  Decoder: 4x4 valid transposed conv back to
    4x4x128 (relu, BatchNorm), two 4x4 stride-2 transposed stages (64, 64; relu, BatchNorm) and a final
    4x4 stride-2 transposed conv to the input channel count with tanh."""

    def __init__(self, input_shape=_DEFAULT_IN, kernel_size: int = 4, latent_dimensions: int = 1_000):
        super().__init__(name="context_encoder")
        k = (kernel_size, kernel_size)
        enc: List[Layer] = [Input(shape=input_shape)] + _down_stages(FILTER_SIZES, k, (2, 2))
        enc.append(Conv2D(latent_dimensions, k, strides=(1, 1), activation=LeakyReLU(0.2), padding="valid"))
        self.encoder = Sequential(enc, name="encoder")
        dec: List[Layer] = [Conv2DTranspose(FILTER_SIZES[-1], k, activation="relu", padding="valid",
                                            input_shape=(1, 1, latent_dimensions)),
                            BatchNormalization()]
        for f in FILTER_SIZES[::-1][1:]:
            dec.append(Conv2DTranspose(f, k, strides=(2, 2), activation="relu", padding="same"))
            dec.append(BatchNormalization())
        dec.append(Conv2DTranspose(input_shape[-1], k, strides=(2, 2), activation="tanh", padding="same"))
        self.decoder = Sequential(dec, name="decoder")


class _DiscBase(Model):
    discriminator: Sequential

    def _build(self, in_shape):
        return self.discriminator.build(in_shape)

    def forward(self, ctx, x):
        return self.discriminator.forward(ctx, x)

    def backward(self, ctx, dy, need_dx=True):
        return self.discriminator.backward(ctx, dy, need_dx)


class Discriminator(_DiscBase):
    """context_encoder.py:318-361: the encoder's three stride-2 stages, a 4x4 valid conv to one sigmoid
    unit, Flatten -> (B, 1) probabilities."""

    def __init__(self, input_shape=_DEFAULT_IN, kernel_size: int = 4):
        super().__init__(name="discriminator")
        k = (kernel_size, kernel_size)
        layers: List[Layer] = [Input(shape=input_shape)] + _down_stages(FILTER_SIZES, k, (2, 2))
        layers.append(Conv2D(1, k, strides=(1, 1), padding="valid", activation="sigmoid"))
        layers.append(Flatten())
        self.discriminator = Sequential(layers, name="discriminator")


class ContextEncoderOneHot(_EncoderDecoder):
    """context_encoder.py:29-133: (1 x k) stride-(1,2) stages on a (4, NUM_SNPS, 3) input, a (4 x k)
    valid collapse, and a decoder that opens with BatchNorm and uses no BatchNorm afterwards."""

    def __init__(self, input_shape=(4, global_vars.NUM_SNPS, 3), kernel_size: int = 4,
                 latent_dimensions: int = 1_000, activation: str = "relu", padding: str = "same"):
        super().__init__(name="context_encoder_onehot")
        ni = int(np.log2(input_shape[1])) - 3
        filter_sizes = FILTER_SIZES[: ni + 1]
        enc: List[Layer] = [Input(shape=input_shape)] + _down_stages(filter_sizes, (1, kernel_size), (1, 2))
        enc.append(Conv2D(latent_dimensions, (4, kernel_size), strides=(1, 1), activation="relu", padding="valid"))
        self.encoder = Sequential(enc, name="encoder")
        dec: List[Layer] = [BatchNormalization(),
                            Conv2DTranspose(filter_sizes[-1], (kernel_size, kernel_size), activation="relu",
                                            padding="valid", input_shape=(1, 1, latent_dimensions))]
        for f in filter_sizes[1:-1][::-1]:
            dec.append(Conv2DTranspose(f, (1, kernel_size), strides=(1, 2), activation="relu", padding="same"))
        dec.append(Conv2DTranspose(input_shape[-1], (1, kernel_size), strides=(1, 2), activation="tanh",
                                   padding="same"))
        self.decoder = Sequential(dec, name="decoder")


class DiscriminatorOneHot(_DiscBase):
    """context_encoder.py:147-198."""

    def __init__(self, input_shape=(4, global_vars.NUM_SNPS // 2, 3), activation: str = "relu",
                 kernel_size: int = 4):
        super().__init__(name="discriminator_onehot")
        ni = int(np.log2(input_shape[1])) - 3
        layers: List[Layer] = [Input(shape=input_shape)] + _down_stages(FILTER_SIZES[: ni + 1], (1, kernel_size),
                                                                        (1, 2))
        layers.append(Conv2D(1, (4, kernel_size), strides=(1, 1), padding="valid", activation="sigmoid"))
        layers.append(Flatten())
        self.discriminator = Sequential(layers, name="discriminator")

"""Layer-descriptor restatement of the reference's model constructors.

Test infrastructure (see ``oracle/__init__.py``).  Each builder walks the same
construction loop as the reference constructor it cites and returns plain
descriptor lists that ``oracle.torch_ref.RefSequential`` executes.  Written
independently of ``spectrum_vae_amd`` so that a bookkeeping slip in one of the
two shows up as a structure mismatch in ``tests/``.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import torch

from .torch_ref import RefSequential

NUM_SNPS = 32  # global_vars.py:3
NUM_HAPLOTYPES = 32  # global_vars.py:5
NUM_CHANNELS = 3  # global_vars.py:6
INPUT_CHANNELS = 1  # global_vars.py:8
OUTPUT_CHANNELS = 3  # global_vars.py:9


def _filter_sizes() -> List[int]:
    """unet.py:13-26 / context_encoder.py:13-26."""
    final_width, n = 4, 0
    while NUM_SNPS / (2 ** n) > final_width:
        n += 1
    fs = [64]
    for fi in range(1, n):
        fs.append(fs[0] if fi == 1 else fs[fi - 1] * 2)
    return fs


FILTER_SIZES = _filter_sizes()


def _conv(filters, k, act, padding, strides=(1, 1), use_bias=True, alpha=0.2, init="glorot_uniform"):
    if isinstance(k, int):
        k = (k, k)
    return dict(type="conv2d", filters=int(filters), kernel=tuple(k), strides=tuple(strides), padding=padding,
                activation=act, alpha=alpha, use_bias=use_bias, init=init)


def _convT(filters, k, act, padding, strides=(1, 1), use_bias=True, alpha=0.2, init="glorot_uniform"):
    d = _conv(filters, k, act, padding, strides, use_bias, alpha, init)
    d["type"] = "conv2d_transpose"
    return d


def _dense(units, act):
    return dict(type="dense", units=int(units), activation=act)


# --------------------------------------------------------------------------- autoencoder.py
def caepadded(input_shape=(32, 32, 3), initial_filters=8, conv_layer_multiplier=2, conv_layers=3, fc_layers=2,
              fc_layer_size=64, dropout=0.5, activation="elu", kernel_size=3, conv_operations=2,
              latent_dimensions=8, batch_norm=False) -> Tuple[list, list]:
    """autoencoder.py:408-528."""
    enc, dec = [], []
    fs = initial_filters
    for block in range(conv_layers):
        for _ in range(conv_operations):
            enc.append(_conv(fs, kernel_size, activation, "same"))
            if batch_norm:
                enc.append(dict(type="batchnorm"))
        enc.append(dict(type="maxpool", pool=(2, 2), padding="same"))
        if block + 1 < conv_layers:
            fs *= conv_layer_multiplier
    if latent_dimensions > 0:
        enc.append(dict(type="flatten"))
        for _ in range(fc_layers):
            enc.append(_dense(fc_layer_size, activation))
            if batch_norm:
                enc.append(dict(type="batchnorm"))
        enc.append(_dense(latent_dimensions, activation))
        for _ in range(fc_layers):
            dec.append(_dense(fc_layer_size, activation))
        fh = int(input_shape[0] / (2 ** conv_layers))
        fw = int(input_shape[1] / (2 ** conv_layers))
        dec.append(_dense(fh * fw * fs, activation))
        dec.append(dict(type="reshape", shape=(fh, fw, int(fs))))
    for block in range(conv_layers):
        final = block + 1 == conv_layers
        dec.append(dict(type="upsample", size=(2, 2)))
        for op in range(conv_operations):
            if op + 1 == conv_operations:
                fs = fs / conv_layer_multiplier
                dec.append(_conv(NUM_CHANNELS if final else int(fs), kernel_size, "tanh" if final else activation,
                                 "same"))
            else:
                dec.append(_conv(int(fs), kernel_size, activation, "same"))
    return enc, dec


class RefVAEPadded:
    """Variational form of ``caepadded``: NO REFERENCE MODEL (the reference ships only the Sampling layer,
    autoencoder.py:264-272).  Conv stack + Flatten + Dense layers of autoencoder.py:461-494 -> two linear Dense heads
    (z_mean, z_log_var) -> z = mean + exp(0.5 * logvar) * eps (autoencoder.py:272) -> the decoder of
    autoencoder.py:496-526.  loss = mse(x, decoded) + beta * mean_b(-1/2 sum_d(1 + logvar - mean^2 - exp(logvar))).
    Parameter order: encoder, z_mean, z_log_var, decoder (each in Keras order)."""

    def __init__(self, input_shape=(32, 32, 3), initial_filters=8, conv_layer_multiplier=2, conv_layers=3, fc_layers=2,
                 fc_layer_size=64, activation="elu", kernel_size=3, conv_operations=2, latent_dimensions=8, beta=1.0,
                 dtype=torch.float64, seed=1833):
        enc, dec = caepadded(input_shape, initial_filters, conv_layer_multiplier, conv_layers, fc_layers, fc_layer_size,
                             0.0, activation, kernel_size, conv_operations, latent_dimensions, False)
        enc = enc[:-1]  # the auto-encoder's bottleneck Dense is replaced by the two heads
        self.beta = float(beta)
        self.encoder = RefSequential(enc, input_shape, dtype, seed)
        feat = self.encoder.output_shape
        self.z_mean = RefSequential([_dense(latent_dimensions, None)], feat, dtype, seed + 1)
        self.z_log_var = RefSequential([_dense(latent_dimensions, None)], feat, dtype, seed + 2)
        self.decoder = RefSequential(dec, (latent_dimensions,), dtype, seed + 3)
        self.dtype = dtype

    @property
    def parts(self):
        return [self.encoder, self.z_mean, self.z_log_var, self.decoder]

    @property
    def params(self):
        return [p for m in self.parts for p in m.params]

    def load_params(self, arrays: Sequence):
        i = 0
        for m in self.parts:
            m.load_params(arrays[i:i + len(m.params)])
            i += len(m.params)
        assert i == len(arrays)

    def __call__(self, x, eps, training=False):
        """Returns (decoded, kl); ``eps``: the (batch, latent) standard-normal draw."""
        h = self.encoder(x, training=training)
        mean, logvar = self.z_mean(h, training=training), self.z_log_var(h, training=training)
        z = mean + torch.exp(0.5 * logvar) * eps
        kl = torch.mean(-0.5 * torch.sum(1 + logvar - mean * mean - torch.exp(logvar), dim=1))
        return self.decoder(z, training=training), kl

    def loss(self, x, eps, training=True):
        dec, kl = self(x, eps, training)
        return torch.mean((dec - x) ** 2) + self.beta * kl


def cae1d(input_shape=(32, 32, 3), initial_filters=8, conv_layer_multiplier=2, conv_layers=3, fc_layers=2,
          fc_layer_size=64, dropout=0.5, activation="elu", kernel_size=3, conv_operations=2,
          latent_dimensions=8) -> Tuple[list, list]:
    """autoencoder.py:275-389."""
    enc, dec = [], []
    fs = initial_filters
    width = input_shape[1]
    for block in range(conv_layers):
        for _ in range(conv_operations):
            enc.append(_conv(fs, (1, kernel_size), activation, "valid"))
            width -= kernel_size - 1
            enc.append(dict(type="batchnorm"))
        enc.append(dict(type="maxpool", pool=(1, 2), padding="valid"))
        width /= 2
        if block + 1 < conv_layers:
            fs *= conv_layer_multiplier
    if latent_dimensions > 0:
        enc.append(dict(type="flatten"))
        for _ in range(fc_layers):
            enc.append(_dense(fc_layer_size, activation))
            enc.append(dict(type="dropout", rate=dropout))
        enc.append(_dense(latent_dimensions, activation))
        for _ in range(fc_layers):
            dec.append(_dense(fc_layer_size, activation))
        dec.append(_dense(int(input_shape[0] * width * fs), activation))
        dec.append(dict(type="reshape", shape=(input_shape[0], int(width), int(fs))))
    for block in range(conv_layers):
        final = block + 1 == conv_layers
        dec.append(dict(type="upsample", size=(1, 2)))
        for op in range(conv_operations):
            if op + 1 == conv_operations:
                fs = fs / conv_layer_multiplier
                dec.append(_convT(NUM_CHANNELS if final else int(fs), (1, kernel_size),
                                  "tanh" if final else activation, "valid"))
            else:
                dec.append(_convT(int(fs), (1, kernel_size), activation, "valid"))
    return enc, dec


def ae(input_shape=(32, 3), fc_layers=3, fc_layer_size=128, dropout=0.5, activation="elu",
       latent_dimensions=8) -> Tuple[list, list]:
    """autoencoder.py:199-244."""
    enc = [dict(type="flatten")]
    for _ in range(fc_layers):
        enc.append(_dense(fc_layer_size, activation))
        enc.append(dict(type="dropout", rate=dropout))
    enc.append(_dense(latent_dimensions, activation))
    dec = [_dense(fc_layer_size, activation) for _ in range(fc_layers)]
    dec.append(_dense(input_shape[0] * input_shape[1], "tanh"))
    dec.append(dict(type="reshape", shape=tuple(input_shape)))
    return enc, dec


def _cnn_classifier(kernel, pool, padding, initial_filters, conv_layer_multiplier, conv_layers, fc_layers,
                    fc_layer_size, dropout, activation, conv_operations) -> list:
    layers = []
    fs = initial_filters
    for block in range(conv_layers):
        for _ in range(conv_operations):
            layers.append(_conv(fs, kernel, activation, padding))
        layers.append(dict(type="maxpool", pool=pool, padding=padding))
        if block + 1 < conv_layers:
            fs *= conv_layer_multiplier
    layers.append(dict(type="flatten"))
    for _ in range(fc_layers):
        layers.append(_dense(fc_layer_size, activation))
        layers.append(dict(type="dropout", rate=dropout))
    layers.append(_dense(1, activation))
    return layers


def cnn1d(initial_filters=8, conv_layer_multiplier=2, conv_layers=3, fc_layers=2, fc_layer_size=64, dropout=0.5,
          activation="elu", kernel_size=3, conv_operations=2) -> list:
    """autoencoder.py:59-112."""
    return _cnn_classifier((1, kernel_size), (1, 2), "valid", initial_filters, conv_layer_multiplier, conv_layers,
                           fc_layers, fc_layer_size, dropout, activation, conv_operations)


def cnnpadded(initial_filters=8, conv_layer_multiplier=2, conv_layers=3, fc_layers=2, fc_layer_size=64, dropout=0.5,
              activation="elu", kernel_size=3, conv_operations=2) -> list:
    """autoencoder.py:129-181."""
    return _cnn_classifier((kernel_size, kernel_size), (2, 2), "same", initial_filters, conv_layer_multiplier,
                           conv_layers, fc_layers, fc_layer_size, dropout, activation, conv_operations)


def cnnriley(activation="relu") -> list:
    """autoencoder.py:16-41."""
    return [
        _conv(32, (1, 5), activation, "valid"),
        dict(type="maxpool", pool=(1, 2), padding="valid"),
        _conv(64, (1, 5), activation, "valid"),
        dict(type="maxpool", pool=(1, 2), padding="valid"),
        dict(type="reduce_sum_h"),
        dict(type="flatten"),
        _dense(128, activation),
        _dense(128, activation),
        _dense(1, None),
    ]


# --------------------------------------------------------------------------- context_encoder.py
def context_encoder(input_shape=(32, 32, 3), kernel_size=4, latent_dimensions=1000) -> Tuple[list, list]:
    """context_encoder.py:212-299."""
    enc = []
    for fs in FILTER_SIZES:
        enc.append(_conv(fs, kernel_size, "lrelu", "same", strides=(2, 2)))
        enc.append(dict(type="batchnorm"))
    enc.append(_conv(latent_dimensions, kernel_size, "lrelu", "valid"))
    dec = [_convT(FILTER_SIZES[-1], kernel_size, "relu", "valid"), dict(type="batchnorm")]
    for fs in FILTER_SIZES[::-1][1:]:
        dec.append(_convT(fs, kernel_size, "relu", "same", strides=(2, 2)))
        dec.append(dict(type="batchnorm"))
    dec.append(_convT(input_shape[-1], kernel_size, "tanh", "same", strides=(2, 2)))
    return enc, dec


def discriminator(kernel_size=4) -> list:
    """context_encoder.py:318-358."""
    layers = []
    for fs in FILTER_SIZES:
        layers.append(_conv(fs, kernel_size, "lrelu", "same", strides=(2, 2)))
        layers.append(dict(type="batchnorm"))
    layers.append(_conv(1, kernel_size, "sigmoid", "valid"))
    layers.append(dict(type="flatten"))
    return layers


def context_encoder_onehot(input_shape=(4, 32, 3), kernel_size=4, latent_dimensions=1000) -> Tuple[list, list]:
    """context_encoder.py:29-128."""
    n = int(math.log2(input_shape[1]))
    filter_sizes = FILTER_SIZES[: n - 3 + 1]
    enc = []
    for fs in filter_sizes:
        enc.append(_conv(fs, (1, kernel_size), "lrelu", "same", strides=(1, 2)))
        enc.append(dict(type="batchnorm"))
    enc.append(_conv(latent_dimensions, (4, kernel_size), "relu", "valid"))
    dec = [dict(type="batchnorm"), _convT(filter_sizes[-1], kernel_size, "relu", "valid")]
    for fs in filter_sizes[1:-1][::-1]:
        dec.append(_convT(fs, (1, kernel_size), "relu", "same", strides=(1, 2)))
    dec.append(_convT(input_shape[-1], (1, kernel_size), "tanh", "same", strides=(1, 2)))
    return enc, dec


def discriminator_onehot(input_shape=(4, 16, 3), kernel_size=4) -> list:
    """context_encoder.py:147-195."""
    n = int(math.log2(input_shape[1]))
    filter_sizes = FILTER_SIZES[: n - 3 + 1]
    layers = []
    for fs in filter_sizes:
        layers.append(_conv(fs, (1, kernel_size), "lrelu", "same", strides=(1, 2)))
        layers.append(dict(type="batchnorm"))
    layers.append(_conv(1, (4, kernel_size), "sigmoid", "valid"))
    layers.append(dict(type="flatten"))
    return layers


# --------------------------------------------------------------------------- unet.py
def unet_down_block(n_filters=64, kernel=(2, 2), stride=(2, 2), apply_batchnorm=True) -> list:
    """unet.py:29-55: Conv(no bias, N(0,.02)) -> BN -> LeakyReLU(.2)."""
    layers = [_conv(n_filters, kernel, None, "same", strides=stride, use_bias=False, init="normal0.02")]
    if apply_batchnorm:
        layers.append(dict(type="batchnorm"))
    layers.append(dict(type="leakyrelu", alpha=0.2))
    return layers


def unet_up_block(n_filters=64, kernel=(2, 2), stride=(2, 2)) -> list:
    """unet.py:58-81: Conv2DTranspose(no bias) -> LeakyReLU(.2) (no BN)."""
    return [_convT(n_filters, kernel, None, "same", strides=stride, use_bias=False, init="normal0.02"),
            dict(type="leakyrelu", alpha=0.2)]


class RefUNet:
    """unet.ContextEncoder (unet.py:84-148): 4 down blocks, 3 up blocks each followed
    by concat([up, skip]) with skips = reversed(first three down outputs), then a
    final Conv2DTranspose to OUTPUT_CHANNELS with tanh and bias (built once here;
    the reference rebuilds it on every call, which re-initialises it)."""

    def __init__(self, input_shape=(32, 32, 1), kernel=(2, 2), stride=(2, 2), seed=1833, dtype=torch.float32,
                 lowp: bool = False):
        self.blocks: List[RefSequential] = []
        shape = tuple(input_shape)
        down_shapes = []
        for i in range(4):
            blk = RefSequential(unet_down_block(64, kernel, stride), shape, dtype, seed + i, lowp=lowp)
            self.blocks.append(blk)
            shape = blk.output_shape
            down_shapes.append(shape)
        skips = down_shapes[:-1][::-1]
        for i in range(3):
            blk = RefSequential(unet_up_block(64, kernel, stride), shape, dtype, seed + 10 + i, lowp=lowp)
            self.blocks.append(blk)
            shape = blk.output_shape
            shape = (shape[0], shape[1], shape[2] + skips[i][2])
        final = [_convT(OUTPUT_CHANNELS, kernel, "tanh", "same", strides=stride, init="normal0.02")]
        self.final = RefSequential(final, shape, dtype, seed + 20, lowp=lowp)
        self.blocks.append(self.final)
        self.output_shape = self.final.output_shape

    @property
    def params(self):
        return [p for b in self.blocks for p in b.params]

    @property
    def state(self):
        return [s for b in self.blocks for s in b.state]

    def num_params(self):
        return sum(p.numel() for p in self.params)

    def load_params(self, arrays: Sequence):
        i = 0
        for b in self.blocks:
            n = len(b.params)
            b.load_params(arrays[i:i + n])
            i += n
        assert i == len(arrays)

    def __call__(self, x, training=False):
        skips = []
        for blk in self.blocks[:4]:
            x = blk(x, training)
            skips.append(x)
        skips = skips[:-1][::-1]
        for blk, skip in zip(self.blocks[4:7], skips):
            x = blk(x, training)
            x = torch.cat([x, skip], dim=-1)
        return self.final(x, training)

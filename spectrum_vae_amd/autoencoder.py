"""Model classes of the reference's ``autoencoder.py`` on the HIP layer engine.

Same class names, constructor arguments and defaults (autoencoder.py:16-26, 59-77, 129-147, 199-213,
264, 275-294, 408-428); each exposes ``encoder`` / ``decoder`` (or one stack), ``build_graph`` and the
Keras call/compile/fit/predict surface inherited from ``model.Model``.
"""
from __future__ import annotations

from typing import List, Tuple

import torch

from . import global_vars, ops
from .model import Model, SequentialModel
from .nn import (BatchNormalization, Conv2D, Conv2DTranspose, Dense, Dropout, Flatten, Input, Layer, MaxPooling2D,
                 ReduceSumH, Reshape, Sequential, UpSampling2D, as_dtensor)

_DEFAULT_IN = (global_vars.NUM_HAPLOTYPES, global_vars.NUM_SNPS, global_vars.NUM_CHANNELS)


class _EncoderDecoder(Model):
    """decoded = decoder(encoder(x)) (autoencoder.py:246-249, 391-394, 530-536)."""

    encoder: Sequential
    decoder: Sequential

    def _build(self, in_shape):
        return self.decoder.build(self.encoder.build(in_shape))

    def _sequentials(self):
        return [self.encoder, self.decoder]

    def forward(self, ctx, x):
        return self.decoder.forward(ctx, self.encoder.forward(ctx, x))

    chain_dgrad_ok = True  # backward() hands gradients from Sequential to Sequential only

    def backward(self, ctx, dy, need_dx=True):
        d = self.decoder.backward(ctx, dy, need_dx=True)
        return self.encoder.backward(ctx, d, need_dx=need_dx)


def _conv_stack(kernel, pool, padding, initial_filters, multiplier, conv_layers, conv_operations, activation,
                batch_norm) -> Tuple[List[Layer], int]:
    """The encoder loop shared by CNN1D / CNNPadded / CAE1D / CAEPadded: per block ``conv_operations``
    convs (each optionally followed by BatchNorm) then one max-pool; filters grow between blocks."""
    layers: List[Layer] = []
    filters = initial_filters
    for block in range(conv_layers):
        for _ in range(conv_operations):
            layers.append(Conv2D(filters, kernel, activation=activation, padding=padding))
            if batch_norm:
                layers.append(BatchNormalization())
        layers.append(MaxPooling2D(pool, padding=padding))
        if block + 1 < conv_layers:
            filters *= multiplier
    return layers, filters


class CNNRiley(SequentialModel):
    """autoencoder.py:16-45: two (1,5) conv+pool stages, permutation-invariant sum over haplotypes, 3 dense."""

    def __init__(self, input_shape=_DEFAULT_IN, activation: str = "relu"):
        super().__init__([
            Input(shape=input_shape),
            Conv2D(32, (1, 5), activation=activation),
            MaxPooling2D((1, 2)),
            Conv2D(64, (1, 5), activation=activation),
            MaxPooling2D((1, 2)),
            ReduceSumH(),
            Flatten(),
            Dense(128, activation=activation),
            Dense(128, activation=activation),
            Dense(1),
        ], name="cnn_riley")


class _CNNClassifier(SequentialModel):
    def __init__(self, name, kernel, pool, padding, input_shape, initial_filters, conv_layer_multiplier, conv_layers,
                 fc_layers, fc_layer_size, dropout, activation, conv_operations):
        layers, _ = _conv_stack(kernel, pool, padding, initial_filters, conv_layer_multiplier, conv_layers,
                                conv_operations, activation, False)
        layers = [Input(shape=input_shape)] + layers + [Flatten()]
        for _ in range(fc_layers):
            layers += [Dense(fc_layer_size, activation=activation), Dropout(dropout)]
        layers.append(Dense(1, activation=activation))
        super().__init__(layers, name=name)


class CNN1D(_CNNClassifier):
    """autoencoder.py:59-116: (1,k) valid convs, (1,2) pools, dense head."""

    def __init__(self, input_shape=_DEFAULT_IN, initial_filters: int = 8, conv_layer_multiplier: int = 2,
                 conv_layers: int = 3, fc_layers: int = 2, fc_layer_size: int = 64, dropout: float = 0.5,
                 activation: str = "elu", kernel_size: int = 3, conv_operations: int = 2):
        super().__init__("cnn1d", (1, kernel_size), (1, 2), "valid", input_shape, initial_filters,
                         conv_layer_multiplier, conv_layers, fc_layers, fc_layer_size, dropout, activation,
                         conv_operations)


class CNNPadded(_CNNClassifier):
    """autoencoder.py:129-185: (k,k) same convs, (2,2) pools, dense head."""

    def __init__(self, input_shape=_DEFAULT_IN, initial_filters: int = 8, conv_layer_multiplier: int = 2,
                 conv_layers: int = 3, fc_layers: int = 2, fc_layer_size: int = 64, dropout: float = 0.5,
                 activation: str = "elu", kernel_size: int = 3, conv_operations: int = 2):
        super().__init__("cnn_padded", (kernel_size, kernel_size), (2, 2), "same", input_shape, initial_filters,
                         conv_layer_multiplier, conv_layers, fc_layers, fc_layer_size, dropout, activation,
                         conv_operations)


class AE(_EncoderDecoder):
    """autoencoder.py:199-249: dense auto-encoder on a flattened (NUM_SNPS, NUM_CHANNELS) input."""

    def __init__(self, input_shape=(global_vars.NUM_SNPS, global_vars.NUM_CHANNELS), fc_layers: int = 3,
                 fc_layer_size: int = 128, dropout: float = 0.5, activation: str = "elu",
                 latent_dimensions: int = 8):
        super().__init__(name="ae")
        enc: List[Layer] = [Input(shape=input_shape), Flatten()]
        for _ in range(fc_layers):
            enc += [Dense(fc_layer_size, activation=activation), Dropout(dropout)]
        enc.append(Dense(latent_dimensions, activation=activation))
        self.encoder = Sequential(enc, name="encoder")
        dec: List[Layer] = [Input(shape=latent_dimensions)]
        dec += [Dense(fc_layer_size, activation=activation) for _ in range(fc_layers)]
        dec.append(Dense(input_shape[0] * input_shape[1], activation="tanh"))
        dec.append(Reshape(input_shape))
        self.decoder = Sequential(dec, name="decoder")


class Sampling(Layer):
    """autoencoder.py:264-272: z = z_mean + exp(0.5 * z_log_var) * epsilon, epsilon ~ N(0, 1).

    The reference defines the layer but no model uses it and no KL term exists there; ``VAEPadded`` (end of this
    module) is where this build uses it, together with the analytic KL to N(0, 1) that the fused kernel returns
    alongside z.  Epsilon comes from a counter-based generator whose position lives on the device (so recorded steps
    keep drawing fresh values); ``replay(eps)`` substitutes supplied values, for replayable tests.

    Two surfaces: ``layer([z_mean, z_log_var])`` as in the reference (eager, returns z; ``last_kl`` holds the KL), and
    ``sample`` / ``sample_backward`` for models of the layer engine."""

    def __init__(self, seed: int = global_vars.DEFAULT_SEED, name=None):
        super().__init__(name)
        self.seed = int(seed)
        self.last_kl = None
        self.stream_id = 0
        self._counter = None
        self._replay = None

    def _position(self, device):
        if self._counter is None or self._counter.device != device:
            self._counter = torch.zeros(1, dtype=torch.int64, device=device)
        return self._counter

    def replay(self, epsilon):
        """Use these epsilon values (array of the (batch, dim) shape of the next calls) instead of drawing; ``None``
        returns to the generator.  The values are copied into a buffer the recorded steps read."""
        if epsilon is None:
            self._replay = None
            return
        from .losses import to_device_f32
        e = to_device_f32(epsilon)
        if self._replay is not None and tuple(self._replay.shape) == tuple(e.shape) and self._replay.device == e.device:
            self._replay.copy_(e)
        else:
            self._replay = e.clone()

    def call(self, inputs, epsilon=None):
        from .losses import to_device_f32
        z_mean = to_device_f32(inputs[0])
        z_log_var = to_device_f32(inputs[1], z_mean.device)
        z = torch.empty_like(z_mean)
        kl = torch.empty(1, dtype=torch.float32, device=z_mean.device)
        part = torch.empty(ops.loss_workspace_floats(), dtype=torch.float32, device=z_mean.device)
        eps_in = to_device_f32(epsilon, z_mean.device) if epsilon is not None else None
        ops.reparam_kl_fwd(z_mean, z_log_var, eps_in, z, None, kl, part, self.seed + 7919 * self.stream_id, 0,
                           self._position(z_mean.device))
        self.last_kl = as_dtensor(kl.view(()))
        return as_dtensor(z)

    __call__ = call

    # ---- layer-engine surface
    def sample(self, ctx, mean, logvar):
        ctx.need_f32(self)
        b, d = mean.shape
        z, eps, kl = ctx.buf(self, "z", (b, d)), ctx.buf(self, "eps", (b, d)), ctx.buf(self, "kl", (1,))
        part = ctx.buf(self, "part", (ops.loss_workspace_floats(),))
        eps_in = self._replay
        if eps_in is not None and tuple(eps_in.shape) != (b, d):
            raise ValueError(f"{self.name}: replayed epsilon {tuple(eps_in.shape)} does not match {(b, d)}")
        ops.reparam_kl_fwd(mean, logvar, eps_in, z, eps, kl, part, self.seed + 7919 * self.stream_id, 0,
                           self._position(mean.device))
        ctx.saved[id(self)] = {"mean": mean, "logvar": logvar, "eps": eps}
        return z, kl

    def sample_backward(self, ctx, dz, kl_weight: float):
        s = ctx.saved[id(self)]
        dmean = ctx.buf(self, "dmean", tuple(dz.shape))
        dlogvar = ctx.buf(self, "dlogvar", tuple(dz.shape))
        ops.reparam_kl_bwd(s["mean"], s["logvar"], s["eps"], dz, dmean, dlogvar, kl_weight)
        return dmean, dlogvar


class CAE1D(_EncoderDecoder):
    """autoencoder.py:275-394: (1,k) valid convs each followed by BatchNorm, (1,2) pools; decoder of
    (1,2) upsampling + (1,k) valid transposed convs.  Width bookkeeping follows the reference's float
    arithmetic (``/= 2``), so odd widths reproduce its Dense/Reshape sizes (and its failures)."""

    def __init__(self, input_shape=_DEFAULT_IN, initial_filters: int = 8, conv_layer_multiplier: int = 2,
                 conv_layers: int = 3, fc_layers: int = 2, fc_layer_size: int = 64, dropout: float = 0.5,
                 activation: str = "elu", kernel_size: int = 3, conv_operations: int = 2,
                 latent_dimensions: int = 8):
        super().__init__(name="cae1d")
        enc, filters = _conv_stack((1, kernel_size), (1, 2), "valid", initial_filters, conv_layer_multiplier,
                                   conv_layers, conv_operations, activation, True)
        width = input_shape[1]
        for _ in range(conv_layers):
            width -= conv_operations * (kernel_size - 1)
            width /= 2
        enc = [Input(shape=input_shape)] + enc
        dec: List[Layer] = []
        if latent_dimensions > 0:
            enc.append(Flatten())
            for _ in range(fc_layers):
                enc += [Dense(fc_layer_size, activation=activation), Dropout(dropout)]
            enc.append(Dense(latent_dimensions, activation=activation))
            dec.append(Input(shape=latent_dimensions))
            dec += [Dense(fc_layer_size, activation=activation) for _ in range(fc_layers)]
            dec.append(Dense(int(input_shape[0] * width * filters), activation=activation))
            dec.append(Reshape((input_shape[0], int(width), int(filters))))
        for block in range(conv_layers):
            final = block + 1 == conv_layers
            dec.append(UpSampling2D((1, 2)))
            for op in range(conv_operations):
                if op + 1 == conv_operations:
                    filters = filters / conv_layer_multiplier
                    dec.append(Conv2DTranspose(global_vars.NUM_CHANNELS if final else filters, (1, kernel_size),
                                               activation="tanh" if final else activation, padding="valid"))
                else:
                    dec.append(Conv2DTranspose(filters, (1, kernel_size), activation=activation, padding="valid"))
        self.encoder = Sequential(enc, name="encoder")
        self.decoder = Sequential(dec, name="decoder")


def _caepadded_decoder_convs(filters, conv_layers, conv_operations, multiplier, kernel_size, activation):
    """Decoder loop of autoencoder.py:496-526: upsample, then ``conv_operations`` convs of which the LAST
    first divides the running filter count by the multiplier (float division, cast by the layer); the
    very last conv emits NUM_CHANNELS with tanh."""
    layers: List[Layer] = []
    for block in range(conv_layers):
        final = block + 1 == conv_layers
        layers.append(UpSampling2D((2, 2)))
        for op in range(conv_operations):
            last_op = op + 1 == conv_operations
            if last_op:
                filters = filters / multiplier
            out_ch = global_vars.NUM_CHANNELS if (final and last_op) else filters
            act = "tanh" if (final and last_op) else activation
            layers.append(Conv2D(out_ch, (kernel_size, kernel_size), activation=act, padding="same"))
    return layers


class CAEPadded(_EncoderDecoder):
    """autoencoder.py:408-536: the convolutional auto-encoder ``run_model.py`` trains.

    Encoder: ``conv_layers`` blocks of ``conv_operations`` same-padded k x k convs (+BatchNorm when
    ``batch_norm``) and a 2x2 max-pool; optional dense bottleneck when ``latent_dimensions > 0``;
    decoder mirrors it with 2x2 nearest upsampling (folded into the following conv's gather here)."""

    def __init__(self, input_shape=_DEFAULT_IN, initial_filters: int = 8, conv_layer_multiplier: int = 2,
                 conv_layers: int = 3, fc_layers: int = 2, fc_layer_size: int = 64, dropout: float = 0.5,
                 activation: str = "elu", kernel_size: int = 3, conv_operations: int = 2,
                 latent_dimensions: int = 8, batch_norm: bool = False):
        super().__init__(name="cae_padded")
        enc, filters = _conv_stack((kernel_size, kernel_size), (2, 2), "same", initial_filters,
                                   conv_layer_multiplier, conv_layers, conv_operations, activation, batch_norm)
        enc = [Input(shape=input_shape)] + enc
        dec: List[Layer] = []
        if latent_dimensions > 0:
            enc.append(Flatten())
            for _ in range(fc_layers):
                enc.append(Dense(fc_layer_size, activation=activation))
                if batch_norm:
                    enc.append(BatchNormalization())
            enc.append(Dense(latent_dimensions, activation=activation))
            dec.append(Input(shape=latent_dimensions))
            dec += [Dense(fc_layer_size, activation=activation) for _ in range(fc_layers)]
            fh = int(input_shape[0] / (2 ** conv_layers))
            fw = int(input_shape[1] / (2 ** conv_layers))
            dec.append(Dense(fh * fw * filters, activation=activation))
            dec.append(Reshape((fh, fw, int(filters))))
        dec += _caepadded_decoder_convs(filters, conv_layers, conv_operations, conv_layer_multiplier, kernel_size,
                                        activation)
        self.encoder = Sequential(enc, name="encoder")
        self.decoder = Sequential(dec, name="decoder")


class VAEPadded(Model):
    """Variational form of ``CAEPadded``.  NO REFERENCE COUNTERPART as a model: the reference ships the ``Sampling`` layer
    (autoencoder.py:264-272) but nothing that uses it.  Built from the reference's pieces: the conv encoder stack and
    dense layers of CAEPadded (autoencoder.py:461-494), two linear Dense heads (z_mean, z_log_var), ``Sampling``, and
    CAEPadded's decoder (autoencoder.py:496-526).

    Training loss = compiled loss(x, decoded) + beta * KL(q(z|x) || N(0, 1)), the KL averaged over the batch; ``fit``
    reports that sum.  ``sampling.replay(eps)`` fixes the noise."""

    def __init__(self, input_shape=_DEFAULT_IN, initial_filters: int = 8, conv_layer_multiplier: int = 2,
                 conv_layers: int = 3, fc_layers: int = 2, fc_layer_size: int = 64, activation: str = "elu",
                 kernel_size: int = 3, conv_operations: int = 2, latent_dimensions: int = 8, beta: float = 1.0,
                 batch_norm: bool = False):
        super().__init__(name="vae_padded")
        if latent_dimensions < 1:
            raise ValueError("VAEPadded needs latent_dimensions >= 1")
        self.beta = float(beta)
        enc, filters = _conv_stack((kernel_size, kernel_size), (2, 2), "same", initial_filters,
                                   conv_layer_multiplier, conv_layers, conv_operations, activation, batch_norm)
        enc = [Input(shape=input_shape)] + enc + [Flatten()]
        for _ in range(fc_layers):
            enc.append(Dense(fc_layer_size, activation=activation))
            if batch_norm:
                enc.append(BatchNormalization())
        self.encoder = Sequential(enc, name="encoder")
        self.z_mean = Dense(latent_dimensions, name="z_mean")
        self.z_log_var = Dense(latent_dimensions, name="z_log_var")
        self.sampling = Sampling()
        dec: List[Layer] = [Input(shape=latent_dimensions)]
        dec += [Dense(fc_layer_size, activation=activation) for _ in range(fc_layers)]
        fh = int(input_shape[0] / (2 ** conv_layers))
        fw = int(input_shape[1] / (2 ** conv_layers))
        dec.append(Dense(fh * fw * filters, activation=activation))
        dec.append(Reshape((fh, fw, int(filters))))
        dec += _caepadded_decoder_convs(filters, conv_layers, conv_operations, conv_layer_multiplier, kernel_size,
                                        activation)
        self.decoder = Sequential(dec, name="decoder")

    def _build(self, in_shape):
        h = self.encoder.build(in_shape)
        self.z_mean.build(h)
        return self.decoder.build(self.z_log_var.build(h))

    def forward(self, ctx, x):
        h = self.encoder.forward(ctx, x)
        mean = self.z_mean.forward(ctx, h)
        logvar = self.z_log_var.forward(ctx, h)
        z, kl = self.sampling.sample(ctx, mean, logvar)
        ctx.extra_losses = [(kl, self.beta)]
        return self.decoder.forward(ctx, z)

    chain_dgrad_ok = True

    def backward(self, ctx, dy, need_dx=True):
        dz = self.decoder.backward(ctx, dy, need_dx=True)
        ctx.flush_dgrad()  # the sampling layer reads dz here, outside any Sequential
        dmean, dlogvar = self.sampling.sample_backward(ctx, dz, self.beta)
        dh = self.z_mean.backward(ctx, dmean, need_dx=True)
        ops.axpy(dh, self.z_log_var.backward(ctx, dlogvar, need_dx=True), 1.0)
        return self.encoder.backward(ctx, dh, need_dx=need_dx)

    def encode(self, x):
        """(z_mean, z_log_var) of a batch, inference mode."""
        from .losses import to_device_f32
        from .nn import CallCtx
        t = to_device_f32(x)
        if not self.built:
            self.build((None,) + tuple(t.shape[1:]))
        self._materialize(t.device)
        ctx = CallCtx(self._pool, ("encode",), False)
        h = self.encoder.forward(ctx, t)
        return as_dtensor(self.z_mean.forward(ctx, h).clone()), as_dtensor(self.z_log_var.forward(ctx, h).clone())

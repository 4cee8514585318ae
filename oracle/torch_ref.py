"""torch-CPU restatement of the tf.keras-2 primitives, models and training step.

Test infrastructure and the timed CPU baseline (see ``oracle/__init__.py``).
It is a *restatement of the reference's semantics*, not the reference: the
reference runs on TensorFlow, which is not installable here.

A model is a list of layer descriptors (dicts, produced by ``oracle/models.py``
from the reference's constructors) executed by :class:`RefSequential` with
torch autograd supplying every gradient.  Parameters are held in the
reference's layouts (Conv2D HWIO, Conv2DTranspose (kh,kw,Cout,Cin), Dense
(in,out)); activations enter and leave as NHWC.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import np_ops


# --------------------------------------------------------------------------- primitives
def _act(z: torch.Tensor, act: Optional[str], alpha: float = 0.2) -> torch.Tensor:
    if act in (None, "linear", "none"):
        return z
    if act == "relu":
        return F.relu(z)
    if act == "elu":
        return F.elu(z, 1.0)
    if act == "lrelu":
        return F.leaky_relu(z, alpha)
    if act == "tanh":
        return torch.tanh(z)
    if act == "sigmoid":
        return torch.sigmoid(z)
    raise ValueError(act)


def conv2d_nchw(x, w_hwio, b, stride, padding):
    """x NCHW, kernel HWIO; SAME padding is asymmetric (extra on the bottom/right)."""
    kh, kw, _, _ = w_hwio.shape
    h, wd = x.shape[2], x.shape[3]
    sh, sw = stride
    if padding == "same":
        _, pt, pb = np_ops.same_pads(h, kh, sh)
        _, pl, pr = np_ops.same_pads(wd, kw, sw)
        if pt or pb or pl or pr:
            x = F.pad(x, (pl, pr, pt, pb))
    # a materialised OIHW copy: torch's float64 CPU conv backward rejects the permuted view when Cout == 1
    w = w_hwio.permute(3, 2, 0, 1).contiguous()
    return F.conv2d(x, w, b, stride=stride)


def conv2d_transpose_nchw(x, w_hwoi, b, stride, padding):
    """x NCHW, kernel (kh,kw,Cout,Cin).  Full scatter output cropped by the SAME
    'before' pad of the forward conv defined on the OUTPUT size."""
    kh, kw, _, _ = w_hwoi.shape
    h, wd = x.shape[2], x.shape[3]
    sh, sw = stride
    ho = np_ops.deconv_out_size(h, kh, sh, padding)
    wo = np_ops.deconv_out_size(wd, kw, sw, padding)
    _, _, pt, pl = np_ops.conv_geometry(ho, wo, kh, kw, sh, sw, padding)
    w = w_hwoi.permute(3, 2, 0, 1)  # torch conv_transpose2d weight: (Cin, Cout, kh, kw)
    full = F.conv_transpose2d(x, w, None, stride=stride)
    # full may be shorter than pt+ho when k < s; pad with zeros in that case
    need_h, need_w = pt + ho, pl + wo
    if full.shape[2] < need_h or full.shape[3] < need_w:
        full = F.pad(full, (0, max(need_w - full.shape[3], 0), 0, max(need_h - full.shape[2], 0)))
    y = full[:, :, pt:pt + ho, pl:pl + wo]
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return y


def maxpool_nchw(x, pool, padding):
    ph, pw = pool
    if padding == "same":
        _, pt, pb = np_ops.same_pads(x.shape[2], ph, ph)
        _, pl, pr = np_ops.same_pads(x.shape[3], pw, pw)
        if pt or pb or pl or pr:
            x = F.pad(x, (pl, pr, pt, pb), value=float("-inf"))
    return F.max_pool2d(x, (ph, pw), (ph, pw))


def upsample_nchw(x, size):
    return x.repeat_interleave(size[0], dim=2).repeat_interleave(size[1], dim=3)


# --------------------------------------------------------------------------- bf16 storage emulation
class _RoundBoth(torch.autograd.Function):
    """A tensor that the bf16 engine writes to memory: the value is rounded to bfloat16 on the way forward and its
    gradient is rounded to bfloat16 on the way back (both are stored as bf16 there)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _RoundFwd(torch.autograd.Function):
    """A float32 master weight read through its bf16 copy: rounded forward, gradient passed through unrounded (kernel
    gradients are accumulated and stored as float32)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def round_act(x):
    return _RoundBoth.apply(x)


# --------------------------------------------------------------------------- losses (R12)
def mse(y_true, y_pred):
    return torch.mean((y_pred - y_true) ** 2)


def mse_weighted(y_true, y_pred, sample_weight):
    per = torch.mean((y_pred - y_true) ** 2, dim=-1) * sample_weight
    return per.sum() / per.numel()


def mae(y_true, y_pred):
    return torch.mean(torch.abs(y_pred - y_true))


def bce_probs(y_true, p):
    eps = np_ops.KERAS_EPS
    p = torch.clamp(p, eps, 1.0 - eps)
    return torch.mean(-(y_true * torch.log(p + eps) + (1 - y_true) * torch.log(1 - p + eps)))


def bce_logits(y_true, z):
    return torch.mean(torch.clamp(z, min=0) - z * y_true + torch.log1p(torch.exp(-torch.abs(z))))


def kl_standard_normal(mu, logvar):
    return torch.mean(-0.5 * torch.sum(1 + logvar - mu * mu - torch.exp(logvar), dim=1))


# --------------------------------------------------------------------------- sequential interpreter
class RefSequential:
    """Executes a list of layer descriptors.  ``params`` (trainable, Keras order:
    kernel, bias / gamma, beta) and ``state`` (BN moving mean, moving variance)
    are flat lists of tensors."""

    def __init__(self, layers: Sequence[dict], input_shape: Sequence[int], dtype=torch.float32,
                 seed: int = 1833, bn_unbiased_moving_var: bool = True, lowp: bool = False):
        # lowp: restate the bf16 policy of the engine - every tensor it stores as bfloat16 (the input, each layer's
        # output, each gradient flowing between layers, the filters the convs read) is rounded at that point, while
        # the arithmetic in between stays in ``dtype`` (pass float64: the engine's float32 sums are exact by comparison)
        self.lowp = bool(lowp)
        self.layers = [dict(l) for l in layers]
        self.input_shape = tuple(int(v) for v in input_shape)
        self.dtype = dtype
        self.bn_unbiased = bn_unbiased_moving_var
        self.params: List[torch.Tensor] = []
        self.state: List[torch.Tensor] = []
        self.param_names: List[str] = []
        self.out_shapes: List[tuple] = []
        gen = torch.Generator().manual_seed(seed)
        shape = self.input_shape
        for li, l in enumerate(self.layers):
            t = l["type"]
            l["_p"] = len(self.params)
            l["_s"] = len(self.state)
            if t in ("conv2d", "conv2d_transpose"):
                kh, kw = l["kernel"]
                cin = shape[-1]
                cout = int(l["filters"])
                wshape = (kh, kw, cin, cout) if t == "conv2d" else (kh, kw, cout, cin)
                self._add_kernel(wshape, l.get("init", "glorot_uniform"), gen, f"{li}.kernel")
                if l.get("use_bias", True):
                    self._add(torch.zeros(cout, dtype=dtype), f"{li}.bias")
                h, w = shape[0], shape[1]
                sh, sw = l.get("strides", (1, 1))
                if t == "conv2d":
                    ho, wo, _, _ = np_ops.conv_geometry(h, w, kh, kw, sh, sw, l["padding"])
                else:
                    ho = np_ops.deconv_out_size(h, kh, sh, l["padding"])
                    wo = np_ops.deconv_out_size(w, kw, sw, l["padding"])
                shape = (ho, wo, cout)
            elif t == "dense":
                fin = int(np.prod(shape))
                assert len(shape) == 1, "Dense expects a flattened input"
                self._add_kernel((fin, int(l["units"])), l.get("init", "glorot_uniform"), gen, f"{li}.kernel")
                self._add(torch.zeros(int(l["units"]), dtype=dtype), f"{li}.bias")
                shape = (int(l["units"]),)
            elif t == "batchnorm":
                c = shape[-1]
                self._add(torch.ones(c, dtype=dtype), f"{li}.gamma")
                self._add(torch.zeros(c, dtype=dtype), f"{li}.beta")
                self.state.append(torch.zeros(c, dtype=dtype))
                self.state.append(torch.ones(c, dtype=dtype))
            elif t == "maxpool":
                ph, pw = l["pool"]
                if l.get("padding", "valid") == "same":
                    shape = (-(-shape[0] // ph), -(-shape[1] // pw), shape[2])
                else:
                    shape = (shape[0] // ph, shape[1] // pw, shape[2])
            elif t == "upsample":
                shape = (shape[0] * l["size"][0], shape[1] * l["size"][1], shape[2])
            elif t == "flatten":
                shape = (int(np.prod(shape)),)
            elif t == "reshape":
                shape = tuple(int(v) for v in l["shape"])
            elif t in ("leakyrelu", "activation", "dropout", "reduce_sum_h"):
                if t == "reduce_sum_h":
                    shape = (shape[1], shape[2])
            else:
                raise ValueError(t)
            self.out_shapes.append(shape)
        self.output_shape = shape

    # ---- parameter helpers
    def _add(self, t, name):
        t.requires_grad_(True)
        self.params.append(t)
        self.param_names.append(name)

    def _add_kernel(self, shape, init, gen, name):
        if init == "glorot_uniform":
            lim = np_ops.glorot_uniform_limit(shape)
            t = (torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * lim
        elif init == "normal0.02":
            t = torch.randn(shape, generator=gen, dtype=torch.float64) * 0.02
        else:
            raise ValueError(init)
        self._add(t.to(self.dtype), name)

    def num_params(self) -> int:
        return sum(p.numel() for p in self.params)

    def load_params(self, arrays: Sequence[np.ndarray]):
        assert len(arrays) == len(self.params)
        with torch.no_grad():
            for p, a in zip(self.params, arrays):
                p.copy_(torch.as_tensor(np.asarray(a), dtype=self.dtype).reshape(p.shape))

    def get_params(self) -> List[np.ndarray]:
        return [p.detach().numpy().copy() for p in self.params]

    # ---- forward
    def __call__(self, x, training: bool = False, dropout_masks: Optional[Dict[int, torch.Tensor]] = None,
                 taps: Optional[Dict[int, torch.Tensor]] = None):
        """x: NHWC (or (B,features)) tensor.  ``taps`` if given receives every
        layer's output in the reference's layout (NHWC)."""
        spatial = x.dim() == 4
        if spatial:
            x = x.permute(0, 3, 1, 2)  # NCHW internally
        lowp = self.lowp
        wq = (lambda w: _RoundFwd.apply(w)) if lowp else (lambda w: w)
        if lowp:
            x = round_act(x)
        for li, l in enumerate(self.layers):
            t = l["type"]
            p = l["_p"]
            if t == "conv2d":
                b = self.params[p + 1] if l.get("use_bias", True) else None
                x = conv2d_nchw(x, wq(self.params[p]), b, l.get("strides", (1, 1)), l["padding"])
                x = _act(x, l.get("activation"), l.get("alpha", 0.2))
            elif t == "conv2d_transpose":
                b = self.params[p + 1] if l.get("use_bias", True) else None
                x = conv2d_transpose_nchw(x, wq(self.params[p]), b, l.get("strides", (1, 1)), l["padding"])
                x = _act(x, l.get("activation"), l.get("alpha", 0.2))
            elif t == "maxpool":
                x = maxpool_nchw(x, l["pool"], l.get("padding", "valid"))
            elif t == "upsample":
                x = upsample_nchw(x, l["size"])
            elif t == "batchnorm":
                x = self._bn(x, l, training, spatial)
            elif t == "leakyrelu":
                x = F.leaky_relu(x, l.get("alpha", 0.3))
            elif t == "activation":
                x = _act(x, l["activation"], l.get("alpha", 0.2))
            elif t == "flatten":
                if spatial:
                    x = x.permute(0, 2, 3, 1)
                    spatial = False
                x = x.reshape(x.shape[0], -1)
            elif t == "reshape":
                x = x.reshape((x.shape[0],) + tuple(l["shape"]))
                if len(l["shape"]) == 3:
                    x = x.permute(0, 3, 1, 2)
                    spatial = True
            elif t == "dense":
                x = x @ self.params[p] + self.params[p + 1]
                x = _act(x, l.get("activation"), l.get("alpha", 0.2))
            elif t == "dropout":
                if training and l["rate"] > 0:
                    if dropout_masks is None or li not in dropout_masks:
                        raise ValueError("training-mode dropout needs an explicit keep mask")
                    x = x * dropout_masks[li] / (1.0 - l["rate"])
            elif t == "reduce_sum_h":
                x = x.sum(dim=2)  # NCHW: sum over H  -> (N, C, W)
                x = x.permute(0, 2, 1)  # (N, W, C) as tf.reduce_sum(axis=1) on NHWC gives
                spatial = False
            else:
                raise ValueError(t)
            if lowp and t in ("conv2d", "conv2d_transpose", "batchnorm", "leakyrelu", "activation"):
                x = round_act(x)
            if taps is not None:
                taps[li] = x.permute(0, 2, 3, 1) if (spatial and x.dim() == 4) else x
        if spatial and x.dim() == 4:
            x = x.permute(0, 2, 3, 1)
        return x

    def _bn(self, x, l, training, spatial, eps=1e-3, momentum=0.99):
        p, s = l["_p"], l["_s"]
        gamma, beta = self.params[p], self.params[p + 1]
        if spatial:
            red, view = (0, 2, 3), (1, -1, 1, 1)
        else:
            red, view = (0,), (1, -1)
        if training:
            mean = x.mean(dim=red)
            var = x.var(dim=red, unbiased=False)
            n = x.numel() // x.shape[1]
            with torch.no_grad():
                mv = var * (n / max(n - 1, 1)) if (spatial and self.bn_unbiased) else var
                self.state[s].mul_(momentum).add_((1 - momentum) * mean.detach())
                self.state[s + 1].mul_(momentum).add_((1 - momentum) * mv.detach())
        else:
            mean, var = self.state[s], self.state[s + 1]
        return (x - mean.view(view)) * torch.rsqrt(var.view(view) + eps) * gamma.view(view) + beta.view(view)


# --------------------------------------------------------------------------- optimiser
class KerasAdam:
    """tf.keras.optimizers.legacy.Adam with an ExponentialDecay(staircase) schedule."""

    def __init__(self, params: Sequence[torch.Tensor], lr0: float, beta1=0.9, beta2=0.999, eps=1e-7,
                 decay_steps: Optional[int] = None, decay_rate: float = 1.0, staircase: bool = True):
        self.params = list(params)
        self.lr0, self.b1, self.b2, self.eps = lr0, beta1, beta2, eps
        self.decay_steps, self.decay_rate, self.staircase = decay_steps, decay_rate, staircase
        self.it = 0
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]

    def lr(self) -> float:
        if self.decay_steps is None:
            return self.lr0
        return np_ops.exponential_decay(self.lr0, self.it, self.decay_steps, self.decay_rate, self.staircase)

    @torch.no_grad()
    def apply_gradients(self, grads: Sequence[torch.Tensor]):
        t = self.it + 1
        alpha = self.lr() * math.sqrt(1 - self.b2 ** t) / (1 - self.b1 ** t)
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            p.addcdiv_(m, v.sqrt().add_(self.eps), value=-alpha)
        self.it += 1


def train_step_mse(model: RefSequential, opt: KerasAdam, x: torch.Tensor, y: torch.Tensor) -> float:
    """One ``Model.fit`` step of run_model.py:281-289: forward, MSE, backward, Adam."""
    for p in model.params:
        p.grad = None
    loss = mse(y, model(x, training=True))
    loss.backward()
    opt.apply_gradients([p.grad for p in model.params])
    return float(loss.detach())

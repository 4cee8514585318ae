"""tf.keras.losses look-alikes backed by the fused loss + dL/dprediction kernels.

Reference call sites: ``losses.MeanSquaredError()`` run_model.py:140,281;
``MeanSquaredError(reduction=...)(y, p, sample_weight=mask[..., 0] * 10)``
run_context_encoder.py:190-198; ``BinaryCrossentropy(from_logits=False)``
run_context_encoder.py:165-187; ``BinaryCrossentropy(from_logits=True)`` and
``MeanAbsoluteError`` run_unet.py:69-161; ``losses.mse`` run_model.py:361-362.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import ops
from .nn import as_dtensor, default_device

_scratch = {}


def _scratch_for(device):
    s = _scratch.get(device)
    if s is None:
        # zero-filled: the tail of the scratch is the arrival counter of the one-launch losses
        s = {"partials": torch.zeros(ops.loss_workspace_floats(), dtype=torch.float32, device=device)}
        _scratch[device] = s
    return s


def to_device_f32(a, device=None) -> torch.Tensor:
    """NumPy (float32/float64) or torch input -> contiguous float32 device tensor, as Keras casts its inputs."""
    device = device or default_device()
    if isinstance(a, torch.Tensor):
        t = a.as_subclass(torch.Tensor)
    else:
        t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device=device, dtype=torch.float32).contiguous()


class Loss:
    """``loss(y_true, y_pred, sample_weight=None)`` -> scalar device tensor.  ``value_and_grad`` is the
    fused form the training steps use: it also writes scale * dL/dy_pred."""

    def __init__(self, reduction: str = "auto", name: Optional[str] = None):
        if reduction not in ("auto", "sum_over_batch_size", "none"):
            raise ValueError(f"reduction {reduction!r} is not supported")
        self.reduction = reduction
        self.name = name or type(self).__name__

    def value_and_grad(self, y_true, y_pred, loss_out, dpred=None, scale: float = 1.0, sample_weight=None,
                       accumulate: bool = False):
        raise NotImplementedError

    def __call__(self, y_true, y_pred, sample_weight=None):
        y_pred = to_device_f32(y_pred)
        y_true = to_device_f32(y_true, y_pred.device)
        if self.reduction == "none":
            return self.per_sample(y_true, y_pred, sample_weight)
        out = torch.empty(1, dtype=torch.float32, device=y_pred.device)
        self.value_and_grad(y_true, y_pred, out, None, 1.0, sample_weight)
        return as_dtensor(out.view(()))

    def per_sample(self, y_true, y_pred, sample_weight=None):
        raise NotImplementedError(f"{self.name}: reduction='none' is not built")


def _weight_view(sample_weight, y_pred):
    """sample_weight of shape y_pred.shape[:-1] (one weight per pixel).  Returns (tensor, stride)."""
    if sample_weight is None:
        return None, 1
    w = to_device_f32(sample_weight, y_pred.device)
    if tuple(w.shape) != tuple(y_pred.shape[:-1]):
        raise ValueError(f"sample_weight shape {tuple(w.shape)} must be {tuple(y_pred.shape[:-1])}")
    return w, 1


class MeanSquaredError(Loss):
    fuses_running_sum = True  # value_and_grad(..., running=, running_weight=) adds the loss to an epoch sum in its launch

    def value_and_grad(self, y_true, y_pred, loss_out, dpred=None, scale=1.0, sample_weight=None, accumulate=False,
                       weight_scale: float = 1.0, weight_stride: Optional[int] = None, running=None,
                       running_weight: float = 0.0):
        if tuple(y_true.shape) != tuple(y_pred.shape):
            raise ValueError(f"MSE: y_true {tuple(y_true.shape)} vs y_pred {tuple(y_pred.shape)}")
        if weight_stride is None:
            w, stride = _weight_view(sample_weight, y_pred)
        else:  # caller passes a raw device tensor read with a stride (channel 0 of an NHWC mask)
            w, stride = sample_weight, weight_stride
        ops.loss_mse(y_true, y_pred, loss_out, _scratch_for(y_pred.device)["partials"], dpred, scale, w, stride,
                     weight_scale, accumulate, running, running_weight)

    def per_sample(self, y_true, y_pred, sample_weight=None):
        w, stride = _weight_view(sample_weight, y_pred)
        out = torch.empty(tuple(y_pred.shape[:-1]), dtype=torch.float32, device=y_pred.device)
        ops.loss_mse_map(y_true, y_pred, out, w, stride, 1.0)
        return as_dtensor(out)


class MeanAbsoluteError(Loss):
    def value_and_grad(self, y_true, y_pred, loss_out, dpred=None, scale=1.0, sample_weight=None, accumulate=False):
        if sample_weight is not None:
            raise NotImplementedError("MeanAbsoluteError: sample_weight is not used by the reference")
        ops.loss_mae(y_true, y_pred, loss_out, _scratch_for(y_pred.device)["partials"], dpred, scale, accumulate)


class BinaryCrossentropy(Loss):
    """Labels in the reference are always ``tf.ones_like`` / ``tf.zeros_like`` of the prediction, so the
    kernel takes one constant label; a label tensor is accepted when it is constant."""

    def __init__(self, from_logits: bool = False, reduction: str = "auto", name=None):
        super().__init__(reduction, name)
        self.from_logits = from_logits

    @staticmethod
    def _const_label(y_true) -> float:
        if isinstance(y_true, (int, float)):
            return float(y_true)
        t = y_true if isinstance(y_true, torch.Tensor) else torch.as_tensor(np.asarray(y_true))
        lo, hi = float(t.min()), float(t.max())
        if lo != hi:
            raise NotImplementedError("BinaryCrossentropy: only constant label tensors (ones_like / zeros_like)")
        return lo

    def value_and_grad(self, y_true, y_pred, loss_out, dpred=None, scale=1.0, sample_weight=None, accumulate=False):
        label = self._const_label(y_true)
        ops.loss_bce(y_pred, label, self.from_logits, loss_out, _scratch_for(y_pred.device)["partials"], dpred, scale,
                     accumulate)

    def __call__(self, y_true, y_pred, sample_weight=None):
        y_pred = to_device_f32(y_pred)
        out = torch.empty(1, dtype=torch.float32, device=y_pred.device)
        self.value_and_grad(y_true, y_pred, out)
        return as_dtensor(out.view(()))


def mse(y_true, y_pred):
    """``losses.mse``: mean over the last axis only (run_model.py:361-362)."""
    return MeanSquaredError(reduction="none")(y_true, y_pred)


def get(loss):
    if isinstance(loss, Loss):
        return loss
    if isinstance(loss, str):
        table = {"mse": MeanSquaredError, "mean_squared_error": MeanSquaredError, "mae": MeanAbsoluteError,
                 "mean_absolute_error": MeanAbsoluteError, "binary_crossentropy": BinaryCrossentropy}
        if loss.lower() in table:
            return table[loss.lower()]()
    raise ValueError(f"unknown loss {loss!r}")

"""tf.keras.optimizers.legacy.Adam + ExponentialDecay, as one launch over a model's flat buffers.

Reference: ``build_optimizer`` run_model.py:77-89, run_unet.py:48-66, run_context_encoder.py:144-162.
The update is the Keras (legacy / ResourceApplyAdam) form, not torch's: epsilon 1e-7 sits outside the
square root and the bias correction is folded into the step size,
``alpha = lr * sqrt(1 - b2^t) / (1 - b1^t)``, ``theta -= alpha * m / (sqrt(v) + eps)``.
The update counter and the decayed learning rate live on the device, so a HIP graph that contains the
step replays the schedule by itself.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Optional, Tuple

import torch

from . import ops


class ExponentialDecay:
    """tf.keras.optimizers.schedules.ExponentialDecay."""

    def __init__(self, initial_learning_rate, decay_steps, decay_rate, staircase=False):
        if decay_steps <= 0:
            raise ValueError("decay_steps must be positive")
        self.initial_learning_rate = float(initial_learning_rate)
        self.decay_steps = int(decay_steps)
        self.decay_rate = float(decay_rate)
        self.staircase = bool(staircase)

    def __call__(self, step: int) -> float:
        p = step / self.decay_steps
        if self.staircase:
            p = math.floor(p)
        return self.initial_learning_rate * self.decay_rate ** p


class Adam:
    def __init__(self, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        if isinstance(learning_rate, ExponentialDecay):
            self.schedule: Optional[ExponentialDecay] = learning_rate
            self.lr0 = learning_rate.initial_learning_rate
        else:
            self.schedule = None
            self.lr0 = float(learning_rate)
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)
        self._state: Optional[torch.Tensor] = None  # int64[3] on the device: update count, {alpha, lr}, arrival counter
        self._slots: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}

    # ---- device state
    def _ensure_state(self, device):
        if self._state is None:
            self._state = torch.zeros(3, dtype=torch.int64, device=device)
        return self._state

    def _slots_for(self, model):
        s = self._slots.get(id(model))
        if s is None:
            s = (torch.zeros_like(model._params), torch.zeros_like(model._params))
            self._slots[id(model)] = s
        return s

    @property
    def iterations(self) -> int:
        return 0 if self._state is None else int(self._state[0].item())

    def current_lr(self) -> float:
        it = self.iterations
        return self.schedule(it) if self.schedule is not None else self.lr0

    def _prepare(self, device):
        st = self._ensure_state(device)
        sc = self.schedule
        ops.adam_prepare(st, self.lr0, self.beta_1, self.beta_2, sc.decay_steps if sc else 0,
                         sc.decay_rate if sc else 1.0, sc.staircase if sc else False)
        return st

    def prepare(self, device):
        """Advance the update counter and refresh the step size on the current stream; the matching
        ``apply_model(..., prepared=True)`` then only runs the parameter update."""
        self._prepare(device)

    # ---- the fused path: one launch over the model's whole flat parameter buffer
    def apply_model(self, model, grad_scale: float = 1.0, prepared: bool = False, zero_grads: bool = False, fold=None,
                    slabs=None):
        """``zero_grads``: the launch also clears the model's gradient buffer (training loops whose next backward
        accumulates into it), which saves the separate fill of the next step.  ``fold``: a pending scalar fold
        ``(partials, n, scale, out, running, running_weight)`` (the loss of a forward launch that left per-workgroup
        sums) carried by the same launch.  ``slabs``: kernel / bias gradients still lying as slabs of partial sums
        (``ops.conv2d_wgrad_batch_slabs``); the one-launch form sums them on the way in."""
        m, v = self._slots_for(model)
        st = self._ensure_state(model._params.device)
        if slabs is not None and prepared:
            raise ValueError("apply_model: gradient slabs need the one-launch form (prepared=False)")
        if prepared:
            if fold is not None:
                ops.fold_scalar(fold[0], fold[1], fold[2], fold[3], False, fold[4], fold[5])
            ops.adam_apply(model._params, model._grads, m, v, st, self.beta_1, self.beta_2, self.epsilon, grad_scale,
                           zero_grads)
            if model._params_lp is not None:
                model._refresh_lp()  # the convs of a bf16 model read this copy
        else:  # schedule, update and (bf16 models) the refresh of the filter copy in one launch
            sc = self.schedule
            ops.adam_step(model._params, model._grads, m, v, st, self.lr0, self.beta_1, self.beta_2, self.epsilon,
                          sc.decay_steps if sc else 0, sc.decay_rate if sc else 1.0, sc.staircase if sc else False,
                          grad_scale, zero_grads, model._params_lp, fold, slabs)
            model._lp_dirty = False
        model._grads_clean = bool(zero_grads)

    # ---- keras surface: optimizer.apply_gradients(zip(grads, variables))
    def apply_gradients(self, grads_and_vars: Iterable):
        pairs = [(g, v) for g, v in grads_and_vars if g is not None]
        if not pairs:
            return
        owners = {id(v.owner) for _, v in pairs}
        model = pairs[0][1].owner
        if len(owners) == 1 and model is not None:
            tv = model.trainable_variables
            whole = len(pairs) == len(tv) and all(
                v is t and isinstance(g, torch.Tensor) and g.data_ptr() == t.grad.data_ptr() for (g, v), t in zip(pairs, tv))
            if whole:
                self.apply_model(model)
                return
        # general path: one launch per variable, moments sliced out of the owner's flat slot buffers
        st = self._prepare(pairs[0][1].data.device)
        for g, var in pairs:
            m, v = self._slots_for(var.owner)
            sl = slice(var.offset, var.offset + var.size)
            g = g.as_subclass(torch.Tensor).to(dtype=torch.float32).contiguous().view(-1)
            ops.adam_apply(var.data.view(-1), g, m[sl], v[sl], st, self.beta_1, self.beta_2, self.epsilon, 1.0)
            if var.owner is not None:
                var.owner._lp_dirty = True

    # ---- checkpointing (not a reference feature; used by the fixed-seed trajectory tests)
    def state_dict(self) -> dict:
        return {"state": None if self._state is None else self._state.cpu(),
                "slots": {k: (m.cpu(), v.cpu()) for k, (m, v) in self._slots.items()}}


def legacy_adam(learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7) -> Adam:
    return Adam(learning_rate, beta_1, beta_2, epsilon)

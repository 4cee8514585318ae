"""tf.keras.mixed_precision look-alike: the policy a model is constructed under decides its compute dtype.

The reference trains in float32 and never touches this API; it exists for the bf16 configuration of the U-Net /
context-encoder steps.  Under ``"mixed_bfloat16"`` a model stores activations, activation gradients and a copy of its
filters as bfloat16 and multiplies them on the bf16 matrix cores; sums, BatchNorm statistics, losses, every weight
gradient, the master weights and the Adam step stay float32 (the split Keras makes between ``compute_dtype`` and
``variable_dtype``).
"""
from __future__ import annotations

import torch

_POLICIES = {"float32": torch.float32, "mixed_bfloat16": torch.bfloat16}
_current = "float32"


class Policy:
    def __init__(self, name: str):
        if name not in _POLICIES:
            raise ValueError(f"unknown policy {name!r}: expected one of {sorted(_POLICIES)}")
        self.name = name
        self.compute_dtype = "bfloat16" if name == "mixed_bfloat16" else "float32"
        self.variable_dtype = "float32"

    @property
    def torch_dtype(self) -> torch.dtype:
        return _POLICIES[self.name]


def set_global_policy(policy) -> None:
    global _current
    name = policy.name if isinstance(policy, Policy) else str(policy)
    Policy(name)  # validates
    _current = name


def global_policy() -> Policy:
    return Policy(_current)

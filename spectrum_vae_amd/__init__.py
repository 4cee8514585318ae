"""MI355X-native training/inference path for spectrum-vae's convolutional models.

Drop-in for the model-class surface of the reference's ``autoencoder.py``, ``unet.py`` and
``context_encoder.py``; every tensor op runs in ``libspectrum_hip.so`` (hand-written gfx950 HIP kernels,
C ABI in ``include/spectrum_hip.h``).  Importing the package loads that library and fails if it is
missing: there is no fallback backend.
"""
from . import _lib  # noqa: F401  (loads libspectrum_hip.so or raises)
from . import autoencoder, context_encoder, global_vars, losses, mixed_precision, nn, optimizers, unet  # noqa: F401
from .model import EarlyStopping, History, Model  # noqa: F401

__all__ = ["autoencoder", "context_encoder", "unet", "global_vars", "losses", "optimizers", "nn", "Model",
           "EarlyStopping", "History"]

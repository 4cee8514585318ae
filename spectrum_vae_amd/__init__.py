"""MI355X-native training/inference path for spectrum-vae's convolutional models.

Drop-in for the model-class surface of the reference's ``autoencoder.py``, ``unet.py`` and
``context_encoder.py``; every tensor op runs in ``libspectrum_hip.so`` (hand-written gfx950 HIP kernels,
C ABI in ``include/spectrum_hip.h``).  Importing the package loads that library and fails if it is
missing: there is no fallback backend.
"""
import os as _os

# The recorded GAN steps fork the two models' backward passes and updates onto two streams (run_unet.py:164-203,
# run_context_encoder.py:225-260 restated in run_unet.py / run_context_encoder.py here).  A graph's branches only run
# side by side when their streams sit on different hardware queues; with HIP's default of 4 queues per process the
# branch usually shares the launch stream's queue and the step is serialised (U-Net step at batch 32: 0.574 ms in five
# of six processes, 0.50 in the sixth; with 8 queues 0.49-0.51 in every one).  The HIP runtime reads the variable when
# it is LOADED: this line takes effect when the package is imported before torch; a program that imports torch first
# exports the variable itself (bench.py and tools/bench_models.py set it at their top).  A value the user exported is
# left alone.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _lib  # noqa: E402,F401  (loads libspectrum_hip.so or raises)
from . import autoencoder, context_encoder, global_vars, losses, mixed_precision, nn, optimizers, unet  # noqa: E402,F401
from .model import EarlyStopping, History, Model  # noqa: E402,F401

__all__ = ["autoencoder", "context_encoder", "unet", "global_vars", "losses", "optimizers", "nn", "Model",
           "EarlyStopping", "History"]

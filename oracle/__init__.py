"""CPU oracle for the spectrum-vae model hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package imports this
directory; it is used by ``tests/``, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py`` as the checker / reported baseline.

PARITY UNPINNED for the model arithmetic: the reference delegates every tensor
op to TensorFlow / tf.keras (``autoencoder.py:4,11-13``, ``unet.py:1,7-8``,
``context_encoder.py:1,7-8``), an unpinned third-party dependency that is not
present in ``/root/reference`` and is not importable here, and the reference
holds no tests, fixtures or golden vectors for it.  The oracle therefore
restates the published tf.keras-2 (TF 2.11-2.15, implied by
``tf.keras.optimizers.legacy.Adam`` at ``run_model.py:88``) semantics twice,
independently:

* ``np_ops``    - naive float64 NumPy loops, one function per primitive;
* ``torch_ref`` - torch-CPU functional restatement with autograd (also the
                  timed CPU baseline),

and the two are cross-checked against each other and against analytic
known-answer cases in ``tests/``.

PINNED by the reference itself: the data-preparation path
(``util.py:34-58,152-256``), whose NumPy code is importable; golden vectors
minted by ``tests/golden/make_dataprep_golden.py`` are committed under
``tests/golden/`` and ``oracle.dataprep`` is checked against them.
"""

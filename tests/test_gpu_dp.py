"""Data-parallel equality on the HIP step: two ranks at batch B each (sharing the one GPU of the test box, gradients
exchanged over gloo) against one process at batch 2B.  The ranks are fresh child processes."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, bpr, steps, out, worker="dp_hip_worker.py", extra=()):
    """``worker`` dp_hip_worker.py: (bpr, steps) are its batch-per-rank and step count; dp_gan_worker.py: they are its
    BACKEND and SHARDS arguments."""
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, worker), str(r), str(world), port, out,
                               str(bpr), str(steps), *extra], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    for p in procs:
        try:
            log, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, log[-3000:]
    return [np.load(os.path.join(out, f"rank{r}.npz")) for r in range(world)]


def test_two_ranks_at_batch_b_equal_one_rank_at_batch_2b(tmp_path):
    steps, bpr = 6, 64
    d2 = tmp_path / "w2"
    d1 = tmp_path / "w1"
    d2.mkdir(); d1.mkdir()
    r0, r1 = _run(2, bpr, steps, str(d2))
    (s0,) = _run(1, 2 * bpr, steps, str(d1))
    # replicas stay bit-identical: same start (broadcast), same summed gradient, same update
    assert np.array_equal(r0["params"], r1["params"])
    # mean-type loss: the single process's loss is the mean of the two shards' losses
    assert np.allclose((r0["losses"] + r1["losses"]) / 2, s0["losses"], rtol=2e-5, atol=0)
    # weights: summed shard gradients / 2 == full-batch gradient up to float32 summation order; after 6 Adam steps of
    # size <= 5e-4 the replicas must sit within a small fraction of one step of the single process
    assert np.max(np.abs(r0["params"] - s0["params"])) <= 1e-5
    moved = np.max(np.abs(s0["params"])) > 0 and steps * 5e-4 > 1e-4
    assert moved



@pytest.mark.parametrize("kind", ["cae_bn", "disc"])
def test_synchronised_batchnorm_two_ranks_equal_one_rank_at_twice_the_batch(tmp_path, kind):
    """BatchNormalization with the reducer's sync_batchnorm: statistics (forward) and the two sums of the input gradient
    (backward) run over the rows of both ranks, so two ranks at batch B are one process at batch 2 B - weights, losses
    and the moving statistics to 1e-5 - for CAEPadded(batch_norm=True) and context_encoder.Discriminator."""
    steps, bpr = 4, 32
    d2, d1 = tmp_path / "w2", tmp_path / "w1"
    d2.mkdir(); d1.mkdir()
    r0, r1 = _run(2, bpr, steps, str(d2), extra=(kind,))
    (s0,) = _run(1, 2 * bpr, steps, str(d1), extra=(kind,))
    assert np.array_equal(r0["params"], r1["params"]) and np.array_equal(r0["state"], r1["state"])
    assert np.allclose((r0["losses"] + r1["losses"]) / 2, s0["losses"], rtol=2e-5, atol=1e-7)
    assert r0["state"].size > 0 and np.max(np.abs(r0["state"] - s0["state"])) <= 1e-5
    assert np.max(np.abs(r0["params"] - s0["params"])) <= 1e-5
    assert np.max(np.abs(s0["params"])) > 0


def test_buckets_sent_during_the_backward_pass_equal_the_reduction_after_it(tmp_path):
    """context_encoder.ContextEncoder on two ranks, 1 MB buckets: with the bucket pipeline the buckets leave last-first
    while the backward pass is still running (all but the ones holding the first layers), and weights, losses and
    BatchNorm state are bit-identical to the steps that reduce the whole buffer after the pass."""
    steps, bpr = 3, 16
    da, db = tmp_path / "overlap", tmp_path / "plain"
    da.mkdir(); db.mkdir()
    a0, a1 = _run(2, bpr, steps, str(da), extra=("ce_overlap",))
    b0, b1 = _run(2, bpr, steps, str(db), extra=("ce_plain",))
    for k in ("params", "losses", "state"):
        assert np.array_equal(a0[k], b0[k]) and np.array_equal(a1[k], b1[k]), k
    assert np.array_equal(a0["params"], a1["params"])
    order = list(a0["order"])
    assert len(order) >= 17 and order == sorted(order, reverse=True)  # 18 MB in 1 MB buckets, the buffer's tail first
    assert int(a0["early"]) >= len(order) - 2 and len(b0["order"]) == 0


PARAM_KEYS = ("ae_params", "ce_params", "ced_params", "un_params", "und_params")
LOSS_KEYS = ("ae_losses", "ce_losses", "un_losses")


def test_recorded_collective_and_gan_steps_equal_the_single_process_steps(tmp_path):
    """One rank with RCCL and the collective forced: the auto-encoder step and both GAN steps run in their
    data-parallel form (bucketed all-reduce of every model's flat gradient buffer, recorded inside the step's HIP graph
    when the probe passes, else between two graphs).  A one-rank sum is the identity, so weights and losses must equal,
    bit for bit, the same steps run serially without any reducer."""
    dn, d0 = tmp_path / "nccl", tmp_path / "none"
    dn.mkdir(); d0.mkdir()
    (a,) = _run(1, "nccl", "same", str(dn), "dp_gan_worker.py")
    (b,) = _run(1, "none", "same", str(d0), "dp_gan_worker.py")
    for k in PARAM_KEYS + LOSS_KEYS:
        assert np.array_equal(a[k], b[k]), k
    # which form ran is decided by the probe; either way the graphs replayed (the worker asserts one reduction per step)
    assert int(a["ae_graph_len"]) == (3 if int(a["capturable"]) else 2)
    assert int(a["ce_graph_len"]) == (1 if int(a["capturable"]) else 2)


def test_gan_steps_two_ranks_over_gloo(tmp_path):
    """Two ranks sharing the GPU (gloo).  Same rows on both ranks: the averaged gradient is exactly the local one, so
    both replicas must equal the single-process run bit for bit.  Different rows: replicas stay bit-identical to each
    other (same start by broadcast, same reduced gradients) and differ from the single-shard run."""
    ds, dd, d0 = tmp_path / "same", tmp_path / "split", tmp_path / "none"
    ds.mkdir(); dd.mkdir(); d0.mkdir()
    s0, s1 = _run(2, "gloo", "same", str(ds), "dp_gan_worker.py")
    (ref,) = _run(1, "none", "same", str(d0), "dp_gan_worker.py")
    for k in PARAM_KEYS + LOSS_KEYS:
        assert np.array_equal(s0[k], s1[k]) and np.array_equal(s0[k], ref[k]), k
    assert int(s0["ae_graph_len"]) == 2 and int(s0["ce_graph_len"]) == 2
    p0, p1 = _run(2, "gloo", "split", str(dd), "dp_gan_worker.py")
    for k in PARAM_KEYS:
        assert np.array_equal(p0[k], p1[k]), k
        assert not np.array_equal(p0[k], ref[k]), k
    assert not np.array_equal(p0["ae_losses"], p1["ae_losses"])

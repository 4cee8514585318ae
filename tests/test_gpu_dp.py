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


def _run(world, bpr, steps, out):
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_hip_worker.py"), str(r), str(world), port, out,
                               str(bpr), str(steps)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
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

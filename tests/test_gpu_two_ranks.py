"""GPU, multi-GPU rehearsal on a one-GPU box: two fresh child processes (ranks 0 and 1 of a gloo process group) share
cuda:0 and run the sharded STOMP and GPMP2 paths; the gathered result must equal the unsharded run of the parent.
This is the N > 1 path of bench.py / parallel.py with real kernels on both ranks (the CPU gloo test moves tensors only)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_on_one_gpu_equal_unsharded(gpu_device, tmp_path):
    sys.path.insert(0, HERE)
    import two_rank_worker as W
    out = str(tmp_path / 'gathered.npz')
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'two_rank_worker.py'), out], env=env))
    for p in procs:
        assert p.wait(timeout=900) == 0
    z = np.load(out)
    dev = gpu_device
    pr = W.problem(dev)
    full = W.run_stomp(pr, dev, 0, pr['P'])
    torch.cuda.synchronize()
    # STOMP: independent particles, Philox keyed by the global particle id -> bit for bit
    assert np.array_equal(z['stomp'], full.cpu().numpy())
    # GPMP2 trust region (quirk Q9): the damping is the batch mean of diag(A^T K A); the two ranks all-reduce their
    # local fp64 sums, the unsharded run sums all particles in one kernel -- the fp64 association differs, the fp32
    # trajectories agree to rounding
    xg = W.run_gpmp2(pr, dev, 0, pr['Bg'], None)
    torch.cuda.synchronize()
    ref = xg.cpu().numpy()
    err = np.abs(z['gpmp2'] - ref).max() / np.abs(ref).max()
    print('sharded vs unsharded GPMP2 rel err', err)
    assert err < 1e-6
    assert np.abs(ref - pr['x0'].cpu().numpy()).max() > 1e-3        # the steps actually moved the trajectories

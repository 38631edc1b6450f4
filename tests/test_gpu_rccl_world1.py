"""-m gpu: RCCL rehearsal on the one GPU of the box.  The N > 1 path of bench.py / parallel.py
talks to torch.distributed's "nccl" backend (= RCCL on ROCm); without an 8-GPU node it had only ever run on gloo.  Here
fresh child processes run it at world size 1 with every collective forced to execute on the MI355X:
init_process_group('nccl', device_id=...), barrier, all_gather of the means, all_reduce (the clock's MAX and GPMP2's
quirk-Q9 damping sums).  No scaling is measured -- the 1 -> 8 curve remains the driver's to take."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from test_gpu_two_ranks import _free_port

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _env(**extra):
    return dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
                HSA_ENABLE_IPC_MODE_LEGACY='0', MPB_FORCE_DIST='1', **extra)


def test_sharded_planners_on_rccl_world1(gpu_device, tmp_path):
    """tests/two_rank_worker.py (sharded STOMP + GPMP2 with the Q9 all-reduce + the final gathers) on the nccl backend:
    the gathered results equal the parent's plain run."""
    sys.path.insert(0, HERE)
    import two_rank_worker as W
    out = str(tmp_path / 'gathered.npz')
    p = subprocess.run([sys.executable, os.path.join(HERE, 'two_rank_worker.py'), out], env=_env(MPB_DIST_BACKEND='nccl'),
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    z = np.load(out)
    dev = gpu_device
    pr = W.problem(dev)
    full = W.run_stomp(pr, dev, 0, pr['P'])
    xg = W.run_gpmp2(pr, dev, 0, pr['Bg'], None)
    torch.cuda.synchronize()
    assert np.array_equal(z['stomp'], full.cpu().numpy())
    ref = xg.cpu().numpy()
    assert np.abs(z['gpmp2'] - ref).max() / np.abs(ref).max() < 1e-6


def test_bench_on_rccl_world1(gpu_device):
    """bench.py itself with the distributed path forced: the line the driver's SCALE run would print at N = 1."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '20', '--warmup', '5',
                        '--no-other-configs', '--no-cpu-baseline'], env=_env(), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line['dist']['backend'] == 'nccl' and line['dist']['forced_at_world_1'] and line['n_gpus'] == 1
    assert line['metric'] == 'stomp_trajectory_update_iters_per_sec' and line['value'] > 1000
    out_dir = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out_dir) and os.access(out_dir, os.W_OK):       # kept for profiles/r03_bench_nccl_world1.json
        with open(os.path.join(out_dir, 'r03_bench_nccl_world1.json'), 'w') as fh:
            fh.write(json.dumps(line) + '\n')


def test_bench_bare_command_launches_its_own_ranks(gpu_device):
    """`python bench.py --gpus N` started BARE (no torch.distributed.run around it, no WORLD_SIZE in the
    environment) -- the driver's command form -- starts its N ranks itself as fresh child processes and relays rank 0's
    line.  Two ranks share the box's one GPU here (MPB_DIST_BACKEND=gloo); on the 8-GPU node the same command runs on RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT',
                                                             'MPB_FORCE_DIST')}
    env.update(MPB_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '20', '--warmup', '5'],
                       env=env, capture_output=True, text=True, timeout=1100)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['dist']['world'] == 2 and line['dist']['rccl_ranks_seen'] == 2
    assert line['dist']['backend'] == 'gloo' and line['scaling'] == 'weak' and 'c5' in line['scaling_note']
    assert line['metric'] == 'stomp_trajectory_update_iters_per_sec' and line['value'] > 1000
    assert line['c5']['value'] > 0 and 'cpu_baseline' not in line
    #  the N > 1 line carries c5 with its roofline (VALU issue from the committed PMC pass of its own
    # instantiation, HBM from the algorithmic bytes, counter traffic) and says what its scaling figures mean
    r5 = line['c5']['roofline']
    assert r5['frac'] > 0 and r5['unit'] in ('G wave-instr/s', 'GB/s') and 'traffic' in r5
    assert 'c5' in line['scaling_note'] and line['c5']['scaling'] == 'weak' 
    # a mismatch between --gpus and an existing WORLD_SIZE is an error message, not an assert
    bad = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1'],
                         env=dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0'), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and 'WORLD_SIZE' in bad.stderr

"""CPU, build container only: the committed recipe tests/golden/make_goldens.py runs end to end against the unmodified
reference at /root/reference and reproduces every committed fixture bit for bit.  Skipped where the reference is
absent (the GPU box)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

REFERENCE = '/root/reference'


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, 'mp_baselines')), reason='reference checkout not present')
def test_generator_reproduces_committed_fixtures(tmp_path):
    env = dict(os.environ, MPB_GOLDEN_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, 'make_goldens.py')], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    made = sorted(f for f in os.listdir(tmp_path) if f.endswith('.npz'))
    committed = sorted(f for f in os.listdir(GOLDEN) if f.endswith('.npz'))
    assert made == committed, (set(made) ^ set(committed))
    for f in made:
        a = np.load(os.path.join(tmp_path, f), allow_pickle=False)
        b = np.load(os.path.join(GOLDEN, f), allow_pickle=False)
        assert sorted(a.files) == sorted(b.files), f
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (f, k)
            assert a[k].tobytes() == b[k].tobytes(), (f, k)

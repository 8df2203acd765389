"""Two data-parallel ranks of the trainer on the one GPU of the test box (gloo carries the gradient buckets):
the gradient exchange driven from autograd's hooks during the backward pass, and the one after it, must leave both
ranks with identical parameters.  The multi-GPU run proper uses RCCL (bench.py --gpus N)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('mode', ['overlapped', 'after'])
def test_two_rank_trainer_keeps_ranks_identical(mode):
    env = dict(os.environ)
    if mode == 'after':
        env['DDP_SMOKE_AFTER'] = '1'
    out = subprocess.run([sys.executable, '-u', os.path.join(ROOT, 'tools', 'ddp_smoke.py')], env=env, capture_output=True,
                         text=True, timeout=110)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert 'OK' in out.stdout

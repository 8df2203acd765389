"""`bench.py --gpus N` without a launcher starts the N ranks itself: environment and argument plumbing, failure
propagation, and the refusal when fewer GPUs are visible (no GPU needed: the child processes are faked)."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class FakeProc:
    def __init__(self, cmd, env, rc):
        self.cmd, self.env, self.rc, self.terminated, self.waited = cmd, env, rc, False, False

    def wait(self):
        self.waited = True
        return self.rc

    def poll(self):
        return self.rc if self.waited else None

    def terminate(self):
        self.terminated = True


def run_launcher(monkeypatch, argv, devices, rcs):
    bench = load_bench()
    started = []

    def popen(cmd, env):
        proc = FakeProc(cmd, env, rcs[len(started)])
        started.append(proc)
        return proc

    import torch

    def forbidden():
        raise AssertionError('the launcher parent must not ask the HIP runtime for the device count')
    monkeypatch.setattr(torch.cuda, 'device_count', forbidden)
    monkeypatch.setattr(bench, 'visible_gpus', lambda: devices)
    monkeypatch.setattr(bench.subprocess, 'Popen', popen)
    monkeypatch.setattr(sys, 'argv', ['bench.py'] + argv)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.delenv('MASTER_PORT', raising=False)
    with pytest.raises(SystemExit) as stop:
        bench.main()
    return stop.value.code, started


def test_launcher_starts_one_rank_per_gpu(monkeypatch):
    code, procs = run_launcher(monkeypatch, ['--gpus', '4', '--steps', '3', '--warmup', '1'], devices=8, rcs=[0, 0, 0, 0])
    assert code == 0 and len(procs) == 4
    ports = {p.env['MASTER_PORT'] for p in procs}
    assert len(ports) == 1 and int(ports.pop()) > 0
    for rank, p in enumerate(procs):
        assert p.cmd[0] == sys.executable and p.cmd[1].endswith('bench.py')
        assert p.cmd[2:] == ['--gpus', '4', '--steps', '3', '--warmup', '1']
        assert (p.env['RANK'], p.env['LOCAL_RANK'], p.env['WORLD_SIZE']) == (str(rank), str(rank), '4')
        assert p.env['MASTER_ADDR'] == '127.0.0.1'


def test_launcher_reports_a_failed_rank(monkeypatch):
    code, procs = run_launcher(monkeypatch, ['--gpus', '2'], devices=2, rcs=[0, 7])
    assert code == 7


def test_launcher_refuses_without_enough_gpus(monkeypatch):
    code, procs = run_launcher(monkeypatch, ['--gpus', '2'], devices=1, rcs=[])
    assert 'needs 2 GPUs' in str(code) and not procs


def test_rank_refuses_mismatched_world(monkeypatch):
    bench = load_bench()
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2'])
    monkeypatch.setenv('WORLD_SIZE', '4')
    with pytest.raises(SystemExit, match='WORLD_SIZE is 4'):
        bench.main()


def test_visible_gpus_reads_the_environment_first(monkeypatch):
    bench = load_bench()
    monkeypatch.delenv('ROCR_VISIBLE_DEVICES', raising=False)
    monkeypatch.delenv('CUDA_VISIBLE_DEVICES', raising=False)
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,2,5')
    assert bench.visible_gpus() == 3
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '')
    assert bench.visible_gpus() == 0
    monkeypatch.delenv('HIP_VISIBLE_DEVICES')
    count = bench.visible_gpus()        # the driver's topology, or None where there is none to read
    assert count is None or count >= 0

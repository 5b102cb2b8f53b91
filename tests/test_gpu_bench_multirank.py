"""GPU: the N > 1 code path of bench.py end to end on a 1-GPU box.  Two ranks share GPU 0 and talk over gloo (RCCL
refuses two ranks on one device): self-launch from a bare shell, rank discovery, rank 0 encodes + broadcast, the
configs[3] grid sharded over the ranks, per-rank timing exchange, the secondary configs[1] grid, ONE JSON line from
rank 0.  Timings of such a run mean nothing; what is checked is that the harness the driver will launch on 8 GPUs
runs and reports the right workload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_gpu():
    env = dict(os.environ, OCC4D_BENCH_SHARE_GPU='1', OCC4D_BENCH_BACKEND='gloo')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1',
                           '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, proc.stdout[-2000:]                    # rank 0 only
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong' and d['steps'] == 1
    cfg = d['config']
    assert 'configs[3]' in cfg['workload'] and '2125568 grid queries, 1062784 per GPU' in cfg['workload']
    assert cfg['rccl_ranks'] == 2 and cfg['backend'] == 'gloo' and cfg['ranks_share_one_gpu'] is True
    assert len(cfg['per_rank_ms_per_step']) == 2 and all(t > 0 for t in cfg['per_rank_ms_per_step'])
    assert abs(d['value'] - 2125568 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6
    s2 = d['strong_config2']
    assert s2['n_gpus'] == 2 and '534528 queries (267264 per GPU)' in s2['workload']
    assert 0.0 < d['roofline']['frac'] <= 1.0 and d['roofline']['launches'] > 0
    assert 'cpu_baseline' not in d and 'alt_precision' not in d    # single-GPU legs stay off the N > 1 line
    # the exchange step is reported per rank: only rank 0 encodes; every rank takes part in the (one, packed) broadcast
    assert len(cfg['encode_ms_per_rank']) == 2 and cfg['encode_ms_per_rank'][0] > 1.0 > cfg['encode_ms_per_rank'][1] >= 0.0
    assert len(cfg['broadcast_ms_per_rank']) == 2 and all(t >= 0.0 for t in cfg['broadcast_ms_per_rank'])
    # N > 1 is timed on the PIPELINED schedule (rank 0's encode + the one packed broadcast of clip i + 1 beside every rank's
    # decode of clip i); the sequential schedule of rounds 1-5 is the informational leg, with identical outputs
    assert cfg['schedule'].startswith('pipelined') and 'one packed broadcast' in cfg['parallelism']
    p = d['sequential']
    assert 'pipelined' not in d and p['ms_per_step'] > 0 and p['max_abs_diff_vs_timed_schedule'] <= 1e-6


def test_bench_refuses_a_mismatched_launch():
    """--gpus 2 under a 1-rank launcher environment: a clear message and exit code 2, not an assert."""
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, capture_output=True,
                          text=True, timeout=300)
    assert proc.returncode == 2 and 'WORLD_SIZE=1' in proc.stderr


def test_bench_train_two_ranks_on_one_gpu():
    """bench_train.py (BASELINE configs[4], the DDP training step) launched as `--gpus 2` from a bare shell: self-launch,
    two ranks (sharing GPU 0, gloo) that each run the full-size step, the flat gradient all-reduce between them, ONE
    JSON line from rank 0 with roofline / rccl_ranks / per-rank times / all-reduce time."""
    env = dict(os.environ, OCC4D_BENCH_SHARE_GPU='1', OCC4D_BENCH_BACKEND='gloo')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench_train.py'), '--gpus', '2', '--steps', '1', '--warmup', '1'],
                          env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, proc.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['steps'] == 1 and d['unit'] == 'examples/s'
    cfg = d['config']
    assert 'configs[4]' in cfg['workload'] and cfg['rccl_ranks'] == 2 and cfg['backend'] == 'gloo'
    assert cfg['ranks_share_one_gpu'] is True and len(cfg['per_rank_ms_per_step']) == 2
    assert cfg['allreduce_ms'] is not None and cfg['allreduce_ms'] > 0
    assert abs(d['value'] - 2 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6         # whole-job examples / s
    r = d['roofline']
    assert r['bound'] == 'mfma' and 0.0 < r['frac'] <= 1.0 and r['frac_as_written'] > 0
    assert 8.0 < r['as_written_tflop_per_step'] < 12.0 and 4.0 < r['executed_tflop_per_step'] < 12.0
    assert r['dominant'] in r['kernels'] and r['kernels'][r['dominant']]['launches'] > 0
    assert all(np_isfinite(v) for v in d['losses'])


def np_isfinite(v):
    return v == v and abs(v) != float('inf')


def test_bench_train_refuses_what_it_cannot_run():
    """`--gpus 2` on a 1-GPU box from a bare shell: "needs 2 visible GPUs", exit code 2; a mismatched launcher
    environment: exit code 2 with the reason -- no assert, no traceback."""
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'OCC4D_BENCH_SHARE_GPU'):
        env.pop(k, None)
    import torch
    if torch.cuda.device_count() < 2:
        proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench_train.py'), '--gpus', '2'], env=env,
                              capture_output=True, text=True, timeout=300)
        assert proc.returncode == 2 and 'needs 2 visible GPUs' in proc.stderr
    env.update(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench_train.py'), '--gpus', '2'], env=env, capture_output=True,
                          text=True, timeout=300)
    assert proc.returncode == 2 and 'WORLD_SIZE=1' in proc.stderr

"""bench.py's N-rank control flow without a GPU (the driver's `python bench.py --gpus N` on an 8-GPU node).

MGX_BENCH_STUB=1 swaps the engine's work for a per-rank sleep and the backend for gloo; everything around it is the code the
GPU run executes: the plain-command launcher (bench.py re-executes itself under torch.distributed.run), the process group,
the per-rank action tapes, the end-of-rollout gather, the all_reduce(MAX) of the ranks' times and the ONE JSON line on the
real stdout (native libraries' chatter goes to stderr)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, launcher=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE', 'MASTER_PORT', 'MASTER_ADDR')}
    env['MGX_BENCH_STUB'] = '1'
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, 'bench.py')] + list(argv)
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, ('exactly one line on stdout', res.stdout)
    return json.loads(lines[0]), res.stderr


@pytest.mark.timeout(300, method='thread')
def test_plain_command_launches_two_ranks_and_emits_one_line():
    out, err = _run('--gpus', '2', '--steps', '10', '--warmup', '2', '--envs', '16')
    assert out['n_gpus'] == 2 and out['collective']['world_size'] == 2 and out['collective']['backend'] == 'gloo'
    assert out['collective']['gathered_rows'] == 32                       # both ranks' shards came through the gather
    st = out['stub']
    assert st['tape_seeds'] == [0, 1] and st['tape_crcs'][0] != st['tape_crcs'][1]        # each rank its own action tape
    from bench import tape_seed
    want = [int(__import__('zlib').crc32(np.random.RandomState(tape_seed(r)).randint(0, 18, size=(12, 16)).astype(np.int32).tobytes())) for r in range(2)]
    assert st['tape_crcs'] == want
    # the job's time is the slowest rank's: rank 1 sleeps twice as long as rank 0
    assert st['elapsed_per_rank'][1] > 1.5 * st['elapsed_per_rank'][0]
    assert st['elapsed_max'] == max(st['elapsed_per_rank']) and abs(out['ms_per_step'] - st['elapsed_max'] / 10 * 1e3) < 1e-9
    assert abs(out['value'] - 16 * 2 * 10 / st['elapsed_max']) < 1e-6 * out['value']      # whole-job throughput over the MAX
    assert out['steps'] == 10 and out['warmup'] == 2 and out['scaling'] == 'weak'


@pytest.mark.timeout(300, method='thread')
def test_config5_plain_command_two_ranks():
    out, err = _run('--gpus', '2', '--config5', '--envs5', '64', '--steps', '5', '--warmup', '1')
    assert out['n_gpus'] == 2 and out['collective']['world_size'] == 2 and out['collective']['gathered_rows'] == 64
    assert out['config']['envs_per_task_per_gpu'] == 32 and out['scaling'] == 'strong'
    assert abs(out['value'] - 8 * 64 * 5 / out['stub']['elapsed_max']) < 1e-6 * out['value']


@pytest.mark.timeout(300, method='thread')
def test_ranks_started_by_the_driver_are_not_relaunched():
    """The contract's own launch line: torch.distributed.run around bench.py (WORLD_SIZE is set: bench.py must NOT launch again)."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    launcher = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port', str(port)]
    out, err = _run('--gpus', '2', '--steps', '4', '--warmup', '1', '--envs', '8', launcher=launcher)
    assert out['n_gpus'] == 2 and out['collective']['world_size'] == 2 and out['stub']['tape_seeds'] == [0, 1]


def test_world_size_mismatch_is_an_error_not_an_assertion():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(MGX_BENCH_STUB='1', WORLD_SIZE='1')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '0', '--envs', '4'],
                         capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert res.returncode != 0 and 'WORLD_SIZE=1' in res.stderr and 'AssertionError' not in res.stderr


# ---- the first 8-GPU lease, as far as it can be rehearsed without one (verdict r5 item 8): world size 8 under gloo -------------------------
def _port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.timeout(600, method='thread')
def test_plain_command_eight_ranks():
    """`python bench.py --gpus 8` as the driver types it on an 8-GPU node: eight ranks, eight tapes, 8 x n gathered rows, the job's time
    the slowest rank's, and every rank's host pool sized cores / 8 (the ranks of a node share its cores: LOCAL_WORLD_SIZE)."""
    out, err = _run('--gpus', '8', '--steps', '5', '--warmup', '1', '--envs', '16', timeout=500)
    assert out['n_gpus'] == 8 and out['collective']['world_size'] == 8 and out['collective']['gathered_rows'] == 128
    st = out['stub']
    assert st['tape_seeds'] == list(range(8)) and len(set(st['tape_crcs'])) == 8
    assert st['elapsed_max'] == max(st['elapsed_per_rank']) and st['elapsed_per_rank'][7] > 3 * st['elapsed_per_rank'][0]
    assert abs(out['value'] - 16 * 8 * 5 / st['elapsed_max']) < 1e-6 * out['value'] and out['scaling'] == 'weak'
    assert st['local_world_size'] == 8
    assert st['host_pool_threads'] == [max(1, min(64, st['cores'] // 8))] * 8, st


@pytest.mark.timeout(600, method='thread')
def test_config5_plain_command_eight_ranks_at_full_size():
    """BASELINE.json configs[4] at its own size: 8192 envs per task sharded 8 ways = 1024 per task per rank, one gather of the [8192, 8]
    score table."""
    out, err = _run('--gpus', '8', '--config5', '--steps', '3', '--warmup', '1', timeout=500)
    assert out['n_gpus'] == 8 and out['collective']['world_size'] == 8 and out['collective']['gathered_rows'] == 8192
    assert out['config']['envs_per_task_per_gpu'] == 1024 and out['scaling'] == 'strong'
    assert abs(out['value'] - 8 * 8192 * 3 / out['stub']['elapsed_max']) < 1e-6 * out['value']
    assert out['stub']['local_world_size'] == 8 and out['stub']['host_pool_threads_rank0'] == max(1, min(64, out['stub']['cores'] // 8))


@pytest.mark.timeout(600, method='thread')
def test_rollout_all_tasks_eight_ranks_equal_one_rank():
    """tools/rollout_all_tasks.py (the config-5 rollout tool) under torch.distributed.run with 8 ranks: the gathered score table of every
    task is the one-rank run's (seeds by global env index, tape slices by shard), 8192 rows."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE', 'MASTER_PORT', 'MASTER_ADDR')}
    env['MGX_BENCH_STUB'] = '1'
    tool = os.path.join(ROOT, 'tools', 'rollout_all_tasks.py')
    runs = []
    for launcher in ([sys.executable], [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
                                        '--master-port', str(_port())]):
        res = subprocess.run(launcher + [tool, '--envs', '8192'], capture_output=True, text=True, timeout=500, env=env, cwd=ROOT)
        assert res.returncode == 0, res.stderr[-3000:]
        runs.append([json.loads(ln) for ln in res.stdout.splitlines() if ln.startswith('{')])
    one, eight = runs
    assert len(one) == len(eight) == 8 and [r['n_gpus'] for r in eight] == [8] * 8 and [r['n_gpus'] for r in one] == [1] * 8
    for a, b in zip(one, eight):
        assert a['task'] == b['task'] and b['gathered_rows'] == 8192 and a['stub_scores_crc'] == b['stub_scores_crc'], (a, b)


def test_local_rank_to_device_mapping():
    """One process per GPU: LOCAL_RANK is the device index where every rank sees the whole node, and 0 where the launcher shows each rank
    one device (HIP_VISIBLE_DEVICES per rank); fewer visible devices than ranks, but more than one, is refused."""
    from magical_amd.distributed import device_index_for_local_rank as f
    assert [f(r, 8) for r in range(8)] == list(range(8))
    assert [f(r, 1) for r in range(8)] == [0] * 8
    assert f(0, 4) == 0 and f(3, 4) == 3
    with pytest.raises(RuntimeError):
        f(5, 4)
    with pytest.raises(RuntimeError):
        f(0, 0)

"""N > 1 path on CPU: world_size-2 gloo processes shard the env range and gather per-env results."""
import os
import socket

import numpy as np
import pytest

from magical_amd.distributed import env_shard


def test_env_shard_partitions_exactly():
    for n, w in [(8192, 8), (4096, 3), (10, 4), (5, 8)]:
        ranges = [env_shard(n, r, w) for r in range(w)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        sizes = [hi - lo for lo, hi in ranges]
        assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n_total, out_dir):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from magical_amd.distributed import env_shard, gather_rollout_results, init_from_env
    r, w, _ = init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    lo, hi = env_shard(n_total, rank, world)
    # per-env "scores" that encode the global env index, plus a pose-like payload
    scores = torch.arange(lo, hi, dtype=torch.float64) / n_total
    poses = torch.stack([torch.arange(lo, hi, dtype=torch.float32), torch.full((hi - lo,), float(rank))], dim=1)
    all_scores = gather_rollout_results(scores, n_total)
    all_poses = gather_rollout_results(poses, n_total)
    np.save(os.path.join(out_dir, f'scores_{rank}.npy'), all_scores.numpy())
    np.save(os.path.join(out_dir, f'poses_{rank}.npy'), all_poses.numpy())
    # equal shards (the weak-scaling bench: every rank the same env count): one collective straight into the result
    even = torch.arange(rank * 4, rank * 4 + 4, dtype=torch.float64).reshape(4, 1).repeat(1, 3)
    np.save(os.path.join(out_dir, f'even_{rank}.npy'), gather_rollout_results(even, 4 * world).numpy())
    np.save(os.path.join(out_dir, f'even_unsized_{rank}.npy'), gather_rollout_results(even).numpy())      # sizes exchanged, not derived
    dist.destroy_process_group()


def test_gather_rollout_results_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    n_total, world = 11, 2      # uneven shards: 6 + 5
    mp.start_processes(_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True, start_method='spawn')
    for rank in range(world):
        scores = np.load(tmp_path / f'scores_{rank}.npy')
        poses = np.load(tmp_path / f'poses_{rank}.npy')
        assert np.array_equal(scores, np.arange(n_total) / n_total)
        assert np.array_equal(poses[:, 0], np.arange(n_total, dtype=np.float32))
        assert np.array_equal(poses[:, 1], np.array([0.0] * 6 + [1.0] * 5, dtype=np.float32))
        want = np.repeat(np.arange(8, dtype=np.float64)[:, None], 3, axis=1)
        assert np.array_equal(np.load(tmp_path / f'even_{rank}.npy'), want) and np.array_equal(np.load(tmp_path / f'even_unsized_{rank}.npy'), want)


# ---- RCCL on the one GPU of the box: a process group of ONE rank, backend "nccl" (= RCCL on ROCm), and the payloads
# of the end-of-rollout gather pushed through it -- the collective really runs (gather_rollout_results does not return early in a
# group), the device tensors come back equal, and librccl is mapped into the process.  In a subprocess: a hung rendezvous or
# communicator set-up must not take the test process with it.
_RCCL_SCRIPT = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
from magical_amd.distributed import gather_rollout_results, init_from_env
assert torch.cuda.is_available()
torch.cuda.set_device(0)
rank, world, _ = init_from_env(backend='nccl', single_process_group=True)
assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == 'nccl'
g = torch.Generator().manual_seed(0)
scores = torch.rand((1024, 8), dtype=torch.float64, generator=g).to('cuda:0')            # config 5: one rank's [envs, tasks] score table
poses = torch.rand((1024, 14, 3), dtype=torch.float64, generator=g).to('cuda:0')         # a [n, B, 3] pose payload
flags = (torch.rand((1000,), generator=g) < 0.25).to('cuda:0')                           # a ragged bool payload
for x in (scores, poses, flags.to(torch.uint8)):
    y = gather_rollout_results(x, x.shape[0])
    assert y is not x and y.device == x.device and y.dtype == x.dtype and torch.equal(x, y)
t = torch.ones(4, device='cuda:0'); dist.all_reduce(t); assert float(t.sum()) == 4.0
torch.cuda.synchronize()
libs = sorted({ln.split()[-1] for ln in open('/proc/self/maps') if 'rccl' in ln.lower() or 'nccl' in ln.lower()})
print('RCCL_LIBS', libs)
assert libs, 'no rccl library mapped'
dist.destroy_process_group()
print('RCCL_OK')
'''


@pytest.mark.gpu
@pytest.mark.timeout(300, method='thread')
def test_rccl_one_rank_gather_on_the_gpu():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    res = subprocess.run([sys.executable, '-c', _RCCL_SCRIPT, root], capture_output=True, text=True, timeout=280, env=env)
    assert res.returncode == 0 and 'RCCL_OK' in res.stdout, f'stdout:\n{res.stdout}\nstderr:\n{res.stderr[-3000:]}'
    print(res.stdout)


def test_one_rank_group_goes_through_the_collective_gloo():
    """The same on CPU (gloo): in a group of one rank the gather is a real collective, not an early return."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = ("import sys; sys.path.insert(0, sys.argv[1]); import torch, torch.distributed as dist\n"
              "from magical_amd.distributed import gather_rollout_results, init_from_env\n"
              "assert init_from_env(backend='gloo', single_process_group=True) == (0, 1, 0) and dist.get_world_size() == 1\n"
              "x = torch.arange(12, dtype=torch.float64).reshape(6, 2); y = gather_rollout_results(x, 6)\n"
              "assert y is not x and torch.equal(x, y); dist.destroy_process_group(); print('OK')\n")
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    res = subprocess.run([sys.executable, '-c', script, root], capture_output=True, text=True, timeout=120, env=env)
    assert res.returncode == 0 and 'OK' in res.stdout, res.stderr[-2000:]

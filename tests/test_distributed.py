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

"""Multi-GPU layer: envs are independent (no cross-env term anywhere in base_env.py:255-292), so the path
shards by env index -- one process per GPU, contiguous env ranges, the world template replicated -- with NO
data-path collective.  The only exchange is the end-of-rollout gather of per-env results (scores, optionally
final poses) over RCCL/xGMI (`backend="nccl"` on ROCm); the payload is KBs, so it is latency-bound, and
observations never leave the GPU that rendered them.
"""
import os


def env_shard(n_total, rank, world_size):
    """Contiguous env-index range [lo, hi) owned by `rank` (the first n_total % world ranks get one extra)."""
    base, extra = divmod(int(n_total), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init_from_env(backend=None):
    """Join the torch.distributed group described by RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns
    (rank, world_size, local_rank); a no-op for single-process runs."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC only on this driver
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def gather_rollout_results(local, n_total=None):
    """All-gather a per-env result tensor [n_local, ...] along dim 0 in rank order -> [n_total, ...] on every
    rank.  Shards may differ in length by one (env_shard), so this pads to the longest shard."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(s.item()) for s in sizes]
    n_max = max(sizes)
    padded = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    parts = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    out = torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)
    if n_total is not None:
        assert out.shape[0] == n_total
    return out

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


def init_from_env(backend=None, single_process_group=False):
    """Join the torch.distributed group described by RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns
    (rank, world_size, local_rank).  A single-process run joins no group -- unless `single_process_group`: then a group of ONE
    rank is formed on 127.0.0.1 (a free port unless MASTER_PORT is set), so that the end-of-rollout gather really goes through the
    backend (RCCL on a GPU box) on the one GPU there is."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or single_process_group) and not dist.is_initialized():
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC only on this driver
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            if 'MASTER_PORT' not in os.environ:
                import socket
                with socket.socket() as s:
                    s.bind(('127.0.0.1', 0))
                    os.environ['MASTER_PORT'] = str(s.getsockname()[1])
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def device_index_for_local_rank(local_rank, n_visible=None):
    """The HIP device index a rank uses.  One process per GPU, and two ways a launcher hands the GPUs out: every rank sees the whole node
    (torchrun as the driver runs it: device = LOCAL_RANK), or every rank sees ONE device (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES set per
    rank: the only device it has is index 0 whatever its LOCAL_RANK).  Anything in between (fewer visible devices than ranks on the node,
    more than one) is a mis-launch and raises instead of silently doubling ranks up on a GPU."""
    if n_visible is None:
        import torch
        n_visible = torch.cuda.device_count()
    local_rank, n_visible = int(local_rank), int(n_visible)
    if n_visible < 1:
        raise RuntimeError('no HIP device visible to this rank (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES empty?)')
    if local_rank < n_visible:
        return local_rank
    if n_visible == 1:
        return 0
    raise RuntimeError(f'LOCAL_RANK={local_rank} but only {n_visible} HIP devices are visible: give every rank the whole node or exactly one device')


def gather_rollout_results(local, n_total=None):
    """All-gather a per-env result tensor [n_local, ...] along dim 0 in rank order -> [n_total, ...] on every
    rank.  Shards may differ in length by one (env_shard), so this pads to the longest shard.  Without a process group the
    local tensor is the result; in a group of one rank (init_from_env(single_process_group=True)) the collective still runs."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size()
    if n_total is not None:
        # the job's env count is known: every rank's shard length follows from env_shard, nothing to exchange or wait for
        sizes = [hi - lo for lo, hi in (env_shard(n_total, r, world) for r in range(world))]
        assert local.shape[0] == sizes[dist.get_rank()], (local.shape[0], sizes, 'not the env_shard of n_total')
    else:
        n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        got = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(got, n_local)
        sizes = [int(s.item()) for s in got]
    n_max = max(sizes)
    if min(sizes) == n_max:
        # equal shards: one collective straight into the result
        out = torch.empty((world * n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    padded = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    parts = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)


class TaskFleet:
    """Several engines (e.g. one per task: SURVEY.md section 8d config 5 = the 8 Demo tasks, 8192 envs each, 1024 per GPU) on ONE
    GPU, each stepping on its own HIP stream, so that one engine's latency-bound `k_step` (a single wavefront per SIMD) runs
    under another's memory-bound `k_raster`: small per-task batches no longer leave most of the chip idle.  Engines are
    independent (no data crosses between them); `step()` issues one env-step of every engine that still has steps to go and
    returns without joining the streams -- the host only synchronises where an engine's episode ends (its scoring)."""

    def __init__(self, names, n_envs, device, seed=0, first_env=0, concurrent=True, **make_kwargs):
        import torch
        import magical_amd
        self.device = torch.device(device)
        self.names, self.concurrent = list(names), concurrent
        self.streams = [torch.cuda.Stream(self.device) if concurrent else torch.cuda.current_stream(self.device) for _ in self.names]
        self.envs = []
        for name, st in zip(self.names, self.streams):
            with torch.cuda.stream(st):
                env = magical_amd.make(name, n_envs=n_envs, device=device, **make_kwargs)
                env.seed(seed + first_env)          # env k of the job draws from RandomState(seed + k) whatever the sharding
                self.envs.append(env)

    def reset(self):
        import torch
        out = []
        for env, st in zip(self.envs, self.streams):
            with torch.cuda.stream(st):
                out.append(env.reset())
        return out

    def step(self, actions, active=None):
        """actions[k]: int32[n_envs] device tensor for engine k (None / inactive: skipped).  Returns the engines' step() results."""
        import torch
        out = [None] * len(self.envs)
        for k, (env, st) in enumerate(zip(self.envs, self.streams)):
            if actions[k] is None or (active is not None and not active[k]):
                continue
            with torch.cuda.stream(st):
                out[k] = env.step(actions[k])
        return out

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def close(self):
        for env in self.envs:
            env.close()

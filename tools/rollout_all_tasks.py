#!/usr/bin/env python
"""one rollout of every task (Demo, or the --variant given / every variant with `all`), `--envs` envs per task sharded over the GPUs of the job,
per-env scores gathered on every rank (RCCL all_gather over xGMI; observations never leave their GPU).

    python tools/rollout_all_tasks.py --envs 8192 [--preproc LoRes4E]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        tools/rollout_all_tasks.py --envs 8192

Prints one JSON line per task on rank 0: env-steps/s of the whole job, mean score, episode length.
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

TASKS = ['MoveToCorner', 'MoveToRegion', 'MatchRegions', 'MakeLine', 'FindDupe', 'FixColour', 'ClusterColour', 'ClusterShape']


class _StubEnv:
    """MGX_BENCH_STUB=1 (tests/test_bench_launcher.py, no GPU): stands in for an engine so that the N-rank control flow of this tool --
    process group (gloo), env shards, per-rank tape slices, seeds by global env index, the end-of-rollout gather, MAX over ranks, one line per
    task on rank 0 -- runs as it is.  An env's score depends on its own tape column and its own seed only: any sharding gives the same table."""

    def __init__(self, name, n):
        import torch
        self.max_episode_steps, self.device, self.n, self.t = 5 + len(name) % 4, torch.device('cpu'), n, 0
        self.acc = np.zeros(n, dtype=np.int64)

    def seed(self, s):
        self.acc += np.arange(s, s + self.n)          # env k of the job is seeded seed + k whatever the sharding

    def reset(self):
        self.t = 0

    def step(self, a):
        self.t += 1
        self.acc += np.asarray(a, dtype=np.int64) * self.t
        done = np.full(self.n, self.t == self.max_episode_steps)
        return None, None, done, {'eval_score': (self.acc % 11) / 11.0}

    def close(self):
        pass


def _stub():
    return bool(os.environ.get('MGX_BENCH_STUB'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=8192, help='envs per task over the whole job')
    ap.add_argument('--preproc', default='LoRes4E')
    ap.add_argument('--variant', default='Demo', help="'Demo', a Test* variant name, or 'all' for every registered variant of every task")
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--warmup-episodes', type=int, default=0, help='untimed whole episodes (with their auto-reset) before the timed one: the timed episode is then a '
                    'steady-state one -- without, a variant\'s first episode also pays the process\'s first use of its kernels, the host pool\'s threads and the growth '
                    'of the pinned staging buffers (per-env-world variants: 65-145 ms in the first episode-end step against 13-18 ms afterwards)')
    ap.add_argument('--concurrent', action='store_true', help='all tasks of the variant at once, one engine + HIP stream per task on every GPU '
                                                              '(magical_amd.distributed.TaskFleet), next to the sequential loop')
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    import magical_amd
    from magical_amd.distributed import env_shard, gather_rollout_results, init_from_env
    from magical_amd.distributed import device_index_for_local_rank
    stub = _stub()
    rank, world, local_rank = init_from_env(backend='gloo' if stub else 'nccl')
    if not stub:
        local_rank = device_index_for_local_rank(local_rank)      # (LOCAL_RANK, or 0 where the launcher shows every rank one device only)
        torch.cuda.set_device(local_rank)
    sync = (lambda: None) if stub else torch.cuda.synchronize
    lo, hi = env_shard(args.envs, rank, world)
    magical_amd.register_envs()
    variants = (sorted({magical_amd.EnvName(n).variant for n in magical_amd.ALL_REGISTERED_ENVS}, key=lambda v: (v != 'Demo', v))
                if args.variant == 'all' else [args.variant])
    names = []
    for variant in variants:
        for task in TASKS:
            name = f'{task}-{variant}-{args.preproc}-v0' if args.preproc else f'{task}-{variant}-v0'
            if name in magical_amd.ALL_REGISTERED_ENVS:
                names.append((task, name))
    if args.concurrent:
        from magical_amd.distributed import TaskFleet
        only = [name for _, name in names]
        res = {}
        for mode in ('sequential', 'concurrent'):
            fleet = TaskFleet(only, hi - lo, f'cuda:{local_rank}', seed=args.seed, first_env=lo, concurrent=(mode == 'concurrent'))
            Ts = [e.max_episode_steps for e in fleet.envs]
            tapes = [torch.as_tensor(np.random.RandomState(args.seed).randint(0, 18, size=(T, args.envs)).astype(np.int32)[:, lo:hi], device=fleet.device) for T in Ts]
            fleet.reset(); fleet.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            scores = [None] * len(only)
            if mode == 'sequential':
                for k, env in enumerate(fleet.envs):
                    for s in range(Ts[k]):
                        _, _, done, info = env.step(tapes[k][s])
                    scores[k] = info['eval_score']
            else:
                for s in range(max(Ts)):
                    out = fleet.step([tapes[k][s] if s < Ts[k] else None for k in range(len(only))])
                    for k, o in enumerate(out):
                        if o is not None and s == Ts[k] - 1:
                            assert o[2].all()
                            scores[k] = o[3]['eval_score']
            fleet.synchronize()
            # ONE end-of-rollout gather for all tasks: [n_tasks, n_local] -> every rank holds [n_tasks, n_total]
            local = torch.as_tensor(np.stack(scores), device=fleet.device).T.contiguous()
            allsc = gather_rollout_results(local, args.envs)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=fleet.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
            res[mode] = (dt, allsc.cpu().numpy())
            fleet.close()
        if rank == 0:
            total = args.envs * sum(Ts)
            assert np.array_equal(res['sequential'][1], res['concurrent'][1]), 'concurrent engines changed the results'
            print(json.dumps({'tasks': only, 'n_envs_per_task': args.envs, 'n_gpus': world, 'env_steps': total,
                              'sequential_env_steps_per_s': total / res['sequential'][0], 'concurrent_env_steps_per_s': total / res['concurrent'][0],
                              'speedup': res['sequential'][0] / res['concurrent'][0], 'identical_scores': True,
                              'mean_score_per_task': [float(v) for v in res['concurrent'][1].mean(axis=0)]}))
        if world > 1:
            dist.destroy_process_group()
        return
    for task, name in names:
        env = _StubEnv(name, hi - lo) if stub else magical_amd.make(name, n_envs=hi - lo, device=f'cuda:{local_rank}')
        T = env.max_episode_steps
        # the action tape of the whole job, every rank takes its slice (same results for any number of GPUs)
        tape = np.random.RandomState(args.seed).randint(0, 18, size=(T, args.envs)).astype(np.int32)[:, lo:hi]
        tape = torch.as_tensor(tape, device=env.device)
        env.seed(args.seed + lo)      # env k of the job draws from RandomState(seed + k) whatever the number of GPUs
        env.reset()
        for _ in range(args.warmup_episodes):
            for s in range(T):
                env.step(tape[s])          # (the last step auto-resets: the timed episode starts from that reset)
        if world > 1:
            dist.barrier()
        sync(); t0 = time.perf_counter()
        for s in range(T):
            _, _, done, info = env.step(tape[s])
        assert done.all()
        scores = gather_rollout_results(torch.as_tensor(info['eval_score'], device=env.device), args.envs)
        sync(); dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=env.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
        if rank == 0:
            print(json.dumps({'task': name, 'n_envs': args.envs, 'n_gpus': world, 'episode_steps': T, 'warmup_episodes': args.warmup_episodes,
                              'env_steps_per_s': args.envs * T / dt, 'mean_score': float(scores.mean().item()),
                              'frac_solved': float((scores > 0.5).double().mean().item()),
                              **({'stub_scores_crc': int(__import__('zlib').crc32(np.ascontiguousarray(scores.numpy()).tobytes())), 'gathered_rows': int(scores.shape[0])} if stub else {})}), flush=True)
        env.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

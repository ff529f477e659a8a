#!/usr/bin/env python
"""bench.py -- BASELINE.json's headline metric on MI355X:

    env-steps/sec (incl. 96x96 LoRes4E render) at N_envs=4096 on MoveToCorner-Demo-LoRes4E-v0

A "step" is one pass of the hot path over one batch: BaseEnv.step() for all 4096 envs of the rank
(set_action + 10 physics substeps x 10 solver iterations, episode bookkeeping with auto-reset and
host scoring at episode ends, ego rasterisation + INTER_AREA + FlattenFrameStack into the
[N,96,96,12] u8 observation tensor) under random actions that are already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N > 1 without WORLD_SIZE set: launches the N ranks itself)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One JSON line on rank 0 (contract in the task statement) + "roofline" for the dominant kernel
(HIP-event timed on the launch stream, inside the timed region) + "cpu_baseline" (the fp64 oracle
port on the host cores, rank 0 at N=1 only, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
TASK = 'MoveToCorner-Demo-LoRes4E-v0'
N_ENVS = 4096


_RESOURCES = None


def kernel_resources():
    """{kernel symbol: {vgpr, agpr, sgpr_spill, vgpr_spill, scratch_bytes_per_lane, lds}} from the notes of the shipped code object
    (llvm-readelf --notes on the gfx950 image inside libmagical_hip.so); {} where the tool is missing."""
    global _RESOURCES
    if _RESOURCES is not None:
        return _RESOURCES
    import re, struct, subprocess, tempfile
    _RESOURCES = {}
    try:
        from magical_amd import _native
        data = open(_native.LIB_PATH, 'rb').read()
        i = data.find(b'__CLANG_OFFLOAD_BUNDLE__')
        n = struct.unpack_from('<Q', data, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from('<QQQ', data, off); off += 24
            triple = data[off:off + tl].decode(); off += tl
            if 'gfx950' in triple:
                with tempfile.NamedTemporaryFile(suffix='.co') as f:
                    f.write(data[i + o:i + o + sz]); f.flush()
                    txt = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', f.name], capture_output=True, text=True, timeout=60).stdout
                for blk in txt.split('- .agpr_count:')[1:]:
                    g = lambda k: int(re.search(r'\.' + k + r':\s+(\d+)', blk).group(1))
                    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
                    _RESOURCES[name] = {'vgpr': g('vgpr_count'), 'agpr': int(blk.split()[0]), 'sgpr_spill': g('sgpr_spill_count'), 'vgpr_spill': g('vgpr_spill_count'),
                                        'scratch_bytes_per_lane': g('private_segment_fixed_size')}
    except Exception:
        pass
    return _RESOURCES


def scratch_bytes(kernel, n_envs, lanes_per_env, dtype='f32', raster_lds=None, layout=1, narrow=True):
    """Static scratch footprint of one launch: private segment bytes per lane x lanes launched (step: n_envs x lanes_per_env;
    raster: 256 per env) of the instantiation this workload runs -- k_step<R,P,L> / k_step_wide (all-fp64) / k_step_env (one env per
    wavefront); k_raster<P,LAYOUT,WAVES> with WAVES from the LDS footprint as the host picks it; None if the code object cannot be read."""
    res = kernel_resources()
    rp = {'f32': 'fd', 'f64': 'dd', 'f32_pure': 'ff'}.get(dtype, 'fd')
    if kernel == 'k_step':
        want = ('k_step_envI%sE' % rp) if lanes_per_env == 64 else (('k_step_wideI%sLi%dE' if dtype == 'f64' else 'k_stepI%sLi%dE') % (rp, lanes_per_env))
    else:
        fit = 5 if not raster_lds else max(3, min(5, (160 * 1024) // ((int(raster_lds) + 1279) // 1280 * 1280)))      # (LDS goes out in 1280-byte pieces: lds_alloc_bytes, csrc/mgx_api.hip)
        want = 'k_rasterI%sLi%dELi%dE%s' % (rp[1], layout, fit, 'm' if narrow is False else 'j')      # (j / m: 32- / 64-bit primitive sets)
    for name, r in res.items():
        if want in name and 'deferred' not in name:
            return r['scratch_bytes_per_lane'] * (n_envs * lanes_per_env if kernel == 'k_step' else n_envs * 256)
    return None


def _cpu_worker(args):
    seed, seconds, use_pymunk = args
    from oracle.env_ref import LoRes4ERef, RefEnv
    env = LoRes4ERef(RefEnv('MoveToCorner'))
    rng = np.random.RandomState(seed)
    env.reset()
    phys = None
    if use_pymunk:
        # the reference's real physics engine where it can be imported (oracle/pymunk_backend.py): pymunk steps the world, the
        # port's painter renders the poses it reached (pyglet / GL cannot be had headless here)
        phys = RefEnv('MoveToCorner', backend='pymunk')
        phys.reset()
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        a = rng.randint(18)
        if phys is not None:
            _, done, _ = phys.step(a)
            b = env.env.bodies(); b[:, :6] = phys.bodies()[:, :6]; env.env.set_bodies(b)
            env.frames.append(env.env.render('ego')); env._obs()
        else:
            _, _, done, _ = env.step(a)
        n += 1
        if done:
            env.reset()
            if phys is not None:
                phys.reset()
    return n, time.perf_counter() - t0


def cpu_baseline(seconds=12.0):
    """The oracle (fp64 restatement of the pymunk + GL + cv2 path; NOT pymunk itself) on the host cores: one
    env per process, random actions, physics + 384x384 ego render + 4-frame stack + INTER_AREA, like
    misc/benchmark_env_perf.py:12-18 drives the reference."""
    import multiprocessing as mp
    # cores this process may actually use (the GPU box exposes 256 logical CPUs; affinity/cgroups can be narrower)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:   # cgroup v2 CPU quota ("max" or "<quota> <period>")
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    cores = min(cores, 64)
    from oracle import pymunk_backend
    use_pymunk, pm_detail = pymunk_backend.probe()
    ctx = mp.get_context('spawn')
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(1000 + k, seconds, use_pymunk) for k in range(cores)])
    steps = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    what = (f'pymunk {pm_detail} physics + the oracle\'s painter' if use_pymunk else 'oracle/ fp64 C port') + ' incl. 384x384 ego render, 4-frame stack, 4x4 box filter'
    return {'value': steps / wall, 'unit': 'env-steps/s', 'cores': cores, 'kind': 'pymunk' if use_pymunk else 'port',
            'sample': f'{steps} env-steps of MoveToCorner-Demo-LoRes4E ({what}), {cores} processes x {seconds:.0f} s, random actions',
            'pymunk_importable': bool(use_pymunk)}


def pose_l2(device, n=32, t_forced=20, t_free=200):
    """The second half of BASELINE.json's metric ("...; pose L2 vs pymunk"), untimed, after the timed region -- like cpu_baseline, the
    oracle is the CHECKER here, never the thing measured.  pymunk cannot be imported in this image, so the comparison is against
    oracle/ (the C restatement of Chipmunk's step, parity UNPINNED) and says so.  32 envs of the headline task
    (state-only engine of the shipped precision, same k_step), one action tape:
      * one-step error, teacher-forced (every env-step starts from the oracle's body state): median / p99;
      * free-running drift at substep 200 (= env-step 20) and at env-step 200 (BASELINE's "200 steps" is ambiguous: both), median / p90;
    each next to the oracle's OWN spread -- a replica of the oracle whose poses start U(-1e-7, 1e-7) off (one fp32 rounding at unit
    scale).  Two norms per env: l2 = Euclidean norm of the pose differences (x, y, angle of every body whose pose is persistent state),
    linf = the largest single difference (what the tests gate)."""
    import magical_amd
    from tests.util import EPS_F32, comparable_mask, new_ref, perturb_bodies, ref_body_index
    task = TASK.split('-')[0]

    def norms(a, b, mask):
        d = np.abs(np.asarray(a) - np.asarray(b))[mask]
        return float(np.sqrt((d * d).sum())), float(d.max())
    q = lambda x, p: float(np.percentile(np.asarray(x), p))
    rs = np.random.RandomState(17)
    # -- teacher-forced one-step error
    env = magical_amd.make(f'{task}-Demo-v0', n_envs=n, device=device, max_episode_steps=10 ** 6)
    env.reset()
    refs, pert = [new_ref(task) for _ in range(n)], [new_ref(task) for _ in range(n)]
    idx, mask = ref_body_index(refs[0]), comparable_mask(refs[0])
    tape = np.random.RandomState(5).randint(0, 18, size=(t_forced, n)).astype(np.int32)
    e_l2, e_li, r_l2, r_li = [], [], [], []
    for s_ in range(t_forced):
        b = env.get_bodies()
        for k, r in enumerate(refs):
            state = r.bodies()
            b[k, 1:, :] = state[idx]
            pert[k].set_bodies(state); perturb_bodies(pert[k], EPS_F32, rs)
        env.set_bodies(b)
        env.step(tape[s_])
        got = env.get_bodies()[:, 1:, :3]
        for k, r in enumerate(refs):
            r.step(tape[s_, k]); pert[k].step(tape[s_, k])
            want = r.bodies()[idx][:, :3]
            a, c = norms(got[k], want, mask); e_l2.append(a); e_li.append(c)
            a, c = norms(pert[k].bodies()[idx][:, :3], want, mask); r_l2.append(a); r_li.append(c)
    env.close()
    one = {'env_steps': t_forced, 'samples': len(e_l2),
           'engine': {'l2_median': q(e_l2, 50), 'l2_p99': q(e_l2, 99), 'linf_median': q(e_li, 50), 'linf_p99': q(e_li, 99)},
           'oracle_replica_1e-7': {'l2_median': q(r_l2, 50), 'l2_p99': q(r_l2, 99), 'linf_median': q(r_li, 50), 'linf_p99': q(r_li, 99)}}
    # -- free-running drift: the shipped fp32 build AND the all-fp64 (reference-precision) build, same 32 envs, same tape, same oracle run
    builds = ('f32', 'f64')
    envs = {d: magical_amd.make(f'{task}-Demo-v0', n_envs=n, device=device, max_episode_steps=10 ** 6, dtype=d) for d in builds}
    for e in envs.values():
        e.reset()
    refs, reps = [new_ref(task) for _ in range(n)], [new_ref(task) for _ in range(n)]
    for r in reps:
        perturb_bodies(r, EPS_F32, rs)
    tape = np.random.RandomState(7).randint(0, 18, size=(t_free, n)).astype(np.int32)
    free = {d: {} for d in builds}
    stats = lambda v: {'l2_median': q([x[0] for x in v], 50), 'l2_p90': q([x[0] for x in v], 90), 'linf_median': q([x[1] for x in v], 50)}
    for s_ in range(t_free):
        for e in envs.values():
            e.step(tape[s_])
        for k in range(n):
            refs[k].step(tape[s_, k]); reps[k].step(tape[s_, k])
        if s_ + 1 in (20, t_free):
            key = 'substep_200' if s_ + 1 == 20 else f'env_step_{t_free}'
            want = [refs[k].bodies()[idx][:, :3] for k in range(n)]
            rn = [norms(reps[k].bodies()[idx][:, :3], want[k], mask) for k in range(n)]
            for d, e in envs.items():
                got = e.get_bodies()[:, 1:, :3]
                free[d][key] = {'engine': stats([norms(got[k], want[k], mask) for k in range(n)]), 'oracle_replica_1e-7': stats(rn)}
    for e in envs.values():
        e.close()
    med = lambda d, key: free[d][key]['engine']['l2_median']
    k200 = f'env_step_{t_free}'
    met = {d: {key: bool(med(d, key) < 1e-3) for key in free[d]} for d in builds}
    if met['f32']['substep_200'] and met['f32'][k200]:
        verdict = 'met by the shipped fp32 build under both readings'
    elif met['f64']['substep_200'] and met['f64'][k200]:
        verdict = 'met by the all-fp64 (reference-precision) build under both readings; not by the shipped fp32 build'
    elif met['f64']['substep_200']:
        verdict = ("BASELINE's target, substep reading (200 substeps = env-step 20), median: met by the all-fp64 (reference-precision) build; "
                   'not by the shipped fp32 build; env-step-200 reading: met by neither')
    else:
        verdict = 'not met by either build under either reading'
    # flat scalars: the driver's record keeps scalars and strings of `config` only (verdict r5, weak 4)
    flat = {'pose_l2_substep200_median_f32': med('f32', 'substep_200'), 'pose_l2_substep200_median_f64': med('f64', 'substep_200'),
            'pose_l2_envstep200_median_f32': med('f32', k200), 'pose_l2_envstep200_median_f64': med('f64', k200),
            'pose_l2_substep200_median_oracle_replica_1e-7': free['f32']['substep_200']['oracle_replica_1e-7']['l2_median'],
            'pose_l2_envstep200_median_oracle_replica_1e-7': free['f32'][k200]['oracle_replica_1e-7']['l2_median'],
            'pose_l2_one_step_median_f32': one['engine']['l2_median'],
            'pose_drift_target_met': bool(met['f32']['substep_200'] and met['f32'][k200]),
            'pose_drift_target_met_f64_substep200': bool(met['f64']['substep_200']),
            'pose_drift_target': verdict + ' (against oracle/, the C restatement of Chipmunk -- pymunk is not importable here; 32 envs, one tape)'}
    return {'against': 'oracle/ C restatement of Chipmunk\'s step (UNPINNED: no pymunk in this image), not pymunk', 'task': f'{task}-Demo-v0', 'n_envs': n, 'dtype': 'f32',
            'norms': 'per env over x, y, angle of every body with persistent pose: l2 = Euclidean norm, linf = largest component (arena = [-1, 1]^2)',
            'one_step_teacher_forced': one, 'free_running': free['f32'], 'free_running_f64': free['f64'],
            'baseline_target_1e-3_met': flat['pose_drift_target_met'], 'baseline_target_1e-3_met_by_build': met, 'flat': flat,
            'note': 'BASELINE.json asks for pose drift < 1e-3 over 200 steps against pymunk.  ' + verdict + '.  The reference pins each finger with a zero-length '
                    'PinJoint (entities.py:334-341) whose direction is normalised round-off, so the oracle parts from its own 1e-7 replica as fast as the fp32 build '
                    'parts from the oracle (the oracle_replica_1e-7 columns).  See DESIGN.md section 5.'}


CONFIG5_TASKS = ['MoveToCorner', 'MoveToRegion', 'MatchRegions', 'MakeLine', 'FindDupe', 'FixColour', 'ClusterColour', 'ClusterShape']


# ---- the N-rank control flow: what every rank does around its own measurement --------------------------------------
def tape_seed(rank, task_index=0):
    """Seed of the action tape A = RandomState(seed).randint(0, 18, (T, N)) of one rank (and, in config 5, one task): every rank steps
    its own env shard under its own random actions (each rank generates only its slice)."""
    return 1000 * task_index + rank


def max_over_ranks(elapsed, device, world):
    """The job's time = the slowest rank's (the contract's MAX over ranks): one all_reduce(MAX) of a double."""
    if world <= 1:
        return float(elapsed)
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def relaunch_under_torchrun(n, argv):
    """`python bench.py --gpus N` as a plain command (N > 1, no WORLD_SIZE in the environment): the same command line again under
    torch.distributed.run, one process per GPU of this node, rendezvous on 127.0.0.1 at a free port (torchrun sets RANK, LOCAL_RANK,
    WORLD_SIZE, LOCAL_WORLD_SIZE -- the host pool's thread cap reads the last).  Rank 0's JSON line goes to this process's stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))      # dmabuf IPC only on this driver
    return subprocess.call(cmd, env=env)


def stub_mode():
    """MGX_BENCH_STUB=1 (tests/test_bench_launcher.py, no GPU): the launcher, the process group (gloo), the per-rank tapes, the gather,
    the MAX over ranks and the one-line emit run as they are; only the engine's work is replaced by a sleep that differs per rank."""
    return bool(os.environ.get('MGX_BENCH_STUB'))


def measure_stub(args, rank, world, device):
    import torch
    import torch.distributed as dist
    from magical_amd.distributed import gather_rollout_results
    n, K, W = args.envs, args.steps, args.warmup
    seed = tape_seed(rank)
    tape = np.random.RandomState(seed).randint(0, 18, size=(W + K, n)).astype(np.int32)
    gather_rollout_results(torch.zeros(n, dtype=torch.float64), n * world)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.002 * K * (rank + 1))                      # "the rollout": the last rank is the slowest
    scores = torch.as_tensor((tape[W:].sum(axis=0) % 7) / 7.0 + rank, dtype=torch.float64)
    own = time.perf_counter() - t0                          # (taken BEFORE the collective, which would even the ranks out: the MAX is what is under test)
    all_scores = gather_rollout_results(scores, n * world)
    if world > 1:
        dist.barrier()
    elapsed = max_over_ranks(own, device, world)
    seeds, owns, pools = [seed], [own], [host_pool_threads()]
    if world > 1:
        got = [None] * world
        dist.all_gather_object(got, (seed, own, zlib_crc(tape), pools[0]))
        seeds, owns, crcs, pools = [g[0] for g in got], [g[1] for g in got], [g[2] for g in got], [g[3] for g in got]
    else:
        crcs = [zlib_crc(tape)]
    if rank != 0:
        return None
    return {'metric': f'env-steps/sec (STUB: no engine, MGX_BENCH_STUB=1) at N_envs={n}', 'value': n * world * K / elapsed, 'unit': 'env-steps/s',
            'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic', 'config': {'workload': f'stub of {args.task}, {n} envs per rank', 'n_envs_per_gpu': n, 'episodes_finished': 0,
                                                                   'mean_eval_score': float(all_scores.mean().item())},
            'collective': {'backend': dist.get_backend() if dist.is_initialized() else None, 'world_size': world, 'gathered_rows': int(all_scores.shape[0])},
            'roofline': None, 'stub': {'tape_seeds': seeds, 'tape_crcs': crcs, 'elapsed_per_rank': owns, 'elapsed_max': elapsed, 'host_pool_threads': pools,
                                       'local_world_size': int(os.environ.get('LOCAL_WORLD_SIZE', '1')), 'cores': os.cpu_count()}}


def host_pool_threads():
    """Threads the native library's host pool (world builds and placement at a reset) takes in THIS process: the node's cores divided by
    the ranks on it (LOCAL_WORLD_SIZE), capped at 64 (csrc/mgx_api.hip host_threads).  The library loads without a GPU."""
    from magical_amd import _native
    return int(_native.lib().mgx_debug_host_threads(1))


def zlib_crc(a):
    import zlib
    return int(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def run_config5(envs5, K, W, rank=0, world=1, device='cuda:0', dtype='f32', concurrent=True):
    """The body of the config-5 line for one rank (also what tests/test_gpu_parity.py runs at rank size, 8 x 1024 envs): the 8 Demo
    tasks as 8 engines on 8 HIP streams over this rank's env shard, W untimed + K timed env-steps of every task with auto-reset at each
    task's own episode length, the last finished episode's score of every env kept, ONE gather of the [envs, 8] score table at the
    end (inside the timed region).  Returns (scores of the whole job [envs5, 8], episodes finished on this rank, seconds)."""
    import torch
    import torch.distributed as dist
    from magical_amd.distributed import TaskFleet, env_shard, gather_rollout_results
    lo, hi = env_shard(envs5, rank, world)
    n = hi - lo
    names = [f'{t}-Demo-LoRes4E-v0' for t in CONFIG5_TASKS]
    fleet = TaskFleet(names, n, device, seed=0, first_env=lo, dtype=dtype, concurrent=concurrent)
    nt = len(names)
    tapes = [torch.as_tensor(np.random.RandomState(tape_seed(rank, k)).randint(0, 18, size=(W + K, n)).astype(np.int32), device=device) for k in range(nt)]
    fleet.reset()
    for s in range(W):
        fleet.step([tp[s] for tp in tapes])
    last = np.zeros((nt, n), dtype=np.float64)
    n_eps = 0
    gather_rollout_results(torch.zeros((n, nt), dtype=torch.float64, device=device), envs5)      # RCCL warm-up

    def barrier():
        fleet.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for s in range(W, W + K):
        for k, o in enumerate(fleet.step([tp[s] for tp in tapes])):
            done = o[2]
            if done.any():
                n_eps += int(done.sum())
                last[k, done] = o[3]['eval_score'][done]
    fleet.synchronize()
    all_scores = gather_rollout_results(torch.as_tensor(last.T.copy(), device=device), envs5)      # [envs5, n_tasks] on every rank
    barrier()
    elapsed = time.perf_counter() - t0
    fleet.close()
    return all_scores, n_eps, elapsed


def run_config5_stub(envs5, K, W, rank=0, world=1, device='cpu', dtype='f32'):
    """run_config5 without engines (stub_mode): the per-task tapes, the [envs, 8] score table and its ONE gather are real."""
    import torch
    import torch.distributed as dist
    from magical_amd.distributed import env_shard, gather_rollout_results
    lo, hi = env_shard(envs5, rank, world)
    n, nt = hi - lo, len(CONFIG5_TASKS)
    tapes = [np.random.RandomState(tape_seed(rank, k)).randint(0, 18, size=(W + K, n)) for k in range(nt)]
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.002 * K * (rank + 1))
    last = np.stack([(tp[W:].sum(axis=0) % 5) / 5.0 for tp in tapes])          # [tasks, n]
    all_scores = gather_rollout_results(torch.as_tensor(last.T.copy()), envs5)
    if world > 1:
        dist.barrier()
    return all_scores, 0, time.perf_counter() - t0


def main_config5(args):
    """BASELINE.json configs[4] / SURVEY.md section 8d config 5: the 8 Demo tasks at once, `--envs5` envs per task sharded by env index
    over the job's GPUs (8192 / 8 = 1024 per task per GPU), every rank stepping its 8 engines concurrently on 8 HIP streams
    (magical_amd.distributed.TaskFleet), auto-reset at each task's own episode length, and ONE all_gather of the per-env
    scores of all tasks at the end of the rollout.  A "step" = one env-step of every task; value = env-steps/s of the job."""
    import torch
    import torch.distributed as dist
    from magical_amd.distributed import env_shard, init_from_env
    stub = stub_mode()
    rank, world, local_rank = init_from_env(backend='gloo' if stub else 'nccl', single_process_group=not args.no_collective)      # (one GPU: a group of one rank, the gather still runs)
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}')
    if stub:
        device = 'cpu'
    else:
        from magical_amd.distributed import device_index_for_local_rank
        dev = device_index_for_local_rank(local_rank)
        torch.cuda.set_device(dev)
        device = f'cuda:{dev}'
    lo, hi = env_shard(args.envs5, rank, world)
    n, K, W, nt = hi - lo, args.steps, args.warmup, len(CONFIG5_TASKS)
    all_scores, n_eps, own = (run_config5_stub if stub else run_config5)(args.envs5, K, W, rank, world, device, args.dtype)
    elapsed = max_over_ranks(own, device, world)
    if rank == 0:
        out = {'metric': f'env-steps/sec (incl. 96x96 LoRes4E render), all 8 tasks x Demo at once, {args.envs5} envs per task', 'value': nt * args.envs5 * K / elapsed,
               'unit': 'env-steps/s', 'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True,
               'scaling': 'strong', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
               'config': {'workload': f'BASELINE.json configs[4]: {", ".join(CONFIG5_TASKS)} x Demo-LoRes4E-v0, {args.envs5} envs per task sharded '
                                      f'{world} ways ({n} per task per GPU), 8 engines per GPU on 8 HIP streams, random actions, auto-reset, one RCCL '
                                      'all_gather of all scores at the end',
                          'envs_per_task_per_gpu': n, 'episodes_finished': n_eps * world, 'mean_eval_score': float(all_scores.mean().item())},
               'collective': {'backend': dist.get_backend() if dist.is_initialized() else None, 'world_size': world,
                              'in_timed_region': f'one all_gather of the score table f64[{n}, {nt}] per rank at the end of the rollout',
                              'gathered_rows': int(all_scores.shape[0])},
               'roofline': None, 'note': 'a step = one env-step of each of the 8 tasks; kernels of different engines overlap, so per-kernel '
                                         'roofline figures are those of the single-task lines (python bench.py --task ...)'}
        if stub:
            out['stub'] = {'elapsed_max': elapsed, 'tape_seeds_rank0': [tape_seed(rank, k) for k in range(nt)], 'host_pool_threads_rank0': host_pool_threads(),
                           'local_world_size': int(os.environ.get('LOCAL_WORLD_SIZE', '1')), 'cores': os.cpu_count()}
        args.emit(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


def window_plan(K, W, ep, n):
    """(untimed preroll steps, envs whose episode ends inside the K-step window, their clock offset) -- see main().
    K >= ep, or a warm-up that leaves no room for K steps inside one episode: (0, 0, 0), the window holds whole episodes as it is."""
    if K >= ep or W + K >= ep:
        return 0, 0, 0
    start_phase = max(W, (ep - K - 1) // 2)         # episode clock of the untouched envs at the start of the window
    preroll = ep + start_phase - W                  # one whole untimed episode first (clocks, allocator, pinned buffers warm)
    share = int(round(n * K / ep))
    ahead = ep - K // 2 - 1 - start_phase           # clock offset of the envs that finish inside the window
    assert 0 <= ahead and ahead + start_phase < ep and start_phase + K < ep
    return preroll, share, ahead


def measure(args, rank, world, device):
    """One bench line: W untimed + K timed env-steps of args.task at args.envs envs per GPU.  Returns the line's dict on rank 0."""
    import torch
    import torch.distributed as dist
    from magical_amd.distributed import gather_rollout_results
    import magical_amd
    env = magical_amd.make(args.task, n_envs=args.envs, device=device, lanes_per_env=args.lanes, dtype=args.dtype, obs_ring=args.obs_ring)
    ring = bool(args.obs_ring) and '-LoResCHW4E-' in args.task
    n, K, W = args.envs, args.steps, args.warmup
    # The metric includes auto-reset + scoring at episode ends: one per env per `ep` env-steps in the long run.  A
    # timed region shorter than an episode (the driver's 20 steps vs 80) would see none, or -- with every env's end rolled into
    # it -- ep / K times its share.  So a K-step window is given exactly its long-run share: the first n * K / ep envs have
    # their episode clocks set ahead (set_episode_steps) so that their episodes end in the middle of the window -- scoring,
    # auto-reset and frame-stack refill of that many envs are inside the timed region -- and the other envs are mid-episode (the
    # window covers the middle K steps of their episode; the untimed preroll takes them there).  K >= ep: the window
    # contains K / ep whole episodes of every env as it is, nothing is set ahead.
    ep = env.max_episode_steps
    preroll, share, ahead = window_plan(K, W, ep, n)
    # synthetic input: A = RandomState(seed).randint(0, 18, (T, N)) uploaded once; each rank its own slice
    tape = torch.as_tensor(np.random.RandomState(tape_seed(rank)).randint(0, 18, size=(preroll + W + K, n)).astype(np.int32), device=device)
    obs = env.reset()
    last_score = torch.zeros(n, dtype=torch.float64, device=device)      # per-env result of the rollout
    for s in range(preroll):
        if share and s in (0, ep):                      # s == ep: every env has just started its second episode
            # (s == 0: the same number of envs also finishes early in the untimed first episode, so that the partial-mask path's
            # first-use costs -- torch's lazily loaded kernels, allocator blocks -- are not inside the timed window)
            clocks = np.zeros(n, dtype=np.int64)
            clocks[:share] = ahead if s == ep else ep // 2
            env.set_episode_steps(clocks)
        obs, rew, done, info = env.step(tape[s])
    tape = tape[preroll:]
    # the collective's first use (RCCL creates its communicator, proxy thread and buffers there) comes BEFORE the warm-up steps: straight
    # before the timed region it left the first steps of the window 0.3 ms slower (stamps: MGX_BENCH_STAMPS=1)
    gather_rollout_results(last_score, n * world)
    torch.cuda.synchronize()
    for s in range(W):
        obs, rew, done, info = env.step(tape[s])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    every = int(os.environ.get('MGX_BENCH_TIMING_EVERY', '0')) or (8 if K >= 160 else 4)
    env.set_timing(every)       # HIP events around every 8th (short windows: every 4th) launch of each kernel inside the timed region
    last_score.zero_()
    # scores are collected on the host -- in PINNED memory, so that the one upload before the gather is asynchronous: the host, which
    # runs a few steps ahead of the GPU, goes straight on into the collective call (c10d's ≈ 0.3 ms of host work) while the GPU is
    # still finishing the rollout; a pageable copy would wait for the GPU first and expose that call
    score_pin = torch.zeros(n, dtype=torch.float64).pin_memory()
    score_host = score_pin.numpy()
    n_eps = 0
    gather_rollout_results(last_score, n * world)      # warm-up of the collective (RCCL sets its channels up on first use)
    last_score.copy_(score_pin, non_blocking=True)                                                     # ... and of the score upload
    barrier()
    t0 = time.perf_counter()
    for s in range(W, W + K):
        obs, rew, done, info = env.step(tape[s])
        if done.any():
            n_eps += int(done.sum())
            score_host[done] = info['eval_score'][done]
    t_loop = time.perf_counter()
    last_score.copy_(score_pin, non_blocking=True)
    # end-of-rollout gather over xGMI (RCCL): per-env scores of every rank; observations never leave their GPU
    ev_g0, ev_g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)      # (device-side clock: no host sync before the collective)
    ev_g0.record()
    all_scores = gather_rollout_results(last_score, n * world)
    ev_g1.record()
    t_call = time.perf_counter()
    barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get('MGX_BENCH_STAMPS'):      # development: where the host was when (stderr)
        print('stamps: host step loop %.2f ms, upload + collective calls %.3f ms, final synchronize %.2f ms' % (
            (t_loop - t0) * 1e3, (t_call - t_loop) * 1e3, (t0 + elapsed - t_call) * 1e3), file=sys.stderr)
    gather_ms = ev_g0.elapsed_time(ev_g1)
    elapsed = max_over_ranks(elapsed, device, world)
    step_ms = env.read_timing('step')
    rast_ms = env.read_timing('render')
    # In the fused step the raster kernel's launch duration includes its hand-off waits for the step kernel (that overlap is the
    # point); the same kernels launched one after the other, in a short UNTIMED phase, give the duration of the kernel's own work
    alone_ms = None
    if getattr(env, 'overlap', False) and len(rast_ms):
        env.overlap = False
        env.set_timing(1)
        for s in range(W, min(W + 40, W + K)):
            env.step(tape[s])
        torch.cuda.synchronize()
        alone_ms = {'k_step': float(env.read_timing('step').mean()), 'k_raster': float(env.read_timing('render').mean())}
        env.overlap = True
    env.set_timing(0)

    out = None
    if rank == 0:
        value = n * world * K / elapsed
        # algorithmic HBM bytes per launch (DESIGN.md "Kernels"): persistent state read once + written once;
        # stacked observation: read the 9 surviving channels, write all 12, per pixel
        rows_p, rows_f = env.state_p.shape[0], env.state_f.shape[0]
        slots = env._info('cache_slots')
        mean_cache = float(env.state_i[1].float().mean().item())
        state_bytes = rows_p * env.state_p.element_size() + (rows_f - 4 * slots) * env.state_f.element_size() + 12 + 20 * mean_cache
        step_bytes = n * (2 * state_bytes + 4 + 1)
        rast_bytes = n * (96 * 96 * (9 + 12) + rows_p * env.state_p.element_size())
        if ring:      # one new planar frame per env and step; the wrap copy (3 frames read + written every R-3 steps) is torch's, not the kernel's
            rast_bytes = n * (96 * 96 * 3 + rows_p * env.state_p.element_size())
        renders = len(rast_ms) > 0        # a task name without a preprocessor is the state-only configuration
        kernels = {'k_step': (float(step_ms.mean()), step_bytes)}
        if renders:
            kernels['k_raster'] = (float(rast_ms.mean()), rast_bytes)
        dom = max(kernels, key=lambda k: kernels[k][0])
        ach = kernels[dom][1] / (kernels[dom][0] * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters: rocprofv3 cannot run inside this process, so the figure is the
        # committed summary of the separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over this same command
        # (profiles/, produced by tools/pmc_summary.py; FETCH_SIZE doubled per MI355X_MICROARCH.md); default workload only
        from tools.csrc_hash import csrc_sha16
        built_from = csrc_sha16(ROOT)            # (magical_amd._native rebuilds the library whenever a source is newer: these ARE its sources)
        # (counter passes are kept for the headline workload, ClusterColour, and -- round 6 -- the two secondary workloads of the driver's line:
        # the all-fp64 build and the state-only engine)
        key = {(TASK, 'f32'): 'mtc_lores4e', ('ClusterColour-Demo-LoRes4E-v0', 'f32'): 'cc_lores4e', (TASK, 'f64'): 'mtc_lores4e_f64',
               ('MoveToCorner-Demo-v0', 'f32'): 'mtc_state_only'}.get((args.task, args.dtype))

        def committed_profile(kind):
            """(dict, source note) of the newest profiles/rNN_pmc_<kind>_<workload>.json whose stamp equals the hash of the kernel sources in
            use; (None, why not) otherwise -- a counter profile of other sources says nothing about the kernels this run timed."""
            if not (key and n == N_ENVS):
                return None, 'no counter passes are kept for this workload'
            why = f'no profiles/rNN_pmc_{kind}_{key}.json committed'
            for rnd in ('r06', 'r05', 'r04', 'r03', 'r02', 'r01'):
                path = os.path.join(ROOT, 'profiles', f'{rnd}_pmc_{kind}_{key}.json')
                try:
                    d = json.load(open(path))
                except Exception:
                    continue
                stamp = (d.get('_stamp') or {}).get('csrc_sha16')
                if stamp == built_from:
                    return d, f'profiles/{rnd}_pmc_{kind}_{key}.json, stamped with the kernel sources this library was built from (csrc {built_from}); collected by rocprofv3 outside this process'
                why = (f'refused: profiles/{rnd}_pmc_{kind}_{key}.json was measured on other kernel sources (its stamp: {stamp or "none"}, this build: csrc {built_from}); '
                       're-run tools/regen_profiles.sh')
                break
            return None, why
        traffic, traffic_src = None, None
        pmc, traffic_src = committed_profile('traffic')
        if pmc is not None:
            try:
                traffic = float(pmc[dom]['hbm_traffic_bytes_per_launch'])
                traffic_src += ' (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE in separate passes, median per launch)'
            except Exception as ex:
                traffic_src = f'unreadable: {type(ex).__name__}: {ex}'
        # what bounds the kernel, from the SQ counter passes of the same command (profiles/rNN_pmc_alu_*.json, tools/pmc_alu_summary.py;
        # like `traffic`, collected by rocprofv3 outside this process and committed): VALU issue utilisation against the dense vector
        # peak, resident wavefronts per SIMD, share of wave-cycles parked on s_waitcnt / stalled at issue, scratch footprint
        alu, alu_src = committed_profile('alu')
        if alu is not None:
            alu_src += ' (three passes of eight SQ counters, kernels one after the other)'
        def scratch_of(kname):
            try:
                from magical_amd import _native
                lds_r = _native.lib().mgx_engine_lds_bytes(env._engine, 1)
            except Exception:
                lds_r = None
            return scratch_bytes(kname, n, env.lanes_per_env, args.dtype, lds_r, 1 if '-LoRes4' in args.task else 0)
        def alu_fields(kname):
            d = ((alu or {}).get(kname) or {}).get('derived') or {}
            return {k: d.get(k) for k in ('valu_util', 'valu_busy', 'occupancy_waves_per_simd', 'wait_share', 'issue_stall_share', 'active_share',
                                          'lds_conflict_share', 'valu_insts_per_wave')}
        def bound_of(kname):
            # HBM-bound only if the counters' traffic moves at more than half the achievable copy rate; else what the SQ counters show
            f = alu_fields(kname)
            if traffic and kname == dom and traffic / (kernels[dom][0] * 1e-3) / 1e9 > 0.5 * 6300:
                return 'hbm'
            if f.get('valu_util') is None:
                return 'hbm (nominal: no SQ counter pass committed for this workload)'
            if (f.get('valu_busy') or 0) > 0.6:
                return 'valu (the SIMDs have a vector instruction in flight in most busy cycles; HBM far from its peak)'
            return 'latency (one dependent instruction stream per wavefront, waiting on LDS for much of it; VALU and HBM both far from their peaks)'
        # SURVEY.md §8(d) has two byte rows for the LoRes4E env-step: the headline one (state + ONE new 96x96x3 frame,
        # 28.3 KB: what a ring of frames would move) and the parenthetical one this layout really needs (the contiguous
        # [96,96,12] stack re-materialised: 9 B read + 12 B written per pixel, 194 KB).  `frac` prices the kernel against the
        # layout it implements; `frac_new_frame_row` is the same launch priced against the headline row.
        ring_bytes = n * (96 * 96 * 3 + rows_p * env.state_p.element_size())
        out = {
            'metric': f'env-steps/sec (incl. 96x96 LoRes4E render) at N_envs={n}' if '-LoRes4E-' in args.task else
                      f'env-steps/sec ({"incl. 96x96 render" if renders else "state-only observation"}) at N_envs={n}', 'value': value, 'unit': 'env-steps/s',
            'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': elapsed / K * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.dtype == 'f32' else args.dtype, 'data': 'synthetic',
            'config': {'workload': f'{args.task}, {n} envs per GPU, random actions, auto-reset every {env.max_episode_steps} steps, '
                                   + (f'obs u8[N,12,96,96] = window of a ring u8[N,{args.obs_ring},3,96,96] of planar frames' if ring else
                                      'obs u8[N,96,96,12] (4 ego frames, oldest first)' if '-LoRes4E-' in args.task else
                                      'rendered observation' if renders else 'obs f32[N,n_bodies,3] poses'),
                       'n_envs_per_gpu': n, 'lanes_per_env': env.lanes_per_env, 'episodes_finished': n_eps * world,
                       'mean_eval_score': float(all_scores.mean().item()),
                       'untimed_preroll_steps': preroll,
                       'episode_ends_in_window': (f'{share} of {n} envs per GPU finish an episode inside the {K}-step window = its long-run share K / {ep} '
                                                  '(their episode clocks were set ahead; scoring + auto-reset + stack refill are timed)') if share else
                                                 f'{K // ep} whole episodes of every env',
                       'arith': 'fp32 velocities/impulses/contacts + fp64 poses; fp64 rasteriser' if args.dtype == 'f32' else args.dtype,
                       'broadphase': 'pre-filtered candidate-pair list, AABB-tested brute force by the env\'s lane group and compacted in pair order '
                                     'through a wavefront ballot.  north_star\'s LDS sort-and-sweep was built (-DMGX_BROAD_SAP=1, same list entry for entry) and measured on the '
                                     'same box: k_step 10-15 % slower in every world (ph_broad 105 k -> 212 k cycles per ClusterColour env-step: '
                                     'profiles/r06_step_broadphase_sap_ab.txt), so the list ships',
                       'roofline_bytes_row': 'SURVEY.md 8(d) headline row: state + ONE new 96x96x3 frame per env-step (ring of planar frames)' if ring else
                                             'SURVEY.md 8(d) parenthetical row: [96,96,12] stack re-materialised each step (9 B read + 12 B '
                                             'written per pixel + pose rows); frac_new_frame_row uses the 28.3 KB headline row'},
            'collective': {'backend': (dist.get_backend() + (' (RCCL)' if dist.get_backend() == 'nccl' else '')) if dist.is_initialized() else None,
                           'world_size': world, 'in_timed_region': f'one all_gather of the per-env scores f64[{n}] per rank at the end of the rollout',
                           'ms_on_stream': gather_ms} if dist.is_initialized() else {'backend': None, 'note': 'no process group: single process, gather skipped'},
            'roofline': {'bound': bound_of(dom), 'kernel': dom, 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': ach / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_src,
                         **alu_fields(dom), 'alu_source': alu_src, 'scratch_bytes': scratch_of(dom),
                         'frac_new_frame_row': (ring_bytes / (kernels[dom][0] * 1e-3) / 1e9 / HBM_PEAK_GBS) if dom == 'k_raster' else None,
                         'launch_mode': ('fused: k_raster runs concurrently with k_step and consumes envs as they finish, so its launch duration '
                                         'includes hand-off waits' if alone_ms else 'one kernel after the other'),
                         'kernel_alone': None if not alone_ms else {
                             'note': 'the same kernels launched one after the other (40 untimed steps after the timed region)',
                             'avg_launch_ms': alone_ms, 'frac': kernels[dom][1] / (alone_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         'avg_launch_ms': kernels[dom][0], 'algorithmic_bytes_per_launch': kernels[dom][1],
                         'other_kernels': {k: {'avg_launch_ms': v[0], 'algorithmic_bytes_per_launch': v[1],
                                               'achieved_GBs': v[1] / (v[0] * 1e-3) / 1e9, 'bound': bound_of(k), **alu_fields(k),
                                               'scratch_bytes': scratch_of(k)} for k, v in kernels.items() if k != dom}},
        }
    env.close()
    return out if rank == 0 else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=400)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--task', default=TASK)
    ap.add_argument('--envs', type=int, default=N_ENVS)
    ap.add_argument('--lanes', type=int, default=0)
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-pose-l2', action='store_true', help='default line only: skip the untimed pose-error leg (engine vs oracle, config.pose_l2)')
    ap.add_argument('--no-secondary', action='store_true', help='default line only: skip the short secondary lines (all-fp64 build, ClusterColour, state-only, config 5 at rank size)')
    ap.add_argument('--obs-ring', type=int, default=0, help='with a -LoResCHW4E- task: frames kept as planes in a ring of this many frames '
                    '(MGX_OBS_PLANAR; the channels-first stack is a window of the ring), priced on the 28.3 KB row of SURVEY.md 8(d)')
    ap.add_argument('--config5', action='store_true', help='BASELINE.json configs[4]: all 8 tasks x Demo-LoRes4E, --envs5 envs per task sharded over '
                                                           'the GPUs, one engine + HIP stream per task on every GPU, one RCCL gather at the end')
    ap.add_argument('--envs5', type=int, default=8192, help='envs per task over the whole job (config 5)')
    ap.add_argument('--no-collective', action='store_true', help='--gpus 1: do not form the one-rank RCCL group (the end-of-rollout gather is then skipped)')
    ap.add_argument('--force-collective', action='store_true', help='--gpus 1: fail instead of carrying on without the gather if the one-rank RCCL group cannot be formed')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # a plain `python bench.py --gpus N`: become the launcher of N ranks (the driver may also start the ranks itself, see the docstring)
        sys.exit(relaunch_under_torchrun(args.gpus, sys.argv[1:]))
    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL's version banner sits in a C stdio buffer
    # until exit): file descriptor 1 is pointed at stderr for the run and the line goes out on a duplicate of the real stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        os.write(real_stdout, (line + '\n').encode())
    args.emit = emit
    if args.config5:
        return main_config5(args)

    import torch
    import torch.distributed as dist
    from magical_amd.distributed import gather_rollout_results, init_from_env
    # "nccl" == RCCL on ROCm.  One GPU: a process group of ONE rank, so that the line's timed region holds the same RCCL all_gather
    # as the N-GPU lines (the collective runs on the one GPU; without a group the gather would be skipped)
    collective_error = None
    stub = stub_mode()
    try:
        rank, world, local_rank = init_from_env(backend='gloo' if stub else 'nccl', single_process_group=not args.no_collective)
    except Exception as ex:
        if args.force_collective or int(os.environ.get('WORLD_SIZE', '1')) > 1:
            raise
        collective_error = f'{type(ex).__name__}: {ex}'
        rank, world, local_rank = 0, 1, 0
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (start N ranks, or run `python bench.py --gpus N` without WORLD_SIZE set: it launches them)')
    if stub:
        device = 'cpu'
        out = measure_stub(args, rank, world, device)
        if rank == 0:
            args.emit(json.dumps(out))
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    from magical_amd.distributed import device_index_for_local_rank
    dev = device_index_for_local_rank(local_rank)       # (LOCAL_RANK, or 0 where the launcher shows every rank one device only)
    torch.cuda.set_device(dev)
    device = f'cuda:{dev}'

    out = measure(args, rank, world, device)
    if rank == 0 and collective_error:
        out['collective'] = {'backend': None, 'error': collective_error}
    if rank == 0:
        # Driver-executed, outside the headline's timed region: the reference-precision build and BASELINE.json configs[3]
        # (ClusterColour, many contacts) on short windows with the same window plan -- so that these lines are not builder-run only
        default_line = (world == 1 and args.task == TASK and args.dtype == 'f32' and args.envs == N_ENVS and not args.obs_ring and not args.lanes)
        if default_line and not args.no_secondary:
            import copy
            sec = {}
            for key, over in (('f64', {'dtype': 'f64'}), ('clustercolour', {'task': 'ClusterColour-Demo-LoRes4E-v0'}),
                              ('state_only', {'task': 'MoveToCorner-Demo-v0'})):       # BASELINE.json configs[2] (reference precision), [3], [1]
                a2 = copy.copy(args)
                a2.steps, a2.warmup = 40, 5
                for k, v in over.items():
                    setattr(a2, k, v)
                try:
                    o2 = measure(a2, rank, world, device)
                    sec[f'{key}_env_steps_per_s'] = o2['value']
                    sec[key] = {'task': a2.task, 'dtype': a2.dtype, 'steps': a2.steps, 'warmup': a2.warmup, 'ms_per_step': o2['ms_per_step'],
                                'episodes_finished': o2['config']['episodes_finished'], 'roofline': o2['roofline']}
                except Exception as ex:          # the headline must not be lost to a secondary line
                    sec[key] = {'error': f'{type(ex).__name__}: {ex}'}
            # BASELINE.json configs[4] at rank size: the 8 Demo tasks x 1024 envs as 8 engines on 8 HIP streams of this one GPU (what each
            # of the 8 ranks of the 8192-env job runs), one short window, the score table through the same gather
            try:
                sc5, eps5, el5 = run_config5(1024, 20, 5, 0, 1, device, 'f32')
                sec['config5_rank_size_env_steps_per_s'] = len(CONFIG5_TASKS) * 1024 * 20 / el5
                sec['config5_rank_size'] = {'tasks': CONFIG5_TASKS, 'envs_per_task': 1024, 'steps': 20, 'warmup': 5, 'ms_per_step': el5 / 20 * 1e3,
                                            'episodes_finished': eps5, 'mean_eval_score': float(sc5.mean().item()),
                                            'note': 'a step = one env-step of each of the 8 tasks (8192 env-steps); one rank\'s share of configs[4]'}
            except Exception as ex:
                sec['config5_rank_size'] = {'error': f'{type(ex).__name__}: {ex}'}
            out['secondary'] = sec
        if default_line and not args.no_secondary and not args.no_pose_l2:       # (the driver's line; --no-secondary runs are development A/Bs)
            try:
                pl = pose_l2(device)
                out['config'].update(pl.pop('flat'))         # flat keys first: what the driver's record keeps
                out['config']['pose_l2'] = pl
            except Exception as ex:
                out['config']['pose_l2'] = {'error': f'{type(ex).__name__}: {ex}'}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        args.emit(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""Which of the oracle's recollected choices could move a result at all?  (CPU only, no GPU, no pymunk.)

oracle/magical_ref.c restates Chipmunk / GL / cv2 behaviour that cannot be checked in this image (its header: PARITY UNPINNED).  This
tool flips each such choice (ref_set_unknowns, oracle/magical_ref.c "unknowns"), one at a time, and measures on the ORACLE itself:

  A. eval_score over --episodes random-action episodes of every Demo task (actions RandomState(episode).randint(18), the same tapes for
     every configuration), against the baseline AND against a control whose only change is a 1e-9 perturbation of the initial poses:
     the reference dynamics are chaotic, so any change re-rolls the episodes; what counts is whether the score
     DISTRIBUTION moves more than it does under the control (paired differences: share of episodes whose score changed, mean change,
     and its z against zero; mean final block displacement as a second, continuous statistic).
  B. one env-step from the same state: max |pose difference| between the flipped and the unflipped oracle over states sampled from
     baseline rollouts (what the flip does before chaos takes over; 0 = the flip never acted).
  C. the render flips on frames of those states: bytes of the 96x96x3 ego observation that differ, and by how much.

  A2. (round 6, verdict r5 item 6) the same pairing on episodes that SCORE.  Random actions score 0 in six tasks of eight, so "no flip moves
     the score distribution" was true of a distribution that is identically zero.  Here every episode starts from the task's reset with its
     blocks moved (set_bodies, blocks only, none overlapping, the robot where the reset put it) to the neighbourhood of the task's own
     scoring threshold -- at distance 0.8 ... 1.5 of MoveToCorner's corner, astride the edges of the goal regions, in a ragged line, in
     loose clusters -- and then runs 40 env-steps of a scripted pusher on the oracle's state (drive at block k, gripper opening and closing:
     the tests' _chase_action; MoveToRegion: drive at a point on the goal's edge), which shoves the blocks across those thresholds or not.
     Baseline episodes with eval_score > 0: >= 20 % in every task (printed).  Same flips, same pairing, same control.

    python tools/oracle_unknowns.py --episodes 4096 --procs 8 > profiles/r06_oracle_unknowns_sensitivity.txt
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

TASKS = ['MoveToCorner', 'MoveToRegion', 'MatchRegions', 'MakeLine', 'FindDupe', 'FixColour', 'ClusterColour', 'ClusterShape']
PHYS = [('baseline', 0), ('control: initial poses +-1e-9', -1), ('arbiter order descending', 1), ('collision_persistence 1', 2), ('collision_persistence 5', 4),
        ('contact impulses matched by index', 8), ('contact impulses never carried over', 16), ('polygon bevel radius 1e-3', 32), ('narrowphase started cold', 256)]
RENDER = [('fill rule: samples ON an edge are outside', 64), ('INTER_AREA rounds half up', 128)]


def _episodes(job):
    task, flags, lo, hi = job
    from oracle._lib import lib
    from oracle.entities_ref import Shape as RefShape
    from oracle.env_ref import RefEnv
    from tests.util import perturb_bodies
    L = lib()
    L.ref_set_unknowns(max(flags, 0))
    env = RefEnv(task)
    out = np.zeros((hi - lo, 2))
    for ep in range(lo, hi):
        env.reset()
        rs = np.random.RandomState(ep)
        if flags < 0:
            perturb_bodies(env, 1e-9, np.random.RandomState(10 ** 6 + ep))
        blocks = [e.bodies[0] for e in env.world.entities if isinstance(e, RefShape)]
        start = env.bodies()[blocks, :2].copy() if blocks else np.zeros((0, 2))
        acts = rs.randint(0, 18, size=env.max_episode_steps)
        for a in acts:
            _, done, info = env.step(int(a))
        assert done
        disp = float(np.linalg.norm(env.bodies()[blocks, :2] - start, axis=1).mean()) if blocks else 0.0
        out[ep - lo] = (info['eval_score'], disp)
    L.ref_set_unknowns(0)
    return task, flags, lo, out


SCORING_STEPS = 40


def _place_blocks(env, rs, wanted, ents, keep=()):
    """Move the block entities `ents` to the positions `wanted(i, rs)` proposes (re-drawn until no two blocks, and no block and the robot /
    a kept entity, are closer than 0.34: a block's circumradius is 0.12-0.15), angles uniform, velocities zero.  Blocks only: single bodies."""
    b = env.bodies()
    rb = env.task.robot.bodies
    fixed = [b[k, :2].copy() for k in rb] + [b[e.bodies[0], :2].copy() for e in keep]
    placed = []
    for i, e in enumerate(ents):
        for attempt in range(200):
            x, y = wanted(i, rs)
            x, y = float(np.clip(x, -0.82, 0.82)), float(np.clip(y, -0.82, 0.82))
            if all(np.hypot(x - px, y - py) > 0.34 for px, py in placed) and all(np.hypot(x - fx, y - fy) > 0.42 for fx, fy in fixed):
                break
        placed.append((x, y))
        k = e.bodies[0]
        b[k, :] = 0.0
        b[k, 0], b[k, 1], b[k, 2] = x, y, rs.uniform(-np.pi, np.pi)
    env.set_bodies(b)


def _edge_point(bb, rs, spread=0.16):
    """a point astride the edge of the box (l, b, r, t): on a random side, U(-spread, spread) across it"""
    l, bt, r, t = bb
    side = rs.randint(4)
    u, d = rs.uniform(0.15, 0.85), rs.uniform(-spread, spread)
    if side == 0: return l + d, bt + u * (t - bt)
    if side == 1: return r + d, bt + u * (t - bt)
    if side == 2: return l + u * (r - l), bt + d
    return l + u * (r - l), t + d


def _scoring_start(env, task, rs):
    """blocks to the neighbourhood of the task's scoring threshold; returns the point MoveToRegion's robot drives at (else None)"""
    from oracle.entities_ref import Shape as RefShape
    tk = env.task
    if task == 'MoveToCorner':
        def want(i, rs):
            d, phi = rs.uniform(0.8, 1.5), rs.uniform(0.15, np.pi / 2 - 0.15)
            return -1 + d * np.cos(phi), 1 - d * np.sin(phi)
        _place_blocks(env, rs, want, [tk.shape])
    elif task == 'MoveToRegion':
        return _edge_point(tk.goal.bb, rs, 0.12)
    elif task == 'MatchRegions':
        ents = tk.target_shapes + tk.distractor_shapes
        _place_blocks(env, rs, lambda i, rs: _edge_point(tk.sensor.bb, rs), ents)
    elif task == 'FindDupe':
        ents = [tk.query_block, *tk.outside_blocks]
        l, bt, r, t = tk.sensor.bb
        def want(i, rs):         # the duplicates mostly inside the region, everything else astride its edge
            if ents[i] in tk.target_set and rs.uniform() < 0.6:
                return rs.uniform(l + 0.08, r - 0.08), rs.uniform(bt + 0.08, t - 0.08)
            return _edge_point(tk.sensor.bb, rs)
        _place_blocks(env, rs, want, ents)
    elif task == 'FixColour':
        # a block whose colour is its region's sits just inside it, the others just outside theirs
        def want(i, rs):
            l, bt, r, t = tk.sensors[i].bb
            if tk.target_blocks[i]:
                return rs.uniform(l + 0.12, r - 0.12), rs.uniform(bt + 0.12, t - 0.12)
            x, y = _edge_point(tk.sensors[i].bb, rs, 0.0)
            cx, cy = (l + r) / 2, (bt + t) / 2
            n = np.hypot(x - cx, y - cy) + 1e-9
            return x + 0.3 * (x - cx) / n, y + 0.3 * (y - cy) / n
        _place_blocks(env, rs, want, tk.blocks)
    elif task == 'MakeLine':
        x0, y0, phi = rs.uniform(-0.35, 0.35), rs.uniform(-0.1, 0.45), rs.uniform(0, np.pi)
        def want(i, rs):
            u, v = 0.36 * (i - 1.5) + rs.uniform(-0.03, 0.03), rs.uniform(-0.2, 0.2)
            return x0 + u * np.cos(phi) - v * np.sin(phi), y0 + u * np.sin(phi) + v * np.cos(phi)
        _place_blocks(env, rs, want, tk.blocks)
    else:   # ClusterColour / ClusterShape: one loose heap per characteristic value
        groups = list(tk.blocks_by_characteristic.values())
        centres = []
        for g in groups:
            for attempt in range(200):
                c = rs.uniform(-0.6, 0.6, 2)
                if all(np.hypot(*(c - o)) > 0.75 for o in centres):
                    break
            centres.append(c)
        ents, cen = [], []
        for g, c in zip(groups, centres):
            for e in g:
                ents.append(e); cen.append(c)
        _place_blocks(env, rs, lambda i, rs: (cen[i][0] + rs.uniform(-0.42, 0.42), cen[i][1] + rs.uniform(-0.42, 0.42)), ents)
    return None


def _drive_at(env, tx, ty, s):
    b = env.bodies()
    rb = env.task.robot.bodies[0]
    x, y, a = b[rb, 0], b[rb, 1], b[rb, 2]
    hx, hy = -np.sin(a), np.cos(a)
    dx, dy = tx - x, ty - y
    if np.hypot(dx, dy) < 0.05:
        return 0
    err = np.arctan2(hx * dy - hy * dx, hx * dx + hy * dy)
    if abs(err) > 0.5:
        return 3 if err > 0 else 6
    if abs(err) > 0.15:
        return (3 if err > 0 else 6) + 1
    return 1


def _episodes_scoring(job):
    task, flags, lo, hi = job
    from oracle._lib import lib
    from oracle.entities_ref import Shape as RefShape
    from oracle.env_ref import RefEnv
    from tests.test_gpu_parity import _chase_action
    from tests.util import perturb_bodies
    L = lib()
    L.ref_set_unknowns(max(flags, 0))
    env = RefEnv(task, max_episode_steps=SCORING_STEPS)
    out = np.zeros((hi - lo, 2))
    for ep in range(lo, hi):
        env.reset()
        target = _scoring_start(env, task, np.random.RandomState(7 * 10 ** 6 + ep))
        if flags < 0:
            perturb_bodies(env, 1e-9, np.random.RandomState(10 ** 6 + ep))
        blocks = [e.bodies[0] for e in env.world.entities if isinstance(e, RefShape)]
        start = env.bodies()[blocks, :2].copy() if blocks else np.zeros((0, 2))
        for s in range(SCORING_STEPS):
            a = _drive_at(env, target[0], target[1], s) if target is not None else _chase_action(env, ep, s + 3)
            _, done, info = env.step(int(a))
        assert done
        disp = float(np.linalg.norm(env.bodies()[blocks, :2] - start, axis=1).mean()) if blocks else 0.0
        out[ep - lo] = (info['eval_score'], disp)
    L.ref_set_unknowns(0)
    return task, flags, lo, out


def _one_step(job):
    """states sampled from baseline rollouts: one env-step with and without each flip; render flips on the sampled states' frames"""
    task, n_states = job
    import ctypes as C
    from oracle._lib import lib
    from oracle.env_ref import FPS, RefEnv, area_downsample
    from tests.util import comparable_mask, ref_body_index
    L = lib()
    L.ref_set_unknowns(0)
    env = RefEnv(task)
    rs = np.random.RandomState(123)
    flips = [(n, f) for n, f in PHYS if f > 0 and f != 32]          # (the bevel radius is set when the shapes are made: trajectories only)
    dpose = {n: [] for n, _ in flips}
    contacts = []
    px = {n: [] for n, _ in RENDER}
    n_contact = 0
    for s in range(n_states):
        # half of the states are to have arbiters (that is where the solver's choices act): rollouts are re-drawn, up to 60 times, until
        # the sampled state has one
        for attempt in range(60):
            env.reset()
            for _ in range(rs.randint(1, env.max_episode_steps)):
                env.step(int(rs.randint(18)))
            if len(env.contacts()) > 0 or 2 * n_contact >= s + 1:
                break
        idx, mask = ref_body_index(env), comparable_mask(env)
        a = int(rs.randint(18))
        contacts.append(len(env.contacts()))
        n_contact += contacts[-1] > 0
        def stepped(flags):
            h = L.ref_clone(env.h)
            L.ref_set_unknowns(flags)
            L.ref_step(h, a, float(FPS))
            L.ref_set_unknowns(0)
            buf = np.zeros((L.ref_nbodies(h), 9))
            L.ref_get_bodies(h, buf.ctypes.data_as(C.POINTER(C.c_double)))
            L.ref_free(h)
            return buf[idx][:, :3]
        base = stepped(0)
        for n, f in flips:
            dpose[n].append(float(np.abs(stepped(f) - base)[mask].max()))
        frame0 = env.render_lores('ego')
        for n, f in RENDER:
            L.ref_set_unknowns(f)
            fr = env.render_lores('ego')
            L.ref_set_unknowns(0)
            d = np.abs(fr.astype(np.int32) - frame0.astype(np.int32))
            px[n].append((int((d != 0).sum()), int(d.max()), float(d.mean())))
    return task, {n: np.array(v) for n, v in dpose.items()}, np.array(contacts), {n: np.array(v) for n, v in px.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--episodes', type=int, default=4096)
    ap.add_argument('--states', type=int, default=256)
    ap.add_argument('--procs', type=int, default=os.cpu_count() or 1)
    ap.add_argument('--tasks', nargs='*', default=TASKS)
    ap.add_argument('--scoring-only', action='store_true', help='skip the random-action table A (round 5\'s), keep A2, B, C')
    args = ap.parse_args()
    t0 = time.time()
    print(__doc__.split('\n\n')[1] + '\n')
    chunk = max(16, args.episodes // (4 * args.procs))
    jobs = [(t, f, lo, min(lo + chunk, args.episodes)) for t in args.tasks for _, f in PHYS for lo in range(0, args.episodes, chunk)]
    # heavy tasks first so that the pool drains evenly
    weight = {'ClusterShape': 8, 'ClusterColour': 8, 'MakeLine': 6, 'MatchRegions': 3, 'FindDupe': 3, 'MoveToCorner': 1, 'FixColour': 1, 'MoveToRegion': 0}
    jobs.sort(key=lambda j: -weight.get(j[0], 1))
    res, res2 = {}, {}
    with mp.get_context('spawn').Pool(args.procs) as pool:
        one = pool.map_async(_one_step, [(t, args.states) for t in args.tasks])
        if not args.scoring_only:
            for task, flags, lo, out in pool.imap_unordered(_episodes, jobs):
                res.setdefault((task, flags), np.zeros((args.episodes, 2)))[lo:lo + len(out)] = out
        for task, flags, lo, out in pool.imap_unordered(_episodes_scoring, jobs):
            res2.setdefault((task, flags), np.zeros((args.episodes, 2)))[lo:lo + len(out)] = out
        one = {r[0]: r[1:] for r in one.get()}
    for label, res in ((f'A. eval_score over {args.episodes} random-action episodes per task and configuration (same action tapes), paired against the baseline', res),
                       (f'A2. eval_score over {args.episodes} episodes per task and configuration that START NEAR THE SCORING THRESHOLD and run {SCORING_STEPS} env-steps of a scripted pusher, paired against the baseline', res2)):
        if res:
            _score_table(args, label, res)
    _rest(args, one, t0)


def _score_table(args, label, res):
    print(label)
    print('   task           configuration                              mean score  (stderr)   score > 0   changed episodes   mean change (z)      mean block displacement (z of change)')
    verdict = {}
    for task in args.tasks:
        base = res[(task, 0)]
        for name, flags in PHYS:
            r = res[(task, flags)]
            sc, n = r[:, 0], len(r)
            d = sc - base[:, 0]
            dd = r[:, 1] - base[:, 1]
            z = d.mean() / (d.std(ddof=1) / np.sqrt(n)) if d.std() > 0 else 0.0
            zd = dd.mean() / (dd.std(ddof=1) / np.sqrt(n)) if dd.std() > 0 else 0.0
            print(f'   {task:14s} {name:42s} {sc.mean():9.5f}  ({sc.std(ddof=1) / np.sqrt(n):.5f})  {np.mean(sc > 0):8.4f}   {np.mean(np.abs(d) > 1e-12):12.4f}      '
                  f'{d.mean():+9.5f} ({z:+5.2f})      {r[:, 1].mean():8.5f} ({zd:+5.2f})')
            verdict.setdefault(name, []).append((task, z, zd, float(np.mean(np.abs(d) > 1e-12))))
        print()
    print('   summary: largest |z| of the mean score change / of the mean displacement change over the 8 tasks (the control shows what chaos alone does; |z| < 3 = no shift seen)')
    for name, flags in PHYS[1:]:
        v = verdict[name]
        worst = max(v, key=lambda x: abs(x[1])); worst_d = max(v, key=lambda x: abs(x[2]))
        print(f'   {name:42s} score: |z| max {abs(worst[1]):5.2f} ({worst[0]})   displacement: |z| max {abs(worst_d[2]):5.2f} ({worst_d[0]})   episodes changed: {min(x[3] for x in v):.3f} .. {max(x[3] for x in v):.3f}')
    print()


def _rest(args, one, t0):
    print(f'\nB. one env-step from the same state ({args.states} states per task sampled from baseline rollouts): max |pose difference| flipped vs unflipped')
    print('   task           flip                                       states with any difference   median      p90        max      (states with >= 1 arbiter)')
    for task in args.tasks:
        dpose, contacts, px = one[task]
        for name, v in dpose.items():
            print(f'   {task:14s} {name:42s} {np.mean(v > 0):10.3f}               {np.median(v):9.2e}  {np.percentile(v, 90):9.2e}  {v.max():9.2e}    ({np.mean(contacts > 0):.2f})')
        print()
    print(f'C. render flips on the ego 96x96x3 frames of the same states: bytes that differ (of 27648), largest and mean difference')
    for task in args.tasks:
        dpose, contacts, px = one[task]
        for name, v in px.items():
            print(f'   {task:14s} {name:44s} frames with any difference {np.mean(v[:, 0] > 0):.3f}; differing bytes mean {v[:, 0].mean():8.2f} max {int(v[:, 0].max()):5d}; '
                  f'largest difference {int(v[:, 1].max())}; mean |difference| over the frame {v[:, 2].mean():.5f}')
    print(f'\n({time.time() - t0:.0f} s on {args.procs} processes)')


if __name__ == '__main__':
    main()

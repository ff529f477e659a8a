// mgx_step.hip -- k_step / k_reset: the physics half of BaseEnv.step() for N envs in lockstep.
//
// Launch geometry (CDNA4): one 64-lane wavefront per workgroup; the wave is split into
// 64/L groups of L lanes and each group owns one env (L = 4..64, chosen per world by the host).
// All of an env's state is staged HBM -> LDS once per env-step, the 10 substeps x 10 solver
// iterations run out of LDS, and the state goes back once: HBM traffic per env-step is exactly
// the persistent state, read once and written once, in [row][env] SoA order so that the loads of
// consecutive envs coalesce.  N envs x L lanes / 64 workgroups: 4096 envs at L = 16 are 1024
// single-wave workgroups = 4 per CU, spread over all 8 XCDs by the dispatcher.
//
// No MFMA: there is no dense contraction anywhere on this path.
#include <hip/hip_runtime.h>

#include "mgx_sim.h"

namespace mgx {

// template blob in global memory: [TmplHeader][int words][R words][P words], P part 8-byte aligned.
// One blob shared by every env (tmpl_stride_words == 0), or one per env at words + env * tmpl_stride_words (tasks whose
// episodes differ in shape types / entity counts): those run one env per workgroup (L = 64), which stages its env's copy.
struct TmplDev {
    const uint32_t *words;
    int n_words;       // 32-bit words of the shared blob (per-env blobs carry their sizes in their headers)
    int off_i;         // word offset of the int array (the R and P arrays follow, see tmpl_off_r / tmpl_off_p)
    long tmpl_stride_words;    // 0: one template for all envs
    int off_r, off_p;          // shared blob: word offsets of the R and P arrays
    int env_off_r, env_off_i;  // shared blob: word offsets of the R and int regions inside an env's slab (P region first)
    int env_stride_words;      // per-env LDS stride of the working set in 32-bit words (multiple of 2)
    int lds_tmpl_words;        // LDS words reserved per template copy (multiple of 2)
    unsigned long long *dbg_clk;   // development probe (MGX_STEP_PROBE builds): per-workgroup phase cycles [blocks][32]
    // longest-first dispatch (worlds whose step workgroups take several dispatch rounds): workgroup blockIdx.x steps the envs of
    // group order[blockIdx.x]; every group leaves its duration (shader clock >> 6) in dur[group] for k_step_order.  NULL = off
    const uint32_t *order;
    uint32_t *dur;
    // heavy envs together (worlds whose step workgroups are all resident at once, several envs per wavefront): the env at position p of
    // the launch is env_order[p], and every env leaves a cost key (contact points and overlapping pairs of its last substep) in
    // env_cost[env] for k_env_order.  A wavefront lasts as long as its slowest env, and a CU gives its SIMDs to the rasteriser only
    // once its slowest step wavefront is through: envs in contact share wavefronts instead of each holding up three light ones.
    // Never changes what an env computes.  NULL = envs in index order
    const uint32_t *env_order;
    uint32_t *env_cost;
};
// Producer side of the step -> raster hand-off (mgx_engine_step_render): a workgroup that has written its envs' state back
// publishes them, one 64-bit entry per env ((epoch << 32) | env), into a queue that raster workgroups of a concurrently
// running k_raster consume in arrival order.  Placement-independent protocol: plain state stores -> s_waitcnt -> agent-scope
// release -> s_waitcnt -> relaxed agent-scope ticket + entry store; the consumer polls its entry relaxed and acquires once.
struct StepHandoff {
    unsigned long long *queue;   // [n_envs]; NULL = no hand-off (plain launch)
    unsigned *tail;              // tickets handed out so far, all calls (monotonic)
    unsigned *started;           // step workgroups that have begun executing, all calls (monotonic)
    unsigned base;               // value of *tail before this call
    unsigned epoch;              // this call's tag
    unsigned delay_every, delay_sleeps;      // tests (mgx_engine_debug_handoff): every delay_every-th workgroup sleeps delay_sleeps x ~1 us before it publishes (0: off)
};
MGX_HD int even_words(int x) { return (x + 1) & ~1; }
template <typename R> MGX_HD int tmpl_off_r(const TmplHeader &h, int off_i) { return even_words(off_i + h.n_words_i); }
template <typename R, typename P> MGX_HD int tmpl_off_p(const TmplHeader &h, int off_i) {
    return even_words(tmpl_off_r<R>(h, off_i) + h.n_words_r * (int)(sizeof(R) / 4));
}
template <typename R, typename P> MGX_HD int tmpl_total_words(const TmplHeader &h, int off_i) {
    return even_words(tmpl_off_p<R, P>(h, off_i) + h.n_words_p * (int)(sizeof(P) / 4));
}

#ifdef MGX_STEP_PROBE
constexpr bool probe_prefix(const char *s, const char *p) { return *p == 0 ? true : (*s == *p && probe_prefix(s + 1, p + 1)); }
constexpr int probe_phase_id(const char *s) {
    const char *names[] = {"ph_init_work", "ph_load_state", "ph_integrate", "ph_shapes", "ph_broad", "ph_broad_(unused)", "ph_narrow",
                           "ph_arbiters_joints", "solve_begin", "solve_warm_contacts", "solve_warm_pg", "solve_iter_publish",
                           "solve_iter_contacts", "solve_iter_pg", "solve_end", "ph_cache_commit", "solve_warm_chain", "solve_iter_chain"};
    for (int i = 0; i < 18; i++) if (probe_prefix(s, names[i])) return i;
    return 19;
}
#endif
// One env per wavefront (L = 64, the per-env-world mode) launches as many waves as envs: capping its registers for
// MGX_L64_WAVES waves per SIMD keeps more of them resident (the narrower groups run one wave per SIMD at 4096 envs and are
// fastest with all the registers)
#ifndef MGX_L64_WAVES
#define MGX_L64_WAVES 2
#endif
#ifndef MGX_LN_WAVES
#define MGX_LN_WAVES 1      // register budget of the several-envs-per-wave instantiations, in waves per SIMD (512 / n VGPRs)
#endif
// Register budget of the several-envs-per-wave instantiations: 224 of the SIMD's 512 (amdgpu_num_vgpr counts in units of two on
// gfx90a+, where VGPRs and AGPRs are one file), a dozen values spilled outside the substep loop -- so that THREE 96-register
// rasteriser wavefronts fit beside a step wavefront in the fused env-step (mgx_engine_step_render) instead of two.
#ifndef MGX_STEP_NUM_VGPR
#define MGX_STEP_NUM_VGPR 112
#endif
#define MGX_STEP_VGPR_ATTR __attribute__((amdgpu_num_vgpr(MGX_STEP_NUM_VGPR)))
template <typename R, typename P, int L>
__device__ __forceinline__ void step_body(const TmplDev &t, P *__restrict__ sp, R *__restrict__ sf, int32_t *__restrict__ si,
                                          const int32_t *__restrict__ actions, uint8_t *__restrict__ done,
                                          int n_envs, int n_sub, int count_step, int iterations, const StepHandoff &ho) {
    extern __shared__ __align__(16) uint32_t lds[];
    const int tid = threadIdx.x;
    if (ho.queue) {
        // consumers only wait for producers that are already running (never for ones that still need the consumers' slots)
        if (tid == 0) __hip_atomic_fetch_add(ho.started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_setprio(3);      // the serial chain below goes first; co-resident raster waves fill its bubbles
    }
    constexpr int EPB = 64 / L;
    const int env_local = tid / L, lane = tid % L, nl = L;
    // consecutive workgroups land on different XCDs (round robin over the 8 L2s); give each XCD a contiguous env
    // range so that the [row][env] lines shared by neighbouring workgroups are fetched into one L2 only
    int wg = blockIdx.x;
    const unsigned long long t_begin = t.dur ? __builtin_amdgcn_s_memtime() : 0ull;
#ifdef MGX_RASTER_CLOCKS    // development build: where and when this workgroup ran (tools/dev/placement_probe.py)
    const unsigned long long pl_begin = wall_clock64();
#endif
    if (t.order) wg = (int)t.order[wg];
    else if (!t.env_order && (gridDim.x & 7) == 0) wg = (wg & 7) * (gridDim.x >> 3) + (wg >> 3);
#ifndef MGX_STEP_CU_COLOCATE
#define MGX_STEP_CU_COLOCATE 1
#endif
    // The dispatcher deals workgroups b, b + 256, b + 512, ... onto the SAME CU (round robin over 8 XCDs x 4 SEs x 8 CUs, measured:
    // tools/dev/placement_probe.py).  A step wavefront that is still running holds 224 registers of its SIMD, and a rasteriser
    // workgroup needs a wavefront on each of the CU's four SIMDs: ONE late step workgroup costs its CU two of five rasteriser
    // workgroups.  So the heaviest positions of the cost-sorted env order go to workgroups of one CU: the late ones then block a
    // quarter as many CUs.
    int pos_wg = wg;
    if (t.env_order && MGX_STEP_CU_COLOCATE && (gridDim.x & 255) == 0) pos_wg = (wg & 255) * (int)(gridDim.x >> 8) + (wg >> 8);
    // validity follows from the POSITION this lane serves, whatever permutation produced it (advisor, round 4)
    long env = (long)pos_wg * EPB + env_local;
    const bool valid = env < n_envs;
    if (!valid) env = n_envs - 1;   // tail lanes shadow another env (no stores) so barriers stay uniform
    if (t.env_order) env = (long)t.env_order[env];      // (position -> env: a permutation of 0 .. n_envs - 1, heaviest first)
    // the template in LDS.  Per-env worlds run one env per workgroup (L = 64), so the copy -- this env's own -- is still
    // shared by all lanes and every template address stays wave-uniform
    const bool per_env = L == 64 && t.tmpl_stride_words != 0;
    {
        const uint32_t *src = t.words;
        int n = t.n_words;
        if (per_env) {
            src += (long)__builtin_amdgcn_readfirstlane((int)env) * t.tmpl_stride_words;
            n = tmpl_total_words<R, P>(*reinterpret_cast<const TmplHeader *>(src), t.off_i);
        }
        for (int i = tid; i < n; i += 64) lds[i] = src[i];
    }
    __syncthreads();
    const TmplHeader *h = reinterpret_cast<const TmplHeader *>(lds);
    const int32_t *ti = reinterpret_cast<const int32_t *>(lds + t.off_i);
    const R *tr = reinterpret_cast<const R *>(lds + (per_env ? tmpl_off_r<R>(*h, t.off_i) : t.off_r));
    const P *tp = reinterpret_cast<const P *>(lds + (per_env ? tmpl_off_p<R, P>(*h, t.off_i) : t.off_p));
    uint32_t *slab = lds + t.lds_tmpl_words + env_local * t.env_stride_words;
    // the working set inside the slab: pose region, real region, int region (per-env worlds: of THIS env's world)
    int slab_off_r = t.env_off_r, slab_off_i = t.env_off_i;
    if (per_env) {
        const WorkOff wo_(*h);
        slab_off_r = even_words(wo_.n_p * (int)(sizeof(P) / 4)); slab_off_i = slab_off_r + even_words(wo_.n_r * (int)(sizeof(R) / 4));
    }
    Env<R, P> e(h, ti, tr, tp, reinterpret_cast<R *>(slab + slab_off_r), reinterpret_cast<P *>(slab),
                reinterpret_cast<int32_t *>(slab + slab_off_i));
    const long stride = n_envs;
    SolveCtx<R> ctx;

#ifdef MGX_STEP_PROBE
    // development build: shader cycles per phase, accumulated over the launch (tools/step_phase_probe.py)
    unsigned long long pacc[20] = {0};
#define SYNC(stmt) { constexpr int pid_ = probe_phase_id(#stmt); const unsigned long long t0_ = __builtin_amdgcn_s_memtime(); \
                     stmt; __syncthreads(); pacc[pid_] += __builtin_amdgcn_s_memtime() - t0_; }
#else
// The workgroup is ONE wavefront: its LDS operations execute in program order, so what a phase boundary needs is only that
// the compiler keeps the phases' LDS accesses in order -- a wavefront-scope fence, not s_waitcnt + s_barrier.
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#ifdef MGX_PHASE_MARKERS   // development: phase names as comments in the -S output (tools/step_asm_phases.py)
#define SYNC(stmt) asm volatile("; MGX_PHASE " #stmt ::: "memory"); stmt; WAVE_SYNC();
#else
#define SYNC(stmt) stmt; WAVE_SYNC();
#endif
#endif
    SYNC(ph_init_work(e, lane, nl))
    SYNC(ph_load_state(e, sp, sf, si, stride, env, lane, nl))
    ph_refresh_trig(e, lane, nl);
    if (lane == 0) {
        E_I(misc, M_ACTION) = actions[env];
        // the episode step counter and the done flag depend on nothing below: settle them here, so that `done` and
        // `count_step` are not carried through the whole kernel
        if (count_step) {
            const int steps = E_I(misc, M_STEPS) + 1;
            E_I(misc, M_STEPS) = steps;
            if (done && valid) done[env] = steps >= h->max_episode_steps ? 1 : 0;
        }
    }
    __syncthreads();
    if (lane == 0) ph_control(e);
    __syncthreads();
    solve_ctx_init(e, ctx, lane, nl);
    // The hand-off to the rasteriser (fused env-step) happens as soon as the poses are final -- after the LAST substep's position update:
    // what that substep still does (collisions at the new positions, the solve) only sets up the velocities of the next env-step.
    // The pose rows go out and are published there, a tenth of the kernel before its end; everything else is stored at the end.
#ifndef MGX_EARLY_HANDOFF
#define MGX_EARLY_HANDOFF 1      // 0: publish at the end of the kernel (A/B builds)
#endif
    const bool early = MGX_EARLY_HANDOFF && ho.queue != nullptr && n_sub > 0;
    for (int sub = 0; sub < n_sub; sub++) {
        SYNC(ph_integrate(e, lane, nl))
        if (early && sub == n_sub - 1) {
            if (valid) ph_store_poses(e, sp, stride, env, lane, nl);
            if (ho.delay_every && blockIdx.x % ho.delay_every == 0) for (unsigned i = 0; i < ho.delay_sleeps; i++) __builtin_amdgcn_s_sleep(127);      // (a late producer, forced: tests)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the compiler may drop the fence's own wait: keep this one)
            if (valid && lane == 0) {
                const unsigned ticket = __hip_atomic_fetch_add(ho.tail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ticket - ho.base < (unsigned)n_envs)       // (never false while the host's mirror of *tail is right: keeps a wrong base in bounds)
                    __hip_atomic_store(&ho.queue[ticket - ho.base], ((unsigned long long)ho.epoch << 32) | (unsigned long long)env,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        MGX_SUBSTEP_AFTER_INTEGRATE(SYNC)
    }
    SYNC(solve_ctx_flush(e, ctx, lane, nl))
    if (valid) ph_store_state(e, sp, sf, si, stride, env, lane, nl, early);
    if (t.env_cost && valid && lane == 0) t.env_cost[env] = (uint32_t)(4 * E_I(misc, M_NK) + E_I(misc, M_NOV));
    if (ho.queue && !early) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the compiler may drop the fence's own wait: keep this one)
        if (valid && lane == 0) {
            const unsigned ticket = __hip_atomic_fetch_add(ho.tail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ticket - ho.base < (unsigned)n_envs)
                __hip_atomic_store(&ho.queue[ticket - ho.base], ((unsigned long long)ho.epoch << 32) | (unsigned long long)env,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (t.dur && tid == 0) t.dur[wg] = (uint32_t)((__builtin_amdgcn_s_memtime() - t_begin) >> 6);
#ifdef MGX_STEP_PROBE
    if (t.dbg_clk && tid == 0) for (int i = 0; i < 20; i++) t.dbg_clk[(long)blockIdx.x * 32 + i] = pacc[i];
#elif defined(MGX_RASTER_CLOCKS)
    if (t.dbg_clk && tid == 0) {
        unsigned long long *d = t.dbg_clk + (long)blockIdx.x * 4;
        d[0] = __builtin_amdgcn_s_getreg(63492); d[1] = __builtin_amdgcn_s_getreg(63508); d[2] = pl_begin; d[3] = wall_clock64();   // HW_ID, XCC_ID
    }
#endif
#undef SYNC
}
// several envs per wavefront (L = 16, 32) ...
template <typename R, typename P, int L>
__global__ __launch_bounds__(64, MGX_LN_WAVES) MGX_STEP_VGPR_ATTR void k_step(TmplDev t, P *__restrict__ sp, R *__restrict__ sf, int32_t *__restrict__ si,
                                                                              const int32_t *__restrict__ actions, uint8_t *__restrict__ done,
                                                                              int n_envs, int n_sub, int count_step, int iterations, StepHandoff ho) {
    step_body<R, P, L>(t, sp, sf, si, actions, done, n_envs, n_sub, count_step, iterations, ho);
}
// (the all-fp64 validation build keeps the whole register file: at 224 its solver would run out of scratch memory)
template <typename R, typename P, int L>
__global__ __launch_bounds__(64, MGX_LN_WAVES) void k_step_wide(TmplDev t, P *__restrict__ sp, R *__restrict__ sf, int32_t *__restrict__ si,
                                                                const int32_t *__restrict__ actions, uint8_t *__restrict__ done,
                                                                int n_envs, int n_sub, int count_step, int iterations, StepHandoff ho) {
    step_body<R, P, L>(t, sp, sf, si, actions, done, n_envs, n_sub, count_step, iterations, ho);
}
// ... and one env per wavefront (the per-env-world mode)
template <typename R, typename P>
__global__ __launch_bounds__(64, MGX_L64_WAVES) void k_step_env(TmplDev t, P *__restrict__ sp, R *__restrict__ sf, int32_t *__restrict__ si,
                                                                const int32_t *__restrict__ actions, uint8_t *__restrict__ done,
                                                                int n_envs, int n_sub, int count_step, int iterations, StepHandoff ho) {
    step_body<R, P, 64>(t, sp, sf, si, actions, done, n_envs, n_sub, count_step, iterations, ho);
}

// Longest-first dispatch order of the step workgroups for the NEXT launch, from the durations the last one left: a counting
// sort on 256 duration classes (one workgroup; the order inside a class is whatever the atomics give -- it only affects when a
// group runs, never what it computes).  A workgroup's slowest envs are slow again a step later (the same blocks still touch), and
// with several dispatch rounds the launch ends when the last-started long workgroup does: longest first, the tail of the launch is
// made of short ones.
__global__ __launch_bounds__(1024) void k_step_order(const uint32_t *__restrict__ dur, uint32_t *__restrict__ order, int n) {
    __shared__ uint32_t s_max, hist[256];
    const int tid = threadIdx.x;
    if (tid == 0) s_max = 1;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    uint32_t m = 0;
    for (int i = tid; i < n; i += 1024) m = dur[i] > m ? dur[i] : m;
    atomicMax(&s_max, m);
    __syncthreads();
    const uint32_t top = s_max;
    auto cls = [&](uint32_t d) { return 255u - (uint32_t)(((unsigned long long)d * 255ull) / top); };     // 0 = longest
    for (int i = tid; i < n; i += 1024) atomicAdd(&hist[cls(dur[i])], 1u);
    __syncthreads();
    if (tid == 0) { uint32_t acc = 0; for (int b = 0; b < 256; b++) { const uint32_t c = hist[b]; hist[b] = acc; acc += c; } }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) order[atomicAdd(&hist[cls(dur[i])], 1u)] = (uint32_t)i;
}

// Env order of the NEXT launch from the cost keys the last one left (TmplDev::env_order): a counting sort, costliest first, ties in
// env order (a stable sort keeps neighbouring envs -- and their shared state lines -- together where nothing touches).
__global__ __launch_bounds__(1024) void k_env_order(const uint32_t *__restrict__ cost, uint32_t *__restrict__ order, int n) {
    constexpr int CHUNK = 8192;                         // envs whose classes are staged in LDS at a time
    __shared__ uint32_t hist[64], base[64];
    __shared__ uint8_t cls_of[CHUNK];
    const int tid = threadIdx.x;
    auto cls = [](uint32_t c) { return 63u - (c > 63u ? 63u : c); };        // 0 = costliest
    if (tid < 64) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) atomicAdd(&hist[cls(cost[i])], 1u);
    __syncthreads();
    if (tid == 0) { uint32_t acc = 0; for (int b = 0; b < 64; b++) { base[b] = acc; acc += hist[b]; } }
    __syncthreads();
    // stable placement: one wavefront per NON-EMPTY class walks the envs in order, their classes read from LDS (the keys are small:
    // a handful of classes are populated; round 4's first form walked all 64 classes over global memory: 166 us per launch)
    const int wave = tid >> 6, lane = tid & 63;
    for (int c0 = 0; c0 < n; c0 += CHUNK) {
        const int m = n - c0 < CHUNK ? n - c0 : CHUNK;
        __syncthreads();
        for (int i = tid; i < m; i += 1024) cls_of[i] = (uint8_t)cls(cost[c0 + i]);
        __syncthreads();
        int slot = 0;                                   // the slot-th non-empty class goes to wavefront slot % 16
        for (int b = 0; b < 64; b++) {
            if (hist[b] == 0) continue;
            if ((slot++ & 15) != wave) continue;
            uint32_t at = base[b];
            for (int i0 = 0; i0 < m; i0 += 64) {
                const int i = i0 + lane;
                const bool mine = i < m && cls_of[i] == (uint8_t)b;
                const unsigned long long mk = __ballot(mine);
                if (mine) order[at + __popcll(mk & ((1ull << lane) - 1ull))] = (uint32_t)(c0 + i);
                at += (uint32_t)__popcll(mk);
            }
            base[b] = at;                               // (only this wavefront touches class b: next chunk continues here)
        }
    }
}

// BaseEnv.reset(): one thread per env writes the template state into the masked envs
template <typename R, typename P>
__global__ __launch_bounds__(64) void k_reset(TmplDev t, P *__restrict__ sp, R *__restrict__ sf, int32_t *__restrict__ si,
                                              const uint8_t *__restrict__ mask, const P *__restrict__ ent_pose, int n_envs) {
    extern __shared__ __align__(16) uint32_t lds[];
    if (t.tmpl_stride_words == 0) for (int i = threadIdx.x; i < t.n_words; i += 64) lds[i] = t.words[i];
    __syncthreads();
    long env = (long)blockIdx.x * 64 + threadIdx.x;
    if (env >= n_envs) return;
    if (mask && !mask[env]) return;
    // per-env worlds: each thread reads its env's template straight from HBM (resets are rare)
    const uint32_t *tl = t.tmpl_stride_words == 0 ? lds : t.words + env * t.tmpl_stride_words;
    const TmplHeader *h = reinterpret_cast<const TmplHeader *>(tl);
    const int32_t *ti = reinterpret_cast<const int32_t *>(tl + t.off_i);
    const R *tr = reinterpret_cast<const R *>(tl + tmpl_off_r<R>(*h, t.off_i));
    const P *tp = reinterpret_cast<const P *>(tl + tmpl_off_p<R, P>(*h, t.off_i));
    reset_env_state<R, P>(*h, ti, tr, tp, sp, sf, si, (long)n_envs, env, ent_pose);
}

}  // namespace mgx

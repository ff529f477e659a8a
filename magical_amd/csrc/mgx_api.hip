// mgx_api.hip -- the C ABI declared in include/mgx.h: world description, engine lifetime and
// kernel launches.  Device state blobs and output tensors are owned by the caller (PyTorch);
// the engine owns only its small constant template buffers and timing events.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <array>
#include <atomic>
#include <mutex>
#include <iterator>
#include <thread>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <unistd.h>
#include <type_traits>

#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mgx.h"
#include "mgx_raster.hip"
#include "mgx_score.hip"
#include "mgx_step.hip"
#include "mgx_world.h"

using namespace mgx;

namespace {
thread_local std::string g_err;
int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIP_OK(expr)                                                                            \
    do {                                                                                        \
        hipError_t err__ = (expr);                                                              \
        if (err__ != hipSuccess) return fail(MGX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(err__)); \
    } while (0)

constexpr int BG_RGB = 231 | (231 << 8) | (234 << 16);   // lighten_rgb(grey, 4), base_env.py:186
constexpr int MAX_LDS_BYTES = 160 * 1024;
// gfx950 hands LDS out in pieces of 320 dwords (160 KB / 128): five workgroups per CU fit below 32 000 bytes each, not 32 768 (measured:
// a 32 352-byte rasteriser layout ran four per CU -- ClusterColour's k_raster 0.514 -> 0.556 ms)
static size_t lds_alloc_bytes(size_t lds) { return (lds + 1279) / 1280 * 1280; }
constexpr int TIMING_RING = 4096;
constexpr int MAX_DEVICES = 64;

// Every entry point that touches the device runs on the ENGINE's device and leaves the caller's current device as it
// found it (a process may drive several GPUs, one engine each).
struct DeviceGuard {
    int prev = -1; bool ok = true;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) ok = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define ON_DEVICE(e) DeviceGuard guard__((e)->device); if (!guard__.ok) return fail(MGX_ERR_HIP, "hipSetDevice failed")

// Dynamic LDS above 64 KB needs an opt-in per kernel function AND per device: remember the largest size granted so far for
// each (kernel function, device).  Keyed by the kernel's address: instantiations of one template share their function-pointer
// type, so a table per type (or per tag declared in a generic lambda) would be shared by all of them.
static int ensure_lds(const void *kern, size_t lds, int device) {
    static std::mutex mu;
    static std::unordered_map<const void *, std::array<size_t, MAX_DEVICES>> granted;
    if (lds > (size_t)MAX_LDS_BYTES) return fail(MGX_ERR_CAPACITY, "kernel working set does not fit the CU's 160 KB of LDS");
    if (lds <= 65536) return MGX_OK;
    const int d = device >= 0 && device < MAX_DEVICES ? device : 0;
    std::lock_guard<std::mutex> lock(mu);
    auto it = granted.find(kern);
    if (it == granted.end()) it = granted.emplace(kern, std::array<size_t, MAX_DEVICES>{}).first;
    if (lds > it->second[d]) {
        HIP_OK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        it->second[d] = lds;
    }
    return MGX_OK;
}
}  // namespace

struct mgx_world { World w; };

// host copy of one world's two device blobs
struct WorldBlobs {
    TmplHeader h;                    // full header (with the draw list)
    std::vector<uint32_t> step;      // [header without prims][ints][R words][P words]
    std::vector<uint32_t> raster;    // [header][ints][prim reals + prim verts, fp64]
    int step_env_stride = 0;         // LDS words of the per-env working set
    int step_off_r = 0, step_off_p = 0, step_env_off_r = 0, step_env_off_i = 0;
    int raster_lds_words = 0;      // the draw-list blob's LDS-resident prefix (header + ints; the fp64 part is read from HBM by the set-up)
    int raster_scratch_d = 0, raster_scratch_dc = 0, raster_n_i = 0;      // doubles (stored / compact vertex records, RasterOff) / ints of the rasteriser's per-env scratch
};

struct mgx_engine {
    World w;
    TmplHeader h;               // of the default world (capacity world once per-env worlds are enabled: sizes only)
    int n_envs = 0, device = 0, dtype = 0, L = 0, L_request = 0;
    uint32_t *d_step = nullptr, *d_raster = nullptr;
    int32_t *d_palette = nullptr;
    TmplDev tdev{};
    RasterDev rdev{};
    int raster_waves = 4;       // k_raster variant: workgroups per CU its register cap is set for (3, 4 or 5)
    int qcap_debug = 0;         // > 0: queue entries in use as set by mgx_engine_debug_raster_qcap (tests); else the layout's capacity
    size_t lds_step = 0, lds_raster = 0;
    int rows_p = 0, rows_f = 0, rows_i = 0;
    // per-env worlds (tasks whose episodes differ in shape types / entity counts)
    bool env_worlds = false;
    std::vector<std::shared_ptr<World>> env_world;                       // [n_envs]; empty pointer = the default world
    std::unordered_map<std::string, std::weak_ptr<World>> world_by_sig;  // live variants, shared between envs
    int step_stride = 0, raster_stride = 0;                             // words per env in the two blob tables
    // footprints of the envs' current worlds: the launch geometry follows their maxima, not the capacity world's
    std::vector<int> fp_step_words, fp_env_stride, fp_raster_words, fp_raster_full, fp_scratch_d, fp_scratch_dc, fp_raster_n_i;
    uint32_t *d_stage = nullptr; size_t stage_words = 0;                 // upload staging (device)
    uint32_t *h_stage = nullptr; size_t h_stage_words = 0;               // upload staging (pinned host memory)
    int32_t *d_stage_idx = nullptr; size_t stage_idx_n = 0;       // (env, offsets, sizes) rows of an upload
    // set_env_variants does not wait for its uploads: what they read stays untouched until the next call has waited for ev_variants
    hipEvent_t ev_variants = nullptr; bool variants_pending = false;
    std::vector<int32_t> v_rows, v_idx; std::vector<int8_t> v_ty; std::vector<uint8_t> v_on;
    int8_t *d_v_ty = nullptr; uint8_t *d_v_on = nullptr; int32_t *d_v_idx = nullptr; size_t d_v_ent_cap = 0, d_v_idx_cap = 0;
    // k_score (goal-region overlap sets): fp64 shape library, entity / body / goal tables of the engine's world, and with
    // per-env worlds the envs' shape types and presence flags
    ScoreLib *d_score_lib = nullptr;
    int32_t *d_score_ent = nullptr, *d_score_prow = nullptr, *d_score_goal_ent = nullptr;
    double *d_score_goal_xyhw = nullptr;
    int8_t *d_ent_type_env = nullptr; uint8_t *d_ent_present_env = nullptr;
    int n_goals = 0;
    // step -> raster hand-off (mgx_engine_step_render): second stream + events, the queue of finished envs and its counters
    // longest-first dispatch of the step workgroups (launch_step_L): last durations, the order made of them, their capacity
    uint32_t *d_dur = nullptr, *d_order = nullptr; int order_cap = 0, n_cus = 0; bool order_valid = false;
    uint32_t *d_env_cost = nullptr, *d_env_order = nullptr; bool env_order_valid = false, env_sort_pending = false;     // heavy envs together (TmplDev::env_order)
    hipStream_t st2 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    unsigned long long *d_queue = nullptr; unsigned *d_hand = nullptr, *d_deferred = nullptr;   // d_hand: tail, started, stats[2]
    unsigned hand_tail = 0, hand_started = 0, hand_epoch = 0;                                    // host mirrors of the monotonic counters
    int cap_prims = 64;         // per-env worlds: primitives of the capacity world's draw list (mgx_engine_enable_env_worlds)
    unsigned dbg_poll_limit = 0, dbg_delay_every = 0, dbg_delay_sleeps = 0;                      // tests: forced hand-off failures (mgx_engine_debug_handoff; 0 = shipped behaviour)
    hipStream_t peek_stream = nullptr; unsigned long long *peek_q = nullptr; unsigned *peek_h = nullptr;   // mgx_engine_debug_handoff_peek's own stream and pinned buffers
    int timing = 0;             // 0 = off, n = bracket every n-th launch of each kind with HIP events
    int launch_count[2] = {0, 0};
    int dbg_iterations = -1;    // development probe: override the solver iteration count
    std::vector<hipEvent_t> ev[2];      // per kernel kind: start/stop pairs
    int ev_count[2] = {0, 0};
};

extern "C" {

const char *mgx_last_error(void) { return g_err.c_str(); }
int mgx_version(void) { return 1; }

// ------------------------------------------------------------------ world
int mgx_world_create(mgx_world **out) {
    if (!out) return fail(MGX_ERR_ARG, "out is NULL");
    *out = new mgx_world();
    return MGX_OK;
}
void mgx_world_destroy(mgx_world *w) { delete w; }
int mgx_world_set_phys_vars(mgx_world *w, const double vars[5]) {
    if (!w || !vars) return fail(MGX_ERR_ARG, "NULL argument");
    if (w->w.finalized) return fail(MGX_ERR_STATE, "world already finalized");
    for (int i = 0; i < 5; i++) {
        if (!(vars[i] > 0)) return fail(MGX_ERR_ARG, "physics variables must be positive");
        w->w.phys_vars[i] = vars[i];
    }
    return MGX_OK;
}
static int add_entity(mgx_world *w, const EntityDef &e) {
    if (!w) return fail(MGX_ERR_ARG, "world is NULL");
    if (w->w.finalized) return fail(MGX_ERR_STATE, "world already finalized");
    w->w.entities.push_back(e);
    return (int)w->w.entities.size() - 1;
}
int mgx_world_add_robot(mgx_world *w, double x, double y, double angle) {
    EntityDef e{}; e.kind = 0; e.x = x; e.y = y; e.angle = angle; e.body = -1;
    return add_entity(w, e);
}
int mgx_world_add_shape(mgx_world *w, int shape_type, int colour, double x, double y, double angle) {
    if (shape_type < 0 || shape_type > MGX_STAR) return fail(MGX_ERR_ARG, "bad shape_type");
    if (colour < 0 || colour > MGX_YELLOW) return fail(MGX_ERR_ARG, "bad colour");
    EntityDef e{}; e.kind = 1; e.shape_type = shape_type; e.colour = colour; e.x = x; e.y = y; e.angle = angle; e.body = -1;
    return add_entity(w, e);
}
int mgx_world_add_goal(mgx_world *w, double x, double y, double h, double w_, int colour) {
    if (!(h > 0) || !(w_ > 0)) return fail(MGX_ERR_ARG, "goal region needs h > 0 and w > 0");
    if (colour < 0 || colour > MGX_YELLOW) return fail(MGX_ERR_ARG, "bad colour");
    EntityDef e{}; e.kind = 2; e.colour = colour; e.x = x; e.y = y; e.h = h; e.w = w_; e.body = -1;
    return add_entity(w, e);
}
int mgx_world_finalize(mgx_world *w, int max_episode_steps) {
    if (!w) return fail(MGX_ERR_ARG, "world is NULL");
    if (max_episode_steps <= 0) return fail(MGX_ERR_ARG, "max_episode_steps must be positive");
    std::string err;
    int rc = w->w.finalize(max_episode_steps, err);
    if (rc) return fail(rc == -2 ? MGX_ERR_CAPACITY : (rc == -3 ? MGX_ERR_STATE : MGX_ERR_ARG), err);
    return MGX_OK;
}
int mgx_world_variant(const mgx_world *w, const uint8_t *enabled, const int32_t *shape_types, mgx_world **out) {
    if (!w || !out) return fail(MGX_ERR_ARG, "NULL argument");
    if (!w->w.finalized) return fail(MGX_ERR_STATE, "world not finalized");
    mgx_world *v = new mgx_world();
    std::string err;
    std::vector<int> st;
    if (shape_types) st.assign(shape_types, shape_types + w->w.entities.size());
    int rc = w->w.variant(enabled, shape_types ? st.data() : nullptr, v->w, err);
    if (rc) { delete v; return fail(rc == -2 ? MGX_ERR_CAPACITY : MGX_ERR_ARG, err); }
    *out = v;
    return MGX_OK;
}
int mgx_world_info(const mgx_world *w, int key, int *out) {
    if (!w || !out) return fail(MGX_ERR_ARG, "NULL argument");
    if (!w->w.finalized) return fail(MGX_ERR_STATE, "world not finalized");
    TmplHeader h; std::vector<int32_t> iw; std::vector<double> rw, pw;
    w->w.serialise(h, iw, rw, pw);
    switch (key) {
        case MGX_INFO_N_BODIES: *out = h.n_bodies; break;
        case MGX_INFO_N_SHAPES: *out = h.n_shapes; break;
        case MGX_INFO_N_JOINTS: *out = h.n_joints; break;
        case MGX_INFO_N_PAIRS: *out = h.n_pairs; break;
        case MGX_INFO_N_PRIMS: *out = h.n_prims; break;
        case MGX_INFO_STATE_ROWS_P: *out = state_rows_p(h); break;
        case MGX_INFO_STATE_ROWS_F: *out = state_rows_f(h); break;
        case MGX_INFO_STATE_ROWS_I: *out = state_rows_i(h); break;
        case MGX_INFO_ROBOT_BODY: *out = h.robot_body; break;
        case MGX_INFO_N_ENTITIES: *out = (int)w->w.entities.size(); break;
        case MGX_INFO_CACHE_SLOTS: *out = h.cache_slots; break;
        case MGX_INFO_MAX_CONTACTS: *out = h.max_contacts; break;
        case MGX_INFO_MAX_EPISODE_STEPS: *out = h.max_episode_steps; break;
        case MGX_INFO_N_JACC: *out = h.n_jacc; break;
        case MGX_INFO_PHYSVAR_ROW: *out = state_row_physvar(h, 0); break;
        default: return fail(MGX_ERR_ARG, "unknown info key");
    }
    return MGX_OK;
}
int mgx_world_entity(const mgx_world *w, int ent, int *kind, int *body, int *shape_type, int *colour) {
    if (!w || !w->w.finalized) return fail(MGX_ERR_STATE, "world not finalized");
    if (ent < 0 || ent >= (int)w->w.entities.size()) return fail(MGX_ERR_ARG, "entity index out of range");
    const EntityDef &e = w->w.entities[ent];
    if (kind) *kind = e.kind;
    if (body) *body = e.body;
    if (shape_type) *shape_type = e.shape_type;
    if (colour) *colour = e.colour;
    return MGX_OK;
}
int mgx_world_body_table(const mgx_world *w, double *mass_inv, double *init_pose) {
    if (!w || !w->w.finalized) return fail(MGX_ERR_STATE, "world not finalized");
    for (size_t b = 0; b < w->w.bodies.size(); b++) {
        if (mass_inv) { mass_inv[2 * b] = w->w.bodies[b].m_inv; mass_inv[2 * b + 1] = w->w.bodies[b].i_inv; }
        if (init_pose) { init_pose[3 * b] = w->w.bodies[b].x; init_pose[3 * b + 1] = w->w.bodies[b].y; init_pose[3 * b + 2] = w->w.bodies[b].a; }
    }
    return MGX_OK;
}
int mgx_world_n_state_entries(const mgx_world *w) {
    if (!w || !w->w.finalized) return fail(MGX_ERR_STATE, "world not finalized");
    return (int)w->w.state_map.size();
}
int mgx_world_state_entry(const mgx_world *w, int k, int *body, int *comp, int *row) {
    if (!w || !w->w.finalized) return fail(MGX_ERR_STATE, "world not finalized");
    if (k < 0 || k >= (int)w->w.state_map.size()) return fail(MGX_ERR_ARG, "state entry out of range");
    int m = w->w.state_map[k];
    if (comp) *comp = m & 15;
    if (body) *body = (m >> 4) & 0xFF;
    if (row) *row = m >> 12;
    return MGX_OK;
}
int mgx_world_goal_bb(const mgx_world *w, int ent, double bb[4]) {
    if (!w || !w->w.finalized) return fail(MGX_ERR_STATE, "world not finalized");
    if (ent < 0 || ent >= (int)w->w.entities.size() || w->w.entities[ent].kind != 2) return fail(MGX_ERR_ARG, "not a goal entity");
    const EntityDef &e = w->w.entities[ent];
    // GoalRegion.setup: body at (x + w/2, y - h/2), box (w, h)  (entities.py:794-797)
    double cx = e.x + e.w / 2, cy = e.y - e.h / 2, hw = e.w / 2, hh = e.h / 2;
    bb[0] = cx - hw; bb[1] = cy - hh; bb[2] = cx + hw; bb[3] = cy + hh;
    return MGX_OK;
}
int mgx_world_prim_table(const mgx_world *w, int *rgb, int *ent, int *role) {
    if (!w || !w->w.finalized) return fail(MGX_ERR_STATE, "world not finalized");
    int n = (int)w->w.prims.size();
    for (int k = 0; k < n; k++) {
        const PrimDef &P = w->w.prims[k];
        if (rgb) rgb[k] = P.rgb[0] | (P.rgb[1] << 8) | (P.rgb[2] << 16);
        if (ent) ent[k] = P.ent;
        if (role) role[k] = P.role;
    }
    return n;
}
int mgx_world_placement_collides(const mgx_world *w, int ent, const double *poses, const uint8_t *enabled, const double *ent_hw) {
    if (!w || !w->w.finalized) return fail(MGX_ERR_STATE, "world not finalized");
    if (ent < 0 || ent >= (int)w->w.entities.size() || !poses || !enabled) return fail(MGX_ERR_ARG, "bad entity / NULL argument");
    return w->w.placement_collides(ent, poses, enabled, ent_hw) ? 1 : 0;
}
int mgx_world_randomise_all_poses(const mgx_world *w, double *poses, const int *ents, int n, const uint8_t *ignore,
                                  const double arena_lrbt[4], const uint8_t *rand_pos, const uint8_t *rand_rot,
                                  const double *pos_limits, const double *rot_limits, uint32_t *mt_key, int *mt_pos, const double *ent_hw) {
    if (!w || !w->w.finalized) return fail(MGX_ERR_STATE, "world not finalized");
    if (!poses || !ents || n < 1 || !arena_lrbt || !rand_pos || !rand_rot || !pos_limits || !rot_limits || !mt_key || !mt_pos)
        return fail(MGX_ERR_ARG, "NULL argument");
    if (*mt_pos < 0 || *mt_pos > 624) return fail(MGX_ERR_ARG, "bad MT19937 position");
    for (int i = 0; i < n; i++) if (ents[i] < 0 || ents[i] >= (int)w->w.entities.size()) return fail(MGX_ERR_ARG, "entity index out of range");
    int rc = w->w.randomise_all_poses(poses, ents, n, ignore, arena_lrbt, rand_pos, rand_rot, pos_limits, rot_limits, mt_key, mt_pos, ent_hw);
    if (rc < 0) return fail(MGX_ERR_CAPACITY, "could not place the entities (PlacementError after 10 retries)");
    return rc;
}
}  // extern "C"

// pm_randomise_all_poses for m envs, env k in the world world_of(k): envs are independent (own stream, own poses), so
// they are spread over a few host threads
static int host_threads(bool allocating = true) {
    // world building and placement at a reset: as many threads as the host has cores, within [1, MGX_HOST_THREADS (default 64)].
    // History (256-core host, ClusterColour-TestAll, 4096 envs per reset): with a World built from hundreds of small allocations, blob
    // vectors per world and threads created per call, 32 was the best (41-62 ms at 16, 31-48 at 32, no better at 64, worse at 128 -- the
    // allocator; threads that only ran the placement left glibc arenas behind that slowed the NEXT reset's builds,
    // profiles/r04_reset_threads.txt).  Since the builds serialise into per-thread buffers and the pinned staging buffer, the block
    // geometry is cached per thread, the placement queries allocate nothing and the threads persist (HostPool below), more threads pay
    // again: 16.0 ms per reset at 32, 15.7 at 48, 12.2-13.8 at 64, 13.7 at 96, 18.1 at 128 (profiles/r04_reset_threads_pool.txt).
    // MGX_PLACE_THREADS (placement alone) still exists and defaults to MGX_HOST_THREADS; a pool larger than a burst wakes threads for nothing.
    static const int cap = [] { const char *v = getenv("MGX_HOST_THREADS"); const int c = v ? atoi(v) : 64; return c < 1 ? 1 : (c > 256 ? 256 : c); }();
    static const int cap_place = [] { const char *v = getenv("MGX_PLACE_THREADS"); const int c = v ? atoi(v) : cap; return c < 1 ? 1 : (c > 256 ? 256 : c); }();
    // (one process per GPU: the ranks of a node share its cores -- torchrun's LOCAL_WORLD_SIZE says how many there are)
    static const int ranks_here = [] { const char *v = getenv("LOCAL_WORLD_SIZE"); const int r = v ? atoi(v) : 1; return r < 1 ? 1 : r; }();
    const int hw = (int)std::thread::hardware_concurrency() / ranks_here, c = allocating ? cap : cap_place;
    return hw < 1 ? 1 : (hw > c ? c : hw);
}
// A few persistent host threads for the batch work of a reset (world builds, placement sampling): a reset of 4096 per-env worlds is
// two bursts of a few ms each, and creating and joining 32 threads per burst was a good part of them.  The threads keep their
// thread-local buffers (blob serialiser, shape geometry, held placement shapes) warm from reset to reset.  The pool is never torn
// down (its threads block on a condition variable of a leaked object: no static destructor to race with at exit); a forked child
// starts its own.  One burst at a time.
struct HostPool {
    std::mutex run_mu, mu; std::condition_variable cv_work, cv_done;
    std::vector<std::thread> threads;
    const std::function<void(int)> *job = nullptr; int n_job = 0, next = 0, pending = 0;
    pid_t pid = 0;
    void worker() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_work.wait(lk, [&] { return job && next < n_job; });
            const int t = next++;
            const std::function<void(int)> *fn = job;
            lk.unlock();
            (*fn)(t);
            lk.lock();
            if (--pending == 0) cv_done.notify_all();
        }
    }
    void run(int n, const std::function<void(int)> &fn) {      // fn(0 .. n-1), fn(0) on the caller
        if (n <= 1) { fn(0); return; }
        std::lock_guard<std::mutex> one(run_mu);
        {
            std::unique_lock<std::mutex> lk(mu);
            while ((int)threads.size() < n - 1) { threads.emplace_back([this] { worker(); }); threads.back().detach(); }
            job = &fn; n_job = n; next = 1; pending = n - 1;
        }
        cv_work.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
        job = nullptr; n_job = 0;
    }
};
static void host_parallel(int n, const std::function<void(int)> &fn) {
    static std::mutex mu; static HostPool *pool = nullptr;
    HostPool *p;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!pool || pool->pid != getpid()) { pool = new HostPool(); pool->pid = getpid(); }
        p = pool;
    }
    p->run(n, fn);
}
template <typename WorldOf>
static int randomise_batch(WorldOf world_of, int ne, int m, double *poses, const int *ents, int n, const uint8_t *ignore,
                           const double arena_lrbt[4], const uint8_t *rand_pos, const uint8_t *rand_rot,
                           const double *pos_limits, const double *rot_limits, int limits_per_env,
                           const uint64_t *mt_state_addr, const double *ent_hw) {
    int n_threads = host_threads(false);
    if (m < 64) n_threads = 1;
    std::vector<long> rej(n_threads, 0);
    std::vector<int> bad(n_threads, 0);
    std::atomic<int> next{0};                         // (envs differ in their number of rejected draws: chunks of 8 are handed out as threads come free)
    auto work = [&](int t) {
        for (int k0 = next.fetch_add(8); k0 < m; k0 = next.fetch_add(8))
        for (int k = k0; k < k0 + 8 && k < m; k++) {
            // numpy's mt19937_state: uint32 key[624]; int pos
            uint32_t *key = reinterpret_cast<uint32_t *>((uintptr_t)mt_state_addr[k]);
            int *pos = reinterpret_cast<int *>((uintptr_t)mt_state_addr[k] + 624 * sizeof(uint32_t));
            if (!key || *pos < 0 || *pos > 624) { bad[t] = 1; return; }
            const size_t lo = limits_per_env ? (size_t)k * n : 0;
            int rc = world_of(k).randomise_all_poses(poses + (size_t)k * ne * 3, ents, n, ignore, arena_lrbt, rand_pos, rand_rot, pos_limits + lo, rot_limits + lo, key, pos,
                                                     ent_hw ? ent_hw + (size_t)k * ne * 2 : nullptr);
            if (rc < 0) { bad[t] = 2; return; }
            rej[t] += rc;
        }
    };
    host_parallel(n_threads, work);
    long rejected = 0;
    for (int t = 0; t < n_threads; t++) {
        if (bad[t] == 1) return fail(MGX_ERR_ARG, "bad MT19937 state");
        if (bad[t] == 2) return fail(MGX_ERR_CAPACITY, "could not place the entities (PlacementError after 10 retries)");
        rejected += rej[t];
    }
    return (int)(rejected > 0x7fffffff ? 0x7fffffff : rejected);
}

extern "C" {

int mgx_world_randomise_all_poses_batch(const mgx_world *w, int m, double *poses, const int *ents, int n, const uint8_t *ignore,
                                        const double arena_lrbt[4], const uint8_t *rand_pos, const uint8_t *rand_rot,
                                        const double *pos_limits, const double *rot_limits, int limits_per_env,
                                        const uint64_t *mt_state_addr, const double *ent_hw) {
    if (!w || !w->w.finalized) return fail(MGX_ERR_STATE, "world not finalized");
    if (m < 0 || !poses || !ents || n < 1 || !arena_lrbt || !rand_pos || !rand_rot || !pos_limits || !rot_limits || !mt_state_addr)
        return fail(MGX_ERR_ARG, "NULL argument");
    const int ne = (int)w->w.entities.size();
    for (int i = 0; i < n; i++) if (ents[i] < 0 || ents[i] >= ne) return fail(MGX_ERR_ARG, "entity index out of range");
    return randomise_batch([&](int) -> const World & { return w->w; }, ne, m, poses, ents, n, ignore, arena_lrbt, rand_pos, rand_rot,
                           pos_limits, rot_limits, limits_per_env, mt_state_addr, ent_hw);
}
// batched RandomState draws: env k's stream at mt_state_addr[k] (numpy's mt19937_state: uint32 key[624]; int pos)
static bool mt_state_ok(uint64_t addr, uint32_t *&key, int *&pos) {
    key = reinterpret_cast<uint32_t *>((uintptr_t)addr);
    pos = reinterpret_cast<int *>((uintptr_t)addr + 624 * sizeof(uint32_t));
    return key && *pos >= 0 && *pos <= 624;
}
// (the m streams are independent and each call touches every env's 2.5 KB state once -- a cache miss per env: large batches are spread
// over the host pool; the addresses of one call must be distinct, one stream per env; per_env returns 0 or which argument was bad)
}  // extern "C"
template <typename F>
static int rng_batch(int m, F per_env) {
    if (m < 1024) {
        for (int k = 0; k < m; k++) if (int rc = per_env(k)) return rc;
        return 0;
    }
    const int nt = std::min(host_threads(false), 16);
    std::atomic<int> err{0};
    host_parallel(nt, [&](int t) {
        const int k0 = (int)((long)m * t / nt), k1 = (int)((long)m * (t + 1) / nt);
        for (int k = k0; k < k1 && err.load(std::memory_order_relaxed) == 0; k++) if (int rc = per_env(k)) { err.store(rc); return; }
    });
    return err.load();
}
static int rng_batch_status(int rc, const char *range_msg) {
    return rc == 0 ? (int)MGX_OK : fail(MGX_ERR_ARG, rc == 1 ? "bad MT19937 state" : range_msg);
}
extern "C" {
int mgx_rng_bounded_batch(int m, const uint64_t *mt_state_addr, const int32_t *counts, int count, int max_inclusive, int32_t *out, int out_stride) {
    if (m < 0 || !mt_state_addr || !out || max_inclusive < 0 || count < 0 || out_stride < count) return fail(MGX_ERR_ARG, "bad argument");
    return rng_batch_status(rng_batch(m, [&](int k) {
        uint32_t *key; int *pos;
        if (!mt_state_ok(mt_state_addr[k], key, pos)) return 1;
        const int n = counts ? counts[k] : count;
        if (n < 0 || n > out_stride) return 2;
        rng_bounded(key, pos, n, (uint32_t)max_inclusive, out + (size_t)k * out_stride);
        return 0;
    }), "count out of range");
}
int mgx_rng_doubles_batch(int m, const uint64_t *mt_state_addr, const int32_t *counts, int count, double *out, int out_stride) {
    if (m < 0 || !mt_state_addr || !out || count < 0 || out_stride < count) return fail(MGX_ERR_ARG, "bad argument");
    return rng_batch_status(rng_batch(m, [&](int k) {
        uint32_t *key; int *pos;
        if (!mt_state_ok(mt_state_addr[k], key, pos)) return 1;
        const int n = counts ? counts[k] : count;
        if (n < 0 || n > out_stride) return 2;
        rng_doubles(key, pos, n, out + (size_t)k * out_stride);
        return 0;
    }), "count out of range");
}
int mgx_rng_shuffle_batch(int m, const uint64_t *mt_state_addr, const int32_t *n_items, int32_t *perm, int perm_stride) {
    if (m < 0 || !mt_state_addr || !n_items || !perm) return fail(MGX_ERR_ARG, "bad argument");
    return rng_batch_status(rng_batch(m, [&](int k) {
        uint32_t *key; int *pos;
        if (!mt_state_ok(mt_state_addr[k], key, pos)) return 1;
        if (n_items[k] < 0 || n_items[k] > perm_stride) return 2;
        rng_shuffle(key, pos, n_items[k], perm + (size_t)k * perm_stride);
        return 0;
    }), "item count out of range");
}
int mgx_world_palette(int colour, int role) {
    if (colour < 0 || colour > 3 || role < 0 || role > 2) return fail(MGX_ERR_ARG, "colour 0..3, role 0..2");
    return palette_rgb(colour, role);
}
int mgx_world_entity_shapes(const mgx_world *w, int ent, int max_shapes, int *kinds, double *radii, int *nverts, double *xy, int xy_stride) {
    if (!w || !w->w.finalized) return fail(MGX_ERR_STATE, "world not finalized");
    if (ent < 0 || ent >= (int)w->w.entities.size()) return fail(MGX_ERR_ARG, "entity index out of range");
    const EntityDef &e = w->w.entities[ent];
    int n = 0;
    for (int s : e.shapes) {
        if (n >= max_shapes) break;
        const ShapeDef &S = w->w.shapes[s];
        if (kinds) kinds[n] = S.kind;
        if (radii) radii[n] = S.radius;
        if (nverts) nverts[n] = (int)S.verts.size();
        if (xy) for (size_t i = 0; i < S.verts.size() && (int)(2 * i + 1) < xy_stride; i++) { xy[n * xy_stride + 2 * i] = S.verts[i].x; xy[n * xy_stride + 2 * i + 1] = S.verts[i].y; }
        n++;
    }
    return n;
}

}  // extern "C"

// ------------------------------------------------------------------ engine
static int even(int x) { return (x + 1) & ~1; }
static const int HDR_WORDS = even((int)sizeof(TmplHeader) / 4);

// serialise a world into the two device blobs (host side) for the engine's precision
template <typename R, typename P>
static void make_blobs_t(const World &w, WorldBlobs &b) {
    // (the serialiser's buffers keep their capacity from world to world: a reset serialises thousands of worlds per thread)
    static thread_local std::vector<int32_t> iw, ir; static thread_local std::vector<double> rw, pw;
    {
        TmplHeader hs;
        w.serialise(hs, iw, rw, pw, /*strip_prims=*/true);
        const int off_i = HDR_WORDS, off_r = tmpl_off_r<R>(hs, off_i), off_p = tmpl_off_p<R, P>(hs, off_i), total = tmpl_total_words<R, P>(hs, off_i);
        b.step.assign(total, 0);
        std::memcpy(b.step.data(), &hs, sizeof(TmplHeader));
        std::memcpy(b.step.data() + off_i, iw.data(), iw.size() * 4);
        { R *d = reinterpret_cast<R *>(b.step.data() + off_r); for (size_t i = 0; i < rw.size(); i++) d[i] = (R)rw[i]; }
        { P *d = reinterpret_cast<P *>(b.step.data() + off_p); for (size_t i = 0; i < pw.size(); i++) d[i] = (P)pw[i]; }
        WorkOff wo(hs);
        b.step_off_r = off_r; b.step_off_p = off_p;
        b.step_env_off_r = even(wo.n_p * (int)(sizeof(P) / 4)); b.step_env_off_i = b.step_env_off_r + even(wo.n_r * (int)(sizeof(R) / 4));
        int stride = even(wo.n_p * (int)(sizeof(P) / 4)) + even(wo.n_r * (int)(sizeof(R) / 4)) + even(wo.n_i);
        while (stride % 32 != 2) stride += 2;     // envs of one wave start on distinct LDS banks
        b.step_env_stride = stride;
    }
    {
        w.serialise(b.h, iw, rw, pw);
        TmplOff o(b.h);
        // the rasteriser's copy keeps only what it reads: body tables, the draw list and the bodies' pose rows -- not the
        // shapes, joints, candidate pairs (hundreds to thousands of words in worlds with stars) and state map of the physics
        TmplHeader hr = b.h;
        hr.n_shapes = hr.n_verts = hr.n_joints = hr.n_pairs = hr.n_state = hr.n_islands = 0;
        TmplOff orr(hr);
        hr.n_words_i = orr.n_i;
        ir.assign(orr.n_i, 0);
        auto keep = [&](int from, int to, int n) { std::memcpy(ir.data() + to, iw.data() + from, (size_t)n * 4); };
        keep(o.body_type, orr.body_type, b.h.n_bodies); keep(o.body_parent, orr.body_parent, b.h.n_bodies); keep(o.body_ent, orr.body_ent, b.h.n_bodies);
        keep(o.prim_i, orr.prim_i, b.h.n_prims * PRIM_IWORDS); keep(o.pv_prim, orr.pv_prim, b.h.n_pverts); keep(o.body_prow, orr.body_prow, 3 * b.h.n_bodies);
        const int off_i = HDR_WORDS, off_q = raster_off_q(hr, off_i), nq = b.h.n_prims * PRIM_RWORDS + 2 * b.h.n_pverts;
        b.raster.assign(raster_blob_words(hr, off_i), 0);
        b.raster_lds_words = off_q;
        std::memcpy(b.raster.data(), &hr, sizeof(TmplHeader));
        std::memcpy(b.raster.data() + off_i, ir.data(), ir.size() * 4);
        std::memcpy(b.raster.data() + off_q, rw.data() + o.prim_r, (size_t)nq * 8);
        RasterOff ro(hr, false), roc(hr, true);
        b.raster_scratch_d = ro.n_d; b.raster_scratch_dc = roc.n_d;
        b.raster_n_i = ro.n_i;
    }
}
static void make_blobs(int dtype, const World &w, WorldBlobs &b) {
    if (dtype == MGX_F32) make_blobs_t<float, double>(w, b);
    else if (dtype == MGX_F64) make_blobs_t<double, double>(w, b);
    else make_blobs_t<float, float>(w, b);
}

static size_t step_lds_bytes(const mgx_engine *e, int L) { return (size_t)(e->tdev.lds_tmpl_words + (64 / L) * e->tdev.env_stride_words) * 4; }
// size the launch geometry (lanes per env, LDS, raster variant) for blobs of the given sizes
// (scratch_d and raster_n_i may be maxima taken from different worlds: the int region starts after the LARGEST double region,
// so the tile / queue area has to start after the largest double region plus the largest int region)
static int configure_launch(mgx_engine *e, int step_words, int step_env_stride, int raster_lds_words, int raster_full_words, int scratch_d_stored, int scratch_d_compact, int raster_n_i) {
    // per-tile (u64 mask + i32 base) + queue (u64 mask + 2 x i32) + counters + overflow bitmap + phase E records (u64 sums, u16 entry)
    // (the pixel queue is 16 B per entry: the smallest worlds -- no goal region, whose dashed outline alone queues ~200 pixels, and at
    // most one block -- never come near 1024 entries and take the 640-entry layout: 30 -> 24 KB in MoveToCorner, which is what lets a
    // THIRD rasteriser workgroup sit beside a CU's four step workgroups in the fused env-step; a frame that does overflow takes
    // another round like any other)
    int n_goals_w = 0, n_blocks_w = 0;
    for (const auto &en : e->w.entities) { n_goals_w += en.kind == 2; n_blocks_w += en.kind == 1; }
    // primitive sets as 32-bit words where no world the engine holds has more than 32 primitives (every Demo / Jitter / Colour / Layout /
    // Dynamics world; per-env worlds are sized by their capacity world and stay at 64): half the words of the pixel queue and of the
    // per-tile sets -- 8 KB of ClusterColour's 40 KB, a FIFTH rasteriser workgroup per CU for the mid-size worlds
    // (per-env worlds, round 6: by their CAPACITY world's draw list -- MoveToCorner / MakeLine / FixColour / FindDupe Test* worlds have <= 32 primitives too;
    // round 5 kept every per-env engine at 64-bit sets)
    const bool narrow = (e->env_worlds ? e->cap_prims : e->h.n_prims) <= 32 && !getenv("MGX_RASTER_WIDE") && !(e->env_worlds && getenv("MGX_RASTER_WIDE_ENV"));
    e->rdev.narrow = narrow ? 1 : 0;
    const int mw = narrow ? 1 : 2;      // words per primitive set
    const int qcap_lds = (narrow && n_goals_w == 0 && n_blocks_w <= 1 && !getenv("MGX_QCAP_FULL")) ? QCAP_SMALL : QCAP;
    e->rdev.qcap_lds = qcap_lds;
    // phase T's wavefronts ahead of the CU's other rasteriser wavefronts: every world but the smallest ones (those of the 640-entry queue:
    // MoveToCorner -0.4 %, the others +1.1 ... 2.7 %; MGX_RASTER_PRIO=0 / 1: never / always)
    { const char *pv = getenv("MGX_RASTER_PRIO"); e->rdev.prio_t = pv ? (atoi(pv) != 0) : (qcap_lds != QCAP_SMALL); }
    // (the layout may grow when per-env worlds are enabled: the entries in use follow it, unless a test has set them)
    e->rdev.qcap = e->qcap_debug > 0 && e->qcap_debug < qcap_lds ? e->qcap_debug : qcap_lds;
    // per-tile (set + i32 base colour + u8 base index) + queue (set + i32 position | base index) + counters + overflow bitmap + phase E
    // records (u64 sums, u16 entry); phase C's partial verdicts (4 x 144 x 2 mw words) lie in the queue's arrays
    const int extra = N_TILES * (mw + 1) + N_TILES / 4 + qcap_lds * (mw + 1) + 8 + OVF_WORDS + ECAP * 2 + ECAP / 2;
    static_assert(4 * N_TILES * 2 <= QCAP_SMALL * 2 && 4 * N_TILES * 4 <= QCAP * 3, "phase C's partial verdicts fit the pixel queue's arrays");
    // As many rasteriser workgroups per CU as LDS allows (allocations round up to 1280 B, lds_alloc_bytes), between 3 and 5; a step is worth 13-15 % of
    // the launch.  Two economies are taken only where they buy such a step, the cheaper one first:
    //  - the draw list's fp64 part (local vertices, radii: read once per frame, by the set-up) stays in HBM instead of being staged
    //    with the header and ints (4 KB in ClusterColour; FixColour 36.4 -> 31.4 KB: five per CU, env-step 1.19 -> 1.12 ms; the per-env
    //    worlds of FixColour / MakeLine: four; where it buys nothing it costs the set-up its LDS reads: MoveToCorner -1 %);
    //  - the draw-list vertex records drop their edge-function coefficients (RasterOff): rebuilding them costs the exact tests a dozen
    //    fp64 operations per edge (MatchRegions' env-step 1.10 -> 1.17 ms when forced; ClusterColour 45.9 -> 40.6 KB: 1.59 -> 1.48 ms)
    auto raster_fit = [&](bool tq_hbm, bool cmp) {
        const size_t lds = (size_t)(even(tq_hbm ? raster_lds_words : raster_full_words) + even(2 * (cmp ? scratch_d_compact : scratch_d_stored) + raster_n_i) + extra) * 4;
        const int fit = (int)((size_t)MAX_LDS_BYTES / lds_alloc_bytes(lds));
        return fit > 5 ? 5 : fit;
    };
    // (rounds 2-4 kept worlds with multi-part polygons at four workgroups per CU: the 96-register variant's coverage code was slow on
    // stars then.  With the two-pass coverage in every variant and 32-bit primitive sets the fifth workgroup pays there too:
    // ClusterColour 3.99 -> 4.21 M env-steps/s, ClusterShape 3.90 -> 4.11, MakeLine 5.16 -> 5.59 (MGX_RASTER_CAP4=1: the old rule))
    bool stars = e->env_worlds;
    for (const auto &pr : e->w.prims) stars = stars || pr.parts.size() > 1;
    const int cap = (stars && getenv("MGX_RASTER_CAP4")) ? 4 : 5;
    bool tq_hbm = false, compact = false;
    if (!getenv("MGX_RASTER_STORED")) {
        int best = raster_fit(false, false);
        const bool opts[3][2] = {{true, false}, {false, true}, {true, true}};
        for (auto &o : opts) {
            const int f = raster_fit(o[0], o[1]) < cap ? raster_fit(o[0], o[1]) : cap;
            if (f > best) { best = f; tq_hbm = o[0]; compact = o[1]; }
        }
    }
    const int scratch_d = compact ? scratch_d_compact : scratch_d_stored, raster_words = tq_hbm ? raster_lds_words : raster_full_words;
    e->rdev.compact = compact ? 1 : 0; e->rdev.tq_hbm = tq_hbm ? 1 : 0;
    const int off_tiles = even(2 * scratch_d + raster_n_i);
    e->tdev.off_i = HDR_WORDS; e->tdev.lds_tmpl_words = even(step_words); e->tdev.env_stride_words = step_env_stride;
    // lanes per env: caller's choice, else the widest group (most narrowphase parallelism) that still lets
    // two workgroups share a CU's LDS; worlds too big for that take the narrowest group that fits at all
    // (-1: the engine's choice for an engine whose env-steps are rendered -- the fused env-step, mgx_engine_step_render)
    const bool rendered_hint = e->L_request == -1;
    int L = rendered_hint ? 0 : e->L_request;
    if (L == 0 && !e->env_worlds) {
        L = 16;
        while (L < 64 && step_lds_bytes(e, L) > (size_t)MAX_LDS_BYTES / 2) L *= 2;
        // the crowded worlds (ClusterColour / ClusterShape: 305 candidate pairs, 27 shapes) whose 16-lane working sets cannot all be
        // resident anyway (54 KB: three workgroups per CU) take 32 lanes per env: the broadphase and narrowphase, a third of their
        // step, run twice as wide.  Measured at 4096 envs: ClusterColour k_step 0.88 -> 0.77 ms, env-step 1.80 -> 1.66 ms; the
        // smaller worlds lose (FindDupe 1.48 -> 1.68 ms, MatchRegions 1.05 -> 1.31, MakeLine 1.16 -> 1.41, FixColour 1.12 -> 1.41)
        // ... on its own.  In the FUSED env-step the narrow groups win there too (fewer, longer step workgroups leave the rasteriser more
        // of the CU): ClusterColour 4.53 -> 4.66 M env-steps/s, ClusterShape 4.40 -> 4.55 with 16 lanes although k_step alone is 12 %
        // slower (round 5, tools/dev/lanes_ab.sh) -- so the rule applies to engines that are not rendered (lanes_per_env 0), not to -1
        // (Since the working set shrank -- manifold slots, packed flags: WorkOff, mgx_tmpl.h -- the reference's crowded worlds are 40.7 KB at
        // 16 lanes, four workgroups per CU and every step workgroup resident at once: neither rule fires for them any more, and 16 lanes
        // are faster for k_step alone too: ClusterColour state-only 8.74 -> 9.76 M env-steps/s.  The rules stay for bigger worlds.)
        if (L == 16 && !rendered_hint && e->h.n_pairs > 256 && step_lds_bytes(e, 16) > (size_t)40 * 1024) L = 32;
        // FindDupe's worlds (45 KB at 16 lanes: three workgroups per CU; <= 256 candidate pairs): 32 lanes either way -- k_step alone
        // 0.36 -> 0.30 ms, fused env-step 4.73 -> 4.80 M env-steps/s (round 5; in round 2 it was the other way round)
        else if (L == 16 && e->h.n_pairs <= 256 && step_lds_bytes(e, 16) > (size_t)40 * 1024) L = 32;
    } else if (L == 0) {
        L = 64;      // per-env templates: one env per wavefront, so that its template copy in LDS is shared by all lanes
    }
    // (a group is one or more whole DPP rows: the solver keeps robot joint j on lane j of every row, mgx_sim.h)
    if (L != 16 && L != 32 && L != 64) return fail(MGX_ERR_ARG, "lanes_per_env must be 0, -1, 16, 32 or 64");
    if (e->h.n_islands > 15) return fail(MGX_ERR_CAPACITY, "more than 15 blocks: every block's joints need a lane of the group's first row");
    if (step_lds_bytes(e, L) > (size_t)MAX_LDS_BYTES) return fail(MGX_ERR_CAPACITY, "world working set does not fit LDS at this lanes_per_env");
    if (getenv("MGX_DEBUG_LAUNCH"))
        fprintf(stderr, "mgx: lanes_per_env %d, k_step LDS bytes at 16 / 32 / 64 lanes: %zu / %zu / %zu\n", L,
                step_lds_bytes(e, 16), step_lds_bytes(e, 32), step_lds_bytes(e, 64));
    if (getenv("MGX_DEBUG_LAUNCH")) {
        const WorkOff wo(e->h);
        fprintf(stderr, "mgx: bodies %d verts %d shapes %d joints %d contacts %d overlaps %d cache %d pairs %d; working set words: P %d (x%d) R %d I %d; template %d words, env stride %d words\n",
                e->h.n_bodies, e->h.n_verts, e->h.n_shapes, e->h.n_joints, e->h.max_contacts, e->h.max_overlaps, e->h.cache_slots, e->h.n_pairs,
                wo.n_p, 2, wo.n_r, wo.n_i, e->tdev.lds_tmpl_words, e->tdev.env_stride_words);
    }
    e->L = L; e->lds_step = step_lds_bytes(e, L);
    e->rdev.off_i = HDR_WORDS; e->rdev.lds_tmpl_words = even(raster_words); e->rdev.scratch_d = scratch_d; e->rdev.off_tiles = off_tiles;
    e->lds_raster = (size_t)(e->rdev.lds_tmpl_words + off_tiles + extra) * 4;
    int fit = (int)((size_t)MAX_LDS_BYTES / lds_alloc_bytes(e->lds_raster));
    e->raster_waves = fit >= 5 ? 5 : (fit == 4 ? 4 : 3);
    if (getenv("MGX_DEBUG_LAUNCH"))
        fprintf(stderr, "mgx: k_raster LDS bytes %zu (draw list %d words, per-env scratch %d words%s, tiles / queues %d words): %d workgroups per CU\n",
                e->lds_raster, e->rdev.lds_tmpl_words, off_tiles, compact ? (tq_hbm ? " [compact vertex records, fp64 part in HBM]" : " [compact vertex records]") : (tq_hbm ? " [fp64 part in HBM]" : ""), extra, e->raster_waves);
    return MGX_OK;
}

static long step_slots(mgx_engine *e);
template <typename R, typename P, int L>
static int launch_step_L(mgx_engine *e, void *sp, void *sf, int32_t *si, const int32_t *actions, uint8_t *done, int n_sub,
                         int count_step, hipStream_t st, const StepHandoff &ho) {
    void (*kern)(TmplDev, P *, R *, int32_t *, const int32_t *, uint8_t *, int, int, int, int, StepHandoff);
    if constexpr (L == 64) kern = k_step_env<R, P>; else if constexpr (sizeof(R) == 8) kern = k_step_wide<R, P, L>; else kern = k_step<R, P, L>;
    size_t lds = step_lds_bytes(e, L);
    if (int rc = ensure_lds((const void *)kern, lds, e->device)) return rc;
    int epb = 64 / L, blocks = (e->n_envs + epb - 1) / epb;
    // More step workgroups than the chip holds at once (LDS: 160 KB per CU; registers: one wavefront per SIMD, two for the
    // one-env-per-wavefront instantiation): they are dispatched longest first, by the durations of the previous launch.
    // Measured at 4096 envs: ClusterColour (2048 workgroups on 1024 slots) k_step 0.89 -> 0.6x ms.
    const long slots = step_slots(e);
    // (in the fused env-step, where a wavefront holds several envs, the env order below does the same job -- costliest envs first -- and
    // more: ClusterColour 4.65 -> 4.88 M env-steps/s, ClusterShape 4.55 -> 4.77 with the envs packed instead of the workgroups
    // re-ordered, although k_step alone is 12 % slower that way; round 5, tools/dev/ab_env.sh)
    static const bool no_pack_env = getenv("MGX_NO_ENV_PACK") != nullptr;
    const bool pack_wanted = 64 / L > 1 && count_step && !no_pack_env && ho.queue != nullptr && e->n_envs >= 8 * (64 / L);
    const bool lpt = blocks > slots && blocks <= (1 << 20) && !getenv("MGX_NO_LPT") && !pack_wanted;
    TmplDev t = e->tdev;
    t.order = nullptr; t.dur = nullptr;
    if (lpt) {
        if (e->order_cap != blocks) {
            if (e->d_dur) (void)hipFree(e->d_dur);
            if (e->d_order) (void)hipFree(e->d_order);
            e->d_dur = e->d_order = nullptr; e->order_cap = 0; e->order_valid = false;
            HIP_OK(hipMalloc(&e->d_dur, (size_t)blocks * 4)); HIP_OK(hipMalloc(&e->d_order, (size_t)blocks * 4));
            e->order_cap = blocks;
        }
        t.dur = e->d_dur;
        t.order = e->order_valid ? e->d_order : nullptr;
    }
    // Heavy envs together: where every step workgroup is resident at once and a wavefront holds several envs, the launch's envs are
    // ordered by the cost keys the previous launch left (contact points, overlapping pairs), costliest first -- envs in contact share
    // wavefronts, and most CUs hold light wavefronts only and hand their SIMDs to the rasteriser early (MGX_NO_ENV_PACK=1: off)
    // (fused env-step only: on its own the step kernel gains nothing -- its longest wavefront gets longer -- and the sort would sit
    // between two step launches: state-only MoveToCorner 15.5 -> 12.6 M env-steps/s when it was tried there)
    const bool pack = !lpt && epb > 1 && count_step && !no_pack_env && ho.queue != nullptr && e->n_envs >= 8 * epb;
    t.env_order = nullptr; t.env_cost = nullptr;
    if (pack) {
        if (!e->d_env_cost) {
            HIP_OK(hipMalloc(&e->d_env_cost, (size_t)e->n_envs * 4)); HIP_OK(hipMalloc(&e->d_env_order, (size_t)e->n_envs * 4));
            e->env_order_valid = false;
        }
        t.env_cost = e->d_env_cost;
        t.env_order = e->env_order_valid ? e->d_env_order : nullptr;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, st, t, (P *)sp, (R *)sf, si, actions, done, e->n_envs, n_sub,
                       count_step, e->dbg_iterations >= 0 ? e->dbg_iterations : PHYS_ITER, ho);
    HIP_OK(hipGetLastError());
    e->env_sort_pending = pack;       // (step_common enqueues k_env_order behind the kernel, outside the timing events around it)
    if (lpt) {
        hipLaunchKernelGGL(k_step_order, dim3(1), dim3(1024), 0, st, (const uint32_t *)e->d_dur, e->d_order, blocks);
        HIP_OK(hipGetLastError());
        e->order_valid = true;
    }
    return MGX_OK;
}
static int step_blocks(const mgx_engine *e) { const int epb = 64 / e->L; return (e->n_envs + epb - 1) / epb; }
// step workgroups the chip holds at once (LDS: 160 KB per CU; registers: one wavefront per SIMD, two for the one-env-per-wavefront
// instantiation)
static long step_slots(mgx_engine *e) {
    if (!e->n_cus) { if (hipDeviceGetAttribute(&e->n_cus, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess || e->n_cus <= 0) e->n_cus = 256; }
    const size_t lds = step_lds_bytes(e, e->L);
    const int by_lds = (int)((size_t)MAX_LDS_BYTES / lds_alloc_bytes(lds)), by_regs = 4 * (e->L == 64 ? MGX_L64_WAVES : MGX_LN_WAVES);
    return (long)e->n_cus * (by_lds < by_regs ? by_lds : by_regs);
}
template <typename R, typename P>
static int launch_step(mgx_engine *e, void *sp, void *sf, int32_t *si, const int32_t *actions, uint8_t *done, int n_sub,
                       int count_step, hipStream_t st, const StepHandoff &ho = StepHandoff{}) {
    if (e->env_worlds && e->L != 64) return fail(MGX_ERR_ARG, "per-env worlds run one env per wavefront (lanes_per_env 64)");
    switch (e->L) {
        case 16: return launch_step_L<R, P, 16>(e, sp, sf, si, actions, done, n_sub, count_step, st, ho);
        case 32: return launch_step_L<R, P, 32>(e, sp, sf, si, actions, done, n_sub, count_step, st, ho);
        case 64: return launch_step_L<R, P, 64>(e, sp, sf, si, actions, done, n_sub, count_step, st, ho);
    }
    return fail(MGX_ERR_ARG, "lanes_per_env must be 16, 32 or 64");
}

// k_score's constant tables: the block shapes of every shape type in fp64 (from variants of the engine's world in which every
// block has that type: all blocks share SHAPE_RAD, entities.py:614-711), the entity list, the bodies' pose rows, the goals
static int build_score_tables(mgx_engine *e) {
    const World &w = e->w;
    const int ne = (int)w.entities.size();
    std::vector<ScoreLib> lib(1);
    std::memset(&lib[0], 0, sizeof(ScoreLib));
    int first_block = -1;
    for (int i = 0; i < ne; i++) if (w.entities[i].kind == 1) { first_block = i; break; }
    if (first_block >= 0) {
        for (int t = 0; t < SC_TYPES; t++) {
            std::vector<int> types(ne, -1);
            for (int i = 0; i < ne; i++) if (w.entities[i].kind == 1) types[i] = t;
            std::vector<uint8_t> on(ne, 1);
            World v; std::string err;
            int rc = w.variant(on.data(), types.data(), v, err);
            if (rc) return fail(rc == -2 ? MGX_ERR_CAPACITY : MGX_ERR_ARG, "score library: " + err);
            const EntityDef &E = v.entities[first_block];
            if ((int)E.shapes.size() > SC_MAX_PARTS) return fail(MGX_ERR_CAPACITY, "score library: block with more than 8 collision shapes");
            lib[0].n_parts[t] = (int)E.shapes.size();
            for (size_t p = 0; p < E.shapes.size(); p++) {
                const ShapeDef &S = v.shapes[E.shapes[p]];
                if ((int)S.verts.size() > SC_MAX_VERTS) return fail(MGX_ERR_CAPACITY, "score library: collision polygon with more than 8 vertices");
                lib[0].kind[t][p] = S.kind; lib[0].nv[t][p] = (int)S.verts.size(); lib[0].radius[t][p] = S.radius;
                for (size_t k = 0; k < S.verts.size(); k++) { lib[0].xy[t][p][2 * k] = S.verts[k].x; lib[0].xy[t][p][2 * k + 1] = S.verts[k].y; }
            }
        }
    }
    std::vector<int32_t> ent(4 * (size_t)ne), prow(3 * w.bodies.size(), -1), goal_ent;
    std::vector<double> goal_xyhw;
    for (int i = 0; i < ne; i++) {
        const EntityDef &E = w.entities[i];
        ent[4 * i] = E.kind; ent[4 * i + 1] = E.body; ent[4 * i + 2] = E.kind == 1 ? E.shape_type : -1; ent[4 * i + 3] = E.enabled ? 1 : 0;
        if (E.kind == 2) { goal_ent.push_back(i); goal_xyhw.insert(goal_xyhw.end(), {E.x, E.y, E.h, E.w}); }
    }
    for (int m : w.state_map) { const int comp = m & 15, b = (m >> 4) & 0xFF, row = m >> 12; if (comp < 3) prow[3 * b + comp] = row; }
    e->n_goals = (int)goal_ent.size();
    auto up = [&](auto **dst, const void *src, size_t bytes) -> bool {
        if (bytes == 0) { *dst = nullptr; return true; }
        return hipMalloc(reinterpret_cast<void **>(dst), bytes) == hipSuccess && hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
    if (!up(&e->d_score_lib, lib.data(), sizeof(ScoreLib)) || !up(&e->d_score_ent, ent.data(), ent.size() * 4) || !up(&e->d_score_prow, prow.data(), prow.size() * 4) ||
        !up(&e->d_score_goal_ent, goal_ent.data(), goal_ent.size() * 4) || !up(&e->d_score_goal_xyhw, goal_xyhw.data(), goal_xyhw.size() * 8))
        return fail(MGX_ERR_HIP, "score table upload failed");
    return MGX_OK;
}

static bool timing_this_launch(const mgx_engine *e, int which) { return e->timing > 0 && e->launch_count[which] % e->timing == 0; }
static int timing_begin(mgx_engine *e, int which, hipStream_t st) {
    if (!timing_this_launch(e, which)) return MGX_OK;
    if (e->ev[which].empty()) {
        e->ev[which].resize(2 * TIMING_RING);
        for (auto &ev : e->ev[which]) HIP_OK(hipEventCreate(&ev));
    }
    int slot = e->ev_count[which] % TIMING_RING;
    HIP_OK(hipEventRecord(e->ev[which][2 * slot], st));
    return MGX_OK;
}
static int timing_end(mgx_engine *e, int which, hipStream_t st) {
    if (!timing_this_launch(e, which)) { e->launch_count[which]++; return MGX_OK; }
    e->launch_count[which]++;
    int slot = e->ev_count[which] % TIMING_RING;
    HIP_OK(hipEventRecord(e->ev[which][2 * slot + 1], st));
    e->ev_count[which]++;
    return MGX_OK;
}

extern "C" {

int mgx_engine_create(const mgx_world *w, int n_envs, int device, int dtype, int lanes_per_env, mgx_engine **out) {
    if (!w || !out) return fail(MGX_ERR_ARG, "NULL argument");
    if (!w->w.finalized) return fail(MGX_ERR_STATE, "world not finalized");
    if (n_envs <= 0) return fail(MGX_ERR_ARG, "n_envs must be positive");
    if (dtype != MGX_F32 && dtype != MGX_F64 && dtype != MGX_F32_PURE) return fail(MGX_ERR_ARG, "bad dtype");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MGX_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(MGX_ERR_ARG, "device index out of range");
    DeviceGuard guard__(device);
    if (!guard__.ok) return fail(MGX_ERR_HIP, "hipSetDevice failed");
    mgx_engine *e = new mgx_engine();
    e->w = w->w; e->n_envs = n_envs; e->device = device; e->dtype = dtype; e->L_request = lanes_per_env;
    WorldBlobs b;
    make_blobs(dtype, e->w, b);
    e->h = b.h;
    e->rows_p = state_rows_p(b.h); e->rows_f = state_rows_f(b.h); e->rows_i = state_rows_i(b.h);
    int rc = configure_launch(e, (int)b.step.size(), b.step_env_stride, b.raster_lds_words, (int)b.raster.size(), b.raster_scratch_d, b.raster_scratch_dc, b.raster_n_i);
    if (rc) { delete e; return rc; }
    int32_t pal[12];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) pal[4 * r + c] = palette_rgb(c, r);
    if (hipMalloc(&e->d_step, b.step.size() * 4) != hipSuccess || hipMemcpy(e->d_step, b.step.data(), b.step.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc(&e->d_raster, b.raster.size() * 4) != hipSuccess || hipMemcpy(e->d_raster, b.raster.data(), b.raster.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc(&e->d_palette, sizeof(pal)) != hipSuccess || hipMemcpy(e->d_palette, pal, sizeof(pal), hipMemcpyHostToDevice) != hipSuccess) {
        mgx_engine_destroy(e); return fail(MGX_ERR_HIP, "template upload failed");
    }
    e->tdev.words = e->d_step; e->tdev.n_words = (int)b.step.size(); e->tdev.tmpl_stride_words = 0;
    e->tdev.off_r = b.step_off_r; e->tdev.off_p = b.step_off_p; e->tdev.env_off_r = b.step_env_off_r; e->tdev.env_off_i = b.step_env_off_i;
    e->rdev.words = e->d_raster; e->rdev.n_words = (int)b.raster.size(); e->rdev.tmpl_stride_words = 0;
    e->rdev.bg_rgb = BG_RGB; e->rdev.qcap = e->rdev.qcap_lds; e->rdev.ecap = ECAP; e->rdev.palette = e->d_palette;
    rc = build_score_tables(e);
    if (rc) { mgx_engine_destroy(e); return rc; }
    *out = e;
    return MGX_OK;
}
void mgx_engine_destroy(mgx_engine *e) {
    if (!e) return;
    DeviceGuard guard__(e->device);
    if (e->st2) { (void)hipStreamSynchronize(e->st2); (void)hipStreamDestroy(e->st2); }
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->peek_stream) { (void)hipStreamSynchronize(e->peek_stream); (void)hipStreamDestroy(e->peek_stream); }
    if (e->peek_q) (void)hipHostFree(e->peek_q);
    if (e->peek_h) (void)hipHostFree(e->peek_h);
    for (void *p : {(void *)e->d_queue, (void *)e->d_hand, (void *)e->d_deferred, (void *)e->d_dur, (void *)e->d_order, (void *)e->d_env_cost, (void *)e->d_env_order}) if (p) (void)hipFree(p);
    for (void *p : {(void *)e->d_score_lib, (void *)e->d_score_ent, (void *)e->d_score_prow, (void *)e->d_score_goal_ent,
                    (void *)e->d_score_goal_xyhw, (void *)e->d_ent_type_env, (void *)e->d_ent_present_env})
        if (p) (void)hipFree(p);
    if (e->d_step) (void)hipFree(e->d_step);
    if (e->d_raster) (void)hipFree(e->d_raster);
    if (e->d_palette) (void)hipFree(e->d_palette);
    if (e->variants_pending) (void)hipEventSynchronize(e->ev_variants);
    if (e->ev_variants) (void)hipEventDestroy(e->ev_variants);
    for (void *p : {(void *)e->d_v_ty, (void *)e->d_v_on, (void *)e->d_v_idx}) if (p) (void)hipFree(p);
    if (e->d_stage) (void)hipFree(e->d_stage);
    if (e->h_stage) (void)hipHostFree(e->h_stage);
    if (e->d_stage_idx) (void)hipFree(e->d_stage_idx);
    for (int k = 0; k < 2; k++) for (auto &ev : e->ev[k]) (void)hipEventDestroy(ev);
    delete e;
}
int mgx_engine_state_shape(const mgx_engine *e, int *rows_p, int *rows_f, int *rows_i, int *size_p, int *size_f) {
    if (!e) return fail(MGX_ERR_ARG, "engine is NULL");
    if (rows_p) *rows_p = e->rows_p;
    if (rows_f) *rows_f = e->rows_f;
    if (rows_i) *rows_i = e->rows_i;
    if (size_p) *size_p = e->dtype == MGX_F32_PURE ? 4 : 8;
    if (size_f) *size_f = e->dtype == MGX_F64 ? 8 : 4;
    return MGX_OK;
}
int mgx_engine_lanes_per_env(const mgx_engine *e) { return e ? e->L : 0; }
int mgx_engine_lds_bytes(const mgx_engine *e, int which) { return e ? (int)(which == 0 ? e->lds_step : e->lds_raster) : 0; }

static int reset_common(mgx_engine *e, void *state_p, void *state_f, int32_t *state_i, const uint8_t *mask, const void *ent_pose, void *stream) {
    if (!e || !state_p || !state_f || !state_i) return fail(MGX_ERR_ARG, "NULL argument");
    ON_DEVICE(e);
    hipStream_t st = (hipStream_t)stream;
    size_t lds = e->env_worlds ? 0 : (size_t)e->tdev.n_words * 4;
    int blocks = (e->n_envs + 63) / 64;
    {
        int rc = e->dtype == MGX_F32 ? ensure_lds((const void *)k_reset<float, double>, lds, e->device)
               : e->dtype == MGX_F64 ? ensure_lds((const void *)k_reset<double, double>, lds, e->device)
                                     : ensure_lds((const void *)k_reset<float, float>, lds, e->device);
        if (rc) return rc;
    }
    if (e->dtype == MGX_F32) hipLaunchKernelGGL((k_reset<float, double>), dim3(blocks), dim3(64), lds, st, e->tdev, (double *)state_p, (float *)state_f, state_i, mask, (const double *)ent_pose, e->n_envs);
    else if (e->dtype == MGX_F64) hipLaunchKernelGGL((k_reset<double, double>), dim3(blocks), dim3(64), lds, st, e->tdev, (double *)state_p, (double *)state_f, state_i, mask, (const double *)ent_pose, e->n_envs);
    else hipLaunchKernelGGL((k_reset<float, float>), dim3(blocks), dim3(64), lds, st, e->tdev, (float *)state_p, (float *)state_f, state_i, mask, (const float *)ent_pose, e->n_envs);
    HIP_OK(hipGetLastError());
    return MGX_OK;
}
int mgx_engine_reset(mgx_engine *e, void *state_p, void *state_f, int32_t *state_i, const uint8_t *mask, void *stream) {
    return reset_common(e, state_p, state_f, state_i, mask, nullptr, stream);
}
int mgx_engine_reset_poses(mgx_engine *e, void *state_p, void *state_f, int32_t *state_i, const uint8_t *mask, const void *ent_pose, void *stream) {
    if (!ent_pose) return fail(MGX_ERR_ARG, "ent_pose is NULL (use mgx_engine_reset for the template poses)");
    return reset_common(e, state_p, state_f, state_i, mask, ent_pose, stream);
}
static int step_common(mgx_engine *e, void *sp, void *sf, int32_t *si, const int32_t *actions, uint8_t *done, int n_sub, int count_step, void *stream,
                       const StepHandoff &ho = StepHandoff{}) {
    if (!e || !sp || !sf || !si || !actions) return fail(MGX_ERR_ARG, "NULL argument");
    if (n_sub < 0) return fail(MGX_ERR_ARG, "negative substep count");
    ON_DEVICE(e);
    hipStream_t st = (hipStream_t)stream;
    int rc = timing_begin(e, 0, st);
    if (rc) return rc;
    rc = e->dtype == MGX_F32 ? launch_step<float, double>(e, sp, sf, si, actions, done, n_sub, count_step, st, ho)
       : e->dtype == MGX_F64 ? launch_step<double, double>(e, sp, sf, si, actions, done, n_sub, count_step, st, ho)
                             : launch_step<float, float>(e, sp, sf, si, actions, done, n_sub, count_step, st, ho);
    if (rc) return rc;
    rc = timing_end(e, 0, st);
    if (rc) return rc;
    if (e->env_sort_pending) {
        // next launch's env order from this launch's cost keys: one small workgroup behind the step kernel (it runs under the rasterisation)
        hipLaunchKernelGGL(k_env_order, dim3(1), dim3(1024), 0, st, (const uint32_t *)e->d_env_cost, e->d_env_order, e->n_envs);
        HIP_OK(hipGetLastError());
        e->env_order_valid = true; e->env_sort_pending = false;
    }
    return MGX_OK;
}
int mgx_engine_step(mgx_engine *e, void *state_p, void *state_f, int32_t *state_i, const int32_t *actions, uint8_t *done, void *stream) {
    return step_common(e, state_p, state_f, state_i, actions, done, PHYS_STEPS, 1, stream);
}
int mgx_engine_substeps(mgx_engine *e, void *state_p, void *state_f, int32_t *state_i, const int32_t *actions, int n_substeps, void *stream) {
    return step_common(e, state_p, state_f, state_i, actions, nullptr, n_substeps, 0, stream);
}

}  // extern "C"

// what both rasteriser entry points need of the world(s) currently loaded
static int raster_capacity_ok(const mgx_engine *e) {
    if (e->h.n_prims > 64) return fail(MGX_ERR_CAPACITY, "draw list longer than 64 primitives");
    if (e->lds_raster > (size_t)MAX_LDS_BYTES) return fail(MGX_ERR_CAPACITY, "draw list does not fit LDS");
    return MGX_OK;
}
template <typename P>
static int launch_raster(mgx_engine *e, const void *sp, uint8_t *out, int64_t env_stride, int view, int layout, const uint8_t *fill, hipStream_t st,
                         const RasterHandoff &ho = RasterHandoff{}, hipEvent_t done = nullptr) {
    size_t lds = e->lds_raster;
    auto go = [&](auto kern) -> int {
        if (int rc = ensure_lds((const void *)kern, lds, e->device)) return rc;
        // done: an event that completes with the kernel (its own completion signal: no marker packet behind it for the join to wait for)
        if (done) hipExtLaunchKernelGGL(kern, dim3(e->n_envs), dim3(256), lds, st, nullptr, done, 0, e->rdev, (const P *)sp, out, (long)env_stride, view, fill, e->n_envs, ho);
        else hipLaunchKernelGGL(kern, dim3(e->n_envs), dim3(256), lds, st, e->rdev, (const P *)sp, out, (long)env_stride, view, fill, e->n_envs, ho);
        return MGX_OK;
    };
    auto by_layout = [&](auto waves) -> int {
        constexpr int W = decltype(waves)::value;
        if (e->rdev.narrow)
            return layout == MGX_OBS_FRAME ? go(k_raster<P, 0, W, uint32_t>) : layout == MGX_OBS_STACK4 ? go(k_raster<P, 1, W, uint32_t>)
                 : layout == MGX_OBS_STACK3_HI ? go(k_raster<P, 2, W, uint32_t>) : layout == MGX_OBS_SLOT_LO ? go(k_raster<P, 3, W, uint32_t>) : go(k_raster<P, 4, W, uint32_t>);
        return layout == MGX_OBS_FRAME ? go(k_raster<P, 0, W, uint64_t>) : layout == MGX_OBS_STACK4 ? go(k_raster<P, 1, W, uint64_t>)
             : layout == MGX_OBS_STACK3_HI ? go(k_raster<P, 2, W, uint64_t>) : layout == MGX_OBS_SLOT_LO ? go(k_raster<P, 3, W, uint64_t>) : go(k_raster<P, 4, W, uint64_t>);
    };
    int rc = e->raster_waves >= 5 ? by_layout(std::integral_constant<int, 5>{})
           : e->raster_waves == 4 ? by_layout(std::integral_constant<int, 4>{}) : by_layout(std::integral_constant<int, 3>{});
    if (rc) return rc;
    HIP_OK(hipGetLastError());
    return MGX_OK;
}
template <typename P>
static int launch_raster_deferred(mgx_engine *e, const void *sp, uint8_t *out, int64_t env_stride, int view, int layout, hipStream_t st,
                                  const RasterHandoff &ho) {
    size_t lds = e->lds_raster;
    auto go = [&](auto kern) -> int {
        if (int rc = ensure_lds((const void *)kern, lds, e->device)) return rc;
        hipLaunchKernelGGL(kern, dim3(e->n_envs), dim3(256), lds, st, e->rdev, (const P *)sp, out, (long)env_stride, view, e->n_envs, ho);
        return MGX_OK;
    };
    int rc = e->rdev.narrow
           ? (layout == MGX_OBS_FRAME ? go(k_raster_deferred<P, 0, uint32_t>) : layout == MGX_OBS_STACK4 ? go(k_raster_deferred<P, 1, uint32_t>)
              : layout == MGX_OBS_STACK3_HI ? go(k_raster_deferred<P, 2, uint32_t>) : layout == MGX_OBS_SLOT_LO ? go(k_raster_deferred<P, 3, uint32_t>) : go(k_raster_deferred<P, 4, uint32_t>))
           : (layout == MGX_OBS_FRAME ? go(k_raster_deferred<P, 0, uint64_t>) : layout == MGX_OBS_STACK4 ? go(k_raster_deferred<P, 1, uint64_t>)
              : layout == MGX_OBS_STACK3_HI ? go(k_raster_deferred<P, 2, uint64_t>) : layout == MGX_OBS_SLOT_LO ? go(k_raster_deferred<P, 3, uint64_t>) : go(k_raster_deferred<P, 4, uint64_t>));
    if (rc) return rc;
    HIP_OK(hipGetLastError());
    return MGX_OK;
}
extern "C" {

int mgx_engine_render(mgx_engine *e, const void *state_p, uint8_t *out, int64_t env_stride, int view, int layout,
                      const uint8_t *fill_mask, void *stream) {
    if (!e || !state_p || !out) return fail(MGX_ERR_ARG, "NULL argument");
    if (view != MGX_VIEW_EGO && view != MGX_VIEW_ALLO) return fail(MGX_ERR_ARG, "bad view");
    if (layout < MGX_OBS_FRAME || layout > MGX_OBS_PLANAR) return fail(MGX_ERR_ARG, "bad layout");
    int64_t need = (int64_t)LORES * LORES * (layout == MGX_OBS_FRAME || layout == MGX_OBS_PLANAR ? 3 : 12);
    if (env_stride < need || (env_stride & 3)) return fail(MGX_ERR_ARG, "env_stride too small or not a multiple of 4");
    if (int rc = raster_capacity_ok(e)) return rc;
    ON_DEVICE(e);
    hipStream_t st = (hipStream_t)stream;
    int rc = timing_begin(e, 1, st);
    if (rc) return rc;
    rc = e->dtype == MGX_F32_PURE ? launch_raster<float>(e, state_p, out, env_stride, view, layout, fill_mask, st)
                                  : launch_raster<double>(e, state_p, out, env_stride, view, layout, fill_mask, st);
    if (rc) return rc;
    return timing_end(e, 1, st);
}
// fused env-step: physics and rasterisation of ONE BaseEnv.step() as a producer / consumer pair (see StepHandoff / RasterHandoff)
int mgx_engine_step_render(mgx_engine *e, void *state_p, void *state_f, int32_t *state_i, const int32_t *actions, uint8_t *done,
                           uint8_t *out, int64_t env_stride, int view, int layout, void *stream) {
    if (!e || !state_p || !state_f || !state_i || !actions || !out) return fail(MGX_ERR_ARG, "NULL argument");
    if (view != MGX_VIEW_EGO && view != MGX_VIEW_ALLO) return fail(MGX_ERR_ARG, "bad view");
    if (layout < MGX_OBS_FRAME || layout > MGX_OBS_PLANAR) return fail(MGX_ERR_ARG, "bad layout");
    int64_t need = (int64_t)LORES * LORES * (layout == MGX_OBS_FRAME || layout == MGX_OBS_PLANAR ? 3 : 12);
    if (env_stride < need || (env_stride & 3)) return fail(MGX_ERR_ARG, "env_stride too small or not a multiple of 4");
    if (int rc = raster_capacity_ok(e)) return rc;
    // Every world takes the fused path: a consumer only waits at length once all producers are resident (k_raster's prologue), so
    // worlds whose step workgroups do not all fit at once (two dispatch rounds: ClusterColour; one env per wavefront: the per-env
    // worlds) are safe -- raster workgroups are dispatched as step workgroups retire and free their LDS, find their queue entry
    // there already, and fill the machine through the step kernel's tail.  Measured at 4096 envs, two-call -> fused: MatchRegions
    // 1.46 -> 1.28 ms per env-step, MakeLine 1.49 -> 1.30, FixColour 1.26 -> 1.17, FindDupe 1.84 -> 1.73, ClusterColour 1.94 -> 1.83.
    // ... up to six dispatch rounds of step workgroups.  Beyond that (MoveToCorner: from ~40 000 envs on one GPU) the raster
    // workgroups that are dispatched into every freed slot outrun the finished envs, give up by the million and leave their work to
    // the clean-up launch: measured 7.85 vs 7.49 M env-steps/s at 32 768 envs (5.3 rounds), 7.46 vs 7.71 M at 65 536 (10.7), 5.02 vs
    // 7.62 M at 131 072 (21) -- there the two calls in sequence are the better schedule
    const bool overlap = !getenv("MGX_NO_OVERLAP") && (long)step_blocks(e) <= 6 * step_slots(e);
    if (!overlap) {
        int rc = step_common(e, state_p, state_f, state_i, actions, done, PHYS_STEPS, 1, stream);
        return rc ? rc : mgx_engine_render(e, state_p, out, env_stride, view, layout, nullptr, stream);
    }
    ON_DEVICE(e);
    hipStream_t st = (hipStream_t)stream;
    if (!e->st2) {
        HIP_OK(hipStreamCreateWithFlags(&e->st2, hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming)); HIP_OK(hipEventCreateWithFlags(&e->ev_join, getenv("MGX_JOIN_MARKER") ? hipEventDisableTiming : hipEventDefault));
        HIP_OK(hipMalloc(&e->d_queue, (size_t)e->n_envs * 8)); HIP_OK(hipMalloc(&e->d_deferred, (size_t)e->n_envs * 4)); HIP_OK(hipMalloc(&e->d_hand, 64));
        HIP_OK(hipMemset(e->d_queue, 0, (size_t)e->n_envs * 8)); HIP_OK(hipMemset(e->d_deferred, 0, (size_t)e->n_envs * 4)); HIP_OK(hipMemset(e->d_hand, 0, 64));
        HIP_OK(hipDeviceSynchronize());
        e->hand_tail = e->hand_started = e->hand_epoch = 0;
    }
    e->hand_epoch++;
    if (e->hand_epoch == 0) e->hand_epoch = 1;        // 0 is what the zeroed tables hold
    StepHandoff sh{e->d_queue, e->d_hand, e->d_hand + 1, e->hand_tail, e->hand_epoch, e->dbg_delay_every, e->dbg_delay_sleeps};
    RasterHandoff rh{e->d_queue, e->d_hand + 1, e->hand_started, (unsigned)step_blocks(e), e->hand_epoch, e->d_deferred, e->d_hand + 2, 1,
                     e->dbg_poll_limit ? e->dbg_poll_limit : HANDOFF_POLL_LIMIT};
    // A failure between here and the second launch leaves the device's hand-off counters and their host mirrors out of step (a
    // later call would hand the producers a wrong base): drain both streams, zero counters and tables, start over at 0.  The
    // mirrors themselves move only once both launches are in flight.
    auto recover = [&](int rc) {
        (void)hipStreamSynchronize(e->st2); (void)hipStreamSynchronize(st);
        (void)hipMemset(e->d_hand, 0, 64); (void)hipMemset(e->d_queue, 0, (size_t)e->n_envs * 8); (void)hipMemset(e->d_deferred, 0, (size_t)e->n_envs * 4);
        (void)hipDeviceSynchronize();
        e->hand_tail = e->hand_started = 0;
        return rc;
    };
    // the raster stream joins the caller's stream here (the observation tensor may still be read by earlier work on it) ...
    HIP_OK(hipEventRecord(e->ev_fork, st));
    HIP_OK(hipStreamWaitEvent(e->st2, e->ev_fork, 0));
    int rc = step_common(e, state_p, state_f, state_i, actions, done, PHYS_STEPS, 1, stream, sh);
    if (rc) return recover(rc);
    rc = timing_begin(e, 1, e->st2);
    if (rc) return recover(rc);
    static const bool join_on_kernel = getenv("MGX_JOIN_MARKER") == nullptr;      // (the join event completes with the rasteriser itself: no marker packet
                                                                               // behind it; raster end -> next launch on the caller's stream 13 -> 11 us)
    const bool jk = join_on_kernel && !timing_this_launch(e, 1);
    rc = e->dtype == MGX_F32_PURE ? launch_raster<float>(e, state_p, out, env_stride, view, layout, nullptr, e->st2, rh, jk ? e->ev_join : nullptr)
                                  : launch_raster<double>(e, state_p, out, env_stride, view, layout, nullptr, e->st2, rh, jk ? e->ev_join : nullptr);
    if (rc) return recover(rc);
    e->hand_tail += (unsigned)e->n_envs; e->hand_started += (unsigned)step_blocks(e);
    rc = timing_end(e, 1, e->st2);
    // ... and the caller's stream waits for it: whatever comes next on `stream` sees the finished observation.  Both kernels are
    // in flight from here on: a failure (of the timing events, of the join itself) must still join, or later work on `stream`
    // could read an unfinished observation -- if the event path is what failed, the host waits for the raster stream instead
    hipError_t jerr = hipSuccess;
    if (!jk || rc) jerr = hipEventRecord(e->ev_join, e->st2);
    if (jerr == hipSuccess) jerr = hipStreamWaitEvent(st, e->ev_join, 0);
    if (jerr != hipSuccess) {
        (void)hipStreamSynchronize(e->st2);
        if (!rc) rc = fail(MGX_ERR_HIP, std::string("joining the raster stream: ") + hipGetErrorString(jerr));
    }
    if (rc) return rc;
    // clean-up: the envs whose consumer gave up (producers not all running yet, or a wait that ran out) -- normally none
    rh.mode = 2;
    return e->dtype == MGX_F32_PURE ? launch_raster_deferred<float>(e, state_p, out, env_stride, view, layout, st, rh)
                                    : launch_raster_deferred<double>(e, state_p, out, env_stride, view, layout, st, rh);
}
int mgx_engine_handoff_stats(mgx_engine *e, unsigned *deferred, unsigned *timeouts) {
    if (!e) return fail(MGX_ERR_ARG, "engine is NULL");
    unsigned h[4] = {0, 0, 0, 0};
    if (e->d_hand) { ON_DEVICE(e); HIP_OK(hipMemcpy(h, e->d_hand, 16, hipMemcpyDeviceToHost)); }
    if (deferred) *deferred = h[2];
    if (timeouts) *timeouts = h[3];
    return MGX_OK;
}
// development: the hand-off's counters read on a stream of their own (works while the caller's streams are stuck).  out[0..3] = the device's
// tail / started / deferred / timeouts, [4..6] = the host's mirrors of tail, started, epoch, [7] = queue entries that carry the current epoch,
// [8..15] = the device's words 4..11 (-DMGX_HANG_DEBUG builds count there: rasteriser workgroups begun / past the hand-off / at phase Q / ended,
// clean-up workgroups begun / ended)
int mgx_engine_debug_handoff_peek(mgx_engine *e, unsigned *out) {
    if (!e || !out) return fail(MGX_ERR_ARG, "NULL argument");
    for (int i = 0; i < 16; i++) out[i] = 0;
    if (!e->d_hand) return MGX_OK;
    ON_DEVICE(e);
    // (stream and pinned buffers belong to the engine -- its device, freed with it; one caller at a time per engine, like every entry point)
    hipStream_t &st_dbg = e->peek_stream;
    unsigned long long *&h_q = e->peek_q; unsigned *&h_h = e->peek_h;
    if (!st_dbg) HIP_OK(hipStreamCreateWithFlags(&st_dbg, hipStreamNonBlocking));
    if (!h_q) HIP_OK(hipHostMalloc(&h_q, (size_t)e->n_envs * 8));
    if (!h_h) HIP_OK(hipHostMalloc(&h_h, 64));
    HIP_OK(hipMemcpyAsync(h_h, e->d_hand, 64, hipMemcpyDeviceToHost, st_dbg));
    HIP_OK(hipMemcpyAsync(h_q, e->d_queue, (size_t)e->n_envs * 8, hipMemcpyDeviceToHost, st_dbg));
    HIP_OK(hipStreamSynchronize(st_dbg));
    for (int i = 0; i < 4; i++) out[i] = h_h[i];
    out[4] = e->hand_tail; out[5] = e->hand_started; out[6] = e->hand_epoch;
    unsigned n = 0;
    for (int i = 0; i < e->n_envs; i++) if ((unsigned)(h_q[i] >> 32) == e->hand_epoch) n++;
    out[7] = n;
    for (int i = 0; i < 8; i++) out[8 + i] = h_h[4 + i];
    return MGX_OK;
}
int mgx_engine_render_native(mgx_engine *e, const void *state_p, int env, uint8_t *out, int view, void *stream) {
    if (!e || !state_p || !out) return fail(MGX_ERR_ARG, "NULL argument");
    if (env < 0 || env >= e->n_envs) return fail(MGX_ERR_ARG, "env index out of range");
    if (view != MGX_VIEW_EGO && view != MGX_VIEW_ALLO) return fail(MGX_ERR_ARG, "bad view");
    if (int rc = raster_capacity_ok(e)) return rc;
    ON_DEVICE(e);
    hipStream_t st = (hipStream_t)stream;
    size_t lds = e->lds_raster;
    int blocks = (NATIVE_RES * NATIVE_RES + 255) / 256;
    {
        int rc = e->dtype == MGX_F32_PURE ? ensure_lds((const void *)k_raster_native<float>, lds, e->device)
                                          : ensure_lds((const void *)k_raster_native<double>, lds, e->device);
        if (rc) return rc;
    }
    if (e->dtype == MGX_F32_PURE) hipLaunchKernelGGL((k_raster_native<float>), dim3(blocks), dim3(256), lds, st, e->rdev, (const float *)state_p, out, view, (long)env, e->n_envs);
    else hipLaunchKernelGGL((k_raster_native<double>), dim3(blocks), dim3(256), lds, st, e->rdev, (const double *)state_p, out, view, (long)env, e->n_envs);
    HIP_OK(hipGetLastError());
    return MGX_OK;
}
int mgx_debug_host_threads(int allocating) { return host_threads(allocating != 0); }
int mgx_engine_debug_handoff(mgx_engine *e, int poll_limit, int delay_every, int delay_sleeps) {
    if (e) { e->dbg_poll_limit = poll_limit > 0 ? (unsigned)poll_limit : 0u; e->dbg_delay_every = delay_every > 0 ? (unsigned)delay_every : 0u; e->dbg_delay_sleeps = delay_sleeps > 0 ? (unsigned)delay_sleeps : 0u; }
    return MGX_OK;
}
int mgx_engine_debug_raster_waves(mgx_engine *e, int n) { if (e && n >= 3 && n <= 5) e->raster_waves = n; return MGX_OK; }
int mgx_engine_debug_raster_ecap(mgx_engine *e, int n) { if (e) e->rdev.ecap = n < 1 ? 1 : (n > ECAP ? ECAP : n); return MGX_OK; }
int mgx_engine_debug_raster_qcap(mgx_engine *e, int n) { if (e) { e->qcap_debug = n < 1 ? 1 : n; e->rdev.qcap = e->qcap_debug > e->rdev.qcap_lds ? e->rdev.qcap_lds : e->qcap_debug; } return MGX_OK; }
int mgx_engine_debug_raster_stop(mgx_engine *e, int phase) { if (e) e->rdev.dbg_stop = phase; return MGX_OK; }
int mgx_engine_debug_raster_clocks(mgx_engine *e, void *buf) { if (e) e->rdev.dbg_clk = (unsigned long long *)buf; return MGX_OK; }
int mgx_engine_debug_step_clocks(mgx_engine *e, void *buf) { if (e) e->tdev.dbg_clk = (unsigned long long *)buf; return MGX_OK; }
int mgx_engine_debug_iterations(mgx_engine *e, int it) { if (e) e->dbg_iterations = it; return MGX_OK; }
int mgx_engine_set_entity_colours(mgx_engine *e, const int32_t *ent_colour) {
    if (!e) return fail(MGX_ERR_ARG, "engine is NULL");
    e->rdev.ent_colour_env = ent_colour;
    return MGX_OK;
}
int mgx_engine_set_goal_rects(mgx_engine *e, const double *goal_xyhw) {
    if (!e) return fail(MGX_ERR_ARG, "engine is NULL");
    if (e->dtype == MGX_F32_PURE && goal_xyhw) return fail(MGX_ERR_ARG, "per-env goal rectangles need the fp64 pose type");
    e->rdev.goal_xyhw_env = goal_xyhw;
    return MGX_OK;
}
int mgx_engine_score_overlaps(mgx_engine *e, const void *state_p, const uint8_t *mask, uint8_t *out, void *stream) {
    if (!e || !state_p || !out) return fail(MGX_ERR_ARG, "NULL argument");
    if (e->n_goals == 0) return MGX_OK;
    ON_DEVICE(e);
    ScoreDev s{};
    s.lib = e->d_score_lib; s.ent = e->d_score_ent; s.body_prow = e->d_score_prow; s.goal_ent = e->d_score_goal_ent;
    s.goal_xyhw = e->d_score_goal_xyhw; s.goal_xyhw_env = e->rdev.goal_xyhw_env;
    s.ent_type_env = e->d_ent_type_env; s.ent_present_env = e->d_ent_present_env;
    s.n_entities = (int)e->w.entities.size(); s.n_goals = e->n_goals;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (e->n_envs + 63) / 64;
    if (e->dtype == MGX_F32_PURE) hipLaunchKernelGGL((k_score<float>), dim3(blocks), dim3(64), 0, st, s, (const float *)state_p, mask, out, e->n_envs);
    else hipLaunchKernelGGL((k_score<double>), dim3(blocks), dim3(64), 0, st, s, (const double *)state_p, mask, out, e->n_envs);
    HIP_OK(hipGetLastError());
    return MGX_OK;
}
int mgx_engine_score_points(mgx_engine *e, const void *state_p, int task, int n, const int32_t *entities, const int32_t *cls_default, int n_classes,
                            const int8_t *cls_env, const double *params, int dot_mode, int mm_mode, const uint8_t *mask, double *out, void *stream) {
    if (!e || !state_p || !out || !entities || !params) return fail(MGX_ERR_ARG, "NULL argument");
    if (task < SP_CORNER || task > SP_CLUSTER) return fail(MGX_ERR_ARG, "task must be MGX_SCORE_CORNER, _LINE or _CLUSTER");
    if (n < 1 || n > SP_MAX_BLOCKS) return fail(MGX_ERR_CAPACITY, "1 .. 16 blocks");
    if (task == SP_CLUSTER && (!cls_default || n_classes < 1 || n_classes > SP_MAX_CLASSES)) return fail(MGX_ERR_ARG, "cluster score: classes 1 .. 8 and the default class table");
    if (dot_mode < 0 || dot_mode > 2 || mm_mode < 0 || mm_mode > 2) return fail(MGX_ERR_ARG, "dot_mode / mm_mode: 0 plain, 1 fma on the second product, 2 fma on the first");
    ON_DEVICE(e);
    const int ne = (int)e->w.entities.size();
    ScorePointsDev s{};
    s.task = task; s.n = n; s.n_classes = n_classes; s.dot_mode = dot_mode; s.mm_mode = mm_mode;
    s.p0 = params[0]; s.p1 = params[1]; s.p2 = 0.0;
    s.cls_env = cls_env; s.ent_present_env = e->d_ent_present_env;
    std::vector<int> prow(3 * e->w.bodies.size(), -1);
    for (int m : e->w.state_map) { const int comp = m & 15, b = (m >> 4) & 0xFF, row = m >> 12; if (comp < 3) prow[3 * b + comp] = row; }
    for (int k = 0; k < n; k++) {
        const int en = entities[k];
        if (en < 0 || en >= ne || e->w.entities[en].kind != 1) return fail(MGX_ERR_ARG, "score entities must be blocks of the engine's world");
        const int body = e->w.entities[en].body;
        s.ent[k] = en; s.row_x[k] = prow[3 * body]; s.row_y[k] = prow[3 * body + 1];
        s.cls_default[k] = cls_default ? cls_default[k] : 0;
        if (s.row_x[k] < 0 || s.row_y[k] < 0) return fail(MGX_ERR_STATE, "block without persistent pose rows");
        if (task == SP_CLUSTER && (s.cls_default[k] < 0 || s.cls_default[k] >= n_classes)) return fail(MGX_ERR_ARG, "class out of range");
    }
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (e->n_envs + 63) / 64;
    if (e->dtype == MGX_F32_PURE) hipLaunchKernelGGL((k_score_points<float>), dim3(blocks), dim3(64), 0, st, s, (const float *)state_p, mask, out, e->n_envs);
    else hipLaunchKernelGGL((k_score_points<double>), dim3(blocks), dim3(64), 0, st, s, (const double *)state_p, mask, out, e->n_envs);
    HIP_OK(hipGetLastError());
    return MGX_OK;
}
int mgx_engine_n_goals(const mgx_engine *e) { return e ? e->n_goals : 0; }
}  // extern "C"

// dst row idx[r] (or r) <- src row r (src_stride 0: the same row for all)
__global__ void k_scatter_rows(uint32_t *dst, long dst_stride, const uint32_t *src, long src_stride, const int32_t *idx, int n_words) {
    const long r = blockIdx.x, d = idx ? idx[r] : r;
    for (int i = threadIdx.x; i < n_words; i += blockDim.x) dst[d * dst_stride + i] = src[r * src_stride + i];
}

// env rows[5r]: its step blob <- stage[rows[5r+1] .. +rows[5r+2]), its raster blob <- stage[rows[5r+3] .. +rows[5r+4])
__global__ void k_place_blobs(uint32_t *tab_s, long stride_s, uint32_t *tab_r, long stride_r, const uint32_t *stage, const int32_t *rows) {
    const int32_t *r = rows + 5 * (long)blockIdx.x;
    const long env = r[0];
    for (int i = threadIdx.x; i < r[2]; i += blockDim.x) tab_s[env * stride_s + i] = stage[(long)r[1] + i];
    for (int i = threadIdx.x; i < r[4]; i += blockDim.x) tab_r[env * stride_r + i] = stage[(long)r[3] + i];
}

extern "C" {

int mgx_engine_enable_env_worlds(mgx_engine *e, const mgx_world *capacity_world) {
    if (!e || !capacity_world) return fail(MGX_ERR_ARG, "NULL argument");
    if (!capacity_world->w.finalized) return fail(MGX_ERR_STATE, "capacity world not finalized");
    if (e->env_worlds) return fail(MGX_ERR_STATE, "per-env worlds already enabled");
    if (capacity_world->w.entities.size() != e->w.entities.size()) return fail(MGX_ERR_ARG, "capacity world must have the engine world's entities");
    ON_DEVICE(e);
    WorldBlobs cb, db;
    make_blobs(e->dtype, capacity_world->w, cb);
    make_blobs(e->dtype, e->w, db);
    // (blob SIZES only: the per-env LDS working set -- step_env_stride -- is not monotone in the number of blocks, WorkOff hands manifold slots
    // out by another rule from 26 cache slots on: a five-block world needs 1858 words, the six-block one 1698; the launch geometry follows the
    // maximum over the envs' current worlds, configure_launch below and in set_env_variants, and that maximum is checked against the CU there)
    if (db.step.size() > cb.step.size() || db.raster.size() > cb.raster.size() ||
        db.raster_lds_words > cb.raster_lds_words || db.raster_scratch_d > cb.raster_scratch_d || db.raster_scratch_dc > cb.raster_scratch_dc || db.raster_n_i > cb.raster_n_i)
        return fail(MGX_ERR_ARG, "the capacity world must be at least as large as the engine's world");
    if (cb.h.n_prims > 64) return fail(MGX_ERR_CAPACITY, "draw list longer than 64 primitives");
    const int step_stride = even((int)cb.step.size()), raster_stride = even((int)cb.raster.size());
    uint32_t *tab_s = nullptr, *tab_r = nullptr;
    HIP_OK(hipMalloc(&tab_s, (size_t)e->n_envs * step_stride * 4));
    if (hipMalloc(&tab_r, (size_t)e->n_envs * raster_stride * 4) != hipSuccess) { (void)hipFree(tab_s); return fail(MGX_ERR_HIP, "per-env draw-list table allocation failed"); }
    // every env starts in the default world (its blobs are still on the device)
    hipLaunchKernelGGL(k_scatter_rows, dim3(e->n_envs), dim3(256), 0, 0, tab_s, (long)step_stride, e->d_step, 0L, (const int32_t *)nullptr, (int)db.step.size());
    hipLaunchKernelGGL(k_scatter_rows, dim3(e->n_envs), dim3(256), 0, 0, tab_r, (long)raster_stride, e->d_raster, 0L, (const int32_t *)nullptr, (int)db.raster.size());
    HIP_OK(hipDeviceSynchronize());
    (void)hipFree(e->d_step); (void)hipFree(e->d_raster);
    e->d_step = tab_s; e->d_raster = tab_r;
    e->env_worlds = true; e->cap_prims = cb.h.n_prims;
    e->step_stride = step_stride; e->raster_stride = raster_stride;
    e->tdev.words = tab_s; e->tdev.tmpl_stride_words = step_stride;
    e->rdev.words = tab_r; e->rdev.tmpl_stride_words = raster_stride;
    e->env_world.assign(e->n_envs, std::shared_ptr<World>());
    {   // k_score: every env's shape types / presence flags, the engine world's to begin with
        const int ne = (int)e->w.entities.size();
        std::vector<int8_t> ty(ne); std::vector<uint8_t> on(ne);
        for (int i = 0; i < ne; i++) { ty[i] = (int8_t)(e->w.entities[i].kind == 1 ? e->w.entities[i].shape_type : -1); on[i] = e->w.entities[i].enabled ? 1 : 0; }
        int8_t *d_ty = nullptr; uint8_t *d_on = nullptr;
        HIP_OK(hipMalloc(&e->d_ent_type_env, (size_t)ne * e->n_envs)); HIP_OK(hipMalloc(&e->d_ent_present_env, (size_t)ne * e->n_envs));
        HIP_OK(hipMalloc(&d_ty, ne)); HIP_OK(hipMalloc(&d_on, ne));
        HIP_OK(hipMemcpy(d_ty, ty.data(), ne, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(d_on, on.data(), ne, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_scatter_ent_rows, dim3(e->n_envs), dim3(64), 0, 0, e->d_ent_type_env, e->d_ent_present_env, d_ty, d_on, (const int32_t *)nullptr, ne, (long)e->n_envs);
        HIP_OK(hipDeviceSynchronize());
        (void)hipFree(d_ty); (void)hipFree(d_on);
    }
    {   // the capacity world itself must be launchable
        int rc = configure_launch(e, (int)cb.step.size(), cb.step_env_stride, cb.raster_lds_words, (int)cb.raster.size(), cb.raster_scratch_d, cb.raster_scratch_dc, cb.raster_n_i);
        if (rc) return rc;
        if (e->lds_raster > (size_t)MAX_LDS_BYTES) return fail(MGX_ERR_CAPACITY, "capacity world's draw list does not fit LDS");
    }
    e->fp_step_words.assign(e->n_envs, (int)db.step.size()); e->fp_env_stride.assign(e->n_envs, db.step_env_stride);
    e->fp_raster_words.assign(e->n_envs, db.raster_lds_words); e->fp_raster_full.assign(e->n_envs, (int)db.raster.size()); e->fp_scratch_d.assign(e->n_envs, db.raster_scratch_d); e->fp_scratch_dc.assign(e->n_envs, db.raster_scratch_dc);
    e->fp_raster_n_i.assign(e->n_envs, db.raster_n_i);
    int rc = configure_launch(e, (int)db.step.size(), db.step_env_stride, db.raster_lds_words, (int)db.raster.size(), db.raster_scratch_d, db.raster_scratch_dc, db.raster_n_i);
    if (rc) return rc;
    e->rows_p = std::max(e->rows_p, state_rows_p(cb.h)); e->rows_f = std::max(e->rows_f, state_rows_f(cb.h)); e->rows_i = std::max(e->rows_i, state_rows_i(cb.h));
    return MGX_OK;
}

int mgx_engine_set_env_variants(mgx_engine *e, int m, const int32_t *env_idx, const uint8_t *enabled, const int32_t *shape_types, void *stream) {
    if (!e || !env_idx || m < 0) return fail(MGX_ERR_ARG, "bad argument");
    if (!e->env_worlds) return fail(MGX_ERR_STATE, "call mgx_engine_enable_env_worlds first");
    if (m == 0) return MGX_OK;
    ON_DEVICE(e);
    if (e->variants_pending) { HIP_OK(hipEventSynchronize(e->ev_variants)); e->variants_pending = false; }      // the previous call's uploads read the buffers reused below
    if (!e->ev_variants) HIP_OK(hipEventCreateWithFlags(&e->ev_variants, hipEventDisableTiming));
    const bool dbg = getenv("MGX_DEBUG_VARIANTS") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tm[8]; int ti = 0; tm[ti++] = now();
    const int ne = (int)e->w.entities.size();
    for (int k = 0; k < m; k++) if (env_idx[k] < 0 || env_idx[k] >= e->n_envs) return fail(MGX_ERR_ARG, "env index out of range");
    // unique signatures of this call; worlds that are still alive are reused
    // (a world's two blobs are serialised into buffers of the building thread and copied straight into the pinned staging buffer, at an
    // offset the thread reserves with one atomic add: what a Uniq keeps of them is sizes and offsets -- round 3 kept 160 MB of blob
    // vectors per reset of 4096 Cluster worlds alive until a second, packing pass had copied them)
    struct BlobMeta { TmplHeader h; int step_words = 0, raster_words = 0, step_env_stride = 0, raster_lds_words = 0, raster_scratch_d = 0, raster_scratch_dc = 0, raster_n_i = 0; int32_t off_s = 0, off_r = 0; };
    struct Uniq { std::string sig; int first; std::shared_ptr<World> world; BlobMeta blobs; std::string err; int rc = 0; };
    std::vector<Uniq> uniq;
    std::vector<int> which(m);
    {
        // the signatures (and their hashes) of all envs on the pool, then one serial pass that dedups them through an open-addressed table
        // of indices (round 3: a string allocated, hashed and inserted into an unordered_map per env, serially: 1 ms of a 10 ms reset)
        std::vector<std::string> sigs(m); std::vector<uint64_t> hs(m);
        const int nt = m >= 1024 ? std::min(host_threads(false), 16) : 1;
        host_parallel(nt, [&](int t) {
            const int k0 = (int)((long)m * t / nt), k1 = (int)((long)m * (t + 1) / nt);
            for (int k = k0; k < k1; k++) {
                std::string &sig = sigs[k];
                sig.assign((size_t)2 * ne, '\0');
                uint64_t h = 1469598103934665603ull;
                for (int i = 0; i < ne; i++) {
                    sig[i] = (char)(enabled ? (enabled[(size_t)k * ne + i] ? 1 : 0) : (e->w.entities[i].enabled ? 1 : 0));
                    int st = shape_types ? shape_types[(size_t)k * ne + i] : -1;
                    sig[ne + i] = (char)((e->w.entities[i].kind == 1 ? (st >= 0 ? st : e->w.entities[i].shape_type) : 0) + 1);
                }
                for (char c : sig) { h ^= (uint8_t)c; h *= 1099511628211ull; }
                hs[k] = h;
            }
        });
        size_t cap = 16;
        while (cap < (size_t)2 * m) cap <<= 1;
        std::vector<int> slot(cap, -1);
        uniq.reserve(m);
        const bool any_live = !e->world_by_sig.empty();
        for (int k = 0; k < m; k++) {
            size_t i = (size_t)hs[k] & (cap - 1);
            for (;; i = (i + 1) & (cap - 1)) {
                const int u = slot[i];
                if (u < 0) {
                    slot[i] = (int)uniq.size(); which[k] = (int)uniq.size();
                    Uniq nu; nu.sig = std::move(sigs[k]); nu.first = k;
                    if (any_live) {
                        auto live = e->world_by_sig.find(nu.sig);
                        if (live != e->world_by_sig.end()) nu.world = live->second.lock();
                    }
                    uniq.push_back(std::move(nu));
                    break;
                }
                if (hs[uniq[u].first] == hs[k] && uniq[u].sig == sigs[k]) { which[k] = u; break; }
            }
        }
    }
    tm[ti++] = now();
    // the staging buffer holds every distinct world's blobs back to back; no blob is larger than the capacity world's, so this call needs
    // at most (distinct worlds) x (both strides) words -- and its word offsets are int32 (BlobMeta::off_s / off_r): that is what the
    // 2^31 check is about.  The buffer grows geometrically up to one world per env (a whole-batch reset of a crowded task reaches that
    // at once; a one-env auto-reset pins one world's worth, not the batch's: 160 MB for 4096 Cluster worlds, GBs at 64k envs)
    const size_t per_world = (size_t)e->step_stride + (size_t)e->raster_stride;
    const size_t stage_need = uniq.size() * per_world;
    if (stage_need > 0x7fffffffull) return fail(MGX_ERR_CAPACITY, "too many distinct worlds in one call (staging offsets are 32-bit words): split the call");
    // (pinned staging: a pageable copy of this size -- 160 MB for 4096 distinct Cluster worlds -- would dominate the reset)
    if (e->h_stage_words < stage_need) {
        const size_t cap = std::max(stage_need, std::min(2 * e->h_stage_words, (size_t)e->n_envs * per_world));
        if (e->h_stage) (void)hipHostFree(e->h_stage);
        e->h_stage = nullptr; e->h_stage_words = 0;
        HIP_OK(hipHostMalloc(reinterpret_cast<void **>(&e->h_stage), cap * 4 + 4096, hipHostMallocDefault));
        e->h_stage_words = cap + 1024;
    }
    uint32_t *host = e->h_stage;
    std::atomic<size_t> cursor{0};
    // build what is missing and serialise everything, a few host threads wide
    int n_threads = host_threads();
    if (uniq.size() < 8) n_threads = 1;
    std::atomic<size_t> next_u{0};
    auto work = [&](int) {
        static thread_local WorldBlobs tb;
        static thread_local std::vector<uint8_t> en; static thread_local std::vector<int> st;
        for (size_t u0 = next_u.fetch_add(4); u0 < uniq.size(); u0 = next_u.fetch_add(4))
        for (size_t u = u0; u < u0 + 4 && u < uniq.size(); u++) {
            Uniq &U = uniq[u];
            if (!U.world) {
                en.resize(ne); st.resize(ne);
                for (int i = 0; i < ne; i++) { en[i] = (uint8_t)U.sig[i]; st[i] = (int)U.sig[ne + i] - 1; }
                auto w = std::make_shared<World>();
                U.rc = e->w.variant(en.data(), st.data(), *w, U.err);
                if (U.rc) continue;
                U.world = std::move(w);
            }
            make_blobs(e->dtype, *U.world, tb);
            BlobMeta &B = U.blobs;
            B.h = tb.h; B.step_words = (int)tb.step.size(); B.raster_words = (int)tb.raster.size(); B.step_env_stride = tb.step_env_stride;
            B.raster_lds_words = tb.raster_lds_words; B.raster_scratch_d = tb.raster_scratch_d; B.raster_scratch_dc = tb.raster_scratch_dc; B.raster_n_i = tb.raster_n_i;
            if (B.step_words > e->step_stride || B.raster_words > e->raster_stride) { U.rc = -2; U.err = "world variant larger than the capacity world"; continue; }
            const size_t off = cursor.fetch_add((size_t)B.step_words + (size_t)B.raster_words);
            B.off_s = (int32_t)off; B.off_r = (int32_t)(off + B.step_words);
            std::memcpy(host + B.off_s, tb.step.data(), (size_t)B.step_words * 4);
            std::memcpy(host + B.off_r, tb.raster.data(), (size_t)B.raster_words * 4);
        }
    };
    host_parallel(n_threads, work);
    tm[ti++] = now();
    for (auto &U : uniq) {
        if (U.rc) return fail(U.rc == -2 ? MGX_ERR_CAPACITY : MGX_ERR_ARG, U.err);
        if (U.blobs.h.n_prims > (e->rdev.narrow ? 32 : 64)) return fail(MGX_ERR_CAPACITY, "world variant larger than the capacity world");
        if (state_rows_p(U.blobs.h) > e->rows_p || state_rows_f(U.blobs.h) > e->rows_f || state_rows_i(U.blobs.h) > e->rows_i)
            return fail(MGX_ERR_CAPACITY, "world variant needs more state rows than the capacity world");
    }
    const size_t total = cursor.load();
    tm[ti++] = now();
    // per env: destination env, source offsets and sizes of its two blobs
    std::vector<int32_t> &rows = e->v_rows;
    rows.resize((size_t)5 * m);
    for (int k = 0; k < m; k++) {
        const int u = which[k];
        rows[5 * k] = env_idx[k]; rows[5 * k + 1] = uniq[u].blobs.off_s; rows[5 * k + 2] = uniq[u].blobs.step_words;
        rows[5 * k + 3] = uniq[u].blobs.off_r; rows[5 * k + 4] = uniq[u].blobs.raster_words;
    }
    if (e->stage_words < total) {
        if (e->d_stage) (void)hipFree(e->d_stage);
        e->d_stage = nullptr; e->stage_words = 0;
        HIP_OK(hipMalloc(&e->d_stage, total * 4 + (total >> 3) * 4 + 4096));       // (some room: the total differs from reset to reset)
        e->stage_words = total + (total >> 3) + 1024;
    }
    if (e->stage_idx_n < rows.size()) {
        if (e->d_stage_idx) (void)hipFree(e->d_stage_idx);
        e->d_stage_idx = nullptr; e->stage_idx_n = 0;
        HIP_OK(hipMalloc(&e->d_stage_idx, rows.size() * 4));
        e->stage_idx_n = rows.size();
    }
    hipStream_t st = (hipStream_t)stream;
    // From the first asynchronous copy on, the engine's staging buffers are being read by the DMA engine: WHATEVER path leaves this
    // function (an error return between two enqueues included) records the event behind what has been enqueued and marks the uploads
    // pending, so that the next call / mgx_engine_destroy waits before it reuses or frees them (advisor, round 4)
    struct PendingGuard {
        mgx_engine *e; hipStream_t st;
        ~PendingGuard() { if (hipEventRecord(e->ev_variants, st) == hipSuccess) e->variants_pending = true; else (void)hipStreamSynchronize(st); }
    } pending_guard{e, st};
    HIP_OK(hipMemcpyAsync(e->d_stage, host, total * 4, hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(e->d_stage_idx, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_place_blobs, dim3(m), dim3(256), 0, st, e->d_step, (long)e->step_stride, e->d_raster, (long)e->raster_stride, e->d_stage, e->d_stage_idx);
    HIP_OK(hipGetLastError());
    // k_score's per-env entity tables (shape type / presence of every entity of these envs)
    std::vector<int8_t> &ent_ty = e->v_ty; std::vector<uint8_t> &ent_on = e->v_on;
    ent_ty.resize((size_t)m * ne); ent_on.resize((size_t)m * ne);
    e->v_idx.assign(env_idx, env_idx + m);
    for (int k = 0; k < m; k++) {
        const std::string &sig = uniq[which[k]].sig;
        for (int i = 0; i < ne; i++) { ent_on[(size_t)k * ne + i] = (uint8_t)sig[i]; ent_ty[(size_t)k * ne + i] = (int8_t)(e->w.entities[i].kind == 1 ? (int)sig[ne + i] - 1 : -1); }
    }
    if (e->d_v_ent_cap < ent_ty.size()) {
        if (e->d_v_ty) (void)hipFree(e->d_v_ty);
        if (e->d_v_on) (void)hipFree(e->d_v_on);
        e->d_v_ty = nullptr; e->d_v_on = nullptr; e->d_v_ent_cap = 0;
        HIP_OK(hipMalloc(&e->d_v_ty, ent_ty.size())); HIP_OK(hipMalloc(&e->d_v_on, ent_on.size()));
        e->d_v_ent_cap = ent_ty.size();
    }
    if (e->d_v_idx_cap < (size_t)m) {
        if (e->d_v_idx) (void)hipFree(e->d_v_idx);
        e->d_v_idx = nullptr; e->d_v_idx_cap = 0;
        HIP_OK(hipMalloc(&e->d_v_idx, (size_t)m * 4));
        e->d_v_idx_cap = (size_t)m;
    }
    HIP_OK(hipMemcpyAsync(e->d_v_ty, ent_ty.data(), ent_ty.size(), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(e->d_v_on, ent_on.data(), ent_on.size(), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(e->d_v_idx, e->v_idx.data(), (size_t)m * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_scatter_ent_rows, dim3(m), dim3(64), 0, st, e->d_ent_type_env, e->d_ent_present_env, e->d_v_ty, e->d_v_on, (const int32_t *)e->d_v_idx, ne, (long)e->n_envs);
    HIP_OK(hipGetLastError());
    // (no wait here: the copies and the two small kernels run while the caller goes on to the placement sampling, which is host
    // work; the staging buffer, the row table and the entity rows are the engine's and stay as they are until the next call:
    // pending_guard records the event when this function returns)
    tm[ti++] = now();
    std::vector<std::shared_ptr<World>> retired(m);      // the envs' previous worlds: freed below, a few threads wide
    for (int k = 0; k < m; k++) {
        const int env = env_idx[k];
        const BlobMeta &B = uniq[which[k]].blobs;
        retired[k] = std::move(e->env_world[env]);
        e->env_world[env] = uniq[which[k]].world;
        e->fp_step_words[env] = B.step_words; e->fp_env_stride[env] = B.step_env_stride;
        e->fp_raster_words[env] = B.raster_lds_words; e->fp_raster_full[env] = B.raster_words; e->fp_scratch_d[env] = B.raster_scratch_d; e->fp_scratch_dc[env] = B.raster_scratch_dc; e->fp_raster_n_i[env] = B.raster_n_i;
    }
    {
        auto mx = [](const std::vector<int> &v) { return *std::max_element(v.begin(), v.end()); };
        int rc = configure_launch(e, mx(e->fp_step_words), mx(e->fp_env_stride), mx(e->fp_raster_words), mx(e->fp_raster_full), mx(e->fp_scratch_d), mx(e->fp_scratch_dc), mx(e->fp_raster_n_i));
        if (rc) return rc;
    }
    // worlds stay findable by signature for later calls -- where signatures repeat at all: with nearly every env its own world
    // (ClusterColour-TestAll: ~4000 distinct of 4096) the table only cost its inserts and, every few resets, a 10 ms purge
    // (small calls -- per-env auto-resets -- register theirs too: otherwise a world that is alive in another env is rebuilt every time)
    if (uniq.size() * 4 <= (size_t)m || m < 64) for (auto &U : uniq) e->world_by_sig[U.sig] = U.world;
    tm[ti++] = now();
    const int n_uniq = (int)uniq.size();
    {
        // a World is a few hundred small allocations, and ~4000 of them retire per reset: they (and this call's blob buffers) are
        // dropped by a detached helper thread while the caller goes on (round 3 freed them here, a few threads wide: 5-14 ms of the
        // call).  The helper owns what it frees -- shared_ptrs and vectors, nothing of the engine.
        // A handful of worlds (per-env auto-resets) are freed here: a thread per call would cost more than the frees.  No exception
        // crosses the C ABI: if the helper cannot be started (std::system_error) the bins are freed inline.
        if (m < 64) { retired.clear(); uniq.clear(); }
        else {
            auto *bin_worlds = new (std::nothrow) std::vector<std::shared_ptr<World>>(std::move(retired));
            auto *bin_uniq = new (std::nothrow) std::vector<Uniq>(std::move(uniq));
            try { std::thread([bin_worlds, bin_uniq] { delete bin_worlds; delete bin_uniq; }).detach(); }
            catch (...) { delete bin_worlds; delete bin_uniq; }
        }
    }
    if (e->world_by_sig.size() > (size_t)4 * e->n_envs + 64)
        for (auto it = e->world_by_sig.begin(); it != e->world_by_sig.end();) it = it->second.expired() ? e->world_by_sig.erase(it) : std::next(it);
    tm[ti++] = now();
    if (dbg) fprintf(stderr, "mgx: set_env_variants %d envs, %zu distinct worlds, %d threads: signatures %.1f ms, build + serialise %.1f, pack %.1f, upload + place %.1f, bookkeeping %.1f, frees %.1f\n",
                     m, (size_t)n_uniq, n_threads, tm[1] - tm[0], tm[2] - tm[1], tm[3] - tm[2], tm[4] - tm[3], tm[5] - tm[4], tm[6] - tm[5]);
    return n_uniq;
}

int mgx_engine_env_randomise_all_poses_batch(const mgx_engine *e, int m, const int32_t *env_idx, double *poses, const int *ents, int n, const uint8_t *ignore,
                                             const double arena_lrbt[4], const uint8_t *rand_pos, const uint8_t *rand_rot,
                                             const double *pos_limits, const double *rot_limits, int limits_per_env,
                                             const uint64_t *mt_state_addr, const double *ent_hw) {
    if (!e) return fail(MGX_ERR_ARG, "engine is NULL");
    if (m < 0 || !env_idx || !poses || !ents || n < 1 || !arena_lrbt || !rand_pos || !rand_rot || !pos_limits || !rot_limits || !mt_state_addr)
        return fail(MGX_ERR_ARG, "NULL argument");
    const int ne = (int)e->w.entities.size();
    for (int i = 0; i < n; i++) if (ents[i] < 0 || ents[i] >= ne) return fail(MGX_ERR_ARG, "entity index out of range");
    for (int k = 0; k < m; k++) if (env_idx[k] < 0 || env_idx[k] >= e->n_envs) return fail(MGX_ERR_ARG, "env index out of range");
    auto world_of = [&](int k) -> const World & {
        if (e->env_worlds && e->env_world[env_idx[k]]) return *e->env_world[env_idx[k]];
        return e->w;
    };
    return randomise_batch(world_of, ne, m, poses, ents, n, ignore, arena_lrbt, rand_pos, rand_rot, pos_limits, rot_limits, limits_per_env, mt_state_addr, ent_hw);
}

int mgx_engine_env_world_info(const mgx_engine *e, int env, int key, int *out) {
    if (!e || !out) return fail(MGX_ERR_ARG, "NULL argument");
    if (env < 0 || env >= e->n_envs) return fail(MGX_ERR_ARG, "env index out of range");
    mgx_world tmp;
    tmp.w = (e->env_worlds && e->env_world[env]) ? *e->env_world[env] : e->w;
    return mgx_world_info(&tmp, key, out);
}

int mgx_engine_set_timing(mgx_engine *e, int enable) {
    if (!e) return fail(MGX_ERR_ARG, "engine is NULL");
    e->timing = enable < 0 ? 0 : enable;
    if (e->timing) {      // the event ring is made here, not at the first sampled launch (which is inside the caller's timed region)
        ON_DEVICE(e);
        for (int which = 0; which < 2; which++)
            if (e->ev[which].empty()) {
                e->ev[which].resize(2 * TIMING_RING);
                for (auto &ev : e->ev[which]) HIP_OK(hipEventCreate(&ev));
            }
    }
    e->ev_count[0] = e->ev_count[1] = 0;
    e->launch_count[0] = e->launch_count[1] = 0;
    return MGX_OK;
}
int mgx_engine_timing_read(mgx_engine *e, int which, float *ms, int max) {
    if (!e || !ms || which < 0 || which > 1) return fail(MGX_ERR_ARG, "bad argument");
    ON_DEVICE(e);
    int n = e->ev_count[which] < TIMING_RING ? e->ev_count[which] : TIMING_RING;
    if (n > max) n = max;
    int first = e->ev_count[which] - n;
    for (int k = 0; k < n; k++) {
        int slot = (first + k) % TIMING_RING;
        HIP_OK(hipEventSynchronize(e->ev[which][2 * slot + 1]));
        HIP_OK(hipEventElapsedTime(&ms[k], e->ev[which][2 * slot], e->ev[which][2 * slot + 1]));
    }
    e->ev_count[which] = 0;
    return n;
}

}  // extern "C"

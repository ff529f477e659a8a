// mgx_api.hip -- the C ABI declared in include/mgx.h: world description, engine lifetime and
// kernel launches.  Device state blobs and output tensors are owned by the caller (PyTorch);
// the engine owns only its small constant template buffers and timing events.
#include <hip/hip_runtime.h>
#include <thread>
#include <type_traits>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mgx.h"
#include "mgx_raster.hip"
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
constexpr int TIMING_RING = 4096;
}  // namespace

struct mgx_world { World w; };

struct mgx_engine {
    World w;
    TmplHeader h;
    int n_envs = 0, device = 0, dtype = 0, L = 0;
    uint32_t *d_step = nullptr, *d_raster = nullptr;
    TmplDev tdev{};
    RasterDev rdev{};
    int raster_waves = 4;       // k_raster variant: workgroups per CU its register cap is set for (3, 4 or 5)
    size_t lds_step = 0, lds_raster = 0;
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
int mgx_world_randomise_all_poses_batch(const mgx_world *w, int m, double *poses, const int *ents, int n, const uint8_t *ignore,
                                        const double arena_lrbt[4], const uint8_t *rand_pos, const uint8_t *rand_rot,
                                        const double *pos_limits, const double *rot_limits, int limits_per_env,
                                        const uint64_t *mt_state_addr, const double *ent_hw) {
    if (!w || !w->w.finalized) return fail(MGX_ERR_STATE, "world not finalized");
    if (m < 0 || !poses || !ents || n < 1 || !arena_lrbt || !rand_pos || !rand_rot || !pos_limits || !rot_limits || !mt_state_addr)
        return fail(MGX_ERR_ARG, "NULL argument");
    const int ne = (int)w->w.entities.size();
    for (int i = 0; i < n; i++) if (ents[i] < 0 || ents[i] >= ne) return fail(MGX_ERR_ARG, "entity index out of range");
    // envs are independent (own stream, own poses): spread them over a few host threads
    int n_threads = (int)std::thread::hardware_concurrency();
    n_threads = n_threads < 1 ? 1 : (n_threads > 16 ? 16 : n_threads);
    if (m < 64) n_threads = 1;
    std::vector<long> rej(n_threads, 0);
    std::vector<int> bad(n_threads, 0);
    auto work = [&](int t) {
        for (int k = t; k < m; k += n_threads) {
            // numpy's mt19937_state: uint32 key[624]; int pos
            uint32_t *key = reinterpret_cast<uint32_t *>((uintptr_t)mt_state_addr[k]);
            int *pos = reinterpret_cast<int *>((uintptr_t)mt_state_addr[k] + 624 * sizeof(uint32_t));
            if (!key || *pos < 0 || *pos > 624) { bad[t] = 1; return; }
            const size_t lo = limits_per_env ? (size_t)k * n : 0;
            int rc = w->w.randomise_all_poses(poses + (size_t)k * ne * 3, ents, n, ignore, arena_lrbt, rand_pos, rand_rot, pos_limits + lo, rot_limits + lo, key, pos,
                                              ent_hw ? ent_hw + (size_t)k * ne * 2 : nullptr);
            if (rc < 0) { bad[t] = 2; return; }
            rej[t] += rc;
        }
    };
    if (n_threads == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; t++) pool.emplace_back(work, t);
        for (auto &th : pool) th.join();
    }
    long rejected = 0;
    for (int t = 0; t < n_threads; t++) {
        if (bad[t] == 1) return fail(MGX_ERR_ARG, "bad MT19937 state");
        if (bad[t] == 2) return fail(MGX_ERR_CAPACITY, "could not place the entities (PlacementError after 10 retries)");
        rejected += rej[t];
    }
    return (int)(rejected > 0x7fffffff ? 0x7fffffff : rejected);
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
template <typename R, typename P>
static int build_step_template(mgx_engine *e, const std::vector<int32_t> &iw, const std::vector<double> &rw, const std::vector<double> &pw) {
    const TmplHeader &h = e->h;
    auto even = [](int x) { return (x + 1) & ~1; };
    int hw = even((int)sizeof(TmplHeader) / 4);
    int off_i = hw, off_r = even(off_i + (int)iw.size());
    int rwords = (int)rw.size() * (int)(sizeof(R) / 4), pwords = (int)pw.size() * (int)(sizeof(P) / 4);
    int off_p = even(off_r + rwords);
    int total = even(off_p + pwords);
    std::vector<uint32_t> blob(total, 0);
    std::memcpy(blob.data(), &h, sizeof(TmplHeader));
    std::memcpy(blob.data() + off_i, iw.data(), iw.size() * 4);
    { R *d = reinterpret_cast<R *>(blob.data() + off_r); for (size_t i = 0; i < rw.size(); i++) d[i] = (R)rw[i]; }
    { P *d = reinterpret_cast<P *>(blob.data() + off_p); for (size_t i = 0; i < pw.size(); i++) d[i] = (P)pw[i]; }
    HIP_OK(hipMalloc(&e->d_step, blob.size() * 4));
    HIP_OK(hipMemcpy(e->d_step, blob.data(), blob.size() * 4, hipMemcpyHostToDevice));
    WorkOff wo(h);
    int words_p = even(wo.n_p * (int)(sizeof(P) / 4)), words_r = even(wo.n_r * (int)(sizeof(R) / 4)), words_i = even(wo.n_i);
    int stride = words_p + words_r + words_i;
    while (stride % 32 != 2) stride += 2;     // envs of one wave start on distinct LDS banks
    e->tdev.words = e->d_step; e->tdev.n_words = total;
    e->tdev.off_i = off_i; e->tdev.off_r = off_r; e->tdev.off_p = off_p;
    e->tdev.env_stride_words = stride; e->tdev.env_off_r = words_p; e->tdev.env_off_i = words_p + words_r;
    e->tdev.lds_tmpl_words = total;
    return MGX_OK;
}

static size_t step_lds_bytes(const mgx_engine *e, int L) { return (size_t)(e->tdev.lds_tmpl_words + (64 / L) * e->tdev.env_stride_words) * 4; }

template <typename R, typename P, int L>
static int launch_step_L(mgx_engine *e, void *sp, void *sf, int32_t *si, const int32_t *actions, uint8_t *done, int n_sub,
                         int count_step, hipStream_t st) {
    auto kern = k_step<R, P, L>;
    size_t lds = step_lds_bytes(e, L);
    static thread_local const void *configured = nullptr;   // per-instantiation, per-thread
    if (configured != (const void *)kern) {
        HIP_OK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds > 65536 ? (int)lds : 65536));
        configured = (const void *)kern;
    }
    int epb = 64 / L, blocks = (e->n_envs + epb - 1) / epb;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, st, e->tdev, (P *)sp, (R *)sf, si, actions, done, e->n_envs, n_sub,
                       count_step, e->dbg_iterations >= 0 ? e->dbg_iterations : PHYS_ITER);
    HIP_OK(hipGetLastError());
    return MGX_OK;
}
template <typename R, typename P>
static int launch_step(mgx_engine *e, void *sp, void *sf, int32_t *si, const int32_t *actions, uint8_t *done, int n_sub,
                       int count_step, hipStream_t st) {
    switch (e->L) {
        case 4: return launch_step_L<R, P, 4>(e, sp, sf, si, actions, done, n_sub, count_step, st);
        case 8: return launch_step_L<R, P, 8>(e, sp, sf, si, actions, done, n_sub, count_step, st);
        case 16: return launch_step_L<R, P, 16>(e, sp, sf, si, actions, done, n_sub, count_step, st);
        case 32: return launch_step_L<R, P, 32>(e, sp, sf, si, actions, done, n_sub, count_step, st);
        case 64: return launch_step_L<R, P, 64>(e, sp, sf, si, actions, done, n_sub, count_step, st);
    }
    return fail(MGX_ERR_ARG, "lanes_per_env must be 4, 8, 16, 32 or 64");
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
    HIP_OK(hipSetDevice(device));
    mgx_engine *e = new mgx_engine();
    e->w = w->w; e->n_envs = n_envs; e->device = device; e->dtype = dtype;
    std::vector<int32_t> iw; std::vector<double> rw, pw;
    e->w.serialise(e->h, iw, rw, pw);
    int rc = dtype == MGX_F32 ? build_step_template<float, double>(e, iw, rw, pw)
           : dtype == MGX_F64 ? build_step_template<double, double>(e, iw, rw, pw)
                              : build_step_template<float, float>(e, iw, rw, pw);
    if (rc) { delete e; return rc; }
    // lanes per env: caller's choice, else the widest group (most narrowphase parallelism) that still lets
    // two workgroups share a CU's LDS
    int L = lanes_per_env;
    if (L == 0) {
        L = 16;
        while (L < 64 && step_lds_bytes(e, L) > (size_t)MAX_LDS_BYTES / 2) L *= 2;
    }
    if (L != 4 && L != 8 && L != 16 && L != 32 && L != 64) { mgx_engine_destroy(e); return fail(MGX_ERR_ARG, "lanes_per_env must be 0, 4, 8, 16, 32 or 64"); }
    if (step_lds_bytes(e, L) > (size_t)MAX_LDS_BYTES) { mgx_engine_destroy(e); return fail(MGX_ERR_CAPACITY, "world working set does not fit LDS at this lanes_per_env"); }
    e->L = L; e->lds_step = step_lds_bytes(e, L);
    // raster template: header + ints + (prim reals, prim verts) in fp64
    {
        TmplOff o(e->h);
        auto even = [](int x) { return (x + 1) & ~1; };
        int hw = even((int)sizeof(TmplHeader) / 4), off_i = hw, off_q = even(off_i + (int)iw.size());
        int nq = e->h.n_prims * PRIM_RWORDS + 2 * e->h.n_pverts;
        int total = off_q + 2 * nq;
        std::vector<uint32_t> blob(total, 0);
        std::memcpy(blob.data(), &e->h, sizeof(TmplHeader));
        std::memcpy(blob.data() + off_i, iw.data(), iw.size() * 4);
        std::memcpy(blob.data() + off_q, rw.data() + o.prim_r, (size_t)nq * 8);
        if (hipMalloc(&e->d_raster, blob.size() * 4) != hipSuccess || hipMemcpy(e->d_raster, blob.data(), blob.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            mgx_engine_destroy(e); return fail(MGX_ERR_HIP, "raster template upload failed");
        }
        RasterOff ro(e->h);
        e->rdev.words = e->d_raster; e->rdev.n_words = total; e->rdev.off_i = off_i; e->rdev.off_q = off_q;
        e->rdev.lds_tmpl_words = total; e->rdev.scratch_d = ro.n_d; e->rdev.bg_rgb = BG_RGB;
        int off_tiles = even(2 * ro.n_d + ro.n_i);
        e->rdev.off_tiles = off_tiles;
        // per-tile (u64 mask + i32 base) + queue (u64 mask + 2 x i32) + counters + overflow bitmap + phase E records (u64 sums, u16 entry)
        int extra = N_TILES * 3 + QCAP * 4 + 8 + OVF_WORDS + ECAP * 2 + ECAP / 2;
        e->rdev.qcap = QCAP; e->rdev.ecap = ECAP;
        e->lds_raster = (size_t)(total + off_tiles + extra) * 4;
        // as many workgroups per CU as LDS allows (512 B allocation slack), between 3 and 5
        int fit = (int)((size_t)MAX_LDS_BYTES / (e->lds_raster + 512));
        e->raster_waves = fit >= 5 ? 5 : (fit == 4 ? 4 : 3);
    }
    *out = e;
    return MGX_OK;
}
void mgx_engine_destroy(mgx_engine *e) {
    if (!e) return;
    if (e->d_step) (void)hipFree(e->d_step);
    if (e->d_raster) (void)hipFree(e->d_raster);
    for (int k = 0; k < 2; k++) for (auto &ev : e->ev[k]) (void)hipEventDestroy(ev);
    delete e;
}
int mgx_engine_state_shape(const mgx_engine *e, int *rows_p, int *rows_f, int *rows_i, int *size_p, int *size_f) {
    if (!e) return fail(MGX_ERR_ARG, "engine is NULL");
    if (rows_p) *rows_p = state_rows_p(e->h);
    if (rows_f) *rows_f = state_rows_f(e->h);
    if (rows_i) *rows_i = state_rows_i(e->h);
    if (size_p) *size_p = e->dtype == MGX_F32_PURE ? 4 : 8;
    if (size_f) *size_f = e->dtype == MGX_F64 ? 8 : 4;
    return MGX_OK;
}
int mgx_engine_lanes_per_env(const mgx_engine *e) { return e ? e->L : 0; }
int mgx_engine_lds_bytes(const mgx_engine *e, int which) { return e ? (int)(which == 0 ? e->lds_step : e->lds_raster) : 0; }

static int reset_common(mgx_engine *e, void *state_p, void *state_f, int32_t *state_i, const uint8_t *mask, const void *ent_pose, void *stream) {
    if (!e || !state_p || !state_f || !state_i) return fail(MGX_ERR_ARG, "NULL argument");
    hipStream_t st = (hipStream_t)stream;
    size_t lds = (size_t)e->tdev.n_words * 4;
    int blocks = (e->n_envs + 63) / 64;
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
static int step_common(mgx_engine *e, void *sp, void *sf, int32_t *si, const int32_t *actions, uint8_t *done, int n_sub, int count_step, void *stream) {
    if (!e || !sp || !sf || !si || !actions) return fail(MGX_ERR_ARG, "NULL argument");
    if (n_sub < 0) return fail(MGX_ERR_ARG, "negative substep count");
    hipStream_t st = (hipStream_t)stream;
    int rc = timing_begin(e, 0, st);
    if (rc) return rc;
    rc = e->dtype == MGX_F32 ? launch_step<float, double>(e, sp, sf, si, actions, done, n_sub, count_step, st)
       : e->dtype == MGX_F64 ? launch_step<double, double>(e, sp, sf, si, actions, done, n_sub, count_step, st)
                             : launch_step<float, float>(e, sp, sf, si, actions, done, n_sub, count_step, st);
    if (rc) return rc;
    return timing_end(e, 0, st);
}
int mgx_engine_step(mgx_engine *e, void *state_p, void *state_f, int32_t *state_i, const int32_t *actions, uint8_t *done, void *stream) {
    return step_common(e, state_p, state_f, state_i, actions, done, PHYS_STEPS, 1, stream);
}
int mgx_engine_substeps(mgx_engine *e, void *state_p, void *state_f, int32_t *state_i, const int32_t *actions, int n_substeps, void *stream) {
    return step_common(e, state_p, state_f, state_i, actions, nullptr, n_substeps, 0, stream);
}

}  // extern "C"

template <typename P>
static int launch_raster(mgx_engine *e, const void *sp, uint8_t *out, int64_t env_stride, int view, int layout, const uint8_t *fill, hipStream_t st) {
    size_t lds = e->lds_raster;
    auto go = [&](auto kern) -> int {
        if (lds > 65536) HIP_OK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(e->n_envs), dim3(256), lds, st, e->rdev, (const P *)sp, out, (long)env_stride, view, fill, e->n_envs);
        return MGX_OK;
    };
    auto by_layout = [&](auto waves) -> int {
        constexpr int W = decltype(waves)::value;
        return layout == MGX_OBS_FRAME ? go(k_raster<P, 0, W>) : layout == MGX_OBS_STACK4 ? go(k_raster<P, 1, W>)
             : layout == MGX_OBS_STACK3_HI ? go(k_raster<P, 2, W>) : go(k_raster<P, 3, W>);
    };
    int rc = e->raster_waves >= 5 ? by_layout(std::integral_constant<int, 5>{})
           : e->raster_waves == 4 ? by_layout(std::integral_constant<int, 4>{}) : by_layout(std::integral_constant<int, 3>{});
    if (rc) return rc;
    HIP_OK(hipGetLastError());
    return MGX_OK;
}
extern "C" {

int mgx_engine_render(mgx_engine *e, const void *state_p, uint8_t *out, int64_t env_stride, int view, int layout,
                      const uint8_t *fill_mask, void *stream) {
    if (!e || !state_p || !out) return fail(MGX_ERR_ARG, "NULL argument");
    if (view != MGX_VIEW_EGO && view != MGX_VIEW_ALLO) return fail(MGX_ERR_ARG, "bad view");
    if (layout < MGX_OBS_FRAME || layout > MGX_OBS_SLOT_LO) return fail(MGX_ERR_ARG, "bad layout");
    int64_t need = (int64_t)LORES * LORES * (layout == MGX_OBS_FRAME ? 3 : 12);
    if (env_stride < need || (env_stride & 3)) return fail(MGX_ERR_ARG, "env_stride too small or not a multiple of 4");
    if (e->h.n_prims > 64) return fail(MGX_ERR_CAPACITY, "draw list longer than 64 primitives");
    if (e->lds_raster > (size_t)MAX_LDS_BYTES) return fail(MGX_ERR_CAPACITY, "draw list does not fit LDS");
    hipStream_t st = (hipStream_t)stream;
    int rc = timing_begin(e, 1, st);
    if (rc) return rc;
    rc = e->dtype == MGX_F32_PURE ? launch_raster<float>(e, state_p, out, env_stride, view, layout, fill_mask, st)
                                  : launch_raster<double>(e, state_p, out, env_stride, view, layout, fill_mask, st);
    if (rc) return rc;
    return timing_end(e, 1, st);
}
int mgx_engine_render_native(mgx_engine *e, const void *state_p, int env, uint8_t *out, int view, void *stream) {
    if (!e || !state_p || !out) return fail(MGX_ERR_ARG, "NULL argument");
    if (env < 0 || env >= e->n_envs) return fail(MGX_ERR_ARG, "env index out of range");
    hipStream_t st = (hipStream_t)stream;
    size_t lds = e->lds_raster;
    int blocks = (NATIVE_RES * NATIVE_RES + 255) / 256;
    if (e->dtype == MGX_F32_PURE) hipLaunchKernelGGL((k_raster_native<float>), dim3(blocks), dim3(256), lds, st, e->rdev, (const float *)state_p, out, view, (long)env, e->n_envs);
    else hipLaunchKernelGGL((k_raster_native<double>), dim3(blocks), dim3(256), lds, st, e->rdev, (const double *)state_p, out, view, (long)env, e->n_envs);
    HIP_OK(hipGetLastError());
    return MGX_OK;
}
int mgx_engine_debug_raster_ecap(mgx_engine *e, int n) { if (e) e->rdev.ecap = n < 1 ? 1 : (n > ECAP ? ECAP : n); return MGX_OK; }
int mgx_engine_debug_raster_qcap(mgx_engine *e, int n) { if (e) e->rdev.qcap = n < 1 ? 1 : (n > QCAP ? QCAP : n); return MGX_OK; }
int mgx_engine_debug_raster_stop(mgx_engine *e, int phase) { if (e) e->rdev.dbg_stop = phase; return MGX_OK; }
int mgx_engine_debug_raster_clocks(mgx_engine *e, void *buf) { if (e) e->rdev.dbg_clk = (unsigned long long *)buf; return MGX_OK; }
int mgx_engine_debug_step_clocks(mgx_engine *e, void *buf) { if (e) e->tdev.dbg_clk = (unsigned long long *)buf; return MGX_OK; }
int mgx_engine_debug_iterations(mgx_engine *e, int it) { if (e) e->dbg_iterations = it; return MGX_OK; }
int mgx_engine_set_prim_colours(mgx_engine *e, const int32_t *prim_rgb) {
    if (!e) return fail(MGX_ERR_ARG, "engine is NULL");
    e->rdev.prim_rgb_env = prim_rgb;
    return MGX_OK;
}
int mgx_engine_set_goal_rects(mgx_engine *e, const double *goal_xyhw) {
    if (!e) return fail(MGX_ERR_ARG, "engine is NULL");
    if (e->dtype == MGX_F32_PURE && goal_xyhw) return fail(MGX_ERR_ARG, "per-env goal rectangles need the fp64 pose type");
    e->rdev.goal_xyhw_env = goal_xyhw;
    return MGX_OK;
}
int mgx_engine_set_timing(mgx_engine *e, int enable) {
    if (!e) return fail(MGX_ERR_ARG, "engine is NULL");
    e->timing = enable < 0 ? 0 : enable;
    e->ev_count[0] = e->ev_count[1] = 0;
    e->launch_count[0] = e->launch_count[1] = 0;
    return MGX_OK;
}
int mgx_engine_timing_read(mgx_engine *e, int which, float *ms, int max) {
    if (!e || !ms || which < 0 || which > 1) return fail(MGX_ERR_ARG, "bad argument");
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

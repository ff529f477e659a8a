// tests/emu/mgx_emu.cpp -- HOST EMULATION OF THE KERNEL PHASES (test harness only).
//
// Compiles magical_amd/csrc/mgx_sim.h as plain C++ and runs the lane-parallel phases of the
// HIP step kernel sequentially (all lanes of phase k, then phase k+1), so the phase logic can
// be compared with the oracle on a machine without a GPU.  It is NOT a product fallback: the
// magical_amd package never loads this library, and the product path fails loudly without the
// HIP extension.
#include <cstdint>
#include <cstring>
#include <string>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../magical_amd/csrc/mgx_raster.h"
#include "../../magical_amd/csrc/mgx_sim.h"
#include "../../magical_amd/csrc/mgx_world.h"

using namespace mgx;

struct EmuBase {
    virtual ~EmuBase() {}
    TmplHeader h;
    std::vector<int32_t> ti;
    int n_envs;
    virtual void reset(void *sp, void *sf, int32_t *si, const uint8_t *mask) = 0;
    virtual void run(void *sp, void *sf, int32_t *si, const int32_t *actions, int n_sub, int nl, bool count_step, uint8_t *done) = 0;
    virtual int contacts(double *out, int max_rows) = 0;
};

template <typename R, typename P> struct Emu : EmuBase {
    std::vector<R> tr; std::vector<P> tp;
    std::vector<R> wr; std::vector<P> wp; std::vector<int32_t> wi;
    Emu(const World &w, int n) {
        n_envs = n;
        std::vector<double> rw, pw;
        w.serialise(h, ti, rw, pw);
        tr.resize(rw.size()); tp.resize(pw.size());
        for (size_t i = 0; i < rw.size(); i++) tr[i] = (R)rw[i];
        for (size_t i = 0; i < pw.size(); i++) tp[i] = (P)pw[i];
        WorkOff wo(h);
        wr.assign(wo.n_r, R(0)); wp.assign(wo.n_p, P(0)); wi.assign(wo.n_i, 0);
    }
    Env<R, P> env() { return Env<R, P>(&h, ti.data(), tr.data(), tp.data(), wr.data(), wp.data(), wi.data()); }
    void reset(void *sp, void *sf, int32_t *si, const uint8_t *mask) override {
        for (int e = 0; e < n_envs; e++)
            if (!mask || mask[e]) reset_env_state<R, P>(h, ti.data(), tr.data(), tp.data(), (P *)sp, (R *)sf, si, n_envs, e);
    }
    void run(void *sp_, void *sf_, int32_t *si, const int32_t *actions, int n_sub, int nl, bool count_step, uint8_t *done) override {
        P *sp = (P *)sp_; R *sf = (R *)sf_;
        const int iterations = 10;
        for (int ei = 0; ei < n_envs; ei++) {
            Env<R, P> e = env();
            std::vector<SolveCtx<R>> ctxs(nl);
            for (int lane = 0; lane < nl; lane++) ctxs[lane].row = &ctxs[lane & ~(ROW - 1)];     // what DPP row_newbcast reads on the device
#define RUN(stmt) for (int lane = 0; lane < nl; lane++) { SolveCtx<R> &ctx = ctxs[lane]; (void)ctx; stmt; }
            RUN(ph_init_work(e, lane, nl))
            RUN(ph_load_state(e, sp, sf, si, (long)n_envs, (long)ei, lane, nl))
            RUN(ph_refresh_trig(e, lane, nl))
            e.wi[e.wo.misc + M_ACTION] = actions[ei];
            ph_control(e);
            RUN(solve_ctx_init(e, ctx, lane, nl))
            for (int sub = 0; sub < n_sub; sub++) { MGX_SUBSTEP_PHASES(RUN) }
            if (count_step) {
                e.wi[e.wo.misc + M_STEPS] += 1;
                if (done) done[ei] = e.wi[e.wo.misc + M_STEPS] >= h.max_episode_steps ? 1 : 0;
            }
            RUN(solve_ctx_flush(e, ctx, lane, nl))
            RUN(ph_store_state(e, sp, sf, si, (long)n_envs, (long)ei, lane, nl))
#undef RUN
        }
    }
    // debug: contacts of the LAST env processed: rows [a, b, nx, ny, p1x, p1y, p2x, p2y, jn, jt, first]
    int contacts(double *out, int max_rows) override {
        Env<R, P> e = env();
        int nk = e.wi[e.wo.misc + M_NK];
        for (int k = 0; k < nk && k < max_rows; k++) {
            int ab = E_I(kab, k), a = ab & 0xFF, b = (ab >> 8) & 0xFF;
            double *o = out + 11 * k;
            o[0] = a; o[1] = b; o[2] = E_R(knx, k); o[3] = E_R(kny, k);
            o[4] = (double)E_P(px, a) + E_R(kr1x, k); o[5] = (double)E_P(py, a) + E_R(kr1y, k);
            o[6] = (double)E_P(px, b) + E_R(kr2x, k); o[7] = (double)E_P(py, b) + E_R(kr2y, k);
            o[8] = E_R(kjn, k); o[9] = E_R(kjt, k); o[10] = ab >> 16;
        }
        return nk;
    }
};

static long g_stat[8];
// host emulation of the raster kernel: same setup + per-pixel functions, tiles replaced by a plain loop
template <typename P>
static void emu_raster(const World &w, const P *sp, int n_envs, int env, int view, int native, uint8_t *out) {
    TmplHeader h; std::vector<int32_t> ti; std::vector<double> rw, pw;
    w.serialise(h, ti, rw, pw);
    TmplOff o(h); RasterOff ro(h);
    std::vector<double> d(ro.n_d, 0.0); std::vector<int32_t> iv(ro.n_i, 0);
    Raster rs(&h, ti.data(), rw.data() + o.prim_r, d.data(), iv.data(), view);
    const int nl = 7;   // odd lane count on purpose
    for (int lane = 0; lane < nl; lane++) raster_setup_bodies<P>(rs, sp, (long)n_envs, (long)env, lane, nl);
    for (int lane = 0; lane < nl; lane++) raster_setup_prims(rs, lane, nl);
    for (int lane = 0; lane < nl; lane++) raster_setup_edges(rs, lane, nl);
    // what the set-up takes from the world builder instead of computing it per lane (round 3): a primitive's first slot in the
    // front-to-back item list = the item counts of the primitives drawn after it, and an n-gon's cos(pi / n)
    for (int k = 0, after = 0; k < h.n_prims; k++) {
        after = 0;
        for (int kk = h.n_prims - 1; kk > k; kk--) after += prim_item_count(rs, kk);
        if (rs.prim_item_start(k) != after) { std::fprintf(stderr, "emu: item start of primitive %d is %d, expected %d\n", k, rs.prim_item_start(k), after); std::abort(); }
        if (rs.prim_kind(k) == PR_NGON && rs.prim_r(k, 4) != std::cos(3.14159265358979323846 / rs.prim_nv(k))) { std::fprintf(stderr, "emu: n-gon %d carries no cos(pi / n)\n", k); std::abort(); }
    }
    const int bg = 231 | (231 << 8) | (234 << 16);
    const uint64_t all = h.n_prims >= 64 ? ~0ull : ((1ull << h.n_prims) - 1ull);
    if (native) {
        for (int row = 0; row < NATIVE_RES; row++) for (int col = 0; col < NATIVE_RES; col++) {
            int c = raster_sample(rs, col + 0.5, (double)(NATIVE_RES - 1 - row) + 0.5, all, bg);
            uint8_t *q = out + 3 * (row * NATIVE_RES + col); q[0] = c & 0xFF; q[1] = (c >> 8) & 0xFF; q[2] = (c >> 16) & 0xFF;
        }
    } else {
        const int n_items = raster_total_items(rs);
        std::vector<Item> all_items(n_items);
        for (int i = 0; i < n_items; i++) all_items[i] = load_item(rs, i);
        for (int tile = 0; tile < TILES_X * TILES_Y; tile++) {
            // phase C: whole-tile classification against the full item list
            float txc, tyc; tile_centre(tile, txc, tyc);
            ClassState ts; ts.init(bg);
            classify_items_array<true>(rs, all_items.data(), n_items, txc, tyc, ts);
            g_stat[0]++; if (ts.mixed) { g_stat[1]++; g_stat[2] += __builtin_popcountll(ts.mixed); }
            // phase T: gather the undecided prims' items (slot order == what the lanes of a wave would hold)
            std::vector<Item> items;
            if (ts.mixed) {
                int n_total = 0;
                masked_item_index(rs, ts.mixed, 0, n_total);
                for (int slot = 0; slot < n_total; slot++) { int nt; items.push_back(load_item(rs, masked_item_index(rs, ts.mixed, slot, nt))); }
            }
            const int tcol = tile % TILES_X, trow = tile / TILES_X;
            for (int ty = 0; ty < TILE_H; ty++) for (int tx = 0; tx < TILE_W; tx++) {
                int X = tcol * TILE_W + tx, Y = trow * TILE_H + ty;
                int c = ts.base;
                if (ts.mixed) {
                    ClassState st; st.init(ts.base);
                    classify_items_array<false>(rs, items.data(), (int)items.size(), 4.0f * X + 2.0f, (float)NATIVE_RES - 4.0f * Y - 2.0f, st);
                    g_stat[3]++;
                    c = st.base;
                    if (st.mixed) { g_stat[4]++; c = pixel_resolve(rs, X, Y, st.mixed, c); }   // phase Q
                }
                uint8_t *q = out + 3 * (Y * LORES + X); q[0] = c & 0xFF; q[1] = (c >> 8) & 0xFF; q[2] = (c >> 16) & 0xFF;
            }
        }
    }
}

struct EmuHandle { World world; EmuBase *emu[3] = {nullptr, nullptr, nullptr}; };   // 0: f32/f32, 1: f32 + f64 poses, 2: f64

extern "C" {
void *emu_world_new() { return new EmuHandle(); }
void emu_free(void *p) { EmuHandle *h = (EmuHandle *)p; for (auto *e : h->emu) delete e; delete h; }
void emu_add_robot(void *p, double x, double y, double a) { EntityDef e{}; e.kind = 0; e.x = x; e.y = y; e.angle = a; ((EmuHandle *)p)->world.entities.push_back(e); }
void emu_add_shape(void *p, int st, int col, double x, double y, double a) { EntityDef e{}; e.kind = 1; e.shape_type = st; e.colour = col; e.x = x; e.y = y; e.angle = a; ((EmuHandle *)p)->world.entities.push_back(e); }
void emu_add_goal(void *p, double x, double y, double hh, double ww, int col) { EntityDef e{}; e.kind = 2; e.x = x; e.y = y; e.h = hh; e.w = ww; e.colour = col; ((EmuHandle *)p)->world.entities.push_back(e); }
int emu_finalize(void *p, int max_steps, int n_envs) {
    EmuHandle *h = (EmuHandle *)p; std::string err;
    int rc = h->world.finalize(max_steps, err);
    if (rc) return rc;
    h->emu[0] = new Emu<float, float>(h->world, n_envs);
    h->emu[1] = new Emu<float, double>(h->world, n_envs);
    h->emu[2] = new Emu<double, double>(h->world, n_envs);
    return 0;
}
int emu_rows(void *p, int which) {
    const TmplHeader &h = ((EmuHandle *)p)->emu[0]->h;
    return which == 0 ? state_rows_p(h) : (which == 1 ? state_rows_f(h) : state_rows_i(h));
}
int emu_n_state(void *p) { return ((EmuHandle *)p)->emu[0]->h.n_state; }
int emu_state_row(void *p, int row) { EmuHandle *h = (EmuHandle *)p; TmplOff o(h->emu[0]->h); return h->emu[0]->ti[o.state_map + row]; }
int emu_n_bodies(void *p) { return ((EmuHandle *)p)->emu[0]->h.n_bodies; }
int emu_contacts(void *p, int mode, double *out, int max_rows) { return ((EmuHandle *)p)->emu[mode]->contacts(out, max_rows); }
void emu_rstats(long *out) { for (int i = 0; i < 16; i++) { out[i] = mgx::g_rstat[i]; mgx::g_rstat[i] = 0; } }
void emu_stats(long *out) { for (int i = 0; i < 8; i++) { out[i] = g_stat[i]; g_stat[i] = 0; } }
void emu_render(void *p, int mode, const void *sp, int env, int view, int native, uint8_t *out) {
    EmuHandle *h = (EmuHandle *)p;
    if (mode == 0) emu_raster<float>(h->world, (const float *)sp, h->emu[0]->n_envs, env, view, native, out);
    else emu_raster<double>(h->world, (const double *)sp, h->emu[0]->n_envs, env, view, native, out);
}
void emu_reset(void *p, int mode, void *sp, void *sf, int32_t *si, const uint8_t *mask) { ((EmuHandle *)p)->emu[mode]->reset(sp, sf, si, mask); }
void emu_run(void *p, int mode, void *sp, void *sf, int32_t *si, const int32_t *actions, int n_sub, int nl, int count_step, uint8_t *done) {
    ((EmuHandle *)p)->emu[mode]->run(sp, sf, si, actions, n_sub, nl, count_step != 0, done);
}
}

// mgx_sim.h -- one physics substep for ONE env, written as lane-parallel phases.
//
// The step order is Chipmunk2D's cpSpaceStep as the reference drives it
// (base_env.py:236-243): integrate positions -> refresh shapes ->
// broadphase -> narrowphase -> arbiter update/prestep -> joint prestep -> warm start ->
// 10 x {contacts in pair order, joints in insertion order}.
//
// An env is owned by a group of `nl` lanes of one wavefront.  Every phase below is called by
// all lanes of the group with (lane, nl); phases are separated by a workgroup barrier in
// mgx_step.hip.  State lives in LDS (the `Env` view) for the whole env-step; HBM is touched
// only at kernel entry/exit.  The narrowphase is a closed-form SAT/closest-feature solver
// (no GJK iteration, no recursion, fixed trip counts) producing the same minimum-separation
// axis, depth and clipped contact points as Chipmunk's GJK/EPA + ContactPoints.
//
// The same header compiles as plain C++ (tests/emu) so the phase logic can be checked against
// the oracle on CPU; that build is a test harness only and is never loaded by the product.
#pragma once
#include <math.h>
#include <stdint.h>

#include "mgx_tmpl.h"

namespace mgx {

// ---------------------------------------------------------------- scalar helpers
template <typename R> MGX_HD R r_sqrt(R x);
template <> MGX_HD float r_sqrt<float>(float x) { return sqrtf(x); }
template <> MGX_HD double r_sqrt<double>(double x) { return sqrt(x); }
template <typename R> MGX_HD void r_sincos(R a, R &s, R &c);
template <> MGX_HD void r_sincos<float>(float a, float &s, float &c) { s = sinf(a); c = cosf(a); }
// the library's sincos, whatever r_sincos<double> is below: the rasteriser's set-up keeps it (three call sites once per frame:
// inlining the short form there costs the 96-register variant 30 more spilled registers)
MGX_HD void r_sincos_lib(double a, double &s, double &c) { s = sin(a); c = cos(a); }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MGX_LIBM_SINCOS)
// cpvforangle for the pose type, once per body and substep.  The library's sincos spends ~220 fp64 instructions on one lane
// group's six angles (Payne-Hanek path and all); body angles stay within a few thousand radians, so: Cody-Waite reduction
// against pi/2 in three parts (exact products through fma) and the fdlibm kernels on [-pi/4, pi/4] -- < 1 ulp, ~45 instructions.
template <> MGX_HD void r_sincos<double>(double a, double &s, double &c) {
    const double n = __builtin_rint(a * 6.36619772367581382433e-01);
    double r = __builtin_fma(-n, 1.57079632679489655800e+00, a);
    r = __builtin_fma(-n, 6.12323399573676603587e-17, r);
    r = __builtin_fma(-n, -1.49738490485916983e-33, r);
    const double z = r * r;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double sr = __builtin_fma(r * z, ps, r);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double cr = w + (((1.0 - w) - hz) + z * (z * pc));
    const int q = (int)n;
    const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}
#else
template <> MGX_HD void r_sincos<double>(double a, double &s, double &c) { s = sin(a); c = cos(a); }
#endif
template <typename R> MGX_HD R r_abs(R x) { return x < R(0) ? -x : x; }
template <typename R> MGX_HD R r_min(R a, R b) { return a < b ? a : b; }
template <typename R> MGX_HD R r_max(R a, R b) { return a > b ? a : b; }
template <typename R> MGX_HD R r_clamp(R f, R lo, R hi) { return r_min(r_max(f, lo), hi); }
template <typename R> MGX_HD R r_clamp01(R f) { return r_max(R(0), r_min(f, R(1))); }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MGX_NO_FAST_MINMAX)
// One instruction each on the device (v_min_f32 / v_max_f32 / v_med3_f32) instead of compare + wait states + select: the same
// values for every finite input (lo <= hi wherever r_clamp is called); a wavefront's time is its instruction count.
// (fminf / fmaxf would add a canonicalising v_max x, x per operand that comes from memory; the median with an infinity does not)
template <> MGX_HD float r_min<float>(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, -__builtin_inff()); }
template <> MGX_HD float r_max<float>(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }
template <> MGX_HD float r_clamp<float>(float f, float lo, float hi) { return __builtin_amdgcn_fmed3f(f, lo, hi); }
template <> MGX_HD float r_clamp01<float>(float f) { return __builtin_amdgcn_fmed3f(f, 0.0f, 1.0f); }
#endif
template <typename R> MGX_HD R r_inf();
template <> MGX_HD float r_inf<float>() { return __builtin_inff(); }
template <> MGX_HD double r_inf<double>() { return __builtin_inf(); }
template <typename R> MGX_HD R r_tiny();
template <> MGX_HD float r_tiny<float>() { return 1.17549435e-38f; }
template <> MGX_HD double r_tiny<double>() { return 2.2250738585072014e-308; }
// non-contractable multiply / explicit fma: the pin-joint anchor expression must round
// identically at reset and at every prestep (zero-length PinJoint)
template <typename R> MGX_HD R r_mul_nc(R a, R b) {
#if defined(__HIP_DEVICE_COMPILE__)
    R r = a * b;
    asm volatile("" : "+v"(r));
    return r;
#else
    volatile R r = a * b;
    return r;
#endif
}
template <typename R> MGX_HD R r_add_nc(R a, R b) {
#if defined(__HIP_DEVICE_COMPILE__)
    R r = a + b;
    asm volatile("" : "+v"(r));
    return r;
#else
    volatile R r = a + b;
    return r;
#endif
}
// world position of body-local anchor (ax, ay): p + rot (x) a, with one fixed rounding sequence
template <typename R> MGX_HD void anchor_rot(R c, R s, R ax, R ay, R &rx, R &ry) {
    rx = r_add_nc(r_mul_nc(c, ax), -r_mul_nc(s, ay));
    ry = r_add_nc(r_mul_nc(c, ay), r_mul_nc(s, ax));
}

// 1 / x.  The fp32 device build takes v_rcp_f32 (1 ulp) instead of the correctly rounded division's ten instructions: the
// effective masses it feeds are rounded to fp32 anyway, like everything downstream of them.
template <typename R> MGX_HD R r_rcp(R x) { return R(1) / x; }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MGX_NO_FAST_MINMAX)
template <> MGX_HD float r_rcp<float>(float x) { return __builtin_amdgcn_rcpf(x); }
#endif
// length and inverse length of a pose-precision vector whose direction is consumed in fp32 (the pin joints' axis): v_rsq_f64 and
// one Newton step (relative error ~1e-15) instead of a correctly rounded fp64 sqrt and division (~55 fp64 instructions).  The
// all-fp64 build keeps the exact forms.  inv = 0 for a zero vector (cpvnormalize of cpvzero).
template <typename R, typename P> MGX_HD void p_len_inv(P d2, P &len, P &inv) {
    len = r_sqrt<P>(d2);
    inv = len != P(0) ? P(1) / len : P(0);
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MGX_NO_FAST_MINMAX)
template <> MGX_HD void p_len_inv<float, double>(double d2, double &len, double &inv) {
    double y = __builtin_amdgcn_rsq(d2);
    y = y * __builtin_fma(-0.5 * d2 * y, y, 1.5);
    const bool pos = d2 > 0.0;
    inv = pos ? y : 0.0;
    len = pos ? d2 * y : 0.0;
}
#endif

// cpvclamp(v, lim): v scaled back to length lim when longer.  The fp32 device build is branch-free with v_rsq_f32 (1 ulp)
// instead of a correctly rounded sqrt and division (37 instructions, and the solver runs it in every iteration for the
// robot's pivot and every sliding block's); a scale of exactly 1 leaves an unclamped vector as it is.
template <typename R> MGX_HD void clamp_len(R &x, R &y, R lim) {
    R l2 = x * x + y * y;
    if (l2 > lim * lim) { R sc = lim / (r_sqrt(l2) + r_tiny<R>()); x *= sc; y *= sc; }
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MGX_NO_FAST_MINMAX)
template <> MGX_HD void clamp_len<float>(float &x, float &y, float lim) {
    float l2 = x * x + y * y;
    float sc = l2 > lim * lim ? lim * __builtin_amdgcn_rsqf(l2) : 1.0f;
    x *= sc; y *= sc;
}
#endif

// ---------------------------------------------------------------- env view
// R: velocity / impulse / contact arithmetic type.  P: pose type (positions, angles, their
// sin/cos and the joint anchor separations).  The reference's zero-length PinJoints take their
// direction from the DIFFERENCE of two nearly equal world positions, so the
// shipped fp32 engine keeps P = double for those few quantities and R = float for everything
// else; P = R = double is the validation build.
template <typename R, typename P> struct Env {
    const TmplHeader *h;
    const int32_t *ti;   // template ints
    const R *tr;         // template reals
    const P *tp;         // template pose-precision reals
    R *wr;               // working set, real region
    P *wp;               // working set, pose region
    int32_t *wi;         // working set, int region
    TmplOff to;
    WorkOff wo;
    MGX_HD Env(const TmplHeader *h_, const int32_t *ti_, const R *tr_, const P *tp_, R *wr_, P *wp_, int32_t *wi_)
        : h(h_), ti(ti_), tr(tr_), tp(tp_), wr(wr_), wp(wp_), wi(wi_), to(*h_), wo(*h_) {}
    MGX_HD R cst(int k) const { return tr[to.consts + k]; }
};

#define E_R(field, i) e.wr[e.wo.field + (i) * WorkOff::S_##field]
#define E_P(field, i) e.wp[e.wo.field + (i) * WorkOff::S_##field]
#define E_I(field, i) e.wi[e.wo.field + (i) * WorkOff::S_##field]
#define E_OV(i) (reinterpret_cast<uint16_t *>(e.wi + e.wo.ov)[i])                 // overlap list: 16-bit candidate-pair numbers
#define E_MATCHED(i) (reinterpret_cast<uint8_t *>(e.wi + e.wo.cmatched)[i])     // cache entry matched this substep
#define T_R(field, i) e.tr[e.to.field + (i) * TmplOff::S_##field]
#define T_P(field, i) e.tp[e.to.field + (i) * TmplOff::S_##field]
#define T_I(field, i) e.ti[e.to.field + (i) * TmplOff::S_##field]

// ---------------------------------------------------------------- phase: action decode + Robot.update
// entities.py:148-190 (id = 9*[close] + 3*lr + ud), :439-457 (set_action), :459-479 (update)
template <typename R, typename P> MGX_HD void ph_control(Env<R, P> &e) {
    int action = E_I(misc, M_ACTION);
    int ud = action % 3, lr = (action / 3) % 3, grip = action / 9;
    R speed = ud == 1 ? e.cst(C_SPEED_FWD) : (ud == 2 ? -e.cst(C_SPEED_BACK) : R(0));
    R turn = lr == 1 ? e.cst(C_TURN) : (lr == 2 ? -e.cst(C_TURN) : R(0));
    R target_finger = grip == 0 ? e.cst(C_FINGER_OPEN) : -e.cst(C_FINGER_CLOSED);
    int rb = e.h->robot_body, cb = e.h->control_body;
    P ar = E_P(ang, rb);
    E_P(ang, cb) = ar + P(turn);
    // control.velocity = robot.rotation_vector.cpvrotate((0, speed))
    E_R(vx, cb) = -R(E_P(s, rb)) * speed;
    E_R(vy, cb) = R(E_P(c, rb)) * speed;
    for (int f = 0; f < 2; f++) {
        R side = f == 0 ? R(-1) : R(1);
        R rel = R(E_P(ang, e.h->finger_body[f]) - ar);
        R err = rel + side * target_finger;
        R rate = r_max(R(-1), r_min(R(1), err * R(10)));
        if (r_abs(rate) < R(1e-4)) rate = R(0);
        E_R(jrate, e.h->motor_joint[f]) = rate;
    }
}

// ---------------------------------------------------------------- phase: cpBodyUpdatePosition
template <typename R, typename P> MGX_HD void ph_integrate(Env<R, P> &e, int lane, int nl) {
    P dt = T_P(p_dt, 0);
    for (int b = lane; b < e.h->n_bodies; b += nl) {
        if (T_I(body_type, b) == BODY_STATIC) continue;
        E_P(px, b) += P(E_R(vx, b) + E_R(vbx, b)) * dt;
        E_P(py, b) += P(E_R(vy, b) + E_R(vby, b)) * dt;
        P a = E_P(ang, b) + P(E_R(w, b) + E_R(wb, b)) * dt;
        E_P(ang, b) = a;
        E_R(vbx, b) = R(0); E_R(vby, b) = R(0); E_R(wb, b) = R(0);
        P s, c;
        r_sincos<P>(a, s, c);
        E_P(c, b) = c; E_P(s, b) = s;
    }
}

// ---------------------------------------------------------------- phase: shape cache (cpShapeUpdate)
template <typename R, typename P> MGX_HD void ph_shapes(Env<R, P> &e, int lane, int nl) {
    for (int sh = lane; sh < e.h->n_shapes; sh += nl) {
        int b = T_I(shape_body, sh), vo = T_I(shape_voff, sh), nv = T_I(shape_nv, sh);
        R bx = R(E_P(px, b)), by = R(E_P(py, b)), c = R(E_P(c, b)), s = R(E_P(s, b)), rad = T_R(shape_r, sh);
        R l = r_inf<R>(), r = -r_inf<R>(), bo = r_inf<R>(), t = -r_inf<R>();
        // (the next vertex's local record is in flight while this one is transformed and stored: the loop would otherwise pay one
        // LDS round trip per vertex on a wavefront that has nothing else to run)
        R lx = T_R(lvx, vo), ly = T_R(lvy, vo), nx = T_R(lnx, vo), ny = T_R(lny, vo);
        for (int i = 0; i < nv; i++) {
            const int in = i + 1 < nv ? i + 1 : i;
            const R lx1 = T_R(lvx, vo + in), ly1 = T_R(lvy, vo + in), nx1 = T_R(lnx, vo + in), ny1 = T_R(lny, vo + in);
            R x = bx + (c * lx - s * ly), y = by + (c * ly + s * lx);
            E_R(wx, vo + i) = x; E_R(wy, vo + i) = y;
            E_R(wnx, vo + i) = c * nx - s * ny; E_R(wny, vo + i) = c * ny + s * nx;
            l = r_min(l, x); r = r_max(r, x); bo = r_min(bo, y); t = r_max(t, y);
            lx = lx1; ly = ly1; nx = nx1; ny = ny1;
        }
        E_R(bbl, sh) = l - rad; E_R(bbb, sh) = bo - rad; E_R(bbr, sh) = r + rad; E_R(bbt, sh) = t + rad;
    }
}

// ---------------------------------------------------------------- phase: broadphase
// The candidate list already encodes QueryReject's body/group filters; what is left per substep is the BB test, and the
// overlapping pairs compacted IN PAIR ORDER (= arbiter solve order).  Device: the group's lanes test nl consecutive pairs at a
// time; a wavefront ballot gives every lane the hits of its group, so a pair's place in the list is the running total plus the
// hits of the lanes before it -- one phase, no LDS counters, no flags (round 2 counted per lane chunk, synchronised, then wrote).
template <typename R, typename P> MGX_HD bool boxes_overlap(const Env<R, P> &e, int pr) {
    const int a = pr & 0xFF, b = pr >> 8;
    // (both boxes read before the first compare, the four results combined without short circuit: `&&` made four dependent LDS round
    // trips of them -- read two sides, compare, branch, read the next two -- on a wavefront with nothing else to run.  Round 5:
    // FindDupe's k_step 0.382 -> 0.365 ms, ClusterColour's 0.480 -> 0.471, profiles/r05_step_broadphase_ab.txt)
    const R la = E_R(bbl, a), ba = E_R(bbb, a), ra = E_R(bbr, a), ta = E_R(bbt, a);
    const R lb = E_R(bbl, b), bb = E_R(bbb, b), rb = E_R(bbr, b), tb = E_R(bbt, b);
    return (la <= rb) & (lb <= ra) & (ba <= tb) & (bb <= ta);
}
template <typename R, typename P> MGX_HD bool pair_boxes_overlap(const Env<R, P> &e, int p) { return boxes_overlap(e, T_I(pair, p)); }
#if MGX_BROAD_SAP
// Sort and sweep (north_star's broadphase; round 6, A/B against the list above: profiles/r06_step_broadphase_sap_ab.txt).  Three phases of the
// env's lane group, scratch in the manifold words (dead between ph_cache_commit and ph_narrow): (1) rank by count -- lane i's shape finds its
// place among the <= 32 min-x keys and leaves (l, r, b, t, shape) there; (2) sweep -- lane s walks the records behind its own while their
// left edge is not beyond its right one (early exit) and notes y-overlaps in a bit matrix, row = lower shape number; (3) emit -- row a
// masked by a's candidate partners (the template's filtered pair list as bit masks), in ascending (a, b) = candidate-pair = arbiter order.
// Same comparisons (<=) on the same box values as boxes_overlap: the list it leaves is the list form's, entry for entry.
#ifndef MGX_BROAD_SAP_MIN
#define MGX_BROAD_SAP_MIN 0          // fewer shapes than this: the list form
#endif
constexpr int SAP_REC = 5, SAP_ROWS = SAP_REC * 32, SAP_WORDS = SAP_ROWS + 32;      // R-typed words of scratch
MGX_HD bool broad_sap(const TmplHeader &h) {
    return h.n_shapes <= 32 && h.n_shapes >= MGX_BROAD_SAP_MIN && 10 * (manifold_slots(h) ? h.cache_slots : h.max_overlaps) >= SAP_WORDS;
}
template <typename R, typename P> MGX_HD void sap_or(uint32_t *p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicOr(p, v);
#else
    *p |= v;
#endif
}
template <typename R, typename P> MGX_HD void ph_broad_sort(Env<R, P> &e, int lane, int nl) {
    const int n = e.h->n_shapes;
    R *rec = &E_R(mn, 0);
    uint32_t *row = reinterpret_cast<uint32_t *>(&E_R(mn, SAP_ROWS));
    for (int i = lane; i < 32; i += nl) row[i] = 0;
    for (int i = lane; i < n; i += nl) {
        const R key = E_R(bbl, i), r = E_R(bbr, i), b = E_R(bbb, i), t = E_R(bbt, i);
        int rank = 0;
        for (int j = 0; j < n; j++) { const R kj = E_R(bbl, j); rank += (int)((kj < key) | ((kj == key) & (j < i))); }
        rec[SAP_REC * rank + 0] = key; rec[SAP_REC * rank + 1] = r; rec[SAP_REC * rank + 2] = b; rec[SAP_REC * rank + 3] = t; rec[SAP_REC * rank + 4] = (R)i;
    }
}
template <typename R, typename P> MGX_HD void ph_broad_sweep(Env<R, P> &e, int lane, int nl) {
    if (!broad_sap(*e.h)) return;
    const int n = e.h->n_shapes;
    const R *rec = &E_R(mn, 0);
    uint32_t *row = reinterpret_cast<uint32_t *>(&E_R(mn, SAP_ROWS));
    for (int s = lane; s < n; s += nl) {
        const R r_ = rec[SAP_REC * s + 1], b_ = rec[SAP_REC * s + 2], t_ = rec[SAP_REC * s + 3];
        const int i = (int)rec[SAP_REC * s + 4];
        // (the next record is read while this one is tested: the lone wavefront pays every dependent LDS round trip in full)
        int tn = s + 1 < n ? s + 1 : s;
        R nl_ = rec[SAP_REC * tn], nb_ = rec[SAP_REC * tn + 2], nt_ = rec[SAP_REC * tn + 3], ni_ = rec[SAP_REC * tn + 4];
        for (int t = s + 1; t < n; t++) {
            const R cl = nl_, cb = nb_, ct = nt_, ci = ni_;
            tn = t + 1 < n ? t + 1 : t;
            nl_ = rec[SAP_REC * tn]; nb_ = rec[SAP_REC * tn + 2]; nt_ = rec[SAP_REC * tn + 3]; ni_ = rec[SAP_REC * tn + 4];
            if (!(cl <= r_)) break;                    // sorted by left edge: nothing further on reaches back to this box
            if ((b_ <= ct) & (cb <= t_)) {
                const int j = (int)ci, a = i < j ? i : j, bb = i < j ? j : i;
                sap_or<R, P>(&row[a], 1u << bb);
            }
        }
    }
}
template <typename R, typename P> MGX_HD void ph_broad_emit(Env<R, P> &e, int lane, int nl) {
    if (!broad_sap(*e.h)) return;
    const int n = e.h->n_shapes, cap = e.h->max_overlaps;
    const uint32_t *row = reinterpret_cast<const uint32_t *>(&E_R(mn, SAP_ROWS));
    int before = 0, total = 0;              // overlapping candidate pairs in the rows before this lane's next one / in all rows
    int a_next = lane;
    for (int a = 0; a < n; a++) {
        const uint32_t allow = (uint32_t)T_I(pair_allow, a);
        uint32_t hits = row[a] & allow;
        if (a == a_next) {
            int pos = before;
            const int first = T_I(pair_row, a);
            for (uint32_t h = hits; h; h &= h - 1) {
                const int b = __builtin_ctz(h);
                if (pos < cap) E_OV(pos) = (uint16_t)(first + __builtin_popcount(allow & ((1u << b) - 1u)));
                pos++;
            }
            a_next += nl;
        }
        before += __builtin_popcount(hits);
    }
    total = before;
    if (lane == 0) {
        if (total > cap) { E_I(misc, M_OVERFLOW) += 1; total = cap; }
        E_I(misc, M_NOV) = total;
        E_I(misc, M_NMAN) = 0;
    }
}
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// the list form, U rounds of nl candidate pairs per turn of the loop: their pair words are read a turn ahead and their box reads go out back
// to back, so that the lone wavefront waits for LDS once per turn, not once per round (round 5: U = 1; profiles/r06_step_broadphase_sap_ab.txt:
// ClusterColour's ph_broad 105 k -> 84 k (U = 2) -> 75 k (U = 4) cycles per env-step, MoveToCorner's 13.8 k -> 11.3 k -> 18.0 k)
template <int U, typename R, typename P> __device__ __forceinline__ int broad_list(Env<R, P> &e, int lane, int nl, int np, int cap) {
    // (n_pairs is the same for every lane of the wavefront: envs of one world, or one env per wavefront)
    const int wave_lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int shift = wave_lane - lane;
    const unsigned long long group_mask = nl >= 64 ? ~0ull : ((1ull << nl) - 1ull);
    int total = 0;
    int pr_next[U];
#pragma unroll
    for (int u = 0; u < U; u++) pr_next[u] = lane + u * nl < np ? T_I(pair, lane + u * nl) : 0;
    for (int base = 0; base < np; base += U * nl) {
        int pr[U]; bool hit[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int p = base + u * nl + lane;
            pr[u] = pr_next[u];
            pr_next[u] = p + U * nl < np ? T_I(pair, p + U * nl) : 0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) hit[u] = boxes_overlap(e, pr[u]) & (base + u * nl + lane < np);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const unsigned long long hits = (__builtin_amdgcn_ballot_w64(hit[u]) >> shift) & group_mask;
            const int pos = total + __builtin_popcountll(hits & ((1ull << lane) - 1ull));
            if (hit[u] && pos < cap) E_OV(pos) = (uint16_t)(base + u * nl + lane);
            total += __builtin_popcountll(hits);
        }
    }
    return total;
}
#endif
#ifndef MGX_BROAD_UNROLL_BIG
#define MGX_BROAD_UNROLL_BIG 4       // rounds per turn where the list has at least MGX_BROAD_BIG_PAIRS candidate pairs per lane-group round ...
#endif
#ifndef MGX_BROAD_UNROLL_SMALL
#define MGX_BROAD_UNROLL_SMALL 2     // ... and elsewhere
#endif
template <typename R, typename P> MGX_HD void ph_broad(Env<R, P> &e, int lane, int nl) {
#if MGX_BROAD_SAP
    if (broad_sap(*e.h)) { ph_broad_sort(e, lane, nl); return; }
#endif
    const int np = e.h->n_pairs, cap = e.h->max_overlaps;
    int total = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    // (wave-uniform: a list of at least eight rounds takes the wider turn)
    total = np >= 8 * nl ? broad_list<MGX_BROAD_UNROLL_BIG>(e, lane, nl, np, cap) : broad_list<MGX_BROAD_UNROLL_SMALL>(e, lane, nl, np, cap);
#else
    // host emulation (lanes run one after the other): lane 0 does the whole list
    if (lane != 0) return;
    for (int p = 0; p < np; p++)
        if (pair_boxes_overlap(e, p)) { if (total < cap) E_OV(total) = (uint16_t)p; total++; }
#endif
    if (lane == 0) {
        if (total > cap) { E_I(misc, M_OVERFLOW) += 1; total = cap; }
        E_I(misc, M_NOV) = total;
        E_I(misc, M_NMAN) = 0;          // ph_narrow's manifold slots
    }
}

// ---------------------------------------------------------------- narrowphase helpers
// Poly-like shapes (convex polygons and 2-vertex segments) keep world verts (wx, wy) and the
// outward normal of edge (i-1 -> i) in (wnx, wny), cpPolyShape "planes" convention.
template <typename R> struct EdgeRef { R ax, ay, bx, by, r, nx, ny; int ha, hb; };

template <typename R, typename P> MGX_HD int support_index(const Env<R, P> &e, int vo, int nv, R nx, R ny) {
    R best = -r_inf<R>(); int idx = 0;
    for (int i = 0; i < nv; i++) {
        R d = E_R(wx, vo + i) * nx + E_R(wy, vo + i) * ny;
        if (d > best) { best = d; idx = i; }
    }
    return idx;
}
// Up to eight world vertices of a shape in registers: the loads go out back to back (one LDS round trip for the shape) and the loops
// over them unroll with static indices -- ph_narrow's inner loops were one round trip per vertex (a pair of squares: 52 of them, now 14).
// Slots past the shape's last vertex repeat vertex 0, which changes no minimum and wins no strict comparison.
#ifndef MGX_NARROW_REGS
#define MGX_NARROW_REGS 1
#endif
constexpr int NV_REGS = 8;
template <typename R> struct Verts8 { R x[NV_REGS], y[NV_REGS]; };
template <typename R, typename P> MGX_HD void load_verts8(const Env<R, P> &e, int vo, int nv, int j0, Verts8<R> &v) {
#pragma unroll
    for (int j = 0; j < NV_REGS; j++) { const int jj = j0 + j < nv ? j0 + j : 0; v.x[j] = E_R(wx, vo + jj); v.y[j] = E_R(wy, vo + jj); }
}
// min over the shape's vertices of f . v (the same products and sums, in vertex order, as the loop it replaces)
template <typename R, typename P> MGX_HD R min_proj(const Env<R, P> &e, int vo, int nv, const Verts8<R> &v, R fx, R fy) {
    R mn = r_inf<R>();
#pragma unroll
    for (int j = 0; j < NV_REGS; j++) mn = r_min(mn, fx * v.x[j] + fy * v.y[j]);
    for (int j0 = NV_REGS; j0 < nv; j0 += NV_REGS) {       // (no shape of the benchmark suite has more than eight vertices)
        Verts8<R> t; load_verts8(e, vo, nv, j0, t);
#pragma unroll
        for (int j = 0; j < NV_REGS; j++) mn = r_min(mn, fx * t.x[j] + fy * t.y[j]);
    }
    return mn;
}
template <typename R, typename P> MGX_HD int support_index_regs(const Env<R, P> &e, int vo, int nv, const Verts8<R> &v, R nx, R ny) {
    if (nv > NV_REGS) return support_index(e, vo, nv, nx, ny);
    R best = -r_inf<R>(); int idx = 0;
#pragma unroll
    for (int i = 0; i < NV_REGS; i++) {
        R d = v.x[i] * nx + v.y[i] * ny;
        if (d > best) { best = d; idx = i; }
    }
    return idx;
}
// cpCollision.c SupportEdgeForPoly / SupportEdgeForSegment
template <typename R, typename P> MGX_HD EdgeRef<R> support_edge(const Env<R, P> &e, int vo, int nv, R rad, R nx, R ny, int i1 = -1) {
    if (i1 < 0) i1 = support_index(e, vo, nv, nx, ny);
    int i0 = (i1 - 1 + nv) % nv, i2 = (i1 + 1) % nv;
    R d1 = nx * E_R(wnx, vo + i1) + ny * E_R(wny, vo + i1);
    R d2 = nx * E_R(wnx, vo + i2) + ny * E_R(wny, vo + i2);
    EdgeRef<R> ed;
    int ia, ib, in;
    if (d1 > d2) { ia = i0; ib = i1; in = i1; } else { ia = i1; ib = i2; in = i2; }
    ed.ax = E_R(wx, vo + ia); ed.ay = E_R(wy, vo + ia); ed.bx = E_R(wx, vo + ib); ed.by = E_R(wy, vo + ib);
    ed.nx = E_R(wnx, vo + in); ed.ny = E_R(wny, vo + in); ed.r = rad; ed.ha = ia; ed.hb = ib;
    return ed;
}

template <typename R> struct ManifoldOut {
    int count; R nx, ny; R p[8]; int h0, h1;
    MGX_HD void push(R p1x, R p1y, R p2x, R p2y, int hash) {
        // (no run-time index into p: the array would live in scratch memory)
        if (count == 0) { p[0] = p1x; p[1] = p1y; p[2] = p2x; p[3] = p2y; h0 = hash; }
        else { p[4] = p1x; p[5] = p1y; p[6] = p2x; p[7] = p2y; h1 = hash; }
        count++;
    }
};

// cpCollision.c ContactPoints: clip the two support edges against each other along n
template <typename R>
MGX_HD void contact_points(const EdgeRef<R> &e1, const EdgeRef<R> &e2, R nx, R ny, R d, ManifoldOut<R> &m) {
    R mindist = e1.r + e2.r;
    if (!(d <= mindist)) return;
    m.nx = nx; m.ny = ny;
    R d_e1_a = e1.ax * ny - e1.ay * nx, d_e1_b = e1.bx * ny - e1.by * nx;
    R d_e2_a = e2.ax * ny - e2.ay * nx, d_e2_b = e2.bx * ny - e2.by * nx;
    R e1_denom = R(1) / (d_e1_b - d_e1_a + r_tiny<R>());
    R e2_denom = R(1) / (d_e2_b - d_e2_a + r_tiny<R>());
    {
        R t1 = r_clamp01((d_e2_b - d_e1_a) * e1_denom), t2 = r_clamp01((d_e1_a - d_e2_a) * e2_denom);
        R p1x = nx * e1.r + (e1.ax * (R(1) - t1) + e1.bx * t1), p1y = ny * e1.r + (e1.ay * (R(1) - t1) + e1.by * t1);
        R p2x = -nx * e2.r + (e2.ax * (R(1) - t2) + e2.bx * t2), p2y = -ny * e2.r + (e2.ay * (R(1) - t2) + e2.by * t2);
        R dist = (p2x - p1x) * nx + (p2y - p1y) * ny;
        if (dist <= R(0)) m.push(p1x, p1y, p2x, p2y, 1 + e1.ha * 8 + e2.hb);
    }
    {
        R t1 = r_clamp01((d_e2_a - d_e1_a) * e1_denom), t2 = r_clamp01((d_e1_b - d_e2_a) * e2_denom);
        R p1x = nx * e1.r + (e1.ax * (R(1) - t1) + e1.bx * t1), p1y = ny * e1.r + (e1.ay * (R(1) - t1) + e1.by * t1);
        R p2x = -nx * e2.r + (e2.ax * (R(1) - t2) + e2.bx * t2), p2y = -ny * e2.r + (e2.ay * (R(1) - t2) + e2.by * t2);
        R dist = (p2x - p1x) * nx + (p2y - p1y) * ny;
        if (dist <= R(0)) m.push(p1x, p1y, p2x, p2y, 1 + e1.hb * 8 + e2.ha);
    }
}

// closest point on segment (ax,ay)-(bx,by) to p; returns squared distance, t in [0,1]
template <typename R> MGX_HD R point_segment(R px, R py, R ax, R ay, R bx, R by, R &t, R &cx, R &cy) {
    R dx = bx - ax, dy = by - ay;
    R l2 = dx * dx + dy * dy;
    t = r_clamp01(((px - ax) * dx + (py - ay) * dy) / l2);
    cx = ax + dx * t; cy = ay + dy * t;
    R ex = px - cx, ey = py - cy;
    return ex * ex + ey * ey;
}

// Minimum-separation axis between two poly-like shapes: returns (n from A to B, signed distance d).
// Overlapping cores: SAT over the face normals of both (== EPA's closest Minkowski edge).
// Separated cores: the exact closest feature pair (== GJK), which matters only inside the
// radius band of bevelled / thick shapes.
template <typename R, typename P>
MGX_HD bool poly_axis(const Env<R, P> &e, int voa, int na, int vob, int nb, R rsum, R &nx, R &ny, R &d, int &sup_a, int &sup_b) {
    R best = -r_inf<R>(); int best_i = 0; bool best_a = true;
    sup_a = -1; sup_b = -1;
#if MGX_NARROW_REGS
    Verts8<R> vb;
    load_verts8(e, vob, nb, 0, vb);
    for (int i = 0; i < na; i++) {
        R fx = E_R(wnx, voa + i), fy = E_R(wny, voa + i);
        R off = fx * E_R(wx, voa + i) + fy * E_R(wy, voa + i);
        R sep = min_proj(e, vob, nb, vb, fx, fy) - off;
        if (sep > best) { best = sep; best_i = i; best_a = true; }
    }
    if (best > rsum) return false;
    Verts8<R> va;
    load_verts8(e, voa, na, 0, va);
    for (int j = 0; j < nb; j++) {
        R fx = E_R(wnx, vob + j), fy = E_R(wny, vob + j);
        R off = fx * E_R(wx, vob + j) + fy * E_R(wy, vob + j);
        R sep = min_proj(e, voa, na, va, fx, fy) - off;
        if (sep > best) { best = sep; best_i = j; best_a = false; }
    }
    if (best > rsum) return false;
#else
    for (int i = 0; i < na; i++) {
        R fx = E_R(wnx, voa + i), fy = E_R(wny, voa + i);
        R off = fx * E_R(wx, voa + i) + fy * E_R(wy, voa + i);
        R mn = r_inf<R>();
        for (int j = 0; j < nb; j++) mn = r_min(mn, fx * E_R(wx, vob + j) + fy * E_R(wy, vob + j));
        R sep = mn - off;
        if (sep > best) { best = sep; best_i = i; best_a = true; }
    }
    if (best > rsum) return false;
    for (int j = 0; j < nb; j++) {
        R fx = E_R(wnx, vob + j), fy = E_R(wny, vob + j);
        R off = fx * E_R(wx, vob + j) + fy * E_R(wy, vob + j);
        R mn = r_inf<R>();
        for (int i = 0; i < na; i++) mn = r_min(mn, fx * E_R(wx, voa + i) + fy * E_R(wy, voa + i));
        R sep = mn - off;
        if (sep > best) { best = sep; best_i = j; best_a = false; }
    }
    if (best > rsum) return false;
#endif
    if (best_a) { nx = E_R(wnx, voa + best_i); ny = E_R(wny, voa + best_i); }
    else { nx = -E_R(wnx, vob + best_i); ny = -E_R(wny, vob + best_i); }
    d = best;
#if MGX_NARROW_REGS   // the vertices the caller's two support edges start from, while both shapes are in registers
    sup_a = support_index_regs(e, voa, na, va, nx, ny); sup_b = support_index_regs(e, vob, nb, vb, -nx, -ny);
#endif
    if (best <= R(0)) return true;
    // separated cores inside the radius band: check the feature pair is vertex/edge, else go exact
    {
        int vo_f = best_a ? voa : vob, n_f = best_a ? na : nb, vo_o = best_a ? vob : voa, n_o = best_a ? nb : na;
        R fx = E_R(wnx, vo_f + best_i), fy = E_R(wny, vo_f + best_i);
        int k = 0; R mn = r_inf<R>();
        for (int j = 0; j < n_o; j++) { R v = fx * E_R(wx, vo_o + j) + fy * E_R(wy, vo_o + j); if (v < mn) { mn = v; k = j; } }
        int i0 = (best_i - 1 + n_f) % n_f;
        R ax = E_R(wx, vo_f + i0), ay = E_R(wy, vo_f + i0), bx = E_R(wx, vo_f + best_i), by = E_R(wy, vo_f + best_i);
        R dx = bx - ax, dy = by - ay;
        R t = ((E_R(wx, vo_o + k) - ax) * dx + (E_R(wy, vo_o + k) - ay) * dy) / (dx * dx + dy * dy);
        if (t >= R(0) && t <= R(1)) return true;
    }
    // exact closest features (vertex/vertex region): brute force both directions
    R bd2 = r_inf<R>(), pax = 0, pay = 0, pbx = 0, pby = 0;
    for (int i = 0; i < na; i++) {
        int i0 = (i - 1 + na) % na;
        R ax = E_R(wx, voa + i0), ay = E_R(wy, voa + i0), bx = E_R(wx, voa + i), by = E_R(wy, voa + i);
        for (int j = 0; j < nb; j++) {
            R t, cx, cy, qx = E_R(wx, vob + j), qy = E_R(wy, vob + j);
            R d2 = point_segment(qx, qy, ax, ay, bx, by, t, cx, cy);
            if (d2 < bd2) { bd2 = d2; pax = cx; pay = cy; pbx = qx; pby = qy; }
        }
    }
    for (int j = 0; j < nb; j++) {
        int j0 = (j - 1 + nb) % nb;
        R ax = E_R(wx, vob + j0), ay = E_R(wy, vob + j0), bx = E_R(wx, vob + j), by = E_R(wy, vob + j);
        for (int i = 0; i < na; i++) {
            R t, cx, cy, qx = E_R(wx, voa + i), qy = E_R(wy, voa + i);
            R d2 = point_segment(qx, qy, ax, ay, bx, by, t, cx, cy);
            if (d2 < bd2) { bd2 = d2; pax = qx; pay = qy; pbx = cx; pby = cy; }
        }
    }
    R dist = r_sqrt(bd2);
    if (dist > rsum) return false;
    R inv = R(1) / (dist + r_tiny<R>());
    nx = (pbx - pax) * inv; ny = (pby - pay) * inv; d = dist;
    sup_a = -1; sup_b = -1;                     // (another axis: the caller looks its support vertices up)
    return true;
}

template <typename R, typename P> MGX_HD void collide_pair(const Env<R, P> &e, int sa, int sb, ManifoldOut<R> &m) {
    m.count = 0; m.h0 = 0; m.h1 = 0; m.nx = 0; m.ny = 0;
    int ka = T_I(shape_kind, sa), kb = T_I(shape_kind, sb);
    int voa = T_I(shape_voff, sa), vob = T_I(shape_voff, sb);
    int na = T_I(shape_nv, sa), nb = T_I(shape_nv, sb);
    R ra = T_R(shape_r, sa), rb = T_R(shape_r, sb);
    if (ka == SH_CIRCLE && kb == SH_CIRCLE) {                 // CircleToCircle
        R cx1 = E_R(wx, voa), cy1 = E_R(wy, voa), cx2 = E_R(wx, vob), cy2 = E_R(wy, vob);
        R mind = ra + rb, dx = cx2 - cx1, dy = cy2 - cy1, d2 = dx * dx + dy * dy;
        if (d2 < mind * mind) {
            R dist = r_sqrt(d2);
            R nx = dist != R(0) ? dx / dist : R(1), ny = dist != R(0) ? dy / dist : R(0);
            m.nx = nx; m.ny = ny;
            m.push(cx1 + nx * ra, cy1 + ny * ra, cx2 - nx * rb, cy2 - ny * rb, 0);
        }
    } else if (ka == SH_CIRCLE && kb == SH_SEGMENT) {         // CircleToSegment
        R cx = E_R(wx, voa), cy = E_R(wy, voa);
        R t, qx, qy;
        R d2 = point_segment(cx, cy, E_R(wx, vob), E_R(wy, vob), E_R(wx, vob + 1), E_R(wy, vob + 1), t, qx, qy);
        R mind = ra + rb;
        if (d2 < mind * mind) {
            R dist = r_sqrt(d2);
            R nx = dist != R(0) ? (qx - cx) / dist : E_R(wnx, vob + 1);
            R ny = dist != R(0) ? (qy - cy) / dist : E_R(wny, vob + 1);
            m.nx = nx; m.ny = ny;
            m.push(cx + nx * ra, cy + ny * ra, qx - nx * rb, qy - ny * rb, 0);
        }
    } else if (ka == SH_CIRCLE) {                             // CircleToPoly
        R cx = E_R(wx, voa), cy = E_R(wy, voa);
        // signed distances to the face planes; inside if all <= 0
        R smax = -r_inf<R>(); int imax = 0;
        for (int i = 0; i < nb; i++) {
            R s = E_R(wnx, vob + i) * (cx - E_R(wx, vob + i)) + E_R(wny, vob + i) * (cy - E_R(wy, vob + i));
            if (s > smax) { smax = s; imax = i; }
        }
        R nx, ny, d, qx, qy;
        if (smax <= R(0)) {                                   // centre inside: closest face (EPA result)
            nx = -E_R(wnx, vob + imax); ny = -E_R(wny, vob + imax); d = smax;
            qx = cx - E_R(wnx, vob + imax) * smax; qy = cy - E_R(wny, vob + imax) * smax;
        } else {                                              // outside: closest boundary point (GJK result)
            R bd2 = r_inf<R>(), bt = 0; int bi = 0; qx = 0; qy = 0;
            for (int i = 0; i < nb; i++) {
                int i0 = (i - 1 + nb) % nb;
                R t, px, py;
                R d2 = point_segment(cx, cy, E_R(wx, vob + i0), E_R(wy, vob + i0), E_R(wx, vob + i), E_R(wy, vob + i), t, px, py);
                if (d2 < bd2) { bd2 = d2; qx = px; qy = py; bt = t; bi = i; }
            }
            if (bt > R(0) && bt < R(1)) {                     // edge interior: axis is the face normal
                nx = -E_R(wnx, vob + bi); ny = -E_R(wny, vob + bi);
                d = (qx - cx) * nx + (qy - cy) * ny;
            } else {
                d = r_sqrt(bd2);
                R inv = R(1) / (d + r_tiny<R>());
                nx = (qx - cx) * inv; ny = (qy - cy) * inv;
            }
        }
        if (d <= ra + rb) {
            m.nx = nx; m.ny = ny;
            m.push(cx + nx * ra, cy + ny * ra, qx - nx * rb, qy - ny * rb, 0);
        }
    } else {                                                  // SegmentToPoly / PolyToPoly
        R nx, ny, d;
        int sup_a, sup_b;
        if (poly_axis(e, voa, na, vob, nb, ra + rb, nx, ny, d, sup_a, sup_b)) {
            EdgeRef<R> e1 = support_edge(e, voa, na, ra, nx, ny, sup_a);
            EdgeRef<R> e2 = support_edge(e, vob, nb, rb, -nx, -ny, sup_b);
            contact_points(e1, e2, nx, ny, d, m);
        }
    }
}

// a counter in the env's LDS working set taken by lanes side by side (the host emulation runs the lanes one after the other)
MGX_HD int lds_fetch_inc(int32_t *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(p, 1);
#else
    return (*p)++;
#endif
}
// ---------------------------------------------------------------- phase: narrowphase
template <typename R, typename P> MGX_HD void ph_narrow(Env<R, P> &e, int lane, int nl) {
    int nov = E_I(misc, M_NOV);
    // Manifold slots of the crowded worlds (manifold_slots, mgx_tmpl.h): a touching pair takes the next slot IN PAIR ORDER -- the running
    // total plus the touching pairs of the lanes before it in this round, from a wavefront ballot like ph_broad's list (round 5 took them
    // off an LDS counter: an atomic round trip per touching pair, and under overflow -- more touching pairs than arbiters can be, counted
    // in M_OVERFLOW -- the pairs that kept a slot depended on lane order, i.e. on lanes_per_env; now the highest pairs are dropped, whatever
    // the group's width, as include/mgx.h promises).  The host emulation runs the lanes one after the other: lane 0 walks the whole list.
    const bool slots = manifold_slots(*e.h);
    int total = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    const int wave_lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int shift = wave_lane - lane;
    const unsigned long long group_mask = nl >= 64 ? ~0ull : ((1ull << nl) - 1ull);
    for (int q = lane; q < nov; q += nl) {
#else
    if (slots && lane != 0) return;
    for (int q = slots ? 0 : lane; q < nov; q += slots ? 1 : nl) {
#endif
        int pr = T_I(pair, (int)E_OV(q));
        ManifoldOut<R> m;
        collide_pair(e, pr & 0xFF, pr >> 8, m);
        // point count (0..2) | the two point hashes << 8 | the manifold's slot << 24; more touching pairs than arbiters can be is an
        // overflow like any other
        int word = 0;
        int slot = q;
        if (slots) {
#if defined(__HIP_DEVICE_COMPILE__)
            const unsigned long long hits = (__builtin_amdgcn_ballot_w64(m.count > 0) >> shift) & group_mask;
            slot = total + __builtin_popcountll(hits & ((1ull << lane) - 1ull));
            total += __builtin_popcountll(hits);
#else
            slot = total;
            total += m.count > 0;
#endif
        }
        if (m.count > 0) {
            if (!slots || slot < e.h->cache_slots) {
                word = m.count | ((m.h0 | (m.h1 << 8)) << 8) | (slot << 24);
                E_R(mn, 2 * slot) = m.nx; E_R(mn, 2 * slot + 1) = m.ny;
                E_R(mp, 8 * slot + 0) = m.p[0]; E_R(mp, 8 * slot + 1) = m.p[1]; E_R(mp, 8 * slot + 2) = m.p[2]; E_R(mp, 8 * slot + 3) = m.p[3];
                if (m.count > 1) { E_R(mp, 8 * slot + 4) = m.p[4]; E_R(mp, 8 * slot + 5) = m.p[5]; E_R(mp, 8 * slot + 6) = m.p[6]; E_R(mp, 8 * slot + 7) = m.p[7]; }
            } else {
                (void)lds_fetch_inc(&E_I(misc, M_OVERFLOW));
            }
        }
        E_I(mcnt, q) = word;
    }
}

// ---------------------------------------------------------------- solver context (registers) + joint preSteps
// Joints only couple bodies inside an ISLAND: the robot's ten joints (Robot.setup, entities.py:238-354: pivot + gear to the
// kinematic control body, two eye springs, {pin, limit, motor} per finger), and each block's {pivot, gear} to the static body
// (entities.py:703-711).  Inside the joint half of a Gauss-Seidel iteration islands can therefore run on different lanes and
// still apply, to every body, exactly the sequence of impulses Chipmunk's serial order applies.
//
// Everything the solver iterates on lives in registers, per lane of the env's group (rowlane = lane & 15; a group of 32 or
// 64 lanes is two or four rows that all carry the robot):
//  * "pg" columns: the lane's own {pivot, gear} pair to a body that impulses do not move -- rowlane 0: the robot's pair to
//    the control body; lanes 1 .. n_islands of the first row: one block's pair each.  Same instructions for all of them.
//  * the robot island, identical in every lane of the env: the velocities of robot / eyes / fingers, the eight joints 2..9 with
//    just the constants their impulse formulas need (33 values; the generic per-joint record of the first design was 12 x 10),
//    and their accumulated impulses, which stay in registers for the whole env-step.
// A joint's preStep runs on its OWNER lane (rowlane j for robot joint j, so both fingers' pins are one instruction stream)
// and the few values it produces are broadcast to the row once per substep (DPP row_newbcast); the 10 iterations then run
// without any cross-lane or LDS traffic unless the env has contacts.
constexpr int RI_JOINTS = 10;      // pivot gear spring spring {pin limit motor} x 2
constexpr int ROW = 16;            // a DPP row

template <typename R> struct SolveCtx {
    // pg pair (own lane)
    R pk0, pk1, pk2, pk3, pb0, pb1, pa0, pa1, plim;   // pivot: K^-1, bias, accumulated impulse, max impulse
    R gim, gb, ga, gratio, glim;                       // gear: effective mass, bias, accumulated impulse, ratio, max impulse
    R avx, avy, aw;                                    // velocity of the pair's first body (control body; 0 for the static body)
    R mvx, mvy, mw, mminv, miinv;                      // the pair's second body: this lane's robot / block
    int mbody, mj;                                     // its index / the pivot's joint index (-1: this lane has no pair)
    // owner lane's preStep results for robot joint `rowlane`, broadcast in solve_begin
    R o[6];
    // robot island (uniform over the env's lanes)
    R rvx, rvy, rw, ew[2], fvx[2], fvy[2], fw[2];
    R r_minv, r_iinv, e_iinv[2], f_minv[2], f_iinv[2];
    R s_im[2], s_coef[2];                              // eye springs: 1 / (i_a + i_b), damping coefficient
    R pin[2][6], pin_lim[2];                           // finger pins: r1x r1y nx ny n_mass bias | max impulse
    R lim_im[2], lim_b[2], lim_lo[2], lim_hi[2], lim_max[2];     // finger limits: effective mass, clamp range of the accumulated impulse
    R mot_im[2], mot_rate[2], mot_lim[2];              // finger motors
    R acc[8];                                          // springs' target_wrn (2) | per finger: pin, limit, motor impulses
    int b_robot, b_eye[2], b_finger[2];                // body indices (read from the header once per env-step, not before every access)
    int has_contacts;
#if !defined(__HIP_DEVICE_COMPILE__)
    const SolveCtx *row;                               // host emulation: the 16 contexts of this lane's row
#endif
};

// value of `field` in lane J of this lane's row
#if defined(__HIP_DEVICE_COMPILE__)
template <int J> MGX_HD float row_bcast(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x150 + J, 0xf, 0xf, false));
}
template <int J> MGX_HD double row_bcast(double x) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, 0x150 + J, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), 0x150 + J, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
#define RB(J, field) row_bcast<J>(c.field)
#else
#define RB(J, field) (c.row[J].field)
#endif

// the pair's pivot + gear preSteps (cpPivotJointPreStep / cpGearJointPreStep) on the lanes that own one; the preSteps of the
// robot's joints 2..9 on their owner lanes.  Anchor separations and angle differences are formed in pose precision.
template <typename R, typename P> MGX_HD void joints_prestep(Env<R, P> &e, SolveCtx<R> &c, int lane, int nl) {
    (void)nl;
    const int rowlane = lane & (ROW - 1);
    if (c.mj >= 0) {
        // both anchors of these pivots are body origins (Robot.setup's pair to the control body, entities.py:255-258; a block's pair
        // to the static body, :703-707): r1 = r2 = 0, K = (m_a^-1 + m_b^-1) I is a constant of the env-step (solve_ctx_init), and
        // the bias is the clamped position error -- zero whenever max_bias is 0, which it is in every world the reference builds
        const int j = c.mj, a = T_I(joint_a, j), b = c.mbody;
        const R *p = &T_R(joint_p, j * JOINT_PARAMS);
        R bx = R(0), by = R(0);
        if (p[8] > R(0)) {
            bx = -R(E_P(px, b) - E_P(px, a)) * p[7]; by = -R(E_P(py, b) - E_P(py, a)) * p[7];
            clamp_len(bx, by, p[8]);
        }
        c.pb0 = bx; c.pb1 = by;
        const R *pg = &T_R(joint_p, (j + 1) * JOINT_PARAMS);
        const P *ppg = &T_P(p_joint, (j + 1) * 7);
        c.gb = r_clamp(-R(E_P(ang, b) * ppg[5] - E_P(ang, a) - ppg[4]) * pg[7], -pg[8], pg[8]);
        c.avx = E_R(vx, a); c.avy = E_R(vy, a); c.aw = E_R(w, a);
    }
    const int j0 = e.h->robot_j0;
    if (rowlane == 2 || rowlane == 3) {                       // cpDampedRotarySpring preStep: the torque it applies
        const int jj = j0 + rowlane, a = T_I(joint_a, jj), b = T_I(joint_b, jj);
        c.o[0] = R((E_P(ang, a) - E_P(ang, b)) - T_P(p_joint, jj * 7 + 4)) * T_R(joint_p, jj * JOINT_PARAMS + 5);
    } else if (rowlane == 4 || rowlane == 7) {                // cpPinJointPreStep
        const int jj = j0 + rowlane, a = T_I(joint_a, jj), b = T_I(joint_b, jj);
        const R *p = &T_R(joint_p, jj * JOINT_PARAMS);
        const P *pp = &T_P(p_joint, jj * 7);
        // (the finger-side anchor is the finger's origin, entities.py:334-341: r2 = 0; the robot-side anchor position is formed
        // with the rounding sequence k_reset places the finger roots with, so the zero-length pin starts exact)
        P q1x, q1y;
        anchor_rot<P>(E_P(c, a), E_P(s, a), pp[0], pp[1], q1x, q1y);
        const R r1x = R(q1x), r1y = R(q1y);
        const P ddx = E_P(px, b) - r_add_nc<P>(E_P(px, a), q1x);
        const P ddy = E_P(py, b) - r_add_nc<P>(E_P(py, a), q1y);
        P dist, inv;
        p_len_inv<R, P>(ddx * ddx + ddy * ddy, dist, inv);
        const R nx = R(ddx * inv), ny = R(ddy * inv);
        const R rcn1 = r1x * ny - r1y * nx;
        const R ma = T_R(body_minv, a), ia = T_R(body_iinv, a), mb = T_R(body_minv, b);
        c.o[0] = r1x; c.o[1] = r1y; c.o[2] = nx; c.o[3] = ny;
        c.o[4] = r_rcp<R>(ma + ia * rcn1 * rcn1 + mb);
        c.o[5] = r_clamp(-R(dist - pp[4]) * p[7], -p[8], p[8]);
    } else if (rowlane == 5 || rowlane == 8) {                // cpRotaryLimitJointPreStep
        const int jj = j0 + rowlane, a = T_I(joint_a, jj), b = T_I(joint_b, jj);
        const R *p = &T_R(joint_p, jj * JOINT_PARAMS);
        const P *pp = &T_P(p_joint, jj * 7);
        const P dist = E_P(ang, b) - E_P(ang, a);
        P pdist = P(0);
        if (dist > pp[5]) pdist = pp[5] - dist; else if (dist < pp[4]) pdist = pp[4] - dist;
        c.o[0] = r_clamp(-R(pdist) * p[7], -p[8], p[8]);
    } else if (rowlane == 6 || rowlane == 9) {                // cpSimpleMotor: the rate Robot.update set (ph_control)
        c.o[0] = E_R(jrate, j0 + rowlane);
    }
}

// ---------------------------------------------------------------- phase: arbiters (cpArbiterUpdate + cpArbiterPreStep)
// and joint preStep for every joint kind that does not touch velocities.
template <typename R, typename P> MGX_HD void ph_arbiters_joints(Env<R, P> &e, SolveCtx<R> &c, int lane, int nl) {
    int nov = E_I(misc, M_NOV), ncache = E_I(misc, M_NCACHE);
    int kcap = e.h->max_contacts, ccap = e.h->cache_slots;
    R dt = e.cst(C_DT), slop = e.cst(C_SLOP), brate = e.cst(C_CONTACT_BIAS_RATE);
    int koff = 0, rank = 0, scanned = 0;
#ifndef MGX_ARB_BALLOT
#define MGX_ARB_BALLOT 1      // 0: round 5's per-lane rescans of the manifold words (A/B builds)
#endif
#if defined(__HIP_DEVICE_COMPILE__) && MGX_ARB_BALLOT
    // A touching pair's first contact slot and arbiter rank are prefix sums over the pairs before it.  Round 5: every lane rescanned the
    // manifold words of all earlier pairs (and lane 0 all of them once more for the totals) -- up to nov DEPENDENT-latency LDS reads on a
    // wavefront with nothing else to run.  Now: per round of nl pairs two ballots of the point count's bits give every lane the sums of the
    // lanes before it, the running totals stay in (group-uniform) registers.  Exact whenever nothing is dropped -- all contacts fit
    // max_contacts, all arbiters cache_slots, which `fits` decides from the same ballots before anything is written; a substep that does
    // overflow (counted, warned about by the host) takes round 5's sequential rule unchanged.
    const int wave_lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int shift = wave_lane - lane;
    const unsigned long long group_mask = nl >= 64 ? ~0ull : ((1ull << nl) - 1ull), below = (1ull << lane) - 1ull;
    int tot_k = 0, tot_r = 0;
    for (int base = 0; base < nov; base += nl) {
        const int q = base + lane, cn = q < nov ? (E_I(mcnt, q) & 3) : 0;
        const unsigned long long b0 = (__builtin_amdgcn_ballot_w64((cn & 1) != 0) >> shift) & group_mask;
        const unsigned long long b1 = (__builtin_amdgcn_ballot_w64((cn & 2) != 0) >> shift) & group_mask;
        tot_k += __builtin_popcountll(b0) + 2 * __builtin_popcountll(b1);
        tot_r += __builtin_popcountll(b0 | b1);
    }
    const bool fits = tot_k <= kcap && tot_r <= ccap;          // (the same in every lane of the group)
    int run_k = 0, run_r = 0;
    for (int q = lane; q < nov; q += nl) {
        const int mc = E_I(mcnt, q), cnt = mc & 3;
        if (fits) {
            const unsigned long long b0 = (__builtin_amdgcn_ballot_w64((cnt & 1) != 0) >> shift) & group_mask;
            const unsigned long long b1 = (__builtin_amdgcn_ballot_w64((cnt & 2) != 0) >> shift) & group_mask;
            koff = run_k + __builtin_popcountll(b0 & below) + 2 * __builtin_popcountll(b1 & below);
            rank = run_r + __builtin_popcountll((b0 | b1) & below);
            run_k += __builtin_popcountll(b0) + 2 * __builtin_popcountll(b1);
            run_r += __builtin_popcountll(b0 | b1);
        } else {
            for (; scanned < q; scanned++) { int c = E_I(mcnt, scanned) & 3; if (c > 0 && koff + c <= kcap && rank < ccap) { koff += c; rank++; } }
            scanned = q + 1;
        }
        if (cnt == 0) continue;
        if (koff + cnt > kcap || rank >= ccap) continue;     // dropped: counted by lane 0 below
#else
    const bool fits = false; const int tot_k = 0, tot_r = 0;
    for (int q = lane; q < nov; q += nl) {
        for (; scanned < q; scanned++) { int c = E_I(mcnt, scanned) & 3; if (c > 0 && koff + c <= kcap && rank < ccap) { koff += c; rank++; } }
        const int mc = E_I(mcnt, q), cnt = mc & 3;
        scanned = q + 1;                                      // this entry is accounted for right below
        if (cnt == 0) continue;
        if (koff + cnt > kcap || rank >= ccap) continue;     // dropped: counted by lane 0 below
#endif
        int p = (int)E_OV(q), pr = T_I(pair, p), sa = pr & 0xFF, sb = pr >> 8;
        int A = T_I(shape_body, sa), B = T_I(shape_body, sb);
        // cached arbiter for this shape pair?
        int ci = -1; uint32_t old = 0;
        // (four headers per turn: the reads go out together -- one LDS round trip per four entries, not per entry)
        for (int c = 0; c < ncache; c += 4) {
            uint32_t hd[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int u = 0; u < 4; u++) hd[u] = c + u < ncache ? (uint32_t)E_I(chead, c + u) : 0xFFFFFFFFu;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int u = 0; u < 4; u++) if (c + u < ncache && (int)(hd[u] & 0xFFFu) == p) { ci = c + u; old = hd[u]; }
        }
        bool first = true;
        if (ci >= 0) { first = ((old >> 12) & 3u) != 0u; E_MATCHED(ci) = 1; }
        int ocnt = ci >= 0 ? (int)((old >> 14) & 3u) : 0;
        int oh[2] = {(int)((old >> 16) & 0xFFu), (int)((old >> 24) & 0xFFu)};
        const int mh = (mc >> 8) & 0xFFFF, ms = (mc >> 24) & 0xFF;      // point hashes, manifold slot
        int nh[2] = {mh & 0xFF, (mh >> 8) & 0xFF};
        R nx = E_R(mn, 2 * ms), ny = E_R(mn, 2 * ms + 1);
        R mu = T_R(shape_u, sa) * T_R(shape_u, sb);
        R ma = T_R(body_minv, A), ia = T_R(body_iinv, A), mb = T_R(body_minv, B), ib = T_R(body_iinv, B);
        R pax = R(E_P(px, A)), pay = R(E_P(py, A)), pbx = R(E_P(px, B)), pby = R(E_P(py, B));
        for (int i = 0; i < cnt; i++) {
            int k = koff + i;
            R r1x = E_R(mp, 8 * ms + 4 * i) - pax, r1y = E_R(mp, 8 * ms + 4 * i + 1) - pay;
            R r2x = E_R(mp, 8 * ms + 4 * i + 2) - pbx, r2y = E_R(mp, 8 * ms + 4 * i + 3) - pby;
            R jn = 0, jt = 0;
            for (int o = 0; o < ocnt; o++) if (oh[o] == nh[i]) { jn = E_R(cj, 4 * ci + 2 * o); jt = E_R(cj, 4 * ci + 2 * o + 1); }
            R rcn1 = r1x * ny - r1y * nx, rcn2 = r2x * ny - r2y * nx;          // cross(r, n)
            R rct1 = r1x * nx + r1y * ny, rct2 = r2x * nx + r2y * ny;          // cross(r, perp(n))
            R kn = ma + ia * rcn1 * rcn1 + mb + ib * rcn2 * rcn2;
            R kt = ma + ia * rct1 * rct1 + mb + ib * rct2 * rct2;
            R dist = ((r2x - r1x) + (pbx - pax)) * nx + ((r2y - r1y) + (pby - pay)) * ny;
            E_I(kab, k) = A | (B << 8) | (first ? 1 << 16 : 0);
            E_R(knx, k) = nx; E_R(kny, k) = ny;
            E_R(kr1x, k) = r1x; E_R(kr1y, k) = r1y; E_R(kr2x, k) = r2x; E_R(kr2y, k) = r2y;
            E_R(knm, k) = r_rcp<R>(kn); E_R(ktm, k) = r_rcp<R>(kt);
            E_R(kbias, k) = -brate * r_min(R(0), dist + slop);
            E_R(kjb, k) = R(0); E_R(kjn, k) = jn; E_R(kjt, k) = jt; E_R(kmu, k) = mu;
        }
        E_I(nchead, rank) = (int32_t)cache_pack((uint32_t)p, 0u, (uint32_t)cnt, (uint32_t)nh[0], (uint32_t)nh[1]);
        E_I(koff, rank) = koff;
        koff += cnt; rank++;
    }
    if (lane == 0) {
        if (fits) {      // totals straight from the ballots
            E_I(misc, M_NK) = tot_k; E_I(misc, M_NARB) = tot_r;
        } else {         // lane 0 rescans everything once; also counts drops
            int k = 0, r = 0, dropped = 0;
            for (int q = 0; q < nov; q++) { int c = E_I(mcnt, q) & 3; if (c > 0) { if (k + c <= kcap && r < ccap) { k += c; r++; } else dropped++; } }
            E_I(misc, M_NK) = k; E_I(misc, M_NARB) = r;
            if (dropped) E_I(misc, M_OVERFLOW) += dropped;
        }
    }
    // joint preSteps: straight into the registers of the lanes that own the joints (SolveCtx, below).
    (void)dt;
    joints_prestep(e, c, lane, nl);
}

// ---------------------------------------------------------------- velocity helpers for the solver
template <typename R> struct Vel { R vx, vy, w; };
#define LOADV(b) Vel<R>{E_R(vx, b), E_R(vy, b), E_R(w, b)}
#define STOREV(b, v) do { E_R(vx, b) = (v).vx; E_R(vy, b) = (v).vy; E_R(w, b) = (v).w; } while (0)

// cpArbiterApplyImpulse for one contact point, split so that the constants of contact k + 1 can be fetched from LDS
// while contact k is being computed (they never alias the velocities contact k writes): ContactK = everything that does
// not depend on the bodies' current velocities.
template <typename R> struct ContactK {
    int a, b;
    R nx, ny, r1x, r1y, r2x, r2y, ma, ia, mb, ib, n_mass, t_mass, bias, mu, jb, jn, jt;
};
template <typename R, typename P> MGX_HD ContactK<R> contact_load(const Env<R, P> &e, int k) {
    ContactK<R> c;
    int ab = E_I(kab, k);
    c.a = ab & 0xFF; c.b = (ab >> 8) & 0xFF;
    c.nx = E_R(knx, k); c.ny = E_R(kny, k);
    c.r1x = E_R(kr1x, k); c.r1y = E_R(kr1y, k); c.r2x = E_R(kr2x, k); c.r2y = E_R(kr2y, k);
    c.ma = T_R(body_minv, c.a); c.ia = T_R(body_iinv, c.a); c.mb = T_R(body_minv, c.b); c.ib = T_R(body_iinv, c.b);
    c.n_mass = E_R(knm, k); c.t_mass = E_R(ktm, k); c.bias = E_R(kbias, k); c.mu = E_R(kmu, k);
    c.jb = E_R(kjb, k); c.jn = E_R(kjn, k); c.jt = E_R(kjt, k);
    return c;
}
template <typename R, typename P> MGX_HD void contact_apply_loaded(Env<R, P> &e, int k, const ContactK<R> &c) {
    const int a = c.a, b = c.b;
    const R nx = c.nx, ny = c.ny, r1x = c.r1x, r1y = c.r1y, r2x = c.r2x, r2y = c.r2y, ma = c.ma, ia = c.ia, mb = c.mb, ib = c.ib;
    Vel<R> va = LOADV(a), vb = LOADV(b);
    R vbax = E_R(vbx, a), vbay = E_R(vby, a), wba = E_R(wb, a);
    R vbbx = E_R(vbx, b), vbby = E_R(vby, b), wbb = E_R(wb, b);
    // bias (pseudo-velocity) part
    R vb1x = vbax - r1y * wba, vb1y = vbay + r1x * wba;
    R vb2x = vbbx - r2y * wbb, vb2y = vbby + r2x * wbb;
    R vbn = (vb2x - vb1x) * nx + (vb2y - vb1y) * ny;
    R vrx = (vb.vx - r2y * vb.w) - (va.vx - r1y * va.w), vry = (vb.vy + r2x * vb.w) - (va.vy + r1x * va.w);
    R vrn = vrx * nx + vry * ny;
    R vrt = -vrx * ny + vry * nx;                           // dot(vr, perp(n))
    R n_mass = c.n_mass;
    R jbn = (c.bias - vbn) * n_mass;
    R jbn_old = c.jb;
    R jb_new = r_max(jbn_old + jbn, R(0));
    E_R(kjb, k) = jb_new;
    R jn = -vrn * n_mass;                                   // bounce = 0 (elasticity 0 everywhere)
    R jn_old = c.jn;
    R jn_new = r_max(jn_old + jn, R(0));
    E_R(kjn, k) = jn_new;
    R jt_max = c.mu * jn_new;
    R jt = -vrt * c.t_mass;
    R jt_old = c.jt;
    R jt_new = r_clamp(jt_old + jt, -jt_max, jt_max);
    E_R(kjt, k) = jt_new;
    R jbx = nx * (jb_new - jbn_old), jby = ny * (jb_new - jbn_old);
    E_R(vbx, a) = vbax - jbx * ma; E_R(vby, a) = vbay - jby * ma; E_R(wb, a) = wba - ia * (r1x * jby - r1y * jbx);
    E_R(vbx, b) = vbbx + jbx * mb; E_R(vby, b) = vbby + jby * mb; E_R(wb, b) = wbb + ib * (r2x * jby - r2y * jbx);
    R dn = jn_new - jn_old, dtg = jt_new - jt_old;
    R jx = nx * dn - ny * dtg, jy = nx * dtg + ny * dn;     // cpvrotate(n, (dn, dt))
    va.vx -= jx * ma; va.vy -= jy * ma; va.w -= ia * (r1x * jy - r1y * jx);
    vb.vx += jx * mb; vb.vy += jy * mb; vb.w += ib * (r2x * jy - r2y * jx);
    STOREV(a, va); STOREV(b, vb);
}
template <typename R, typename P> MGX_HD void contact_apply_impulse(Env<R, P> &e, int k) {
    contact_apply_loaded(e, k, contact_load(e, k));
}

// ---------------------------------------------------------------- solve
// Chipmunk's order per substep: (arbiter preStep, joint preStep incl. the spring torque) -> cached arbiter impulses ->
// cached joint impulses -> 10 x { all arbiters in pair order ; all joints in insertion order }.  Contacts (which couple
// islands) stay sequential on lane 0 against the LDS copy of the velocities, with a store / load hand-over around them only
// when the env has contacts at all.

template <typename R, typename P> MGX_HD int ri_body(const Env<R, P> &e, int slot) {   // 1 robot, 2 / 3 eyes, 4 / 5 fingers
    const TmplHeader &h = *e.h;
    return slot == 0 ? h.control_body : slot == 1 ? h.robot_body : slot == 2 ? h.eye_body[0] : slot == 3 ? h.eye_body[1]
         : slot == 4 ? h.finger_body[0] : h.finger_body[1];
}
// the robot island's velocities, LDS <-> the (uniform) registers of every lane
template <typename R, typename P> MGX_HD void ri_load_vel(const Env<R, P> &e, SolveCtx<R> &c) {
    c.rvx = E_R(vx, c.b_robot); c.rvy = E_R(vy, c.b_robot); c.rw = E_R(w, c.b_robot);
    c.ew[0] = E_R(w, c.b_eye[0]); c.ew[1] = E_R(w, c.b_eye[1]);
    MGX_UNROLL for (int f = 0; f < 2; f++) { const int b = c.b_finger[f]; c.fvx[f] = E_R(vx, b); c.fvy[f] = E_R(vy, b); c.fw[f] = E_R(w, b); }
}
template <typename R, typename P> MGX_HD void ri_store_vel(Env<R, P> &e, const SolveCtx<R> &c) {
    E_R(vx, c.b_robot) = c.rvx; E_R(vy, c.b_robot) = c.rvy; E_R(w, c.b_robot) = c.rw;
    E_R(w, c.b_eye[0]) = c.ew[0]; E_R(w, c.b_eye[1]) = c.ew[1];
    MGX_UNROLL for (int f = 0; f < 2; f++) { const int b = c.b_finger[f]; E_R(vx, b) = c.fvx[f]; E_R(vy, b) = c.fvy[f]; E_R(w, b) = c.fw[f]; }
}
// does this lane keep a block's pair?  (lane 1 + k <-> island k; rowlane 0 keeps the robot's)
template <typename R> MGX_HD bool is_block_lane(const SolveCtx<R> &c, int lane) { return c.mj >= 0 && (lane & (ROW - 1)) != 0; }

// Once per env-step, after ph_load_state: who owns what, the constants that do not change during the env-step, and the
// accumulated impulses (they stay in registers until solve_ctx_flush).
template <typename R, typename P> MGX_HD void solve_ctx_init(Env<R, P> &e, SolveCtx<R> &c, int lane, int nl) {
    (void)nl;
    const TmplHeader &h = *e.h;
    const int rowlane = lane & (ROW - 1), j0 = h.robot_j0;
    c.mj = -1; c.mbody = -1;
    c.b_robot = h.robot_body; c.b_eye[0] = h.eye_body[0]; c.b_eye[1] = h.eye_body[1]; c.b_finger[0] = h.finger_body[0]; c.b_finger[1] = h.finger_body[1];
    if (rowlane == 0) { c.mj = j0; c.mbody = h.robot_body; }
    else if (lane < ROW && lane - 1 < h.n_islands) { c.mj = T_I(island_j, lane - 1); c.mbody = T_I(joint_b, c.mj); }
    c.pk0 = c.pk1 = c.pk2 = c.pk3 = c.pb0 = c.pb1 = c.pa0 = c.pa1 = c.plim = R(0);
    c.gim = c.gb = c.ga = c.gratio = c.glim = R(0);
    c.avx = c.avy = c.aw = c.mvx = c.mvy = c.mw = c.mminv = c.miinv = R(0);
    if (c.mj >= 0) {
        const int j = c.mj;
        c.pa0 = E_R(ja0, j); c.pa1 = E_R(ja1, j); c.ga = E_R(ja0, j + 1);
        c.plim = E_R(jlim, j); c.glim = E_R(jlim, j + 1);
        c.gim = T_R(joint_p, (j + 1) * JOINT_PARAMS); c.gratio = T_R(joint_p, (j + 1) * JOINT_PARAMS + 5);
        c.mminv = T_R(body_minv, c.mbody); c.miinv = T_R(body_iinv, c.mbody);
        // cpPivotJointPreStep's K^-1 with r1 = r2 = 0 (joints_prestep): k11 = k22 = m_a^-1 + m_b^-1, k12 = 0
        const R msum = T_R(body_minv, T_I(joint_a, j)) + c.mminv;
        const R det_inv = R(1) / (msum * msum);
        c.pk0 = msum * det_inv; c.pk3 = msum * det_inv;
    }
    c.r_minv = T_R(body_minv, h.robot_body); c.r_iinv = T_R(body_iinv, h.robot_body);
    MGX_UNROLL for (int k = 0; k < 2; k++) {
        c.e_iinv[k] = T_R(body_iinv, h.eye_body[k]);
        c.f_minv[k] = T_R(body_minv, h.finger_body[k]); c.f_iinv[k] = T_R(body_iinv, h.finger_body[k]);
        c.s_im[k] = T_R(joint_p, (j0 + 2 + k) * JOINT_PARAMS); c.s_coef[k] = T_R(joint_p, (j0 + 2 + k) * JOINT_PARAMS + 6);
        const int jp = j0 + 4 + 3 * k;            // this finger's pin, limit, motor
        c.pin_lim[k] = E_R(jlim, jp); c.lim_max[k] = E_R(jlim, jp + 1); c.mot_lim[k] = E_R(jlim, jp + 2);
        c.lim_im[k] = T_R(joint_p, (jp + 1) * JOINT_PARAMS); c.mot_im[k] = T_R(joint_p, (jp + 2) * JOINT_PARAMS);
        c.acc[k] = R(0);
        c.acc[2 + 3 * k] = E_R(ja0, jp); c.acc[3 + 3 * k] = E_R(ja0, jp + 1); c.acc[4 + 3 * k] = E_R(ja0, jp + 2);
        c.lim_lo[k] = c.lim_hi[k] = c.mot_rate[k] = R(0);
        MGX_UNROLL for (int i = 0; i < 6; i++) c.pin[k][i] = R(0);
    }
    MGX_UNROLL for (int i = 0; i < 6; i++) c.o[i] = R(0);
    c.rvx = c.rvy = c.rw = R(0);
    MGX_UNROLL for (int k = 0; k < 2; k++) c.ew[k] = c.fvx[k] = c.fvy[k] = c.fw[k] = R(0);
    c.has_contacts = 0;
}
// ... and back into the working set, before ph_store_state
template <typename R, typename P> MGX_HD void solve_ctx_flush(Env<R, P> &e, const SolveCtx<R> &c, int lane, int nl) {
    (void)nl;
    if (c.mj >= 0 && lane < ROW) { E_R(ja0, c.mj) = c.pa0; E_R(ja1, c.mj) = c.pa1; E_R(ja0, c.mj + 1) = c.ga; }
    if (lane == 0) {
        const int j0 = e.h->robot_j0;
        MGX_UNROLL for (int k = 0; k < 2; k++) {
            const int jp = j0 + 4 + 3 * k;
            E_R(ja0, jp) = c.acc[2 + 3 * k]; E_R(ja0, jp + 1) = c.acc[3 + 3 * k]; E_R(ja0, jp + 2) = c.acc[4 + 3 * k];
        }
    }
}

// cpArbiterApplyCachedImpulse for every warm contact (lane 0, LDS velocities)
template <typename R, typename P> MGX_HD void contacts_warm_start(Env<R, P> &e) {
    int nk = E_I(misc, M_NK);
    for (int k = 0; k < nk; k++) {
        int ab = E_I(kab, k), a = ab & 0xFF, b = (ab >> 8) & 0xFF;
        if (ab >> 16) continue;                   // (the arbiter's first substep: no cached impulse)
        R nx = E_R(knx, k), ny = E_R(kny, k), jn = E_R(kjn, k), jt = E_R(kjt, k);
        R jx = nx * jn - ny * jt, jy = nx * jt + ny * jn;
        R r1x = E_R(kr1x, k), r1y = E_R(kr1y, k), r2x = E_R(kr2x, k), r2y = E_R(kr2y, k);
        E_R(vx, a) -= jx * T_R(body_minv, a); E_R(vy, a) -= jy * T_R(body_minv, a); E_R(w, a) -= T_R(body_iinv, a) * (r1x * jy - r1y * jx);
        E_R(vx, b) += jx * T_R(body_minv, b); E_R(vy, b) += jy * T_R(body_minv, b); E_R(w, b) += T_R(body_iinv, b) * (r2x * jy - r2y * jx);
    }
}

// solve step A (after the preStep phase): lane 0 ages the contact cache; every lane picks up the owners' preStep results
// and the island's velocities, and the springs apply their torque (cpDampedRotarySpring preStep, in joint order)
template <typename R, typename P> MGX_HD void solve_begin(Env<R, P> &e, SolveCtx<R> &c, int lane, int nl) {
    (void)nl;
    const TmplHeader &h = *e.h;
    c.has_contacts = E_I(misc, M_NK) > 0;
    if (lane == 0) {
        int narb = E_I(misc, M_NARB), ncache = E_I(misc, M_NCACHE);
        // untouched cached arbiters age; they survive collision_persistence = 3 steps (cpSpaceArbiterSetFilter)
        int n = narb;
        for (int q = 0; q < ncache; q++) {
            if (E_MATCHED(q)) { E_MATCHED(q) = 0; continue; }
            uint32_t hd = (uint32_t)E_I(chead, q);
            uint32_t age = ((hd >> 12) & 3u) + 1u;
            if (age <= 2u && n < h.cache_slots) {
                E_I(nchead, n) = (int32_t)((hd & ~(3u << 12)) | (age << 12));
                for (int i = 0; i < 4; i++) E_R(ncj, 4 * n + i) = E_R(cj, 4 * q + i);
                E_I(koff, n) = -1;
                n++;
            }
        }
        E_I(misc, M_NNCACHE) = n;
    }
    const R js0 = RB(2, o[0]), js1 = RB(3, o[0]);
#define MGX_PIN_BCAST(f, J) c.pin[f][0] = RB(J, o[0]); c.pin[f][1] = RB(J, o[1]); c.pin[f][2] = RB(J, o[2]); \
                            c.pin[f][3] = RB(J, o[3]); c.pin[f][4] = RB(J, o[4]); c.pin[f][5] = RB(J, o[5]);
    MGX_PIN_BCAST(0, 4) MGX_PIN_BCAST(1, 7)
#undef MGX_PIN_BCAST
    const R lb[2] = {RB(5, o[0]), RB(8, o[0])};
    c.mot_rate[0] = RB(6, o[0]); c.mot_rate[1] = RB(9, o[0]);
    MGX_UNROLL for (int f = 0; f < 2; f++) {
        // cpRotaryLimitJoint: inside its range (bias 0) the joint does nothing and forgets its impulse; else the
        // accumulated impulse is clamped to the side that pushes back
        c.lim_lo[f] = lb[f] < R(0) ? R(0) : (lb[f] > R(0) ? -c.lim_max[f] : R(0));
        c.lim_hi[f] = lb[f] < R(0) ? c.lim_max[f] : R(0);
        if (lb[f] == R(0)) c.acc[3 + 3 * f] = R(0);
        c.lim_b[f] = lb[f];
    }
    c.acc[0] = R(0); c.acc[1] = R(0);
    ri_load_vel(e, c);
    c.rw -= js0 * c.r_iinv; c.ew[0] += js0 * c.e_iinv[0];
    c.rw -= js1 * c.r_iinv; c.ew[1] += js1 * c.e_iinv[1];
    if (is_block_lane(c, lane)) { c.mvx = E_R(vx, c.mbody); c.mvy = E_R(vy, c.mbody); c.mw = E_R(w, c.mbody); }
    // the contact warm start (next step) must see the spring impulses
    if (c.has_contacts && lane == 0) { E_R(w, c.b_robot) = c.rw; E_R(w, c.b_eye[0]) = c.ew[0]; E_R(w, c.b_eye[1]) = c.ew[1]; }
}
// solve step B: cached arbiter impulses (lane 0, LDS)
template <typename R, typename P> MGX_HD void solve_warm_contacts(Env<R, P> &e, const SolveCtx<R> &c, int lane) {
    if (c.has_contacts && lane == 0) contacts_warm_start(e);
}
// after contacts ran on the LDS velocities: everybody reloads
template <typename R, typename P> MGX_HD void solve_reload(const Env<R, P> &e, SolveCtx<R> &c, int lane) {
    ri_load_vel(e, c);
    if (is_block_lane(c, lane)) { c.mvx = E_R(vx, c.mbody); c.mvy = E_R(vy, c.mbody); c.mw = E_R(w, c.mbody); }
}
// the lanes that keep the robot's pair take the robot's velocity from the island's registers
template <typename R> MGX_HD void pg_take_robot(SolveCtx<R> &c, int lane) {
    if ((lane & (ROW - 1)) == 0) { c.mvx = c.rvx; c.mvy = c.rvy; c.mw = c.rw; }
}
// solve step C: cached joint impulses (cpConstraint applyCachedImpulse, joint order): the pairs ...
template <typename R, typename P> MGX_HD void solve_warm_pg(Env<R, P> &e, SolveCtx<R> &c, int lane) {
    if (c.has_contacts) solve_reload(e, c, lane);
    pg_take_robot(c, lane);
    c.mvx += c.pa0 * c.mminv; c.mvy += c.pa1 * c.mminv;      // (both anchors are body origins: no torque)
    c.mw += c.ga * c.miinv;
}
// ... then the robot's joints 2..9 (springs have no cached impulse)
template <typename R> MGX_HD void solve_warm_chain(SolveCtx<R> &c) {
    c.rvx = RB(0, mvx); c.rvy = RB(0, mvy); c.rw = RB(0, mw);
    MGX_UNROLL for (int f = 0; f < 2; f++) {
        const R *pn = c.pin[f];
        const R jn = c.acc[2 + 3 * f];
        const R jx = pn[2] * jn, jy = pn[3] * jn;
        c.rvx -= jx * c.r_minv; c.rvy -= jy * c.r_minv; c.rw -= c.r_iinv * (pn[0] * jy - pn[1] * jx);
        c.fvx[f] += jx * c.f_minv[f]; c.fvy[f] += jy * c.f_minv[f];
        const R jl = c.acc[3 + 3 * f];
        c.rw -= jl * c.r_iinv; c.fw[f] += jl * c.f_iinv[f];
        const R jm = c.acc[4 + 3 * f];
        c.rw -= jm * c.r_iinv; c.fw[f] += jm * c.f_iinv[f];
    }
}
// solve step D: one Gauss-Seidel iteration = [publish island velocities] [contacts] [pairs] [robot joints 2..9]
template <typename R, typename P> MGX_HD void solve_iter_publish(Env<R, P> &e, const SolveCtx<R> &c, int lane) {
    if (!c.has_contacts) return;
    if (lane == 0) ri_store_vel(e, c);
    if (is_block_lane(c, lane)) { E_R(vx, c.mbody) = c.mvx; E_R(vy, c.mbody) = c.mvy; E_R(w, c.mbody) = c.mw; }
}
// cpArbiterApplyImpulse's two halves on two lanes.  A contact point carries two impulse chains that never meet inside the
// iterations: the velocity chain (jnAcc / jtAcc against v, w) and the bias chain (jBias against v_bias, w_bias) -- same
// constants, same expression trees (relative velocity at the point, normal impulse, clamp at 0, apply to both bodies), disjoint
// rows of the working set.  Lane 0 of the env runs the first and lane 1 the second in ONE instruction stream: the operands that
// differ (which velocity rows, jnAcc or jBias, the bias term, the tangent part) are per-lane selects, every arithmetic expression is
// the one contact_apply_loaded evaluates for that chain, in its order.  The contact chain -- what the slowest workgroups of a
// launch spend their time in -- then costs the velocity half alone.
#ifndef MGX_CONTACT_SPLIT
#define MGX_CONTACT_SPLIT 1
#endif
template <typename R, typename P> MGX_HD void contact_apply_split(Env<R, P> &e, int k, const ContactK<R> &c, bool bl) {
    const int a = c.a, b = c.b;
    const R nx = c.nx, ny = c.ny, r1x = c.r1x, r1y = c.r1y, r2x = c.r2x, r2y = c.r2y, ma = c.ma, ia = c.ia, mb = c.mb, ib = c.ib;
    const int ox = bl ? e.wo.vbx : e.wo.vx, oy = bl ? e.wo.vby : e.wo.vy, ow = bl ? e.wo.wb : e.wo.w;
    R *wr = e.wr;
    R vax = wr[ox + a * WorkOff::S_vx], vay = wr[oy + a * WorkOff::S_vy], wa = wr[ow + a * WorkOff::S_w];
    R vbx = wr[ox + b * WorkOff::S_vx], vby = wr[oy + b * WorkOff::S_vy], wb = wr[ow + b * WorkOff::S_w];
    R rx = (vbx - r2y * wb) - (vax - r1y * wa), ry = (vby + r2x * wb) - (vay + r1x * wa);
    R vn = rx * nx + ry * ny;
    R vrt = -rx * ny + ry * nx;                             // dot(vr, perp(n)): the velocity lane's
    R acc_old = bl ? c.jb : c.jn;
    R jd = bl ? (c.bias - vn) * c.n_mass : -vn * c.n_mass;  // bounce = 0 (elasticity 0 everywhere)
    R acc_new = r_max(acc_old + jd, R(0));
    wr[(bl ? e.wo.kjb : e.wo.kjn) + k * WorkOff::S_kjn] = acc_new;
    static_assert(WorkOff::S_kjb == WorkOff::S_kjn, "contact record strides");
    R jt_max = c.mu * acc_new;
    R jt = -vrt * c.t_mass;
    R jt_old = c.jt;
    R jt_new = r_clamp(jt_old + jt, -jt_max, jt_max);
    if (!bl) E_R(kjt, k) = jt_new;
    R dn = acc_new - acc_old, dtg = jt_new - jt_old;
    R jx = bl ? nx * dn : nx * dn - ny * dtg, jy = bl ? ny * dn : nx * dtg + ny * dn;     // cpvrotate(n, (dn, dt)) / n * dn
    vax -= jx * ma; vay -= jy * ma; wa -= ia * (r1x * jy - r1y * jx);
    vbx += jx * mb; vby += jy * mb; wb += ib * (r2x * jy - r2y * jx);
    wr[ox + a * WorkOff::S_vx] = vax; wr[oy + a * WorkOff::S_vy] = vay; wr[ow + a * WorkOff::S_w] = wa;
    wr[ox + b * WorkOff::S_vx] = vbx; wr[oy + b * WorkOff::S_vy] = vby; wr[ow + b * WorkOff::S_w] = wb;
}
template <typename R, typename P> MGX_HD void solve_iter_contacts(Env<R, P> &e, const SolveCtx<R> &c, int lane) {
    if (!c.has_contacts || lane > (MGX_CONTACT_SPLIT ? 1 : 0)) return;
    int nk = E_I(misc, M_NK);
    if (nk == 0) return;
    ContactK<R> cur = contact_load(e, 0);
    for (int k = 0; k < nk; k++) {
        // the next contact's constants are in flight while this one's dependent chain runs
        const ContactK<R> nxt = contact_load(e, k + 1 < nk ? k + 1 : k);
        if (MGX_CONTACT_SPLIT) contact_apply_split(e, k, cur, lane == 1); else contact_apply_loaded(e, k, cur);
        cur = nxt;
    }
}
// cpPivotJoint + cpGearJoint applyImpulse of the lane's pair: first body immovable (kinematic / static), both anchors at the
// body origins (Robot.setup's pair to the control body, entities.py:255-263; a block's pair, :703-711)
template <typename R, typename P> MGX_HD void solve_iter_pg(Env<R, P> &e, SolveCtx<R> &c, int lane) {
    if (c.has_contacts) solve_reload(e, c, lane);
    pg_take_robot(c, lane);
    {
        const R vrx = c.mvx - c.avx, vry = c.mvy - c.avy;
        const R dx = c.pb0 - vrx, dy = c.pb1 - vry;
        R jx = dx * c.pk0 + dy * c.pk1, jy = dx * c.pk2 + dy * c.pk3;
        const R ox = c.pa0, oy = c.pa1;
        R nxv = ox + jx, nyv = oy + jy;
        clamp_len(nxv, nyv, c.plim);
        c.pa0 = nxv; c.pa1 = nyv;
        jx = nxv - ox; jy = nyv - oy;
        c.mvx += jx * c.mminv; c.mvy += jy * c.mminv;
    }
    {
        const R wr = c.mw * c.gratio - c.aw;
        R jj = (c.gb - wr) * c.gim;
        const R jold = c.ga;
        const R jn = r_clamp(jold + jj, -c.glim, c.glim);
        c.ga = jn; jj = jn - jold;
        c.mw = c.mw + jj * c.miinv;
    }
}
// robot joints 2..9 in Robot.setup order: two eye springs, then per finger {pin (anchored at the finger's origin), limit,
// motor}; cpDampedRotarySpring / cpPinJoint / cpRotaryLimitJoint / cpSimpleMotor applyImpulse
template <typename R> MGX_HD void solve_iter_chain(SolveCtx<R> &c) {
    c.rvx = RB(0, mvx); c.rvy = RB(0, mvy); c.rw = RB(0, mw);
    MGX_UNROLL for (int k = 0; k < 2; k++) {
        const R wrn = c.rw - c.ew[k];
        const R w_damp = (c.acc[k] - wrn) * c.s_coef[k];
        c.acc[k] = wrn + w_damp;                              // target_wrn
        const R j_damp = w_damp * c.s_im[k];
        c.rw = c.rw + j_damp * c.r_iinv; c.ew[k] = c.ew[k] - j_damp * c.e_iinv[k];
    }
    MGX_UNROLL for (int f = 0; f < 2; f++) {
        const R ma = c.r_minv, ia = c.r_iinv, mb = c.f_minv[f], ib = c.f_iinv[f];
        {
            const R *pn = c.pin[f];                           // r1x r1y nx ny n_mass bias
            const R vrx = c.fvx[f] - (c.rvx - pn[1] * c.rw), vry = c.fvy[f] - (c.rvy + pn[0] * c.rw);
            const R vrn = vrx * pn[2] + vry * pn[3];
            R jn = (pn[5] - vrn) * pn[4];
            const R jold = c.acc[2 + 3 * f];
            const R jnew = r_clamp(jold + jn, -c.pin_lim[f], c.pin_lim[f]);
            c.acc[2 + 3 * f] = jnew; jn = jnew - jold;
            const R jx = pn[2] * jn, jy = pn[3] * jn;
            c.rvx -= jx * ma; c.rvy -= jy * ma; c.rw -= ia * (pn[0] * jy - pn[1] * jx);
            c.fvx[f] += jx * mb; c.fvy[f] += jy * mb;
        }
        {
            const R wr = c.fw[f] - c.rw;
            R jj = -(c.lim_b[f] + wr) * c.lim_im[f];
            const R jold = c.acc[3 + 3 * f];
            const R jn = r_clamp(jold + jj, c.lim_lo[f], c.lim_hi[f]);
            c.acc[3 + 3 * f] = jn; jj = jn - jold;
            c.rw = c.rw - jj * ia; c.fw[f] = c.fw[f] + jj * ib;
        }
        {
            const R wr = c.fw[f] - c.rw + c.mot_rate[f];
            R jj = -wr * c.mot_im[f];
            const R jold = c.acc[4 + 3 * f];
            const R jn = r_clamp(jold + jj, -c.mot_lim[f], c.mot_lim[f]);
            c.acc[4 + 3 * f] = jn; jj = jn - jold;
            c.rw = c.rw - jj * ia; c.fw[f] = c.fw[f] + jj * ib;
        }
    }
}
// solve step E: velocities back into the working set, then next substep's Robot.update
template <typename R, typename P> MGX_HD void solve_end(Env<R, P> &e, SolveCtx<R> &c, int lane) {
    if (lane == 0) { ri_store_vel(e, c); ph_control(e); }
    if (is_block_lane(c, lane) && lane < ROW) { E_R(vx, c.mbody) = c.mvx; E_R(vy, c.mbody) = c.mvy; E_R(w, c.mbody) = c.mw; }
}

// ---------------------------------------------------------------- phase: publish the contact cache for the next substep
template <typename R, typename P> MGX_HD void ph_cache_commit(Env<R, P> &e, int lane, int nl) {
    int n = E_I(misc, M_NNCACHE);
    for (int c = lane; c < n; c += nl) {
        uint32_t hd = (uint32_t)E_I(nchead, c);
        int k0 = E_I(koff, c);
        E_I(chead, c) = (int32_t)hd;
        if (k0 >= 0) {
            int cnt = (int)((hd >> 14) & 3u);
            E_R(cj, 4 * c + 0) = E_R(kjn, k0); E_R(cj, 4 * c + 1) = E_R(kjt, k0);
            E_R(cj, 4 * c + 2) = cnt > 1 ? E_R(kjn, k0 + 1) : R(0); E_R(cj, 4 * c + 3) = cnt > 1 ? E_R(kjt, k0 + 1) : R(0);
        } else {
            for (int i = 0; i < 4; i++) E_R(cj, 4 * c + i) = E_R(ncj, 4 * c + i);
        }
    }
    if (lane == 0) E_I(misc, M_NCACHE) = n;
}

// ---------------------------------------------------------------- state blobs <-> working set
// pose blob (P)  rows: [0, n_state_p)                 x / y / angle of the persistent bodies
// vel blob  (R)  rows: [0, n_state - n_state_p)       velocities + bias velocities
//                      then the env's five force limits, n_jacc joint accumulators, 4 * cache_slots contact impulses
//                      (the rows the host addresses come first: they are the same for every world of a task)
// int blob       rows: 0 episode steps, 1 n_cache, 2 overflow count, 3.. cache headers
// All blobs are [rows][N] (env index fastest) so lane<->env loads coalesce.
MGX_HD int state_rows_p(const TmplHeader &h) { return h.n_state_p; }
// velocities | the env's five force limits (max impulse per substep) | joint accumulators | contact cache impulses
MGX_HD int state_rows_f(const TmplHeader &h) { return (h.n_state - h.n_state_p) + N_PHYS_VARS + h.n_jacc + 4 * h.cache_slots; }
MGX_HD int state_row_physvar(const TmplHeader &h, int k) { return (h.n_state - h.n_state_p) + k; }
MGX_HD int state_row_jacc0(const TmplHeader &h) { return (h.n_state - h.n_state_p) + N_PHYS_VARS; }
MGX_HD int state_rows_i(const TmplHeader &h) { return 3 + h.cache_slots; }

template <typename R, typename P>
MGX_HD void ph_init_work(Env<R, P> &e, int lane, int nl) {
    // zero everything that is not persistent, set the static frame
    for (int b = lane; b < e.h->n_bodies; b += nl) {
        E_P(px, b) = P(0); E_P(py, b) = P(0); E_P(ang, b) = P(0); E_P(c, b) = P(1); E_P(s, b) = P(0);
        E_R(vx, b) = R(0); E_R(vy, b) = R(0); E_R(w, b) = R(0); E_R(vbx, b) = R(0); E_R(vby, b) = R(0); E_R(wb, b) = R(0);
    }
    for (int j = lane; j < e.h->n_joints; j += nl) { E_R(ja0, j) = R(0); E_R(ja1, j) = R(0); E_R(jrate, j) = R(0); }
    for (int c = lane; c < e.h->cache_slots; c += nl) E_MATCHED(c) = 0;
    if (lane == 0) for (int i = 0; i < M_N; i++) E_I(misc, i) = 0;
}

template <typename R, typename P>
MGX_HD void ph_load_state(Env<R, P> &e, const P *sp, const R *sf, const int32_t *si, long stride, long env, int lane, int nl) {
    const TmplHeader &h = *e.h;
    int nvel = state_row_jacc0(h);      // first row after the velocities and force limits
    for (int k = lane; k < h.n_state; k += nl) {
        int m = T_I(state_map, k), comp = m & 15, b = (m >> 4) & 0xFF, row = m >> 12;
        if (comp < 3) {
            P v = sp[(long)row * stride + env];
            if (comp == 0) E_P(px, b) = v; else if (comp == 1) E_P(py, b) = v; else E_P(ang, b) = v;
        } else {
            R v = sf[(long)row * stride + env];
            switch (comp) {
                case 3: E_R(vx, b) = v; break; case 4: E_R(vy, b) = v; break; case 5: E_R(w, b) = v; break;
                case 6: E_R(vbx, b) = v; break; case 7: E_R(vby, b) = v; break; default: E_R(wb, b) = v; break;
            }
        }
    }
    for (int j = lane; j < h.n_joints; j += nl) {
        int kind = T_I(joint_kind, j), off = nvel + T_I(joint_acc, j), pv = T_I(joint_pv, j);
        // max impulse per substep: this env's PhysicsVariables (rand_dynamics) or the template's constant
        E_R(jlim, j) = pv >= 0 ? sf[(long)state_row_physvar(h, pv) * stride + env] : T_R(joint_p, j * JOINT_PARAMS + 9);
        if (kind == J_SPRING) continue;
        E_R(ja0, j) = sf[(long)off * stride + env];
        if (kind == J_PIVOT) E_R(ja1, j) = sf[(long)(off + 1) * stride + env];
    }
    int ncache = si[1 * stride + env];
    for (int c = lane; c < ncache; c += nl) {
        E_I(chead, c) = si[(long)(3 + c) * stride + env];
        for (int i = 0; i < 4; i++) E_R(cj, 4 * c + i) = sf[(long)(nvel + h.n_jacc + 4 * c + i) * stride + env];
    }
    if (lane == 0) { E_I(misc, M_NCACHE) = ncache; E_I(misc, M_STEPS) = si[env]; E_I(misc, M_OVERFLOW) = si[2 * stride + env]; }
}
// after ph_load_state + barrier: trig of the loaded angles
template <typename R, typename P> MGX_HD void ph_refresh_trig(Env<R, P> &e, int lane, int nl) {
    for (int b = lane; b < e.h->n_bodies; b += nl) {
        if (T_I(body_type, b) == BODY_STATIC) continue;
        P s, c;
        r_sincos<P>(E_P(ang, b), s, c);
        E_P(c, b) = c; E_P(s, b) = s;
    }
}
// the pose rows alone (the state the rasteriser reads): final after the last substep's ph_integrate
template <typename R, typename P>
MGX_HD void ph_store_poses(Env<R, P> &e, P *sp, long stride, long env, int lane, int nl) {
    const TmplHeader &h = *e.h;
    for (int k = lane; k < h.n_state; k += nl) {
        int m = T_I(state_map, k), comp = m & 15, b = (m >> 4) & 0xFF, row = m >> 12;
        if (comp < 3) sp[(long)row * stride + env] = comp == 0 ? E_P(px, b) : (comp == 1 ? E_P(py, b) : E_P(ang, b));
    }
}
template <typename R, typename P>
MGX_HD void ph_store_state(Env<R, P> &e, P *sp, R *sf, int32_t *si, long stride, long env, int lane, int nl, bool poses_stored = false) {
    const TmplHeader &h = *e.h;
    int nvel = state_row_jacc0(h);      // first row after the velocities and force limits
    for (int k = lane; k < h.n_state; k += nl) {
        int m = T_I(state_map, k), comp = m & 15, b = (m >> 4) & 0xFF, row = m >> 12;
        if (comp < 3) {
            // (poses_stored: the rows went out after the last position update, ph_store_poses; the kinematic control body's angle is
            // set again by Robot.update at the end of the substep -- not drawn, re-derived at the next step's start -- and is stored here)
            if (!poses_stored || b == h.control_body) sp[(long)row * stride + env] = comp == 0 ? E_P(px, b) : (comp == 1 ? E_P(py, b) : E_P(ang, b));
        } else {
            R v;
            switch (comp) {
                case 3: v = E_R(vx, b); break; case 4: v = E_R(vy, b); break; case 5: v = E_R(w, b); break;
                case 6: v = E_R(vbx, b); break; case 7: v = E_R(vby, b); break; default: v = E_R(wb, b); break;
            }
            sf[(long)row * stride + env] = v;
        }
    }
    for (int j = lane; j < h.n_joints; j += nl) {
        int kind = T_I(joint_kind, j), off = nvel + T_I(joint_acc, j);
        if (kind == J_SPRING) continue;
        sf[(long)off * stride + env] = E_R(ja0, j);
        if (kind == J_PIVOT) sf[(long)(off + 1) * stride + env] = E_R(ja1, j);
    }
    int ncache = E_I(misc, M_NCACHE);
    for (int c = lane; c < ncache; c += nl) {
        si[(long)(3 + c) * stride + env] = E_I(chead, c);
        for (int i = 0; i < 4; i++) sf[(long)(nvel + h.n_jacc + 4 * c + i) * stride + env] = E_R(cj, 4 * c + i);
    }
    if (lane == 0) { si[env] = E_I(misc, M_STEPS); si[1 * stride + env] = ncache; si[2 * stride + env] = E_I(misc, M_OVERFLOW); }
}

// BaseEnv.reset() for one env: template poses, zero velocities / accumulators / cache.
// Bodies with a parent (finger roots) are placed with the SAME rounding sequence the pin-joint
// preStep uses, so the zero-length PinJoint starts with delta == 0 exactly, as in the reference.
// ent_pose (optional): [n_entities * 3][stride] per-env (x, y, angle) of every entity (Test*Jitter / Layout variants,
// geom.py pm_shift_bodies): the entity's bodies follow it rigidly -- the main body takes the pose, the eyes its angle,
// the finger roots are re-derived from the moved robot exactly as at construction.
template <typename R, typename P>
MGX_HD void reset_env_state(const TmplHeader &h, const int32_t *ti, const R *tr, const P *tp, P *sp, R *sf, int32_t *si, long stride, long env,
                            const P *ent_pose = nullptr) {
    TmplOff to(h);
    for (int k = 0; k < h.n_state; k++) {
        int m = ti[to.state_map + k], comp = m & 15, b = (m >> 4) & 0xFF, row = m >> 12;
        if (comp >= 3) { sf[(long)row * stride + env] = R(0); continue; }
        P v;
        int parent = ti[to.body_parent + b];
        int ent = ent_pose ? ti[to.body_ent + b] : -1;
        if (parent >= 0 && comp < 2) {
            // parent = the robot's main body: its pose is the template's or this env's entity pose
            P px = tp[to.p_body_init + 3 * parent], py = tp[to.p_body_init + 3 * parent + 1], pa = tp[to.p_body_init + 3 * parent + 2];
            if (ent >= 0) { px = ent_pose[(long)(3 * ent) * stride + env]; py = ent_pose[(long)(3 * ent + 1) * stride + env];
                            pa = ent_pose[(long)(3 * ent + 2) * stride + env] + tp[to.p_body_aoff + parent]; }
            P s, c, rx, ry;
            r_sincos<P>(pa, s, c);
            anchor_rot<P>(c, s, tp[to.p_body_anchor + 2 * b], tp[to.p_body_anchor + 2 * b + 1], rx, ry);
            v = comp == 0 ? r_add_nc<P>(px, rx) : r_add_nc<P>(py, ry);
        } else if (ent >= 0) {
            v = ent_pose[(long)(3 * ent + comp) * stride + env];
            if (comp == 2) v = v + tp[to.p_body_aoff + b];
        } else {
            v = tp[to.p_body_init + 3 * b + comp];
        }
        sp[(long)row * stride + env] = v;
    }
    int nvel = state_row_jacc0(h);      // first row after the velocities and force limits
    for (int k = 0; k < h.n_jacc + 4 * h.cache_slots; k++) sf[(long)(nvel + k) * stride + env] = R(0);
    for (int k = 0; k < N_PHYS_VARS; k++) sf[(long)state_row_physvar(h, k) * stride + env] = tr[to.consts + C_PV0 + k];
    for (int k = 0; k < 3 + h.cache_slots; k++) si[(long)k * stride + env] = 0;
}

}  // namespace mgx

// One physics substep as a list of phases; X(stmt) runs `stmt` for (lane, nl) and then
// synchronises the env's lane group.  Used by mgx_step.hip (device) and tests/emu (host).
// (everything after the position update: a substep's poses are final once ph_integrate has run -- what follows finds the collisions
// at the new positions and solves for the velocities that the NEXT substep integrates, cpSpaceStep's order)
#if MGX_BROAD_SAP
#define MGX_BROAD_SAP_PHASES(X) X(ph_broad_sweep(e, lane, nl)) X(ph_broad_emit(e, lane, nl))
#else
#define MGX_BROAD_SAP_PHASES(X)
#endif
#define MGX_SUBSTEP_PHASES(X)                                      \
    X(ph_integrate(e, lane, nl))                                   \
    MGX_SUBSTEP_AFTER_INTEGRATE(X)
#define MGX_SUBSTEP_AFTER_INTEGRATE(X)                             \
    X(ph_shapes(e, lane, nl))                                      \
    X(ph_broad(e, lane, nl))                                       \
    MGX_BROAD_SAP_PHASES(X)                                        \
    X(ph_narrow(e, lane, nl))                                      \
    X(ph_arbiters_joints(e, ctx, lane, nl))                        \
    X(solve_begin(e, ctx, lane, nl))                               \
    X(solve_warm_contacts(e, ctx, lane))                           \
    X(solve_warm_pg(e, ctx, lane))                                 \
    X(solve_warm_chain(ctx))                                       \
    MGX_SOLVE_ITERATIONS(X)                                        \
    X(solve_end(e, ctx, lane))                                     \
    X(ph_cache_commit(e, lane, nl))

// `iterations` Gauss-Seidel sweeps; each is four lane-group-synchronised steps (ctx = this lane's SolveCtx)
#define MGX_SOLVE_ITERATIONS(X)                                    \
    for (int it_ = 0; it_ < iterations; it_++) {                   \
        X(solve_iter_publish(e, ctx, lane))                        \
        X(solve_iter_contacts(e, ctx, lane))                       \
        X(solve_iter_pg(e, ctx, lane))                             \
        X(solve_iter_chain(ctx))                                   \
    }


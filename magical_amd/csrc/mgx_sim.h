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
template <> MGX_HD void r_sincos<double>(double a, double &s, double &c) { s = sin(a); c = cos(a); }
template <typename R> MGX_HD R r_abs(R x) { return x < R(0) ? -x : x; }
template <typename R> MGX_HD R r_min(R a, R b) { return a < b ? a : b; }
template <typename R> MGX_HD R r_max(R a, R b) { return a > b ? a : b; }
template <typename R> MGX_HD R r_clamp(R f, R lo, R hi) { return r_min(r_max(f, lo), hi); }
template <typename R> MGX_HD R r_clamp01(R f) { return r_max(R(0), r_min(f, R(1))); }
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
        for (int i = 0; i < nv; i++) {
            R lx = T_R(lvx, vo + i), ly = T_R(lvy, vo + i);
            R x = bx + (c * lx - s * ly), y = by + (c * ly + s * lx);
            E_R(wx, vo + i) = x; E_R(wy, vo + i) = y;
            R nx = T_R(lnx, vo + i), ny = T_R(lny, vo + i);
            E_R(wnx, vo + i) = c * nx - s * ny; E_R(wny, vo + i) = c * ny + s * nx;
            l = r_min(l, x); r = r_max(r, x); bo = r_min(bo, y); t = r_max(t, y);
        }
        E_R(bbl, sh) = l - rad; E_R(bbb, sh) = bo - rad; E_R(bbr, sh) = r + rad; E_R(bbt, sh) = t + rad;
    }
}

// ---------------------------------------------------------------- phase: broadphase
// The candidate list already encodes QueryReject's body/group filters; what is left per substep is
// the BB test.  Each lane owns a contiguous chunk so that the compacted list keeps pair order
// (= arbiter solve order).
template <typename R, typename P> MGX_HD void ph_broad_count(Env<R, P> &e, int lane, int nl) {
    int np = e.h->n_pairs, chunk = (np + nl - 1) / nl;
    int p0 = lane * chunk, p1 = p0 + chunk < np ? p0 + chunk : np;
    uint8_t *flag = reinterpret_cast<uint8_t *>(&E_I(flag, 0));
    int count = 0;
    for (int p = p0; p < p1; p++) {
        int pr = T_I(pair, p), a = pr & 0xFF, b = pr >> 8;
        bool hit = E_R(bbl, a) <= E_R(bbr, b) && E_R(bbl, b) <= E_R(bbr, a) &&
                   E_R(bbb, a) <= E_R(bbt, b) && E_R(bbb, b) <= E_R(bbt, a);
        flag[p] = hit ? 1 : 0;
        count += hit ? 1 : 0;
    }
    E_I(cnt, lane) = count;
}
template <typename R, typename P> MGX_HD void ph_broad_write(Env<R, P> &e, int lane, int nl) {
    int np = e.h->n_pairs, chunk = (np + nl - 1) / nl;
    int p0 = lane * chunk, p1 = p0 + chunk < np ? p0 + chunk : np;
    const uint8_t *flag = reinterpret_cast<const uint8_t *>(&E_I(flag, 0));
    int off = 0, total = 0;
    for (int l = 0; l < nl; l++) { int c = E_I(cnt, l); if (l < lane) off += c; total += c; }
    int cap = e.h->max_overlaps;
    for (int p = p0; p < p1; p++)
        if (flag[p]) { if (off < cap) E_I(ov, off) = p; off++; }
    if (lane == 0) {
        if (total > cap) { E_I(misc, M_OVERFLOW) += 1; total = cap; }
        E_I(misc, M_NOV) = total;
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
// cpCollision.c SupportEdgeForPoly / SupportEdgeForSegment
template <typename R, typename P> MGX_HD EdgeRef<R> support_edge(const Env<R, P> &e, int vo, int nv, R rad, R nx, R ny) {
    int i1 = support_index(e, vo, nv, nx, ny);
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
        R *q = p + 4 * count; q[0] = p1x; q[1] = p1y; q[2] = p2x; q[3] = p2y;
        if (count == 0) h0 = hash; else h1 = hash;
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
MGX_HD bool poly_axis(const Env<R, P> &e, int voa, int na, int vob, int nb, R rsum, R &nx, R &ny, R &d) {
    R best = -r_inf<R>(); int best_i = 0; bool best_a = true;
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
    if (best_a) { nx = E_R(wnx, voa + best_i); ny = E_R(wny, voa + best_i); }
    else { nx = -E_R(wnx, vob + best_i); ny = -E_R(wny, vob + best_i); }
    d = best;
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
        if (poly_axis(e, voa, na, vob, nb, ra + rb, nx, ny, d)) {
            EdgeRef<R> e1 = support_edge(e, voa, na, ra, nx, ny);
            EdgeRef<R> e2 = support_edge(e, vob, nb, rb, -nx, -ny);
            contact_points(e1, e2, nx, ny, d, m);
        }
    }
}

// ---------------------------------------------------------------- phase: narrowphase
template <typename R, typename P> MGX_HD void ph_narrow(Env<R, P> &e, int lane, int nl) {
    int nov = E_I(misc, M_NOV);
    for (int q = lane; q < nov; q += nl) {
        int pr = T_I(pair, E_I(ov, q));
        ManifoldOut<R> m;
        collide_pair(e, pr & 0xFF, pr >> 8, m);
        E_I(mcnt, q) = m.count | ((m.h0 | (m.h1 << 8)) << 8);        // point count (0..2) | the two point hashes
        E_R(mn, 2 * q) = m.nx; E_R(mn, 2 * q + 1) = m.ny;
        for (int i = 0; i < 4 * m.count; i++) E_R(mp, 8 * q + i) = m.p[i];
    }
}

// ---------------------------------------------------------------- phase: arbiters (cpArbiterUpdate + cpArbiterPreStep)
// and joint preStep for every joint kind that does not touch velocities.
template <typename R, typename P> MGX_HD void ph_arbiters_joints(Env<R, P> &e, int lane, int nl) {
    int nov = E_I(misc, M_NOV), ncache = E_I(misc, M_NCACHE);
    int kcap = e.h->max_contacts, ccap = e.h->cache_slots;
    R dt = e.cst(C_DT), slop = e.cst(C_SLOP), brate = e.cst(C_CONTACT_BIAS_RATE);
    int koff = 0, rank = 0, scanned = 0;
    for (int q = lane; q < nov; q += nl) {
        for (; scanned < q; scanned++) { int c = E_I(mcnt, scanned) & 3; if (c > 0 && koff + c <= kcap && rank < ccap) { koff += c; rank++; } }
        const int mc = E_I(mcnt, q), cnt = mc & 3;
        scanned = q + 1;                                      // this entry is accounted for right below
        if (cnt == 0) continue;
        if (koff + cnt > kcap || rank >= ccap) continue;     // dropped: counted by lane 0 below
        int p = E_I(ov, q), pr = T_I(pair, p), sa = pr & 0xFF, sb = pr >> 8;
        int A = T_I(shape_body, sa), B = T_I(shape_body, sb);
        // cached arbiter for this shape pair?
        int ci = -1; uint32_t old = 0;
        for (int c = 0; c < ncache; c++) { uint32_t hd = (uint32_t)E_I(chead, c); if ((int)(hd & 0xFFFu) == p) { ci = c; old = hd; } }
        bool first = true;
        if (ci >= 0) { first = ((old >> 12) & 3u) != 0u; E_I(cmatched, ci) = 1; }
        int ocnt = ci >= 0 ? (int)((old >> 14) & 3u) : 0;
        int oh[2] = {(int)((old >> 16) & 0xFFu), (int)((old >> 24) & 0xFFu)};
        int mh = mc >> 8;
        int nh[2] = {mh & 0xFF, (mh >> 8) & 0xFF};
        R nx = E_R(mn, 2 * q), ny = E_R(mn, 2 * q + 1);
        R mu = T_R(shape_u, sa) * T_R(shape_u, sb);
        R ma = T_R(body_minv, A), ia = T_R(body_iinv, A), mb = T_R(body_minv, B), ib = T_R(body_iinv, B);
        R pax = R(E_P(px, A)), pay = R(E_P(py, A)), pbx = R(E_P(px, B)), pby = R(E_P(py, B));
        for (int i = 0; i < cnt; i++) {
            int k = koff + i;
            R r1x = E_R(mp, 8 * q + 4 * i) - pax, r1y = E_R(mp, 8 * q + 4 * i + 1) - pay;
            R r2x = E_R(mp, 8 * q + 4 * i + 2) - pbx, r2y = E_R(mp, 8 * q + 4 * i + 3) - pby;
            R jn = 0, jt = 0;
            for (int o = 0; o < ocnt; o++) if (oh[o] == nh[i]) { jn = E_R(cj, 4 * ci + 2 * o); jt = E_R(cj, 4 * ci + 2 * o + 1); }
            R rcn1 = r1x * ny - r1y * nx, rcn2 = r2x * ny - r2y * nx;          // cross(r, n)
            R rct1 = r1x * nx + r1y * ny, rct2 = r2x * nx + r2y * ny;          // cross(r, perp(n))
            R kn = ma + ia * rcn1 * rcn1 + mb + ib * rcn2 * rcn2;
            R kt = ma + ia * rct1 * rct1 + mb + ib * rct2 * rct2;
            R dist = ((r2x - r1x) + (pbx - pax)) * nx + ((r2y - r1y) + (pby - pay)) * ny;
            E_I(kab, k) = A | (B << 8); E_I(kfirst, k) = first ? 1 : 0;
            E_R(knx, k) = nx; E_R(kny, k) = ny;
            E_R(kr1x, k) = r1x; E_R(kr1y, k) = r1y; E_R(kr2x, k) = r2x; E_R(kr2y, k) = r2y;
            E_R(knm, k) = R(1) / kn; E_R(ktm, k) = R(1) / kt;
            E_R(kbias, k) = -brate * r_min(R(0), dist + slop);
            E_R(kjb, k) = R(0); E_R(kjn, k) = jn; E_R(kjt, k) = jt; E_R(kmu, k) = mu;
        }
        E_I(nchead, rank) = (int32_t)cache_pack((uint32_t)p, 0u, (uint32_t)cnt, (uint32_t)nh[0], (uint32_t)nh[1]);
        E_I(koff, rank) = koff;
        koff += cnt; rank++;
    }
    if (lane == 0) {   // totals (lane 0 rescans everything once; also counts drops)
        int k = 0, r = 0, dropped = 0;
        for (int q = 0; q < nov; q++) { int c = E_I(mcnt, q) & 3; if (c > 0) { if (k + c <= kcap && r < ccap) { k += c; r++; } else dropped++; } }
        E_I(misc, M_NK) = k; E_I(misc, M_NARB) = r;
        if (dropped) E_I(misc, M_OVERFLOW) += dropped;
    }
    // joint preStep (all kinds except the damped rotary spring, which applies its torque right away).
    // Anchor separations and angle differences are formed in pose precision, then narrowed.
    (void)dt;
    for (int j = lane; j < e.h->n_joints; j += nl) {
        int kind = T_I(joint_kind, j), a = T_I(joint_a, j), b = T_I(joint_b, j);
        const R *p = &T_R(joint_p, j * JOINT_PARAMS);
        const P *pp = &T_P(p_joint, j * 7);
        if (kind == J_PIVOT || kind == J_PIN) {
            P q1x, q1y, q2x, q2y;
            anchor_rot<P>(E_P(c, a), E_P(s, a), pp[0], pp[1], q1x, q1y);
            anchor_rot<P>(E_P(c, b), E_P(s, b), pp[2], pp[3], q2x, q2y);
            R r1x = R(q1x), r1y = R(q1y), r2x = R(q2x), r2y = R(q2y);
            E_R(jr1x, j) = r1x; E_R(jr1y, j) = r1y; E_R(jr2x, j) = r2x; E_R(jr2y, j) = r2y;
            P ddx = r_add_nc<P>(E_P(px, b), q2x) - r_add_nc<P>(E_P(px, a), q1x);
            P ddy = r_add_nc<P>(E_P(py, b), q2y) - r_add_nc<P>(E_P(py, a), q1y);
            R ma = T_R(body_minv, a), ia = T_R(body_iinv, a), mb = T_R(body_minv, b), ib = T_R(body_iinv, b);
            if (kind == J_PIVOT) {
                R dx = R(ddx), dy = R(ddy);
                R msum = ma + mb;
                R k11 = msum + r1y * r1y * ia + r2y * r2y * ib;
                R k12 = -r1x * r1y * ia - r2x * r2y * ib;
                R k22 = msum + r1x * r1x * ia + r2x * r2x * ib;
                R det_inv = R(1) / (k11 * k22 - k12 * k12);
                E_R(jk0, j) = k22 * det_inv; E_R(jk1, j) = -k12 * det_inv; E_R(jk2, j) = -k12 * det_inv; E_R(jk3, j) = k11 * det_inv;
                R bx = -dx * p[7], by = -dy * p[7], mb_ = p[8];
                R bl2 = bx * bx + by * by;
                if (bl2 > mb_ * mb_) { R sc = mb_ / (r_sqrt(bl2) + r_tiny<R>()); bx *= sc; by *= sc; }
                E_R(jb0, j) = bx; E_R(jb1, j) = by;
            } else {
                P dist = r_sqrt<P>(ddx * ddx + ddy * ddy);
                P inv = dist != P(0) ? P(1) / dist : P(0);
                R nx = R(ddx * inv), ny = R(ddy * inv);
                R rcn1 = r1x * ny - r1y * nx, rcn2 = r2x * ny - r2y * nx;
                E_R(jk0, j) = nx; E_R(jk1, j) = ny;
                E_R(jk2, j) = R(1) / (ma + ia * rcn1 * rcn1 + mb + ib * rcn2 * rcn2);
                E_R(jb0, j) = r_clamp(-R(dist - pp[4]) * p[7], -p[8], p[8]);
            }
        } else if (kind == J_GEAR) {
            E_R(jb0, j) = r_clamp(-R(E_P(ang, b) * pp[5] - E_P(ang, a) - pp[4]) * p[7], -p[8], p[8]);
        } else if (kind == J_LIMIT) {
            P dist = E_P(ang, b) - E_P(ang, a), pdist = P(0);
            if (dist > pp[5]) pdist = pp[5] - dist; else if (dist < pp[4]) pdist = pp[4] - dist;
            R bias = r_clamp(-R(pdist) * p[7], -p[8], p[8]);
            E_R(jb0, j) = bias;
            if (bias == R(0)) E_R(ja0, j) = R(0);
        }
    }
}

// ---------------------------------------------------------------- velocity helpers for the solver
template <typename R> struct Vel { R vx, vy, w; };
#define LOADV(b) Vel<R>{E_R(vx, b), E_R(vy, b), E_R(w, b)}
#define STOREV(b, v) do { E_R(vx, b) = (v).vx; E_R(vy, b) = (v).vy; E_R(w, b) = (v).w; } while (0)

template <typename R, typename P> MGX_HD void joint_apply_cached(Env<R, P> &e, int j) {
    int kind = T_I(joint_kind, j), a = T_I(joint_a, j), b = T_I(joint_b, j);
    const R *p = &T_R(joint_p, j * JOINT_PARAMS);
    R ma = T_R(body_minv, a), ia = T_R(body_iinv, a), mb = T_R(body_minv, b), ib = T_R(body_iinv, b);
    if (kind == J_PIVOT || kind == J_PIN) {
        R jx, jy;
        if (kind == J_PIVOT) { jx = E_R(ja0, j); jy = E_R(ja1, j); }
        else { jx = E_R(jk0, j) * E_R(ja0, j); jy = E_R(jk1, j) * E_R(ja0, j); }
        R r1x = E_R(jr1x, j), r1y = E_R(jr1y, j), r2x = E_R(jr2x, j), r2y = E_R(jr2y, j);
        E_R(vx, a) -= jx * ma; E_R(vy, a) -= jy * ma; E_R(w, a) -= ia * (r1x * jy - r1y * jx);
        E_R(vx, b) += jx * mb; E_R(vy, b) += jy * mb; E_R(w, b) += ib * (r2x * jy - r2y * jx);
    } else if (kind == J_GEAR) {
        R jj = E_R(ja0, j);
        E_R(w, a) -= jj * ia * (R(1) / p[5]); E_R(w, b) += jj * ib;
    } else if (kind == J_LIMIT || kind == J_MOTOR) {
        R jj = E_R(ja0, j);
        E_R(w, a) -= jj * ia; E_R(w, b) += jj * ib;
    }
}

template <typename R, typename P> MGX_HD void joint_apply_impulse(Env<R, P> &e, int j) {
    int kind = T_I(joint_kind, j), a = T_I(joint_a, j), b = T_I(joint_b, j);
    const R *p = &T_R(joint_p, j * JOINT_PARAMS);
    R ia = T_R(body_iinv, a), ib = T_R(body_iinv, b);
    switch (kind) {
    case J_PIVOT: {
        R ma = T_R(body_minv, a), mb = T_R(body_minv, b);
        R r1x = E_R(jr1x, j), r1y = E_R(jr1y, j), r2x = E_R(jr2x, j), r2y = E_R(jr2y, j);
        Vel<R> va = LOADV(a), vb = LOADV(b);
        R vrx = (vb.vx - r2y * vb.w) - (va.vx - r1y * va.w), vry = (vb.vy + r2x * vb.w) - (va.vy + r1x * va.w);
        R dx = E_R(jb0, j) - vrx, dy = E_R(jb1, j) - vry;
        R jx = dx * E_R(jk0, j) + dy * E_R(jk1, j), jy = dx * E_R(jk2, j) + dy * E_R(jk3, j);
        R ox = E_R(ja0, j), oy = E_R(ja1, j);
        R nxv = ox + jx, nyv = oy + jy, lim = E_R(jlim, j);
        R l2 = nxv * nxv + nyv * nyv;
        if (l2 > lim * lim) { R sc = lim / (r_sqrt(l2) + r_tiny<R>()); nxv *= sc; nyv *= sc; }
        E_R(ja0, j) = nxv; E_R(ja1, j) = nyv;
        jx = nxv - ox; jy = nyv - oy;
        va.vx -= jx * ma; va.vy -= jy * ma; va.w -= ia * (r1x * jy - r1y * jx);
        vb.vx += jx * mb; vb.vy += jy * mb; vb.w += ib * (r2x * jy - r2y * jx);
        STOREV(a, va); STOREV(b, vb);
    } break;
    case J_GEAR: {
        R ratio = p[5], ratio_inv = R(1) / ratio;
        R wa = E_R(w, a), wb = E_R(w, b);
        R wr = wb * ratio - wa;
        R jmax = E_R(jlim, j);
        R jj = (E_R(jb0, j) - wr) * p[0];
        R jold = E_R(ja0, j);
        R jn = r_clamp(jold + jj, -jmax, jmax);
        E_R(ja0, j) = jn; jj = jn - jold;
        E_R(w, a) = wa - jj * ia * ratio_inv; E_R(w, b) = wb + jj * ib;
    } break;
    case J_SPRING: {
        R wa = E_R(w, a), wb = E_R(w, b);
        R wrn = wa - wb;
        R w_damp = (E_R(jrate, j) - wrn) * p[6];
        E_R(jrate, j) = wrn + w_damp;                       // target_wrn
        R j_damp = w_damp * p[0];
        E_R(w, a) = wa + j_damp * ia; E_R(w, b) = wb - j_damp * ib;
    } break;
    case J_PIN: {
        R ma = T_R(body_minv, a), mb = T_R(body_minv, b);
        R r1x = E_R(jr1x, j), r1y = E_R(jr1y, j), r2x = E_R(jr2x, j), r2y = E_R(jr2y, j);
        R nx = E_R(jk0, j), ny = E_R(jk1, j);
        Vel<R> va = LOADV(a), vb = LOADV(b);
        R vrx = (vb.vx - r2y * vb.w) - (va.vx - r1y * va.w), vry = (vb.vy + r2x * vb.w) - (va.vy + r1x * va.w);
        R vrn = vrx * nx + vry * ny;
        R jmax = E_R(jlim, j);
        R jn = (E_R(jb0, j) - vrn) * E_R(jk2, j);
        R jold = E_R(ja0, j);
        R jnew = r_clamp(jold + jn, -jmax, jmax);
        E_R(ja0, j) = jnew; jn = jnew - jold;
        R jx = nx * jn, jy = ny * jn;
        va.vx -= jx * ma; va.vy -= jy * ma; va.w -= ia * (r1x * jy - r1y * jx);
        vb.vx += jx * mb; vb.vy += jy * mb; vb.w += ib * (r2x * jy - r2y * jx);
        STOREV(a, va); STOREV(b, vb);
    } break;
    case J_LIMIT: {
        R bias = E_R(jb0, j);
        if (bias == R(0)) return;
        R wa = E_R(w, a), wb = E_R(w, b);
        R wr = wb - wa;
        R jmax = E_R(jlim, j);
        R jj = -(bias + wr) * p[0];
        R jold = E_R(ja0, j);
        R jn = bias < R(0) ? r_clamp(jold + jj, R(0), jmax) : r_clamp(jold + jj, -jmax, R(0));
        E_R(ja0, j) = jn; jj = jn - jold;
        E_R(w, a) = wa - jj * ia; E_R(w, b) = wb + jj * ib;
    } break;
    case J_MOTOR: {
        R wa = E_R(w, a), wb = E_R(w, b);
        R wr = wb - wa + E_R(jrate, j);
        R jmax = E_R(jlim, j);
        R jj = -wr * p[0];
        R jold = E_R(ja0, j);
        R jn = r_clamp(jold + jj, -jmax, jmax);
        E_R(ja0, j) = jn; jj = jn - jold;
        E_R(w, a) = wa - jj * ia; E_R(w, b) = wb + jj * ib;
    } break;
    }
}

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
    c.a = ab & 0xFF; c.b = ab >> 8;
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
// Chipmunk's order per substep: (arbiter preStep, joint preStep incl. the spring torque) -> cached arbiter
// impulses -> cached joint impulses -> 10 x { all arbiters in pair order ; all joints in insertion order }.
// Joints only couple bodies inside an ISLAND (the robot's 10 joints; each block's {pivot, gear} to the static
// body), so inside the joint half of an iteration islands can run on different lanes and still apply, to every
// body, exactly the sequence of impulses the serial order applies: the result is bit-identical.  The robot
// island is solved entirely in registers (6 bodies x 3 velocities + 10 unrolled joint rows); contacts stay
// sequential on lane 0 against the LDS copy of the velocities, with a store / barrier / load hand-over around
// them only when the env has contacts at all.

constexpr int RI_JOINTS = 10, RI_BODIES = 6;   // Robot.setup (entities.py:238-354): pivot gear spring spring {pin limit motor} x2
// kinds and body slots (0 control, 1 robot, 2 eye L, 3 eye R, 4 finger L, 5 finger R) of the robot's joints
#define RI_KIND(j) ((j) == 0 ? J_PIVOT : (j) == 1 ? J_GEAR : (j) <= 3 ? J_SPRING : ((j) - 4) % 3 == 0 ? J_PIN : ((j) - 4) % 3 == 1 ? J_LIMIT : J_MOTOR)
#define RI_SA(j) ((j) <= 1 ? 0 : 1)
#define RI_SB(j) ((j) <= 1 ? 1 : (j) == 2 ? 2 : (j) == 3 ? 3 : (j) <= 6 ? 4 : 5)

template <typename R> struct SolveCtx {
    // robot island (only meaningful on lane 0 of the env's group)
    R vx[RI_BODIES], vy[RI_BODIES], w[RI_BODIES], minv[RI_BODIES], iinv[RI_BODIES];
    R f[RI_JOINTS][12], lim[RI_JOINTS];
    // one block island per lane (lanes 1..): pivot k (2x2), accumulators, limits, gear effective mass
    R bvx, bvy, bw, bminv, biinv, bk[4], bacc[3], blim[2], bgear, bbias[3];
    int bbody, bj;            // body / first joint of the register-resident block (-1: none)
    int has_contacts;
};

// ZA / ZB: the joint's anchor on body a / b is the body origin (r = 0, so the r x j terms vanish); KA: body a is the
// kinematic control body (inverse mass and inertia 0: impulses leave it unchanged).  What is skipped is x - 0 * y and
// x + 0 * y on finite values, i.e. x: Robot.setup's pivot / gear to the control body and the finger pins (entities.py:
// 255-263,334-341) have exactly these shapes.
template <typename R, bool ZA = false, bool ZB = false, bool KA = false>
MGX_HD void reg_apply_joint(int kind, R *f, R lim, R ma, R ia, R mb, R ib,
                            R &avx, R &avy, R &aw, R &bvx, R &bvy, R &bw) {
    switch (kind) {
    case J_PIVOT: {   // f: r1x r1y r2x r2y k0 k1 k2 k3 bias0 bias1 acc0 acc1
        R vbx_ = ZB ? bvx : bvx - f[3] * bw, vby_ = ZB ? bvy : bvy + f[2] * bw;
        R vax_ = ZA ? avx : avx - f[1] * aw, vay_ = ZA ? avy : avy + f[0] * aw;
        R vrx = vbx_ - vax_, vry = vby_ - vay_;
        R dx = f[8] - vrx, dy = f[9] - vry;
        R jx = dx * f[4] + dy * f[5], jy = dx * f[6] + dy * f[7];
        R ox = f[10], oy = f[11];
        R nxv = ox + jx, nyv = oy + jy;
        R l2 = nxv * nxv + nyv * nyv;
        if (l2 > lim * lim) { R sc = lim / (r_sqrt(l2) + r_tiny<R>()); nxv *= sc; nyv *= sc; }
        f[10] = nxv; f[11] = nyv;
        jx = nxv - ox; jy = nyv - oy;
        if (!KA) { avx -= jx * ma; avy -= jy * ma; if (!ZA) aw -= ia * (f[0] * jy - f[1] * jx); }
        bvx += jx * mb; bvy += jy * mb; if (!ZB) bw += ib * (f[2] * jy - f[3] * jx);
    } break;
    case J_GEAR: {    // f: imass bias acc ratio 1/ratio
        R ratio = f[3], ratio_inv = f[4];
        R wr = bw * ratio - aw;
        R jj = (f[1] - wr) * f[0];
        R jold = f[2];
        R jn = r_clamp(jold + jj, -lim, lim);
        f[2] = jn; jj = jn - jold;
        if (!KA) aw = aw - jj * ia * ratio_inv;
        bw = bw + jj * ib;
    } break;
    case J_SPRING: {  // f: imass w_coef target_wrn
        R wrn = aw - bw;
        R w_damp = (f[2] - wrn) * f[1];
        f[2] = wrn + w_damp;
        R j_damp = w_damp * f[0];
        aw = aw + j_damp * ia; bw = bw - j_damp * ib;
    } break;
    case J_PIN: {     // f: r1x r1y r2x r2y nx ny nmass bias acc
        R vrx = (ZB ? bvx : bvx - f[3] * bw) - (avx - f[1] * aw), vry = (ZB ? bvy : bvy + f[2] * bw) - (avy + f[0] * aw);
        R vrn = vrx * f[4] + vry * f[5];
        R jn = (f[7] - vrn) * f[6];
        R jold = f[8];
        R jnew = r_clamp(jold + jn, -lim, lim);
        f[8] = jnew; jn = jnew - jold;
        R jx = f[4] * jn, jy = f[5] * jn;
        avx -= jx * ma; avy -= jy * ma; aw -= ia * (f[0] * jy - f[1] * jx);
        bvx += jx * mb; bvy += jy * mb; if (!ZB) bw += ib * (f[2] * jy - f[3] * jx);
    } break;
    case J_LIMIT: {   // f: imass bias acc
        R bias = f[1];
        if (bias == R(0)) return;
        R wr = bw - aw;
        R jj = -(bias + wr) * f[0];
        R jold = f[2];
        R jn = bias < R(0) ? r_clamp(jold + jj, R(0), lim) : r_clamp(jold + jj, -lim, R(0));
        f[2] = jn; jj = jn - jold;
        aw = aw - jj * ia; bw = bw + jj * ib;
    } break;
    default: {        // J_MOTOR, f: imass rate acc
        R wr = bw - aw + f[1];
        R jj = -wr * f[0];
        R jold = f[2];
        R jn = r_clamp(jold + jj, -lim, lim);
        f[2] = jn; jj = jn - jold;
        aw = aw - jj * ia; bw = bw + jj * ib;
    } break;
    }
}
template <typename R> MGX_HD void reg_apply_cached(int kind, const R *f, R ma, R ia, R mb, R ib,
                                                   R &avx, R &avy, R &aw, R &bvx, R &bvy, R &bw) {
    if (kind == J_PIVOT || kind == J_PIN) {
        R jx, jy;
        if (kind == J_PIVOT) { jx = f[10]; jy = f[11]; } else { jx = f[4] * f[8]; jy = f[5] * f[8]; }
        avx -= jx * ma; avy -= jy * ma; aw -= ia * (f[0] * jy - f[1] * jx);
        bvx += jx * mb; bvy += jy * mb; bw += ib * (f[2] * jy - f[3] * jx);
    } else if (kind == J_GEAR) {
        R jj = f[2];
        aw -= jj * ia * (R(1) / f[3]); bw += jj * ib;
    } else if (kind == J_LIMIT || kind == J_MOTOR) {
        R jj = f[2];
        aw -= jj * ia; bw += jj * ib;
    }
}

template <typename R, typename P> MGX_HD int ri_body(const Env<R, P> &e, int slot) {
    const TmplHeader &h = *e.h;
    return slot == 0 ? h.control_body : slot == 1 ? h.robot_body : slot == 2 ? h.eye_body[0] : slot == 3 ? h.eye_body[1]
         : slot == 4 ? h.finger_body[0] : h.finger_body[1];
}
template <typename R, typename P> MGX_HD void ri_load_vel(const Env<R, P> &e, SolveCtx<R> &c) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int s = 0; s < RI_BODIES; s++) { int b = ri_body(e, s); c.vx[s] = E_R(vx, b); c.vy[s] = E_R(vy, b); c.w[s] = E_R(w, b); }
}
template <typename R, typename P> MGX_HD void ri_store_vel(Env<R, P> &e, const SolveCtx<R> &c) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int s = 1; s < RI_BODIES; s++) { int b = ri_body(e, s); E_R(vx, b) = c.vx[s]; E_R(vy, b) = c.vy[s]; E_R(w, b) = c.w[s]; }
}
// which block (island) does this lane keep in registers?  lane 1 + k <-> island k when every island gets a lane
template <typename R, typename P> MGX_HD int lane_island(const Env<R, P> &e, int lane, int nl) {
    return (e.h->n_islands <= nl - 1 && lane >= 1 && lane - 1 < e.h->n_islands) ? lane - 1 : -1;
}
template <typename R, typename P> MGX_HD void bi_load_vel(const Env<R, P> &e, SolveCtx<R> &c) {
    if (c.bbody >= 0) { c.bvx = E_R(vx, c.bbody); c.bvy = E_R(vy, c.bbody); c.bw = E_R(w, c.bbody); }
}
template <typename R, typename P> MGX_HD void bi_store_vel(Env<R, P> &e, const SolveCtx<R> &c) {
    if (c.bbody >= 0) { E_R(vx, c.bbody) = c.bvx; E_R(vy, c.bbody) = c.bvy; E_R(w, c.bbody) = c.bw; }
}
template <typename R> MGX_HD void bi_iterate(SolveCtx<R> &c) {
    if (c.bbody < 0) return;
    // PivotJoint(static, block): a = static (m_inv = i_inv = 0, v = 0), r1 = r2 = 0
    {
        R dx = c.bbias[0] - c.bvx, dy = c.bbias[1] - c.bvy;
        R jx = dx * c.bk[0] + dy * c.bk[1], jy = dx * c.bk[2] + dy * c.bk[3];
        R ox = c.bacc[0], oy = c.bacc[1];
        R nxv = ox + jx, nyv = oy + jy, lim = c.blim[0];
        R l2 = nxv * nxv + nyv * nyv;
        if (l2 > lim * lim) { R sc = lim / (r_sqrt(l2) + r_tiny<R>()); nxv *= sc; nyv *= sc; }
        c.bacc[0] = nxv; c.bacc[1] = nyv;
        jx = nxv - ox; jy = nyv - oy;
        c.bvx += jx * c.bminv; c.bvy += jy * c.bminv; c.bw += c.biinv * (R(0) * jy - R(0) * jx);
    }
    // GearJoint(static, block, 0, 1)
    {
        R wr = c.bw * R(1) - R(0);
        R jj = (c.bbias[2] - wr) * c.bgear;
        R jold = c.bacc[2];
        R jn = r_clamp(jold + jj, -c.blim[1], c.blim[1]);
        c.bacc[2] = jn; jj = jn - jold;
        c.bw = c.bw + jj * c.biinv;
    }
}

// cpArbiterApplyCachedImpulse for every warm contact (lane 0, LDS velocities)
template <typename R, typename P> MGX_HD void contacts_warm_start(Env<R, P> &e) {
    int nk = E_I(misc, M_NK);
    for (int k = 0; k < nk; k++) {
        if (E_I(kfirst, k)) continue;
        int ab = E_I(kab, k), a = ab & 0xFF, b = ab >> 8;
        R nx = E_R(knx, k), ny = E_R(kny, k), jn = E_R(kjn, k), jt = E_R(kjt, k);
        R jx = nx * jn - ny * jt, jy = nx * jt + ny * jn;
        R r1x = E_R(kr1x, k), r1y = E_R(kr1y, k), r2x = E_R(kr2x, k), r2y = E_R(kr2y, k);
        E_R(vx, a) -= jx * T_R(body_minv, a); E_R(vy, a) -= jy * T_R(body_minv, a); E_R(w, a) -= T_R(body_iinv, a) * (r1x * jy - r1y * jx);
        E_R(vx, b) += jx * T_R(body_minv, b); E_R(vy, b) += jy * T_R(body_minv, b); E_R(w, b) += T_R(body_iinv, b) * (r2x * jy - r2y * jx);
    }
}

// solve step A (after the preStep phase): lane 0 ages the contact cache, loads the robot island into registers and
// applies the spring torques (cpDampedRotarySpring preStep, in joint order); block lanes load their island
template <typename R, typename P> MGX_HD void solve_begin(Env<R, P> &e, SolveCtx<R> &c, int lane, int nl) {
    const TmplHeader &h = *e.h;
    c.has_contacts = E_I(misc, M_NK) > 0;
    c.bbody = -1; c.bj = -1;
    if (lane == 0) {
        int narb = E_I(misc, M_NARB), ncache = E_I(misc, M_NCACHE);
        // untouched cached arbiters age; they survive collision_persistence = 3 steps (cpSpaceArbiterSetFilter)
        int n = narb;
        for (int q = 0; q < ncache; q++) {
            if (E_I(cmatched, q)) { E_I(cmatched, q) = 0; continue; }
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
        ri_load_vel(e, c);
        int j0 = h.robot_j0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int s = 0; s < RI_BODIES; s++) { int b = ri_body(e, s); c.minv[s] = T_R(body_minv, b); c.iinv[s] = T_R(body_iinv, b); }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int j = 0; j < RI_JOINTS; j++) {
            const int kind = RI_KIND(j), jj = j0 + j;
            const R *p = &T_R(joint_p, jj * JOINT_PARAMS);
            R *f = c.f[j];
            c.lim[j] = E_R(jlim, jj);
            if (kind == J_PIVOT || kind == J_PIN) {
                f[0] = E_R(jr1x, jj); f[1] = E_R(jr1y, jj); f[2] = E_R(jr2x, jj); f[3] = E_R(jr2y, jj);
                if (kind == J_PIVOT) {
                    f[4] = E_R(jk0, jj); f[5] = E_R(jk1, jj); f[6] = E_R(jk2, jj); f[7] = E_R(jk3, jj);
                    f[8] = E_R(jb0, jj); f[9] = E_R(jb1, jj); f[10] = E_R(ja0, jj); f[11] = E_R(ja1, jj);
                } else {
                    f[4] = E_R(jk0, jj); f[5] = E_R(jk1, jj); f[6] = E_R(jk2, jj); f[7] = E_R(jb0, jj); f[8] = E_R(ja0, jj);
                }
            } else if (kind == J_GEAR) {
                f[0] = p[0]; f[1] = E_R(jb0, jj); f[2] = E_R(ja0, jj); f[3] = p[5]; f[4] = R(1) / p[5];      // 1 / ratio once per substep, not per iteration
            } else if (kind == J_SPRING) {
                f[0] = p[0]; f[1] = p[6]; f[2] = R(0);
                // spring torque, applied right here exactly as cpDampedRotarySpring's preStep does
                int a = T_I(joint_a, jj), b = T_I(joint_b, jj);
                R j_spring = R((E_P(ang, a) - E_P(ang, b)) - T_P(p_joint, jj * 7 + 4)) * p[5];
                c.w[RI_SA(j)] -= j_spring * c.iinv[RI_SA(j)]; c.w[RI_SB(j)] += j_spring * c.iinv[RI_SB(j)];
            } else if (kind == J_LIMIT) {
                f[0] = p[0]; f[1] = E_R(jb0, jj); f[2] = E_R(ja0, jj);
            } else {
                f[0] = p[0]; f[1] = E_R(jrate, jj); f[2] = E_R(ja0, jj);
            }
        }
        ri_store_vel(e, c);       // the contact warm start (next step) must see the spring impulses
    } else {
        int isl = lane_island(e, lane, nl);
        if (isl >= 0) {
            int jp = T_I(island_j, isl), jg = jp + 1, b = T_I(joint_b, jp);
            c.bbody = b; c.bj = jp;
            c.bminv = T_R(body_minv, b); c.biinv = T_R(body_iinv, b);
            c.bk[0] = E_R(jk0, jp); c.bk[1] = E_R(jk1, jp); c.bk[2] = E_R(jk2, jp); c.bk[3] = E_R(jk3, jp);
            c.bbias[0] = E_R(jb0, jp); c.bbias[1] = E_R(jb1, jp); c.bbias[2] = E_R(jb0, jg);
            c.bacc[0] = E_R(ja0, jp); c.bacc[1] = E_R(ja1, jp); c.bacc[2] = E_R(ja0, jg);
            c.blim[0] = E_R(jlim, jp); c.blim[1] = E_R(jlim, jg);
            c.bgear = T_R(joint_p, jg * JOINT_PARAMS);
        }
    }
}
// solve step B: cached arbiter impulses (lane 0, LDS)
template <typename R, typename P> MGX_HD void solve_warm_contacts(Env<R, P> &e, int lane) {
    if (lane == 0) contacts_warm_start(e);
}
// solve step C: islands pick up the velocities and apply their cached joint impulses
template <typename R, typename P> MGX_HD void solve_warm_joints(Env<R, P> &e, SolveCtx<R> &c, int lane, int nl) {
    if (lane == 0) {
        ri_load_vel(e, c);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int j = 0; j < RI_JOINTS; j++)
            reg_apply_cached<R>(RI_KIND(j), c.f[j], c.minv[RI_SA(j)], c.iinv[RI_SA(j)], c.minv[RI_SB(j)], c.iinv[RI_SB(j)],
                                c.vx[RI_SA(j)], c.vy[RI_SA(j)], c.w[RI_SA(j)], c.vx[RI_SB(j)], c.vy[RI_SB(j)], c.w[RI_SB(j)]);
    }
    if (c.bbody >= 0) {
        bi_load_vel(e, c);
        // pivot then gear cached impulses (a = static body)
        c.bvx += c.bacc[0] * c.bminv; c.bvy += c.bacc[1] * c.bminv; c.bw += c.biinv * (R(0) * c.bacc[1] - R(0) * c.bacc[0]);
        c.bw += c.bacc[2] * c.biinv;
    } else if (lane != 0 || e.h->n_islands > nl - 1) {
        // islands without a lane of their own: through LDS (lane 0 takes them all when L is small)
        if (e.h->n_islands > nl - 1 && lane == 0)
            for (int k = 0; k < e.h->n_islands; k++) { int jp = T_I(island_j, k); joint_apply_cached(e, jp); joint_apply_cached(e, jp + 1); }
    }
}
// solve step D1 / D2 / D3: one Gauss-Seidel iteration = [publish island velocities] [contacts] [islands]
template <typename R, typename P> MGX_HD void solve_iter_publish(Env<R, P> &e, SolveCtx<R> &c, int lane) {
    if (!c.has_contacts) return;
    if (lane == 0) ri_store_vel(e, c);
    bi_store_vel(e, c);
}
template <typename R, typename P> MGX_HD void solve_iter_contacts(Env<R, P> &e, SolveCtx<R> &c, int lane) {
    if (!c.has_contacts || lane != 0) return;
    int nk = E_I(misc, M_NK);
    if (nk == 0) return;
    ContactK<R> cur = contact_load(e, 0);
    for (int k = 0; k < nk; k++) {
        // the next contact's constants are in flight while this one's dependent chain runs
        const ContactK<R> nxt = contact_load(e, k + 1 < nk ? k + 1 : k);
        contact_apply_loaded(e, k, cur);
        cur = nxt;
    }
}
template <typename R, typename P> MGX_HD void solve_iter_joints(Env<R, P> &e, SolveCtx<R> &c, int lane, int nl) {
    if (c.has_contacts) { if (lane == 0) ri_load_vel(e, c); bi_load_vel(e, c); }
    if (lane == 0) {
        if (e.h->n_islands > nl - 1) {
            // not enough lanes for the block islands: lane 0 runs them through LDS, in joint order relative to the robot
            if (!c.has_contacts) { /* LDS copy of block velocities is current: blocks never enter the robot's registers */ }
            for (int k = 0; k < e.h->n_islands; k++) { int jp = T_I(island_j, k); joint_apply_impulse(e, jp); joint_apply_impulse(e, jp + 1); }
        }
#define MGX_RI_APPLY(j, ZA, ZB, KA) reg_apply_joint<R, ZA, ZB, KA>(RI_KIND(j), c.f[j], c.lim[j], c.minv[RI_SA(j)], c.iinv[RI_SA(j)], \
            c.minv[RI_SB(j)], c.iinv[RI_SB(j)], c.vx[RI_SA(j)], c.vy[RI_SA(j)], c.w[RI_SA(j)], c.vx[RI_SB(j)], c.vy[RI_SB(j)], c.w[RI_SB(j)])
        // Robot.setup order: pivot + gear from the kinematic control body (anchors at both origins), two eye springs, then per
        // finger {pin (anchored at the finger's origin), limit, motor}
        MGX_RI_APPLY(0, true, true, true); MGX_RI_APPLY(1, false, false, true); MGX_RI_APPLY(2, false, false, false); MGX_RI_APPLY(3, false, false, false);
        MGX_RI_APPLY(4, false, true, false); MGX_RI_APPLY(5, false, false, false); MGX_RI_APPLY(6, false, false, false);
        MGX_RI_APPLY(7, false, true, false); MGX_RI_APPLY(8, false, false, false); MGX_RI_APPLY(9, false, false, false);
#undef MGX_RI_APPLY
    }
    bi_iterate(c);
}
// solve step E: write velocities and accumulators back, then next substep's Robot.update
template <typename R, typename P> MGX_HD void solve_end(Env<R, P> &e, SolveCtx<R> &c, int lane) {
    if (lane == 0) {
        ri_store_vel(e, c);
        int j0 = e.h->robot_j0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int j = 0; j < RI_JOINTS; j++) {
            const int kind = RI_KIND(j), jj = j0 + j;
            if (kind == J_PIVOT) { E_R(ja0, jj) = c.f[j][10]; E_R(ja1, jj) = c.f[j][11]; }
            else if (kind == J_PIN) E_R(ja0, jj) = c.f[j][8];
            else if (kind != J_SPRING) E_R(ja0, jj) = c.f[j][2];
        }
        ph_control(e);
    }
    if (c.bbody >= 0) {
        bi_store_vel(e, c);
        E_R(ja0, c.bj) = c.bacc[0]; E_R(ja1, c.bj) = c.bacc[1]; E_R(ja0, c.bj + 1) = c.bacc[2];
    }
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
    for (int j = lane; j < e.h->n_joints; j += nl) { E_R(ja0, j) = R(0); E_R(ja1, j) = R(0); E_R(jrate, j) = R(0); E_R(jb0, j) = R(0); E_R(jb1, j) = R(0); }
    for (int c = lane; c < e.h->cache_slots; c += nl) E_I(cmatched, c) = 0;
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
template <typename R, typename P>
MGX_HD void ph_store_state(Env<R, P> &e, P *sp, R *sf, int32_t *si, long stride, long env, int lane, int nl) {
    const TmplHeader &h = *e.h;
    int nvel = state_row_jacc0(h);      // first row after the velocities and force limits
    for (int k = lane; k < h.n_state; k += nl) {
        int m = T_I(state_map, k), comp = m & 15, b = (m >> 4) & 0xFF, row = m >> 12;
        if (comp < 3) {
            sp[(long)row * stride + env] = comp == 0 ? E_P(px, b) : (comp == 1 ? E_P(py, b) : E_P(ang, b));
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
#define MGX_SUBSTEP_PHASES(X)                                      \
    X(ph_integrate(e, lane, nl))                                   \
    X(ph_shapes(e, lane, nl))                                      \
    X(ph_broad_count(e, lane, nl))                                 \
    X(ph_broad_write(e, lane, nl))                                 \
    X(ph_narrow(e, lane, nl))                                      \
    X(ph_arbiters_joints(e, lane, nl))                             \
    X(solve_begin(e, ctx, lane, nl))                               \
    X(solve_warm_contacts(e, lane))                                \
    X(solve_warm_joints(e, ctx, lane, nl))                         \
    MGX_SOLVE_ITERATIONS(X)                                        \
    X(solve_end(e, ctx, lane))                                     \
    X(ph_cache_commit(e, lane, nl))

// `iterations` Gauss-Seidel sweeps; each is three lane-group-synchronised steps (ctx = this lane's SolveCtx)
#define MGX_SOLVE_ITERATIONS(X)                                    \
    for (int it_ = 0; it_ < iterations; it_++) {                   \
        X(solve_iter_publish(e, ctx, lane))                        \
        X(solve_iter_contacts(e, ctx, lane))                       \
        X(solve_iter_joints(e, ctx, lane, nl))                     \
    }


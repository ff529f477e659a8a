// mgx_raster.h -- egocentric / allocentric rasteriser core (host+device compilable).
//
// Replaces, for one env:  BaseEnv.render('rgb_array') -> Viewer.render (gym_render.py:208-249:
// clear to the background, painter's-order fill of convex polygons, GL_LINE_SMOOTH loops)
// at 384x384, then cv2.resize(INTER_AREA) to 96x96 (benchmarks/__init__.py:234) -- fused:
// each 96x96 output pixel is the cvRound'ed mean of its 4x4 block of 384-grid point samples.
//
// Exactness strategy: all geometry is set up and tested in fp64 (MI355X runs fp64 vector math
// at half the fp32 rate, and this kernel is nowhere near ALU-bound), so the u8 output can be
// compared bit-for-bit with the fp64 oracle.  Speed comes from classification, not from lower
// precision: per output pixel every primitive is first classified ALL / NONE / MIXED against
// the 4x4 sample block with a conservative bound; pixels without a MIXED primitive (the large
// majority) take the colour of the topmost ALL primitive without touching a single sample.
#pragma once
#include "mgx_sim.h"

namespace mgx {

constexpr int NATIVE_RES = 384;   // benchmarks/__init__.py:23 DEFAULT_RES
constexpr int LORES = 96;         // LoRes* preprocessors
constexpr double BLOCK_HALF_DIAG = 2.1213203435596424 + 1e-9;   // 1.5*sqrt(2): 4x4 sample block half diagonal
constexpr double CLASS_EPS = 1e-9;

// per-env raster scratch (LDS on device)
struct RasterOff {
    int bx, by, ba, bc, bs;             // per body pose (doubles)
    int svx, svy, ea, eb, ec;           // per prim vertex: screen position + edge function of edge (i -> i+1)
    int pcx, pcy, prad, papo, pphi;     // per prim (n-gon centre/radius/apothem/phase; line half width in prad)
    int n_d;
    int bb;                             // per prim bbox in 384-grid units: x0 y0 x1 y1 (ints, inclusive, may be empty)
    int n_i;
    MGX_HD explicit RasterOff(const TmplHeader &h) {
        int o = 0;
        bx = o; o += h.n_bodies; by = o; o += h.n_bodies; ba = o; o += h.n_bodies; bc = o; o += h.n_bodies; bs = o; o += h.n_bodies;
        svx = o; o += h.n_pverts; svy = o; o += h.n_pverts; ea = o; o += h.n_pverts; eb = o; o += h.n_pverts; ec = o; o += h.n_pverts;
        pcx = o; o += h.n_prims; pcy = o; o += h.n_prims; prad = o; o += h.n_prims; papo = o; o += h.n_prims; pphi = o; o += h.n_prims;
        n_d = o;
        o = 0;
        bb = o; o += 4 * h.n_prims;
        n_i = o;
    }
};

struct Raster {
    const TmplHeader *h;
    const int32_t *ti;
    const double *tq;     // prim reals + prim verts in fp64 (template copy kept in double for the rasteriser)
    double *d;
    int32_t *i;
    TmplOff to;
    RasterOff ro;
    int view;
    MGX_HD Raster(const TmplHeader *h_, const int32_t *ti_, const double *tq_, double *d_, int32_t *i_, int view_)
        : h(h_), ti(ti_), tq(tq_), d(d_), i(i_), to(*h_), ro(*h_), view(view_) {}
    MGX_HD int prim_kind(int k) const { return ti[to.prim_i + k * PRIM_IWORDS]; }
    MGX_HD int prim_nv(int k) const { return ti[to.prim_i + k * PRIM_IWORDS + 1]; }
    MGX_HD int prim_voff(int k) const { return ti[to.prim_i + k * PRIM_IWORDS + 2]; }
    MGX_HD int prim_xf(int k) const { return ti[to.prim_i + k * PRIM_IWORDS + 3]; }
    MGX_HD int prim_rgb(int k) const { return ti[to.prim_i + k * PRIM_IWORDS + 4]; }
    MGX_HD int prim_stipple(int k) const { return ti[to.prim_i + k * PRIM_IWORDS + 5]; }
    // tq layout: [n_prims * PRIM_RWORDS][pvx n_pverts][pvy n_pverts]
    MGX_HD double prim_r(int k, int j) const { return tq[k * PRIM_RWORDS + j]; }
    MGX_HD double pvx(int v) const { return tq[h->n_prims * PRIM_RWORDS + v]; }
    MGX_HD double pvy(int v) const { return tq[h->n_prims * PRIM_RWORDS + h->n_pverts + v]; }
};

#define RD(field, k) rs.d[rs.ro.field + (k)]
#define RI(field, k) rs.i[rs.ro.field + (k)]

MGX_HD double rz_floor(double x) { return floor(x); }

// ---- setup phase 1: body poses from the pose blob (lane per body)
template <typename P>
MGX_HD void raster_setup_bodies(Raster &rs, const P *sp, long stride, long env, int lane, int nl) {
    const TmplHeader &h = *rs.h;
    for (int b = lane; b < h.n_bodies; b += nl) {
        double v[3] = {0.0, 0.0, 0.0};
        for (int c = 0; c < 3; c++) {
            int row = rs.ti[rs.to.body_prow + 3 * b + c];
            if (row >= 0) v[c] = (double)sp[(long)row * stride + env];
        }
        double s, c;
        r_sincos<double>(v[2], s, c);
        RD(bx, b) = v[0]; RD(by, b) = v[1]; RD(ba, b) = v[2]; RD(bc, b) = c; RD(bs, b) = s;
    }
}

// world -> screen affine for this env (base_env.py:294-307, gym_render.py:176-200,372-377)
MGX_HD void raster_camera(const Raster &rs, double *cam) {
    const double zoom = 1.02, arena = 2.0;                       // style.py ARENA_ZOOM_OUT, base_env.py:65
    const double world = arena * zoom, sc = (double)NATIVE_RES / world;
    if (rs.view == 1) {
        cam[0] = sc; cam[1] = 0; cam[2] = 0; cam[3] = sc; cam[4] = zoom * sc; cam[5] = zoom * sc;
    } else {
        int rb = rs.h->robot_body;
        double rx = RD(bx, rb), ry = RD(by, rb);
        double c = RD(bc, rb), s = -RD(bs, rb);                  // rotation by -theta
        double npx = world * 0.5, npy = world * 0.15;
        cam[0] = sc * c; cam[1] = -sc * s; cam[2] = sc * s; cam[3] = sc * c;
        cam[4] = sc * (npx + (c * -rx - s * -ry));
        cam[5] = sc * (npy + (s * -rx + c * -ry));
    }
}

// ---- setup phase 2: screen-space vertices (lane per vertex) and n-gon records (lane per prim)
MGX_HD void raster_setup_prims(Raster &rs, int lane, int nl) {
    const TmplHeader &h = *rs.h;
    double cam[6];
    raster_camera(rs, cam);
    for (int k = lane; k < h.n_prims; k += nl) {
        int kind = rs.prim_kind(k), xfw = rs.prim_xf(k);
        int xf = xfw & 0xFF, body = (xfw >> 8) & 0xFF, eye_body = ((xfw >> 16) & 0xFF) - 1;
        int nv = rs.prim_nv(k), vo = rs.prim_voff(k);
        double bx = RD(bx, body), by = RD(by, body), bc = RD(bc, body), bs = RD(bs, body);
        double ebx = rs.prim_r(k, 0), eby = rs.prim_r(k, 1), epx = rs.prim_r(k, 2), epy = rs.prim_r(k, 3);
        double da = 0.0, dc = 1.0, ds = 0.0;
        if (xf == XF_EYE && eye_body >= 0) {
            da = RD(ba, eye_body) - RD(ba, body);
            r_sincos<double>(da, ds, dc);
        }
        if (kind == PR_NGON) {
            // centre = image of the local origin; phase = world angle of vertex 0 minus camera rotation
            double lx = 0.0, ly = 0.0;
            if (xf == XF_EYE) { double qx = lx + epx, qy = ly + epy; lx = (dc * qx - ds * qy) + ebx; ly = (ds * qx + dc * qy) + eby; }
            double wx = lx, wy = ly;
            if (xf != XF_WORLD) { wx = bx + (bc * lx - bs * ly); wy = by + (bc * ly + bs * lx); }
            RD(pcx, k) = cam[0] * wx + cam[1] * wy + cam[4];
            RD(pcy, k) = cam[2] * wx + cam[3] * wy + cam[5];
            double sc = (double)NATIVE_RES / 2.04;
            double rad = rs.prim_r(k, 5) * sc;
            RD(prad, k) = rad;
            RD(papo, k) = rad * cos(3.14159265358979323846 / nv);
            double phi = 0.0;
            if (xf != XF_WORLD) phi += RD(ba, body);
            if (xf == XF_EYE) phi += da;
            if (rs.view == 0) phi -= RD(ba, rs.h->robot_body);
            RD(pphi, k) = phi;
            int x0 = (int)rz_floor(RD(pcx, k) - rad - 0.5), x1 = (int)ceil(RD(pcx, k) + rad - 0.5);
            int y0 = (int)rz_floor(RD(pcy, k) - rad - 0.5), y1 = (int)ceil(RD(pcy, k) + rad - 0.5);
            RI(bb, 4 * k) = x0; RI(bb, 4 * k + 1) = y0; RI(bb, 4 * k + 2) = x1; RI(bb, 4 * k + 3) = y1;
            continue;
        }
        double minx = 1e30, maxx = -1e30, miny = 1e30, maxy = -1e30;
        for (int i = 0; i < nv; i++) {
            double lx = rs.pvx(vo + i), ly = rs.pvy(vo + i);
            double wx = lx, wy = ly;
            if (xf != XF_WORLD) { wx = bx + (bc * lx - bs * ly); wy = by + (bc * ly + bs * lx); }
            double sx = cam[0] * wx + cam[1] * wy + cam[4], sy = cam[2] * wx + cam[3] * wy + cam[5];
            RD(svx, vo + i) = sx; RD(svy, vo + i) = sy;
            minx = r_min(minx, sx); maxx = r_max(maxx, sx); miny = r_min(miny, sy); maxy = r_max(maxy, sy);
        }
        double pad = 0.0;
        if (kind == PR_LINELOOP) { RD(prad, k) = rs.prim_r(k, 4); pad = rs.prim_r(k, 4) + 1.0; }
        RI(bb, 4 * k) = (int)rz_floor(minx - pad - 0.5); RI(bb, 4 * k + 1) = (int)rz_floor(miny - pad - 0.5);
        RI(bb, 4 * k + 2) = (int)ceil(maxx + pad - 0.5); RI(bb, 4 * k + 3) = (int)ceil(maxy + pad - 0.5);
        if (kind == PR_POLY) {
            double area2 = 0.0;
            for (int i = 0; i < nv; i++) {
                int j = (i + 1) % nv;
                area2 += RD(svx, vo + i) * RD(svy, vo + j) - RD(svy, vo + i) * RD(svx, vo + j);
            }
            double sgn = area2 >= 0.0 ? 1.0 : -1.0;
            for (int i = 0; i < nv; i++) {
                int j = (i + 1) % nv;
                double ax = RD(svx, vo + i), ay = RD(svy, vo + i), ex = RD(svx, vo + j) - ax, ey = RD(svy, vo + j) - ay;
                double inv = sgn / sqrt(ex * ex + ey * ey);
                // E(p) = sgn * cross(e, p - a) / |e|   (>= 0 inside)
                RD(ea, vo + i) = -ey * inv; RD(eb, vo + i) = ex * inv; RD(ec, vo + i) = (ey * ax - ex * ay) * inv;
            }
        }
    }
}

// ---- exact sample tests (fp64)
MGX_HD bool poly_contains(const Raster &rs, int k, double x, double y) {
    int nv = rs.prim_nv(k), vo = rs.prim_voff(k);
    for (int i = 0; i < nv; i++)
        if (RD(ea, vo + i) * x + RD(eb, vo + i) * y + RD(ec, vo + i) < 0.0) return false;
    return true;
}
MGX_HD bool ngon_contains(const Raster &rs, int k, double x, double y) {
    double qx = x - RD(pcx, k), qy = y - RD(pcy, k);
    double d2 = qx * qx + qy * qy, apo = RD(papo, k), rad = RD(prad, k);
    if (d2 <= (apo - CLASS_EPS) * (apo - CLASS_EPS)) return true;
    if (d2 > (rad + CLASS_EPS) * (rad + CLASS_EPS)) return false;
    // thin annulus: test against the edge of the sector the point falls in
    int n = rs.prim_nv(k);
    double step = 6.283185307179586476925 / n;
    double th = atan2(qy, qx) - RD(pphi, k);
    double kk = rz_floor(th / step);
    double mid = RD(pphi, k) + (kk + 0.5) * step, s, c;
    r_sincos<double>(mid, s, c);
    return qx * c + qy * s <= apo;
}
// max coverage alpha of a smooth line loop at a sample; OUR model of GL_LINE_SMOOTH (driver-defined):
// alpha = clamp(halfwidth - dist, 0, 1), halfwidth = (w + 1) / 2, 16-px stipple by arclength.
MGX_HD double lineloop_alpha(const Raster &rs, int k, double x, double y) {
    int nv = rs.prim_nv(k), vo = rs.prim_voff(k), stipple = rs.prim_stipple(k);
    double hw = RD(prad, k), best = 0.0, arc = 0.0;
    for (int i = 0; i < nv; i++) {
        int j = (i + 1) % nv;
        double ax = RD(svx, vo + i), ay = RD(svy, vo + i), dx = RD(svx, vo + j) - ax, dy = RD(svy, vo + j) - ay;
        double l2 = dx * dx + dy * dy;
        double t = l2 > 0.0 ? r_clamp01(((x - ax) * dx + (y - ay) * dy) / l2) : 0.0;
        double ex = x - (ax + dx * t), ey = y - (ay + dy * t);
        double dist = sqrt(ex * ex + ey * ey);
        double alpha = r_clamp01(hw - dist);
        double len = sqrt(l2);
        if (alpha > 0.0 && stipple) {
            int bit = ((int)rz_floor(arc + t * len)) & 15;
            if (!((stipple >> bit) & 1)) alpha = 0.0;
        }
        if (alpha > best) best = alpha;
        arc += len;
    }
    return best;
}

// one 384-grid sample, painter's order over the primitives in `mask` (bit k = prim k), starting from `base_rgb`
MGX_HD int raster_sample(const Raster &rs, double x, double y, uint64_t mask, int base_rgb) {
    int r = base_rgb & 0xFF, g = (base_rgb >> 8) & 0xFF, b = (base_rgb >> 16) & 0xFF;
    while (mask) {
        int k = __builtin_ctzll(mask);
        mask &= mask - 1;
        int kind = rs.prim_kind(k), col = rs.prim_rgb(k);
        if (kind == PR_LINELOOP) {
            double a = lineloop_alpha(rs, k, x, y);
            if (a > 0.0) {
                r = (int)rz_floor(a * (double)(col & 0xFF) + (1.0 - a) * (double)r + 0.5);
                g = (int)rz_floor(a * (double)((col >> 8) & 0xFF) + (1.0 - a) * (double)g + 0.5);
                b = (int)rz_floor(a * (double)((col >> 16) & 0xFF) + (1.0 - a) * (double)b + 0.5);
            }
        } else {
            bool in = kind == PR_POLY ? poly_contains(rs, k, x, y) : ngon_contains(rs, k, x, y);
            if (in) { r = col & 0xFF; g = (col >> 8) & 0xFF; b = (col >> 16) & 0xFF; }
        }
    }
    return r | (g << 8) | (b << 16);
}

enum { CLS_NONE = 0, CLS_ALL = 1, CLS_MIXED = 2 };
// classify prim k against the 4x4 sample block centred at (xc, yc)
MGX_HD int classify_block(const Raster &rs, int k, double xc, double yc) {
    int kind = rs.prim_kind(k);
    if (kind == PR_POLY) {
        int nv = rs.prim_nv(k), vo = rs.prim_voff(k);
        bool all = true;
        for (int i = 0; i < nv; i++) {
            double a = RD(ea, vo + i), b = RD(eb, vo + i);
            double ec = a * xc + b * yc + RD(ec, vo + i);
            double ext = 1.5 * (r_abs(a) + r_abs(b)) + CLASS_EPS;
            if (ec + ext < 0.0) return CLS_NONE;
            if (ec - ext < 0.0) all = false;
        }
        return all ? CLS_ALL : CLS_MIXED;
    } else if (kind == PR_NGON) {
        double qx = xc - RD(pcx, k), qy = yc - RD(pcy, k);
        double dc = sqrt(qx * qx + qy * qy);
        if (dc + BLOCK_HALF_DIAG < RD(papo, k) - CLASS_EPS) return CLS_ALL;
        if (dc - BLOCK_HALF_DIAG > RD(prad, k) + CLASS_EPS) return CLS_NONE;
        return CLS_MIXED;
    } else {
        int nv = rs.prim_nv(k), vo = rs.prim_voff(k);
        double hw = RD(prad, k);
        for (int i = 0; i < nv; i++) {
            int j = (i + 1) % nv;
            double ax = RD(svx, vo + i), ay = RD(svy, vo + i), dx = RD(svx, vo + j) - ax, dy = RD(svy, vo + j) - ay;
            double l2 = dx * dx + dy * dy;
            double t = l2 > 0.0 ? r_clamp01(((xc - ax) * dx + (yc - ay) * dy) / l2) : 0.0;
            double ex = xc - (ax + dx * t), ey = yc - (ay + dy * t);
            if (sqrt(ex * ex + ey * ey) - BLOCK_HALF_DIAG <= hw + CLASS_EPS) return CLS_MIXED;
        }
        return CLS_NONE;
    }
}

// one 96x96 output pixel (X, Y), Y = 0 at the top; `tile_mask`: prims whose bbox touches the tile
MGX_HD int raster_pixel_lores(const Raster &rs, int X, int Y, uint64_t tile_mask, int bg_rgb) {
    const double xc = 4.0 * X + 2.0, yc = (double)NATIVE_RES - 4.0 * Y - 2.0;
    const int gx0 = 4 * X, gx1 = 4 * X + 3, gy1 = NATIVE_RES - 1 - 4 * Y, gy0 = gy1 - 3;   // 384-grid index range of the block
    int base = bg_rgb;
    uint64_t mixed = 0;
    // front to back: stop at the topmost primitive that covers the whole block
    uint64_t m = tile_mask;
    while (m) {
        int k = 63 - __builtin_clzll(m);
        m &= ~(1ull << k);
        if (RI(bb, 4 * k) > gx1 || RI(bb, 4 * k + 2) < gx0 || RI(bb, 4 * k + 1) > gy1 || RI(bb, 4 * k + 3) < gy0) continue;
        int cls = classify_block(rs, k, xc, yc);
        if (cls == CLS_ALL) { base = rs.prim_rgb(k); break; }
        if (cls == CLS_MIXED) mixed |= 1ull << k;
    }
    if (!mixed) return base;
    int sr = 0, sg = 0, sb = 0;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) {
            int c = raster_sample(rs, 4.0 * X + i + 0.5, (double)NATIVE_RES - 0.5 - 4.0 * Y - j, mixed, base);
            sr += c & 0xFF; sg += (c >> 8) & 0xFF; sb += (c >> 16) & 0xFF;
        }
    // cv2 INTER_AREA integer-factor path: saturate_cast<uchar>(sum * (1/16)) = round half to even
    int r = (sr + 7 + ((sr >> 4) & 1)) >> 4, g = (sg + 7 + ((sg >> 4) & 1)) >> 4, b = (sb + 7 + ((sb >> 4) & 1)) >> 4;
    return r | (g << 8) | (b << 16);
}

}  // namespace mgx

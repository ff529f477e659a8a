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
constexpr double CLASS_EPS = 1e-9;

// per-env raster scratch (LDS on device)
struct RasterOff {
    int bx, by, ba, bc, bs;             // per body pose (doubles)
    int svx, svy, ea, eb, ec;           // per prim vertex: screen position + normalised edge function of edge (i -> i+1)
    int elen, earc;                     // line loops: segment length and arclength at the segment start
    int pcx, pcy, prad, papo, pphi;     // per prim (n-gon centre/radius/apothem/phase; line half width in prad)
    int n_d;
    int bb;                             // per prim bbox in 384-grid units: x0 y0 x1 y1 (ints, inclusive, may be empty)
    int n_i;
    MGX_HD explicit RasterOff(const TmplHeader &h) {
        int o = 0;
        bx = o; o += h.n_bodies; by = o; o += h.n_bodies; ba = o; o += h.n_bodies; bc = o; o += h.n_bodies; bs = o; o += h.n_bodies;
        svx = o; o += h.n_pverts; svy = o; o += h.n_pverts; ea = o; o += h.n_pverts; eb = o; o += h.n_pverts; ec = o; o += h.n_pverts;
        elen = o; o += h.n_pverts; earc = o; o += h.n_pverts;
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
        {
            // normalised edge functions E(p) = sgn * cross(e, p - a) / |e|: >= 0 inside a polygon; for a line
            // loop |E| is the distance to the segment's carrier line and (eb, -ea) is its unit direction
            double sgn = 1.0;
            if (kind == PR_POLY) {
                double area2 = 0.0;
                for (int i = 0; i < nv; i++) {
                    int j = (i + 1) % nv;
                    area2 += RD(svx, vo + i) * RD(svy, vo + j) - RD(svy, vo + i) * RD(svx, vo + j);
                }
                sgn = area2 >= 0.0 ? 1.0 : -1.0;
            }
            double arc = 0.0;
            for (int i = 0; i < nv; i++) {
                int j = (i + 1) % nv;
                double ax = RD(svx, vo + i), ay = RD(svy, vo + i), ex = RD(svx, vo + j) - ax, ey = RD(svy, vo + j) - ay;
                double len = sqrt(ex * ex + ey * ey);
                double inv = sgn / len;
                RD(ea, vo + i) = -ey * inv; RD(eb, vo + i) = ex * inv; RD(ec, vo + i) = (ey * ax - ex * ay) * inv;
                RD(elen, vo + i) = len; RD(earc, vo + i) = arc;
                arc += len;
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
    double hw = RD(prad, k), best = 0.0;
    for (int i = 0; i < nv; i++) {
        double a = RD(ea, vo + i), b = RD(eb, vo + i);
        double e = a * x + b * y + RD(ec, vo + i);
        if (r_abs(e) >= hw) continue;                       // distance to the segment >= distance to its line
        double ax = RD(svx, vo + i), ay = RD(svy, vo + i), len = RD(elen, vo + i);
        double sl = r_clamp((x - ax) * b - (y - ay) * a, 0.0, len);   // arclength of the closest point
        double qx = x - (ax + b * sl), qy = y - (ay - a * sl);
        double alpha = r_clamp01(hw - sqrt(qx * qx + qy * qy));
        if (alpha > 0.0 && stipple) {
            int bit = ((int)rz_floor(RD(earc, vo + i) + sl)) & 15;
            if (!((stipple >> bit) & 1)) alpha = 0.0;
        }
        if (alpha > best) best = alpha;
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
// Classify prim k against the block of 384-grid samples centred at (xc, yc) with half extents (hx, hy)
// (sample centres, so a 4x4 block has hx = hy = 1.5): ALL = every sample inside an opaque prim, NONE = no
// sample touched, MIXED = decide per sample.  Division- and sqrt-free; conservative by CLASS_EPS.
MGX_HD int classify_rect(const Raster &rs, int k, double xc, double yc, double hx, double hy) {
    int kind = rs.prim_kind(k);
    if (kind == PR_POLY) {
        int nv = rs.prim_nv(k), vo = rs.prim_voff(k);
        bool all = true;
        for (int i = 0; i < nv; i++) {
            double a = RD(ea, vo + i), b = RD(eb, vo + i);
            double e = a * xc + b * yc + RD(ec, vo + i);
            double ext = hx * r_abs(a) + hy * r_abs(b) + CLASS_EPS;
            if (e + ext < 0.0) return CLS_NONE;
            if (e - ext < 0.0) all = false;
        }
        return all ? CLS_ALL : CLS_MIXED;
    } else if (kind == PR_NGON) {
        double qx = r_abs(xc - RD(pcx, k)), qy = r_abs(yc - RD(pcy, k));
        double nx = r_max(qx - hx, 0.0), ny = r_max(qy - hy, 0.0);          // nearest point of the rect
        double fx = qx + hx, fy = qy + hy;                                    // farthest corner
        double apo = RD(papo, k) - CLASS_EPS, rad = RD(prad, k) + CLASS_EPS;
        if (fx * fx + fy * fy < apo * apo) return CLS_ALL;
        if (nx * nx + ny * ny > rad * rad) return CLS_NONE;
        return CLS_MIXED;
    } else {
        int nv = rs.prim_nv(k), vo = rs.prim_voff(k);
        double hw = RD(prad, k) + CLASS_EPS;
        for (int i = 0; i < nv; i++) {
            double a = RD(ea, vo + i), b = RD(eb, vo + i);
            double e = a * xc + b * yc + RD(ec, vo + i);
            if (r_abs(e) - (hx * r_abs(a) + hy * r_abs(b)) > hw) continue;       // off the carrier line
            double sl = (xc - RD(svx, vo + i)) * b - (yc - RD(svy, vo + i)) * a;  // along the segment
            double es = hx * r_abs(b) + hy * r_abs(a);
            if (sl + es < -hw || sl - es > RD(elen, vo + i) + hw) continue;       // beyond its ends
            return CLS_MIXED;
        }
        return CLS_NONE;
    }
}

constexpr int TILE_W = 16, TILE_H = 4, TILES_X = LORES / TILE_W, TILES_Y = LORES / TILE_H;

// tile (tcol, trow) of 16x4 output pixels = 64x16 samples: class of prim k for the whole tile
MGX_HD int classify_tile(const Raster &rs, int k, int tcol, int trow) {
    const int gx0 = 4 * TILE_W * tcol, gx1 = gx0 + 4 * TILE_W - 1;
    const int gy1 = NATIVE_RES - 1 - 4 * TILE_H * trow, gy0 = gy1 - 4 * TILE_H + 1;
    if (RI(bb, 4 * k) > gx1 || RI(bb, 4 * k + 2) < gx0 || RI(bb, 4 * k + 1) > gy1 || RI(bb, 4 * k + 3) < gy0) return CLS_NONE;
    return classify_rect(rs, k, 0.5 * (gx0 + gx1 + 1), 0.5 * (gy0 + gy1 + 1), 2.0 * TILE_W - 0.5, 2.0 * TILE_H - 0.5);
}
// combine the per-prim tile classes (as bit masks) into the tile's base colour and its set of undecided prims
MGX_HD void tile_resolve(const Raster &rs, uint64_t all_mask, uint64_t &mixed_mask, int &base_rgb) {
    if (all_mask) {
        int ka = 63 - __builtin_clzll(all_mask);
        base_rgb = rs.prim_rgb(ka);
        mixed_mask &= ~((2ull << ka) - 1ull);       // everything at or below the topmost covering prim is hidden
    }
}

// 16-bit coverage of the 4x4 sample block whose top-left sample is (x0, y0) (bit 4*j + i = sample (x0 + i, y0 - j))
MGX_HD uint32_t poly_coverage16(const Raster &rs, int k, double x0, double y0) {
    int nv = rs.prim_nv(k), vo = rs.prim_voff(k);
    uint32_t cov = 0xFFFFu;
    for (int e = 0; e < nv && cov; e++) {
        double a = RD(ea, vo + e), b = RD(eb, vo + e);
        double row = a * x0 + b * y0 + RD(ec, vo + e);
        // the block spans x0..x0+3, y0-3..y0: if even its worst corner is inside this edge, nothing to test
        double worst = row + r_min(0.0, 3.0 * a) - r_max(0.0, 3.0 * b);
        if (worst >= CLASS_EPS) continue;
        uint32_t m = 0;
        for (int j = 0; j < 4; j++) {
            double v = row;
            for (int i = 0; i < 4; i++) { m |= (v >= 0.0 ? 1u : 0u) << (4 * j + i); v += a; }
            row -= b;
        }
        cov &= m;
    }
    return cov;
}
MGX_HD uint32_t ngon_coverage16(const Raster &rs, int k, double x0, double y0) {
    double cx = RD(pcx, k), cy = RD(pcy, k);
    double apo = RD(papo, k) - CLASS_EPS, rad = RD(prad, k) + CLASS_EPS, apo2 = apo * apo, rad2 = rad * rad;
    uint32_t cov = 0;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) {
            double qx = x0 + i - cx, qy = y0 - j - cy, d2 = qx * qx + qy * qy;
            bool in = d2 <= apo2;
            if (!in && d2 <= rad2) in = ngon_contains(rs, k, x0 + i, y0 - j);     // thin annulus: exact sector test
            cov |= (in ? 1u : 0u) << (4 * j + i);
        }
    return cov;
}

// max line-loop alpha for the 4 samples (x0 + i, y), i = 0..3: segment parameters are loaded once per row
MGX_HD void lineloop_alpha_row(const Raster &rs, int k, double x0, double y, uint32_t segmask, double *best) {
    int nv = rs.prim_nv(k), vo = rs.prim_voff(k), stipple = rs.prim_stipple(k);
    double hw = RD(prad, k);
    best[0] = best[1] = best[2] = best[3] = 0.0;
    for (int i = 0; i < nv; i++) {
        if (!((segmask >> i) & 1u)) continue;
        double a = RD(ea, vo + i), b = RD(eb, vo + i);
        double e0 = a * x0 + b * y + RD(ec, vo + i);
        // all four samples off this segment's carrier line?  (e is affine in x)
        double e3 = e0 + 3.0 * a;
        if ((e0 >= hw && e3 >= hw) || (e0 <= -hw && e3 <= -hw)) continue;
        double ax = RD(svx, vo + i), ay = RD(svy, vo + i), len = RD(elen, vo + i), arc = RD(earc, vo + i);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int t = 0; t < 4; t++) {
            double x = x0 + t, e = a * x + b * y + RD(ec, vo + i);
            if (r_abs(e) >= hw) continue;
            double sl = r_clamp((x - ax) * b - (y - ay) * a, 0.0, len);
            double qx = x - (ax + b * sl), qy = y - (ay - a * sl);
            double alpha = r_clamp01(hw - sqrt(qx * qx + qy * qy));
            if (alpha > 0.0 && stipple) {
                int bit = ((int)rz_floor(arc + sl)) & 15;
                if (!((stipple >> bit) & 1)) alpha = 0.0;
            }
            if (alpha > best[t]) best[t] = alpha;
        }
    }
}

// segments of line loop k that can touch the 4x4 block whose top-left sample is (x0, y0)
MGX_HD uint32_t lineloop_block_segments(const Raster &rs, int k, double x0, double y0) {
    int nv = rs.prim_nv(k), vo = rs.prim_voff(k);
    double hw = RD(prad, k) + CLASS_EPS, xc = x0 + 1.5, yc = y0 - 1.5;
    uint32_t mask = 0;
    for (int i = 0; i < nv && i < 16; i++) {
        double a = RD(ea, vo + i), b = RD(eb, vo + i);
        double e = a * xc + b * yc + RD(ec, vo + i);
        if (r_abs(e) - 1.5 * (r_abs(a) + r_abs(b)) > hw) continue;
        double sl = (xc - RD(svx, vo + i)) * b - (yc - RD(svy, vo + i)) * a, es = 1.5 * (r_abs(a) + r_abs(b));
        if (sl + es < -hw || sl - es > RD(elen, vo + i) + hw) continue;
        mask |= 1u << i;
    }
    return mask;
}

// Painter's-order resolution of the prims in `lower` (which contains at least one line loop) for the samples in
// `remaining`, accumulated into (sr, sg, sb).  Opaque prims act through coverage masks, line loops blend row by row.
MGX_HD void resolve_lower_stack(const Raster &rs, uint64_t lower, double x0, double y0, int base, uint32_t remaining,
                                int &sr, int &sg, int &sb) {
    constexpr int MAXL = 6;
    int lk[MAXL]; uint32_t lcov[MAXL]; int nl = 0;
    uint64_t m = lower;
    while (m && nl < MAXL) {
        int k = __builtin_ctzll(m);
        m &= m - 1;
        int kind = rs.prim_kind(k);
        lk[nl] = k;
        // line loops: bit 16 marks the kind, bits 0..15 the segments that can touch this block
        lcov[nl] = kind == PR_POLY ? poly_coverage16(rs, k, x0, y0)
                 : (kind == PR_NGON ? ngon_coverage16(rs, k, x0, y0) : (0x10000u | lineloop_block_segments(rs, k, x0, y0)));
        nl++;
    }
    if (m) {   // unusually deep translucent stack: plain per-sample painter
        for (int s = 0; s < 16; s++)
            if ((remaining >> s) & 1u) {
                int c = raster_sample(rs, x0 + (s & 3), y0 - (s >> 2), lower, base);
                sr += c & 0xFF; sg += (c >> 8) & 0xFF; sb += (c >> 16) & 0xFF;
            }
        return;
    }
    for (int j = 0; j < 4; j++) {
        uint32_t rm = (remaining >> (4 * j)) & 0xFu;
        if (!rm) continue;
        int c[4] = {base, base, base, base};
        for (int q = 0; q < nl; q++) {
            if (lcov[q] & 0x10000u) {
                if (!(lcov[q] & 0xFFFFu)) continue;
                double best[4];
                lineloop_alpha_row(rs, lk[q], x0, y0 - j, lcov[q] & 0xFFFFu, best);
                int col = rs.prim_rgb(lk[q]);
                double lr = (double)(col & 0xFF), lg = (double)((col >> 8) & 0xFF), lb = (double)((col >> 16) & 0xFF);
                for (int t = 0; t < 4; t++)
                    if (best[t] > 0.0) {
                        double a = best[t];
                        int r = (int)rz_floor(a * lr + (1.0 - a) * (double)(c[t] & 0xFF) + 0.5);
                        int g = (int)rz_floor(a * lg + (1.0 - a) * (double)((c[t] >> 8) & 0xFF) + 0.5);
                        int b = (int)rz_floor(a * lb + (1.0 - a) * (double)((c[t] >> 16) & 0xFF) + 0.5);
                        c[t] = r | (g << 8) | (b << 16);
                    }
            } else {
                uint32_t cv = (lcov[q] >> (4 * j)) & 0xFu;
                int col = rs.prim_rgb(lk[q]);
                for (int t = 0; t < 4; t++) if ((cv >> t) & 1u) c[t] = col;
            }
        }
        for (int t = 0; t < 4; t++)
            if ((rm >> t) & 1u) { sr += c[t] & 0xFF; sg += (c[t] >> 8) & 0xFF; sb += (c[t] >> 16) & 0xFF; }
    }
}

// ---- one 96x96 output pixel (X, Y), Y = 0 at the top, in two steps
// step 1: classify the tile's undecided prims against this pixel's 4x4 sample block.  Returns the prims still
// undecided (0 = the pixel is `base`, done) and updates `base` to the colour under them.
MGX_HD uint64_t pixel_classify(const Raster &rs, int X, int Y, uint64_t tile_mixed, int &base) {
    const double xc = 4.0 * X + 2.0, yc = (double)NATIVE_RES - 4.0 * Y - 2.0;
    const int gx0 = 4 * X, gx1 = 4 * X + 3, gy1 = NATIVE_RES - 1 - 4 * Y, gy0 = gy1 - 3;   // 384-grid index range of the block
    uint64_t mixed = 0;
    // front to back: stop at the topmost primitive that covers the whole 4x4 block
    uint64_t m = tile_mixed;
    while (m) {
        int k = 63 - __builtin_clzll(m);
        m &= ~(1ull << k);
        if (RI(bb, 4 * k) > gx1 || RI(bb, 4 * k + 2) < gx0 || RI(bb, 4 * k + 1) > gy1 || RI(bb, 4 * k + 3) < gy0) continue;
        int cls = classify_rect(rs, k, xc, yc, 1.5, 1.5);
        if (cls == CLS_ALL) { base = rs.prim_rgb(k); break; }
        if (cls == CLS_MIXED) mixed |= 1ull << k;
    }
    return mixed;
}
// step 2: resolve the 16 samples of an undecided pixel.  Opaque prims (front to back) claim samples through
// coverage masks; once a translucent line loop is reached, the samples still unclaimed are blended per sample.
MGX_HD int pixel_resolve(const Raster &rs, int X, int Y, uint64_t mixed, int base) {
    const double x0 = 4.0 * X + 0.5, y0 = (double)NATIVE_RES - 0.5 - 4.0 * Y;
    uint32_t remaining = 0xFFFFu;
    int sr = 0, sg = 0, sb = 0;
    uint64_t m = mixed;
    while (m && remaining) {
        int k = 63 - __builtin_clzll(m);
        m &= ~(1ull << k);
        int kind = rs.prim_kind(k);
        if (kind == PR_LINELOOP) {
            // translucent: everything from here down is blended per sample, for the samples nobody above claimed
            resolve_lower_stack(rs, mixed & ((2ull << k) - 1ull), x0, y0, base, remaining, sr, sg, sb);
            remaining = 0;
            break;
        }
        uint32_t cov = (kind == PR_POLY ? poly_coverage16(rs, k, x0, y0) : ngon_coverage16(rs, k, x0, y0)) & remaining;
        if (cov) {
            int n = __builtin_popcount(cov), col = rs.prim_rgb(k);
            sr += n * (col & 0xFF); sg += n * ((col >> 8) & 0xFF); sb += n * ((col >> 16) & 0xFF);
            remaining &= ~cov;
        }
    }
    if (remaining) {
        int n = __builtin_popcount(remaining);
        sr += n * (base & 0xFF); sg += n * ((base >> 8) & 0xFF); sb += n * ((base >> 16) & 0xFF);
    }
    // cv2 INTER_AREA integer-factor path: saturate_cast<uchar>(sum * (1/16)) = round half to even
    int r = (sr + 7 + ((sr >> 4) & 1)) >> 4, g = (sg + 7 + ((sg >> 4) & 1)) >> 4, b = (sb + 7 + ((sb >> 4) & 1)) >> 4;
    return r | (g << 8) | (b << 16);
}
MGX_HD int raster_pixel_lores(const Raster &rs, int X, int Y, uint64_t tile_mixed, int base) {
    uint64_t mixed = pixel_classify(rs, X, Y, tile_mixed, base);
    return mixed ? pixel_resolve(rs, X, Y, mixed, base) : base;
}

// whole-tile classification of every prim (one lane per tile): base colour + undecided set
MGX_HD void classify_tile_all(const Raster &rs, int tile, int bg_rgb, int &base, uint64_t &mixed) {
    const int tcol = tile % TILES_X, trow = tile / TILES_X;
    uint64_t all_mask = 0, mixed_mask = 0;
    for (int k = rs.h->n_prims - 1; k >= 0; k--) {       // front to back: nothing under a covering prim matters
        int cls = classify_tile(rs, k, tcol, trow);
        if (cls == CLS_ALL) { all_mask = 1ull << k; break; }
        if (cls == CLS_MIXED) mixed_mask |= 1ull << k;
    }
    base = bg_rgb;
    tile_resolve(rs, all_mask, mixed_mask, base);
    mixed = mixed_mask;
}

}  // namespace mgx

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
constexpr float CLASS_EPS_F = 2e-3f;   // fp32 classification margin (px): covers coefficient + evaluation rounding at |x|,|y| <= 384

// per-env raster scratch (LDS on device)
struct RasterOff {
    int bx, by, ba, bc, bs;             // per body pose (doubles)
    int svx, svy, einv, ea, eb, ec;     // per prim vertex: screen position, sign / length of edge (i -> i+1) and -- unless `compact` --
                                        // the normalised edge function's coefficients; compact: the exact tests rebuild them (edge_coeffs)
    int vstride, compact;               // doubles per vertex record (6, compact 3)
    int elen, earc;                     // per LINE-LOOP vertex: segment length and arclength at the segment start
    int pcx, pcy, prad, papo, pphi;     // per prim (n-gon centre/radius/apothem/phase; line half width in prad)
    int n_d;
    int prgb;                           // per prim colour of THIS env (the template's, or the env's own: TestColour variants)
    // fp32 classification items (8 words each) in FRONT-TO-BACK prim order + per-prim (start | count << 16)
    int items, pitem, n_items;
    int n_i;
#ifndef MGX_RAOS
#define MGX_RAOS 1      // records: bit 0 per draw-list vertex (7 doubles: -1.6 %), 1 per primitive (5: slower), 2 per body (5: no change)
#endif
    static constexpr bool A_V = MGX_RAOS & 1, A_P = MGX_RAOS & 2, A_B = MGX_RAOS & 4;
    static constexpr int S_bx = A_B ? 5 : 1, S_by = S_bx, S_ba = S_bx, S_bc = S_bx, S_bs = S_bx;
    static constexpr int S_elen = 1, S_earc = 1;      // (the vertex records have a run-time stride: RDV)
    static constexpr int S_pcx = A_P ? 5 : 1, S_pcy = S_pcx, S_prad = S_pcx, S_papo = S_pcx, S_pphi = S_pcx;
    static constexpr int S_prgb = 1, S_items = 1, S_pitem = 1;
    // compact: three doubles per draw-list vertex less (2.6 KB of LDS in ClusterColour: a fourth workgroup per CU) for a dozen more
    // fp64 operations per edge in the exact tests; the host takes it where it buys a workgroup per CU (configure_launch)
    MGX_HD explicit RasterOff(const TmplHeader &h, bool compact_ = true) {
        int o = 0;
        compact = compact_ ? 1 : 0; vstride = compact_ ? 3 : 6;
        { const int d = A_B ? 1 : h.n_bodies; bx = o; by = o + d; ba = o + 2 * d; bc = o + 3 * d; bs = o + 4 * d; o += 5 * h.n_bodies; }
        svx = o; svy = o + 1; einv = o + 2; ea = o + 3; eb = o + 4; ec = o + 5; o += vstride * h.n_pverts;
        elen = o; o += h.n_lverts; earc = o; o += h.n_lverts;
        { const int d = A_P ? 1 : h.n_prims; pcx = o; pcy = o + d; prad = o + 2 * d; papo = o + 3 * d; pphi = o + 4 * d; o += 5 * h.n_prims; }
        n_d = o;
        o = 0;
        prgb = o; o += h.n_prims;
        n_items = h.n_pverts + h.n_prims;          // upper bound: one per polygon edge / line segment / n-gon (the exact count,
                                                   // 0.4-0.8 KB less, buys no workgroup anywhere and shifts MoveToCorner's tables: -0.4 %)
        o = (o + 3) & ~3;                          // 16-byte aligned records
        items = o; o += 8 * n_items;
        pitem = o; o += h.n_prims;
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
    MGX_HD Raster(const TmplHeader *h_, const int32_t *ti_, const double *tq_, double *d_, int32_t *i_, int view_, bool compact = true)
        : h(h_), ti(ti_), tq(tq_), d(d_), i(i_), to(*h_), ro(*h_, compact), view(view_) {}
    MGX_HD int prim_kind(int k) const { return ti[to.prim_i + k * PRIM_IWORDS]; }
    MGX_HD int prim_nv(int k) const { return ti[to.prim_i + k * PRIM_IWORDS + 1]; }
    MGX_HD int prim_voff(int k) const { return ti[to.prim_i + k * PRIM_IWORDS + 2] & 0xFFFF; }
    MGX_HD int prim_lvoff(int k) const { return ti[to.prim_i + k * PRIM_IWORDS + 2] >> 16; }     // line loops: first slot of elen / earc
    MGX_HD int prim_xf(int k) const { return ti[to.prim_i + k * PRIM_IWORDS + 3]; }
    MGX_HD int prim_rgb(int k) const { return i[ro.prgb + k]; }       // after raster_setup_prims
    MGX_HD int prim_rgb_template(int k) const { return ti[to.prim_i + k * PRIM_IWORDS + 4]; }
    MGX_HD int prim_stipple(int k) const { return ti[to.prim_i + k * PRIM_IWORDS + 5] & 0xFFFF; }
    MGX_HD int prim_goal(int k) const { return ((ti[to.prim_i + k * PRIM_IWORDS + 5] >> 16) & 0x1F) - 1; }   // goal region ordinal, -1: none
    MGX_HD int prim_item_start(int k) const { return (int)((uint32_t)ti[to.prim_i + k * PRIM_IWORDS + 5] >> 21); }   // first classification item (front-to-back list)
    MGX_HD uint32_t prim_ends(int k) const { return (uint32_t)ti[to.prim_i + k * PRIM_IWORDS + 6]; }    // bit i: vertex i ends a convex part
    // tq layout: [n_prims * PRIM_RWORDS][pvx n_pverts][pvy n_pverts]
    MGX_HD double prim_r(int k, int j) const { return tq[k * PRIM_RWORDS + j]; }
    MGX_HD double pvx(int v) const { return tq[h->n_prims * PRIM_RWORDS + v]; }
    MGX_HD double pvy(int v) const { return tq[h->n_prims * PRIM_RWORDS + h->n_pverts + v]; }
};

#define RD(field, k) rs.d[rs.ro.field + (k) * RasterOff::S_##field]
#define RI(field, k) rs.i[rs.ro.field + (k) * RasterOff::S_##field]
#define RDV(field, v) rs.d[rs.ro.field + (v) * rs.ro.vstride]      // fields of the draw-list vertex records

MGX_HD double rz_floor(double x) { return floor(x); }

#ifdef MGX_RASTER_STATS     // host emulation only (tests/emu): operation counts of the resolve path
inline long g_rstat[16];
#define MGX_RSTAT(i, n) g_rstat[i] += (n)
#else
#define MGX_RSTAT(i, n)
#endif

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
        r_sincos_lib(v[2], s, c);
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

// ---------------------------------------------------------------- fp32 classification items
// The conservative ALL / NONE / MIXED classification runs in fp32 on a flat list of ITEMS -- one per polygon
// edge, per n-gon, per line-loop segment -- stored front-to-back.  A wavefront loads up to 64 items (one per
// lane) and every lane (a tile in phase C, an output pixel in phase T) consumes them through v_readlane
// broadcasts: the item fields become scalar operands, the loop is wave-uniform and touches no LDS.
enum { IT_EDGE = 0, IT_NGON = 1, IT_SEG = 2 };
// A polygon made of several convex parts (a star) is classified part by part, each part like a primitive of its own
// with the same index: the union touches the block when a part does (bit k of `mixed`) and covers it when a part does.
// A block covered only by several parts together counts as MIXED, which costs time, not correctness.
constexpr int IT_LAST = 4;      // meta bit: last item of its convex part
// meta = kind | IT_LAST | (items left in the part, this one included) << 3 | prim << 12
constexpr int IT_REM_SHIFT = 3, IT_REM_MASK = 0x1FF, IT_K_SHIFT = 12;
struct Item {
    // (two 16-byte halves: a polygon edge's PIXEL-level classification reads the second one only -- g0 g1 g2 + meta --, its tile-level
    // one the first -- a b c g3)
    float a, b, c;              // EDGE/SEG: normalised line a x + b y + c;  NGON: centre x, y, apothem
    float g3;
    float g0, g1, g2;           // EDGE: g0 .. g2 = edge function over its extent over a 4x4 block, g3 = 1 / extent over a tile;  SEG: start x, y, length, g3 = half width;  NGON: g0 = circumradius
    int meta;                   // kind | IT_LAST | items left in the part << 3 | prim << 12
};
static_assert(sizeof(Item) == 32, "two 16-byte halves");
// per-primitive word of the item list (RasterOff::pitem): first item | items << 16 | primitive kind << 24 | (several convex parts) << 26
constexpr int PI_CNT_MASK = 0xFF;
constexpr int TILE_W = 16, TILE_H = 4, TILES_X = LORES / TILE_W, TILES_Y = LORES / TILE_H;
constexpr float TILE_HX = 2.0f * TILE_W - 0.5f, TILE_HY = 2.0f * TILE_H - 0.5f;


// ---- setup phase 2: per-primitive records (lane per prim) and screen-space vertices (lane per prim vertex)
MGX_HD int prim_item_count(const Raster &rs, int k) { return rs.prim_kind(k) == PR_NGON ? 1 : rs.prim_nv(k); }
// env_col (optional): [n_entities][stride] colour index of every entity in THIS env (Test*Colour variants); a primitive
// painted by an entity takes palette[4 * role + colour] (role 0 darkened outline, 1 base, 2 lightened interior)
MGX_HD void raster_setup_prims(Raster &rs, int lane, int nl, const int32_t *env_col = nullptr, long stride = 0, long env = 0,
                               const double *env_goal = nullptr, const int32_t *palette = nullptr) {
    const TmplHeader &h = *rs.h;
    double cam[6];
    raster_camera(rs, cam);
    for (int k = lane; k < h.n_prims; k += nl) {
        {
            const int xw = rs.prim_xf(k), role = ((xw >> 24) & 3) - 1, ent = ((xw >> 26) & 0x3F) - 1;
            RI(prgb, k) = (env_col && ent >= 0 && role >= 0) ? palette[4 * role + env_col[(long)ent * stride + env]] : rs.prim_rgb_template(k);
        }
        // the prim's slice of the item list: front (top) prims first (the world builder did the sum)
        const int start = rs.prim_item_start(k);
        RI(pitem, k) = start | (prim_item_count(rs, k) << 16) | (rs.prim_kind(k) << 24) |
                       ((rs.prim_kind(k) == PR_POLY && (rs.prim_ends(k) & ~(1u << (rs.prim_nv(k) - 1))) != 0) ? 1 << 26 : 0);
        const int kind = rs.prim_kind(k);
        if (kind == PR_LINELOOP) RD(prad, k) = rs.prim_r(k, 4);
        if (kind != PR_NGON) continue;
        int xfw = rs.prim_xf(k);
        int xf = xfw & 0xFF, body = (xfw >> 8) & 0xFF, eye_body = ((xfw >> 16) & 0xFF) - 1;
        int nv = rs.prim_nv(k);
        double bx = RD(bx, body), by = RD(by, body), bc = RD(bc, body), bs = RD(bs, body);
        double ebx = rs.prim_r(k, 0), eby = rs.prim_r(k, 1), epx = rs.prim_r(k, 2), epy = rs.prim_r(k, 3);
        double da = 0.0, dc = 1.0, ds = 0.0;
        if (xf == XF_EYE && eye_body >= 0) {
            da = RD(ba, eye_body) - RD(ba, body);
            r_sincos_lib(da, ds, dc);
        }
        // centre = image of the local origin; phase = world angle of vertex 0 minus camera rotation
        double lx = 0.0, ly = 0.0;
        if (xf == XF_EYE) { double qx = lx + epx, qy = ly + epy; lx = (dc * qx - ds * qy) + ebx; ly = (ds * qx + dc * qy) + eby; }
        double wx = lx, wy = ly;
        if (xf != XF_WORLD) { wx = bx + (bc * lx - bs * ly); wy = by + (bc * ly + bs * lx); }
        const double pcx = cam[0] * wx + cam[1] * wy + cam[4], pcy = cam[2] * wx + cam[3] * wy + cam[5];
        RD(pcx, k) = pcx; RD(pcy, k) = pcy;
        double sc = (double)NATIVE_RES / 2.04;
        double rad = rs.prim_r(k, 5) * sc, apo = rad * rs.prim_r(k, 4);       // (prim_r 4 of an n-gon: cos(pi / nv), from the world builder)
        RD(prad, k) = rad;
        RD(papo, k) = apo;
        double phi = 0.0;
        if (xf != XF_WORLD) phi += RD(ba, body);
        if (xf == XF_EYE) phi += da;
        if (rs.view == 0) phi -= RD(ba, rs.h->robot_body);
        RD(pphi, k) = phi;
        Item *it = reinterpret_cast<Item *>(&RI(items, 8 * start));
        it[0].a = (float)pcx; it[0].b = (float)pcy; it[0].c = (float)apo; it[0].g0 = (float)rad;
        it[0].g1 = it[0].g2 = it[0].g3 = 0.0f;
        it[0].meta = IT_NGON | IT_LAST | (1 << IT_REM_SHIFT) | (k << IT_K_SHIFT);
    }
    for (int v = lane; v < h.n_pverts; v += nl) {
        const int k = rs.ti[rs.to.pv_prim + v], i = v - rs.prim_voff(k);
        const int xfw = rs.prim_xf(k), xf = xfw & 0xFF, body = (xfw >> 8) & 0xFF;
        double lx = rs.pvx(v), ly = rs.pvy(v);
        // goal regions whose rectangle differs per env (x, y = top-left corner, then h, w): rebuild the four corners
        // the way the world builder does (gym_render.py:449-453 order around the box centre, entities.py:794-797)
        const int goal = env_goal ? rs.prim_goal(k) : -1;
        if (goal >= 0) {
            const double gx = env_goal[(long)(4 * goal) * stride + env], gy = env_goal[(long)(4 * goal + 1) * stride + env];
            const double gh = env_goal[(long)(4 * goal + 2) * stride + env], gw = env_goal[(long)(4 * goal + 3) * stride + env];
            const double gcx = gx + gw / 2, gcy = gy - gh / 2;
            lx = ((i == 1 || i == 2) ? gw / 2 : -gw / 2) + gcx; ly = ((i < 2) ? gh / 2 : -gh / 2) + gcy;
        }
        double wx = lx, wy = ly;
        if (xf != XF_WORLD) {
            const double bx = RD(bx, body), by = RD(by, body), bc = RD(bc, body), bs = RD(bs, body);
            wx = bx + (bc * lx - bs * ly); wy = by + (bc * ly + bs * lx);
        }
        RDV(svx, v) = cam[0] * wx + cam[1] * wy + cam[4];
        RDV(svy, v) = cam[2] * wx + cam[3] * wy + cam[5];
    }
}

// coefficients of the normalised edge function a x + b y + c of the edge that starts at (ax, ay) with direction (ex, ey), inv =
// sign / length.  One definition for the set-up (which derives the fp32 items from it) and for the exact tests (which rebuild it
// instead of keeping three doubles per edge in LDS), without contraction so that both get the same bits.
MGX_HD void edge_coeffs(double ax, double ay, double ex, double ey, double inv, double &a, double &b, double &c) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    a = -ey * inv; b = ex * inv;
    const double t0 = ey * ax, t1 = ex * ay;
    c = (t0 - t1) * inv;
}
// ... of edge i of primitive k (vertices at vo ..): its end point is the next vertex, or the first vertex of its convex part
// (`first`) when i closes the part
MGX_HD void edge_coeffs_of(const Raster &rs, int vo, int i, int first, bool closes, double &a, double &b, double &c) {
    if (!rs.ro.compact) { a = RDV(ea, vo + i); b = RDV(eb, vo + i); c = RDV(ec, vo + i); return; }
    const int j = closes ? first : i + 1;
    const double ax = RDV(svx, vo + i), ay = RDV(svy, vo + i);
    edge_coeffs(ax, ay, RDV(svx, vo + j) - ax, RDV(svy, vo + j) - ay, RDV(einv, vo + i), a, b, c);
}
// ---- setup phase 3 (after a barrier): lane per prim vertex = per polygon edge / line segment (vertex i -> its successor):
// normalised edge function E(p) = sgn * cross(e, p - a) / |e| (>= 0 inside a polygon; for a line loop |E| is the
// distance to the segment's carrier line and (eb, -ea) is its unit direction) and the edge's classification item.
// A polygon's convex parts are closed loops of their own: the edge after a part's last vertex returns to its first.
MGX_HD void raster_setup_edges(Raster &rs, int lane, int nl) {
    const TmplHeader &h = *rs.h;
    for (int v = lane; v < h.n_pverts; v += nl) {
        const int k = rs.ti[rs.to.pv_prim + v], vo = rs.prim_voff(k), nv = rs.prim_nv(k), i = v - vo;
        const int kind = rs.prim_kind(k);
        const uint32_t ends = rs.prim_ends(k);
        const int p1 = i + __builtin_ctz(ends >> i);                                              // last vertex of this part
        const uint32_t before = ends & ((1u << i) - 1u);
        const int p0 = before ? 32 - __builtin_clz(before) : 0;                                  // first vertex of this part
        double sgn = 1.0;
        if (kind == PR_POLY) {
            double area2 = 0.0;
            for (int a = p0; a <= p1; a++) {
                int b = a == p1 ? p0 : a + 1;
                area2 += RDV(svx, vo + a) * RDV(svy, vo + b) - RDV(svy, vo + a) * RDV(svx, vo + b);
            }
            sgn = area2 >= 0.0 ? 1.0 : -1.0;
        }
        const int j = i == p1 ? p0 : i + 1;
        const double ax = RDV(svx, vo + i), ay = RDV(svy, vo + i), ex = RDV(svx, vo + j) - ax, ey = RDV(svy, vo + j) - ay;
        const double len = sqrt(ex * ex + ey * ey);
        const double inv = sgn / len;
        double ea, eb, ec;
        edge_coeffs(ax, ay, ex, ey, inv, ea, eb, ec);
        RDV(einv, v) = inv;
        if (!rs.ro.compact) { RDV(ea, v) = ea; RDV(eb, v) = eb; RDV(ec, v) = ec; }
        Item *it = reinterpret_cast<Item *>(&RI(items, 8 * ((RI(pitem, k) & 0xFFFF) + i)));
        it->a = (float)ea; it->b = (float)eb; it->c = (float)ec;
        if (kind == PR_LINELOOP) {
            // arclength at the segment's start: the lengths of the loop's earlier segments, summed in order
            double arc = 0.0;
            for (int a = 0; a < i; a++) {
                const double dx = RDV(svx, vo + a + 1) - RDV(svx, vo + a), dy = RDV(svy, vo + a + 1) - RDV(svy, vo + a);
                arc += sqrt(dx * dx + dy * dy);
            }
            const int lv = rs.prim_lvoff(k) + i;
            RD(elen, lv) = len; RD(earc, lv) = arc;
            it->g0 = (float)ax; it->g1 = (float)ay; it->g2 = (float)len; it->g3 = (float)RD(prad, k);
        } else {
            // edge function divided by its conservative half-extent over a 4x4 sample block (g0..g2: the block is
            // entirely inside / outside this edge when the scaled value at its centre is > 1 / < -1), and the
            // reciprocal half-extent over a whole tile (g3)
            const float fa = r_abs(it->a), fb = r_abs(it->b);
            const float ip = 1.0f / (1.5f * (fa + fb) + CLASS_EPS_F);
            it->g0 = it->a * ip; it->g1 = it->b * ip; it->g2 = it->c * ip;
            it->g3 = 1.0f / (TILE_HX * fa + TILE_HY * fb + CLASS_EPS_F);
        }
        it->meta = (kind == PR_POLY ? IT_EDGE : IT_SEG) | (i == p1 ? IT_LAST : 0) | ((p1 - i + 1) << IT_REM_SHIFT) | (k << IT_K_SHIFT);
        (void)nv;
    }
}

// ---- exact sample tests (fp64)
MGX_HD bool poly_contains(const Raster &rs, int k, double x, double y) {
    int nv = rs.prim_nv(k), vo = rs.prim_voff(k);
    const uint32_t ends = rs.prim_ends(k);
    bool in = true;                                 // inside the part being walked; the polygon is the union of its parts
    int first = 0;
    for (int i = 0; i < nv; i++) {
        const bool closes = (ends >> i) & 1u;
        double a, b, c;
        edge_coeffs_of(rs, vo, i, first, closes, a, b, c);
        if (a * x + b * y + c < 0.0) in = false;
        if (closes) { if (in) return true; in = true; first = i + 1; }
    }
    return false;
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
    r_sincos_lib(mid, s, c);
    return qx * c + qy * s <= apo;
}
// max coverage alpha of a smooth line loop at a sample; OUR model of GL_LINE_SMOOTH (driver-defined):
// alpha = clamp(halfwidth - dist, 0, 1), halfwidth = (w + 1) / 2, 16-px stipple by arclength.
MGX_HD double lineloop_alpha(const Raster &rs, int k, double x, double y) {
    int nv = rs.prim_nv(k), vo = rs.prim_voff(k), lvo = rs.prim_lvoff(k), stipple = rs.prim_stipple(k);
    double hw = RD(prad, k), best = 0.0;
    for (int i = 0; i < nv; i++) {
        double a, b, c;
        edge_coeffs_of(rs, vo, i, 0, i == nv - 1, a, b, c);
        double e = a * x + b * y + c;
        if (r_abs(e) >= hw) continue;                       // distance to the segment >= distance to its line
        double ax = RDV(svx, vo + i), ay = RDV(svy, vo + i), len = RD(elen, lvo + i);
        double sl = r_clamp((x - ax) * b - (y - ay) * a, 0.0, len);   // arclength of the closest point
        double qx = x - (ax + b * sl), qy = y - (ay - a * sl);
        double alpha = r_clamp01(hw - sqrt(qx * qx + qy * qy));
        if (alpha > 0.0 && stipple) {
            int bit = ((int)rz_floor(RD(earc, lvo + i) + sl)) & 15;
            if (!((stipple >> bit) & 1)) alpha = 0.0;
        }
        if (alpha > best) best = alpha;
    }
    return best;
}

// as lineloop_alpha, restricted to the segments in `segmask` (the others were excluded by the fp32 conservative test)
MGX_HD double lineloop_alpha_masked(const Raster &rs, int k, double x, double y, uint32_t segmask) {
    int nv = rs.prim_nv(k), vo = rs.prim_voff(k), lvo = rs.prim_lvoff(k), stipple = rs.prim_stipple(k);
    double hw = RD(prad, k), best = 0.0;
    for (; segmask; segmask &= segmask - 1) {
        const int i = __builtin_ctz(segmask);
        double a, b, c;
        edge_coeffs_of(rs, vo, i, 0, i == nv - 1, a, b, c);
        double e = a * x + b * y + c;
        if (r_abs(e) >= hw) continue;
        double ax = RDV(svx, vo + i), ay = RDV(svy, vo + i), len = RD(elen, lvo + i);
        double sl = r_clamp((x - ax) * b - (y - ay) * a, 0.0, len);
        double qx = x - (ax + b * sl), qy = y - (ay - a * sl);
        double alpha = r_clamp01(hw - sqrt(qx * qx + qy * qy));
        if (alpha > 0.0 && stipple) {
            int bit = ((int)rz_floor(RD(earc, lvo + i) + sl)) & 15;
            if (!((stipple >> bit) & 1)) alpha = 0.0;
        }
        if (alpha > best) best = alpha;
    }
    return best;
}

// Primitive sets are bit masks (bit k = prim k): 64 bits in general, 32 where every world the engine holds has at most 32 primitives
// (all Demo worlds: half the words of the pixel queue and of the per-tile tables in LDS, one instruction instead of two per mask
// operation; the host's choice, configure_launch)
MGX_HD int mask_top(uint64_t m) { return 63 - __builtin_clzll(m); }
MGX_HD int mask_top(uint32_t m) { return 31 - __builtin_clz(m); }
MGX_HD int mask_low(uint64_t m) { return __builtin_ctzll(m); }
MGX_HD int mask_low(uint32_t m) { return __builtin_ctz(m); }
// one 384-grid sample, painter's order over the primitives in `mask` (bit k = prim k), starting from `base_rgb`
template <typename M> MGX_HD int raster_sample(const Raster &rs, double x, double y, M mask, int base_rgb) {
    int r = base_rgb & 0xFF, g = (base_rgb >> 8) & 0xFF, b = (base_rgb >> 16) & 0xFF;
    while (mask) {
        int k = mask_low(mask);
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

MGX_HD int raster_total_items(const Raster &rs) {
    int pi = RI(pitem, 0);                      // prim 0 is the rearmost: its items end the list
    return (pi & 0xFFFF) + ((pi >> 16) & PI_CNT_MASK);
}
MGX_HD Item load_item(const Raster &rs, int idx) { return reinterpret_cast<const Item *>(&RI(items, 0))[idx]; }

// item index of `slot` in the concatenation (front to back) of the items of the prims in `mask`; -1 past the end
template <typename M> MGX_HD int masked_item_index(const Raster &rs, M mask, int slot, int &n_total) {
    int acc = 0, found = -1;
    while (mask) {
        int k = mask_top(mask);
        mask &= ~(M(1) << k);
        int pi = RI(pitem, k), start = pi & 0xFFFF, cnt = (pi >> 16) & PI_CNT_MASK;
        if (slot >= acc && slot < acc + cnt) found = start + (slot - acc);
        acc += cnt;
    }
    n_total = acc;
    return found;
}

// per-lane classification state carried across item chunks.  For the polygon being consumed `lo` is the minimum over
// its edges of the edge function at the block centre divided by the edge's conservative half-extent over the block:
//   lo < -1 -> NONE (some edge excludes the whole block),  lo > +1 -> ALL (every edge contains the whole block)
// (n-gons report -2 / 0 / +2 on the same scale; for line loops `lo` is the max touch margin: touched when lo >= 0).
struct ClassState {
    uint64_t mixed; int base; int decided; float lo;
    int line;                        // some primitive of `mixed` is a line loop (the pixel queue keeps those apart, mgx_raster_body.inc)
    MGX_HD void init(int bg) { mixed = 0; base = bg; decided = 0; lo = 1e30f; line = 0; }
};
constexpr float BIG_F = 1e30f;
// Consume one item for the sample block centred at (xc, yc): TILE = whole 16x4-pixel tile (half extents TILE_HX/HY),
// otherwise one output pixel's 4x4 sample block (half extents 1.5).  ALL = every sample inside an opaque prim,
// NONE = no sample touched, MIXED = decide per sample; conservative by CLASS_EPS_F.  `I` is wave-uniform on the
// device (readlane), so the branches on its kind are scalar.
template <bool TILE> MGX_HD void classify_item(const Raster &rs, const Item &I, float xc, float yc, ClassState &st) {
    const int kind = I.meta & 3;
    const float hx = TILE ? TILE_HX : 1.5f, hy = TILE ? TILE_HY : 1.5f;
    if (kind == IT_EDGE) {
        // scaled edge function: the block is outside this edge below -1, inside above +1; `lo` keeps the minimum
        const float e = TILE ? (I.a * xc + (I.b * yc + I.c)) * I.g3 : I.g0 * xc + (I.g1 * yc + I.g2);
        st.lo = r_min(st.lo, e);
    } else if (kind == IT_NGON) {
        const float qx = r_abs(xc - I.a), qy = r_abs(yc - I.b);
        const float nx = r_max(qx - hx, 0.0f), ny = r_max(qy - hy, 0.0f);        // nearest point of the rect
        const float fx = qx + hx, fy = qy + hy;                                    // farthest corner
        const float apo = I.c - CLASS_EPS_F, rad = I.g0 + CLASS_EPS_F;
        const float lo = rad * rad - (nx * nx + ny * ny);
        const float hi = apo > 0.0f ? apo * apo - (fx * fx + fy * fy) : -1.0f;
        st.lo = lo < 0.0f ? -2.0f : (hi > 0.0f ? 2.0f : 0.0f);                    // same scale as the edges
    } else {
        const float hw = I.g3 + CLASS_EPS_F;
        const float e = I.a * xc + (I.b * yc + I.c);
        const float sl = (xc - I.g0) * I.b - (yc - I.g1) * I.a;                    // along the segment
        const float el = hx * r_abs(I.a) + hy * r_abs(I.b), es = hx * r_abs(I.b) + hy * r_abs(I.a);
        // touch margin: >= 0 iff the block is within hw of the carrier line and not beyond the segment's ends
        const float t = r_min(r_min(hw + el - r_abs(e), sl + es + hw), I.g2 + hw + es - sl);
        st.lo = st.lo >= BIG_F ? t : r_max(st.lo, t);
    }
    if (I.meta & IT_LAST) {
        const int k = I.meta >> IT_K_SHIFT;
        if (!st.decided) {
            if (kind == IT_SEG) { if (st.lo >= 0.0f) { st.mixed |= 1ull << k; st.line = 1; } }
            else if (!(st.lo < -1.0f)) {
                if (st.lo > 1.0f) { st.base = rs.prim_rgb(k); st.decided = 1; }     // topmost covering prim hides the rest
                else st.mixed |= 1ull << k;
            }
        }
        st.lo = BIG_F;
    }
}

// host-side item source (the device uses registers + readlane, see mgx_raster.hip)
template <bool TILE> MGX_HD void classify_items_array(const Raster &rs, const Item *items, int n, float xc, float yc, ClassState &st) {
    for (int i = 0; i < n && !st.decided; i++) classify_item<TILE>(rs, items[i], xc, yc, st);
}

MGX_HD void tile_centre(int tile, float &xc, float &yc) {
    const int tcol = tile % TILES_X, trow = tile / TILES_X;
    const int gx0 = 4 * TILE_W * tcol, gy1 = NATIVE_RES - 1 - 4 * TILE_H * trow, gy0 = gy1 - 4 * TILE_H + 1;
    xc = 0.5f * (gx0 + gx0 + 4 * TILE_W); yc = 0.5f * (gy0 + gy1 + 1);
}
// ---------------------------------------------------------------- resolving an undecided pixel (one lane per pixel)
// Phase Q's sixteen-sample loops, written for their instruction count (a wavefront's time is its instruction count, and these loops are a
// quarter of the rasteriser's): a sample's yes / no is the SIGN of a difference, shifted into a bit mask by one v_alignbit_b32 -- instead of
// compare + select + or -- in reverse sample order (q_rev16 turns the mask round at the end).
#if defined(__HIP_DEVICE_COMPILE__)
typedef float float2_t __attribute__((ext_vector_type(2)));       // (a pair the compiler keeps in two adjacent registers: v_pk_add_f32)
#else
struct float2_t { float x, y; };
inline float2_t operator-(float2_t a, float2_t b) { return {a.x - b.x, a.y - b.y}; }
inline float2_t &operator+=(float2_t &a, float2_t b) { a.x += b.x; a.y += b.y; return a; }
#endif
MGX_HD uint32_t q_push_sign(uint32_t acc, float v) {          // (acc << 1) | (v < 0 or v == -0)
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(acc, __builtin_bit_cast(uint32_t, v), 31);
#else
    return (acc << 1) | (__builtin_bit_cast(uint32_t, v) >> 31);
#endif
}
MGX_HD uint32_t q_rev16(uint32_t acc) {                         // sixteen pushes: sample q sits at bit 15 - q
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse32(acc) >> 16;
#else
    uint32_t r = 0;
    for (int q = 0; q < 16; q++) r |= ((acc >> (15 - q)) & 1u) << q;
    return r;
#endif
}
MGX_HD int q_bit_mask(uint32_t word, uint32_t pos) {            // all ones when bit (pos & 31) of word is set, else 0: one v_bfe_i32
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sbfe((int)word, pos, 1u);
#else
    return -(int)((word >> (pos & 31u)) & 1u);
#endif
}
MGX_HD float q_sqrt(float x) {                                  // v_sqrt_f32, 1 ulp (the library's correctly rounded sqrtf is 17 instructions)
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
MGX_HD float q_fract(float x) {                                 // x - floor(x), exact for 0 <= x < 2^23
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fractf(x);
#else
    return x - floorf(x);
#endif
}
MGX_HD float q_abs(float x) { return __builtin_fabsf(x); }      // (an operand modifier on the device, where r_abs is compare + select)
MGX_HD float q_and(float x, int mask) {
    return __builtin_bit_cast(float, __builtin_bit_cast(int, x) & mask);
}
// 16-bit coverage of the 4x4 sample block whose top-left sample is (x0, y0); bit 4*j + i = sample (x0 + i, y0 - j).
// fp32 first: a sample is decided in fp32 when |E| > CLASS_EPS_F, otherwise that one sample is re-evaluated in fp64
// against the fp64 edge function -- the result equals the all-fp64 test.
// (Rounds 2-3 kept an edge-by-edge form of this for the rasteriser's 96-register variant, which then spilled 63 registers; since phase Q was
// shrunk in round 4 the two-pass form wins there too: MoveToCorner 7.65 -> 7.77 M env-steps/s, profiles/r04_raster_phase_q_shrink_ab.txt.)
MGX_HD uint32_t poly_coverage16(const Raster &rs, int k, int X, int Y, uint32_t &unc) {
#ifdef MGX_Q_NO_POLY      // development probe: what phase Q costs without the polygon coverage arithmetic (wrong pixels)
    return 0x0F0Fu;
#endif
    const int nv = rs.prim_nv(k), i0 = RI(pitem, k) & 0xFFFF;
    const uint32_t ends = rs.prim_ends(k) | (1u << (nv - 1));               // bit e: edge e closes a convex part
    const float x0 = 4.0f * X + 0.5f, y0 = (float)NATIVE_RES - 0.5f - 4.0f * Y;
    const Item *items = reinterpret_cast<const Item *>(&RI(items, 0)) + i0;
    // Two passes, so that the lanes of a wavefront (each on its own pixel, often on different polygons) diverge as little as
    // possible.  Pass 1, the same few operations per edge for every lane: is the block wholly outside the edge (`dead`), does
    // the edge cross it (`cross`), or is the block wholly inside?  Pass 2 evaluates the 16 samples only for the crossing edges
    // of parts that are still alive -- one to three per pixel, whatever the polygon (a star has 36 edges in 12 parts).
    uint32_t cross = 0, dead = 0;
    for (int e = 0; e < nv; e++) {
        const float a = items[e].a, b = items[e].b;
        const float row = a * x0 + b * y0 + items[e].c;
        // block spans x0..x0+3, y0-3..y0: worst / best corner value of this edge function
        const float lo = row + r_min(0.0f, 3.0f * a) - r_max(0.0f, 3.0f * b);
        const float hi = row + r_max(0.0f, 3.0f * a) - r_min(0.0f, 3.0f * b);
        if (hi < -CLASS_EPS_F) dead |= 1u << e;                             // whole block outside this edge
        else if (lo < CLASS_EPS_F) cross |= 1u << e;                        // (else: whole block inside it)
    }
    // the polygon is the union of its parts: a part with a dead edge holds no sample; one with neither a dead nor a crossing
    // edge holds them all
    uint32_t todo = 0;
    {
        int e0 = 0;
        for (uint32_t er = ends; er; er &= er - 1) {
            const int pe = __builtin_ctz(er);
            const uint32_t rm = (2u << pe) - (1u << e0);                    // edges e0 .. pe
            if (!(dead & rm)) {
                if (!(cross & rm)) return 0xFFFFu;
                todo |= cross & rm;
            }
            e0 = pe + 1;
        }
    }
    // `part` = samples inside every crossing edge of the part being walked (a sample that is ambiguous for one part stays
    // flagged even when another part holds it: the exact painter then decides it)
    uint32_t cov = 0, part = 0xFFFFu;
    int cur_end = -1;
    while (todo) {
        const int e = __builtin_ctz(todo);
        todo &= todo - 1;
        const int pe = e + __builtin_ctz(ends >> e);                        // the edge that closes e's part
        if (pe != cur_end) {
            if (cur_end >= 0) { cov |= part; if (cov == 0xFFFFu) return cov; }
            part = 0xFFFFu; cur_end = pe;
        }
        const float a = items[e].a, b = items[e].b;
        const float row = a * x0 + b * y0 + items[e].c;
        // w = (E - eps, E + eps) of the sample, stepped along the row by one packed add: inside (with the margin) = E - eps >= 0;
        // too close to call = E - eps < 0 <= E + eps
        uint32_t nin = 0, nlow = 0;
        const float2_t step = {a, a};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int j = 0; j < 4; j++) {
            const float rj = row - b * (float)j;
            float2_t w = {rj - CLASS_EPS_F, rj + CLASS_EPS_F};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int i = 0; i < 4; i++) {
                nin = q_push_sign(nin, w.x);
                nlow = q_push_sign(nlow, w.y);
                w += step;
            }
        }
        const uint32_t out = q_rev16(nin), in = out ^ 0xFFFFu, amb = out & ~q_rev16(nlow);
        unc |= amb & part;                                                 // samples too close to call in fp32
        part &= in;
    }
    if (cur_end >= 0) cov |= part;
    return cov;
}
// regular n-gon (circles): inside the in-circle / outside the circum-circle is decided by the radius; in the thin annulus
// between them the sample is tested against the edges of its angular sector and of both neighbours (robust against an
// off-by-one sector from the fp32 angle; the largest of the three equals the polygon's support function there)
constexpr float NGON_TOL_F = 2e-4f;
MGX_HD uint32_t ngon_coverage16(const Raster &rs, int k, int X, int Y, uint32_t &unc) {
    const Item &it = reinterpret_cast<const Item *>(&RI(items, 0))[RI(pitem, k) & 0xFFFF];
    const float x0 = 4.0f * X + 0.5f - it.a, y0 = (float)NATIVE_RES - 0.5f - 4.0f * Y - it.b;
    const float apo = it.c - CLASS_EPS_F, rad = it.g0 + CLASS_EPS_F, apo2 = apo > 0.0f ? apo * apo : -1.0f, rad2 = rad * rad;
    // (apo2 - d2, rad2 - d2) by one packed subtraction per sample: outside the in-circle = the first one negative, beyond the circum-circle =
    // the second (a difference of two floats has the sign of the comparison, exactly)
    uint32_t nout = 0, nfar = 0;
    const float2_t lim = {apo2, rad2};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 4; j++) {
        const float qy = y0 - (float)j, qy2 = qy * qy;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int i = 0; i < 4; i++) {
            const float qx = x0 + (float)i, d2 = qx * qx + qy2;
            const float2_t dd = {d2, d2};
            const float2_t t = lim - dd;
            nout = q_push_sign(nout, t.x);
            nfar = q_push_sign(nfar, t.y);
        }
    }
    const uint32_t outside = q_rev16(nout);
    uint32_t cov = outside ^ 0xFFFFu, ann = outside & ~q_rev16(nfar);
    if (ann) {
        const int n = rs.prim_nv(k);
        const float step = 6.283185307179586f / (float)n, phi = (float)RD(pphi, k);
        float ss, cs;
        r_sincos<float>(step, ss, cs);
        for (; ann; ann &= ann - 1) {
            const int sidx = __builtin_ctz(ann);
            const float qx = x0 + (float)(sidx & 3), qy = y0 - (float)(sidx >> 2);
            const float kk = floorf((atan2f(qy, qx) - phi) / step);
            float s, c;
            r_sincos<float>(phi + (kk + 0.5f) * step, s, c);
            const float d0 = qx * c + qy * s;                                      // this sector's edge
            const float dp = qx * (c * cs - s * ss) + qy * (s * cs + c * ss);      // next sector
            const float dm = qx * (c * cs + s * ss) + qy * (s * cs - c * ss);      // previous sector
            const float d = r_max(d0, r_max(dp, dm)) - it.c;
            if (r_abs(d) < NGON_TOL_F) unc |= 1u << sidx;
            if (d <= 0.0f) cov |= 1u << sidx;
        }
    }
    return cov;
}
// The segments of a line loop that can touch the pixel's 4x4 block (fp32, conservative), as a bit mask; the return value is the set of
// samples the line then CLAIMS: all sixteen.  Alpha is exactly 0 wherever the line is not, and an alpha of 0 leaves the colour underneath, so
// blending every sample of a crossed block gives the painter's pixel -- round 3 first tested the sixteen samples against every crossing
// segment to claim only the touched ones: 64 more instructions per segment in a kernel that pays for its code in spilled registers.
MGX_HD uint32_t lineloop_touch16(const Raster &rs, int k, int X, int Y, uint32_t &segmask) {
    segmask = 0;
    const int nv = rs.prim_nv(k), i0 = RI(pitem, k) & 0xFFFF;
    const float x0 = 4.0f * X + 0.5f, y0 = (float)NATIVE_RES - 0.5f - 4.0f * Y;
    const Item *items = reinterpret_cast<const Item *>(&RI(items, 0)) + i0;
    for (int e = 0; e < nv; e++) {
        const float a = items[e].a, b = items[e].b, hw = items[e].g3 + CLASS_EPS_F;
        const float row = a * x0 + b * y0 + items[e].c;
        const float lo = row + r_min(0.0f, 3.0f * a) - r_max(0.0f, 3.0f * b), hi = row + r_max(0.0f, 3.0f * a) - r_min(0.0f, 3.0f * b);
        if (lo > hw || hi < -hw) continue;                                  // block entirely off the carrier line
        segmask |= 1u << e;
    }
    return segmask ? 0xFFFFu : 0u;
}

// fp32 alpha of all 16 samples of the block for the segments in `segmask`, in block-local coordinates: the fp64 edge
// and arclength functions are evaluated once at sample (0, 0) and stepped in fp32, so the absolute error of alpha
// stays below ALPHA_ERR_F (terms are O(1) wherever alpha can be non-zero).  Samples whose stipple bit cannot be
// decided in fp32 are flagged in `amb` and must be redone with lineloop_alpha_masked.
constexpr float ALPHA_ERR_F = 4e-6f;
constexpr float STIPPLE_TOL_F = 1e-3f;
MGX_HD void lineloop_alpha16(const Raster &rs, int k, int X, int Y, uint32_t segmask, float (&alpha)[16], uint32_t &amb) {
    const int nv = rs.prim_nv(k), vo = rs.prim_voff(k), lvo = rs.prim_lvoff(k), stipple = rs.prim_stipple(k);
    const float hw = (float)RD(prad, k);
    const double x0 = 4.0 * X + 0.5, y0 = (double)NATIVE_RES - 0.5 - 4.0 * Y;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int q = 0; q < 16; q++) alpha[q] = 0.0f;
    // the dash pattern twice over, so that bit (n & 31) is the pattern's bit (n & 15); for a solid line the stipple arithmetic below is skipped
    const uint32_t pattern = (uint32_t)stipple | ((uint32_t)stipple << 16);
    for (; segmask; segmask &= segmask - 1) {
        const int i = __builtin_ctz(segmask);
        uint32_t namb = 0, nzero = 0;
        double a, b, c;
        edge_coeffs_of(rs, vo, i, 0, i == nv - 1, a, b, c);
        const double ax = RDV(svx, vo + i), ay = RDV(svy, vo + i), len = RD(elen, lvo + i);
        const double E0 = a * x0 + b * y0 + c, S0 = (x0 - ax) * b - (y0 - ay) * a;
        const float af = (float)a, bf = (float)b, e0 = (float)E0, s0 = (float)S0, s1 = (float)(S0 - len), lenf = (float)len;
        float u0 = 0.0f;
        if (stipple) { const double arc = RD(earc, lvo + i); u0 = (float)(arc - 16.0 * rz_floor(arc * 0.0625)); }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int j = 0; j < 4; j++) {
            const float ej = e0 - bf * (float)j, dj = af * (float)j;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int ii = 0; ii < 4; ii++) {
                const float e = ej + af * (float)ii;
                const float ds = dj + bf * (float)ii;
                const float s = s0 + ds, sb = s1 + ds;                         // arclength from the start / beyond the end
                const float t = r_max(r_max(-s, sb), 0.0f);
                float al = r_clamp01(hw - q_sqrt(e * e + t * t));
                if (stipple) {
                    // no branches: position along the pattern, its fraction within STIPPLE_TOL_F of a whole number = undecidable in fp32 (where
                    // the line has any alpha at all), the pattern's bit as an all-ones / zero mask over alpha
                    const float u = u0 + r_clamp(s, 0.0f, lenf);
                    namb = q_push_sign(namb, (0.5f - STIPPLE_TOL_F) - q_abs(q_fract(u) - 0.5f));
                    nzero = q_push_sign(nzero, al - 1e-30f);
                    al = q_and(al, q_bit_mask(pattern, (uint32_t)u));
                }
                alpha[4 * j + ii] = r_max(alpha[4 * j + ii], al);
            }
        }
        amb |= q_rev16(namb & ~nzero);
    }
}

// Sum of the 4x4 samples of an undecided pixel, fp32 only.  Opaque prims (front to back) claim samples through
// coverage masks; at a translucent line loop the still-unclaimed samples it may touch are blended over the colour of
// the opaque prims below it.  Samples are independent of each other, so whenever a decision for one sample cannot be
// guaranteed to agree with the fp64 painter (within the fp32 margin of an edge, a blended channel within TAU of a
// rounding boundary, an undecidable stipple bit, two line loops on top of each other) that sample is left out of the
// sums and reported in `uncertain`, to be added with pixel_add_exact.  sums = r | g << 12 | b << 24 (each <= 4080).
#ifndef MGX_Q_MAXL
#define MGX_Q_MAXL 1          // undecided opaque primitives remembered under a line loop (more than that under one line: the exact painter)
#endif
// One walk down the pixel's primitives, every coverage function called from ONE place.  A line loop that touches still-unclaimed samples
// becomes PENDING: the walk goes on below it and notes, for the opaque primitives it meets, mask and colour (at most MGX_Q_MAXL of them);
// when the list ends -- or every sample is accounted for -- the pending line's alphas are computed and blended over what lies underneath.
// (Round 3 evaluated the primitives below a line from inside the line's branch -- the coverage code inlined twice -- remembered four of them and
// tested all sixteen samples against every crossing segment first; the rasteriser's 96-register variant pays for every instruction and every
// live value of phase Q in spilled registers, and each of the three cuts made it faster: profiles/r04_raster_phase_q_shrink_ab.txt.)
template <typename M> MGX_HD uint64_t pixel_resolve_fast(const Raster &rs, int X, int Y, M mixed, int base, uint32_t &uncertain) {
    constexpr int MAXL = MGX_Q_MAXL;
    uint32_t remaining = 0xFFFFu, unc = 0;
    int sr = 0, sg = 0, sb = 0;
    int pk = -1;                              // the pending line loop, the samples it claimed, its segments that cross the block
    uint32_t pcov = 0, psegmask = 0, pneed = 0, lunc = 0;
    uint32_t lcov[MAXL]; int lcol[MAXL]; int nlow = 0;
    M m = mixed;
    MGX_RSTAT(0, 1); MGX_RSTAT(1, __builtin_popcountll((uint64_t)mixed));
    while (m && (remaining | pneed)) {
        const int k = mask_top(m);
        m &= ~(M(1) << k);
        const int kind = rs.prim_kind(k);
        if (kind == PR_LINELOOP) {
#ifdef MGX_Q_NO_LINE      // development probe: ... without the line loops
            continue;
#endif
            uint32_t segmask;
            const uint32_t cov = lineloop_touch16(rs, k, X, Y, segmask) & remaining;
            MGX_RSTAT(2, 1);
            if (pk >= 0) {
                // a second line loop below a pending one: the pending line's samples and this one's go to the exact painter
                lunc = 0xFFFFu; unc |= cov;
            } else if (cov) {
                MGX_RSTAT(3, 1); MGX_RSTAT(4, __builtin_popcount(cov)); MGX_RSTAT(5, __builtin_popcount(segmask));
                pk = k; pcov = cov; psegmask = segmask; pneed = cov;
            }
            remaining &= ~cov;
            continue;
        }
        uint32_t punc = 0;
        const uint32_t full = kind == PR_POLY ? poly_coverage16(rs, k, X, Y, punc) : ngon_coverage16(rs, k, X, Y, punc);
        MGX_RSTAT(9, 1);
        const int col = rs.prim_rgb(k);
        if (pneed) {
            // under the pending line: remember this primitive (front to back), its uncertain samples are the line's too
            lunc |= punc & pcov;
            if (full & pneed) {
                if (nlow == MAXL) lunc = 0xFFFFu;
                else { lcov[nlow] = full; lcol[nlow] = col; nlow++; }
                pneed &= ~full;
            }
        }
        uint32_t cov = full & remaining;
        punc &= remaining;
        unc |= punc;
        cov &= ~punc;
        if (cov) {
            const int n = __builtin_popcount(cov);
            sr += n * (col & 0xFF); sg += n * ((col >> 8) & 0xFF); sb += n * ((col >> 16) & 0xFF);
        }
        remaining &= ~(cov | punc);
    }
    if (remaining) {
        const int n = __builtin_popcount(remaining);
        sr += n * (base & 0xFF); sg += n * ((base >> 8) & 0xFF); sb += n * ((base >> 16) & 0xFF);
    }
    if (pk >= 0) {
        const int col = rs.prim_rgb(pk);
        float alpha[16];
        lineloop_alpha16(rs, pk, X, Y, psegmask, alpha, lunc);
        const float lrf = (float)(col & 0xFF), lgf = (float)((col >> 8) & 0xFF), lbf = (float)((col >> 16) & 0xFF);
        constexpr float TAU = 255.0f * ALPHA_ERR_F + 5e-4f;
        // Sixteen samples, no branches: a sample that is not the line's to blend (claimed above it, or already uncertain) takes colour 0 and
        // alpha 0 and adds nothing; one whose blended channel lies within TAU of a rounding boundary is flagged and its share multiplied
        // away (`keep`).  Whole parts of the channels are summed in fp32 (exact: <= 4080).  v = c (1 - a) + (a l + 0.5); floor(v) = v - fract(v).
        static_assert(MAXL == 1, "one remembered primitive under a line loop");
        const uint32_t valid = pcov & ~lunc, under = nlow ? lcov[0] : 0u;
        const int ucol = nlow ? lcol[0] : base;
        float fsr = 0.0f, fsg = 0.0f, fsb = 0.0f;
        uint32_t nunc = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int sidx = 0; sidx < 16; sidx++) {
            const int vm = q_bit_mask(valid, sidx), um = q_bit_mask(under, sidx);
            const int c = ((ucol & um) | (base & ~um)) & vm;
            const float a = q_and(alpha[sidx], vm), om = 1.0f - a;
            const float cr = (float)(c & 0xFF), cg = (float)((c >> 8) & 0xFF), cb = (float)((c >> 16) & 0xFF);
            const float vr = cr * om + (a * lrf + 0.5f), vg = cg * om + (a * lgf + 0.5f), vb = cb * om + (a * lbf + 0.5f);
            const float qr = q_fract(vr), qg = q_fract(vg), qb = q_fract(vb);
            const float margin = (0.5f - TAU) - r_max(r_max(q_abs(qr - 0.5f), q_abs(qg - 0.5f)), q_abs(qb - 0.5f));   // < 0: too close to call
            nunc = q_push_sign(nunc, margin);
            const float keep = margin < 0.0f ? 0.0f : 1.0f;
            fsr += keep * (vr - qr); fsg += keep * (vg - qg); fsb += keep * (vb - qb);
        }
        sr += (int)fsr; sg += (int)fsg; sb += (int)fsb;
        lunc |= q_rev16(nunc);
        unc |= lunc & pcov;
    }
    MGX_RSTAT(6, unc ? 1 : 0); MGX_RSTAT(7, __builtin_popcount(unc));
    uncertain = unc;
    return (uint64_t)sr | ((uint64_t)sg << 12) | ((uint64_t)sb << 24);
}
// cv2 INTER_AREA integer-factor path: saturate_cast<uchar>(sum * (1/16)) = round half to even
MGX_HD int pixel_finish(uint64_t sums) {
    const int sr = (int)(sums & 0xFFF), sg = (int)((sums >> 12) & 0xFFF), sb = (int)((sums >> 24) & 0xFFF);
    const int r = (sr + 7 + ((sr >> 4) & 1)) >> 4, g = (sg + 7 + ((sg >> 4) & 1)) >> 4, b = (sb + 7 + ((sb >> 4) & 1)) >> 4;
    return r | (g << 8) | (b << 16);
}
// add the samples in `uncertain` with the fp64 painter (what k_raster_native and the oracle do for every sample)
template <typename M> MGX_HD uint64_t pixel_add_exact(const Raster &rs, int X, int Y, M mixed, int base, uint64_t sums, uint32_t uncertain) {
    for (; uncertain; uncertain &= uncertain - 1) {
        const int sidx = __builtin_ctz(uncertain);
        const double x = 4.0 * X + (sidx & 3) + 0.5, y = (double)NATIVE_RES - 0.5 - 4.0 * Y - (sidx >> 2);
        const int c = raster_sample(rs, x, y, mixed, base);
        sums += (uint64_t)(c & 0xFF) | ((uint64_t)((c >> 8) & 0xFF) << 12) | ((uint64_t)((c >> 16) & 0xFF) << 24);
    }
    return sums;
}
template <typename M> MGX_HD int pixel_resolve(const Raster &rs, int X, int Y, M mixed, int base) {
    uint32_t unc;
    uint64_t sums = pixel_resolve_fast(rs, X, Y, mixed, base, unc);
    if (unc) sums = pixel_add_exact(rs, X, Y, mixed, base, sums, unc);
    return pixel_finish(sums);
}

}  // namespace mgx

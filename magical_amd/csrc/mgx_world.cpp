// mgx_world.cpp -- entity list -> rigid bodies / shapes / joints / collision pairs / draw list.
// See mgx_world.h for the reference lines each builder follows.
#include "mgx_world.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace mgx {

namespace {

const double PI = 3.14159265358979323846;
const double INF = std::numeric_limits<double>::infinity();

// ---- palette: magical/style.py:28-37 evaluated to RGB8; a GL float colour
// lands in the RGBA8 framebuffer as round(c*255).  tests/ check this table against colorsys.
struct Rgb { int r, g, b; };
const Rgb BASE[4] = {{245, 129, 165}, {195, 208, 130}, {135, 185, 211}, {254, 213, 123}};      // red green blue yellow
const Rgb DARK[4] = {{243, 94, 141}, {183, 198, 105}, {110, 170, 202}, {254, 201, 86}};       // darken_rgb
const Rgb LIGHT2[4] = {{250, 190, 209}, {224, 231, 191}, {194, 219, 233}, {254, 234, 188}};   // lighten_rgb(times=2)
const Rgb GREY = {162, 163, 175}, GREY_DARK = {144, 145, 159}, BACKGROUND = {231, 231, 234};
const Rgb WHITE = {255, 255, 255}, PUPIL = {26, 26, 26};

const double ROBOT_LINE = 0.01, SHAPE_LINE = 0.015;   // style.py:25-27

Vec2 rot(Vec2 v, double a) {   // pymunk Vec2d.rotated
    double c = std::cos(a), s = std::sin(a);
    return {v.x * c - v.y * s, v.x * s + v.y * c};
}
double moment_for_circle(double m, double r_in, double r_out) { return m * 0.5 * (r_in * r_in + r_out * r_out); }
double moment_for_poly(double m, const std::vector<Vec2> &v) {   // cpMomentForPoly, offset 0, radius ignored
    double sum1 = 0, sum2 = 0;
    size_t n = v.size();
    for (size_t i = 0; i < n; i++) {
        Vec2 v1 = v[i], v2 = v[(i + 1) % n];
        double a = v2.x * v1.y - v2.y * v1.x;
        double b = (v1.x * v1.x + v1.y * v1.y) + (v1.x * v2.x + v1.y * v2.y) + (v2.x * v2.x + v2.y * v2.y);
        sum1 += a * b; sum2 += a;
    }
    return (m * sum1) / (6.0 * sum2);
}
std::vector<Vec2> rect_verts(double w, double h) {   // geom.py:101-108
    return {{w / 2, h / 2}, {-w / 2, h / 2}, {-w / 2, -h / 2}, {w / 2, -h / 2}};
}
std::vector<Vec2> draw_rect(double w, double h) {    // gym_render.py:449-453
    return {{-w / 2, h / 2}, {w / 2, h / 2}, {w / 2, -h / 2}, {-w / 2, -h / 2}};
}
std::vector<Vec2> regular_poly(int n, double side) { // geom.py:13-15,35-46
    double radius = side / (2 * std::sin(PI / n));
    std::vector<Vec2> out;
    for (int k = 0; k < n; k++) out.push_back(rot({0, radius}, k * (2 * PI / n)));
    return out;
}
double area_equiv_side(int n, double rad) {          // geom.py:18-22
    double p_n = PI / n;
    return 2 * rad * std::sqrt(p_n * std::tan(p_n));
}
std::vector<Vec2> star_verts(int n, double r_out, double r_in) {   // geom.py:49-63
    std::vector<Vec2> out;
    for (int k = 0; k < n; k++) {
        out.push_back(rot({0, r_out}, k * 2 * PI / n));
        out.push_back(rot({0, r_in}, (2 * k + 1) * PI / n));
    }
    return out;
}
// Convex parts of a star.  The reference asks Chipmunk's autogeometry (entities.py:653-654), whose
// partition is not observable here; we use 5 tip triangles + the inner pentagon (same union).
std::vector<std::vector<Vec2>> star_parts(const std::vector<Vec2> &sv) {
    int n = (int)sv.size() / 2;
    std::vector<std::vector<Vec2>> parts;
    for (int k = 0; k < n; k++) parts.push_back({sv[2 * ((k - 1 + n) % n) + 1], sv[2 * k], sv[2 * k + 1]});
    std::vector<Vec2> inner;
    for (int k = 0; k < n; k++) inner.push_back(sv[2 * k + 1]);
    parts.push_back(inner);
    return parts;
}
// entities.py:193-214
void finger_vertices(double upper_len, double fore_len, double thick, double side,
                     std::vector<Vec2> &upper, std::vector<Vec2> &fore) {
    double up_shift = upper_len / 2;
    upper = rect_verts(thick, upper_len);
    fore = rect_verts(thick, fore_len);
    Vec2 upper_start = {side * thick / 2, upper_len / 2};
    Vec2 off = rot({-side * thick / 2, fore_len / 2}, side * PI / 8);
    Vec2 trans = {upper_start.x + off.x, upper_start.y + off.y + up_shift};
    for (auto &v : fore) { Vec2 r = rot(v, side * PI / 8); v = {r.x + trans.x, r.y + trans.y}; }
    for (auto &v : upper) v.y += up_shift;
}

JointDef joint(int kind, int a, int b) {
    JointDef j{};
    j.kind = kind; j.a = a; j.b = b;
    j.error_bias = std::pow(1.0 - 0.1, 60.0);   // Chipmunk default
    j.max_bias = INF; j.max_force = INF;
    return j;
}
PrimDef prim(int kind, Rgb c, int xform, int body) {
    PrimDef p{};
    p.kind = kind; p.xform = xform; p.body = body; p.eye_body = -1;
    p.rgb[0] = c.r; p.rgb[1] = c.g; p.rgb[2] = c.b;
    return p;
}

}  // namespace

int palette_rgb(int colour, int role) {
    const Rgb c = role == 0 ? DARK[colour] : (role == 1 ? BASE[colour] : LIGHT2[colour]);
    return c.r | (c.g << 8) | (c.b << 16);
}


// ---------------------------------------------------------------- placement queries (host only)
namespace {
// (fixed capacity, no allocation: a reset with per-env worlds runs millions of these queries)
constexpr int WS_MAXV = 16, WS_MAXS = 8;          // vertices per shape (finalize() checks), shapes per entity (robot 5, star 6)
struct VList {
    Vec2 d[WS_MAXV]; int n = 0;
    size_t size() const { return (size_t)n; }
    const Vec2 &operator[](size_t i) const { return d[i]; }
    const Vec2 *begin() const { return d; }
    const Vec2 *end() const { return d + n; }
    void push_back(Vec2 p) { d[n++] = p; }
};
struct WShape {       // world-space: circle centre / polygon verts / segment ends, and the box around it (radius included)
    int kind; double r; VList v; int group; double lox, loy, hix, hiy;
    void bound() {
        lox = loy = 1e300; hix = hiy = -1e300;
        for (auto &p : v) { lox = std::min(lox, p.x); hix = std::max(hix, p.x); loy = std::min(loy, p.y); hiy = std::max(hiy, p.y); }
        lox -= r; loy -= r; hix += r; hiy += r;
    }
};
// shapes whose boxes are further apart than round-off could explain cannot touch (cpCollide needs distance <= r_a + r_b)
inline bool boxes_apart(const WShape &a, const WShape &b) {
    const double s = 1e-9;
    return a.lox > b.hix + s || b.lox > a.hix + s || a.loy > b.hiy + s || b.loy > a.hiy + s;
}

double pt_seg_dist2(Vec2 p, Vec2 a, Vec2 b) {
    double dx = b.x - a.x, dy = b.y - a.y, l2 = dx * dx + dy * dy;
    double t = l2 > 0 ? ((p.x - a.x) * dx + (p.y - a.y) * dy) / l2 : 0.0;
    t = t < 0 ? 0 : (t > 1 ? 1 : t);
    double qx = a.x + t * dx - p.x, qy = a.y + t * dy - p.y;
    return qx * qx + qy * qy;
}
// signed distance of a point to a convex CCW polygon (negative inside)
double pt_poly_dist(Vec2 p, const VList &v) {
    bool inside = true; double best = 1e300; size_t n = v.size();
    for (size_t i = 0; i < n; i++) {
        Vec2 a = v[i], b = v[(i + 1) % n];
        double cr = (b.x - a.x) * (p.y - a.y) - (b.y - a.y) * (p.x - a.x);
        if (cr < 0) inside = false;
        double d2 = pt_seg_dist2(p, a, b);
        if (d2 < best) best = d2;
    }
    return inside ? -std::sqrt(best) : std::sqrt(best);
}
// distance between two convex CCW polygons: negative when they overlap, else the exact gap
double poly_poly_dist(const VList &A, const VList &B) {
    auto separated = [](const VList &P, const VList &Q) {
        size_t n = P.size();
        for (size_t i = 0; i < n; i++) {
            Vec2 a = P[i], b = P[(i + 1) % n];
            double nx = b.y - a.y, ny = -(b.x - a.x);                 // outward normal of a CCW polygon
            bool all_out = true;
            for (auto &q : Q) if ((q.x - a.x) * nx + (q.y - a.y) * ny <= 0) { all_out = false; break; }
            if (all_out) return true;
        }
        return false;
    };
    if (!separated(A, B) && !separated(B, A)) return -1.0;
    double best = 1e300;
    for (auto &p : A) for (size_t i = 0; i < B.size(); i++) best = std::min(best, pt_seg_dist2(p, B[i], B[(i + 1) % B.size()]));
    for (auto &p : B) for (size_t i = 0; i < A.size(); i++) best = std::min(best, pt_seg_dist2(p, A[i], A[(i + 1) % A.size()]));
    return std::sqrt(best);
}
// cpCollide(a, b).count > 0 (cpCollision.c: CircleToCircle / CircleToSegment use distsq < mindist^2, the GJK cases d <= mindist)
bool shapes_touch(const WShape &a, const WShape &b) {
    if (a.kind > b.kind) return shapes_touch(b, a);
    double rr = a.r + b.r;
    if (a.kind == SH_CIRCLE && b.kind == SH_CIRCLE) {
        double dx = b.v[0].x - a.v[0].x, dy = b.v[0].y - a.v[0].y;
        return dx * dx + dy * dy < rr * rr;
    }
    if (a.kind == SH_CIRCLE && b.kind == SH_SEGMENT) return pt_seg_dist2(a.v[0], b.v[0], b.v[1]) < rr * rr;
    if (a.kind == SH_CIRCLE && b.kind == SH_POLY) return pt_poly_dist(a.v[0], b.v) <= rr;
    if (a.kind == SH_SEGMENT && b.kind == SH_POLY) {
        double best = 1e300;
        for (auto &p : b.v) best = std::min(best, pt_seg_dist2(p, a.v[0], a.v[1]));
        for (int k = 0; k < 2; k++) { double d = pt_poly_dist(a.v[k], b.v); if (d <= 0) return true; best = std::min(best, d * d); }
        return std::sqrt(best) - rr <= 0.0;
    }
    if (a.kind == SH_POLY && b.kind == SH_POLY) return poly_poly_dist(a.v, b.v) - rr <= 0.0;
    return false;   // segment-segment: both static
}
// world-space shapes of entity e of world W with the entity at (x, y, a) (bodies follow the entity rigidly, as in finalize());
// hw: this env's (h, w) of a goal region, or NULL.  Returns how many (<= WS_MAXS)
int world_shapes_of(const World &W, int e, double x, double y, double a, const double *hw_of_e, WShape *out) {
    const EntityDef &E = W.entities[e];
    if (!E.enabled) return 0;
    if (E.kind == 2) {                      // goal sensor: box (w, h) around its centre, never rotated
        double hw = (hw_of_e ? hw_of_e[1] : E.w) / 2, hh = (hw_of_e ? hw_of_e[0] : E.h) / 2;
        WShape &G = out[0];
        G.kind = SH_POLY; G.r = 0.0; G.group = 0; G.v.n = 0;
        G.v.push_back({x - hw, y - hh}); G.v.push_back({x + hw, y - hh}); G.v.push_back({x + hw, y + hh}); G.v.push_back({x - hw, y + hh});
        G.bound();
        return 1;
    }
    int n = 0;
    for (int si : E.shapes) {
        if (n == WS_MAXS) break;
        const ShapeDef &S = W.shapes[si];
        const BodyDef &B = W.bodies[S.body];
        double bx = x, by = y, ba = a + B.aoff;
        if (B.parent >= 0) { Vec2 r = rot({B.ax, B.ay}, a + W.bodies[B.parent].aoff); bx = x + r.x; by = y + r.y; }
        WShape &O = out[n++];
        O.kind = S.kind; O.r = S.radius; O.group = S.group; O.v.n = 0;
        if (S.kind == SH_CIRCLE) O.v.push_back({bx, by});
        else for (auto &lv : S.verts) { Vec2 r = rot(lv, ba); O.v.push_back({bx + r.x, by + r.y}); }
        O.bound();
    }
    return n;
}
int wall_shapes_of(const World &W, WShape *out) {
    int n = 0;
    for (const ShapeDef &S : W.shapes) if (S.kind == SH_SEGMENT && n < WS_MAXS) {
        WShape &O = out[n++];
        O.kind = SH_SEGMENT; O.r = S.radius; O.group = 0; O.v.n = 0;
        for (auto &p : S.verts) O.v.push_back(p);
        O.bound();
    }
    return n;
}
inline bool any_touch(const WShape *A, int na, const WShape *B, int nb, bool groups) {
    for (int i = 0; i < na; i++) for (int j = 0; j < nb; j++) {
        if (groups && A[i].group != 0 && A[i].group == B[j].group) continue;       // ShapeFilter groups (cpShapeFilterReject)
        if (boxes_apart(A[i], B[j])) continue;
        if (shapes_touch(A[i], B[j])) return true;
    }
    return false;
}
}  // namespace

bool World::placement_collides(int ent, const double *poses, const uint8_t *enabled, const double *ent_hw) const {
    WShape mine[WS_MAXS], walls[WS_MAXS], theirs[WS_MAXS];
    const int n_mine = world_shapes_of(*this, ent, poses[3 * ent], poses[3 * ent + 1], poses[3 * ent + 2], ent_hw ? ent_hw + 2 * ent : nullptr, mine);
    const int n_walls = wall_shapes_of(*this, walls);
    if (any_touch(mine, n_mine, walls, n_walls, false)) return true;
    for (int e = 0; e < (int)entities.size(); e++) {
        if (e == ent || !enabled[e]) continue;
        const int n_theirs = world_shapes_of(*this, e, poses[3 * e], poses[3 * e + 1], poses[3 * e + 2], ent_hw ? ent_hw + 2 * e : nullptr, theirs);
        if (any_touch(mine, n_mine, theirs, n_theirs, true)) return true;
    }
    return false;
}

// ---------------------------------------------------------------- np.random.RandomState's generator (host only)
namespace {
// MT19937 exactly as numpy's legacy RandomState runs it (randomkit / mt19937.c): same tempering, same refill, and
// random_sample() = (a >> 5, b >> 6) -> (a * 2^26 + b) / 2^53; uniform(lo, hi) = lo + (hi - lo) * random_sample()
struct Mt19937 {
    uint32_t *key; int *pos;
    void refill() {
        constexpr uint32_t N = 624, M = 397, MATRIX_A = 0x9908b0dfu, UPPER = 0x80000000u, LOWER = 0x7fffffffu;
        uint32_t y; uint32_t i;
        for (i = 0; i < N - M; i++) { y = (key[i] & UPPER) | (key[i + 1] & LOWER); key[i] = key[i + M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A); }
        for (; i < N - 1; i++) { y = (key[i] & UPPER) | (key[i + 1] & LOWER); key[i] = key[i + (M - N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A); }
        y = (key[N - 1] & UPPER) | (key[0] & LOWER);
        key[N - 1] = key[M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
        *pos = 0;
    }
    uint32_t next32() {
        if (*pos == 624) refill();
        uint32_t y = key[(*pos)++];
        y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
        return y;
    }
    double next_double() { int32_t a = next32() >> 5, b = next32() >> 6; return (a * 67108864.0 + b) / 9007199254740992.0; }
    double uniform(double lo, double hi) { return lo + (hi - lo) * next_double(); }
};
}  // namespace

// ---- the other draws of the tasks' on_reset() (counts, colours, shape types, sizes), for many envs per call.  numpy's legacy
// generator: randint(lo, hi) / choice / shuffle all come down to a masked rejection on 32-bit outputs (max = hi - lo - 1:
// mask = next power of two minus one, redraw while (next32() & mask) > max; max == 0 draws nothing), shuffle(list of n) swaps
// x[i], x[random_interval(i)] for i = n - 1 .. 1, random_sample() is next_double().
void rng_bounded(uint32_t *key, int *pos, int count, uint32_t max_inclusive, int32_t *out) {
    Mt19937 rng{key, pos};
    uint32_t mask = max_inclusive;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    for (int i = 0; i < count; i++) {
        uint32_t v = 0;
        if (max_inclusive != 0) { do { v = rng.next32() & mask; } while (v > max_inclusive); }
        out[i] = (int32_t)v;
    }
}
void rng_doubles(uint32_t *key, int *pos, int count, double *out) {
    Mt19937 rng{key, pos};
    for (int i = 0; i < count; i++) out[i] = rng.next_double();
}
void rng_shuffle(uint32_t *key, int *pos, int n, int32_t *perm) {
    for (int i = 0; i < n; i++) perm[i] = i;
    for (int i = n - 1; i >= 1; i--) {
        int32_t j;
        rng_bounded(key, pos, 1, (uint32_t)i, &j);
        const int32_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
}

int World::randomise_all_poses(double *poses, const int *ents, int n, const uint8_t *ignore, const double arena[4],
                               const uint8_t *rand_pos, const uint8_t *rand_rot, const double *pos_limits, const double *rot_limits,
                               uint32_t *mt_key, int *mt_pos, const double *ent_hw) const {
    Mt19937 rng{mt_key, mt_pos};
    const int ne = (int)entities.size(), max_retries = 10, max_tries = 10000;
    int rejected = 0;
    // World-space shapes of the entities that stay where they are during an attempt: worked out when an entity is first asked
    // about (the ones this call does not move) or when its pose is accepted (the ones it does), not once per attempt -- only the
    // entity being placed moves between attempts.  Same tests on the same numbers as placement_collides(), so the same draws.
    static thread_local std::vector<WShape> held; static thread_local std::vector<int> n_held; static thread_local std::vector<uint8_t> enabled;
    held.resize((size_t)ne * WS_MAXS); n_held.assign(ne, -1);
    WShape walls[WS_MAXS], mine[WS_MAXS];
    const int n_walls = wall_shapes_of(*this, walls);
    auto hw_of = [&](int e) { return ent_hw ? ent_hw + 2 * e : nullptr; };
    for (int retry = 0; retry < max_retries; retry++) {
        enabled.assign(ne, 1);
        for (int i = 0; i < n; i++) { enabled[ents[i]] = 0; n_held[ents[i]] = -1; }       // categories = 0 until its turn
        for (int e = 0; e < ne; e++) if (ignore && ignore[e]) enabled[e] = 0;
        bool failed = false;
        for (int i = 0; i < n && !failed; i++) {
            const int e = ents[i];
            if (!entities[e].enabled) continue;                            // not in this episode's world: nothing to place, no draws
            enabled[e] = 1;
            // ---- pm_randomise_pose (geom.py:116-262)
            const double ox = poses[3 * e], oy = poses[3 * e + 1], oa = poses[3 * e + 2];
            double x0 = arena[0], x1 = arena[1], y0 = arena[2], y1 = arena[3], r0 = -PI, r1 = PI;
            if (pos_limits[i] >= 0) {
                x0 = std::max(arena[0], ox - pos_limits[i]); x1 = std::min(arena[1], ox + pos_limits[i]);
                y0 = std::max(arena[2], oy - pos_limits[i]); y1 = std::min(arena[3], oy + pos_limits[i]);
            }
            if (rot_limits[i] >= 0) { r0 = oa - rot_limits[i]; r1 = oa + rot_limits[i]; }
            int n_tries = 0, n_mine = 0;
            for (; n_tries < max_tries; n_tries++) {
                if (rand_pos[i]) { poses[3 * e] = rng.uniform(x0, x1); poses[3 * e + 1] = rng.uniform(y0, y1); }
                if (rand_rot[i]) poses[3 * e + 2] = rng.uniform(r0, r1);
                n_mine = world_shapes_of(*this, e, poses[3 * e], poses[3 * e + 1], poses[3 * e + 2], hw_of(e), mine);
                bool hit = any_touch(mine, n_mine, walls, n_walls, false);
                for (int o = 0; o < ne && !hit; o++) {
                    if (o == e || !enabled[o]) continue;
                    if (n_held[o] < 0) n_held[o] = world_shapes_of(*this, o, poses[3 * o], poses[3 * o + 1], poses[3 * o + 2], hw_of(o), &held[(size_t)o * WS_MAXS]);
                    hit = any_touch(mine, n_mine, &held[(size_t)o * WS_MAXS], n_held[o], true);
                }
                if (!hit) break;
            }
            rejected += n_tries;
            if (n_tries == max_tries) {                                    // PlacementError: put it back, start over
                poses[3 * e] = ox; poses[3 * e + 1] = oy; poses[3 * e + 2] = oa;
                failed = true;
            } else {
                for (int k = 0; k < n_mine; k++) held[(size_t)e * WS_MAXS + k] = mine[k];
                n_held[e] = n_mine;
            }
        }
        if (!failed) return rejected;
    }
    return -1;
}

int World::finalize(int max_steps, std::string &err) {
    if (finalized) { err = "world already finalized"; return -3; }
    max_episode_steps = max_steps;
    const double *pv = phys_vars;
    {   // (no regrowth of the tables while they fill: a reset with per-env worlds finalizes thousands of worlds)
        const size_t n = entities.size();
        bodies.reserve(8 + n); shapes.reserve(16 + 6 * n); joints.reserve(12 + 2 * n); prims.reserve(16 + 2 * n);
        island_j.reserve(n); joint_acc_off.reserve(12 + 2 * n);
    }
    // body 0: every static body (space.static_body, the arena body, goal bodies) in one world frame
    bodies.push_back({BODY_STATIC, 0, 0, 0, 0, 0, -1, 0, 0, 0});

    // ---- arena (entities.py:502-537): added first by BaseEnv.reset (base_env.py:213)
    {
        double l = -1, r = 1, t = 1, b = -1, rad = 1;
        Vec2 pts[4] = {{l - rad, t + rad}, {r + rad, t + rad}, {r + rad, b - rad}, {l - rad, b - rad}};
        for (int i = 0; i < 4; i++) {
            ShapeDef s{SH_SEGMENT, 0, rad, 0.8, 0, -1, {pts[i], pts[(i + 1) % 4]}};
            shapes.push_back(std::move(s));
        }
        PrimDef q = prim(PR_POLY, WHITE, XF_WORLD, 0);
        q.verts = draw_rect(r - l, t - b);
        prims.push_back(std::move(q));
        // PolyLine: attrs enabled in reverse add order (gym_render.py:306-311) -> glLineWidth(1) wins
        PrimDef ln = prim(PR_LINELOOP, GREY, XF_WORLD, 0);
        ln.verts = draw_rect(r - l, t - b); ln.line_width = 1.0;
        prims.push_back(std::move(ln));
    }

    int n_blocks = 0, n_goals = 0;
    for (size_t ei = 0; ei < entities.size(); ei++) {
        EntityDef &e = entities[ei];
        if (e.kind == 0) {
            // ---------------- Robot (entities.py:238-437)
            if (robot_body >= 0) { err = "only one robot per world"; return -1; }
            double radius = ROBOT_RAD, mass = ROBOT_MASS;
            int body = (int)bodies.size();
            robot_body = body; e.body = body;
            bodies.push_back({BODY_DYNAMIC, 1.0 / mass, 1.0 / moment_for_circle(mass, 0, radius), e.x, e.y, e.angle, -1, 0, 0, 0x1FF, (int)ei, 0.0});
            control_body = (int)bodies.size();
            bodies.push_back({BODY_KINEMATIC, 0, 0, e.x, e.y, e.angle, -1, 0, 0, 0, (int)ei, 0.0});
            robot_j0 = (int)joints.size();
            JointDef pj = joint(J_PIVOT, control_body, body);              // :255-258
            pj.max_bias = 0; pj.max_force = pv[0]; pj.pv = 0;
            joints.push_back(pj);
            JointDef gj = joint(J_GEAR, control_body, body);               // :259-263
            gj.p0 = 0.0; gj.p1 = 1.0; gj.error_bias = 0.0; gj.max_bias = 2.5; gj.max_force = pv[1]; gj.pv = 1;
            joints.push_back(gj);
            int eye_bodies[2];
            for (int k = 0; k < 2; k++) {                                  // :267-277
                double em = mass / 10;
                eye_bodies[k] = (int)bodies.size();
                eye_body[k] = eye_bodies[k];
                bodies.push_back({BODY_DYNAMIC, 1.0 / em, 1.0 / moment_for_circle(em, 0, radius), 0, 0, e.angle, -1, 0, 0,
                                  (1 << 2) | (1 << 5), (int)ei, 0.0});
                JointDef sj = joint(J_SPRING, body, eye_bodies[k]);
                sj.p0 = 0; sj.p1 = 0.1; sj.p2 = 3e-3;
                joints.push_back(sj);
            }
            double thick = 0.25 * radius, upper_len = 1.1 * radius, fore_len = 0.7 * radius;   // :280-282
            std::vector<Vec2> f_upper[2], f_fore[2], fi_upper[2], fi_fore[2];
            for (int k = 0; k < 2; k++) {                                  // :288-354
                double side = k == 0 ? -1.0 : 1.0;
                finger_vertices(upper_len, fore_len, thick, side, f_upper[k], f_fore[k]);
                finger_vertices(upper_len - ROBOT_LINE * 2, fore_len - ROBOT_LINE * 2, thick - ROBOT_LINE * 2, side,
                                fi_upper[k], fi_fore[k]);
                for (auto &v : fi_upper[k]) v.y += ROBOT_LINE;
                for (auto &v : fi_fore[k]) v.y += ROBOT_LINE;
                double lim_outer = PI / 8, lim_inner = 0.0;
                double lo = side < 0 ? -lim_inner : -lim_outer;
                double hi = side < 0 ? lim_outer : lim_inner;
                double fm = mass / 8;
                std::vector<Vec2> cat = f_upper[k];
                cat.insert(cat.end(), f_fore[k].begin(), f_fore[k].end());
                double fi = moment_for_poly(fm, cat);                      // 8-vertex concatenation (:314-315)
                double delta = side < 0 ? hi : lo;
                Vec2 rel = {side * radius * 0.45, radius * 0.1};
                Vec2 rr = rot(rel, e.angle);
                int fb = (int)bodies.size();
                finger_body[k] = fb;
                bodies.push_back({BODY_DYNAMIC, 1.0 / fm, 1.0 / fi, e.x + rr.x, e.y + rr.y, e.angle + delta, body, rel.x, rel.y, 0x1FF, (int)ei, delta});
                JointDef pin = joint(J_PIN, body, fb);                     // :334-341
                pin.ax = rel.x; pin.ay = rel.y; pin.bx = 0; pin.by = 0; pin.error_bias = 0.0;
                pin.p0 = 0.0;   // anchors coincide at construction -> rest length exactly 0
                joints.push_back(pin);
                JointDef lim = joint(J_LIMIT, body, fb);                   // :343-346
                lim.p0 = lo; lim.p1 = hi; lim.error_bias = 0.0;
                joints.push_back(lim);
                JointDef mot = joint(J_MOTOR, body, fb);                   // :349-354
                mot.max_bias = 0.0; mot.max_force = pv[2]; mot.pv = 2;
                motor_joint[k] = (int)joints.size();
                joints.push_back(mot);
            }
            int robot_group = 1;                                           // :358-375
            e.shapes.push_back((int)shapes.size());
            shapes.push_back({SH_CIRCLE, body, radius, 0.5, robot_group, (int)ei, {}});
            for (int k = 0; k < 2; k++) {
                e.shapes.push_back((int)shapes.size());
                shapes.push_back({SH_POLY, finger_body[k], 0.0, 5.0, robot_group, (int)ei, f_upper[k]});
                e.shapes.push_back((int)shapes.size());
                shapes.push_back({SH_POLY, finger_body[k], 0.0, 5.0, robot_group, (int)ei, f_fore[k]});
            }
            // graphics (:377-437): finger outers, finger inners, then the body compound
            for (int k = 0; k < 2; k++) {
                PrimDef a = prim(PR_POLY, GREY, XF_BODY, finger_body[k]); a.verts = f_upper[k]; prims.push_back(std::move(a));
                PrimDef b = prim(PR_POLY, GREY, XF_BODY, finger_body[k]); b.verts = f_fore[k]; prims.push_back(std::move(b));
            }
            for (int k = 0; k < 2; k++) {
                PrimDef a = prim(PR_POLY, BACKGROUND, XF_BODY, finger_body[k]); a.verts = fi_upper[k]; prims.push_back(std::move(a));
                PrimDef b = prim(PR_POLY, BACKGROUND, XF_BODY, finger_body[k]); b.verts = fi_fore[k]; prims.push_back(std::move(b));
            }
            PrimDef co = prim(PR_NGON, GREY_DARK, XF_BODY, body); co.ngon = 100; co.radius = radius; prims.push_back(std::move(co));
            PrimDef ci = prim(PR_NGON, GREY, XF_BODY, body); ci.ngon = 100; ci.radius = radius - ROBOT_LINE; prims.push_back(std::move(ci));
            for (int k = 0; k < 2; k++) {
                double xs = k == 0 ? -1.0 : 1.0;
                PrimDef eye = prim(PR_NGON, WHITE, XF_EYE, body);
                eye.ngon = 20; eye.radius = 0.2 * radius; eye.eye_base[0] = xs * 0.4 * radius; eye.eye_base[1] = 0.3 * radius;
                prims.push_back(std::move(eye));
                PrimDef pup = prim(PR_NGON, PUPIL, XF_EYE, body);
                pup.ngon = 10; pup.radius = 0.12 * radius; pup.eye_base[0] = eye.eye_base[0]; pup.eye_base[1] = eye.eye_base[1];
                pup.eye_body = eye_bodies[k]; pup.eye_pre[0] = 0; pup.eye_pre[1] = radius * 0.07;
                prims.push_back(std::move(pup));
            }
        } else if (e.kind == 1 && !e.enabled) {
            // not part of this env's episode: an inert body that only keeps body indices and state rows aligned
            e.body = (int)bodies.size();
            bodies.push_back({BODY_STATIC, 0, 0, e.x, e.y, e.angle, -1, 0, 0, 0x1FF, (int)ei, 0.0});
        } else if (e.kind == 1) {
            // ---------------- Shape (entities.py:614-757)
            n_blocks++;
            double mass = SHAPE_MASS, size = SHAPE_RAD;
            int body = (int)bodies.size();
            e.body = body;
            // a block's geometry depends on its shape type alone: worked out once per thread and type (a reset with per-env worlds
            // finalizes thousands of worlds per thread, and the trigonometry and the star's decomposition were a third of that)
            struct ShapeGeom { bool have = false, circle = false; double inertia = 0, poly_radius = 0; std::vector<std::vector<Vec2>> phys_parts, draw_outer, draw_inner; };
            static thread_local ShapeGeom geom_cache[7];
            if (e.shape_type < 0 || e.shape_type > 6) { err = "bad shape type"; return -1; }
            ShapeGeom &G = geom_cache[e.shape_type];
            if (!G.have) {
                std::vector<std::vector<Vec2>> phys_parts, draw_outer, draw_inner;
                double inertia = 0, poly_radius = 0;
                bool circle = false;
                if (e.shape_type == 1) {                                       // SQUARE :620-635
                    double side = std::sqrt(PI) * size, hw = side / 2;
                    std::vector<Vec2> box = {{hw, -hw}, {hw, hw}, {-hw, hw}, {-hw, -hw}};   // cpBoxShapeInit2 order
                    inertia = mass * moment_for_poly(1.0, box);               // mass comes from shape.mass
                    poly_radius = 0.01 * side;
                    phys_parts = {box};
                    draw_outer = {draw_rect(side, side)};
                    draw_inner = {draw_rect(side - 2 * SHAPE_LINE, side - 2 * SHAPE_LINE)};
                } else if (e.shape_type == 5) {                                // CIRCLE :636-645
                    inertia = moment_for_circle(mass, 0, size);
                    circle = true;
                } else if (e.shape_type == 6) {                                // STAR :646-668
                    double r_out = 1.3 * size, r_in = 0.5 * r_out;
                    std::vector<Vec2> sv = star_verts(5, r_out, r_in);
                    std::vector<Vec2> hull;                                    // to_convex_hull -> the 5 tips
                    for (int k = 0; k < 5; k++) hull.push_back(sv[2 * k]);
                    inertia = moment_for_poly(mass, hull);
                    phys_parts = star_parts(sv);
                    draw_outer = phys_parts;
                    draw_inner = star_parts(star_verts(5, r_out - SHAPE_LINE, r_in - SHAPE_LINE));
                } else {                                                       // regular polygons :669-697
                    int ns; double factor = 1.0;
                    switch (e.shape_type) {
                        case 0: ns = 3; factor = 0.8; break;
                        case 2: ns = 5; break;
                        case 3: ns = 6; break;
                        case 4: ns = 8; break;
                        default: err = "bad shape type"; return -1;
                    }
                    double side = factor * area_equiv_side(ns, size);
                    std::vector<Vec2> pv_ = regular_poly(ns, side);
                    inertia = moment_for_poly(mass, pv_);
                    phys_parts = {pv_};
                    double apothem = side / (2 * std::tan(PI / ns));
                    double short_side = 2 * (apothem - SHAPE_LINE) * std::tan(PI / ns);
                    draw_outer = {pv_};
                    draw_inner = {regular_poly(ns, short_side)};
                }
                G.have = true; G.circle = circle; G.inertia = inertia; G.poly_radius = poly_radius;
                G.phys_parts = std::move(phys_parts); G.draw_outer = std::move(draw_outer); G.draw_inner = std::move(draw_inner);
            }
            const std::vector<std::vector<Vec2>> &phys_parts = G.phys_parts, &draw_outer = G.draw_outer, &draw_inner = G.draw_inner;
            const double inertia = G.inertia, poly_radius = G.poly_radius;
            const bool circle = G.circle;
            int group = 0;
            if (e.shape_type == 6) group = ++group_ctr;                    // generate_group_id :60-66
            Rgb col = BASE[e.colour], dark = DARK[e.colour];
            bodies.push_back({BODY_DYNAMIC, 1.0 / mass, 1.0 / inertia, e.x, e.y, e.angle, -1, 0, 0, 0x1FF, (int)ei, 0.0});
            if (circle) {
                e.shapes.push_back((int)shapes.size());
                shapes.push_back({SH_CIRCLE, body, size, 0.5, 0, (int)ei, {}});
            } else {
                for (auto &part : phys_parts) {
                    e.shapes.push_back((int)shapes.size());
                    shapes.push_back({SH_POLY, body, poly_radius, 0.5, group, (int)ei, part});
                }
            }
            island_j.push_back((int)joints.size());
            JointDef tj = joint(J_PIVOT, 0, body);                         // :703-707
            tj.max_bias = 0; tj.max_force = pv[3]; tj.pv = 3;
            joints.push_back(tj);
            JointDef rj = joint(J_GEAR, 0, body);                          // :708-711
            rj.p0 = 0.0; rj.p1 = 1.0; rj.max_bias = 0; rj.max_force = pv[4]; rj.pv = 4;
            joints.push_back(rj);
            if (circle) {
                PrimDef o = prim(PR_NGON, dark, XF_BODY, body); o.ngon = 100; o.radius = size; o.ent = (int)ei; o.role = 0; prims.push_back(std::move(o));
                PrimDef i = prim(PR_NGON, col, XF_BODY, body); i.ngon = 100; i.radius = size - SHAPE_LINE; i.ent = (int)ei; i.role = 1; prims.push_back(std::move(i));
            } else {
                // the convex parts of one compound are painted back to back in one opaque colour (entities.py:750-757), so
                // each compound is a single multi-part primitive: the union of its parts
                auto compound = [&](const std::vector<std::vector<Vec2>> &geoms, Rgb c, int role) {
                    PrimDef q = prim(PR_POLY, c, XF_BODY, body);
                    for (auto &g : geoms) { q.verts.insert(q.verts.end(), g.begin(), g.end()); q.parts.push_back((int)g.size()); }
                    q.ent = (int)ei; q.role = role;
                    prims.push_back(std::move(q));
                };
                compound(draw_outer, dark, 0);
                compound(draw_inner, col, 1);
            }
        } else {
            // ---------------- GoalRegion (entities.py:790-819): static sensor, drawn only
            e.body = -1;
            if (!e.enabled) { n_goals++; continue; }      // keeps the later regions' ordinals
            double cx = e.x + e.w / 2, cy = e.y - e.h / 2;
            std::vector<Vec2> rect = draw_rect(e.w, e.h);
            for (auto &v : rect) { v.x += cx; v.y += cy; }
            const int goal_ord = n_goals++;
            PrimDef fill = prim(PR_POLY, LIGHT2[e.colour], XF_WORLD, 0); fill.verts = rect; fill.ent = (int)ei; fill.role = 2; fill.goal = goal_ord; prims.push_back(std::move(fill));
            PrimDef outl = prim(PR_LINELOOP, BASE[e.colour], XF_WORLD, 0);
            outl.verts = rect; outl.line_width = 2.5; outl.stipple = 0x00FF; outl.ent = (int)ei; outl.role = 1; outl.goal = goal_ord;
            prims.push_back(std::move(outl));
        }
    }
    if (robot_body < 0) { err = "world has no robot"; return -1; }

    // ---- filtered candidate pairs (cpSpaceCollideShapes QueryReject minus the BB test)
    pairs.reserve(shapes.size() * (shapes.size() - 1) / 2);
    state_map.reserve(bodies.size() * 9);
    for (int i = 0; i < (int)shapes.size(); i++)
        for (int j = i + 1; j < (int)shapes.size(); j++) {
            const ShapeDef &a = shapes[i], &b = shapes[j];
            if (bodies[a.body].type == BODY_STATIC && bodies[b.body].type == BODY_STATIC) continue;
            if (a.body == b.body) continue;
            if (a.group != 0 && a.group == b.group) continue;
            int sa = i, sb = j;
            if (shapes[sa].kind > shapes[sb].kind) { sa = j; sb = i; }   // cpCollide type ordering
            pairs.push_back({sa, sb});
        }
    // ---- persistent state rows
    // comp | body<<4 | row-within-blob<<12; pose components (x y a) go to the pose blob, the rest to the
    // velocity blob
    n_state_p = 0;
    {
        int row_p = 0, row_v = 0;
        for (int b = 0; b < (int)bodies.size(); b++)
            for (int c = 0; c < 9; c++)
                if (bodies[b].state_mask & (1 << c)) {
                    int row = c < 3 ? row_p++ : row_v++;
                    state_map.push_back(c | (b << 4) | (row << 12));
                }
        n_state_p = row_p;
    }
    n_jacc = 0;
    for (auto &j : joints) {
        joint_acc_off.push_back(n_jacc);
        n_jacc += (j.kind == J_PIVOT) ? 2 : (j.kind == J_SPRING ? 0 : 1);
    }
    cache_slots = 10 + 3 * n_blocks;
    if (cache_slots > 60) cache_slots = 60;
    max_contacts = 2 * cache_slots;
    max_overlaps = 2 * cache_slots + 8;

    int nverts = 0, npv = 0;
    for (auto &s : shapes) nverts += (s.kind == SH_CIRCLE) ? 1 : (int)s.verts.size();
    for (auto &p : prims) npv += (int)p.verts.size();
    for (auto &p : prims) if (p.verts.size() > 32) { err = "primitive with more than 32 vertices"; return -2; }
    for (auto &s : shapes) if (s.verts.size() > 16) { err = "collision shape with more than 16 vertices"; return -2; }      // (WS_MAXV of the placement queries)
    for (auto &e : entities) if (e.shapes.size() > 8) { err = "entity with more than 8 collision shapes"; return -2; }
    for (auto &p : prims) if (p.goal + 1 > 31) { err = "more than 30 goal regions"; return -2; }      // (5-bit field of the primitive record)
    if ((int)bodies.size() > CAP_BODIES || (int)shapes.size() > CAP_SHAPES || nverts > CAP_VERTS ||
        (int)joints.size() > CAP_JOINTS || (int)pairs.size() > CAP_PAIRS || (int)prims.size() > CAP_PRIMS ||
        npv > CAP_PVERTS) {
        err = "world exceeds compiled capacities (bodies " + std::to_string(bodies.size()) + ", shapes " +
              std::to_string(shapes.size()) + ", verts " + std::to_string(nverts) + ", joints " +
              std::to_string(joints.size()) + ", pairs " + std::to_string(pairs.size()) + ", prims " +
              std::to_string(prims.size()) + ", prim verts " + std::to_string(npv) + ")";
        return -2;
    }
    finalized = true;
    return 0;
}

int World::variant(const uint8_t *enabled, const int *shape_types, World &out, std::string &err) const {
    out = World();
    for (int k = 0; k < 5; k++) out.phys_vars[k] = phys_vars[k];
    out.entities = entities;
    for (size_t i = 0; i < out.entities.size(); i++) {
        EntityDef &e = out.entities[i];
        e.body = -1; e.shapes.clear();
        if (enabled) e.enabled = enabled[i] != 0;
        if (e.kind == 0 && !e.enabled) { err = "the robot cannot be disabled"; return -1; }
        if (shape_types && e.kind == 1 && shape_types[i] >= 0) {
            if (shape_types[i] > 6) { err = "bad shape type"; return -1; }
            e.shape_type = shape_types[i];
        }
    }
    return out.finalize(max_episode_steps, err);
}

void World::serialise(TmplHeader &h, std::vector<int32_t> &iw, std::vector<double> &rw, std::vector<double> &pw, bool strip_prims) const {
    std::memset(&h, 0, sizeof(h));
    h.n_bodies = (int)bodies.size();
    h.n_shapes = (int)shapes.size();
    h.n_joints = (int)joints.size();
    h.n_pairs = (int)pairs.size();
    h.n_prims = strip_prims ? 0 : (int)prims.size();
    int nverts = 0, npv = 0;
    for (auto &s : shapes) nverts += (s.kind == SH_CIRCLE) ? 1 : (int)s.verts.size();
    if (!strip_prims) for (auto &p : prims) npv += (int)p.verts.size();
    h.n_verts = nverts; h.n_pverts = npv;
    h.n_lverts = 0;
    if (!strip_prims) for (auto &p : prims) if (p.kind == PR_LINELOOP) h.n_lverts += (int)p.verts.size();
    h.n_state = (int)state_map.size();
    h.n_state_p = n_state_p;
    h.n_jacc = n_jacc;
    h.cache_slots = cache_slots; h.max_contacts = max_contacts; h.max_overlaps = max_overlaps;
    h.robot_body = robot_body; h.control_body = control_body;
    h.finger_body[0] = finger_body[0]; h.finger_body[1] = finger_body[1];
    h.motor_joint[0] = motor_joint[0]; h.motor_joint[1] = motor_joint[1];
    h.max_episode_steps = max_episode_steps;
    h.robot_j0 = robot_j0; h.n_islands = (int)island_j.size(); h.eye_body[0] = eye_body[0]; h.eye_body[1] = eye_body[1];
    TmplOff o(h);
    h.n_words_i = o.n_i; h.n_words_r = o.n_r; h.n_words_p = o.n_p;
    iw.assign(o.n_i, 0);
    rw.assign(o.n_r, 0.0);
    pw.assign(o.n_p, 0.0);
    const double dt = 1.0 / FPS / PHYS_STEPS;
    for (int b = 0; b < h.n_bodies; b++) {
        const BodyDef &B = bodies[b];
        iw[o.body_type + b] = B.type; iw[o.body_parent + b] = B.parent; iw[o.body_ent + b] = B.ent;
        pw[o.p_body_aoff + b] = B.aoff;
        rw[o.body_minv + (b) * TmplOff::S_body_minv] = B.m_inv; rw[o.body_iinv + (b) * TmplOff::S_body_iinv] = B.i_inv;
        rw[o.body_init + 3 * b] = B.x; rw[o.body_init + 3 * b + 1] = B.y; rw[o.body_init + 3 * b + 2] = B.a;
        rw[o.body_anchor + 2 * b] = B.ax; rw[o.body_anchor + 2 * b + 1] = B.ay;
        pw[o.p_body_init + 3 * b] = B.x; pw[o.p_body_init + 3 * b + 1] = B.y; pw[o.p_body_init + 3 * b + 2] = B.a;
        pw[o.p_body_anchor + 2 * b] = B.ax; pw[o.p_body_anchor + 2 * b + 1] = B.ay;
    }
    int voff = 0;
    for (int s = 0; s < h.n_shapes; s++) {
        const ShapeDef &S = shapes[s];
        int nv = (S.kind == SH_CIRCLE) ? 1 : (int)S.verts.size();
        iw[o.shape_kind + (s) * TmplOff::S_shape_kind] = S.kind; iw[o.shape_body + (s) * TmplOff::S_shape_body] = S.body;
        iw[o.shape_voff + (s) * TmplOff::S_shape_voff] = voff; iw[o.shape_nv + (s) * TmplOff::S_shape_nv] = nv;
        rw[o.shape_r + (s) * TmplOff::S_shape_r] = S.radius; rw[o.shape_u + (s) * TmplOff::S_shape_u] = S.friction;
        if (S.kind == SH_CIRCLE) {
            rw[o.lvx + (voff) * TmplOff::S_lvx] = 0; rw[o.lvy + (voff) * TmplOff::S_lvy] = 0;
        } else {
            for (int i = 0; i < nv; i++) { rw[o.lvx + (voff + i) * TmplOff::S_lvx] = S.verts[i].x; rw[o.lvy + (voff + i) * TmplOff::S_lvy] = S.verts[i].y; }
            if (S.kind == SH_SEGMENT) {
                // a segment is treated as a 2-vertex polygon: plane 1 = edge (a -> b) carries
                // cpSegmentShape's n = rperp(normalize(b - a)), plane 0 = edge (b -> a) carries -n
                double dx = S.verts[1].x - S.verts[0].x, dy = S.verts[1].y - S.verts[0].y;
                double len = std::sqrt(dx * dx + dy * dy);
                rw[o.lnx + (voff + 1) * TmplOff::S_lnx] = dy / len; rw[o.lny + (voff + 1) * TmplOff::S_lny] = -dx / len;
                rw[o.lnx + (voff) * TmplOff::S_lnx] = -dy / len; rw[o.lny + (voff) * TmplOff::S_lny] = dx / len;
            } else {
                // cpPolyShape SetVerts: plane i = edge (i-1 -> i), outward normal rperp(b - a)/|b - a|
                for (int i = 0; i < nv; i++) {
                    Vec2 a = S.verts[(i - 1 + nv) % nv], b = S.verts[i];
                    double ex = b.x - a.x, ey = b.y - a.y, len = std::sqrt(ex * ex + ey * ey);
                    rw[o.lnx + (voff + i) * TmplOff::S_lnx] = ey / len; rw[o.lny + (voff + i) * TmplOff::S_lny] = -ex / len;
                }
            }
        }
        voff += nv;
    }
    for (int j = 0; j < h.n_joints; j++) {
        const JointDef &J = joints[j];
        iw[o.joint_kind + j] = J.kind; iw[o.joint_a + j] = J.a; iw[o.joint_b + j] = J.b;
        iw[o.joint_acc + j] = joint_acc_off[j];
        iw[o.joint_pv + j] = J.pv;
        double *p = &rw[o.joint_p + j * JOINT_PARAMS];
        double ia = bodies[J.a].i_inv, ib = bodies[J.b].i_inv;
        p[0] = J.ax; p[1] = J.ay; p[2] = J.bx; p[3] = J.by; p[4] = J.p0; p[5] = J.p1; p[6] = J.p2;
        {
            double *q = &pw[o.p_joint + j * 7];
            q[0] = J.ax; q[1] = J.ay; q[2] = J.bx; q[3] = J.by; q[4] = J.p0; q[5] = J.p1; q[6] = J.p2;
        }
        p[7] = (1.0 - std::pow(J.error_bias, dt)) / dt;   // bias_coef / dt
        p[8] = J.max_bias;
        p[9] = J.max_force * dt;
        switch (J.kind) {                                   // constant angular effective masses
            case J_GEAR: p[0] = 1.0 / (ia * (1.0 / J.p1) + J.p1 * ib); break;
            case J_LIMIT: case J_MOTOR: p[0] = 1.0 / (ia + ib); break;
            case J_SPRING:
                p[0] = 1.0 / (ia + ib);
                p[5] = J.p1 * dt;                            // stiffness * dt
                p[6] = 1.0 - std::exp(-J.p2 * dt * (ia + ib));   // w_coef
                break;
            default: break;
        }
    }
    for (int k = 0; k < h.n_pairs; k++) iw[o.pair + k] = pairs[k].first | (pairs[k].second << 8);
#if MGX_BROAD_SAP
    // the candidate list is in (lower shape, higher shape) order: per lower shape the partners as a bit mask and the number of its first pair
    if (h.n_shapes <= 32) {
        for (int a = 0; a < h.n_shapes; a++) { iw[o.pair_allow + a] = 0; iw[o.pair_row + a] = 0; }
        for (int k = h.n_pairs - 1; k >= 0; k--) {
            const int a = std::min(pairs[k].first, pairs[k].second), b = std::max(pairs[k].first, pairs[k].second);
            iw[o.pair_allow + a] |= (int32_t)(1u << b);
            iw[o.pair_row + a] = k;
        }
    }
#endif
    for (int k = 0; k < h.n_state; k++) iw[o.state_map + k] = state_map[k];
    for (int k = 0; k < h.n_islands; k++) iw[o.island_j + k] = island_j[k];
    for (int k = 0; k < 3 * h.n_bodies; k++) iw[o.body_prow + k] = -1;
    for (int k = 0; k < h.n_state; k++) {
        int m = state_map[k], comp = m & 15, b = (m >> 4) & 0xFF, row = m >> 12;
        if (comp < 3) iw[o.body_prow + 3 * b + comp] = row;
    }
    int pvoff = 0, lvoff = 0;
    // the rasteriser's classification items (one per polygon edge / line segment / n-gon) are listed FRONT TO BACK: primitive k's
    // first item comes after those of every primitive drawn later
    std::vector<int> item_start(h.n_prims + 1, 0);
    for (int k = h.n_prims - 1; k >= 0; k--)
        item_start[k] = item_start[k + 1] + (k + 1 < h.n_prims ? (prims[k + 1].kind == PR_NGON ? 1 : (int)prims[k + 1].verts.size()) : 0);
    for (int k = 0; k < h.n_prims; k++) {
        const PrimDef &P = prims[k];
        int32_t *pi = &iw[o.prim_i + k * PRIM_IWORDS];
        double *pr = &rw[o.prim_r + k * PRIM_RWORDS];
        int nv = (P.kind == PR_NGON) ? P.ngon : (int)P.verts.size();
        pi[0] = P.kind; pi[1] = nv; pi[2] = pvoff | (lvoff << 16);      // (both < 2^15: CAP_PVERTS)
        if (P.kind == PR_LINELOOP) lvoff += (int)P.verts.size();
        pi[3] = P.xform | (P.body << 8) | ((P.eye_body + 1) << 16) | ((P.role + 1) << 24) | (int32_t)((uint32_t)(P.ent + 1) << 26);
        pi[4] = P.rgb[0] | (P.rgb[1] << 8) | (P.rgb[2] << 16);
        // low 16 bits: line stipple; 5 bits: 1 + goal ordinal; 11 bits: first classification item (< CAP_PVERTS + CAP_PRIMS)
        static_assert(CAP_PVERTS + CAP_PRIMS <= 2048, "item index field");
        pi[5] = P.stipple | ((P.goal + 1) << 16) | (int32_t)((uint32_t)item_start[k] << 21);
        {
            uint32_t ends = 0; int at = 0;
            for (int n : P.parts) { at += n; ends |= 1u << (at - 1); }
            if (P.parts.empty() && nv > 0 && P.kind != PR_NGON) ends = 1u << (nv - 1);
            pi[6] = (int32_t)ends;
        }
        pr[0] = P.eye_base[0]; pr[1] = P.eye_base[1]; pr[2] = P.eye_pre[0]; pr[3] = P.eye_pre[1];
        pr[4] = P.kind == PR_NGON ? std::cos(3.14159265358979323846 / nv) : 0.5 * (P.line_width + 1.0);   // apothem / circumradius; line half width
        pr[5] = P.radius;
        for (size_t i = 0; i < P.verts.size(); i++) { rw[o.pvx + pvoff + i] = P.verts[i].x; rw[o.pvy + pvoff + i] = P.verts[i].y; }
        for (size_t i = 0; i < P.verts.size(); i++) iw[o.pv_prim + pvoff + i] = k;
        pvoff += (int)P.verts.size();
    }
    double *c = &rw[o.consts];
    c[C_DT] = dt;
    pw[o.p_dt] = dt;
    c[C_CONTACT_BIAS_RATE] = (1.0 - std::pow(std::pow(1.0 - 0.1, 60.0), dt)) / dt;   // collision_bias default
    c[C_SLOP] = COLLISION_SLOP;
    c[C_SPEED_FWD] = 4.0 * ROBOT_RAD; c[C_SPEED_BACK] = 3.0 * ROBOT_RAD; c[C_TURN] = 1.5;   // entities.py:439-451
    for (int k = 0; k < N_PHYS_VARS; k++) c[C_PV0 + k] = phys_vars[k] * dt;                    // = p[9] of the joints they limit
    c[C_FINGER_OPEN] = PI / 8; c[C_FINGER_CLOSED] = 0.0;                                       // :227-228,452-457
}

}  // namespace mgx

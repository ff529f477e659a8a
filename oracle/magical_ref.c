/*
 * oracle/magical_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * fp64, one-env-at-a-time CPU restatement of the arithmetic behind the
 * reference hot path  BaseEnv.step()  (magical/base_env.py:255-292):
 *
 *   - Robot.update control law            magical/entities.py:459-479
 *   - pm.Space.step(dt)                   magical/base_env.py:243
 *        -> Chipmunk2D 7.0.x cpSpaceStep (third-party; pymunk~=5.6.0,
 *           setup.py:32-42; NOT vendored under /root/reference).  The step
 *           order, joint formulas, GJK/EPA narrowphase and contact solver
 *           below restate Chipmunk's published algorithm (cpSpaceStep.c,
 *           cpArbiter.c, cpCollision.c, cp*Joint.c, cpDampedRotarySpring.c)
 *           as summarised in SURVEY.md Appendix B.
 *   - Viewer.render / FilledPolygon / PolyLine  magical/gym_render.py:208-249,
 *           421-435,495-510 (GL point-sampled painter's fill at 384x384)
 *   - cv2.resize INTER_AREA 384->96        magical/benchmarks/__init__.py:234
 *
 * PARITY UNPINNED: pymunk / pyglet / cv2 are not installable in the build
 * container and the reference's tests hold no numeric vectors, so this file
 * cannot be checked against the real engines here.  What IS pinned: analytic
 * known-answer tests (tests/test_oracle_*.py), Appendix-D constants and the
 * reference's images/static-*.png initial frames.  (The Python side of the
 * oracle -- palette, force-limit draws, polygon sizing, scores -- IS pinned on
 * outputs of the reference's own code: oracle/__init__.py.)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load the library built from this file.  Build: see oracle/Makefile
 * (gcc -O2 -ffp-contract=off; no FMA contraction so results do not depend on
 * the host ISA).
 *
 * The world (bodies / shapes / joints / drawables) is assembled by
 * oracle/entities_ref.py, which restates magical/entities.py; this file only
 * knows generic rigid bodies, three collision-shape kinds, six joint kinds and
 * three drawable kinds.
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAX_BODIES 32
#define MAX_SHAPES 64
#define MAX_VERTS 8
#define MAX_JOINTS 64
#define MAX_ARB 256
#define MAX_GEOMS 128
#define MAX_GVERTS 100

typedef struct { double x, y; } v2;

static inline v2 V(double x, double y) { v2 r = {x, y}; return r; }
static inline v2 vadd(v2 a, v2 b) { return V(a.x + b.x, a.y + b.y); }
static inline v2 vsub(v2 a, v2 b) { return V(a.x - b.x, a.y - b.y); }
static inline v2 vneg(v2 a) { return V(-a.x, -a.y); }
static inline v2 vmul(v2 a, double s) { return V(a.x * s, a.y * s); }
static inline double vdot(v2 a, v2 b) { return a.x * b.x + a.y * b.y; }
static inline double vcross(v2 a, v2 b) { return a.x * b.y - a.y * b.x; }
static inline v2 vperp(v2 a) { return V(-a.y, a.x); }
static inline v2 vrperp(v2 a) { return V(a.y, -a.x); }
static inline double vlensq(v2 a) { return vdot(a, a); }
static inline double vlen(v2 a) { return sqrt(vdot(a, a)); }
static inline v2 vnormalize(v2 a) { return vmul(a, 1.0 / (vlen(a) + DBL_MIN)); }
static inline v2 vrotate(v2 a, v2 b) { return V(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
static inline v2 vlerp(v2 a, v2 b, double t) { return vadd(vmul(a, 1.0 - t), vmul(b, t)); }
static inline double fclamp(double f, double lo, double hi) { return fmin(fmax(f, lo), hi); }
static inline double fclamp01(double f) { return fmax(0.0, fmin(f, 1.0)); }
static inline v2 vclamp(v2 v, double len) {
    return (vdot(v, v) > len * len) ? vmul(vnormalize(v), len) : v;
}

/* ------------------------------------------------------------------ types */

enum { BODY_STATIC = 0, BODY_KINEMATIC = 1, BODY_DYNAMIC = 2 };
enum { SH_CIRCLE = 0, SH_SEGMENT = 1, SH_POLY = 2 };
enum { J_PIVOT = 0, J_GEAR = 1, J_SPRING = 2, J_PIN = 3, J_LIMIT = 4, J_MOTOR = 5 };
enum { ARB_FIRST = 0, ARB_NORMAL = 1, ARB_CACHED = 2 };

typedef struct {
    int type;
    double m_inv, i_inv;
    v2 p, v; double a, w;
    v2 v_bias; double w_bias;
    v2 rot;               /* (cos a, sin a), refreshed whenever a changes */
} Body;

typedef struct {
    int type, body;
    double r;             /* circle radius / segment radius / poly bevel radius */
    int n;                /* poly vertex count (circle: 1 = centre offset, segment: 2) */
    v2 lv[MAX_VERTS];     /* local verts (CCW) */
    v2 ln[MAX_VERTS];     /* local outward normal of edge (i-1 -> i)   (cpPolyShape planes) */
    v2 tv[MAX_VERTS];     /* world verts */
    v2 tn[MAX_VERTS];     /* world normals */
    double bb_l, bb_b, bb_r, bb_t;
    double u;             /* friction */
    int group;
    int sensor;
} Shape;

typedef struct {
    int type, a, b;
    v2 anchor_a, anchor_b;
    double p0, p1, p2;    /* gear: phase, ratio | spring: rest, k, damping | pin: dist | limit: min,max | motor: rate */
    double error_bias, max_bias, max_force;
    /* solver scratch + warm-start accumulators */
    v2 r1, r2, n, bias_v, jacc_v;
    double k[4];
    double bias, jacc, imass, w_coef, target_wrn;
} Joint;

typedef struct {
    v2 r1, r2;
    double n_mass, t_mass, bias, j_bias, jn_acc, jt_acc, bounce;
    uint32_t hash;
} Contact;

typedef struct {
    int used;
    int sa, sb;           /* shape indices as ordered by the collision function */
    int key_lo, key_hi;   /* min / max shape index: cache key and solve order */
    int state, stamp, count, active;
    v2 n;
    double u;
    uint32_t gjk_id;
    Contact c[2];
} Arbiter;

/* drawables: painter's order list, gym_render.py Geom tree flattened */
enum { G_POLY = 0, G_LINELOOP = 1 };
enum { X_WORLD = 0, X_BODY = 1, X_EYE = 2 };
typedef struct {
    int kind, nverts;
    v2 verts[MAX_GVERTS];
    double rgb[3];
    int xform, body;      /* X_BODY: body pose; X_EYE: robot pose o eye chain */
    v2 eye_base; int eye_body; v2 eye_pre;
    double line_width; int stipple;   /* G_LINELOOP */
} Geom;

typedef struct {
    int nbodies, nshapes, njoints, ngeoms;
    Body bodies[MAX_BODIES];
    Shape shapes[MAX_SHAPES];
    Joint joints[MAX_JOINTS];
    Geom geoms[MAX_GEOMS];
    Arbiter arbs[MAX_ARB];
    int order[MAX_ARB], norder;      /* active arbiters in solve order */
    /* space params (base_env.py:194-196 + Chipmunk defaults) */
    int iterations;
    double collision_slop, collision_bias, damping;
    int collision_persistence;
    int stamp;
    double prev_dt;
    /* robot control (entities.py:439-479) */
    int robot_body, control_body, finger_body[2], finger_motor[2];
    double robot_radius, rel_turn_angle, target_speed, target_finger_angle;
    double finger_rot_limit_outer, finger_rot_limit_inner;
    double bg_rgb[3];
    int episode_steps;
    int gjk_warm;   /* use cached GJK ids like Chipmunk (1) or cold start (0) */
} World;

/* ------------------------------------------------------------ unknowns (sensitivity study only)
 *
 * The choices this restatement makes "from recollection" where the third-party engines cannot be consulted (header: PARITY
 * UNPINNED).  tools/oracle_unknowns.py flips them one at a time and measures what moves (profiles/r05_oracle_unknowns_sensitivity.txt,
 * DESIGN.md section 6).  Process-global, default 0 = the oracle as every test uses it; never set by tests/, smoke() or bench.py. */
enum {
    UNK_ARB_DESCENDING = 1,    /* arbiters solved in descending (shape_i, shape_j) order instead of ascending (Chipmunk: BBTree order) */
    UNK_PERSISTENCE_1 = 2,     /* collision_persistence 1 instead of Chipmunk's documented default 3 */
    UNK_PERSISTENCE_5 = 4,     /* ... 5 */
    UNK_MATCH_BY_INDEX = 8,    /* cached contact impulses matched by point index instead of by feature hash (cpArbiterUpdate) */
    UNK_NO_WARM_CONTACTS = 16, /* cached contact impulses never carried over */
    UNK_POLY_RADIUS = 32,      /* polygons bevelled by 1e-3 instead of pymunk's documented default radius 0 */
    UNK_FILL_EXCLUSIVE = 64,   /* a sample exactly ON a polygon edge is outside (GL: top-left rule; here: inclusive) */
    UNK_RESIZE_HALF_UP = 128,  /* INTER_AREA block mean rounded half up instead of cvRound's ties-to-even */
    UNK_GJK_COLD = 256,        /* narrowphase started cold every step instead of from the arbiter's cached feature pair */
};
static int g_unknowns = 0;
void ref_set_unknowns(int flags) { g_unknowns = flags; }
int ref_get_unknowns(void) { return g_unknowns; }

/* ------------------------------------------------------------ world build */

World *ref_new(void) {
    World *w = (World *)calloc(1, sizeof(World));
    w->iterations = 10;
    w->collision_slop = 0.1;               /* Chipmunk default; reference overrides to 0.01 */
    w->collision_bias = pow(1.0 - 0.1, 60.0);
    w->collision_persistence = 3;          /* (UNK_PERSISTENCE_*: read at step time) */
    w->damping = 1.0;
    w->robot_body = w->control_body = -1;
    w->gjk_warm = 1;
    return w;
}
void ref_free(World *w) { free(w); }
World *ref_clone(const World *w) {
    World *c = (World *)malloc(sizeof(World));
    memcpy(c, w, sizeof(World));
    return c;
}
/* Sensitivity probe (tests only): every warm-start quantity the solver carries from one substep to the next -- the joints'
 * accumulated impulses and the cached contact impulses -- scaled by (1 + rel * u), u uniform in [-1, 1] from a small LCG.
 * What a second implementation of the same step differs from this one by after a few steps is exactly such a perturbation
 * (plus one of the poses, which the callers apply through ref_set_bodies). */
void ref_perturb_warm(World *w, double rel, unsigned seed) {
    unsigned s = seed * 2654435761u + 12345u;
#define PERT(x) do { s = s * 1664525u + 1013904223u; (x) *= 1.0 + rel * (((double)(s >> 8) / 8388608.0) - 1.0); } while (0)
    for (int i = 0; i < w->njoints; i++) { Joint *j = &w->joints[i]; PERT(j->jacc); PERT(j->jacc_v.x); PERT(j->jacc_v.y); }
    for (int i = 0; i < MAX_ARB; i++) {
        Arbiter *a = &w->arbs[i];
        if (!a->used) continue;
        for (int k = 0; k < a->count; k++) { PERT(a->c[k].jn_acc); PERT(a->c[k].jt_acc); }
    }
#undef PERT
}
/* Cold solver state (tests only): every accumulated impulse zeroed -- joints and cached contact points -- with the arbiters (and so
 * the contact persistence) left as they are.  Full-state teacher forcing: two implementations that start an env-step from equal
 * poses, velocities AND zero warm start differ afterwards by their arithmetic alone. */
void ref_clear_warm(World *w) {
    for (int i = 0; i < w->njoints; i++) { Joint *j = &w->joints[i]; j->jacc = 0.0; j->jacc_v.x = j->jacc_v.y = 0.0; }
    for (int i = 0; i < MAX_ARB; i++) {
        Arbiter *a = &w->arbs[i];
        if (!a->used) continue;
        for (int k = 0; k < 2; k++) { a->c[k].jn_acc = 0.0; a->c[k].jt_acc = 0.0; }
    }
}
void ref_set_space(World *w, int iterations, double slop) { w->iterations = iterations; w->collision_slop = slop; }
void ref_set_bg(World *w, double r, double g, double b) { w->bg_rgb[0] = r; w->bg_rgb[1] = g; w->bg_rgb[2] = b; }
void ref_set_gjk_warm(World *w, int on) { w->gjk_warm = on; }

static void body_set_angle(Body *b, double a) { b->a = a; b->rot = V(cos(a), sin(a)); }

int ref_add_body(World *w, int type, double mass, double moment, double x, double y, double angle) {
    Body *b = &w->bodies[w->nbodies];
    memset(b, 0, sizeof(*b));
    b->type = type;
    if (type == BODY_DYNAMIC) { b->m_inv = 1.0 / mass; b->i_inv = 1.0 / moment; }
    b->p = V(x, y);
    body_set_angle(b, angle);
    return w->nbodies++;
}
void ref_body_set_pos(World *w, int b, double x, double y) { w->bodies[b].p = V(x, y); }

static void poly_set_verts(Shape *s, int n, const double *xy) {
    s->n = n;
    for (int i = 0; i < n; i++) s->lv[i] = V(xy[2 * i], xy[2 * i + 1]);
    for (int i = 0; i < n; i++) {   /* cpPolyShape SetVerts: plane i = edge (i-1 -> i), n = rperp */
        v2 a = s->lv[(i - 1 + n) % n], b = s->lv[i];
        s->ln[i] = vnormalize(vrperp(vsub(b, a)));
    }
}
int ref_add_circle(World *w, int body, double radius, double friction, int group, int sensor) {
    Shape *s = &w->shapes[w->nshapes]; memset(s, 0, sizeof(*s));
    s->type = SH_CIRCLE; s->body = body; s->r = radius; s->n = 1; s->lv[0] = V(0, 0);
    s->u = friction; s->group = group; s->sensor = sensor;
    return w->nshapes++;
}
int ref_add_poly(World *w, int body, int n, const double *xy, double radius, double friction, int group, int sensor) {
    Shape *s = &w->shapes[w->nshapes]; memset(s, 0, sizeof(*s));
    s->type = SH_POLY; s->body = body; s->r = radius + ((g_unknowns & UNK_POLY_RADIUS) && !sensor ? 1e-3 : 0.0);
    poly_set_verts(s, n, xy);
    s->u = friction; s->group = group; s->sensor = sensor;
    return w->nshapes++;
}
int ref_add_segment(World *w, int body, double ax, double ay, double bx, double by, double radius, double friction) {
    Shape *s = &w->shapes[w->nshapes]; memset(s, 0, sizeof(*s));
    s->type = SH_SEGMENT; s->body = body; s->r = radius; s->n = 2;
    s->lv[0] = V(ax, ay); s->lv[1] = V(bx, by);
    s->ln[0] = vrperp(vnormalize(vsub(s->lv[1], s->lv[0])));   /* cpSegmentShapeInit */
    s->u = friction;
    return w->nshapes++;
}
static Joint *new_joint(World *w, int type, int a, int b) {
    Joint *j = &w->joints[w->njoints++]; memset(j, 0, sizeof(*j));
    j->type = type; j->a = a; j->b = b;
    j->error_bias = pow(1.0 - 0.1, 60.0); j->max_bias = INFINITY; j->max_force = INFINITY;
    return j;
}
int ref_add_pivot(World *w, int a, int b, double ax, double ay, double bx, double by) {
    Joint *j = new_joint(w, J_PIVOT, a, b); j->anchor_a = V(ax, ay); j->anchor_b = V(bx, by); return w->njoints - 1;
}
int ref_add_gear(World *w, int a, int b, double phase, double ratio) {
    Joint *j = new_joint(w, J_GEAR, a, b); j->p0 = phase; j->p1 = ratio; return w->njoints - 1;
}
int ref_add_spring(World *w, int a, int b, double rest, double k, double damping) {
    Joint *j = new_joint(w, J_SPRING, a, b); j->p0 = rest; j->p1 = k; j->p2 = damping; return w->njoints - 1;
}
int ref_add_pin(World *w, int a, int b, double ax, double ay, double bx, double by) {
    Joint *j = new_joint(w, J_PIN, a, b); j->anchor_a = V(ax, ay); j->anchor_b = V(bx, by);
    /* cpPinJointInit: rest length = world-space anchor separation at construction */
    Body *A = &w->bodies[a], *B = &w->bodies[b];
    v2 p1 = vadd(A->p, vrotate(A->rot, j->anchor_a)), p2 = vadd(B->p, vrotate(B->rot, j->anchor_b));
    j->p0 = vlen(vsub(p2, p1));
    return w->njoints - 1;
}
int ref_add_limit(World *w, int a, int b, double lo, double hi) {
    Joint *j = new_joint(w, J_LIMIT, a, b); j->p0 = lo; j->p1 = hi; return w->njoints - 1;
}
int ref_add_motor(World *w, int a, int b, double rate) {
    Joint *j = new_joint(w, J_MOTOR, a, b); j->p0 = rate; return w->njoints - 1;
}
void ref_joint_params(World *w, int j, double error_bias, double max_bias, double max_force) {
    Joint *J = &w->joints[j];
    if (!isnan(error_bias)) J->error_bias = error_bias;
    if (!isnan(max_bias)) J->max_bias = max_bias;
    if (!isnan(max_force)) J->max_force = max_force;
}
void ref_set_robot(World *w, int robot_body, int control_body, int finger_l, int finger_r,
                   int motor_l, int motor_r, double radius, double lim_outer, double lim_inner) {
    w->robot_body = robot_body; w->control_body = control_body;
    w->finger_body[0] = finger_l; w->finger_body[1] = finger_r;
    w->finger_motor[0] = motor_l; w->finger_motor[1] = motor_r;
    w->robot_radius = radius; w->finger_rot_limit_outer = lim_outer; w->finger_rot_limit_inner = lim_inner;
}
int ref_add_geom(World *w, int kind, int n, const double *xy, double r, double g, double b,
                 int xform, int body, double ebx, double eby, int eye_body, double epx, double epy,
                 double line_width, int stipple) {
    Geom *G = &w->geoms[w->ngeoms]; memset(G, 0, sizeof(*G));
    G->kind = kind; G->nverts = n;
    for (int i = 0; i < n; i++) G->verts[i] = V(xy[2 * i], xy[2 * i + 1]);
    G->rgb[0] = r; G->rgb[1] = g; G->rgb[2] = b;
    G->xform = xform; G->body = body; G->eye_base = V(ebx, eby); G->eye_body = eye_body; G->eye_pre = V(epx, epy);
    G->line_width = line_width; G->stipple = stipple;
    return w->ngeoms++;
}

/* ---------------------------------------------------------------- shapes */

static void shape_update(World *w, Shape *s) {
    Body *b = &w->bodies[s->body];
    /* cpTransformPoint with cog = 0: p' = rot (x) p + pos */
    if (s->type == SH_CIRCLE) {
        s->tv[0] = vadd(b->p, vrotate(b->rot, s->lv[0]));
        s->bb_l = s->tv[0].x - s->r; s->bb_r = s->tv[0].x + s->r;
        s->bb_b = s->tv[0].y - s->r; s->bb_t = s->tv[0].y + s->r;
    } else if (s->type == SH_SEGMENT) {
        s->tv[0] = vadd(b->p, vrotate(b->rot, s->lv[0]));
        s->tv[1] = vadd(b->p, vrotate(b->rot, s->lv[1]));
        s->tn[0] = vrotate(b->rot, s->ln[0]);
        s->bb_l = fmin(s->tv[0].x, s->tv[1].x) - s->r; s->bb_r = fmax(s->tv[0].x, s->tv[1].x) + s->r;
        s->bb_b = fmin(s->tv[0].y, s->tv[1].y) - s->r; s->bb_t = fmax(s->tv[0].y, s->tv[1].y) + s->r;
    } else {
        double l = INFINITY, r = -INFINITY, bo = INFINITY, t = -INFINITY;
        for (int i = 0; i < s->n; i++) {
            v2 v = vadd(b->p, vrotate(b->rot, s->lv[i]));
            s->tv[i] = v; s->tn[i] = vrotate(b->rot, s->ln[i]);
            l = fmin(l, v.x); r = fmax(r, v.x); bo = fmin(bo, v.y); t = fmax(t, v.y);
        }
        s->bb_l = l - s->r; s->bb_b = bo - s->r; s->bb_r = r + s->r; s->bb_t = t + s->r;
    }
}

/* ------------------------------------------------- narrowphase (cpCollision.c) */

typedef struct { v2 p; int index; } SupportPoint;
typedef struct { v2 a, b, ab; uint32_t id; } MinkowskiPoint;
typedef struct { v2 a, b, n; double d; uint32_t id; } ClosestPoints;
typedef struct { v2 pa; uint32_t ha; v2 pb; uint32_t hb; double r; v2 n; } Edge;
typedef struct { const Shape *s1, *s2; } SupportContext;

static int poly_support_index(const Shape *s, v2 n) {
    double max = -INFINITY; int index = 0;
    for (int i = 0; i < s->n; i++) {
        double d = vdot(s->tv[i], n);
        if (d > max) { max = d; index = i; }
    }
    return index;
}
static SupportPoint shape_support(const Shape *s, v2 n) {
    SupportPoint sp;
    if (s->type == SH_CIRCLE) { sp.p = s->tv[0]; sp.index = 0; }
    else if (s->type == SH_SEGMENT) {
        if (vdot(s->tv[0], n) > vdot(s->tv[1], n)) { sp.p = s->tv[0]; sp.index = 0; }
        else { sp.p = s->tv[1]; sp.index = 1; }
    } else { int i = poly_support_index(s, n); sp.p = s->tv[i]; sp.index = i; }
    return sp;
}
static SupportPoint shape_point(const Shape *s, int i) {
    SupportPoint sp;
    if (s->type == SH_CIRCLE) { sp.p = s->tv[0]; sp.index = 0; }
    else if (s->type == SH_SEGMENT) { sp.p = s->tv[i ? 1 : 0]; sp.index = i ? 1 : 0; }
    else { int k = (i < s->n ? i : 0); sp.p = s->tv[k]; sp.index = k; }
    return sp;
}
static MinkowskiPoint mk_new(SupportPoint a, SupportPoint b) {
    MinkowskiPoint m; m.a = a.p; m.b = b.p; m.ab = vsub(b.p, a.p);
    m.id = ((uint32_t)(a.index & 0xFF) << 8) | (uint32_t)(b.index & 0xFF);
    return m;
}
static MinkowskiPoint support(const SupportContext *ctx, v2 n) {
    return mk_new(shape_support(ctx->s1, vneg(n)), shape_support(ctx->s2, n));
}
static double closest_t(v2 a, v2 b) {
    v2 delta = vsub(b, a);
    return -fclamp(vdot(delta, vadd(a, b)) / vlensq(delta), -1.0, 1.0);
}
static v2 lerp_t(v2 a, v2 b, double t) {
    double ht = 0.5 * t;
    return vadd(vmul(a, 0.5 - ht), vmul(b, 0.5 + ht));
}
static double closest_dist(v2 v0, v2 v1) { return vlensq(lerp_t(v0, v1, closest_t(v0, v1))); }
static int check_point_greater(v2 a, v2 b, v2 c) {
    return (b.y - a.y) * (a.x + b.x - 2 * c.x) > (b.x - a.x) * (a.y + b.y - 2 * c.y);
}
static int check_axis(v2 v0, v2 v1, v2 p, v2 n) { return vdot(p, n) <= fmax(vdot(v0, n), vdot(v1, n)); }

static ClosestPoints closest_points_new(MinkowskiPoint v0, MinkowskiPoint v1) {
    double t = closest_t(v0.ab, v1.ab);
    v2 p = lerp_t(v0.ab, v1.ab, t);
    v2 pa = lerp_t(v0.a, v1.a, t), pb = lerp_t(v0.b, v1.b, t);
    uint32_t id = ((v0.id & 0xFFFF) << 16) | (v1.id & 0xFFFF);
    v2 delta = vsub(v1.ab, v0.ab);
    v2 n = vnormalize(vrperp(delta));
    double d = vdot(n, p);
    ClosestPoints cp;
    if (d <= 0.0 || (-1.0 < t && t < 1.0)) {
        cp.a = pa; cp.b = pb; cp.n = n; cp.d = d; cp.id = id;
    } else {   /* vertex/vertex: axis is not an edge normal of the Minkowski difference */
        double d2 = vlen(p);
        v2 n2 = vmul(p, 1.0 / (d2 + DBL_MIN));
        cp.a = pa; cp.b = pb; cp.n = n2; cp.d = d2; cp.id = id;
    }
    return cp;
}

#define MAX_GJK_ITERATIONS 30
#define MAX_EPA_ITERATIONS 30

static ClosestPoints epa(const SupportContext *ctx, MinkowskiPoint v0, MinkowskiPoint v1, MinkowskiPoint v2_) {
    MinkowskiPoint hull[MAX_EPA_ITERATIONS + 4], hull2[MAX_EPA_ITERATIONS + 4];
    int count = 3; hull[0] = v0; hull[1] = v1; hull[2] = v2_;
    for (int iteration = 1;; iteration++) {
        int mini = 0; double min_dist = INFINITY;
        for (int j = 0, i = count - 1; j < count; i = j, j++) {
            double d = closest_dist(hull[i].ab, hull[j].ab);
            if (d < min_dist) { min_dist = d; mini = i; }
        }
        MinkowskiPoint e0 = hull[mini], e1 = hull[(mini + 1) % count];
        MinkowskiPoint p = support(ctx, vperp(vsub(e1.ab, e0.ab)));
        int duplicate = (p.id == e0.id || p.id == e1.id);
        if (!duplicate && check_point_greater(e0.ab, e1.ab, p.ab) && iteration < MAX_EPA_ITERATIONS) {
            int count2 = 1; hull2[0] = p;
            for (int i = 0; i < count; i++) {
                int index = (mini + 1 + i) % count;
                v2 h0 = hull2[count2 - 1].ab, h1 = hull[index].ab;
                v2 h2 = (i + 1 < count ? hull[(index + 1) % count] : p).ab;
                if (check_point_greater(h0, h2, h1)) hull2[count2++] = hull[index];
            }
            memcpy(hull, hull2, sizeof(MinkowskiPoint) * count2); count = count2;
        } else {
            return closest_points_new(e0, e1);
        }
    }
}

static ClosestPoints gjk(const SupportContext *ctx, uint32_t *id, int warm) {
    MinkowskiPoint v0, v1;
    if (*id && warm) {
        v0 = mk_new(shape_point(ctx->s1, (*id >> 24) & 0xFF), shape_point(ctx->s2, (*id >> 16) & 0xFF));
        v1 = mk_new(shape_point(ctx->s1, (*id >> 8) & 0xFF), shape_point(ctx->s2, (*id) & 0xFF));
    } else {
        v2 c1 = V((ctx->s1->bb_l + ctx->s1->bb_r) * 0.5, (ctx->s1->bb_b + ctx->s1->bb_t) * 0.5);
        v2 c2 = V((ctx->s2->bb_l + ctx->s2->bb_r) * 0.5, (ctx->s2->bb_b + ctx->s2->bb_t) * 0.5);
        v2 axis = vperp(vsub(c1, c2));
        v0 = support(ctx, axis); v1 = support(ctx, vneg(axis));
    }
    ClosestPoints pts;
    for (int iteration = 1;;) {
        if (iteration > MAX_GJK_ITERATIONS) { pts = closest_points_new(v0, v1); break; }
        if (check_point_greater(v1.ab, v0.ab, V(0, 0))) {   /* origin behind axis: flip */
            MinkowskiPoint t = v0; v0 = v1; v1 = t; continue;
        }
        double t = closest_t(v0.ab, v1.ab);
        v2 n = (-1.0 < t && t < 1.0) ? vperp(vsub(v1.ab, v0.ab)) : vneg(lerp_t(v0.ab, v1.ab, t));
        MinkowskiPoint p = support(ctx, n);
        if (check_point_greater(p.ab, v0.ab, V(0, 0)) && check_point_greater(v1.ab, p.ab, V(0, 0))) {
            pts = epa(ctx, v0, p, v1); break;
        }
        if (check_axis(v0.ab, v1.ab, p.ab, n)) { pts = closest_points_new(v0, v1); break; }
        if (closest_dist(v0.ab, p.ab) < closest_dist(p.ab, v1.ab)) { v1 = p; } else { v0 = p; }
        iteration++;
    }
    *id = pts.id;
    return pts;
}

/* contact ids: only equality between consecutive steps of the SAME shape pair
 * matters (cpArbiterUpdate hash match), so a (feature a, feature b) code does
 * the job of CP_HASH_PAIR(hashid, index) pairs. */
static uint32_t feat_hash(uint32_t fa, uint32_t fb) { return 1u + fa * 64u + fb; }

static Edge support_edge_poly(const Shape *s, v2 n) {
    int count = s->n;
    int i1 = poly_support_index(s, n);
    int i0 = (i1 - 1 + count) % count, i2 = (i1 + 1) % count;
    Edge e;
    if (vdot(n, s->tn[i1]) > vdot(n, s->tn[i2])) {
        e.pa = s->tv[i0]; e.ha = (uint32_t)i0; e.pb = s->tv[i1]; e.hb = (uint32_t)i1; e.r = s->r; e.n = s->tn[i1];
    } else {
        e.pa = s->tv[i1]; e.ha = (uint32_t)i1; e.pb = s->tv[i2]; e.hb = (uint32_t)i2; e.r = s->r; e.n = s->tn[i2];
    }
    return e;
}
static Edge support_edge_segment(const Shape *s, v2 n) {
    Edge e;
    if (vdot(s->tn[0], n) > 0.0) {
        e.pa = s->tv[0]; e.ha = 0; e.pb = s->tv[1]; e.hb = 1; e.r = s->r; e.n = s->tn[0];
    } else {
        e.pa = s->tv[1]; e.ha = 1; e.pb = s->tv[0]; e.hb = 0; e.r = s->r; e.n = vneg(s->tn[0]);
    }
    return e;
}

typedef struct { int count; v2 n; v2 p1[2], p2[2]; uint32_t hash[2]; uint32_t id; } CollisionInfo;

static void push_contact(CollisionInfo *info, v2 p1, v2 p2, uint32_t hash) {
    info->p1[info->count] = p1; info->p2[info->count] = p2; info->hash[info->count] = hash; info->count++;
}
static void contact_points(Edge e1, Edge e2, ClosestPoints points, CollisionInfo *info) {
    double mindist = e1.r + e2.r;
    if (points.d <= mindist) {
        v2 n = info->n = points.n;
        double d_e1_a = vcross(e1.pa, n), d_e1_b = vcross(e1.pb, n);
        double d_e2_a = vcross(e2.pa, n), d_e2_b = vcross(e2.pb, n);
        double e1_denom = 1.0 / (d_e1_b - d_e1_a + DBL_MIN);
        double e2_denom = 1.0 / (d_e2_b - d_e2_a + DBL_MIN);
        {
            v2 p1 = vadd(vmul(n, e1.r), vlerp(e1.pa, e1.pb, fclamp01((d_e2_b - d_e1_a) * e1_denom)));
            v2 p2 = vadd(vmul(n, -e2.r), vlerp(e2.pa, e2.pb, fclamp01((d_e1_a - d_e2_a) * e2_denom)));
            double dist = vdot(vsub(p2, p1), n);
            if (dist <= 0.0) push_contact(info, p1, p2, feat_hash(e1.ha, e2.hb));
        }
        {
            v2 p1 = vadd(vmul(n, e1.r), vlerp(e1.pa, e1.pb, fclamp01((d_e2_a - d_e1_a) * e1_denom)));
            v2 p2 = vadd(vmul(n, -e2.r), vlerp(e2.pa, e2.pb, fclamp01((d_e1_b - d_e2_a) * e2_denom)));
            double dist = vdot(vsub(p2, p1), n);
            if (dist <= 0.0) push_contact(info, p1, p2, feat_hash(e1.hb, e2.ha));
        }
    }
}

static void circle_to_circle(const Shape *c1, const Shape *c2, CollisionInfo *info) {
    double mindist = c1->r + c2->r;
    v2 delta = vsub(c2->tv[0], c1->tv[0]);
    double distsq = vlensq(delta);
    if (distsq < mindist * mindist) {
        double dist = sqrt(distsq);
        v2 n = info->n = (dist ? vmul(delta, 1.0 / dist) : V(1.0, 0.0));
        push_contact(info, vadd(c1->tv[0], vmul(n, c1->r)), vadd(c2->tv[0], vmul(n, -c2->r)), 0);
    }
}
static void circle_to_segment(const Shape *c, const Shape *seg, CollisionInfo *info) {
    v2 seg_a = seg->tv[0], seg_b = seg->tv[1], center = c->tv[0];
    v2 seg_delta = vsub(seg_b, seg_a);
    double closest_t_ = fclamp01(vdot(seg_delta, vsub(center, seg_a)) / vlensq(seg_delta));
    v2 closest = vadd(seg_a, vmul(seg_delta, closest_t_));
    double mindist = c->r + seg->r;
    v2 delta = vsub(closest, center);
    double distsq = vlensq(delta);
    if (distsq < mindist * mindist) {
        double dist = sqrt(distsq);
        v2 n = info->n = (dist ? vmul(delta, 1.0 / dist) : seg->tn[0]);
        /* a_tangent = b_tangent = 0 (pymunk never sets neighbours): no end-cap rejection */
        push_contact(info, vadd(center, vmul(n, c->r)), vadd(closest, vmul(n, -seg->r)), 0);
    }
}
static void circle_to_poly(const Shape *c, const Shape *poly, CollisionInfo *info, int warm) {
    SupportContext ctx = {c, poly};
    ClosestPoints points = gjk(&ctx, &info->id, warm);
    if (points.d <= c->r + poly->r) {
        v2 n = info->n = points.n;
        push_contact(info, vadd(points.a, vmul(n, c->r)), vadd(points.b, vmul(n, -poly->r)), 0);
    }
}
static void segment_to_poly(const Shape *seg, const Shape *poly, CollisionInfo *info, int warm) {
    SupportContext ctx = {seg, poly};
    ClosestPoints points = gjk(&ctx, &info->id, warm);
    v2 n = points.n;
    if (points.d - seg->r - poly->r <= 0.0) {
        contact_points(support_edge_segment(seg, n), support_edge_poly(poly, vneg(n)), points, info);
    }
}
static void poly_to_poly(const Shape *p1, const Shape *p2, CollisionInfo *info, int warm) {
    SupportContext ctx = {p1, p2};
    ClosestPoints points = gjk(&ctx, &info->id, warm);
    if (points.d - p1->r - p2->r <= 0.0) {
        contact_points(support_edge_poly(p1, points.n), support_edge_poly(p2, vneg(points.n)), points, info);
    }
}

/* cpCollide: order the pair by shape type (circle < segment < poly) */
static void collide(const Shape *a, const Shape *b, CollisionInfo *info, int warm) {
    info->count = 0;
    if (a->type == SH_CIRCLE && b->type == SH_CIRCLE) circle_to_circle(a, b, info);
    else if (a->type == SH_CIRCLE && b->type == SH_SEGMENT) circle_to_segment(a, b, info);
    else if (a->type == SH_CIRCLE && b->type == SH_POLY) circle_to_poly(a, b, info, warm);
    else if (a->type == SH_SEGMENT && b->type == SH_POLY) segment_to_poly(a, b, info, warm);
    else if (a->type == SH_POLY && b->type == SH_POLY) poly_to_poly(a, b, info, warm);
    /* segment-segment: both static here, never queried */
}

/* ------------------------------------------------------------ arbiters */

static Arbiter *arb_find(World *w, int lo, int hi) {
    for (int i = 0; i < MAX_ARB; i++)
        if (w->arbs[i].used && w->arbs[i].key_lo == lo && w->arbs[i].key_hi == hi) return &w->arbs[i];
    return NULL;
}
static Arbiter *arb_alloc(World *w) {
    for (int i = 0; i < MAX_ARB; i++) if (!w->arbs[i].used) { memset(&w->arbs[i], 0, sizeof(Arbiter)); w->arbs[i].used = 1; return &w->arbs[i]; }
    return NULL;
}

static int query_reject(const World *w, const Shape *a, const Shape *b) {
    if (!(a->bb_l <= b->bb_r && b->bb_l <= a->bb_r && a->bb_b <= b->bb_t && b->bb_b <= a->bb_t)) return 1;
    if (a->body == b->body) return 1;
    if (a->group != 0 && a->group == b->group) return 1;
    (void)w;
    return 0;
}

/* cpSpaceCollideShapes + cpArbiterUpdate for one candidate pair (i < j) */
static void collide_pair(World *w, int i, int j) {
    Shape *si = &w->shapes[i], *sj = &w->shapes[j];
    const Body *bi = &w->bodies[si->body], *bj = &w->bodies[sj->body];
    if (bi->type == BODY_STATIC && bj->type == BODY_STATIC) return;   /* static index is never self-queried */
    if (query_reject(w, si, sj)) return;
    int sa = i, sb = j;
    if (w->shapes[sa].type > w->shapes[sb].type) { sa = j; sb = i; }
    Arbiter *arb = arb_find(w, i, j);
    CollisionInfo info; memset(&info, 0, sizeof(info));
    info.id = arb ? arb->gjk_id : 0;
    collide(&w->shapes[sa], &w->shapes[sb], &info, w->gjk_warm && !(g_unknowns & UNK_GJK_COLD));
    if (info.count == 0) { if (arb) arb->gjk_id = info.id; return; }
    if (!arb) { arb = arb_alloc(w); if (!arb) return; arb->key_lo = i; arb->key_hi = j; arb->state = ARB_FIRST; arb->count = 0; }
    arb->gjk_id = info.id;
    const Body *A = &w->bodies[w->shapes[sa].body], *B = &w->bodies[w->shapes[sb].body];
    Contact nc[2];
    for (int k = 0; k < info.count; k++) {
        memset(&nc[k], 0, sizeof(Contact));
        nc[k].r1 = vsub(info.p1[k], A->p); nc[k].r2 = vsub(info.p2[k], B->p);
        nc[k].hash = info.hash[k];
        if (g_unknowns & UNK_NO_WARM_CONTACTS) continue;
        if (g_unknowns & UNK_MATCH_BY_INDEX) { if (k < arb->count) { nc[k].jn_acc = arb->c[k].jn_acc; nc[k].jt_acc = arb->c[k].jt_acc; } continue; }
        for (int q = 0; q < arb->count; q++)
            if (arb->c[q].hash == nc[k].hash) { nc[k].jn_acc = arb->c[q].jn_acc; nc[k].jt_acc = arb->c[q].jt_acc; }
    }
    arb->sa = sa; arb->sb = sb;
    arb->count = info.count; arb->c[0] = nc[0]; arb->c[1] = nc[1];
    arb->n = info.n;
    arb->u = w->shapes[sa].u * w->shapes[sb].u;
    if (arb->state == ARB_CACHED) arb->state = ARB_FIRST;
    arb->stamp = w->stamp;
    int sensor = w->shapes[sa].sensor || w->shapes[sb].sensor;
    int both_inf = (A->m_inv == 0.0 && B->m_inv == 0.0);
    if (!sensor && !both_inf) { arb->active = 1; }
    else { arb->active = 0; arb->count = 0; arb->state = ARB_NORMAL; }
}

/* ------------------------------------------------------------ solver helpers */

static inline v2 relative_velocity(const Body *a, const Body *b, v2 r1, v2 r2) {
    v2 v1_sum = vadd(a->v, vmul(vperp(r1), a->w));
    v2 v2_sum = vadd(b->v, vmul(vperp(r2), b->w));
    return vsub(v2_sum, v1_sum);
}
static inline void apply_impulse(Body *b, v2 j, v2 r) {
    b->v = vadd(b->v, vmul(j, b->m_inv));
    b->w += b->i_inv * vcross(r, j);
}
static inline void apply_impulses(Body *a, Body *b, v2 r1, v2 r2, v2 j) {
    apply_impulse(a, vneg(j), r1); apply_impulse(b, j, r2);
}
static inline void apply_bias_impulse(Body *b, v2 j, v2 r) {
    b->v_bias = vadd(b->v_bias, vmul(j, b->m_inv));
    b->w_bias += b->i_inv * vcross(r, j);
}
static inline double k_scalar_body(const Body *b, v2 r, v2 n) { double rcn = vcross(r, n); return b->m_inv + b->i_inv * rcn * rcn; }
static inline double k_scalar(const Body *a, const Body *b, v2 r1, v2 r2, v2 n) { return k_scalar_body(a, r1, n) + k_scalar_body(b, r2, n); }
static void k_tensor(const Body *a, const Body *b, v2 r1, v2 r2, double *k) {
    double m_sum = a->m_inv + b->m_inv;
    double k11 = m_sum, k12 = 0.0, k21 = 0.0, k22 = m_sum;
    double a_i = a->i_inv;
    double r1xsq = r1.x * r1.x * a_i, r1ysq = r1.y * r1.y * a_i, r1nxy = -r1.x * r1.y * a_i;
    k11 += r1ysq; k12 += r1nxy; k21 += r1nxy; k22 += r1xsq;
    double b_i = b->i_inv;
    double r2xsq = r2.x * r2.x * b_i, r2ysq = r2.y * r2.y * b_i, r2nxy = -r2.x * r2.y * b_i;
    k11 += r2ysq; k12 += r2nxy; k21 += r2nxy; k22 += r2xsq;
    double det_inv = 1.0 / (k11 * k22 - k12 * k21);
    k[0] = k22 * det_inv; k[1] = -k12 * det_inv; k[2] = -k21 * det_inv; k[3] = k11 * det_inv;
}
static inline double bias_coef(double error_bias, double dt) { return 1.0 - pow(error_bias, dt); }

/* ------------------------------------------------------------ joints */

static void joint_prestep(World *w, Joint *j, double dt) {
    Body *a = &w->bodies[j->a], *b = &w->bodies[j->b];
    switch (j->type) {
    case J_PIVOT: {
        j->r1 = vrotate(a->rot, j->anchor_a); j->r2 = vrotate(b->rot, j->anchor_b);
        k_tensor(a, b, j->r1, j->r2, j->k);
        v2 delta = vsub(vadd(b->p, j->r2), vadd(a->p, j->r1));
        j->bias_v = vclamp(vmul(delta, -bias_coef(j->error_bias, dt) / dt), j->max_bias);
    } break;
    case J_GEAR: {
        double ratio = j->p1, ratio_inv = 1.0 / ratio;
        j->imass = 1.0 / (a->i_inv * ratio_inv + ratio * b->i_inv);
        j->bias = fclamp(-bias_coef(j->error_bias, dt) * (b->a * ratio - a->a - j->p0) / dt, -j->max_bias, j->max_bias);
    } break;
    case J_SPRING: {
        double moment = a->i_inv + b->i_inv;
        j->imass = 1.0 / moment;
        j->w_coef = 1.0 - exp(-j->p2 * dt * moment);
        j->target_wrn = 0.0;
        double j_spring = ((a->a - b->a) - j->p0) * j->p1 * dt;
        j->jacc = j_spring;
        a->w -= j_spring * a->i_inv; b->w += j_spring * b->i_inv;
    } break;
    case J_PIN: {
        j->r1 = vrotate(a->rot, j->anchor_a); j->r2 = vrotate(b->rot, j->anchor_b);
        v2 delta = vsub(vadd(b->p, j->r2), vadd(a->p, j->r1));
        double dist = vlen(delta);
        j->n = vmul(delta, 1.0 / (dist ? dist : INFINITY));
        j->imass = 1.0 / k_scalar(a, b, j->r1, j->r2, j->n);
        j->bias = fclamp(-bias_coef(j->error_bias, dt) * (dist - j->p0) / dt, -j->max_bias, j->max_bias);
    } break;
    case J_LIMIT: {
        double dist = b->a - a->a, pdist = 0.0;
        if (dist > j->p1) pdist = j->p1 - dist; else if (dist < j->p0) pdist = j->p0 - dist;
        j->imass = 1.0 / (a->i_inv + b->i_inv);
        j->bias = fclamp(-bias_coef(j->error_bias, dt) * pdist / dt, -j->max_bias, j->max_bias);
        if (!j->bias) j->jacc = 0.0;
    } break;
    case J_MOTOR: j->imass = 1.0 / (a->i_inv + b->i_inv); break;
    }
}
static void joint_apply_cached(World *w, Joint *j, double dt_coef) {
    Body *a = &w->bodies[j->a], *b = &w->bodies[j->b];
    switch (j->type) {
    case J_PIVOT: apply_impulses(a, b, j->r1, j->r2, vmul(j->jacc_v, dt_coef)); break;
    case J_GEAR: { double jj = j->jacc * dt_coef; a->w -= jj * a->i_inv * (1.0 / j->p1); b->w += jj * b->i_inv; } break;
    case J_SPRING: break;
    case J_PIN: apply_impulses(a, b, j->r1, j->r2, vmul(j->n, j->jacc * dt_coef)); break;
    case J_LIMIT: case J_MOTOR: { double jj = j->jacc * dt_coef; a->w -= jj * a->i_inv; b->w += jj * b->i_inv; } break;
    }
}
static void joint_apply_impulse(World *w, Joint *j, double dt) {
    Body *a = &w->bodies[j->a], *b = &w->bodies[j->b];
    switch (j->type) {
    case J_PIVOT: {
        v2 vr = relative_velocity(a, b, j->r1, j->r2);
        v2 d = vsub(j->bias_v, vr);
        v2 jj = V(d.x * j->k[0] + d.y * j->k[1], d.x * j->k[2] + d.y * j->k[3]);
        v2 j_old = j->jacc_v;
        j->jacc_v = vclamp(vadd(j->jacc_v, jj), j->max_force * dt);
        jj = vsub(j->jacc_v, j_old);
        apply_impulses(a, b, j->r1, j->r2, jj);
    } break;
    case J_GEAR: {
        double ratio = j->p1, ratio_inv = 1.0 / ratio;
        double wr = b->w * ratio - a->w;
        double j_max = j->max_force * dt;
        double jj = (j->bias - wr) * j->imass;
        double j_old = j->jacc;
        j->jacc = fclamp(j_old + jj, -j_max, j_max);
        jj = j->jacc - j_old;
        a->w -= jj * a->i_inv * ratio_inv; b->w += jj * b->i_inv;
    } break;
    case J_SPRING: {
        double wrn = a->w - b->w;
        double w_damp = (j->target_wrn - wrn) * j->w_coef;
        j->target_wrn = wrn + w_damp;
        double j_damp = w_damp * j->imass;
        j->jacc += j_damp;
        a->w += j_damp * a->i_inv; b->w -= j_damp * b->i_inv;
    } break;
    case J_PIN: {
        v2 n = j->n;
        double vrn = vdot(relative_velocity(a, b, j->r1, j->r2), n);
        double jn_max = j->max_force * dt;
        double jn = (j->bias - vrn) * j->imass;
        double jn_old = j->jacc;
        j->jacc = fclamp(jn_old + jn, -jn_max, jn_max);
        jn = j->jacc - jn_old;
        apply_impulses(a, b, j->r1, j->r2, vmul(n, jn));
    } break;
    case J_LIMIT: {
        if (!j->bias) return;
        double wr = b->w - a->w;
        double j_max = j->max_force * dt;
        double jj = -(j->bias + wr) * j->imass;
        double j_old = j->jacc;
        if (j->bias < 0.0) j->jacc = fclamp(j_old + jj, 0.0, j_max);
        else j->jacc = fclamp(j_old + jj, -j_max, 0.0);
        jj = j->jacc - j_old;
        a->w -= jj * a->i_inv; b->w += jj * b->i_inv;
    } break;
    case J_MOTOR: {
        double wr = b->w - a->w + j->p0;
        double j_max = j->max_force * dt;
        double jj = -wr * j->imass;
        double j_old = j->jacc;
        j->jacc = fclamp(j_old + jj, -j_max, j_max);
        jj = j->jacc - j_old;
        a->w -= jj * a->i_inv; b->w += jj * b->i_inv;
    } break;
    }
}

/* ------------------------------------------------------------ cpSpaceStep */

void ref_space_step(World *w, double dt) {
    if (dt == 0.0) return;
    w->stamp++;
    double prev_dt = w->prev_dt; w->prev_dt = dt;

    /* arbiters that were active last step go back to NORMAL (cpSpaceStep top) */
    for (int i = 0; i < MAX_ARB; i++) if (w->arbs[i].used && w->arbs[i].active) { w->arbs[i].state = ARB_NORMAL; w->arbs[i].active = 0; }

    /* 1. integrate positions (cpBodyUpdatePosition) -- dynamic AND kinematic bodies */
    for (int i = 0; i < w->nbodies; i++) {
        Body *b = &w->bodies[i];
        if (b->type == BODY_STATIC) continue;
        b->p = vadd(b->p, vmul(vadd(b->v, b->v_bias), dt));
        body_set_angle(b, b->a + (b->w + b->w_bias) * dt);
        b->v_bias = V(0, 0); b->w_bias = 0.0;
    }
    /* 2. shape cache + collide.  Broadphase is an accelerator only: the pair set
     *    is decided by QueryReject on the true BBs, so all-pairs is equivalent.
     *    Arbiter order: Chipmunk uses BBTree traversal order (not reproducible
     *    without the tree); we fix ascending (shape_i, shape_j). */
    for (int i = 0; i < w->nshapes; i++) shape_update(w, &w->shapes[i]);
    for (int i = 0; i < w->nshapes; i++)
        for (int j = i + 1; j < w->nshapes; j++) collide_pair(w, i, j);
    w->norder = 0;
    for (int i = 0; i < w->nshapes; i++)
        for (int j = i + 1; j < w->nshapes; j++) {
            Arbiter *arb = arb_find(w, i, j);
            if (arb && arb->active) { w->order[w->norder++] = (int)(arb - w->arbs); }
        }
    if (g_unknowns & UNK_ARB_DESCENDING)
        for (int lo = 0, hi = w->norder - 1; lo < hi; lo++, hi--) { int tmp = w->order[lo]; w->order[lo] = w->order[hi]; w->order[hi] = tmp; }
    /* 3. cache filter (cpSpaceArbiterSetFilter) */
    for (int i = 0; i < MAX_ARB; i++) {
        Arbiter *arb = &w->arbs[i];
        if (!arb->used) continue;
        int ticks = w->stamp - arb->stamp;
        if (ticks >= 1 && arb->state != ARB_CACHED) arb->state = ARB_CACHED;
        if (ticks >= ((g_unknowns & UNK_PERSISTENCE_1) ? 1 : ((g_unknowns & UNK_PERSISTENCE_5) ? 5 : w->collision_persistence))) arb->used = 0;
    }
    /* 4. prestep arbiters then joints */
    double slop = w->collision_slop;
    double bias_c = 1.0 - pow(w->collision_bias, dt);
    for (int q = 0; q < w->norder; q++) {
        Arbiter *arb = &w->arbs[w->order[q]];
        Body *a = &w->bodies[w->shapes[arb->sa].body], *b = &w->bodies[w->shapes[arb->sb].body];
        v2 n = arb->n;
        v2 body_delta = vsub(b->p, a->p);
        for (int i = 0; i < arb->count; i++) {
            Contact *con = &arb->c[i];
            con->n_mass = 1.0 / k_scalar(a, b, con->r1, con->r2, n);
            con->t_mass = 1.0 / k_scalar(a, b, con->r1, con->r2, vperp(n));
            double dist = vdot(vadd(vsub(con->r2, con->r1), body_delta), n);
            con->bias = -bias_c * fmin(0.0, dist + slop) / dt;
            con->j_bias = 0.0;
            con->bounce = 0.0;    /* elasticity 0 everywhere */
        }
    }
    for (int i = 0; i < w->njoints; i++) joint_prestep(w, &w->joints[i], dt);
    /* 5. integrate velocities (cpBodyUpdateVelocity; gravity 0, no forces) */
    double damping = pow(w->damping, dt);
    for (int i = 0; i < w->nbodies; i++) {
        Body *b = &w->bodies[i];
        if (b->type != BODY_DYNAMIC) continue;
        b->v = vadd(vmul(b->v, damping), vmul(V(0, 0), dt));
        b->w = b->w * damping + 0.0 * b->i_inv * dt;
    }
    /* 6. warm start */
    double dt_coef = (prev_dt == 0.0 ? 0.0 : dt / prev_dt);
    for (int q = 0; q < w->norder; q++) {
        Arbiter *arb = &w->arbs[w->order[q]];
        if (arb->state == ARB_FIRST) continue;
        Body *a = &w->bodies[w->shapes[arb->sa].body], *b = &w->bodies[w->shapes[arb->sb].body];
        for (int i = 0; i < arb->count; i++) {
            Contact *con = &arb->c[i];
            v2 jj = vrotate(arb->n, V(con->jn_acc, con->jt_acc));
            apply_impulses(a, b, con->r1, con->r2, vmul(jj, dt_coef));
        }
    }
    for (int i = 0; i < w->njoints; i++) joint_apply_cached(w, &w->joints[i], dt_coef);
    /* 7. iterations */
    for (int it = 0; it < w->iterations; it++) {
        for (int q = 0; q < w->norder; q++) {
            Arbiter *arb = &w->arbs[w->order[q]];
            Body *a = &w->bodies[w->shapes[arb->sa].body], *b = &w->bodies[w->shapes[arb->sb].body];
            v2 n = arb->n; double friction = arb->u;
            for (int i = 0; i < arb->count; i++) {
                Contact *con = &arb->c[i];
                double n_mass = con->n_mass;
                v2 r1 = con->r1, r2 = con->r2;
                v2 vb1 = vadd(a->v_bias, vmul(vperp(r1), a->w_bias));
                v2 vb2 = vadd(b->v_bias, vmul(vperp(r2), b->w_bias));
                v2 vr = relative_velocity(a, b, r1, r2);
                double vbn = vdot(vsub(vb2, vb1), n);
                double vrn = vdot(vr, n);
                double vrt = vdot(vr, vperp(n));
                double jbn = (con->bias - vbn) * n_mass;
                double jbn_old = con->j_bias;
                con->j_bias = fmax(jbn_old + jbn, 0.0);
                double jn = -(con->bounce + vrn) * n_mass;
                double jn_old = con->jn_acc;
                con->jn_acc = fmax(jn_old + jn, 0.0);
                double jt_max = friction * con->jn_acc;
                double jt = -vrt * con->t_mass;
                double jt_old = con->jt_acc;
                con->jt_acc = fclamp(jt_old + jt, -jt_max, jt_max);
                v2 jb = vmul(n, con->j_bias - jbn_old);
                apply_bias_impulse(a, vneg(jb), r1); apply_bias_impulse(b, jb, r2);
                apply_impulses(a, b, r1, r2, vrotate(n, V(con->jn_acc - jn_old, con->jt_acc - jt_old)));
            }
        }
        for (int i = 0; i < w->njoints; i++) joint_apply_impulse(w, &w->joints[i], dt);
    }
}

/* ------------------------------------------------------------ env step */

/* entities.py:148-190: id = 9*[close] + 3*lr + ud */
void ref_set_action(World *w, int action) {
    int ud = action % 3, lr = (action / 3) % 3, grip = action / 9;
    double r = w->robot_radius;
    w->rel_turn_angle = 0.0; w->target_speed = 0.0;
    if (ud == 1) w->target_speed += 4.0 * r;
    if (ud == 2) w->target_speed -= 3.0 * r;
    if (lr == 1) w->rel_turn_angle += 1.5;
    if (lr == 2) w->rel_turn_angle -= 1.5;
    if (grip == 0) w->target_finger_angle = w->finger_rot_limit_outer;
    else w->target_finger_angle = -w->finger_rot_limit_inner;
}
/* entities.py:459-479 */
void ref_robot_update(World *w) {
    if (w->robot_body < 0) return;
    Body *rb = &w->bodies[w->robot_body], *cb = &w->bodies[w->control_body];
    body_set_angle(cb, rb->a + w->rel_turn_angle);
    cb->v = vrotate(rb->rot, V(0.0, w->target_speed));
    for (int f = 0; f < 2; f++) {
        double side = (f == 0 ? -1.0 : 1.0);
        Body *fb = &w->bodies[w->finger_body[f]];
        double rel_angle = fb->a - rb->a;
        double angle_error = rel_angle + side * w->target_finger_angle;
        double target_rate = fmax(-1.0, fmin(1.0, angle_error * 10));
        if (fabs(target_rate) < 1e-4) target_rate = 0.0;
        w->joints[w->finger_motor[f]].p0 = target_rate;
    }
}
/* base_env.py:236-243 -- one physics substep */
void ref_substep(World *w, double dt) { ref_robot_update(w); ref_space_step(w, dt); }
/* base_env.py:255-270 (physics part of step) */
void ref_step(World *w, int action, double fps) {
    ref_set_action(w, action);
    int phys_steps = 10;
    double spf = 1 / fps, dt = spf / phys_steps;
    for (int i = 0; i < phys_steps; i++) ref_substep(w, dt);
    w->episode_steps++;
}

/* ------------------------------------------------------------ state access */

int ref_nbodies(const World *w) { return w->nbodies; }
int ref_nshapes(const World *w) { return w->nshapes; }
/* body, ShapeFilter group and sensor flag of shape s (placement queries, geom.py:116-262) */
void ref_shape_info(const World *w, int s, int *out) { out[0] = w->shapes[s].body; out[1] = w->shapes[s].group; out[2] = w->shapes[s].sensor; }
int ref_njoints(const World *w) { return w->njoints; }
int ref_narbiters(const World *w) { return w->norder; }
/* out[nbodies][9] = x y a vx vy w vbx vby wb */
void ref_get_bodies(const World *w, double *out) {
    for (int i = 0; i < w->nbodies; i++) {
        const Body *b = &w->bodies[i]; double *o = out + 9 * i;
        o[0] = b->p.x; o[1] = b->p.y; o[2] = b->a; o[3] = b->v.x; o[4] = b->v.y; o[5] = b->w;
        o[6] = b->v_bias.x; o[7] = b->v_bias.y; o[8] = b->w_bias;
    }
}
void ref_set_bodies(World *w, const double *in) {
    for (int i = 0; i < w->nbodies; i++) {
        Body *b = &w->bodies[i]; const double *o = in + 9 * i;
        b->p = V(o[0], o[1]); body_set_angle(b, o[2]); b->v = V(o[3], o[4]); b->w = o[5];
        b->v_bias = V(o[6], o[7]); b->w_bias = o[8];
    }
}
void ref_get_body_mass(const World *w, double *out) {
    for (int i = 0; i < w->nbodies; i++) { out[2 * i] = w->bodies[i].m_inv; out[2 * i + 1] = w->bodies[i].i_inv; }
}
void ref_get_joint_acc(const World *w, double *out) {
    for (int i = 0; i < w->njoints; i++) {
        const Joint *j = &w->joints[i];
        out[2 * i] = (j->type == J_PIVOT ? j->jacc_v.x : j->jacc); out[2 * i + 1] = (j->type == J_PIVOT ? j->jacc_v.y : 0.0);
    }
}
/* active contacts of the last step: rows of [shape_a, shape_b, nx, ny, count, (p1x p1y p2x p2y jn jt hash) x2] = 19 */
int ref_get_contacts(const World *w, double *out, int max_rows) {
    int rows = 0;
    for (int q = 0; q < w->norder && rows < max_rows; q++) {
        const Arbiter *arb = &w->arbs[w->order[q]];
        const Body *a = &w->bodies[w->shapes[arb->sa].body], *b = &w->bodies[w->shapes[arb->sb].body];
        double *o = out + 19 * rows++;
        memset(o, 0, 19 * sizeof(double));
        o[0] = arb->sa; o[1] = arb->sb; o[2] = arb->n.x; o[3] = arb->n.y; o[4] = arb->count;
        for (int i = 0; i < arb->count; i++) {
            const Contact *c = &arb->c[i]; double *p = o + 5 + 7 * i;
            p[0] = a->p.x + c->r1.x; p[1] = a->p.y + c->r1.y; p[2] = b->p.x + c->r2.x; p[3] = b->p.y + c->r2.y;
            p[4] = c->jn_acc; p[5] = c->jt_acc; p[6] = c->hash;
        }
    }
    return rows;
}
/* stand-alone narrowphase on the CURRENT poses: same row layout as above, 1 row */
int ref_collide_shapes(World *w, int i, int j, double *out) {
    shape_update(w, &w->shapes[i]); shape_update(w, &w->shapes[j]);
    int sa = i, sb = j;
    if (w->shapes[sa].type > w->shapes[sb].type) { sa = j; sb = i; }
    CollisionInfo info; memset(&info, 0, sizeof(info));
    collide(&w->shapes[sa], &w->shapes[sb], &info, 0);
    memset(out, 0, 19 * sizeof(double));
    out[0] = sa; out[1] = sb; out[2] = info.n.x; out[3] = info.n.y; out[4] = info.count;
    for (int k = 0; k < info.count; k++) {
        double *p = out + 5 + 7 * k;
        p[0] = info.p1[k].x; p[1] = info.p1[k].y; p[2] = info.p2[k].x; p[3] = info.p2[k].y; p[6] = info.hash[k];
    }
    return info.count;
}
/* world-space verts of a shape at the current pose (for overlap queries in scoring) */
int ref_shape_world(World *w, int s, double *out_xy, double *out_r, int *out_type) {
    Shape *S = &w->shapes[s]; shape_update(w, S);
    for (int i = 0; i < S->n; i++) { out_xy[2 * i] = S->tv[i].x; out_xy[2 * i + 1] = S->tv[i].y; }
    *out_r = S->r; *out_type = S->type;
    return S->n;
}
int ref_episode_steps(const World *w) { return w->episode_steps; }

/* ------------------------------------------------------------ rasteriser */

/* world -> screen affine: s = M p + t (pixels, origin bottom-left, y up) */
typedef struct { double m00, m01, m10, m11, tx, ty; } Aff;
static inline v2 aff_apply(const Aff *A, v2 p) { return V(A->m00 * p.x + A->m01 * p.y + A->tx, A->m10 * p.x + A->m11 * p.y + A->ty); }

/* gym_render.py:176-200,372-377 + base_env.py:294-307.  view 0 = ego, 1 = allo */
static Aff camera(const World *w, int view, int res) {
    double arena = 2.0, zoom = 1.02;
    double world_w = arena * zoom, world_h = arena * zoom;
    double sx = res / world_w, sy = res / world_h;
    Aff A;
    if (view == 1) {
        double left = -1.0 * zoom, bottom = -1.0 * zoom;
        A.m00 = sx; A.m01 = 0; A.m10 = 0; A.m11 = sy; A.tx = -left * sx; A.ty = -bottom * sy;
    } else {
        const Body *rb = &w->bodies[w->robot_body];
        double c = cos(-rb->a), s = sin(-rb->a);
        double npx = world_w * 0.5, npy = world_h * 0.15;
        /* scale o translate(newpos) o rotate(-theta) o translate(-centre) */
        A.m00 = sx * c; A.m01 = -sx * s; A.m10 = sy * s; A.m11 = sy * c;
        A.tx = sx * (npx + (c * -rb->p.x - s * -rb->p.y));
        A.ty = sy * (npy + (s * -rb->p.x + c * -rb->p.y));
    }
    return A;
}

static v2 geom_vertex_world(const World *w, const Geom *G, v2 v) {
    if (G->xform == X_WORLD) return v;
    const Body *b = &w->bodies[G->body];
    if (G->xform == X_BODY) return vadd(b->p, vrotate(b->rot, v));
    /* X_EYE (entities.py:414-437,488-490): robot o T(eye_base) o R(eye.a - robot.a) o T(eye_pre) */
    double da = (G->eye_body >= 0) ? (w->bodies[G->eye_body].a - b->a) : 0.0;
    v2 q = vadd(v, G->eye_pre);
    q = vrotate(V(cos(da), sin(da)), q);
    q = vadd(q, G->eye_base);
    return vadd(b->p, vrotate(b->rot, q));
}

static inline unsigned char to_u8(double c) { double v = floor(c * 255.0 + 0.5); return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* distance from point to segment (pixels) */
static double seg_dist(v2 p, v2 a, v2 b, double *t_out) {
    v2 d = vsub(b, a); double L2 = vlensq(d);
    double t = L2 > 0 ? fclamp01(vdot(vsub(p, a), d) / L2) : 0.0;
    if (t_out) *t_out = t;
    return vlen(vsub(p, vadd(a, vmul(d, t))));
}

/* Render at `res` x `res` into out[res][res][3] (row 0 = top).
 * Filled polygons: point-sampled at pixel centres, painter's order, inclusive edges.
 * Line loops (GL_LINE_SMOOTH, blending on): OUR MODEL (driver-defined in GL):
 *   coverage alpha = clamp(0.5*(w+1) - dist_to_segment, 0, 1), optional 16-px on/off stipple by
 *   arclength along the loop, blended src-over and re-quantised to u8 per sample. */
void ref_render(const World *w, int view, int res, unsigned char *out) {
    Aff A = camera(w, view, res);
    unsigned char bg[3] = {to_u8(w->bg_rgb[0]), to_u8(w->bg_rgb[1]), to_u8(w->bg_rgb[2])};
    for (int i = 0; i < res * res; i++) { out[3 * i] = bg[0]; out[3 * i + 1] = bg[1]; out[3 * i + 2] = bg[2]; }
    static v2 sv[MAX_GVERTS];
    for (int g = 0; g < w->ngeoms; g++) {
        const Geom *G = &w->geoms[g];
        int n = G->nverts;
        double minx = INFINITY, maxx = -INFINITY, miny = INFINITY, maxy = -INFINITY;
        for (int i = 0; i < n; i++) {
            sv[i] = aff_apply(&A, geom_vertex_world(w, G, G->verts[i]));
            minx = fmin(minx, sv[i].x); maxx = fmax(maxx, sv[i].x); miny = fmin(miny, sv[i].y); maxy = fmax(maxy, sv[i].y);
        }
        unsigned char col[3] = {to_u8(G->rgb[0]), to_u8(G->rgb[1]), to_u8(G->rgb[2])};
        double pad = (G->kind == G_LINELOOP) ? (0.5 * (G->line_width + 1.0) + 1.0) : 0.0;
        int x0 = (int)floor(minx - pad - 0.5), x1 = (int)ceil(maxx + pad - 0.5);
        int y0 = (int)floor(miny - pad - 0.5), y1 = (int)ceil(maxy + pad - 0.5);
        if (x0 < 0) x0 = 0; if (y0 < 0) y0 = 0; if (x1 > res - 1) x1 = res - 1; if (y1 > res - 1) y1 = res - 1;
        if (G->kind == G_POLY) {
            /* orientation: accept either winding (GL fills both) */
            double area2 = 0; for (int i = 0; i < n; i++) area2 += vcross(sv[i], sv[(i + 1) % n]);
            double sgn = area2 >= 0 ? 1.0 : -1.0;
            for (int py = y0; py <= y1; py++) for (int px = x0; px <= x1; px++) {
                v2 p = V(px + 0.5, py + 0.5);
                int inside = 1;
                for (int i = 0; i < n && inside; i++) {
                    v2 a = sv[i], b = sv[(i + 1) % n];
                    double ef = sgn * vcross(vsub(b, a), vsub(p, a));
                    if (ef < 0.0 || ((g_unknowns & UNK_FILL_EXCLUSIVE) && ef == 0.0)) inside = 0;
                }
                if (inside) { unsigned char *o = out + 3 * ((res - 1 - py) * res + px); o[0] = col[0]; o[1] = col[1]; o[2] = col[2]; }
            }
        } else {
            double hw = 0.5 * (G->line_width + 1.0);
            for (int py = y0; py <= y1; py++) for (int px = x0; px <= x1; px++) {
                v2 p = V(px + 0.5, py + 0.5);
                double best = 0.0, arc = 0.0;
                for (int i = 0; i < n; i++) {
                    v2 a = sv[i], b = sv[(i + 1) % n]; double t;
                    double d = seg_dist(p, a, b, &t);
                    double alpha = fclamp01(hw - d);
                    double len = vlen(vsub(b, a));
                    if (alpha > 0 && G->stipple) {
                        double s = arc + t * len;
                        int bit = ((int)floor(s)) & 15;
                        if (!((G->stipple >> bit) & 1)) alpha = 0.0;
                    }
                    if (alpha > best) best = alpha;
                    arc += len;
                }
                if (best > 0.0) {
                    unsigned char *o = out + 3 * ((res - 1 - py) * res + px);
                    for (int c = 0; c < 3; c++) {
                        double v = best * (double)col[c] + (1.0 - best) * (double)o[c];
                        o[c] = (unsigned char)floor(v + 0.5);
                    }
                }
            }
        }
    }
}

/* cv2.resize(..., INTER_AREA) for an integer factor: exact block mean, cvRound (ties to even). */
void ref_area_downsample(const unsigned char *in, int res_in, int factor, int channels, unsigned char *out) {
    int res_out = res_in / factor; int area = factor * factor;
    for (int y = 0; y < res_out; y++) for (int x = 0; x < res_out; x++) for (int c = 0; c < channels; c++) {
        int sum = 0;
        for (int dy = 0; dy < factor; dy++) for (int dx = 0; dx < factor; dx++)
            sum += in[((y * factor + dy) * res_in + (x * factor + dx)) * channels + c];
        double v = (double)sum / (double)area;
        out[(y * res_out + x) * channels + c] = (unsigned char)((g_unknowns & UNK_RESIZE_HALF_UP) ? floor(v + 0.5) : nearbyint(v));
    }
}

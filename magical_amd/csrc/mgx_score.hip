// mgx_score.hip -- k_score: the goal-region overlap sets of score_on_end_of_traj(), on the device.
//
// GoalRegion.get_overlapping_ents(com_overlap=True) (entities.py:821-881): an entity counts for a region iff EVERY one of its
// collision shapes overlaps the sensor rectangle (space.shape_query -> cpShapesCollide(...).count > 0) AND its body position
// lies inside the sensor's bounding box (bb.contains_vect).  MoveToRegion's score (move_to_region.py:85-94) is the second
// test alone, for the robot.  Both are booleans of the final poses, so they are evaluated here, once per finished episode,
// straight from the pose blob: one byte per (region, entity, env) goes to the host instead of the poses, and the tasks' float
// arithmetic (match_regions.py:193-213, find_dupe.py:203-216, fix_colour.py:193-202) runs on those booleans unchanged.
//
// Geometry in fp64 whatever the engine's dtype: the block shapes come from an fp64 library (one entry per shape type; all
// blocks have the same size), not from the step kernel's fp32 template.  The separating-axis arithmetic is, operation for
// operation, that of the host restatement magical_amd/benchmarks/_scoring.py (_shape_hits_box) -- contraction off, so that
// the two agree bit for bit up to the last-place differences of sin / cos.
//
// One thread per env; state rows are [row][env], so the loads of a wavefront coalesce.  No MFMA (no contraction anywhere).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mgx {

constexpr int SC_TYPES = 7, SC_MAX_PARTS = 8, SC_MAX_VERTS = 8;

struct ScoreLib {     // collision shapes of a block of shape type t (entities.py:614-711): convex parts, local fp64 vertices
    int32_t n_parts[SC_TYPES];
    int32_t kind[SC_TYPES][SC_MAX_PARTS];      // 0 circle, 2 polygon
    int32_t nv[SC_TYPES][SC_MAX_PARTS];
    double radius[SC_TYPES][SC_MAX_PARTS];     // circle radius / polygon bevel radius
    double xy[SC_TYPES][SC_MAX_PARTS][2 * SC_MAX_VERTS];
};

struct ScoreDev {
    const ScoreLib *lib;
    const int32_t *ent;             // [n_entities][4] of the engine's world: kind (0 robot, 1 shape, 2 goal), main body, shape type, present
    const int32_t *body_prow;       // [n_bodies][3]: pose-blob row of x, y, angle (-1: not persistent)
    const int32_t *goal_ent;        // [n_goals] entity index of every goal region, in entity order
    const double *goal_xyhw;        // [n_goals][4] the world's own rectangles: x, y (top-left), h, w (entities.py:769-797)
    const double *goal_xyhw_env;    // [n_goals * 4][N] per-env rectangles (Test*Jitter / Layout) or NULL
    const int8_t *ent_type_env;     // [n_entities][N] per-env shape types (per-env worlds) or NULL
    const uint8_t *ent_present_env; // [n_entities][N] per-env presence or NULL
    int n_entities, n_goals;
};

enum { SC_COM_INSIDE = 1, SC_SHAPES_OVERLAP = 2 };

// does the convex part (at x, y with rotation c, s) overlap the box l b r t?  Minimum separation of the cores <= radius.
__device__ inline bool part_hits_box(int kind, double radius, int nv, const double *v, double x, double y, double c, double s,
                                     double l, double b, double r, double t) {
#pragma clang fp contract(off)
    if (kind == 0) {
        const double dx = fmax(fmax(l - x, 0.0), x - r), dy = fmax(fmax(b - y, 0.0), y - t);
        return dx * dx + dy * dy <= radius * radius;
    }
    double wx[SC_MAX_VERTS], wy[SC_MAX_VERTS];
    double xmin = 0, xmax = 0, ymin = 0, ymax = 0;
    for (int i = 0; i < nv; i++) {
        const double vx = v[2 * i], vy = v[2 * i + 1];
        wx[i] = x + (c * vx - s * vy); wy[i] = y + (c * vy + s * vx);
        if (i == 0) { xmin = xmax = wx[0]; ymin = ymax = wy[0]; }
        else { xmin = fmin(xmin, wx[i]); xmax = fmax(xmax, wx[i]); ymin = fmin(ymin, wy[i]); ymax = fmax(ymax, wy[i]); }
    }
    double sep = fmax(fmax(fmax(l - xmax, xmin - r), b - ymax), ymin - t);
    for (int i = 0; i < nv; i++) {
        const int j = i + 1 == nv ? 0 : i + 1;
        const double ex = wx[j] - wx[i], ey = wy[j] - wy[i];
        const double ln = sqrt(ex * ex + ey * ey);
        const double nx = ey / ln, ny = -ex / ln;            // outward normal of a CCW polygon
        const double d0 = nx * (l - wx[i]) + ny * (b - wy[i]), d1 = nx * (r - wx[i]) + ny * (b - wy[i]);
        const double d2 = nx * (r - wx[i]) + ny * (t - wy[i]), d3 = nx * (l - wx[i]) + ny * (t - wy[i]);
        sep = fmax(sep, fmin(fmin(fmin(d0, d1), d2), d3));
    }
    return sep <= radius;
}

template <typename P>
__global__ __launch_bounds__(64) void k_score(ScoreDev s, const P *__restrict__ sp, const uint8_t *__restrict__ mask,
                                              uint8_t *__restrict__ out, int n_envs) {
#pragma clang fp contract(off)
    const long env = (long)blockIdx.x * 64 + threadIdx.x;
    if (env >= n_envs) return;
    const long N = n_envs;
    const bool live = !mask || mask[env];
    for (int g = 0; g < s.n_goals; g++) {
        double gx, gy, gh, gw;
        if (s.goal_xyhw_env) {
            gx = s.goal_xyhw_env[(long)(4 * g) * N + env]; gy = s.goal_xyhw_env[(long)(4 * g + 1) * N + env];
            gh = s.goal_xyhw_env[(long)(4 * g + 2) * N + env]; gw = s.goal_xyhw_env[(long)(4 * g + 3) * N + env];
        } else {
            gx = s.goal_xyhw[4 * g]; gy = s.goal_xyhw[4 * g + 1]; gh = s.goal_xyhw[4 * g + 2]; gw = s.goal_xyhw[4 * g + 3];
        }
        // GoalRegion.setup (entities.py:794-797): body at (x + w/2, y - h/2), box (w, h)
        const double cx = gx + gw / 2, cy = gy - gh / 2, hw = gw / 2, hh = gh / 2;
        const double l = cx - hw, b = cy - hh, r = cx + hw, t = cy + hh;
        const int ge = s.goal_ent[g];
        const bool goal_present = !s.ent_present_env || s.ent_present_env[(long)ge * N + env];
        for (int e = 0; e < s.n_entities; e++) {
            uint8_t flags = 0;
            const int kind = s.ent[4 * e], body = s.ent[4 * e + 1];
            const bool present = s.ent_present_env ? s.ent_present_env[(long)e * N + env] != 0 : s.ent[4 * e + 3] != 0;
            if (live && goal_present && present && kind != 2 && body >= 0) {
                const int rx = s.body_prow[3 * body], ry = s.body_prow[3 * body + 1], ra = s.body_prow[3 * body + 2];
                const double x = (double)sp[(long)rx * N + env], y = (double)sp[(long)ry * N + env];
                if (l <= x && r >= x && b <= y && t >= y) flags |= SC_COM_INSIDE;          // bb.contains_vect(body.position)
                if (kind == 1) {
                    const int ty = s.ent_type_env ? (int)s.ent_type_env[(long)e * N + env] : s.ent[4 * e + 2];
                    const int np = ty >= 0 && ty < SC_TYPES ? s.lib->n_parts[ty] : 0;
                    const double a = (double)sp[(long)ra * N + env];
                    const double c = cos(a), sn = sin(a);
                    bool all = np > 0;
                    for (int p = 0; p < np && all; p++)
                        all = part_hits_box(s.lib->kind[ty][p], s.lib->radius[ty][p], s.lib->nv[ty][p], s.lib->xy[ty][p], x, y, c, sn, l, b, r, t);
                    if (all) flags |= SC_SHAPES_OVERLAP;
                }
            }
            out[((long)g * s.n_entities + e) * N + env] = flags;
        }
    }
}

// per-env entity tables: row e of env env_idx[k] <- src[k][e]
__global__ void k_scatter_ent_rows(int8_t *type_tab, uint8_t *present_tab, const int8_t *src_type, const uint8_t *src_present,
                                   const int32_t *env_idx, int n_entities, long n_envs) {
    const long k = blockIdx.x;
    const long env = env_idx ? env_idx[k] : k;
    for (int e = threadIdx.x; e < n_entities; e += blockDim.x) {
        type_tab[(long)e * n_envs + env] = src_type[env_idx ? k * n_entities + e : e];
        present_tab[(long)e * n_envs + env] = src_present[env_idx ? k * n_entities + e : e];
    }
}

}  // namespace mgx

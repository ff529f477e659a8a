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

// ---------------------------------------------------------------- the point tasks' score_on_end_of_traj(), whole, on the device
// MoveToCorner (move_to_corner.py:66-75), MakeLine (make_line.py:31-71,142-152), ClusterColour / ClusterShape (cluster.py:166-216)
// score from block POSITIONS alone.  fp64, contraction off, every operation in numpy's order, so the result is the reference's
// bit for bit (tests: the reference's own outputs in tests/golden/reference_vectors.json).  Two of numpy's primitives go through
// library kernels whose use of FMA depends on the host's BLAS / numpy build -- np.linalg.norm of a 1-D vector (BLAS ddot) and
// `offs @ unit` (matmul's small-matrix kernel): the host finds out which form its numpy takes (benchmarks/_scoring.py probes
// exactly rounded candidates against the live primitives) and hands the answer down as dot_mode / mm_mode.
constexpr int SP_MAX_BLOCKS = 16, SP_MAX_CLASSES = 8;
enum { SP_CORNER = 1, SP_LINE = 2, SP_CLUSTER = 3 };
enum { SP_DOT_PLAIN = 0, SP_DOT_FMA_SECOND = 1, SP_DOT_FMA_FIRST = 2 };      // x*x + y*y | fma(y, y, x*x) | fma(x, x, y*y)

struct ScorePointsDev {
    int task, n;                        // blocks of the task, in the task's order
    int32_t ent[SP_MAX_BLOCKS];         // their entity indices (presence table) ...
    int32_t row_x[SP_MAX_BLOCKS], row_y[SP_MAX_BLOCKS];      // ... and the pose-blob rows of their positions
    int32_t cls_default[SP_MAX_BLOCKS]; // cluster: class of every block in the world's own layout
    const int8_t *cls_env;              // cluster: [N][n] per-env classes (Test* variants that redraw them) or NULL
    const uint8_t *ent_present_env;     // [n_entities][N] or NULL (all present)
    int n_classes, dot_mode, mm_mode;
    double p0, p1, p2;                  // corner: furthest distance, range | line: inlier distance, max separation
};

__device__ inline double sp_dot2(double ax, double ay, double bx, double by, int mode) {
#pragma clang fp contract(off)
    if (mode == SP_DOT_FMA_SECOND) return __builtin_fma(ay, by, ax * bx);
    if (mode == SP_DOT_FMA_FIRST) return __builtin_fma(ax, bx, ay * by);
    return ax * bx + ay * by;
}

// The per-env tables (positions, presence, classes, projections) are indexed by loop variables: as local arrays they lived in scratch
// memory (round 3: 5 120 spilled registers, 2.8 KB of private segment per lane); they are columns of LDS now, [k][lane], and the
// inliers' projections are sorted by a fixed compare-exchange network over registers (static indices).
#define SP_CE(a, b) { const double lo_ = fmin(v[a], v[b]), hi_ = fmax(v[a], v[b]); v[a] = lo_; v[b] = hi_; }
// Batcher's odd-even merge sort of 16 values (63 compare-exchanges); +inf pads sort to the end
__device__ __forceinline__ void sp_sort16(double (&v)[SP_MAX_BLOCKS]) {
#pragma unroll
    for (int p = 1; p < SP_MAX_BLOCKS; p <<= 1)
#pragma unroll
        for (int k = p; k >= 1; k >>= 1)
#pragma unroll
            for (int j = k % p; j + k < SP_MAX_BLOCKS; j += 2 * k)
#pragma unroll
                for (int i = 0; i < k; i++)
                    if (i + j + k < SP_MAX_BLOCKS && (i + j) / (2 * p) == (i + j + k) / (2 * p)) SP_CE(i + j, i + j + k)
}
#undef SP_CE

template <typename P>
__global__ __launch_bounds__(64) void k_score_points(ScorePointsDev s, const P *__restrict__ sp, const uint8_t *__restrict__ mask,
                                                     double *__restrict__ out, int n_envs) {
#pragma clang fp contract(off)
    __shared__ double l_px[SP_MAX_BLOCKS][64], l_py[SP_MAX_BLOCKS][64], l_cx[SP_MAX_CLASSES][64], l_cy[SP_MAX_CLASSES][64], l_cn[SP_MAX_CLASSES][64];
    __shared__ int8_t l_cls[SP_MAX_BLOCKS][64];
    __shared__ uint8_t l_on[SP_MAX_BLOCKS][64];
    const int t = threadIdx.x;
    const long env = (long)blockIdx.x * 64 + t;
    if (env >= n_envs) return;
    const long N = n_envs;
    if (mask && !mask[env]) { out[env] = 0.0; return; }
    int n_on = 0;
    for (int k = 0; k < s.n; k++) {
        l_px[k][t] = (double)sp[(long)s.row_x[k] * N + env]; l_py[k][t] = (double)sp[(long)s.row_y[k] * N + env];
        const bool on = !s.ent_present_env || s.ent_present_env[(long)s.ent[k] * N + env] != 0;
        l_on[k][t] = on ? 1 : 0;
        n_on += on ? 1 : 0;
    }
    double score = 0.0;
    if (s.task == SP_CORNER) {
        // dist = np.linalg.norm((-1, 1) - shape_pos); score = min(1, max(0, furthest - dist) / range)
        const double dx = -1.0 - l_px[0][t], dy = 1.0 - l_py[0][t];
        const double dist = sqrt(sp_dot2(dx, dy, dx, dy, s.dot_mode));
        score = fmin(1.0, fmax(0.0, s.p0 - dist) / s.p1);
    } else if (s.task == SP_LINE) {
        // the episode's blocks are the first n_on of the list (make_line.py:100-102); longest_line over their positions
        const int npts = n_on;
        int best = npts < 1 ? npts : 1;
        for (int i = 0; i + 1 < npts; i++)
            for (int j = i + 1; j < npts; j++) {
                const double pix = l_px[i][t], piy = l_py[i][t];
                const double jx = l_px[j][t] - pix, jy = l_py[j][t] - piy;
                const double nrm = sqrt(sp_dot2(jx, jy, jx, jy, s.dot_mode));
                const double ux = jx / nrm, uy = jy / nrm;
                double proj[SP_MAX_BLOCKS];
                int n_in = 0;
#pragma unroll
                for (int p = 0; p < SP_MAX_BLOCKS; p++) {
                    proj[p] = __builtin_inf();
                    if (p < npts) {
                        const double ox = l_px[p][t] - pix, oy = l_py[p][t] - piy;
                        const double pl = sp_dot2(ox, oy, ux, uy, s.mm_mode);
                        const double ex = ox - pl * ux, ey = oy - pl * uy;
                        const double d = sqrt(ex * ex + ey * ey);
                        if (d <= s.p0) { proj[p] = pl; n_in++; }            // (NaN: coincident points -> no inliers)
                    }
                }
                if (n_in <= best) continue;
                // np.sort of the inliers' projections, then the longest run of neighbours at most max_separation apart
                sp_sort16(proj);
                int run = 0, longest = 0;
#pragma unroll
                for (int k = 0; k + 1 < SP_MAX_BLOCKS; k++) {
                    if (k + 1 < n_in) {
                        run = fabs(proj[k + 1] - proj[k]) <= s.p1 ? run + 1 : 0;
                        longest = longest > run ? longest : run;
                    }
                }
                if (longest + 1 > best) best = longest + 1;
            }
        const int max_len = npts, min_len = (npts - 2) > 2 ? (npts - 2) : 2;
        const int num = best - min_len > 0 ? best - min_len : 0;
        score = (double)num / (double)(max_len - min_len);
    } else {
        // centroid of every class = sum of its members in block order / their number -- (0, 0) for a class without members, as the
        // reference has it (cluster.py:173-175); a block is correct when it is closer to its own centroid than to the nearest
        // other one by the margin (a squared distance: reference quirk, cluster.py:203-206)
        for (int c = 0; c < s.n_classes; c++) { l_cx[c][t] = 0.0; l_cy[c][t] = 0.0; l_cn[c][t] = 0.0; }
        for (int k = 0; k < s.n; k++) {
            const int c = s.cls_env ? (int)s.cls_env[env * s.n + k] : s.cls_default[k];
            l_cls[k][t] = (int8_t)c;
            if (!l_on[k][t] || c < 0 || c >= s.n_classes) continue;
            l_cx[c][t] += l_px[k][t]; l_cy[c][t] += l_py[k][t]; l_cn[c][t] += 1.0;
        }
        for (int c = 0; c < s.n_classes; c++) {
            const double cn = l_cn[c][t];
            l_cx[c][t] = cn > 0.0 ? l_cx[c][t] / cn : 0.0; l_cy[c][t] = cn > 0.0 ? l_cy[c][t] / cn : 0.0;
        }
        int n_correct = 0;
        for (int k = 0; k < s.n; k++) {
            double true_sse = 0.0, bad = __builtin_inf();
            bool bad_nan = false;
            const double x = l_px[k][t], y = l_py[k][t];
            const int ck = l_cls[k][t];
            for (int c = 0; c < s.n_classes; c++) {
                const double dx = x - l_cx[c][t], dy = y - l_cy[c][t];
                const double sse = dx * dx + dy * dy;
                if (c == ck) true_sse = sse;
                else if (sse != sse) bad_nan = true;         // np.min propagates a NaN
                else bad = sse < bad ? sse : bad;
            }
            const double margin = 2.0 * true_sse;
            const bool ok = !bad_nan && sqrt(true_sse) < sqrt(bad) - margin;
            n_correct += (ok && l_on[k][t]) ? 1 : 0;
        }
        const double frac = (double)n_correct / (double)(n_on > 1 ? n_on : 1);
        score = fmax(frac - 0.75, 0.0) / (1.0 - 0.75);
    }
    out[env] = score;
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

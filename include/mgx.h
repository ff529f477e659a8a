/*
 * mgx.h -- C ABI of libmagical_hip.so: the MI355X-native batched replacement for
 * MAGICAL's pymunk/pyglet-backed BaseEnv.step() hot path.
 *
 * The reference (qxcv/magical, pure Python) has no FFI of its own; its hot path
 * sits on the pymunk object API (downward) and the Gym env protocol (upward),
 * SURVEY.md §8b.  Each entry point below names the reference interface it
 * replaces.  Conventions:
 *   - plain pointers and sizes only; device buffers are OWNED BY THE CALLER
 *     (PyTorch tensors in the shipped host) and only borrowed for the call;
 *   - every function returns 0 on success or a negative mgx_status;
 *     mgx_last_error() returns a thread-local message for the last failure;
 *   - all device work is enqueued on the caller-supplied hipStream_t (passed
 *     as void*); nothing synchronises the host unless the name says _sync;
 *   - one mgx_engine per GPU per world template; one host thread per engine.
 */
#ifndef MGX_H_
#define MGX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mgx_world mgx_world;    /* host-side world description (entity list -> template) */
typedef struct mgx_engine mgx_engine;  /* per-GPU batch of N envs stepping one template */

enum mgx_status {
    MGX_OK = 0,
    MGX_ERR_ARG = -1,        /* bad argument / out of range */
    MGX_ERR_CAPACITY = -2,   /* world too large for the compiled capacities */
    MGX_ERR_STATE = -3,      /* call order violated (e.g. add after finalize) */
    MGX_ERR_HIP = -4,        /* HIP runtime error (message has hipGetErrorString) */
    MGX_ERR_NO_DEVICE = -5,  /* no gfx950 device visible */
};

/* entities.py:545-554 ShapeType order / :557-562 ShapeColour order */
enum mgx_shape_type { MGX_TRIANGLE = 0, MGX_SQUARE, MGX_PENTAGON, MGX_HEXAGON, MGX_OCTAGON, MGX_CIRCLE, MGX_STAR };
enum mgx_colour { MGX_RED = 0, MGX_GREEN, MGX_BLUE, MGX_YELLOW };
enum mgx_dtype { MGX_F32 = 0, MGX_F64 = 1 };
enum mgx_view { MGX_VIEW_EGO = 0, MGX_VIEW_ALLO = 1 };
/* output layouts of mgx_engine_render */
enum mgx_obs_layout {
    MGX_OBS_FRAME = 0,   /* u8[N][96][96][3]: just the new frame (caller owns any ring) */
    MGX_OBS_STACK4 = 1,  /* u8[N][96][96][12]: FlattenFrameStack semantics, oldest first; shifted in place */
    /* the two halves of LoRes3EA = FlattenFrameStack({'allo': 1, 'ego': 3}) (benchmarks/__init__.py:219-251), both on
     * the same u8[N][96][96][12] tensor, one launch per view: */
    MGX_OBS_STACK3_HI = 2,  /* channels 3..11: depth-3 stack of this view, shifted in place; channels 0..2 untouched */
    MGX_OBS_SLOT_LO = 3,    /* channels 0..2 <- this view's new frame; channels 3..11 untouched */
    /* u8[3][96][96] per env: the new frame as three channel planes, nothing else read or written.  `out` + env * env_stride
     * addresses the frame's slot in a caller-owned ring u8[N][R][3][96][96]: with frames kept as planes the channels-first stack
     * of the 4 newest frames (LoResCHW4E, benchmarks/__init__.py:262-268) is a contiguous window u8[12][96][96] of the ring, so a
     * step writes 27.6 KB per env (SURVEY.md 8(d)'s headline row) instead of re-materialising the 110 KB stack */
    MGX_OBS_PLANAR = 4,
};

const char *mgx_last_error(void);
int mgx_version(void);

/* ---- world description: replaces BaseEnv.reset()'s entity construction -------------------
 * base_env.py:177-234 (new pm.Space, collision_slop=0.01, iterations=phys_iter, arena added
 * first) + entities.py Entity.setup() for each entity, in add_entities() order. */
int mgx_world_create(mgx_world **out);
void mgx_world_destroy(mgx_world *w);
/* base_env.py:49-57 PhysicsVariables: robot_pos, robot_rot, robot_finger, shape_trans, shape_rot */
int mgx_world_set_phys_vars(mgx_world *w, const double vars[5]);
/* entities.py:217-490 Robot(radius=0.2, init_pos, init_angle, mass=1.0) */
int mgx_world_add_robot(mgx_world *w, double x, double y, double angle);
/* entities.py:584-761 Shape(shape_type, colour, shape_size=0.12, init_pos, init_angle, mass=0.5); returns entity id */
int mgx_world_add_shape(mgx_world *w, int shape_type, int colour, double x, double y, double angle);
/* entities.py:769-886 GoalRegion(x, y, h, w, colour); returns entity id */
int mgx_world_add_goal(mgx_world *w, double x, double y, double h, double w_, int colour);
/* freeze the entity list and build bodies / shapes / joints / collision pairs / draw list */
int mgx_world_finalize(mgx_world *w, int max_episode_steps);
/* a finalized copy of `w` with other per-episode choices (Test*Shape / CountPlus / All): entity k present iff
 * enabled[k] (NULL: all as in w), blocks of shape type shape_types[k] (NULL or < 0: as in w).  The robot cannot be
 * absent.  Entity and body indices are those of `w`. */
int mgx_world_variant(const mgx_world *w, const uint8_t *enabled, const int32_t *shape_types, mgx_world **out);

/* introspection (used by the host for scoring and by the parity tests) */
enum mgx_info_key {
    MGX_INFO_N_BODIES = 0, MGX_INFO_N_SHAPES, MGX_INFO_N_JOINTS, MGX_INFO_N_PAIRS, MGX_INFO_N_PRIMS,
    MGX_INFO_STATE_ROWS_P, MGX_INFO_STATE_ROWS_F, MGX_INFO_STATE_ROWS_I, MGX_INFO_ROBOT_BODY, MGX_INFO_N_ENTITIES,
    MGX_INFO_CACHE_SLOTS, MGX_INFO_MAX_CONTACTS, MGX_INFO_MAX_EPISODE_STEPS, MGX_INFO_N_JACC,
    /* rand_dynamics (base_env.py:198-203, phys_vars.py): rows PHYSVAR_ROW .. +4 of the motion blob hold each env's
     * five joint force limits as max impulse per substep = max_force x (1 / fps / phys_steps); mgx_engine_reset()
     * writes the world's defaults, the host overwrites them for the envs it has sampled PhysicsVariables for */
    MGX_INFO_PHYSVAR_ROW,
};
int mgx_world_info(const mgx_world *w, int key, int *out);
/* body index of entity `ent` (shape body; -1 for goals) and its type/colour */
int mgx_world_entity(const mgx_world *w, int ent, int *kind, int *body, int *shape_type, int *colour);
/* out[n_bodies][2] = (1/m, 1/I); out_init[n_bodies][3] = initial (x,y,angle) */
int mgx_world_body_table(const mgx_world *w, double *mass_inv, double *init_pose);
/* k-th persistent body component (k < rows_p + motion body rows): component comp (0..8 = x y a vx vy w vbx vby wb)
 * of body `body` lives in row `row` of the pose blob (comp < 3) or of the motion blob (comp >= 3) */
int mgx_world_state_entry(const mgx_world *w, int k, int *body, int *comp, int *row);
int mgx_world_n_state_entries(const mgx_world *w);
/* goal region `ent`: sensor box (l, b, r, t) */
int mgx_world_goal_bb(const mgx_world *w, int ent, double bb[4]);
/* shapes of entity `ent`: count; per shape kind (0 circle, 2 poly), radius, nverts and local verts xy[nverts*2] */
int mgx_world_entity_shapes(const mgx_world *w, int ent, int max_shapes, int *kinds, double *radii, int *nverts, double *xy, int xy_stride);

/* draw list: per primitive its template colour (r | g << 8 | b << 16), the entity whose colour paints it (-1: none) and
 * how (0 darkened, 1 base, 2 lightened twice: entities.py:712-757,807-819); returns the number of primitives */
int mgx_world_prim_table(const mgx_world *w, int *rgb, int *ent, int *role);
/* geom.py:116-262 pm_randomise_pose's rejection test, on the host: with entity e at poses[3e..3e+2] = (x, y, angle)
 * (goals: their box centre), does entity `ent` touch the arena walls or a shape of an entity with enabled[e] != 0
 * (space.shape_query of each of its shapes, i.e. cpCollide(...).count > 0, ShapeFilter groups honoured)?  1 / 0.
 * ent_hw: NULL, or [n_entities][2] = this env's (h, w) of every goal region (geom.py:344-360 randomise_hw) */
int mgx_world_placement_collides(const mgx_world *w, int ent, const double *poses, const uint8_t *enabled, const double *ent_hw);
/* geom.py:285-341 pm_randomise_all_poses for the entities ents[0..n) of ONE env, natively: poses [n_entities][3] in/out;
 * ignore[n_entities] (or NULL); per listed entity rand_pos / rand_rot flags and position / rotation limits (< 0: none).
 * The draws come from the np.random.RandomState stream handed over as its MT19937 state (key[624], pos) and are the
 * ones RandomState.uniform would make (x, y, angle per attempt); the state is advanced in place.  Returns the number
 * of rejected attempts, or MGX_ERR_CAPACITY when placement fails 10 times over */
int mgx_world_randomise_all_poses(const mgx_world *w, double *poses, const int *ents, int n, const uint8_t *ignore,
                                  const double arena_lrbt[4], const uint8_t *rand_pos, const uint8_t *rand_rot,
                                  const double *pos_limits, const double *rot_limits, uint32_t *mt_key, int *mt_pos, const double *ent_hw);
/* the same for m envs in one call: poses [m][n_entities][3]; mt_state_addr[k] = address of env k's live numpy
 * mt19937_state { uint32 key[624]; int pos; } (RandomState._bit_generator.ctypes.state_address), advanced in place */
int mgx_world_randomise_all_poses_batch(const mgx_world *w, int m, double *poses, const int *ents, int n, const uint8_t *ignore,
                                        const double arena_lrbt[4], const uint8_t *rand_pos, const uint8_t *rand_rot,
                                        const double *pos_limits, const double *rot_limits, int limits_per_env /* limits are [m][n] */,
                                        const uint64_t *mt_state_addr, const double *ent_hw /* NULL or [m][n_entities][2] */);
/* the other per-episode draws of the tasks' on_reset() (counts, colours, shape types, region sizes: e.g. cluster.py:81-110,
 * match_regions.py:51-117, find_dupe.py:84-112, fix_colour.py:78-113, make_line.py:100-110, base_env.py:198-203), for m envs
 * per call, each on its own live np.random.RandomState stream (mt_state_addr[k] as above), advanced exactly as numpy would:
 *   bounded: counts[k] (NULL: `count`) draws of rng.randint(0, max_inclusive + 1) -- what rng.choice(seq) indexes with --
 *            into out[k][..] (row stride out_stride);
 *   doubles: likewise rng.random_sample() (rng.uniform(lo, hi) = lo + (hi - lo) * it);
 *   shuffle: the permutation rng.shuffle() applies to a list of n_items[k] items: shuffled[i] = original[perm[k][i]].
 * The m addresses of one call must be distinct (one stream per env): batches of 1024 envs or more are spread over host threads. */
int mgx_rng_bounded_batch(int m, const uint64_t *mt_state_addr, const int32_t *counts, int count, int max_inclusive, int32_t *out, int out_stride);
int mgx_rng_doubles_batch(int m, const uint64_t *mt_state_addr, const int32_t *counts, int count, double *out, int out_stride);
int mgx_rng_shuffle_batch(int m, const uint64_t *mt_state_addr, const int32_t *n_items, int32_t *perm, int perm_stride);
/* style.py:28-37 evaluated to RGB8 for entity colour 0..3 (red green blue yellow) in `role` */
int mgx_world_palette(int colour, int role);

/* ---- engine: replaces BaseEnv.step()/render() for N envs ---------------------------------
 * Per-env persistent state lives in three caller-owned DEVICE blobs, all [rows][N] with the env
 * index fastest (coalesced lane<->env access):
 *   state_p : pose blob   (x, y, angle rows)        element = pose type   (f64 for MGX_F32 and MGX_F64)
 *   state_f : motion blob (velocities, joint and contact impulse accumulators) element = f32 / f64
 *   state_i : int32 blob  (episode step counter, contact-cache headers)
 * dtype: MGX_F32 = fp32 velocities/impulses/contacts with fp64 poses (the shipped engine: the
 * reference's zero-length pin joints difference nearly equal world positions), MGX_F64 = all fp64
 * (validation), MGX_F32_PURE = all fp32 (ablation).
 * lanes_per_env: lanes of a wavefront that step one env together: 0 = the engine's choice, -1 = its choice for an engine whose env-steps
 * are rendered (mgx_engine_step_render: the crowded worlds then keep 16 lanes, which is slower for k_step alone), else 16, 32 or 64 (whole DPP
 * rows: the solver keeps the robot's joint j on lane j of each row); the result does not depend on it.  Other values (4 and 8 were
 * accepted up to round 2) return MGX_ERR_ARG; a world with more than 15 blocks returns MGX_ERR_CAPACITY (each block's joints take a
 * lane of the group's first row; every task of the reference has at most 10). */
enum mgx_engine_dtype { MGX_F32_PURE = 2 };
int mgx_engine_create(const mgx_world *w, int n_envs, int device, int dtype, int lanes_per_env, mgx_engine **out);
void mgx_engine_destroy(mgx_engine *e);
/* rows of the three blobs, element sizes in bytes of state_p / state_f */
int mgx_engine_state_shape(const mgx_engine *e, int *rows_p, int *rows_f, int *rows_i, int *size_p, int *size_f);
int mgx_engine_lanes_per_env(const mgx_engine *e);
int mgx_engine_lds_bytes(const mgx_engine *e, int which);   /* 0 = step kernel, 1 = raster kernel */
/* BaseEnv.reset() (base_env.py:177-234): template state -> envs where mask[i] != 0 (mask NULL = all).
 * mask is a DEVICE pointer to u8[N]. */
int mgx_engine_reset(mgx_engine *e, void *state_p, void *state_f, int32_t *state_i, const uint8_t *mask, void *stream);
/* the same with per-env initial entity poses (Test*Jitter / TestLayout variants; geom.py:116-341 picks them on the host):
 * ent_pose = DEVICE [n_entities * 3][N] (x, y, angle per entity, element type = the pose type of state_p); every body
 * of an entity follows it rigidly (geom.py pm_shift_bodies); rows of goal entities are ignored */
int mgx_engine_reset_poses(mgx_engine *e, void *state_p, void *state_f, int32_t *state_i, const uint8_t *mask,
                           const void *ent_pose, void *stream);
/* BaseEnv.step() physics (base_env.py:255-274): set_action + 10 x {Robot.update; space.step(dt)} +
 * episode step counter.  actions: DEVICE i32[N] in [0,18).  done: DEVICE u8[N] (may be NULL). */
int mgx_engine_step(mgx_engine *e, void *state_p, void *state_f, int32_t *state_i, const int32_t *actions,
                    uint8_t *done, void *stream);
/* n_substeps physics substeps under `actions` without touching the episode counter (parity tests) */
int mgx_engine_substeps(mgx_engine *e, void *state_p, void *state_f, int32_t *state_i, const int32_t *actions,
                        int n_substeps, void *stream);
/* BaseEnv.render('rgb_array') ego/allo view + ResizeObservation(96) + FlattenFrameStack
 * (base_env.py:309-338, benchmarks/__init__.py:80-136,219-256).  out: DEVICE u8, layout per `layout`;
 * env_stride in bytes (multiple of 4).  fill_mask (DEVICE u8[N] or NULL): envs whose stack is (re)filled
 * with 4 copies of the new frame (FlattenFrameStack.reset). */
/* Test*Colour variants: the colour (mgx_colour) of every entity per env, DEVICE int32 [n_entities][N] owned by the
 * caller and read by every later render call (NULL = the world's own colours again); rows of the robot are ignored.
 * A shape is painted base colour + darkened outline, a goal region lightened interior + base outline
 * (entities.py:750-753,807-819; style.py:28-37) */
int mgx_engine_set_entity_colours(mgx_engine *e, const int32_t *ent_colour);
/* ---- per-env worlds: Test*Shape / TestCountPlus / TestAll variants draw the blocks' shape types and the number of
 * entities anew every episode (e.g. cluster.py:81-110, match_regions.py:101-128).  The engine then keeps one world
 * per env: same entity list as the engine's world (which lists every entity an episode can have), each entity
 * present or absent and each block with its own shape type.  Body indices, entity indices and the state rows the
 * host addresses (poses, velocities, force limits) are the same in every variant; an absent block has an inert body.
 *
 * enable: once, right after mgx_engine_create and before mgx_engine_state_shape; capacity_world = the largest world
 * an episode can have (all entities present, all blocks stars) - it sizes the per-env template tables and LDS. */
int mgx_engine_enable_env_worlds(mgx_engine *e, const mgx_world *capacity_world);
/* the worlds of the m envs env_idx[k] (HOST arrays): enabled[m][n_entities] (NULL: as in the engine's world),
 * shape_types[m][n_entities] (mgx_shape_type; NULL or < 0: as in the engine's world; ignored for non-blocks).
 * Builds / shares the variants, uploads their templates in stream order; call before the reset of those envs.
 * Returns the number of distinct worlds in the call (>= 0).  The uploads are enqueued on `stream` and not waited for (like every
 * other call's device work: later calls on the same stream see the new worlds); the arguments are copied before the call returns, and
 * the next mgx_engine_set_env_variants / mgx_engine_destroy waits for the uploads before the engine's staging buffers are reused. */
int mgx_engine_set_env_variants(mgx_engine *e, int m, const int32_t *env_idx, const uint8_t *enabled, const int32_t *shape_types, void *stream);
/* mgx_world_randomise_all_poses_batch with env env_idx[k] placed in ITS world */
int mgx_engine_env_randomise_all_poses_batch(const mgx_engine *e, int m, const int32_t *env_idx, double *poses, const int *ents, int n,
                                             const uint8_t *ignore, const double arena_lrbt[4], const uint8_t *rand_pos, const uint8_t *rand_rot,
                                             const double *pos_limits, const double *rot_limits, int limits_per_env,
                                             const uint64_t *mt_state_addr, const double *ent_hw);
/* mgx_world_info of env's current world */
int mgx_engine_env_world_info(const mgx_engine *e, int env, int key, int *out);
/* Test*Jitter / TestLayout variants with goal regions: the regions' rectangles per env, DEVICE double [n_goals * 4][N]
 * = x, y (top-left corner), h, w per goal in entity order (entities.py:769-819), caller-owned, read by every later
 * render call (NULL = the world's own rectangles again).  Goal regions are sensors: physics never sees them */
int mgx_engine_set_goal_rects(mgx_engine *e, const double *goal_xyhw);
/* GoalRegion.get_overlapping_ents(com_overlap=True) (entities.py:821-881) for every goal region x every entity x every env,
 * from the pose blob: out = DEVICE u8 [n_goals][n_entities][N] (goals in entity order, mgx_engine_n_goals of them), bit 0 =
 * the entity's body position lies inside the region's box (bb.contains_vect; for the robot this is MoveToRegion's
 * score, move_to_region.py:85-94), bit 1 = every collision shape of the block overlaps the region's rectangle
 * (space.shape_query(...) non-empty for each shape).  An entity counts for a region iff both are set.  Per-env goal
 * rectangles (mgx_engine_set_goal_rects) and per-env worlds (shape types, absent entities) are honoured; mask (DEVICE
 * u8[N] or NULL): envs with mask == 0 get 0.  The tasks' score arithmetic on these sets stays with the caller
 * (match_regions.py:193-213, find_dupe.py:203-216, fix_colour.py:193-202). */
int mgx_engine_score_overlaps(mgx_engine *e, const void *state_p, const uint8_t *mask, uint8_t *out, void *stream);
int mgx_engine_n_goals(const mgx_engine *e);
/* score_on_end_of_traj() of the tasks that score from block POSITIONS, whole, on the device: out = DEVICE double[N] (0 where
 * mask == 0), fp64 with every operation in numpy's order (contraction off), so that the score equals the reference's bit for bit:
 *   MGX_SCORE_CORNER   move_to_corner.py:66-75   entities = {the block}; params = {furthest distance sqrt(2), range sqrt(2) - sqrt(2)/2}
 *   MGX_SCORE_LINE     make_line.py:31-71,142-152 entities = the blocks in task order (an episode has the first k present ones);
 *                                                 params = {inlier distance, max separation}
 *   MGX_SCORE_CLUSTER  cluster.py:166-216         entities = the blocks, cls_default[k] = class of block k in the world's own layout,
 *                                                 cls_env = DEVICE int8 [N][n] per-env classes (variants that redraw colours /
 *                                                 types) or NULL; n_classes <= 8
 * Blocks an env's episode does not have (per-env worlds) take no part.  dot_mode / mm_mode say how the HOST's numpy evaluates the
 * two library primitives the reference's code goes through -- np.linalg.norm of a 2-vector (BLAS ddot) and the [n,2] @ [2,1]
 * product of make_line.py:47 -- 0: x*x + y*y, 1: fma(y, y, x*x), 2: fma(x, x, y*y); the caller finds out by probing its numpy
 * (magical_amd/benchmarks/_scoring.py), because that is what "the reference's result" is on its machine. */
enum mgx_score_task { MGX_SCORE_CORNER = 1, MGX_SCORE_LINE = 2, MGX_SCORE_CLUSTER = 3 };
int mgx_engine_score_points(mgx_engine *e, const void *state_p, int task, int n, const int32_t *entities, const int32_t *cls_default,
                            int n_classes, const int8_t *cls_env, const double *params, int dot_mode, int mm_mode,
                            const uint8_t *mask, double *out, void *stream);
int mgx_engine_render(mgx_engine *e, const void *state_p, uint8_t *out, int64_t env_stride, int view, int layout,
                      const uint8_t *fill_mask, void *stream);
/* BaseEnv.step() physics + its observation in ONE call (base_env.py:255-292 with the wrappers of
 * benchmarks/__init__.py:80-136,219-256): = mgx_engine_step followed by mgx_engine_render(fill_mask = NULL) on the same
 * buffers, same results bit for bit, but issued as a producer / consumer pair: the step kernel publishes every env as soon as its
 * POSES are final and written back (after the last substep's position update: cpSpaceStep moves the bodies first, the rest of the
 * substep only prepares the next step's velocities; the motion blob follows at the kernel's end), and the raster kernel -- on a stream of the engine's own, joined to `stream` before and after --
 * rasterises envs in the order they finish, so the long tail of the physics (a few envs with many contacts) runs under the
 * rasterisation of the others.  Not for steps in which envs are reset between physics and rendering (episode ends:
 * use the two calls).  Every world takes this path (a consumer waits at length only once all producers are resident, otherwise
 * it takes an entry that is there or leaves its env to a clean-up launch); MGX_NO_OVERLAP=1 in the environment makes it the
 * two calls in sequence. */
int mgx_engine_step_render(mgx_engine *e, void *state_p, void *state_f, int32_t *state_i, const int32_t *actions, uint8_t *done,
                           uint8_t *out, int64_t env_stride, int view, int layout, void *stream);
/* diagnostics of the hand-off (synchronises): consumer workgroups that gave up and were served by the clean-up launch */
int mgx_engine_handoff_stats(mgx_engine *e, unsigned *deferred, unsigned *timeouts);
/* native-resolution (384x384x3, no box filter) render of ONE env, for tests against the oracle/images */
int mgx_engine_render_native(mgx_engine *e, const void *state_p, int env, uint8_t *out, int view, void *stream);
/* HIP-event timing: set_timing(e, n) with n > 0 brackets every n-th step (which=0) / render (which=1) launch with
 * events on the launch stream (n = 1: every launch; event records serialise dispatch, so sampling keeps the
 * measured region undisturbed); 0 switches it off; timing_read synchronises those events and returns up to `max` most recent launch
 * durations in ms (oldest first) and clears the log. */
int mgx_engine_set_timing(mgx_engine *e, int enable);
int mgx_engine_timing_read(mgx_engine *e, int which, float *ms, int max);

#ifdef __cplusplus
}
#endif
#endif /* MGX_H_ */

// mgx_world.h -- host-side world description and template builder (C++, fp64).
//
// Restates the physical + visual spec the reference builds at reset time:
//   base_env.py:177-234 (space params, arena first), entities.py:217-437 (Robot.setup),
//   :502-537 (ArenaBoundaries.setup), :614-757 (Shape.setup), :790-819 (GoalRegion.setup),
//   geom.py:13-63,101-108 (vertex maths), style.py (palette).
// Independent of oracle/ (which holds its own Python restatement); tests compare the two.
#pragma once
#include <string>
#include <vector>

#include "mgx_tmpl.h"

namespace mgx {

struct Vec2 { double x, y; };

struct BodyDef {
    int type;
    double m_inv, i_inv;
    double x, y, a;          // initial pose
    int parent;              // >= 0: position = parent pose applied to (ax, ay) at reset (finger roots)
    double ax, ay;
    int state_mask;          // bit c set: component c (x y a vx vy w vbx vby wb) is persistent state
    int ent = -1;            // entity this body belongs to (-1: the static body)
    double aoff = 0;         // body angle = entity angle + aoff at reset (finger roots: +-pi/8)
};
struct ShapeDef {
    int kind, body;
    double radius, friction;
    int group, entity;
    std::vector<Vec2> verts;  // local; circle: none; segment: 2
};
struct JointDef {
    int kind, a, b;
    double ax, ay, bx, by, p0, p1, p2;
    double error_bias, max_bias, max_force;
    int pv = -1;            // index of the PhysicsVariable that is this joint's max_force (-1: constant)
};
struct PrimDef {
    int kind, xform, body, eye_body;
    std::vector<Vec2> verts;
    std::vector<int> parts;    // PR_POLY made of several convex parts: their vertex counts (empty: one part)
    int rgb[3];
    double eye_base[2], eye_pre[2];
    double line_width; int stipple;
    double radius; int ngon;
    int goal = -1;             // ordinal of the goal region this prim draws (-1: none): per-env rectangles (Test*Jitter / Layout)
    int ent = -1, role = -1;   // entity whose colour paints this prim and how (0 darkened, 1 base, 2 lightened x2; -1: fixed colour)
};
struct EntityDef {
    int kind;                // 0 robot, 1 shape, 2 goal
    int shape_type, colour;
    double x, y, angle, h, w;
    int body;                // main body index after finalize (-1 for goals)
    std::vector<int> shapes;
    bool enabled = true;     // false: this env's episode does not have the entity (Test*CountPlus): it keeps its index and,
                             // for a block, an inert body with its state rows, so that indices and rows agree across envs
};

struct World {
    double phys_vars[5] = {3.0, 1.0, 4.0, 1.5, 0.1};   // base_env.py:49-57
    std::vector<EntityDef> entities;
    bool finalized = false;
    int max_episode_steps = 0;
    // built by finalize()
    std::vector<BodyDef> bodies;
    std::vector<ShapeDef> shapes;
    std::vector<JointDef> joints;
    std::vector<PrimDef> prims;
    std::vector<std::pair<int, int>> pairs;
    std::vector<int> state_map;        // comp | body<<4 | row<<12
    int n_state_p = 0;
    std::vector<int> joint_acc_off;
    int n_jacc = 0, cache_slots = 0, max_contacts = 0, max_overlaps = 0;
    int robot_body = -1, control_body = -1, finger_body[2] = {-1, -1}, motor_joint[2] = {-1, -1};
    int group_ctr = 999;
    int robot_j0 = -1, eye_body[2] = {-1, -1};
    std::vector<int> island_j;         // first joint (pivot) of every block's {pivot, gear} pair

    int finalize(int max_steps, std::string &err);
    // the same task with other per-episode choices (Test*Shape / CountPlus / All): entity e is present iff enabled[e]
    // (NULL: as here) and blocks take shape_types[e] (NULL or < 0: as here); finalized like this world
    int variant(const uint8_t *enabled, const int *shape_types, World &out, std::string &err) const;
    // geom.py:116-262 pm_randomise_pose's collision test, on the host: would entity `ent`, with every entity at
    // poses[3 * e .. 3 * e + 2] (x, y, angle; goals: their box centre), touch the arena walls or a shape of an entity
    // whose `enabled` flag is set?  (space.shape_query of each of its shapes: Chipmunk's cpCollide count > 0)
    // ent_hw (optional): [n_entities][2] = this env's (h, w) of every goal region (rows of other entities ignored)
    bool placement_collides(int ent, const double *poses, const uint8_t *enabled, const double *ent_hw = nullptr) const;
    // geom.py:285-341 pm_randomise_all_poses for the entities ents[0..n): draws from the MT19937 stream (key[624], pos)
    // exactly what np.random.RandomState.uniform would (x, y, angle per attempt), entity after entity; limits < 0 = none.
    // Returns the number of rejected attempts, or -1 after max_retries placement failures.
    int randomise_all_poses(double *poses, const int *ents, int n, const uint8_t *ignore, const double arena_lrbt[4],
                            const uint8_t *rand_pos, const uint8_t *rand_rot, const double *pos_limits, const double *rot_limits,
                            uint32_t *mt_key, int *mt_pos, const double *ent_hw = nullptr) const;
    // serialise: header + int words + real words (as double; caller narrows to float if needed)
    // strip_prims: without the draw list (the physics kernels' copy)
    void serialise(TmplHeader &h, std::vector<int32_t> &iw, std::vector<double> &rw, std::vector<double> &pw, bool strip_prims = false) const;
};

// np.random.RandomState primitives on a live mt19937 state (key[624], pos), advanced in place exactly as numpy would:
// `count` draws of randint(0, max_inclusive + 1); `count` draws of random_sample(); the permutation shuffle() applies to n items
void rng_bounded(uint32_t *key, int *pos, int count, uint32_t max_inclusive, int32_t *out);
void rng_doubles(uint32_t *key, int *pos, int count, double *out);
void rng_shuffle(uint32_t *key, int *pos, int n, int32_t *perm);

// RGB8 (r | g << 8 | b << 16) of entity colour 0..3 in role 0 darkened / 1 base / 2 lightened twice (style.py:28-37)
int palette_rgb(int colour, int role);

// physics constants of the reference (base_env.py:62-64,194-196,236-239; benchmarks/__init__.py:401-404)
constexpr double ROBOT_RAD = 0.2;
constexpr double ROBOT_MASS = 1.0;
constexpr double SHAPE_RAD = ROBOT_RAD * 0.6;
constexpr double SHAPE_MASS = 0.5;
constexpr double FPS = 8.0;
constexpr int PHYS_STEPS = 10;
constexpr int PHYS_ITER = 10;
constexpr double COLLISION_SLOP = 0.01;

}  // namespace mgx

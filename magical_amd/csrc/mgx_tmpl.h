// mgx_tmpl.h -- world template + per-env working-set layout shared by the host builder
// (mgx_world.cpp) and the kernels (mgx_step.hip, mgx_raster.hip).
//
// A "template" is everything that is identical for all envs of a batch: body masses, local
// shape geometry, joints, the filtered collision-pair list and the draw list.  It is
// serialised as a header of counts plus two flat arrays (int32 words, real words) that each
// workgroup copies into LDS.  Array offsets are derived from the counts by the same
// constructor on host and device, so the header stays small.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MGX_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define MGX_HD inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define MGX_UNROLL _Pragma("unroll")
#else
#define MGX_UNROLL
#endif

namespace mgx {

enum BodyType { BODY_STATIC = 0, BODY_KINEMATIC = 1, BODY_DYNAMIC = 2 };
enum ShapeKind { SH_CIRCLE = 0, SH_SEGMENT = 1, SH_POLY = 2 };
enum JointKind { J_PIVOT = 0, J_GEAR = 1, J_SPRING = 2, J_PIN = 3, J_LIMIT = 4, J_MOTOR = 5 };
enum PrimKind { PR_POLY = 0, PR_NGON = 1, PR_LINELOOP = 2 };
enum PrimXform { XF_WORLD = 0, XF_BODY = 1, XF_EYE = 2 };

// capacities (index fields are 8 bits for bodies / shapes, 12 bits for candidate pairs; the rasteriser keeps per-tile
// primitive sets in one 64-bit mask).  Largest Demo world: Cluster* = 14 moving bodies, 27 shapes, 26 joints; largest
// Test world: Cluster*-TestAll with ten stars = 17 bodies, 69 shapes, ~2200 candidate pairs.
constexpr int CAP_BODIES = 32;
constexpr int CAP_SHAPES = 128;
constexpr int CAP_VERTS = 512;
constexpr int CAP_JOINTS = 64;
constexpr int CAP_PAIRS = 4095;
constexpr int CAP_PRIMS = 64;
constexpr int CAP_PVERTS = 1024;

constexpr int N_PHYS_VARS = 5;    // robot_pos, robot_rot, finger, shape_trans, shape_rot joint max forces (phys_vars.py)
constexpr int JOINT_PARAMS = 10;   // ax ay bx by p0 p1 p2 bias_rate max_bias max_impulse
constexpr int PRIM_IWORDS = 7;     // kind, nverts, voff | line-vertex offset << 16, xform|body<<8|(eye_body+1)<<16|(role+1)<<24|(entity+1)<<26, rgb(packed), stipple | (goal+1) << 16 | first classification item << 21, part ends
// A PR_POLY primitive is a union of convex parts drawn in one colour (a star: five triangles + a pentagon, entities.py:
// 723-734): its vertices are the parts' vertices back to back and bit i of the `part ends` word marks vertex i as the
// last one of its part (a plain convex polygon has the single bit nverts - 1) -- hence at most 32 vertices per primitive.
constexpr int PRIM_RWORDS = 6;     // eye_base(2) eye_pre(2) line_halfwidth (n-gons: cos(pi / nverts)) radius

#ifndef MGX_BROAD_SAP
#define MGX_BROAD_SAP 0      // 1: the sort-and-sweep broadphase (ph_broad, mgx_sim.h) where a world has at most 32 shapes; 0: the candidate-pair list
#endif
struct TmplHeader {
    int32_t n_bodies, n_shapes, n_verts, n_joints, n_pairs, n_prims, n_pverts;
    int32_t n_state, n_state_p, n_jacc, cache_slots, max_contacts, max_overlaps;   // n_state_p of the n_state rows are pose rows
    int32_t robot_body, control_body, finger_body[2], motor_joint[2];
    int32_t max_episode_steps, n_words_i, n_words_r, n_words_p;
    // joint islands (sets of joints that share no dynamic body with any other set): the robot's 10 joints
    // start at robot_j0 in Robot.setup order; every block contributes {pivot, gear} at island_j[k], +1
    int32_t robot_j0, n_islands, eye_body[2];
    int32_t n_lverts, pad_;       // draw-list vertices that belong to line loops (they carry segment length + arclength in the rasteriser)
};

// indices into the consts block
enum ConstIdx {
    C_DT = 0, C_CONTACT_BIAS_RATE, C_SLOP, C_SPEED_FWD, C_SPEED_BACK, C_TURN, C_FINGER_OPEN, C_FINGER_CLOSED,
    C_PV0,                       // default max impulse (max_force x dt) of the five PhysicsVariables, base_env.py:49-57
    C_N = C_PV0 + 5
};

// offsets into the template's int / real arrays
#ifndef MGX_TAOS
#define MGX_TAOS 15     // template records (all four together: -6 %): bit 0 body (minv iinv), 1 shape ints (kind body voff nv), 2 shape reals (r u), 3 local verts (x y nx ny)
#endif
struct TmplOff {
    static constexpr bool A_B = MGX_TAOS & 1, A_SI = MGX_TAOS & 2, A_SR = MGX_TAOS & 4, A_V = MGX_TAOS & 8;
    static constexpr int S_body_minv = A_B ? 2 : 1, S_body_iinv = A_B ? 2 : 1;
    static constexpr int S_shape_kind = A_SI ? 4 : 1, S_shape_body = A_SI ? 4 : 1, S_shape_voff = A_SI ? 4 : 1, S_shape_nv = A_SI ? 4 : 1;
    static constexpr int S_shape_r = A_SR ? 2 : 1, S_shape_u = A_SR ? 2 : 1;
    static constexpr int S_lvx = A_V ? 4 : 1, S_lvy = A_V ? 4 : 1, S_lnx = A_V ? 4 : 1, S_lny = A_V ? 4 : 1;
    // every other field is a plain array
    static constexpr int S_body_type = 1, S_body_parent = 1, S_body_ent = 1, S_joint_kind = 1, S_joint_a = 1, S_joint_b = 1, S_joint_acc = 1,
                         S_joint_pv = 1, S_pair = 1, S_state_map = 1, S_prim_i = 1, S_pv_prim = 1, S_body_prow = 1, S_island_j = 1,
                         S_body_init = 1, S_body_anchor = 1, S_joint_p = 1, S_prim_r = 1, S_pvx = 1, S_pvy = 1, S_consts = 1,
                         S_p_body_init = 1, S_p_body_anchor = 1, S_p_body_aoff = 1, S_p_joint = 1, S_p_dt = 1, S_pair_allow = 1, S_pair_row = 1;
    // ints
    int body_type, body_parent, body_ent, shape_kind, shape_body, shape_voff, shape_nv;
    int joint_kind, joint_a, joint_b, joint_acc, joint_pv, pair, state_map, prim_i, pv_prim, body_prow, island_j, n_i;
    int pair_allow, pair_row;       // MGX_BROAD_SAP builds: per shape a the candidate partners b > a as a bit mask, and the number of a's first candidate pair
    // reals
    int body_minv, body_iinv, body_init, body_anchor, shape_r, shape_u, lvx, lvy, lnx, lny;
    int joint_p, prim_r, pvx, pvy, consts, n_r;
    // pose-precision copies (P-typed array): initial poses, body anchors, joint anchors/angles, dt
    int p_body_init, p_body_anchor, p_body_aoff, p_joint, p_dt, n_p;
    MGX_HD explicit TmplOff(const TmplHeader &h) {
        int o = 0;
        body_type = o; o += h.n_bodies;
        body_parent = o; o += h.n_bodies;
        body_ent = o; o += h.n_bodies;        // entity of the body (per-env reset poses), -1 for the static body
        { const int d = A_SI ? 1 : h.n_shapes; shape_kind = o; shape_body = o + d; shape_voff = o + 2 * d; shape_nv = o + 3 * d; o += 4 * h.n_shapes; }
        joint_kind = o; o += h.n_joints;
        joint_a = o; o += h.n_joints;
        joint_b = o; o += h.n_joints;
        joint_acc = o; o += h.n_joints;
        joint_pv = o; o += h.n_joints;        // which PhysicsVariable limits this joint's impulse (-1: the template's own value)
        pair = o; o += h.n_pairs;
        state_map = o; o += h.n_state;
        prim_i = o; o += h.n_prims * PRIM_IWORDS;
        pv_prim = o; o += h.n_pverts;         // primitive of every draw-list vertex (the rasteriser sets vertices up one per lane)
        body_prow = o; o += h.n_bodies * 3;   // pose-blob row of (x, y, angle) per body, -1 if not persistent
        island_j = o; o += h.n_islands;
        pair_allow = pair_row = o;
#if MGX_BROAD_SAP
        if (h.n_shapes <= 32) { pair_allow = o; o += h.n_shapes; pair_row = o; o += h.n_shapes; }
#endif
        n_i = o;
        o = 0;
        { const int d = A_B ? 1 : h.n_bodies; body_minv = o; body_iinv = o + d; o += 2 * h.n_bodies; }
        body_init = o; o += h.n_bodies * 3;
        body_anchor = o; o += h.n_bodies * 2;
        { const int d = A_SR ? 1 : h.n_shapes; shape_r = o; shape_u = o + d; o += 2 * h.n_shapes; }
        { const int d = A_V ? 1 : h.n_verts; lvx = o; lvy = o + d; lnx = o + 2 * d; lny = o + 3 * d; o += 4 * h.n_verts; }
        joint_p = o; o += h.n_joints * JOINT_PARAMS;
        prim_r = o; o += h.n_prims * PRIM_RWORDS;
        pvx = o; o += h.n_pverts;
        pvy = o; o += h.n_pverts;
        consts = o; o += C_N;
        n_r = o;
        o = 0;
        p_body_init = o; o += h.n_bodies * 3;
        p_body_anchor = o; o += h.n_bodies * 2;
        p_body_aoff = o; o += h.n_bodies;     // body angle - entity angle at reset
        p_joint = o; o += h.n_joints * 7;
        p_dt = o; o += 1;
        n_p = o;
    }
};

// misc int slots per env
enum MiscIdx { M_NOV = 0, M_NK, M_NARB, M_NCACHE, M_NNCACHE, M_OVERFLOW, M_ACTION, M_STEPS, M_NMAN, M_N };      // (M_NMAN: manifold slots handed out this substep)

// per-env LDS working set: offsets in real-sized words (R region) and 32-bit words (int region).
// The per-body, per-joint and per-contact fields are RECORDS (array of structs): field f of element i sits at
// wo.f + i * WorkOff::S_f, where wo.f = record base + the field's position.  One base register and immediate offsets then
// address a whole record (the solver reads all 13 fields of a contact, all 6 velocities of a body), instead of one
// scalar offset per field held live across the kernel.  Record strides are odd (or lanes walk them with distinct banks)
// so that lanes working on consecutive elements do not collide in LDS.
// Manifold scratch per TOUCHING pair (slots handed out in pair order, ph_narrow) in every world.  Round 5 handed the slots out by an LDS
// counter, which cost the small worlds' step kernel 1.5 %, and kept one slot per overlapping pair for worlds below 26 cache slots -- a
// per-env working set that was not monotone in the number of blocks (five blocks 1858 words, six 1698: round-5 advisor).  The ballot form
// costs nothing (profiles/r06_step_broadphase_sap_ab.txt, the `slotsall` lines), so every world takes it (MGX_MANIFOLD_SLOTS_ALL=0: round 5's rule).
#ifndef MGX_MANIFOLD_SLOTS_ALL
#define MGX_MANIFOLD_SLOTS_ALL 1
#endif
MGX_HD bool manifold_slots(const TmplHeader &h) { return MGX_MANIFOLD_SLOTS_ALL || h.cache_slots > 25; }
struct WorkOff {
    // bodies: poses live in the pose-precision region (P words), velocities in the R region
    int px, py, ang, c, s, n_p;
    int vx, vy, w, vbx, vby, wb;
    // world-space shape data
    int wx, wy, wnx, wny, bbl, bbb, bbr, bbt;
    // joints: accumulated impulses (2; in registers during the env-step, here only between ph_load_state / ph_store_state and
    // the solver context), motor rate (Robot.update -> the motors' preStep), max impulse per substep
    int ja0, ja1, jrate, jlim;
    // contact points
    int knx, kny, kr1x, kr1y, kr2x, kr2y, knm, ktm, kbias, kjb, kjn, kjt, kmu;
    // manifold scratch per TOUCHING pair (slot): n(2) + 2 x (p1, p2)(4)  [mn and mp are adjacent: ncj lies over both]
    int mn, mp;
    // persistent contact cache impulses (jn, jt per point) and the one being built
    int cj, ncj;
    int n_r;
    // int region
    int ov, mcnt, koff, kab, chead, nchead, cmatched, misc;        // (ov: uint16[max_overlaps], cmatched: uint8[cache_slots], the rest int32)
    int n_i;
    // element strides of the fields (1 = plain array)
#ifndef MGX_AOS
#define MGX_AOS 507     // bit 0 body poses, 1 body velocities, 2 joints, 3 contacts (joint records measured slower), 4.. below
#endif
#ifndef MGX_ALIAS
#define MGX_ALIAS 1
#endif
    static constexpr bool ALIAS = MGX_ALIAS != 0;
    static constexpr bool AOS_BP = MGX_AOS & 1, AOS_BR = MGX_AOS & 2, AOS_J = MGX_AOS & 4, AOS_K = MGX_AOS & 8;
    // 4 world vertices (x y nx ny: -1.4 %), 5 shape boxes, 6 overlap records (pair count hash offset), 7 contact ints, 8 cache ints
    // (5..8: nothing measurable one by one, -1.5 % together)
    static constexpr bool AOS_V = MGX_AOS & 16, AOS_BB = MGX_AOS & 32, AOS_OV = MGX_AOS & 64, AOS_KI = MGX_AOS & 128, AOS_C = MGX_AOS & 256;
    static constexpr int BODY_P = AOS_BP ? 5 : 1, BODY_R = AOS_BR ? 6 : 1, JOINT_R = AOS_J ? 4 : 1, CONTACT_R = AOS_K ? 13 : 1;
    static constexpr int S_px = BODY_P, S_py = BODY_P, S_ang = BODY_P, S_c = BODY_P, S_s = BODY_P;
    static constexpr int S_vx = BODY_R, S_vy = BODY_R, S_w = BODY_R, S_vbx = BODY_R, S_vby = BODY_R, S_wb = BODY_R;
    static constexpr int VERT_R = AOS_V ? 4 : 1, BOX_R = AOS_BB ? 4 : 1, OV_I = AOS_OV ? 2 : 1, KI_I = AOS_KI ? 2 : 1, C_I = AOS_C ? 3 : 1;
    static constexpr int S_wx = VERT_R, S_wy = VERT_R, S_wnx = VERT_R, S_wny = VERT_R, S_bbl = BOX_R, S_bbb = BOX_R, S_bbr = BOX_R, S_bbt = BOX_R;
    static constexpr int S_ja0 = JOINT_R, S_ja1 = JOINT_R, S_jrate = JOINT_R, S_jlim = JOINT_R;
    static constexpr int S_knx = CONTACT_R, S_kny = CONTACT_R, S_kr1x = CONTACT_R, S_kr1y = CONTACT_R, S_kr2x = CONTACT_R, S_kr2y = CONTACT_R,
                         S_knm = CONTACT_R, S_ktm = CONTACT_R, S_kbias = CONTACT_R, S_kjb = CONTACT_R, S_kjn = CONTACT_R, S_kjt = CONTACT_R, S_kmu = CONTACT_R;
    static constexpr int S_mn = 1, S_mp = 1, S_cj = 1, S_ncj = 1;
    // (round 5: the overlap list is 16-bit, the matched flags bytes, a contact's `first` flag a bit of its body word: plain arrays)
    static constexpr int S_mcnt = 1, S_koff = 1, S_kab = 1, S_chead = 1, S_nchead = 1, S_misc = 1;
    MGX_HD explicit WorkOff(const TmplHeader &h) {
        int nb = h.n_bodies, nv = h.n_verts, ns = h.n_shapes, nj = h.n_joints;
        int nk = h.max_contacts, nov = h.max_overlaps, nc = h.cache_slots;
        int o = 0;
        const int sbp = AOS_BP ? 1 : nb, sbr = AOS_BR ? 1 : nb, sj = AOS_J ? 1 : nj, sk = AOS_K ? 1 : nk;     // distance between a record's fields
        px = o; py = o + sbp; ang = o + 2 * sbp; c = o + 3 * sbp; s = o + 4 * sbp; o += 5 * nb;
        n_p = o;
        o = 0;
        vx = o; vy = o + sbr; w = o + 2 * sbr; vbx = o + 3 * sbr; vby = o + 4 * sbr; wb = o + 5 * sbr; o += 6 * nb;
        const int sv = AOS_V ? 1 : nv, sbb = AOS_BB ? 1 : ns;
        // The world-space vertices and boxes live from ph_shapes to ph_narrow, the contact records from ph_arbiters_joints to
        // ph_cache_commit of the same substep: the two share their words (MGX_ALIAS; 2 KB per env in ClusterColour).
        const int shared = o;
        wx = o; wy = o + sv; wnx = o + 2 * sv; wny = o + 3 * sv; o += 4 * nv;
        bbl = o; bbb = o + sbb; bbr = o + 2 * sbb; bbt = o + 3 * sbb; o += 4 * ns;
        const int geom_end = o;
        if (ALIAS) o = shared;
        knx = o; kny = o + sk; kr1x = o + 2 * sk; kr1y = o + 3 * sk; kr2x = o + 4 * sk; kr2y = o + 5 * sk; knm = o + 6 * sk; ktm = o + 7 * sk;
        kbias = o + 8 * sk; kjb = o + 9 * sk; kjn = o + 10 * sk; kjt = o + 11 * sk; kmu = o + 12 * sk; o += 13 * nk;
        if (o < geom_end) o = geom_end;
        ja0 = o; ja1 = o + sj; jrate = o + 2 * sj; jlim = o + 3 * sj; o += 4 * nj;
        // manifolds: one slot per overlapping pair that TOUCHES (handed out in pair order by ph_narrow -- round 5: by an LDS counter --, its number kept in the pair's
        // count word) -- as many as there can be arbiters (nc), not as many as there can be overlapping boxes (nov: round 5, 420 words of
        // ClusterColour's 2562 per env; with the three packings below the 16-lane working set fits a CU four times)
        const int mcap = manifold_slots(h) ? nc : nov;
        mn = o; o += mcap * 2; mp = o; o += mcap * 8;
        cj = o; o += nc * 4;
        // the cache being built is written from solve_begin on, when the manifolds (ph_narrow .. ph_arbiters_joints) are dead
        if (ALIAS && 4 * nc <= 10 * mcap) ncj = mn; else { ncj = o; o += nc * 4; }
        n_r = o;
        o = 0;
        // overlapping pairs: candidate pair (16 bit) and manifold word (point count | point hashes << 8 | slot << 24) per pair; first contact
        // of each arbiter, by rank; cache headers old / new; matched flags (bytes); per contact: body a | body b << 8 | first << 16
        ov = o; o += (nov + 1) / 2;
        mcnt = o; o += nov;
        koff = o; o += nc;
        chead = o; o += nc; nchead = o; o += nc;
        cmatched = o; o += (nc + 3) / 4;
        misc = o; o += M_N;
        kab = o; o += nk;
        n_i = o;
    }
};

// cache entry header: pair(12) | age(2)<<12 | count(2)<<14 | hash0(8)<<16 | hash1(8)<<24
MGX_HD uint32_t cache_pack(uint32_t pair, uint32_t age, uint32_t count, uint32_t h0, uint32_t h1) {
    return (pair & 0xFFFu) | ((age & 3u) << 12) | ((count & 3u) << 14) | ((h0 & 0xFFu) << 16) | ((h1 & 0xFFu) << 24);
}

}  // namespace mgx

// mgx_raster.hip -- k_raster: 96x96 egocentric (or allocentric) observation frames for N envs,
// written straight into the caller's (PyTorch) u8 tensor.
//
// Launch geometry (CDNA4): one 256-thread workgroup (4 wavefronts) per env.  The workgroup
// stages the draw list + this env's poses into LDS, sets up screen-space primitives once
// (lane per body, then lane per primitive), then each wavefront walks 16x4-pixel tiles:
// a 64-bit ballot builds the tile's primitive set (wave-uniform, so the per-pixel loop over
// it does not diverge on the loop structure), each lane resolves one output pixel, and the
// frame goes out as packed dwords (FRAME layout, via cross-lane shuffles) or as a 12-byte
// load-shift-store per pixel (STACK4 layout = FlattenFrameStack's [96,96,12], oldest first).
// 4096 envs -> 4096 workgroups = 16 per CU, dispatched round-robin over the 8 XCDs.
#include <hip/hip_runtime.h>

#include "mgx_raster.h"

namespace mgx {

struct RasterDev {
    const uint32_t *words;   // [TmplHeader][int words][pad][tq doubles]; env e's own blob at words + e * tmpl_stride_words.  Header and ints are
                             // staged in LDS; the doubles (local vertices, radii) are read once per frame by the set-up, straight from HBM
    long tmpl_stride_words;  // 0: one draw list for all envs
    int n_words;             // words of the shared blob (per-env blobs: see raster_blob_words)
    int off_i;               // word offset of the int array (the doubles follow at raster_off_q)
    int lds_tmpl_words;      // reserved words for the template (even)
    int scratch_d;           // doubles of per-env scratch
    int bg_rgb;
    int off_tiles;           // word offset (inside the scratch area) of the per-tile / queue region, 8-byte aligned
    unsigned long long *dbg_clk;   // development probe: per-block phase clocks [n_envs][16] (NULL = off)
    int dbg_stop;                  // development probe: return after phase k (0 = run everything)
    const int32_t *ent_colour_env; // [n_entities][n_envs] per-env entity colour indices (NULL: the template's colours)
    const int32_t *palette;        // device int32[12]: RGB8 of colour c in role r at [4 * r + c] (style.py:28-37)
    const double *goal_xyhw_env;   // [n_goals * 4][n_envs] per-env goal rectangles x, y (top-left), h, w (NULL: the template's)
    int tq_hbm;                    // the draw list's fp64 part is not staged in LDS: the set-up reads it from HBM (the host's choice)
    int compact;                   // draw-list vertex records without the edge-function coefficients (RasterOff; the host's choice)
    int qcap;                      // queue entries in use (<= qcap_lds; tests shrink it to exercise the overflow rounds)
    int qcap_lds;                  // queue entries the LDS layout holds (<= QCAP; the host's choice by world, configure_launch)
    int ecap;                      // phase E records in use (<= ECAP; likewise)
    int prio_t;                    // wavefronts in phase T run at s_setprio 2 (the host's choice by world, configure_launch)
    int narrow;                    // primitive sets are 32-bit words (every world loaded has <= 32 primitives): the uint32_t instantiations
};

// Consumer side of the step -> raster hand-off (mgx_engine_step_render; producer: StepHandoff in mgx_step.hip).
//   mode 0  plain launch: workgroup b rasterises env b
//   mode 1  runs concurrently with k_step: workgroup b rasterises the b-th env k_step finishes.  It waits for its queue entry
//           only if every producer workgroup is already executing (otherwise the consumers' own residency could be what keeps the
//           producers from starting); if not, or if the bounded wait runs out, it marks itself deferred and exits
//   mode 2  clean-up launch after both kernels: workgroup b rasterises queue[b]'s env iff b was deferred (normally none)
struct RasterHandoff {
    const unsigned long long *queue;
    const unsigned *started;
    unsigned started_base, n_producers;    // producers of this call have all begun once *started - started_base >= n_producers
    unsigned epoch;
    unsigned *deferred;                    // [n_envs] = epoch where workgroup b gave up
    unsigned *stats;                       // [0] deferred workgroups so far (diagnostic), [1] bounded waits that ran out
    int mode;
    unsigned poll_limit;                   // polls a consumer spends on its entry (HANDOFF_POLL_LIMIT; tests force the run-out: mgx_engine_debug_handoff)
};
#ifdef MGX_HANG_DEBUG     // development: monotonic counters behind the hand-off's words (mgx_engine_debug_handoff_peek)
#define HDBG(i) if (ho.mode && threadIdx.x == 0) atomicAdd(&ho.stats[2 + (i)], 1u);
#else
#define HDBG(i)
#endif
// x (s_sleep 16 + an L2 round trip) ~ 0.1 s: far beyond any step kernel.  It does run out: the two kernels sit on two hardware queues, and
// the system context-switches wavefronts (queues) now and then -- with the producers' queue switched out the consumers spin for nothing.
// One 20 000-step run in ten sees a burst of ~2000 such waits; at 2^20 polls (round 4) that burst cost 2 s, now 0.15 s.  The envs are
// then rasterised by the clean-up launch like any other deferred one.
constexpr unsigned HANDOFF_POLL_LIMIT = 1u << 16;

MGX_HD int raster_off_q(const TmplHeader &h, int off_i) { return (off_i + h.n_words_i + 1) & ~1; }
MGX_HD int raster_blob_words(const TmplHeader &h, int off_i) { return raster_off_q(h, off_i) + 2 * (h.n_prims * PRIM_RWORDS + 2 * h.n_pverts); }
constexpr int N_TILES = TILES_X * TILES_Y;
// words of a draw-list blob that a workgroup stages in LDS, and where its fp64 part is then read
__device__ __forceinline__ int raster_staged_words(const RasterDev &t, const uint32_t *blob) {
    const TmplHeader &h = *reinterpret_cast<const TmplHeader *>(blob);
    return t.tq_hbm ? raster_off_q(h, t.off_i) : raster_blob_words(h, t.off_i);
}
__device__ __forceinline__ const double *raster_tq(const RasterDev &t, const uint32_t *blob, const uint32_t *lds, const TmplHeader &h) {
    return reinterpret_cast<const double *>((t.tq_hbm ? blob : lds) + raster_off_q(h, t.off_i));
}
constexpr int QCAP = 1024;     // LDS queue of undecided pixels per env (largest layout); what does not fit waits in a bitmap for another round
constexpr int QCAP_SMALL = 640;   // ... of the worlds without a goal region and with at most one block (MoveToCorner: 440 queued pixels per frame
                                  // at the median, 550 at p99.9, 660 at most over 2 x 10^5 frames, tools/dev/nq_stats.py): 6 KB less LDS
constexpr int OVF_WORDS = LORES * LORES / 32;
constexpr int ECAP = 256;      // phase E records per round (uncertain pixels beyond that wait in the bitmap like queue overflow)
// k_raster is instantiated for 3, 4 and 5 workgroups per CU (= waves per SIMD: VGPR caps 168 / 128 / 96); the host picks
// the one the world's LDS footprint allows, so that LDS-bound worlds are not squeezed into fewer registers for nothing
#ifndef MGX_RASTER_WAVES
#define MGX_RASTER_WAVES 4      // k_raster_native only
#endif
#ifndef MGX_STACK_GROUP
#define MGX_STACK_GROUP 6      // tiles whose old pixels are fetched ahead, per wavefront (stack layouts; round 3: 4 -> 6, +0.3 ... 0.9 % on every task measured)
#endif

// Output layouts (include/mgx.h mgx_obs_layout).  The three stacked ones address a 12 B pixel of u8[N][96][96][12]:
//   1 STACK4   FlattenFrameStack of one view, depth 4: bytes 0..8 <- bytes 3..11, bytes 9..11 <- new frame
//   2 STACK3HI depth 3 in bytes 3..11 (the 3 ego frames of LoRes3EA): bytes 3..8 <- bytes 6..11, bytes 9..11 <- new
//   3 SLOT0    bytes 0..2 <- new frame, the rest untouched (the allo frame of LoRes3EA)
// after a reset (`fill`) every frame of the stack is the new one.
//   4 PLANAR   u8[3][96][96] channel planes of the new frame only (one slot of a caller-owned ring of frames: channels-first
//              stacks are then contiguous windows of the ring, nothing is shifted)
constexpr int LAY_FRAME = 0, LAY_STACK4 = 1, LAY_STACK3HI = 2, LAY_SLOT0 = 3, LAY_PLANAR = 4;
struct OldPx { uint32_t o0, o1, o2; };
template <int LAYOUT> __device__ __forceinline__ bool layout_needs_old(bool fill) { return LAYOUT == LAY_STACK4 ? !fill : true; }
template <int LAYOUT> __device__ __forceinline__ OldPx load_old(const uint8_t *frame, int X, int Y) {
    const uint32_t *px = reinterpret_cast<const uint32_t *>(frame + (uint32_t)((Y * LORES + X) * 12));     // uniform base + 32-bit offset
    OldPx o; o.o0 = px[0];
    if (LAYOUT == LAY_SLOT0) { o.o1 = 0; o.o2 = 0; } else { o.o1 = px[1]; o.o2 = px[2]; }
    return o;
}
// the pixel update with the old pixel already in registers (loaded a tile ahead so its latency hides behind classification)
template <int LAYOUT> __device__ __forceinline__ void store_pre(uint8_t *frame, int X, int Y, int c, bool fill, const OldPx &o) {
    uint32_t *px = reinterpret_cast<uint32_t *>(frame + (uint32_t)((Y * LORES + X) * 12));
    const uint32_t r = c & 0xFF, g = (c >> 8) & 0xFF, b = (c >> 16) & 0xFF;
    if (LAYOUT == LAY_SLOT0) { px[0] = (o.o0 & 0xFF000000u) | ((uint32_t)c & 0xFFFFFFu); return; }
    uint32_t d0, d1, d2;
    if (fill) {
        d0 = LAYOUT == LAY_STACK4 ? (r | (g << 8) | (b << 16) | (r << 24)) : ((o.o0 & 0xFFFFFFu) | (r << 24));
        d1 = g | (b << 8) | (r << 16) | (g << 24);
        d2 = b | (r << 8) | (g << 16) | (b << 24);
    } else {
        d0 = LAYOUT == LAY_STACK4 ? ((o.o0 >> 24) | (o.o1 << 8)) : ((o.o0 & 0xFFFFFFu) | ((o.o1 >> 16) << 24));
        d1 = (o.o1 >> 24) | (o.o2 << 8);
        d2 = (o.o2 >> 24) | ((uint32_t)c << 8);
    }
    px[0] = d0; px[1] = d1; px[2] = d2;
}
// Two horizontally adjacent pixels (24 B) per lane: 16 lanes cover a 32-pixel row = 384 B = three whole 128 B lines, so
// the in-place shift of a PAIR of 16x4 tiles reads and writes complete lines only (a single tile's row is 1.5 lines).
struct OldPx2 { OldPx a, b; };
template <int LAYOUT> __device__ __forceinline__ OldPx2 load_old2(const uint8_t *frame, uint32_t off) {
    const uint32_t *px = reinterpret_cast<const uint32_t *>(frame + off);
    OldPx2 o;
    o.a.o0 = px[0]; o.b.o0 = px[3];
    if (LAYOUT == LAY_SLOT0) { o.a.o1 = o.a.o2 = o.b.o1 = o.b.o2 = 0; }
    else { o.a.o1 = px[1]; o.a.o2 = px[2]; o.b.o1 = px[4]; o.b.o2 = px[5]; }
    return o;
}
template <int LAYOUT> __device__ __forceinline__ void shifted_px(int c, bool fill, const OldPx &o, uint32_t &d0, uint32_t &d1, uint32_t &d2) {
    const uint32_t r = c & 0xFF, g = (c >> 8) & 0xFF, b = (c >> 16) & 0xFF;
    if (fill) {
        d0 = LAYOUT == LAY_STACK4 ? (r | (g << 8) | (b << 16) | (r << 24)) : ((o.o0 & 0xFFFFFFu) | (r << 24));
        d1 = g | (b << 8) | (r << 16) | (g << 24);
        d2 = b | (r << 8) | (g << 16) | (b << 24);
    } else {
        d0 = LAYOUT == LAY_STACK4 ? ((o.o0 >> 24) | (o.o1 << 8)) : ((o.o0 & 0xFFFFFFu) | ((o.o1 >> 16) << 24));
        d1 = (o.o1 >> 24) | (o.o2 << 8);
        d2 = (o.o2 >> 24) | ((uint32_t)c << 8);
    }
}
template <int LAYOUT> __device__ __forceinline__ void store_pre2(uint8_t *frame, uint32_t off, int c0, int c1, bool fill, const OldPx2 &o) {
    uint32_t *px = reinterpret_cast<uint32_t *>(frame + off);
    if (LAYOUT == LAY_SLOT0) {
        px[0] = (o.a.o0 & 0xFF000000u) | ((uint32_t)c0 & 0xFFFFFFu);
        px[3] = (o.b.o0 & 0xFF000000u) | ((uint32_t)c1 & 0xFFFFFFu);
        return;
    }
    uint32_t d[6];
    shifted_px<LAYOUT>(c0, fill, o.a, d[0], d[1], d[2]);
    shifted_px<LAYOUT>(c1, fill, o.b, d[3], d[4], d[5]);
    // 24 B, 8-byte aligned
    uint2 *q = reinterpret_cast<uint2 *>(px);
    q[0] = make_uint2(d[0], d[1]); q[1] = make_uint2(d[2], d[3]); q[2] = make_uint2(d[4], d[5]);
}
// phases Q / E: the pixel has already been shifted by phase T; write only the bytes that hold the new frame
template <int LAYOUT> __device__ __forceinline__ void store_patch(uint8_t *frame, int X, int Y, int c, bool fill) {
    uint8_t *px = frame + (long)(Y * LORES + X) * 12;
    const uint8_t r = c & 0xFF, g = (c >> 8) & 0xFF, b = (c >> 16) & 0xFF;
    const int first = LAYOUT == LAY_SLOT0 ? 0 : (fill ? (LAYOUT == LAY_STACK4 ? 0 : 3) : 9);   // after a reset every frame of the stack
    const int last = LAYOUT == LAY_SLOT0 ? 3 : 12;
    for (int k = first; k < last; k += 3) { px[k] = r; px[k + 1] = g; px[k + 2] = b; }
}
__device__ __forceinline__ void store_planar_px(uint8_t *frame, int X, int Y, int c) {
    uint8_t *q = frame + (Y * LORES + X);          // a tile row = 16 consecutive bytes per plane
    q[0] = c & 0xFF; q[LORES * LORES] = (c >> 8) & 0xFF; q[2 * LORES * LORES] = (c >> 16) & 0xFF;
}
__device__ __forceinline__ void store_frame_px(uint8_t *frame, int X, int Y, int c) {
    uint8_t *q = frame + (long)(Y * LORES + X) * 3;
    q[0] = c & 0xFF; q[1] = (c >> 8) & 0xFF; q[2] = (c >> 16) & 0xFF;
}

// masked_item_index for a prim set that is the same in every lane (a tile's): the walk over the set runs on the scalar
// unit, each lane only compares its slot against the running item count
template <typename M> __device__ __forceinline__ M mask_uniform(M v);
template <> __device__ __forceinline__ uint64_t mask_uniform<uint64_t>(uint64_t v) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
}
template <> __device__ __forceinline__ uint32_t mask_uniform<uint32_t>(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane(v); }
template <typename M> __device__ __forceinline__ int masked_item_index_uniform(const Raster &rs, M mask_v, int slot, int &n_total) {
    M mask = mask_uniform<M>(mask_v);
    int acc = 0, found = -1;
    while (mask) {
        const int k = mask_top(mask);
        mask &= ~(M(1) << k);
        const int pi = __builtin_amdgcn_readfirstlane(RI(pitem, k)), start = pi & 0xFFFF, cnt = (pi >> 16) & PI_CNT_MASK;
        if (slot >= acc && slot < acc + cnt) found = start + (slot - acc);
        acc += cnt;
    }
    n_total = acc;
    return found;
}

// The device's classification state: like ClassState (mgx_raster.h), with the covering primitive's INDEX in place of its colour -- the
// colour is looked up once per tile / pixel at the end instead of once per primitive of the walk, and a queued pixel's entry holds
// the index (7 bits) beside its position (14), which makes the entry 8 bytes with a 32-bit primitive set.  BK_BG = the background.
constexpr int BK_BG = 64, BK_TILE = 255;       // (BK_TILE: phase T's walk has met no covering primitive yet -- the tile's own base holds)
template <typename M> struct PixState {
    M mixed; int bk; int decided; float lo; int line;
    float gate;        // pixel-level walks: 0, then -inf once a primitive has covered the lane's block (added to every later lo: far outside)
    __device__ __forceinline__ void init(int bk0) { mixed = 0; bk = bk0; decided = 0; lo = 1e30f; line = 0; gate = 0.0f; }
};
__device__ __forceinline__ int rgb_of(const Raster &rs, int bk, int bg_rgb) { return bk >= BK_BG ? bg_rgb : rs.prim_rgb(bk < BK_BG ? bk : 0); }

// v_min_f32 as it is: fminf first canonicalises an operand the compiler cannot prove quiet (NaNs cannot arise where this is used:
// finite coefficients, finite coordinates)
__device__ __forceinline__ float raw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float raw_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// one Item per lane; get(i) broadcasts lane i's record to the whole wave as scalar operands
struct RegItems {
    Item my;
    __device__ __forceinline__ Item get(int i) const {
        Item r;
        r.a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my.a), i));
        r.b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my.b), i));
        r.c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my.c), i));
        r.g0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my.g0), i));
        r.g1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my.g1), i));
        r.g2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my.g2), i));
        r.g3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my.g3), i));
        r.meta = __builtin_amdgcn_readlane(my.meta, i);
        return r;
    }
};
// A tile's line-loop SEGMENTS (round 5): phase C notes, per tile, which segments of the tile's line loops can touch it -- bit 8 s + e for
// segment e of the s-th line-loop primitive (s < 4, loops of at most 8 segments; others: no note, never skipped) -- and phase T's pixels
// test only those.  The arena's boundary runs through two tiles in three of MoveToCorner's mixed ones, and one of its four segments does.
__device__ __forceinline__ int seg_note_base(const Raster &rs, uint64_t lm, int k) {      // 8 s, or -1: this primitive's segments are not noted
    const int s = __builtin_popcountll(lm & ((1ull << k) - 1ull));
    const int cnt = (RI(pitem, k) >> 16) & PI_CNT_MASK;
    return (s < 4 && cnt <= 8) ? 8 * s : -1;
}
// consume items [0, n) held one per lane; stops at a primitive boundary once every lane of the wave is decided.
// Same arithmetic as classify_item, organised as runs of one primitive's items so that the polygon-edge loop is
// branch-free (3 broadcasts + 2 fma + 1 min per edge) and the per-primitive verdict is computed without divergence.
template <bool TILE, typename M>
__device__ __forceinline__ void classify_items_regs(const Raster &rs, const RegItems &src, int n, float xc, float yc,
                                                    PixState<M> &st, bool active, uint32_t segw = ~0u, uint64_t lm = 0) {
    auto bc = [&](float v, int i) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i)); };
    const float hx = TILE ? TILE_HX : 1.5f, hy = TILE ? TILE_HY : 1.5f;
    int i = 0;
    while (i < n) {
        const int meta = __builtin_amdgcn_readlane(src.my.meta, i);
        const int kind = meta & 3, rem = (meta >> IT_REM_SHIFT) & IT_REM_MASK, k = meta >> IT_K_SHIFT;
        const int run = rem < n - i ? rem : n - i;
        float lo = st.lo;
        if (kind == IT_EDGE) {
            for (int j = i; j < i + run; j++) {
                // (explicit fma and the raw v_min: 7 instead of 10 instructions per edge -- the compiler's form multiplied in pairs and
                // canonicalised the running minimum each turn)
                const float e = TILE ? __builtin_fmaf(bc(src.my.a, j), xc, __builtin_fmaf(bc(src.my.b, j), yc, bc(src.my.c, j))) * bc(src.my.g3, j)
                                     : __builtin_fmaf(bc(src.my.g0, j), xc, __builtin_fmaf(bc(src.my.g1, j), yc, bc(src.my.g2, j)));
                lo = raw_min(lo, e);
            }
        } else if (kind == IT_NGON) {
            const float qx = __builtin_fabsf(xc - bc(src.my.a, i)), qy = __builtin_fabsf(yc - bc(src.my.b, i));
            const float nx = r_max(qx - hx, 0.0f), ny = r_max(qy - hy, 0.0f);
            const float fx = qx + hx, fy = qy + hy;
            const float apo = bc(src.my.c, i) - CLASS_EPS_F, rad = bc(src.my.g0, i) + CLASS_EPS_F;
            const float l = rad * rad - (nx * nx + ny * ny);
            const float h = apo > 0.0f ? apo * apo - (fx * fx + fy * fy) : -1.0f;
            lo = l < 0.0f ? -2.0f : (h > 0.0f ? 2.0f : 0.0f);
        } else {
            // (pixel level: only the segments that phase C found touching this tile)
            const int nb = TILE ? -1 : seg_note_base(rs, lm, k);
            const int e0 = ((RI(pitem, k) >> 16) & PI_CNT_MASK) - rem;           // the run's first segment within its loop
            for (int j = i; j < i + run; j++) {
                if (!TILE && nb >= 0 && !((segw >> (nb + e0 + (j - i))) & 1u)) continue;
                const float a = bc(src.my.a, j), b = bc(src.my.b, j);
                const float hw = bc(src.my.g3, j) + CLASS_EPS_F;
                const float e = a * xc + (b * yc + bc(src.my.c, j));
                const float sl = (xc - bc(src.my.g0, j)) * b - (yc - bc(src.my.g1, j)) * a;
                const float el = hx * __builtin_fabsf(a) + hy * __builtin_fabsf(b), es = hx * __builtin_fabsf(b) + hy * __builtin_fabsf(a);
                const float t = r_min(r_min(hw + el - __builtin_fabsf(e), sl + es + hw), bc(src.my.g2, j) + hw + es - sl);
                lo = lo >= BIG_F ? t : r_max(lo, t);
            }
        }
        i += run;
        if (run == rem) {
            // the primitive (or one convex part of it) is complete: verdict for this lane's block
            if (TILE) {
                const bool open = !st.decided;
                // (one compare against a wave-uniform threshold: `lo` is never NaN -- finite coefficients at finite coordinates --, so
                // lo >= 0 for a line loop and !(lo < -1) for the others are both lo >= thr; six vector instructions less per primitive
                // than selecting between the two compares' results)
                const float thr = __builtin_bit_cast(float, kind == IT_SEG ? 0u : 0xBF800000u);
                const bool touch = lo >= thr;
                const bool all = kind != IT_SEG && lo > 1.0f;
                const bool mix = open && touch && !all, cover = open && touch && all;
                const M bit = M(1) << k;
                st.mixed |= mix ? bit : M(0);
                st.line |= (mix && kind == IT_SEG) ? 1 : 0;
                st.bk = cover ? k : st.bk;
                st.decided |= cover ? 1 : 0;
                lo = BIG_F;
                st.lo = lo;
                if (__all(st.decided || !active)) break;
            } else {
                // pixel level: the gate form (classify_tile_direct) -- a covered lane's later lo are far outside; the kind is wave-uniform
                const float l2 = lo + st.gate;
                const M bit = M(1) << k;
                if (kind == IT_SEG) {
                    const bool mix = l2 >= 0.0f;
                    st.mixed |= mix ? bit : M(0);
                    st.line |= mix ? 1 : 0;
                } else {
                    const bool cover = l2 > 1.0f;
                    st.bk = cover ? k : st.bk;
                    st.gate = cover ? -__builtin_inff() : st.gate;
                    st.mixed |= __builtin_fabsf(l2) <= 1.0f ? bit : M(0);
                }
                lo = BIG_F;
                st.lo = lo;
                if (__all(st.gate < 0.0f || !active)) break;
            }
        } else {
            st.lo = lo;                                  // the primitive continues in the next 64-item chunk
        }
    }
}

// Phase T's classification of a mixed tile's 64 pixels WITHOUT gathering the items into lanes first (round 5): a scalar walk down the
// tile's primitive set, the items read straight from LDS at wave-uniform addresses (one 16-byte read per polygon edge: the second half of
// its record), the values used as vector operands.  Against the gather + v_readlane form above: no per-lane index computation per
// primitive (8 vector + 10 scalar instructions), no item load into lanes, no meta decode per run, 3 v_readlane + v_mov less per edge
// -- a wavefront's time follows its instruction count.  Same arithmetic per item and the same verdict per convex part, in the same order.
template <typename M>
__device__ __forceinline__ void classify_tile_direct(const Raster &rs, M tmixed, float xc, float yc, PixState<M> &st) {
    const float4 *items4 = reinterpret_cast<const float4 *>(&RI(items, 0));       // (two float4 per item)
    M m = mask_uniform<M>(tmixed);
    // A lane's verdict on primitive k from lo, the smallest of its edge functions over the lane's 4x4 block (in units where +-1 is the
    // block's reach): > 1 the primitive covers the block, < -1 it misses it, between them the pixel is undecided.  A lane that a
    // primitive has covered is done: `gate` (0, then -inf) pushes every later lo of that lane far outside, so that no verdict has to ask
    // first -- seven vector instructions and a scalar compare where the lane-mask form had twelve and ten.
    auto verdict = [&](int, int k, float lo) -> bool {
        const float l2 = lo + st.gate;
        const bool cover = l2 > 1.0f;
        st.bk = cover ? k : st.bk;
        st.gate = cover ? -__builtin_inff() : st.gate;
        st.mixed |= __builtin_fabsf(l2) <= 1.0f ? M(1) << k : M(0);
        return __all(st.gate < 0.0f);
    };
    while (m) {
        const int k = mask_top(m);
        m &= ~(M(1) << k);
        const int pi = __builtin_amdgcn_readfirstlane(RI(pitem, k));
        const int start = pi & 0xFFFF, cnt = (pi >> 16) & PI_CNT_MASK, pkind = (pi >> 24) & 3, multi = (pi >> 26) & 1;
        if (pkind == PR_POLY && !multi) {
            // one convex part: cnt edges, two per turn
            float lo = BIG_F;
            int e = 0;
            for (; e + 1 < cnt; e += 2) {
                const float4 g = items4[2 * (start + e) + 1], h = items4[2 * (start + e) + 3];
                lo = raw_min(lo, __builtin_fmaf(g.x, xc, __builtin_fmaf(g.y, yc, g.z)));
                lo = raw_min(lo, __builtin_fmaf(h.x, xc, __builtin_fmaf(h.y, yc, h.z)));
            }
            if (e < cnt) { const float4 g = items4[2 * (start + e) + 1]; lo = raw_min(lo, __builtin_fmaf(g.x, xc, __builtin_fmaf(g.y, yc, g.z))); }
            if (verdict(IT_EDGE, k, lo)) break;
        } else if (pkind == PR_NGON) {
            const float4 u = items4[2 * start], g = items4[2 * start + 1];          // a b c g3 | g0 ...
            const float qx = __builtin_fabsf(xc - u.x), qy = __builtin_fabsf(yc - u.y);
            const float nx = r_max(qx - 1.5f, 0.0f), ny = r_max(qy - 1.5f, 0.0f);
            const float fx = qx + 1.5f, fy = qy + 1.5f;
            const float apo = u.z - CLASS_EPS_F, rad = g.x + CLASS_EPS_F;
            const float l = rad * rad - (nx * nx + ny * ny);
            const float hh = apo > 0.0f ? apo * apo - (fx * fx + fy * fy) : -1.0f;
            const float lo = l < 0.0f ? -2.0f : (hh > 0.0f ? 2.0f : 0.0f);
            if (verdict(IT_NGON, k, lo)) break;
        } else {
            // several convex parts (a star): a verdict at every part's last edge, as the item stream's IT_LAST marks them
            // (line loops never come here: a tile with one among its primitives takes the gather form, mgx_raster_body.inc)
            float lo = BIG_F;
            bool stop = false;
            for (int e = 0; e < cnt; e++) {
                const float4 g = items4[2 * (start + e) + 1];
                lo = raw_min(lo, __builtin_fmaf(g.x, xc, __builtin_fmaf(g.y, yc, g.z)));
                if (__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, g.w)) & IT_LAST) {
                    if (verdict(IT_EDGE, k, lo)) { stop = true; break; }
                    lo = BIG_F;
                }
            }
            if (stop) break;
        }
    }
}

// Phase C's form: NT tiles per lane (144 tiles = 48 lanes x 3), so that every wavefront sees the whole tile grid and the ITEM LIST can be
// dealt out over the workgroup's four wavefronts instead (mgx_raster_body.inc).  Per item three broadcasts + (2 FMA + 1 min) per tile.
// Same arithmetic per tile as classify_items_regs<true>; consumes items [0, n) of `src`, which never end inside a convex part.
template <int NT, typename M>
__device__ __forceinline__ void classify_items_regs_tiles(const Raster &rs, const RegItems &src, int n, const float (&xc)[NT], const float (&yc)[NT],
                                                          PixState<M> (&st)[NT], bool active, uint32_t *tile_seg, uint64_t lm, int lane) {
    auto bc = [&](float v, int i) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i)); };
    const float hx = TILE_HX, hy = TILE_HY;
    int i = 0;
    while (i < n) {
        const int meta = __builtin_amdgcn_readlane(src.my.meta, i);
        const int kind = meta & 3, rem = (meta >> IT_REM_SHIFT) & IT_REM_MASK, k = meta >> IT_K_SHIFT;
        const int run = rem < n - i ? rem : n - i;
        float lo[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) lo[t] = st[t].lo;
        if (kind == IT_EDGE) {
            for (int j = i; j < i + run; j++) {
                const float a = bc(src.my.a, j), b = bc(src.my.b, j), c = bc(src.my.c, j), g3 = bc(src.my.g3, j);
#pragma unroll
                for (int t = 0; t < NT; t++) lo[t] = raw_min(lo[t], __builtin_fmaf(a, xc[t], __builtin_fmaf(b, yc[t], c)) * g3);
            }
        } else if (kind == IT_NGON) {
            const float ca = bc(src.my.a, i), cb = bc(src.my.b, i);
            const float apo = bc(src.my.c, i) - CLASS_EPS_F, rad = bc(src.my.g0, i) + CLASS_EPS_F;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const float qx = __builtin_fabsf(xc[t] - ca), qy = __builtin_fabsf(yc[t] - cb);
                const float nx = r_max(qx - hx, 0.0f), ny = r_max(qy - hy, 0.0f);
                const float fx = qx + hx, fy = qy + hy;
                const float l = rad * rad - (nx * nx + ny * ny);
                const float h = apo > 0.0f ? apo * apo - (fx * fx + fy * fy) : -1.0f;
                lo[t] = l < 0.0f ? -2.0f : (h > 0.0f ? 2.0f : 0.0f);
            }
        } else {
            const int nb = seg_note_base(rs, lm, k);
            const int e0 = ((RI(pitem, k) >> 16) & PI_CNT_MASK) - rem;
            for (int j = i; j < i + run; j++) {
                const float a = bc(src.my.a, j), b = bc(src.my.b, j);
                const float hw = bc(src.my.g3, j) + CLASS_EPS_F;
                const float c = bc(src.my.c, j), g0 = bc(src.my.g0, j), g1 = bc(src.my.g1, j), g2 = bc(src.my.g2, j);
                const float el = hx * __builtin_fabsf(a) + hy * __builtin_fabsf(b), es = hx * __builtin_fabsf(b) + hy * __builtin_fabsf(a);
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    const float e = a * xc[t] + (b * yc[t] + c);
                    const float sl = (xc[t] - g0) * b - (yc[t] - g1) * a;
                    const float tt = r_min(r_min(hw + el - __builtin_fabsf(e), sl + es + hw), g2 + hw + es - sl);
                    lo[t] = lo[t] >= BIG_F ? tt : r_max(lo[t], tt);
                    if (nb >= 0 && active && tt >= 0.0f) atomicOr(&tile_seg[lane + t * (N_TILES / NT)], 1u << (nb + e0 + (j - i)));
                }
            }
        }
        i += run;
        if (run == rem) {
            bool all_decided = true;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const bool open = !st[t].decided;
                const bool touch = kind == IT_SEG ? lo[t] >= 0.0f : !(lo[t] < -1.0f);
                const bool all = kind != IT_SEG && lo[t] > 1.0f;
                const bool mix = open && touch && !all, cover = open && touch && all;
                const M bit = M(1) << k;
                st[t].mixed |= mix ? bit : M(0);
                st[t].bk = cover ? k : st[t].bk;
                st[t].decided |= cover ? 1 : 0;
                st[t].lo = BIG_F;
                all_decided = all_decided && st[t].decided;
            }
            if (__all(all_decided || !active)) break;
        } else {
#pragma unroll
            for (int t = 0; t < NT; t++) st[t].lo = lo[t];
        }
    }
}

template <typename P, int LAYOUT, int WAVES, typename M>
__global__ __launch_bounds__(256, WAVES) void k_raster(RasterDev t, const P *__restrict__ sp, uint8_t *__restrict__ out,
                                                long env_stride, int view, const uint8_t *__restrict__ fill_mask, int n_envs, RasterHandoff ho) {
    extern __shared__ __align__(16) uint32_t lds[];
    const int tid = threadIdx.x;
    // phase clocks of a workgroup (tools/dev/raster_phase_clocks.py, fused_timeline.py): in -DMGX_RASTER_CLOCKS / -DMGX_RASTER_PROBE builds only
    // -- the start time and the table pointer are four scalar registers held from the first instruction to the last, in a kernel that
    // spills seventy of them
#if defined(MGX_RASTER_PROBE) && !defined(MGX_RASTER_CLOCKS)
#define MGX_RASTER_CLOCKS 1
#endif
#ifdef MGX_RASTER_CLOCKS
    unsigned long long clk0 = wall_clock64();
#endif
#ifdef MGX_RASTER_PROBE
#define PROBE(...) __VA_ARGS__
#else
#define PROBE(...)
#endif
#ifdef MGX_Q_NO_STORE      // development probe: what the byte patches of phases Q / E cost (the frame is wrong without them)
#define Q_STORE(x) do { if (c == 0x7FFFFFFF) { x; } } while (0)
#else
#define Q_STORE(x) x
#endif
#ifdef MGX_RASTER_MARKERS   // development aid: names in the assembly (hipcc -S) to find a phase's instructions by
#define RMARK(name) asm volatile("; MGX_MARK " #name);
#else
#define RMARK(name)
#endif
#if defined(MGX_RASTER_PROBE)   // development build: also allows truncating the kernel after phase i (tools/raster_phase_probe.py)
#define CLK(i) if (t.dbg_clk && tid == 0) t.dbg_clk[blockIdx.x * 16 + (i)] = wall_clock64() - clk0; if ((i) > 0 && t.dbg_stop == (i)) return;
#elif defined(MGX_RASTER_CLOCKS)
#define CLK(i) if (t.dbg_clk && tid == 0) t.dbg_clk[blockIdx.x * 16 + (i)] = wall_clock64() - clk0;
#else
#define CLK(i)
#endif
    long env = blockIdx.x;
#ifdef MGX_RASTER_CLOCKS
    if (t.dbg_clk && tid == 0) t.dbg_clk[blockIdx.x * 16 + 9] = clk0;        // (absolute: tools/dev/fused_timeline.py)
#ifndef MGX_RASTER_PROBE
    if (t.dbg_clk && (tid & 63) == 0)                                        // where each wavefront runs (slots 6, 7, 8, 10; tools/dev/placement_probe.py)
        t.dbg_clk[blockIdx.x * 16 + ((tid >> 6) < 3 ? 6 + (tid >> 6) : 10)] = __builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
#endif
#endif
#if defined(MGX_HANG_DEBUG)
    const unsigned hwid_at_start = __builtin_amdgcn_s_getreg(10244);
#endif
    HDBG(0)
    // the shared draw list does not depend on the env: stage it before waiting for the hand-off
    if (!t.tmpl_stride_words) { const int n = raster_staged_words(t, t.words); for (int i = tid; i < n; i += 256) lds[i] = t.words[i]; }
    if (ho.mode) {
        // one lane waits / decides, one agent-scope acquire per workgroup, then plain loads (cdna guide, G16)
        int32_t *slot = reinterpret_cast<int32_t *>(lds + t.lds_tmpl_words + t.off_tiles) + (N_TILES * (1 + (int)(sizeof(M) / 4)) + N_TILES / 4 + t.qcap_lds * (1 + (int)(sizeof(M) / 4)) + 5);    // = q_count[5], unused below
        if (tid == 0) {
            long got = -1;
            if (ho.mode == 1) {
                // An entry that is already there is taken at once.  Otherwise the wait is long only once every producer is
                // resident (then entry blockIdx.x is certain to come); while some producer has not begun -- it may be waiting for
                // the very resources this workgroup holds -- the wait is a short bounded one (the two kernels are released
                // together and the producers need a microsecond to begin), after which the env is left to the clean-up launch.
                bool all = false;
                for (unsigned n = 0; n < ho.poll_limit; n++) {
                    const unsigned long long ent = __hip_atomic_load(&ho.queue[blockIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(ent >> 32) == ho.epoch) { got = (long)(ent & 0xFFFFFFFFull); break; }
                    if (!all) {
                        all = __hip_atomic_load(ho.started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ho.started_base >= ho.n_producers;
                        if (!all && n >= 64) break;
                    }
                    if (all) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(32);
                }
                if (got < 0 && all) atomicAdd(&ho.stats[1], 1u);
                if (got < 0) { ho.deferred[blockIdx.x] = ho.epoch; atomicAdd(&ho.stats[0], 1u); }
                else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            *slot = (int32_t)got;
        }
        __syncthreads();
        env = __builtin_amdgcn_readfirstlane(*slot);      // (the same in every lane: keeps the env's frame pointer and row offsets on the scalar unit)
        if (env < 0) return;
        __syncthreads();           // (the slot is reused as a counter below)
    }
    HDBG(1)
#ifndef MGX_T_DYNAMIC
#define MGX_T_DYNAMIC 1
#endif
#ifndef MGX_Q_DYNAMIC
#define MGX_Q_DYNAMIC 1
#endif
    constexpr bool SCALAR_WAVE = WAVES < 5, T_UNIFORM_FAST = WAVES >= 5, E_SPLIT = WAVES < 5, T_DYNAMIC = MGX_T_DYNAMIC, Q_DYNAMIC = MGX_Q_DYNAMIC;
#include "mgx_raster_body.inc"
    HDBG(3)
#if defined(MGX_HANG_DEBUG)
    if (ho.mode && (threadIdx.x & 63) == 0 && __builtin_amdgcn_s_getreg(10244) != hwid_at_start) atomicAdd(&ho.stats[2 + 7], 1u);      // a wavefront that moved
#endif
#undef CLK
}

// the clean-up launch of mgx_engine_step_render: workgroup b rasterises queue[b]'s env iff consumer b gave up (normally none: 4.4 us,
// the floor of a launch); a kernel of its own, so that the launch statistics of k_raster are those of real work (one occupancy
// variant serves every world).  One workgroup per entry: where several engines share the GPU (the task fleet of BASELINE.json's
// configs[4]) consumers give up by the hundred, and a launch of 16 workgroups that each walk 256 entries -- tried: no faster when
// there is nothing to do -- rasterised them one after the other (8 x 1024 envs on one GPU: 3.59 -> 2.74 M env-steps/s)
template <typename P, int LAYOUT, typename M>
__global__ __launch_bounds__(256, 3) void k_raster_deferred(RasterDev t, const P *__restrict__ sp, uint8_t *__restrict__ out,
                                                   long env_stride, int view, int n_envs, RasterHandoff ho) {
    extern __shared__ __align__(16) uint32_t lds[];
    const int tid = threadIdx.x;
#define CLK(i)
#ifdef MGX_RASTER_PROBE
    const unsigned long long clk0 = wall_clock64();     // (the body's PROBE lines refer to it)
#endif
    HDBG(4)
    if (ho.deferred[blockIdx.x] != ho.epoch) { HDBG(5) return; }                        // (workgroup-uniform)
    const long env = (long)(ho.queue[blockIdx.x] & 0xFFFFFFFFull);         // written by a kernel that has completed
    const uint8_t *fill_mask = nullptr;
    if (!t.tmpl_stride_words) { const int n = raster_staged_words(t, t.words); for (int i = tid; i < n; i += 256) lds[i] = t.words[i]; }
    constexpr bool SCALAR_WAVE = true, T_UNIFORM_FAST = false, E_SPLIT = false, T_DYNAMIC = false, Q_DYNAMIC = false;
#include "mgx_raster_body.inc"
#undef CLK
}

// 384x384x3 point-sampled frame of ONE env (no box filter): parity tests against the oracle / reference PNGs
template <typename P>
__global__ __launch_bounds__(256, MGX_RASTER_WAVES) void k_raster_native(RasterDev t, const P *__restrict__ sp, uint8_t *__restrict__ out,
                                                       int view, long env, int n_envs) {
    extern __shared__ __align__(16) uint32_t lds[];
    const int tid = threadIdx.x;
    {
        const uint32_t *src = t.words + env * t.tmpl_stride_words;
        const int n = raster_staged_words(t, src);
        for (int i = tid; i < n; i += 256) lds[i] = src[i];
    }
    __syncthreads();
    const TmplHeader *h = reinterpret_cast<const TmplHeader *>(lds);
    Raster rs(h, reinterpret_cast<const int32_t *>(lds + t.off_i), raster_tq(t, t.words + env * t.tmpl_stride_words, lds, *h),
              reinterpret_cast<double *>(lds + t.lds_tmpl_words),
              reinterpret_cast<int32_t *>(lds + t.lds_tmpl_words + 2 * t.scratch_d), view, t.compact != 0);
    raster_setup_bodies<P>(rs, sp, (long)n_envs, env, tid, 256);
    __syncthreads();
    raster_setup_prims(rs, tid, 256, t.ent_colour_env, (long)n_envs, env, t.goal_xyhw_env, t.palette);
    __syncthreads();
    raster_setup_edges(rs, tid, 256);
    __syncthreads();
    const int pix = blockIdx.x * 256 + tid;
    if (pix >= NATIVE_RES * NATIVE_RES) return;
    const int col = pix % NATIVE_RES, row = pix / NATIVE_RES;
    const uint64_t all = h->n_prims >= 64 ? ~0ull : ((1ull << h->n_prims) - 1ull);
    const int c = raster_sample<uint64_t>(rs, col + 0.5, (double)(NATIVE_RES - 1 - row) + 0.5, all, t.bg_rgb);
    out[3 * pix] = c & 0xFF; out[3 * pix + 1] = (c >> 8) & 0xFF; out[3 * pix + 2] = (c >> 16) & 0xFF;
}

}  // namespace mgx

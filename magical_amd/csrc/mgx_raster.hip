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
    const uint32_t *words;   // [TmplHeader][int words][pad][tq doubles]
    int n_words;
    int off_i, off_q;        // word offsets of the int array and the double array
    int lds_tmpl_words;      // reserved words for the template (even)
    int scratch_d;           // doubles of per-env scratch
    int bg_rgb;
    int off_tiles;           // word offset (inside the scratch area) of the per-tile / queue region, 8-byte aligned
};

constexpr int N_TILES = TILES_X * TILES_Y;
constexpr int QCAP = 1024;     // LDS queue of undecided pixels per env (overflow is resolved in place)

// FlattenFrameStack shift of one pixel: 12 B read-modify-write (or 4 copies of the frame after a reset)
__device__ __forceinline__ void store_stack4(uint8_t *frame, int X, int Y, int c, bool fill) {
    uint32_t *px = reinterpret_cast<uint32_t *>(frame + (long)(Y * LORES + X) * 12);
    const uint32_t r = c & 0xFF, g = (c >> 8) & 0xFF, b = (c >> 16) & 0xFF;
    uint32_t d0, d1, d2;
    if (fill) {
        d0 = r | (g << 8) | (b << 16) | (r << 24);
        d1 = g | (b << 8) | (r << 16) | (g << 24);
        d2 = b | (r << 8) | (g << 16) | (b << 24);
    } else {
        const uint32_t o0 = px[0], o1 = px[1], o2 = px[2];
        d0 = (o0 >> 24) | (o1 << 8);
        d1 = (o1 >> 24) | (o2 << 8);
        d2 = (o2 >> 24) | ((uint32_t)c << 8);
    }
    px[0] = d0; px[1] = d1; px[2] = d2;
}
__device__ __forceinline__ void store_frame_px(uint8_t *frame, int X, int Y, int c) {
    uint8_t *q = frame + (long)(Y * LORES + X) * 3;
    q[0] = c & 0xFF; q[1] = (c >> 8) & 0xFF; q[2] = (c >> 16) & 0xFF;
}

template <typename P, int LAYOUT>
__global__ __launch_bounds__(256) void k_raster(RasterDev t, const P *__restrict__ sp, uint8_t *__restrict__ out,
                                                long env_stride, int view, const uint8_t *__restrict__ fill_mask, int n_envs) {
    extern __shared__ __align__(16) uint32_t lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < t.n_words; i += 256) lds[i] = t.words[i];
    __syncthreads();
    const long env = blockIdx.x;
    const TmplHeader *h = reinterpret_cast<const TmplHeader *>(lds);
    uint32_t *scratch = lds + t.lds_tmpl_words;
    Raster rs(h, reinterpret_cast<const int32_t *>(lds + t.off_i), reinterpret_cast<const double *>(lds + t.off_q),
              reinterpret_cast<double *>(scratch), reinterpret_cast<int32_t *>(scratch + 2 * t.scratch_d), view);
    // phase-local LDS: per-tile results and the queue of undecided pixels
    uint64_t *tile_mixed = reinterpret_cast<uint64_t *>(scratch + t.off_tiles);
    int32_t *tile_base = reinterpret_cast<int32_t *>(tile_mixed + N_TILES);
    uint64_t *q_mask = reinterpret_cast<uint64_t *>(tile_base + N_TILES);
    int32_t *q_pix = reinterpret_cast<int32_t *>(q_mask + QCAP);
    int32_t *q_base = q_pix + QCAP;
    int32_t *q_count = q_base + QCAP;
    if (tid == 0) *q_count = 0;
    // phase S: screen-space setup (lane per body, then lane per primitive)
    raster_setup_bodies<P>(rs, sp, (long)n_envs, env, tid, 256);
    __syncthreads();
    raster_setup_prims(rs, tid, 256);
    __syncthreads();
    // phase C: one lane per 16x4 tile classifies every primitive against it (uniform loops, broadcast LDS reads)
    if (tid < N_TILES) {
        int base; uint64_t mixed;
        classify_tile_all(rs, tid, t.bg_rgb, base, mixed);
        tile_base[tid] = base; tile_mixed[tid] = mixed;
    }
    __syncthreads();

    // phase T: each wavefront walks its tiles, one lane per output pixel
    const int wave = tid >> 6, lane = tid & 63;
    const int tx = lane & (TILE_W - 1), ty = lane >> 4;
    const bool fill = LAYOUT == 1 && fill_mask != nullptr && fill_mask[env] != 0;
    uint8_t *frame = out + env * env_stride;
    for (int tile = wave; tile < N_TILES; tile += 4) {
        const int tcol = tile % TILES_X, trow = tile / TILES_X;
        const int X = tcol * TILE_W + tx, Y = trow * TILE_H + ty;
        const uint64_t tmixed = tile_mixed[tile];
        int c = tile_base[tile];
        if (tmixed == 0) {
            // decided for the whole tile: packed stores
            if (LAYOUT == 0) {
                // 16 pixels x 3 B = 12 dwords per tile row: lanes tx < 12 each assemble one dword
                const int d = tx < 12 ? tx : 0;
                const int p0 = (4 * d) / 3, o = (4 * d) % 3;
                const uint32_t c0 = (uint32_t)__shfl(c, (ty << 4) + p0), c1 = (uint32_t)__shfl(c, (ty << 4) + p0 + 1);
                uint32_t w = o == 0 ? (c0 | (c1 << 24)) : (o == 1 ? ((c0 >> 8) | (c1 << 16)) : ((c0 >> 16) | (c1 << 8)));
                if (tx < 12) reinterpret_cast<uint32_t *>(frame + (long)(Y * LORES + tcol * TILE_W) * 3)[d] = w;
            } else {
                store_stack4(frame, X, Y, c, fill);
            }
            continue;
        }
        const uint64_t pmixed = pixel_classify(rs, X, Y, tmixed, c);
        if (pmixed == 0) {
            if (LAYOUT == 0) store_frame_px(frame, X, Y, c); else store_stack4(frame, X, Y, c, fill);
        } else {
            // undecided: hand the pixel to phase Q so that finished lanes do not wait for it
            int slot = atomicAdd(q_count, 1);
            if (slot < QCAP) {
                q_mask[slot] = pmixed; q_pix[slot] = X | (Y << 8); q_base[slot] = c;
            } else {
                c = pixel_resolve(rs, X, Y, pmixed, c);
                if (LAYOUT == 0) store_frame_px(frame, X, Y, c); else store_stack4(frame, X, Y, c, fill);
            }
        }
    }
    __syncthreads();
    // phase Q: all 256 lanes resolve the queued pixels (16 samples each)
    const int nq = *q_count < QCAP ? *q_count : QCAP;
    for (int i = tid; i < nq; i += 256) {
        const int X = q_pix[i] & 0xFF, Y = q_pix[i] >> 8;
        const int c = pixel_resolve(rs, X, Y, q_mask[i], q_base[i]);
        if (LAYOUT == 0) store_frame_px(frame, X, Y, c); else store_stack4(frame, X, Y, c, fill);
    }
}

// 384x384x3 point-sampled frame of ONE env (no box filter): parity tests against the oracle / reference PNGs
template <typename P>
__global__ __launch_bounds__(256) void k_raster_native(RasterDev t, const P *__restrict__ sp, uint8_t *__restrict__ out,
                                                       int view, long env, int n_envs) {
    extern __shared__ __align__(16) uint32_t lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < t.n_words; i += 256) lds[i] = t.words[i];
    __syncthreads();
    const TmplHeader *h = reinterpret_cast<const TmplHeader *>(lds);
    Raster rs(h, reinterpret_cast<const int32_t *>(lds + t.off_i), reinterpret_cast<const double *>(lds + t.off_q),
              reinterpret_cast<double *>(lds + t.lds_tmpl_words),
              reinterpret_cast<int32_t *>(lds + t.lds_tmpl_words + 2 * t.scratch_d), view);
    raster_setup_bodies<P>(rs, sp, (long)n_envs, env, tid, 256);
    __syncthreads();
    raster_setup_prims(rs, tid, 256);
    __syncthreads();
    const int pix = blockIdx.x * 256 + tid;
    if (pix >= NATIVE_RES * NATIVE_RES) return;
    const int col = pix % NATIVE_RES, row = pix / NATIVE_RES;
    const uint64_t all = h->n_prims >= 64 ? ~0ull : ((1ull << h->n_prims) - 1ull);
    const int c = raster_sample(rs, col + 0.5, (double)(NATIVE_RES - 1 - row) + 0.5, all, t.bg_rgb);
    out[3 * pix] = c & 0xFF; out[3 * pix + 1] = (c >> 8) & 0xFF; out[3 * pix + 2] = (c >> 16) & 0xFF;
}

}  // namespace mgx

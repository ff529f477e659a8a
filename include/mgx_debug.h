/* mgx_debug.h -- development / test hooks of libmagical_hip.so.  NOT part of the drop-in boundary (include/mgx.h):
 * nothing in magical_amd/ calls these; tests/ and tools/ do.
 *
 * All return MGX_OK.  Defaults restore the shipped behaviour. */
#ifndef MGX_DEBUG_H
#define MGX_DEBUG_H
#include "mgx.h"
#ifdef __cplusplus
extern "C" {
#endif

/* k_raster: entries of the undecided-pixel queue / of the uncertain-sample record list actually used (<= the compiled
 * capacities, clamped).  tests/test_gpu_parity.py shrinks them to 1 to force the overflow rounds. */
int mgx_engine_debug_raster_qcap(mgx_engine *e, int n);
int mgx_engine_debug_raster_ecap(mgx_engine *e, int n);
/* k_raster: DEVICE u64 [N][16] written by lane 0 of every workgroup at the end of each phase (100 MHz wall clock since
 * kernel start; [5] = queued pixels); NULL = off.  tools/raster_probe.py */
int mgx_engine_debug_raster_clocks(mgx_engine *e, void *buf);
/* k_raster, -DMGX_RASTER_PROBE builds only: return after phase `phase` (1 S, 2 C, 3 T, 4 Q+E; 9 = no per-pixel
 * classification), 0 = run everything.  tools/raster_phase_probe.py */
int mgx_engine_debug_raster_stop(mgx_engine *e, int phase);
/* k_step, -DMGX_STEP_PROBE builds only: DEVICE u64 [workgroups][32] shader cycles per phase; NULL = off.
 * tools/step_phase_probe.py */
int mgx_engine_debug_step_clocks(mgx_engine *e, void *buf);
/* k_raster: force the occupancy variant (3, 4 or 5 workgroups per CU = VGPR caps 168 / 128 / 96) instead of the one the
 * world's LDS footprint selects; the world must still fit that many workgroups.  tools/raster_probe.py */
int mgx_engine_debug_raster_waves(mgx_engine *e, int n);
/* k_step: override the solver iteration count (-1 = the reference's 10).  tools/step_probe.py */
int mgx_engine_debug_iterations(mgx_engine *e, int it);
/* the fused env-step's hand-off counters read on a stream of its own -- works while the caller's streams are stuck (tools/dev/hang_hunt.py).
 * out[16]: [0..3] the device's tail / started / deferred / timeouts, [4..6] the host's mirrors of tail, started, epoch, [7] queue entries
 * that carry the current epoch, [8..15] per-phase workgroup counters of -DMGX_HANG_DEBUG builds (0 otherwise) */
int mgx_engine_debug_handoff_peek(mgx_engine *e, unsigned *out);

/* threads the host pool (world builds, placement sampling at a reset) takes in this process: hardware threads / LOCAL_WORLD_SIZE (the ranks
 * that share the node), within [1, MGX_HOST_THREADS (64)]; `allocating` 0 = the placement-only bound MGX_PLACE_THREADS.  No GPU needed. */
int mgx_debug_host_threads(int allocating);
/* the fused env-step's hand-off, failure paths forced (tests/test_gpu_parity.py, tools/dev/hang_hunt.py --fleet): poll_limit > 0 = polls a
 * consumer workgroup spends on its queue entry before it gives the env up to the clean-up launch (shipped: 2^16; 1 = every consumer that does
 * not find its entry at once gives up: `deferred` and -- once all producers are resident -- `timeouts` of mgx_engine_handoff_stats count);
 * delay_every > 0 = every delay_every-th step workgroup sleeps delay_sleeps x ~1 us before it publishes its envs (late producers).
 * 0, 0, 0 restores the shipped behaviour.  The observations are the two-call result byte for byte whatever these are set to. */
int mgx_engine_debug_handoff(mgx_engine *e, int poll_limit, int delay_every, int delay_sleeps);

#ifdef __cplusplus
}
#endif
#endif

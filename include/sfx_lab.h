/* sfx_lab.h -- the LAB build of libsfx (libsfx_lab.so: csrc/build.sh with SFX_LAB=1, i.e. -DSFX_LAB).
 *
 * The product library (libsfx.so, include/sfx.h) has ONE form of every step and reads no environment variable.  The lab build is
 * the same source plus what the measurements of rounds 2-5 were made with: interchangeable forms of two steps (bit for bit the
 * product's results -- tests/ check that on this build: tools/run_gpu_suite.sh), phase clocks inside the kernels, and the
 * switches below.  None of them changes a result.  LAB_NOTES.md holds what was measured with them.
 *
 * Environment switches of the lab build (read once per process unless noted):
 *   SFX_LBS_DENSE=16|17|32   which dense LBS kernel the rounds launch (= sfx_debug_lbs_dense_form)
 *   SFX_LBS_W=3|4|5          one workgroup width for every launch of k_lbs_dense16
 *   SFX_TICK_THREADS=256     the four-wavefront k_tick_dense of round 3 at <= 256 frames
 *   SFX_ADJ_KI=4             the 64-k tile of the adjoint GEMM (rounds 3-5)
 *   SFX_PEN_FORM=0|1|2       form of the interpenetration step for new handles (= sfx_debug_pen_form)
 *   SFX_PEN_GRAPH=0          the interpenetration step launched kernel by kernel instead of as a captured graph
 *   SFX_PEN_FLAT_OFF, SFX_PEN_WALK_CHUNKS_OFF, SFX_PEN_ROWS_OFF (per call)   one grid row per column instead of flat work lists
 *   SFX_PEN_FAST_PAIRS=n     forms 1 / 2: columns with more than n pairs go to the general kernels
 *   SFX_POLL_ROUNDS=n, SFX_POLL_AHEAD=n   rounds per polled batch / batches queued ahead in the dense fitting loop
 *   SFX_DEBUG_NACT, SFX_DEBUG_HOST        loop diagnostics on stderr
 * Deleted in round 6 (result-changing, or measured and closed; the last tree that holds them is commit e8a08e9):
 *   SFX_PEN_REWALK_OFF, SFX_HIST_ZEROPAGE, SFX_PEN_BRANCHES, SFX_PEN_FORM=3 (k_pen_frame), sfx_fit_multi / sfx_debug_overlap_test.
 */
#ifndef SFX_LAB_H_
#define SFX_LAB_H_
#include "sfx.h"
#ifdef __cplusplus
extern "C" {
#endif

/* debug: elapsed 100 MHz wall-clock ticks at the end of k_pen_grid's seven steps (triangle boxes,
 * frame box, part boxes, part culling, grid histogram, scan, scatter; [7..9] unused) of the most
 * recent evaluation, then the number of grid entries: HOST [B][11].                               */
int  sfx_pen_phase_clocks(sfx_pen* h, int32_t B, int32_t* clocks_host);


/* Debug: shader-clock stamps at the phase boundaries of one closure launch (block 0). */
int  sfx_debug_phase_clocks(sfx_batch* b, int32_t stage, int64_t* out /* [32] */);

/* Debug / A-B measurements: which dense LBS kernel the rounds launch -- 16 = k_lbs_dense16 (16 frames per wavefront: the
 * product kernel; at <= 32 active frames its form with one coordinate per wavefront, k_lbs_dense16c), 17 = k_lbs_dense16 at
 * every size, 32 = k_lbs_dense (32 frames per wavefront).  The same chain of fp32 operations per vertex and frame in all of
 * them, so the same bits.  Process-wide; any other value only queries.  Returns the previous setting.                   */
int  sfx_debug_lbs_dense_form(int32_t form);

/* Debug / A-B measurements: which form of the interpenetration term handles and batches created FROM NOW ON take -- 0 = the
 * product's: the general kernels on every column, every step dealt flat over the chip; 1 = grid build and pair tests spread over
 * the chip, then ONE workgroup per column from the accepted pairs to the gradient (k_pen_narrow), plus the general kernels on the
 * columns it hands over; 2 = form 1 with every column handed over.  Form 1 was built in round 5 and measured slower on whole fits
 * (a round lasts as long as its most crowded column); all forms produce the same bits (pair list, loss, gradients:
 * tests/test_gpu_topology.py).  Any other value only queries.  Returns the previous setting.  Environment: SFX_PEN_FORM.     */
int  sfx_debug_pen_form(int32_t form);
/* Debug: 100-MHz ticks the workgroups of k_pen_narrow spent in their phases since sfx_pen_work_reset, summed over the column
 * evaluations: HOST [8] = entry, pair list, pair evaluation, triangle sums, vertices + loss, evaluations, their ordered pairs, 0. */
int  sfx_debug_pen_phase_ticks(int64_t* ticks_host);

/* Debug: attach (enable>=1) a 64-slot clock buffer to the batch, run any entry point, then read it
 * and detach (enable=0): out[0..18] = closure phase stamps of the last launch, out[32+i] =
 * shader-clock cycles frame 0 spent between optimiser-tick marks i-1 and i, out[63] = ticks.
 * enable = N > 1: the stamps of the dense tick kernel freeze after its N-th launch (default 40),
 * out[24..26] = its start / end of adjoint+tick / end, out[40..56] = phases of its adjoint pass.  */
int  sfx_debug_clocks(sfx_batch* b, int32_t enable, int64_t* out /* [64] or NULL */);

#ifdef __cplusplus
}
#endif
#endif /* SFX_LAB_H_ */

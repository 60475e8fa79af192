/*
 * prmf_hip_debug.h -- validation and measurement hooks exported by libprmf_hip.so next to the product ABI
 * (include/prmf_hip.h).  Nothing on the product path calls them: the full-size parity tests read the sweeps' products
 * through prmf_debug_slab_sum, the tools under tools/ use the rest.  Same conventions as prmf_hip.h (0 or a negative
 * prmf_status, never throws).
 */
#ifndef PRMF_HIP_DEBUG_H
#define PRMF_HIP_DEBUG_H

#include "prmf_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Dense row-major copy of the product the last sweep left in its partial slabs: which = 0: P = X V [m_local][k]
 * (valid between prmf_phase_u and prmf_phase_q), which = 1: Q = X^T U [n][k] (valid between prmf_phase_q and the next
 * prmf_phase_u).  out_dev: float32 device buffer.  tests/test_gpu_fullsize.py compares both with a float64 restatement
 * of prmf_runner.py:420 / :424. */
int prmf_debug_slab_sum(prmf_ctx* ctx, int which, float* out_dev, void* stream);

/* The three terms of the Gram identity behind record `it` of the last deferred-objective batch (single rank, after
 * prmf_end): *cross = <X^T U, V>, gu / gv = U^T U / V^T V (k*k doubles each, host).  tests/test_gpu_recon.py measures
 * each against float64.  PRMF_ERR_STATE if the last batch was not a deferred one. */
int prmf_debug_recon_terms(prmf_ctx* ctx, int it, double* cross, double* gu, double* gv);

/* Device buffer of `n_entries` uint64 receiving, from the next sweep launches: per-workgroup start / end wall-clock stamps
 * ([2 b], [2 b + 1] for workgroup b), where each workgroup ran ([where_offset + b] = XCC_ID << 32 | HW_ID) and, in a
 * -DPRMF_PHASE_PROF=1 build, 8 phase sums per wave at [phase_offset + (8 b + wave) * 8].  The regions are fixed and
 * disjoint; prmf_debug_stamp_layout returns the entries the buffer must have (and the offsets) -- a shorter buffer is
 * refused with PRMF_ERR_ARG.  NULL (any length) switches it off.  tools/wg_spread.py, tools/wg_where.py,
 * tools/sweep_phases.py */
int prmf_debug_set_stamps(void* dev_buf, size_t n_entries);
int prmf_debug_stamp_layout(int* where_offset, int* phase_offset, int* max_workgroups);
/* The same for the fused epilogues: 8 stamps per workgroup.  tools/epi_phases.py */
int prmf_debug_set_epi_stamps(void* dev_buf);

/* Advance the iteration cursor without prmf_phase_post, so that partial iterations can be timed.  tools/sweep_context.py */
int prmf_debug_bump_cursor(prmf_ctx* ctx);

/* Swap one device buffer between two contexts of identical geometry: 0 X, 1 X^T, 2 slabs, 3 packed V, 4 packed U.
 * tools/placement_probe.py */
int prmf_debug_swap(prmf_ctx* a, prmf_ctx* b, int which);

/* The resident copies as the sweeps read them, padding included: which = 0: X, 1: X^T; prmf_debug_resident_shape gives
 * the padded rows and the leading dimension (elements of the storage type), prmf_debug_resident_x copies rows x ld elements
 * to HOST memory (synchronises the device).  tests/test_gpu_setx.py compares every upload path bit for bit. */
int prmf_debug_resident_shape(const prmf_ctx* ctx, int which, int64_t* rows, int64_t* ld);
int prmf_debug_resident_x(prmf_ctx* ctx, int which, void* out_host);

/* The sweep planner's decision for one sweep C = A (rows x depth) * packed factor (depth x k) -- pure host arithmetic, no GPU
 * touched (the CPU test suite pins the plan of every benchmark shape with it).  x_dtype as in prmf_create; n_cus = compute
 * units of the device (256 on MI355X).  out[8] = { waves per workgroup, 32-row tiles per wave, panel rows, workgroups,
 * reduction slices, slabs per panel (upper bound), bit 0: factor chunk double-buffered, bit 1: ping-pong chunk loop, bit 2: the
 * exact short-reduction product (padded depth x column tiles <= 512), panels }. */
int prmf_debug_plan_sweep(int x_dtype, long long rows, long long depth, int k, int n_cus, int* out);

/* Calibration: a pure stream of the resident matrix (which = 0: the padded copy of X, 1: of X^T) -- one workgroup per
 * resident slot of the sweep, every wave instruction 1 KiB contiguous, 8 non-temporal loads in flight per lane: the
 * sweep's loads without its LDS bounce, factor and MFMAs.  One warm-up launch, then `reps` launches between two HIP
 * events on `stream`; *ms_out = average launch time, *bytes_out (nullable) = bytes read per launch.  bench.py reports the
 * sweep's rate as a fraction of this one (`roofline.frac_of_stream`), which separates the kernel from the box it ran on.
 * Synchronises `stream`. */
int prmf_debug_stream(prmf_ctx* ctx, int which, int reps, double* ms_out, int64_t* bytes_out, void* stream);

/* Probe of a hand-off inside the sweep's tail (tools/tail_probe.py): with a device buffer of 64 + 2 * 4096 ints set, every
 * workgroup of the following `which` sweeps (0: X V, 1: X^T U; unsliced plans) publishes its slabs behind a release fence
 * and an agent-scope atomic per panel, claims groups of `tiles_per_group` output tiles from a ticket counter, waits for
 * the group's panel (acquire) and reads its tiles of slab 0 -- what a fused epilogue would pay before any arithmetic.
 * Results are unchanged.  dev_buf[3] != 0 afterwards: a wait gave up.  NULL switches it off. */
int prmf_debug_set_tail_probe(prmf_ctx* ctx, int which, int* dev_buf, int tiles_per_group);

/* The library's bounded wait for a stream that may carry collectives (prmf_set_comm_timeout), callable on its own:
 * PRMF_OK when the stream has drained, PRMF_ERR_HIP (and the communicator aborted) when the time limit passed first.
 * tests/test_gpu_eightway.py drives the timeout path with it. */
int prmf_debug_sync_stream(prmf_ctx* ctx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PRMF_HIP_DEBUG_H */

/*
 * f2v_test.h -- self-test hooks of libf2v.  NOT part of the product library: they exist only in
 * force2vec_amd/libf2v_selftest.so, the same sources built with -DF2V_TEST_HOOKS (Makefile), which tests/ and a few
 * tools/ load instead of libf2v.so where they need fault injection or a look inside.  That build also understands
 * the environment variable F2V_PUSH_CHAOS=<seed>: every rank of f2v_train_sharded drains its stream and sleeps up to
 * 3 ms at random minibatches (other ones on every rank) -- the protocol test under rank skew; F2V_TEST_WITHHOLD_SLOT / F2V_TEST_WITHHOLD_ROW
 * (f2v_test_withhold_flag / _row from outside the process: the CLI's fault tests) and F2V_TEST_RING_REFUSE (the ring of matrices of
 * "wide_epochs" is refused as if the device had no room: f2v_train falls back to one epoch per launch).
 */
#ifndef F2V_TEST_H_
#define F2V_TEST_H_

#include "f2v.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Runs the wavefront tree reduction on `rows` rows of `width` (<=512) floats; out[r] = sum. */
F2V_API int f2v_test_wave_reduce(int device, const float *in, uint32_t rows, uint32_t width, float *out);
/* The push exchange between engines of ONE process on one device (direct pointers instead of HIP IPC; every
 * engine is driven by its own host thread): runs the push kernels, masks and flag barriers under a profiler. */
F2V_API int f2v_test_push_attach_local(f2v_handle h, uint32_t rank, uint32_t world, const f2v_handle *all);
/* PMC calibration: `reps` launches that each gather `rows` distinct 512-byte rows exactly once with
 * the step kernel's access pattern (known HBM read volume rows*516 bytes per launch). */
F2V_API int f2v_test_gather_calibration(int device, uint32_t rows, uint32_t reps);
/* Fault injection for the one-launch minibatch: the hub piece that owns partial-sum slot `slot` never announces its
 * sum (0xFFFFFFFF: none), so the combine-tree node that adds it has to give up its wait ("tree_timeout_ms"). */
F2V_API int f2v_test_withhold_flag(f2v_handle h, uint32_t slot);
/* Fault injection for chained launches: the flag of `row` is never stored (0xFFFFFFFF: none), so every item of a later
 * minibatch of the same launch that reads the row has to give up its wait ("chain_timeout_ms"). */
F2V_API int f2v_test_withhold_row(f2v_handle h, uint32_t row);
/* Timing experiment: chained launches skip their row waits (the results are then WRONG): what the launch structure costs
 * without the dependency chain. */
F2V_API int f2v_test_chain_nowait(f2v_handle h, int on);

/* Where a chained launch spends its time: with `on`, chained launches record per row four words of the 100-MHz device wall
 * clock -- [0] when its last hub piece announced its partial sum, [1] when its last inner combine-tree node did, [2] when its
 * row flag was stored, [3] the bitwise complement of the first time a waiter that had to wait saw that flag.  `out` (4*n words,
 * may be null) receives what has been recorded so far; the words are cleared whenever `on` is set.  tools/chain_hops.py. */
F2V_API int f2v_test_stamps(f2v_handle h, int on, unsigned long long *out);

/* Host-only (no device): builds the wide form's launch plans of one epoch for this graph, batch and tunables (`names` / `values`:
 * hub_chunk, hub_fanin, class_cut, wide_phases, wide_rounds, wide_span, wide_finish, wide_order, wide_rows, wide_min_width; plan_threads = T
 * builds them on T host threads as f2v_train does for large graphs) and
 * checks what the kernel relies on: every row finished once, every neighbour in one piece, rounds / phases / slots / passes
 * well-formed, a job adds consecutive pieces of one row, and every wait (an imported group sum, a tree node's inputs) points at a
 * workgroup with a SMALLER index.  stats_out[7]: workgroups, helpers, finishers, packed, node workgroups, partial-sum slots, a checksum of
 * the resident plan arrays. */
F2V_API int f2v_test_wide_plan_check(const uint32_t *rowptr, const uint32_t *colids, uint32_t n, uint64_t nnz, uint32_t dim, uint32_t batch, int walk,
                             const char *const *names, const int64_t *values, uint32_t n_params, uint64_t *stats_out);

/* Per-XCD timing of the launches that follow (the sub-wave step kernel, one launch per minibatch): on = 1 clears 32 words -- per XCD k: [k] the latest end
 * of a workgroup, [8 + k] the earliest start, [16 + k] the sum of its workgroups' durations, [24 + k] its workgroups (100-MHz device wall clock) --, `out`
 * (may be null) receives what has been recorded since the last call; on = 0 frees them.  tools/xcd_balance_probe.py. */
F2V_API int f2v_test_xcd_times(f2v_handle h, int on, unsigned long long *out);

/* The memory side of one real launch, alone (D = 128): the launch plan of minibatch [row_lo, row_hi) replayed by a kernel that only gathers --
 * the plan's items in the plan's order, lane groups in lockstep, 4 rows in flight, nothing computed, no negative samples, no combine trees;
 * mode bit 0: every item also reads its own row, bit 1: every whole-row item stores a row into the second matrix (the embeddings are garbage
 * afterwards).  Best of `reps` launches in microseconds.  tools/plan_gather_probe.py. */
F2V_API int f2v_test_plan_gather(f2v_handle h, uint32_t row_lo, uint32_t row_hi, uint32_t mode, uint32_t reps, double *us_out);

#ifdef __cplusplus
}
#endif
#endif /* F2V_TEST_H_ */

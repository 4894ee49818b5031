/*
 * f2v.h -- C ABI of libf2v, the MI355X (gfx950) Force2Vec embedding engine.
 *
 * Drop-in boundary for the ONE hot path of HipGraph/Force2Vec: the per-minibatch
 * attraction/repulsion force kernels + SGD row update of tForce2Vec / sForce2Vec /
 * rForce2Vec (CLI options 5-7, and 8-11 as their load-balanced equivalents).  The
 * reference has no FFI of its own; each entry point below names the reference interface
 * it replaces (file:line under the reference tree).  Plain pointers and sizes only.
 *
 * Conventions: every function returns 0 on success and a negative F2V_E* code on failure
 * (f2v_last_error() then holds a message for the calling thread).  The caller owns every
 * host array it passes in (inputs are copied to HBM); the library owns all device state.
 * One handle per host thread; no global mutable state besides the per-thread error text.
 * There is NO CPU fallback: without a usable HIP device f2v_create fails with F2V_ENODEV.
 */
#ifndef F2V_H_
#define F2V_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The ABI: libf2v.so is built with -fvisibility=hidden -Wl,-Bsymbolic and exports exactly the functions marked F2V_API
 * (tests/test_host_boundary.py checks `nm -D`).  Nothing else -- no kernel launch stub, no C++ helper -- can be interposed by,
 * or interpose on, another library of the process (two builds of libf2v in one process are two independent engines). */
#ifndef F2V_API
#define F2V_API __attribute__((visibility("default")))
#endif

#define F2V_OK 0
#define F2V_EINVAL (-1)  /* bad argument */
#define F2V_ENODEV (-2)  /* no usable HIP device / HIP runtime error */
#define F2V_ENOMEM (-3)
#define F2V_EIO (-4)
#define F2V_ESTATE (-5)  /* call order violated (e.g. training before init) */

#define F2V_INIT_SYMMETRIC 0 /* randInitF: U[-1,1)  sample/algorithms.cpp:47-53 (options 5,8,11) */
#define F2V_INIT_UNIT 1      /* randInit : U[0,1)   sample/algorithms.cpp:38-45 (options 6,7,9,10) */

typedef struct f2v_ctx *f2v_handle;

F2V_API const char *f2v_last_error(void);
F2V_API const char *f2v_version(void);

/* ---- engine life cycle -----------------------------------------------------------------
 * Replaces `algorithms::algorithms(CSR&, input, outputdir, dim, gamma, batch)`
 * (sample/algorithms.h:60-70): copies the CSR (rowptr u32[n+1], colids u32[nnz], ascending
 * inside each row, duplicates kept -- sample/CSR.h:89-96) to HBM and allocates the N x D
 * fp32 embedding matrix there.  `device` is the HIP device ordinal. */
F2V_API int f2v_create(const uint32_t *rowptr, const uint32_t *colids, uint32_t n, uint64_t nnz, uint32_t dim, int device,
               f2v_handle *out);
F2V_API int f2v_destroy(f2v_handle h);

/* srand(seed) of Test/Force2Vec.cpp:126; the handle carries the libc rand() stream. */
F2V_API int f2v_srand(f2v_handle h, uint32_t seed);
/* randInitF / randInit (sample/algorithms.cpp:38-53): N*D rand() draws, uploaded to HBM. */
F2V_API int f2v_init_embeddings(f2v_handle h, int kind);
/* Direct access to the embedding matrix `nCoordinates` (sample/algorithms.h:55), host N x D row-major. */
F2V_API int f2v_set_embeddings(f2v_handle h, const float *x);
F2V_API int f2v_get_embeddings(f2v_handle h, float *x_out);

/* Tunables.  "hub_chunk": neighbours per work item before a row is split (0 = never split: the
 * reference's summation order for every row; unset = chosen by f2v_train from the rows one launch covers
 * -- the batch, or a rank's slice of it in f2v_train_sharded -- or by setting "hub_chunk_for_batch" = B;
 * the chunk is part of the summation order, so pin it where bits must not depend on the number of GPUs);
 * "hub_fanin": fan-in of the tree that adds a split row's partial sums (0 = one sequential pass);
 * "merge_finalize" (default 1): the trees' nodes run in the step kernel's own grid (one launch per
 * minibatch; every in-grid wait is bounded by "tree_timeout_ms", default 5000, those of chained launches by
 * "chain_timeout_ms", default 200 -- a wait that gives up stores nothing, the launch drains, the handle falls back to 0, and
 * f2v_train repeats the call from the snapshot it took at its start; where it has none -- "recover" = 0, no room for one more
 * matrix, f2v_train_sharded, f2v_minibatch_step -- the call fails with F2V_ESTATE within an epoch or two and the embeddings
 * must be set again), 0 = one launch per tree level;
 * SINGLE TENANT: launches with in-grid waits ("merge_finalize", "chain_batches") count on this process having the GPU to
 * itself -- a second process (or a second handle of this one training at the same time) whose waiting workgroups fill the card
 * can keep the workgroups they wait for from starting; the bounded waits and "recover" turn that into a slower, correct
 * run ("recoveries" counts them), never into a hang or a wrong result.  Set both to 0 where the card is shared on purpose; f2v_create selects 0 by itself when its dispatch probe does not
 * find 8 XCDs taking workgroups round robin ("xcc_count", "xcc_round_robin" answer what it saw); "chain_batches" (default 1): f2v_train runs minibatches of up to "chain_max_batch" (4096) rows in groups of
 * "chain_rows" (65536) rows per launch, ordered by row-level data dependencies inside the launch instead of launch boundaries
 * (same results; batch 256 on RMAT-20: 0.60 -> 2.0 G edges/s); "chain_wide" (default 1): minibatches of up to "wide_max_batch" (2048) rows run
 * in the WIDE form of such launches -- the pieces of a split row meet in LDS inside one workgroup (finisher + helpers) instead of travelling
 * through HBM and combine-tree nodes, "wide_rows" (262144) rows per launch; same pieces, same fan-in groups, same results (batch 256: 2.0 -> 4.3 G
 * edges/s); needs 2 <= "hub_fanin" <= 32; "wide_phases", "wide_rounds", "wide_span", "wide_finish", "wide_order" (where a minibatch's whole-row
 * workgroups go: measured neutral), "wide_min_width" (0 = automatic) shape its
 * workgroup programs; "wide_epochs" (0 = automatic: 32 on graphs of up to 2 M nonzeros that one launch covers, else 1): that many EPOCHS of options 5 / 6 run in
 * one launch -- a ring of matrices, epoch e reads matrix e and writes matrix e + 1 behind its own row flags; same results, and what was a
 * launch boundary per epoch becomes one more hop of the dependency chain ("last_wide_epochs" answers what the last f2v_train did);
 * "wide_samples_early" (-1 = automatic: small graphs): the sample rows a launch itself writes are awaited before
 * the first neighbour waits instead of after them (placement in time only); "last_train_form" answers how the last f2v_train launched (0 one launch per minibatch, 1 chained, 2 wide, 3 hipGraph replay: set where the launches are made), "last_wide_width"
 * the sub-wave layout a wide run used, "last_wide_early" whether it ran the kernel's EARLY form; "class_cut" (default 1): a split row's pieces also end where its ascending neighbour ids cross
 * from one eighth of the id range into the next (part of the summation order, restated by the oracle; it is what makes
 * "piece_affinity" pure); "piece_affinity" (default 1): a split row's pieces run on the XCD that owns
 * the id range of their neighbours, so that each of the eight L2s caches its own eighth of the matrix (placement only:
 * results do not change; ranks of a push exchange that share one GPU switch it off by themselves); "quarter_wave": 0 selects the one-item-per-wavefront
 * kernel for every D; "waves_per_block"; "rows_in_flight" (0 = the kernels' default | 4 | 8); "use_graph" = 1 makes f2v_train replay a
 * captured hipGraph per epoch instead of launching eagerly (same results; measured no faster); "count_compulsory" = 1
 * makes new launch plans count their compulsory bytes (f2v_stats.compulsory_bytes).
 * Launch plans: the first f2v_train of a batch size builds the epoch's plans on the host (the wide form's on the host's threads:
 * F2V_IO_THREADS bounds them; RMAT-24 at batch 384: 65 plans, 0.8 s) and keeps them resident on the host and in HBM -- "plan_resident_bytes"
 * answers how much (RMAT-20 at batch 256: 184 MB; RMAT-24 at batch 384: 2.9 GB); the cache holds up to three epochs' worth of items and is
 * dropped and rebuilt on demand beyond that (a run that alternates many batch sizes).  "wide_single" = 1 (measurement only) lets the wide form run
 * launches of one minibatch.
 * Sharded runs: "push_fused" (default 1: the step kernels push their rows themselves, 0: a kernel behind
 * them does), "push_timeout_ms"; "replicate_small" (default 1): f2v_train_sharded at a batch size that f2v_train runs chained (up to
 * "chain_max_batch" rows: the reference's default 384 is one) runs the whole call on EVERY rank and exchanges nothing -- such an epoch is one
 * row-to-row dependency chain that hops over xGMI can only lengthen (sharded: 41 ms per epoch at batch 384 on two ranks; one GPU: 6.7 ms) --
 * where no two attached ranks share a GPU and every rank's GPU can chain (both learnt by f2v_push_attach, so every rank decides alike);
 * 2: also on a shared GPU; 0: never.  Bits, rand() state and the matrix every rank holds afterwards are f2v_train's.
 * f2v_get_param("last_train_replicated") tells.
 * "fast_rng" = 1 selects the NON-PARITY fast mode: initial embeddings and the option-7
 * walks are generated on the device by a counter-based RNG (same distributions, different numbers than
 * the reference's libc rand() stream); negative-sample ids still come from the handle's rand() stream.
 * f2v_get_param also answers "dim", "n", "nnz", "hub_chunk_auto" (1 while the chunk is still chosen per call), and for
 * a handle attached to a push exchange "push_rank", "push_world", "shared_card" (1: a peer runs on this very GPU). */
F2V_API int f2v_set_param(f2v_handle h, const char *name, int64_t value);
F2V_API int f2v_get_param(f2v_handle h, const char *name, int64_t *value_out);

/* ---- training --------------------------------------------------------------------------
 * Replaces vector<float> algorithms::AlgoForce2VecNS / NSBS / NSRW / NSRWBS / NSRWEFF and
 * their AVX512 twins (sample/algorithms.h:86-102; bodies sample/algorithms.cpp:544-1203,
 * 1230-4051): `iters` epochs of minibatch SGD over all N vertices in batches of `batch`,
 * drawing negative samples (and, for option 7, walks) from the handle's rand() stream in
 * the reference's order.  option: 5|6|7 (8,11 -> 5 ; 9 -> 6 with its own negative-sample range ; 10 -> 7 without the division of
 * the attraction by deg + 1, as AlgoForce2VecNSRWEFF_SREAL_D128/D64_AVXZ have it: `degi = 1.0`, sample/algorithms.cpp:2155, :3793).  bs_mode: the
 * CLI's "-bs" (1 = ns*batch samples per minibatch, row i uses samples [i, i+ns)).
 * seconds_out (may be NULL) receives the device time of the epoch loop alone (HIP events);
 * the embeddings stay in HBM (fetch with f2v_get_embeddings). */
F2V_API int f2v_train(f2v_handle h, int option, uint32_t iters, uint32_t batch, uint32_t ns, float lr, int bs_mode,
              double *seconds_out);
/* (While "recover" is on -- the default -- and the handle uses in-grid waits, f2v_train keeps a copy of the matrix and of the
 * rand() state as they were when the call began: one more N x D matrix of HBM, one device-to-device copy per call.  A call
 * that loses a launch is run again from there with one launch per minibatch and per tree level: same bits, F2V_OK,
 * f2v_last_error() says what happened, "recoveries" counts.) */

/* One minibatch: the kgen row-kernel boundary Calc_<pre>frc_<tdist|sigmoid>_DIM<D>_VL<V>
 * (sample/kgen/genDimFrc.base:36-57) lifted to a batch, and the unit the multi-GPU driver
 * shards.  Computes the new embeddings of rows [row_lo,row_hi) of minibatch
 * [batch_lo,batch_hi) from the pre-batch matrix into the epoch's second matrix ("staged": later
 * minibatches read them there, f2v_flush / the end of the epoch makes them the matrix; other
 * ranks' rows of the same minibatch are merged first with f2v_stage_write / an all-gather).  sample_ids: host array of the minibatch's
 * negative-sample vertex ids (ns of them, or (batch_hi-batch_lo)+ns-1 in bs_mode).
 * Option 7 uses the walks set by f2v_set_walks. */
F2V_API int f2v_minibatch_step(f2v_handle h, int option, uint32_t batch_lo, uint32_t batch_hi, uint32_t row_lo,
                       uint32_t row_hi, const uint32_t *sample_ids, uint32_t n_sample_ids, uint32_t ns, float lr,
                       int bs_mode);
/* The same step with the sample ids already in HBM: f2v_upload_sample_ids copies a host array (e.g. one
 * epoch's ids, drawn up-front: they do not depend on the embeddings) once, f2v_minibatch_step_at names the
 * minibatch's ids by their offset in it.  No host-device synchronisation per step: the multi-GPU driver
 * enqueues step and exchange back to back. */
F2V_API int f2v_upload_sample_ids(f2v_handle h, const uint32_t *ids, uint64_t count);
F2V_API int f2v_minibatch_step_at(f2v_handle h, int option, uint32_t batch_lo, uint32_t batch_hi, uint32_t row_lo,
                          uint32_t row_hi, uint64_t ids_offset, uint32_t ns, float lr, int bs_mode);
/* Commit the staged minibatch into the matrix (K5, sample/algorithms.cpp:629-639 / 913-921). */
F2V_API int f2v_flush(f2v_handle h);
/* Option 7 walk samples of the current epoch, uint32[5*n] (sample/algorithms.cpp:1097-1118). */
F2V_API int f2v_set_walks(f2v_handle h, const uint32_t *walks);
/* Draw this epoch's walks from the handle's rand() stream exactly as the reference does. */
F2V_API int f2v_generate_walks(f2v_handle h, uint32_t *walks_out /* may be NULL */);
/* randIndex(max,min) of sample/algorithms.cpp:55-58 on the handle's stream. */
F2V_API int f2v_rand_index(f2v_handle h, uint32_t max_num, uint32_t min_num, uint32_t *out);
/* `count` consecutive randIndex(max,min) draws; the first `keep` (<= count) are stored in out.
 * (One minibatch's sample loop, sample/algorithms.cpp:577-586; -bs 1 draws ns*BATCH, :686.) */
F2V_API int f2v_rand_indices(f2v_handle h, uint32_t max_num, uint32_t min_num, uint64_t count, uint64_t keep, uint32_t *out);

/* Multi-GPU exchange: device address of the staged rows of the last stepped minibatch (row r of the
 * batch at float offset (r-batch_lo)*dim; it points into the second matrix), the number of rows that may
 * be written from there (the matrix has slack behind row N for a padded all-gather), and a host
 * read/write of a row range of it (gloo / test path). */
F2V_API int f2v_stage_device_ptr(f2v_handle h, uint64_t *devptr_out, uint32_t *capacity_rows_out);
F2V_API int f2v_stage_read(f2v_handle h, uint32_t row_lo, uint32_t row_hi, float *out);
F2V_API int f2v_stage_write(f2v_handle h, uint32_t row_lo, uint32_t row_hi, const float *in);
F2V_API int f2v_stage_reserve(f2v_handle h, uint32_t rows);
/* Arbitrary rows of the matrix the staged rows live in (the epoch's second matrix), by vertex id: the
 * per-destination exchange ("send a row only to the ranks that read it") packs and unpacks with these. */
F2V_API int f2v_rows_read(f2v_handle h, const uint32_t *ids, uint32_t count, float *out);
F2V_API int f2v_rows_write(f2v_handle h, const uint32_t *ids, uint32_t count, const float *in);
/* Device address of the embedding matrix and the HIP stream (as integers) for zero-copy wrapping. */
F2V_API int f2v_embeddings_device_ptr(f2v_handle h, uint64_t *devptr_out);
F2V_API int f2v_stream(f2v_handle h, uint64_t *stream_out);
F2V_API int f2v_synchronize(f2v_handle h);

/* ---- multi-GPU: the push exchange over xGMI -----------------------------------------------
 * One process per GPU, the graph and both matrices replicated, rank r computes the r-th contiguous slice
 * of every minibatch (north_star's "1-D vertex partition ... each minibatch"; the reference has no
 * multi-device path).  What crosses GPUs is the NEW ROWS of a minibatch (forces only ever update the source
 * row): after its step kernel a rank's push kernel stores each new row straight into the second matrix of
 * every peer that READS that row -- the peers' matrices are mapped through HIP IPC, the stores travel
 * over the direct xGMI link to that peer -- and a device-side flag barrier (one small kernel, no host
 * round trip, no collective) separates minibatches.  Who reads a row is static for options 5/6: the ranks
 * owning a CSR neighbour of it, plus everyone for a vertex some minibatch samples (f2v_push_masks).
 * Results are bit-identical to the single-GPU f2v_train for any world size.
 *
 *   f2v_push_export   this rank's F2V_PUSH_EXPORT_BYTES bytes: IPC handles of both matrices and the flags, n, dim;
 *                     the host gathers them from all ranks with whatever it has (torch.distributed, MPI, a file)
 *   f2v_push_attach   map the peers (all_handles = world exports, in rank order); world <= F2V_PUSH_MAX_RANKS
 *   f2v_push_selftest every rank writes a pattern into every peer's slack rows, barrier, verifies what it
 *                     received: F2V_OK only if IPC mapping, remote stores and the flag barrier all work
 *   f2v_train_sharded f2v_train over the attached ranks (every rank calls it with the same arguments after
 *                     the same f2v_srand / f2v_init_embeddings); on return every replica is complete
 *   f2v_push_detach   unmap the peers (also done by f2v_destroy)
 * "push_timeout_ms" (f2v_set_param, default 20000) bounds every wait of the flag barrier: a missing peer
 * makes the calls fail with F2V_ESTATE instead of hanging the GPU.
 * Matrices of 2 GiB and more cannot be mapped through HIP IPC (hipIpcOpenMemHandle does not return): such
 * engines exchange through a mapped landing buffer of one minibatch (two halves of at most 512 MiB) that a
 * small kernel unpacks behind the barrier -- automatically, or for any size with "push_landing" = 1 (set
 * before f2v_push_export, on every rank alike).  A minibatch must then fit one half.
 * (Fault injection for the protocol tests lives in the self-test build only: include/f2v_test.h.) */
#define F2V_PUSH_MAX_RANKS 8
#define F2V_PUSH_EXPORT_BYTES 256
F2V_API int f2v_push_export(f2v_handle h, void *handles_out);
F2V_API int f2v_push_attach(f2v_handle h, uint32_t rank, uint32_t world, const void *all_handles);
F2V_API int f2v_push_selftest(f2v_handle h);
F2V_API int f2v_push_detach(f2v_handle h);
F2V_API int f2v_train_sharded(f2v_handle h, int option, uint32_t iters, uint32_t batch, uint32_t ns, float lr, int bs_mode,
                      double *seconds_out);
/* Host-only: the slices f2v_train_sharded cuts minibatch [lo,hi) into: bounds_out[0..world], slice r = rows
 * [bounds_out[r], bounds_out[r+1]), contiguous and balanced by work (weight of a row = its degree + 4), not by
 * row count.  (Option 7 uses equal row counts: its rows all have five pairs.) */
F2V_API int f2v_shard_bounds(const uint32_t *rowptr, uint32_t lo, uint32_t hi, uint32_t world, uint32_t *bounds_out);
/* Host-only: masks_out[v] = bit r set when rank r READS row v without owning it -- v is a CSR neighbour of a
 * row in one of r's slices (f2v_shard_bounds of every minibatch), or one of `sample_ids` (read by every row of a
 * minibatch, hence by every rank). */
F2V_API int f2v_push_masks(const uint32_t *rowptr, const uint32_t *colids, uint32_t n, uint32_t batch, uint32_t world,
                   const uint32_t *sample_ids, uint64_t n_ids, uint32_t *masks_out);
/* Rows pushed to peers / rows a full all-gather would have sent (per peer copies), since the last f2v_train_sharded began. */
F2V_API int f2v_push_stats(f2v_handle h, uint64_t *rows_pushed_out, uint64_t *rows_allgather_out);

/* Statistics of the last f2v_train: launches of the step kernel, rows and nonzeros they
 * processed, algorithmic bytes (SURVEY 8d formula), device seconds. */
typedef struct {
    uint64_t step_launches;
    uint64_t rows;
    uint64_t nnz;
    uint64_t algorithmic_bytes;
    double device_seconds;
    uint64_t hub_rows;
    uint64_t hub_chunks;
    /* with "count_compulsory" = 1: sum over the launches of (distinct embedding rows read + rows written) * 4D + 4 bytes per
     * neighbour id + 16 per work item -- the bytes a launch must move even if every re-read inside it hit a cache: the
     * numerator of a roofline fraction that cannot exceed 1 (the SURVEY 8d figure above charges every neighbour row to
     * HBM and does, on power-law graphs whose hub rows are cache hits). */
    uint64_t compulsory_bytes;
    /* (library 0.5) f2v_train keeps a snapshot of the matrix while "recover" is on and the handle's launches may hold in-grid waits:
     * one device-to-device copy of N x D floats per CALL, on the stream in front of the epoch loop and NOT part of device_seconds /
     * seconds_out -- its own device time is reported here (a caller that trains one epoch per call pays it every epoch: RMAT-20
     * ~0.2 ms, RMAT-24 ~3 ms; "recover" = 0 drops the copy and frees the matrix). */
    double snapshot_seconds;
    uint64_t recoveries;      /* give-ups f2v_train has recovered from since the handle was created ("recoveries") */
    uint32_t recovered;       /* 1: the LAST f2v_train lost a launch and ran again without in-grid waits: its device_seconds are those of
                               * the slow launch forms -- a benchmark must not quote them as the fast path's */
    uint32_t merge_finalize;  /* the handle's "merge_finalize" now: 0 after a give-up = one launch per minibatch and tree level until the
                               * in-grid waits come back (after 1, 2, 4 ... 64 healthy f2v_train calls) */
} f2v_stats;
F2V_API int f2v_get_stats(f2v_handle h, f2v_stats *out);
/* With "epoch_marks" = k > 0 the next f2v_train records a HIP event on its stream after every k-th epoch (at most 4096 of
 * them); afterwards f2v_train_marks copies the device time from the start of the epoch loop to each mark into `seconds_out`
 * (up to `cap` values; `count_out` receives how many there are): the rate over a long run second by second, without a host
 * synchronisation inside the loop. */
F2V_API int f2v_train_marks(f2v_handle h, double *seconds_out, uint32_t cap, uint32_t *count_out);

/* ---- host-side I/O of the drop-in boundary (no device needed) ----------------------------
 * f2v_read_mtx replaces SetInputMatricesAsCSR (sample/commonutility.h:44-54 -> ReadASCII
 * sample/IO.h:59-156, CSC sample/CSC.h:146-188, CSR sample/CSR.h:154-186): MatrixMarket
 * coordinate text; "symmetric" mirrors off-diagonal entries and drops self-loops;
 * duplicates kept; colids ascending per row.  Arrays are malloc'ed; free with f2v_free. */
F2V_API int f2v_read_mtx(const char *path, uint32_t *n_out, uint64_t *nnz_out, uint32_t **rowptr_out, uint32_t **colids_out);
F2V_API void f2v_free(void *p);
/* Replaces algorithms::writeToFile (sample/algorithms.h:118-136): "<N> <D>\n", then
 * "<i+1> v0 v1 ... \n" with 6 significant digits (%g) and a trailing space. */
F2V_API int f2v_write_embd(const char *path, const float *x, uint32_t n, uint32_t dim);
/* Output file name rule of writeToFile + the per-option suffixes (sample/algorithms.cpp:650,
 * 752, 930, 1059, 1201, 1635, 2047, 2409, 2860): outdir + basename(input) + suffix + ".embd". */
F2V_API int f2v_output_name(const char *input, const char *outdir, int option, int bs_mode, uint32_t batch, uint32_t dim,
                    uint32_t iters, uint32_t ns, char *out, size_t out_len);

/* SURVEY 8f "next" rows on the data-format side of the path.
 * Binary CSR cache (text parsing of 10^8-edge files dominates wall time otherwise): little-endian
 * "F2VCSR1\0", u32 n, u32 reserved, u64 nnz, u32 rowptr[n+1], u32 colids[nnz] -- exactly the arrays
 * f2v_read_mtx returns, so a cached graph trains bit-identically. */
F2V_API int f2v_write_csr_bin(const char *path, const uint32_t *rowptr, const uint32_t *colids, uint32_t n, uint64_t nnz);
F2V_API int f2v_read_csr_bin(const char *path, uint32_t *n_out, uint64_t *nnz_out, uint32_t **rowptr_out, uint32_t **colids_out);
/* Raw fp32 N x D embedding file, the format the reference's scorers read with readBinEmbeddings
 * (performancescores/runnodeclassclust.py:81-100); text .embd of 16 M x 128 values is ~19 GB. */
F2V_API int f2v_write_embd_bin(const char *path, const float *x, uint32_t n, uint32_t dim);
/* The readers of both embedding formats (what performancescores/runnodeclassclust.py:57-100 reads): warm starts
 * (f2v_set_embeddings; `Force2Vec -init <file>`) and scoring without the text round trip.  f2v_read_embd allocates *x_out (N x D
 * floats, release with f2v_free); rows may come in any order, ids are 1-based. */
F2V_API int f2v_read_embd(const char *path, uint32_t *n_out, uint32_t *dim_out, float **x_out);
F2V_API int f2v_read_embd_bin(const char *path, uint32_t n, uint32_t dim, float *x_out);

/* Stand-alone libc rand() stream (glibc TYPE_3), for hosts that pre-draw sample ids. */
typedef struct f2v_rng f2v_rng;
F2V_API f2v_rng *f2v_rng_create(uint32_t seed);
F2V_API void f2v_rng_destroy(f2v_rng *g);
F2V_API int f2v_rng_next(f2v_rng *g);
/* Skip k draws in O(log k) (the generator is linear: a 31x31 matrix over Z/2^32 per jump). */
F2V_API void f2v_rng_jump(f2v_rng *g, uint64_t k);
/* `count` values as randInitF (kind 0) / randInit (kind 1) would store them, drawn from ONE serial stream but filled
 * in parallel from jump-ahead states; the stream ends where `count` serial draws would leave it. */
F2V_API int f2v_rng_fill(f2v_rng *g, float *out, uint64_t count, int kind);

/* One epoch's option-7 walk samples uint32[5*n] from stream g, exactly the reference's draws in the reference's order
 * (sample/algorithms.cpp:1097-1118) -- what f2v_generate_walks does with the handle's own stream, without a device. */
F2V_API int f2v_rng_walks(f2v_rng *g, const uint32_t *rowptr, const uint32_t *colids, uint32_t n, uint64_t nnz, uint32_t *walks_out);

/* The 2048-entry sigmoid table of init_SM_TABLE (sample/algorithms.cpp:757-764) as the source defines it. */
F2V_API int f2v_sm_table(float *table_out /* 2048 */);

/* ---- diagnostics (bench.py, tools/ipc_preflight.py, bin/Force2Vec -gpus) ------------------ */
/* Rehearsal of the push exchange's needs (IPC mapping of `bytes` of device memory and of fine-grained flags between
 * `world` processes that meet through files in `dir`, remote stores from a kernel), for a throw-away process to run
 * before the real engines exist: a mapping call that never returns or a faulting remote store then costs only it. */
F2V_API int f2v_diag_ipc_preflight(int device, uint32_t rank, uint32_t world, const char *dir, uint64_t bytes, double timeout_s);
/* On-box streaming-copy ceiling: read + written bytes per second (GB/s) of a 16-byte-per-lane copy of `bytes`, best of `reps`. */
F2V_API int f2v_diag_stream_copy(int device, uint64_t bytes, uint32_t reps, double *gbps_out);
/* On-box random-row gather ceiling (GB/s, ids included): every 512-byte row of a `table_bytes` table fetched once per pass
 * in random order with the step kernel's access pattern.  32 MiB: the Infinity Cache / L2 rate; 4 GiB: the HBM rate. */
F2V_API int f2v_diag_gather_rate(int device, uint64_t table_bytes, uint32_t reps, double *gbps_out);

#ifdef __cplusplus
}
#endif
#endif /* F2V_H_ */

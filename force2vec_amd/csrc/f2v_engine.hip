// f2v_engine.hip -- HBM-resident Force2Vec engine behind the C ABI of include/f2v.h.
//
// Device layout (all in the HBM of one MI355X):
//   X[2]       fp32 [(N+pad) x D] row-major (rows 4*D bytes apart: 512 B at D = 128): the current matrix and
//              the one this epoch's new rows are written to; they swap when an epoch completes
//   rowptr     u32  [N+1], colids u32 [nnz]           (the reference's CSR, sample/CSR.h:89-96)
//   partials   fp32 [hub chunks x D]  partial force sums of split hub rows
//   sample ids u32, walks u32 [5N], sigmoid table fp32 [2048]
// The host side only sequences launches; every arithmetic step runs in f2v_kernels.hip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <condition_variable>
#include <map>
#include <mutex>
#include <memory>
#include <string>
#include <atomic>
#include <thread>
#include <tuple>
#include <type_traits>
#include <vector>

#include "f2v.h"
#ifdef F2V_TEST_HOOKS
#include "f2v_test.h"
#endif
#include "f2v_internal.h"
#include "f2v_kernels.hip.h"

using namespace f2v;

#define HIPC(expr)                                                                                     \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return fail(e__ == hipErrorOutOfMemory ? F2V_ENOMEM : F2V_ENODEV, "%s: %s (%s:%d)", #expr,  \
                        hipGetErrorString(e__), __FILE__, __LINE__);                                   \
    } while (0)

constexpr size_t kIpcMaxBytes = (size_t)1 << 31;  // allocations of this size and more cannot be opened through HIP IPC (f2v_push_export)
constexpr uint32_t kPadRows = 4096;  // slack behind row N for the padded in-place all-gather of the last minibatch
constexpr int kMaxFinLevels = 32;  // fan-in >= 2: 2^32 chunks
struct Plan {
    size_t item_off = 0;
    uint32_t n_items = 0, n_hubs = 0, n_chunks = 0, n_slots = 0;
    int n_levels = 0;  // levels of the hub combine trees
    size_t fin_off[kMaxFinLevels] = {};
    uint32_t fin_cnt[kMaxFinLevels] = {};
    uint64_t nnz = 0;
    uint64_t compulsory = 0;  // bytes this launch must move even with perfect caching inside the launch ("count_compulsory")
};

// The resident plan arrays reach gigabytes (RMAT-24 at batch 384: 2.9 GB) and are filled by memcpy from the parts the host's threads
// built: resize() must not zero them first (one thread, page by page) -- an allocator whose construct() of no arguments default-initialises.
template <class T>
struct RawInit : std::allocator<T> {
    template <class U> struct rebind { using other = RawInit<U>; };
    using std::allocator<T>::allocator;
    template <class U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
template <class T> using PlanVec = std::vector<T, RawInit<T>>;

struct SeenScratch {  // "count_compulsory": has this row been counted in this minibatch?  One stamp per vertex (a plan-building thread owns one)
    std::vector<uint32_t> seen;
    uint32_t stamp = 0;
};

// One launch of qstep_chain_kernel: up to 64 consecutive minibatches (f2v_kernels.hip.h, "chained minibatches")
struct ChainPlan {
    size_t item_off = 0, fin_off = 0, wg_off = 0;
    uint32_t n_wgs = 0, n_batches = 0, n_slots = 0, n_fin = 0;
    uint32_t first_batch = 0;  // global minibatch index of its first minibatch
    uint32_t lo = 0, hi = 0, last_lo = 0;   // rows covered; first row of the last minibatch
    uint64_t nnz = 0, compulsory = 0;
    uint32_t n_hubs = 0, n_chunks = 0;
};

// One launch of qwide_chain_kernel: the same minibatches as workgroup PROGRAMS (f2v_kernels.hip.h, "wide form")
struct WidePlan {
    size_t item_off = 0, fin_off = 0, wg_off = 0, job_off = 0;
    uint32_t n_wgs = 0, n_batches = 0, n_slots = 0;
    uint32_t first_batch = 0;
    uint32_t lo = 0, hi = 0, last_lo = 0;
    uint64_t nnz = 0, compulsory = 0;
    uint32_t n_hubs = 0, n_chunks = 0;
    uint32_t n_helpers = 0, n_finishers = 0, n_packed = 0, n_node_wgs = 0;  // workgroups by role
    uint32_t width = 0;  // the sub-wave layout its programs are laid out for (wide_width)
};

// what a rank hands its peers (f2v_push_export): F2V_PUSH_EXPORT_BYTES bytes
struct PushExport {
    hipIpcMemHandle_t x[2], flags;  // landing-buffer mode: x[0] is the landing buffer, x[1] unused
    uint32_t magic, n, D, cur;
    uint32_t landing, landing_cap;
    char bus_id[16];  // PCI bus id of the exporting rank's GPU: ranks that share a card run without "piece_affinity"
    uint32_t caps;    // bit 0: this rank's GPU starts workgroups XCD-round-robin in index order (what chained launches count on)
    char reserved[F2V_PUSH_EXPORT_BYTES - 3 * sizeof(hipIpcMemHandle_t) - 28 - 16];
};
static_assert(sizeof(PushExport) == F2V_PUSH_EXPORT_BYTES, "export blob layout");
static_assert(kMaxRanks == F2V_PUSH_MAX_RANKS, "rank limit");
constexpr uint32_t kPushMagic = 0x46325650u;  // "F2VP"

struct f2v_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t n = 0, D = 0;
    uint64_t nnz = 0;
    int vec = 1;
    bool exact = false;
    std::vector<uint32_t> rowptr, colids;  // host copies: hub planning, walk generation, statistics
    uint32_t *d_rowptr = nullptr, *d_colids = nullptr, *d_walks = nullptr, *d_ids = nullptr;
    uint32_t *d_walks_alt = nullptr;  // second walk buffer: f2v_train (option 7) alternates between the two, epoch by epoch
    size_t ids_cap = 0, ids_valid = 0;  // ids_valid: prefix uploaded by f2v_upload_sample_ids
    float *d_X[2] = {nullptr, nullptr}, *d_partials = nullptr, *d_table = nullptr;
    uint32_t *d_ready = nullptr, *d_kerr = nullptr;  // combine-tree flags (one per partial slot), kernel error word
    uint32_t launch_seq = 0;
    uint32_t xcc_count = 0;      // XCDs seen by the dispatch probe of f2v_create
    bool xcc_round_robin = false;  // ... and workgroup b ran on the XCD of workgroup b mod 8
    int64_t tree_timeout_ms = 5000;
    int64_t chain_timeout_ms = 200;  // chained launches: a healthy one lasts a millisecond or two, so a lost one is noticed in milliseconds
    // f2v_train survives a give-up ("recover"): the matrix and the rand() state as they were when the call began
    bool recover = true;
    float *d_snap = nullptr;
    hipEvent_t ev_snap[2] = {nullptr, nullptr};  // around the snapshot copy: f2v_stats.snapshot_seconds
    // in-grid waits that were switched off by a RECOVERED give-up come back by themselves: after `waits_backoff` healthy f2v_train
    // calls (1, then 2, 4 ... 64: a card that keeps losing launches is asked less and less often); a give-up that was not recovered
    // from, the dispatch probe's verdict and the caller's own "merge_finalize" = 0 stay
    bool waits_suspended = false;
    uint32_t waits_backoff = 1, calls_since_give_up = 0;
    uint32_t recoveries = 0;
    bool unit_degi = false;  // the option being run is 10 (StepArgs::unit_degi): set by every entry point that takes an option
    uint32_t mark_every = 0;       // "epoch_marks"
    std::vector<double> marks;     // f2v_train_marks
    uint32_t last_wide_width = 0;  // the layout width of the last wide-form f2v_train ("last_wide_width")
    bool last_wide_early = false;  // ... and whether it ran the kernel's EARLY form ("last_wide_early")
    int last_train_form = 0;  // how the last f2v_train launched: 0 one launch per minibatch, 1 chained, 2 chained in the wide form ("last_train_form")
    bool plan_overflow = false;  // a launch plan needed more than 2^28 partial-sum slots (kItemSlotMask)
    uint32_t *h_kerr = nullptr;  // pinned: the kernel error words as of the last completed epoch-end copy (train_impl)
#ifdef F2V_TEST_HOOKS
    uint32_t test_withhold_slot = kNoSlot, test_withhold_row = kNoSlot;
    uint32_t test_chain_mode = 0;  // f2v_test_chain_nowait's argument as given (bit 0 = the old on/off)
    bool test_chain_nowait = false;  // timing experiment: chained launches without their row waits (results are then wrong)
    unsigned long long *d_xcd = nullptr;     // f2v_test_xcd_times (StepArgs::xcd_times)
    unsigned long long *d_stamps = nullptr;  // f2v_test_stamps: 4 wall-clock words per row (StepArgs::stamps)
#endif
    bool merge_fin = true, capturing = false;  // all combine-tree levels in one launch (not while a hipGraph is captured)
    int cur = 0;  // d_X[cur]: current matrix; d_X[cur^1]: receives the rows updated this epoch
    bool have_x = false, have_walks = false;
    bool x_invalid = false;  // a lost launch left the embeddings half-updated: they must be set or initialised again
    Rand rng;
    // work-item plans (one per distinct launch: row range x neighbour source), see plan_for()
    uint32_t chunk = 64, fanin = 32;
    bool chunk_auto = true;  // f2v_train / "hub_chunk_for_batch" pick the chunk from the batch size
    bool use_quarter = true;  // sub-wave kernel when D is a multiple of 4 up to 256
    std::map<std::tuple<uint32_t, uint32_t, int>, Plan> plans;
    // chained minibatches ("chain_batches"): one launch per group of minibatches of f2v_train
    std::map<std::tuple<uint32_t, uint32_t, uint32_t, int>, ChainPlan> chains;  // (first minibatch, minibatches, batch size, walk samples instead of CSR neighbours)
    std::vector<WgDesc> h_wg;
    WgDesc *d_wg = nullptr;
    size_t d_wg_cap = 0, d_wg_valid = 0;
    // ... in the wide form ("chain_wide", the default where the fan-in allows it): workgroup programs and their jobs
    std::map<std::tuple<uint32_t, uint32_t, uint32_t, int>, WidePlan> wides;
    PlanVec<WideDesc> h_wide;
    PlanVec<WJob> h_jobs;
    WideDesc *d_wide = nullptr;
    WJob *d_jobs = nullptr;
    size_t d_wide_cap = 0, d_wide_valid = 0, d_jobs_cap = 0, d_jobs_valid = 0;
    bool wide = true;
    uint32_t wide_max_batch = 2048;  // larger chained minibatches are throughput-bound: they keep the HBM form (tools/wide_sweep.py)
    uint32_t wide_rows = 262144;     // rows one launch of the wide form covers ("chain_rows" is the HBM form's)
    uint32_t wide_order = 0;    // workgroups of a minibatch: 0 helpers, finishers, packed rows; 1 helpers, packed, finishers; 2 packed, helpers, finishers
    uint32_t wide_phases = 1;   // phases (of 32 piece slots) a workgroup of small rows runs
    uint32_t wide_min_width = 0;   // "wide_min_width" (wide_width; 0: chosen from the graph and the batch)
    // "wide_epochs": up to this many EPOCHS of a small graph in one wide-form launch (WideArgs::wgs_per_epoch): ring of matrices, per-epoch
    // row flags / partial-sum slots / their flags; 0 = automatic (32 for graphs of up to 2 M nonzeros that fit one launch, else 1)
    uint32_t wide_epochs = 0, last_wide_epochs = 1;
    float *d_ring = nullptr, *d_ring_partials = nullptr;
    uint32_t *d_ring_flags = nullptr, *d_ring_ready = nullptr;
    uint32_t ring_epochs = 0;   // epochs the ring buffers are sized for
    bool ring_refused = false;  // the device had no room for them
    size_t ring_slots = 0;      // partial-sum slots per epoch they are sized for
    // "wide_single" (measurement only): the wide form also where a launch holds ONE minibatch -- nothing is handed on inside such a
    // launch, what is left of the form is "a split row's pieces meet in LDS" (no partial sums through HBM, no tree nodes below
    // fanin^2 pieces): the plain launch form's alternative for large minibatches, measured and rejected (profiles/r04_*)
    bool wide_single = false;
    // "replicate_small": f2v_train_sharded at a batch size f2v_train would run chained (the reference's default 384 is one) runs the WHOLE
    // epoch on every rank and exchanges nothing: such an epoch is one chain of row-to-row dependencies (1 432 hops of ~5 us at batch 256
    // on RMAT-20) that a hop over xGMI can only lengthen -- sharded with one launch and one barrier per minibatch it took 41 ms per epoch
    // at batch 384 on two ranks where one GPU takes 6.7.  Every rank holds the same matrix and the same rand() state before and after
    // (the chained forms are deterministic).  1 (default): where no peer shares this rank's GPU (waiting workgroups of two processes on
    // one card can starve each other: bounded and recovered, but slow); 2: always; 0: never.
    int replicate_small = 1;
    int wide_samples_early = -1;  // "wide_samples_early": StepArgs::samples_early; -1 = automatic (graphs of up to 2 M nonzeros: the launch is one dependency chain)
    uint32_t wide_rounds = 0;   // rounds per phase of such a workgroup (0: one for minibatches of up to 512 rows, else as many as fill the piece slots)
    uint32_t wide_span = 2;     // fan-in groups per helper workgroup
    uint32_t wide_finish = 4;   // fan-in groups the finisher workgroup keeps for itself (the ones that wait longest)
    uint32_t *d_rowflag = nullptr;  // per row: sequence number of the chained launch that last wrote it
    bool chain = true;            // "chain_batches"
    uint32_t chain_max_batch = 4096, chain_rows = 65536;  // measured on RMAT-20 (tools/small_batch.py, tools/chain_sweep.py)
    PlanVec<Item> h_items;
    PlanVec<FinItem> h_hubs;
    Item *d_items = nullptr;
    FinItem *d_hubs = nullptr;
    size_t d_items_cap = 0, d_hubs_cap = 0, d_items_valid = 0, d_hubs_valid = 0;
    size_t partial_slots = 0, max_slots = 0;
    // rows [upd_lo, upd_hi) have been updated since the last swap/fold and live in d_X[cur^1]
    uint32_t upd_lo = 0, upd_hi = 0;
    bool pending = false;             // a minibatch has been stepped since the last flush
    uint32_t p_lo = 0, p_hi = 0;      // ... and this is it (multi-GPU exchange window)
    int waves_per_block = 4;
    bool fast_rng = false;        // non-parity mode: device-side init and option-7 walks (counter-based RNG)
    uint64_t fast_seed = 1, fast_epoch = 0;
    int rows_in_flight = 0;  // 0: the kernels' default (4 at D = 128 and 256, 8 below)
    bool class_cut = true;         // split rows are also cut where their neighbour ids cross into the next eighth of the id range (piece_cuts)
    bool piece_affinity = true;    // hub pieces are placed on the XCD that owns their neighbours' id range (see plan_for)
    bool last_replicated = false;  // the last f2v_train_sharded ran the whole epoch on this rank ("replicate_small")
    bool shared_card = false;      // a peer of the push exchange runs on the same GPU: placement goes back to same-XCD groups
    bool count_compulsory = false;  // plans also count their compulsory bytes (f2v_stats.compulsory_bytes; costs O(nnz) per new plan)
    SeenScratch seen_scratch;  // ... with this marker array
    bool use_graph = false;  // f2v_train replays one hipGraph per epoch parity instead of launching eagerly
    f2v_stats stats{};
    // multi-GPU push exchange (include/f2v.h): peers' matrices and flags mapped through HIP IPC
    struct Push {
        bool exported = false, attached = false, local = false;  // local: peers are handles of this process (self-test)
        uint32_t rank = 0, world = 1;
        float *peer_X[2][kMaxRanks] = {};
        unsigned long long *flags = nullptr, *peer_flags[kMaxRanks] = {};
        uint32_t *d_err = nullptr;
        unsigned long long seq = 0;
        int64_t timeout_ms = 20000;
        bool fused = true;  // rows are pushed by the step / finalize kernels themselves ("push_fused" = 0: by a kernel after them)
        // reader masks: neighbour part cached per (batch, world), sampled vertices patched in per run
        std::vector<uint32_t> base_masks, patched_ids;
        uint32_t mask_batch = 0, mask_world = 0;
        uint32_t *d_masks = nullptr, *d_patch = nullptr;
        size_t patch_cap = 0;
        uint64_t pushed_per_epoch = 0, rows_pushed = 0, rows_allgather = 0;
        bool any_shared = false, all_chain = true;  // over all attached ranks (f2v_push_attach): "replicate_small"
        // Landing-buffer mode: hipIpcOpenMemHandle never returns for an allocation of 2 GiB or more (ROCm 7.x), so a
        // matrix that large cannot be mapped into the peers.  The peers then push a minibatch's rows into a small
        // mapped buffer (two halves, alternating per exchange) and unpack_rows_kernel moves them into the matrix
        // behind the barrier.
        bool landing = false, force_landing = false;
        float *landing_buf = nullptr, *peer_landing[kMaxRanks] = {};
        uint32_t landing_cap = 0;  // rows per half
        uint32_t round = 0;        // exchanges done: (round & 1) is the half in use
    } push;
};

namespace {

int pick_vec(uint32_t D) {
    int v = 1;
    while ((uint32_t)(64 * v) < D) v <<= 1;
    return v;
}

// Hub chunk for minibatches of `batch` rows: a chunk is one quarter-wave's serial stretch (about
// 0.6 us per neighbour), the launch as a whole streams ~516 B per nonzero at ~6 TB/s; keeping the
// longest stretch at about half the launch's streaming time gives chunk ~ batch nonzeros / 14000
// (measured optimum on RMAT-20: 16 at B=4096, 32 at B=16384, 128 at B=65536).  It depends only on
// the graph and the number of rows one launch covers (the batch, or a rank's slice of it in f2v_train_sharded).
uint32_t auto_chunk(const f2v_ctx *c, uint32_t batch) {
    const double est = (double)std::min(batch, c->n) * ((double)c->nnz / (double)c->n) / 14000.0;
    // (from 4: minibatches of up to ~2000 rows are bound by the latency of their longest item, not by throughput -- batch 256
    // chained: 15.6 ms per epoch with 4-neighbour pieces, 18.2 with 8, 23.6 with 16)
    uint32_t ch = 4;
    while (ch < 512 && (double)ch * 1.5 < est) ch <<= 1;
    return ch;
}

void drop_plans(f2v_ctx *c) {
    c->plans.clear();
    c->chains.clear();
    c->h_wg.clear();
    c->d_wg_valid = 0;
    c->wides.clear();
    c->h_wide.clear();
    c->h_jobs.clear();
    c->d_wide_valid = c->d_jobs_valid = 0;
    c->max_slots = 0;
    c->h_items.clear();
    c->h_hubs.clear();
    c->d_items_valid = c->d_hubs_valid = 0;
}

// Width of the sub-wave layout that serves this D -- the smallest of 16, 32, 64, 128, 256 that holds it, for any D that
// is a multiple of 4 (rows then stay 16-byte aligned; dims past D are the tree's zero padding) -- or 0: generic layout.
uint32_t subwave_width(const f2v_ctx *c) {
    if (!c->use_quarter || c->D % 4u != 0u || c->D > 256u) return 0u;
    uint32_t w = 16;
    while (w < c->D) w <<= 1;
    return w;
}

// items one workgroup of the step kernel covers, and tree nodes one workgroup covers (= its wavefronts)
uint32_t items_per_block(const f2v_ctx *c) {
    const uint32_t w = subwave_width(c);
    const uint32_t per_wave = !w ? 1u : (w == 16 ? 16u : w == 32 ? 8u : 4u);
    return per_wave * (uint32_t)c->waves_per_block;
}

// Where a split row (degree > chunk) is cut into pieces: after every `chunk` neighbours, and -- "class_cut", the default --
// also wherever the (ascending) neighbour ids cross from one eighth of the id range into the next, so that all neighbours of
// a piece belong to ONE of the kIdClasses id ranges and the piece can run on the XCD whose L2 caches that range
// ("piece_affinity").  The cuts are part of the summation order (a piece accumulates from zero, the pieces' sums are combined
// in order): the oracle restates exactly this rule (oracle/f2v_oracle.c: piece_cuts), and it depends on nothing but the
// row's neighbour ids, the chunk and N -- not on the machine, the placement or the number of GPUs.
// Appends the offsets of the pieces' first neighbours and, last, the degree.
constexpr uint32_t kIdClasses = 8;
void piece_cuts(const f2v_ctx *c, uint32_t row, std::vector<uint32_t> &cuts) {
    const uint32_t rp = c->rowptr[row], deg = c->rowptr[row + 1] - rp;
    if (!c->class_cut) {
        for (uint32_t b = 0; b < deg; b += c->chunk) cuts.push_back(b);
        cuts.push_back(deg);
        return;
    }
    auto cls_of = [&](uint32_t e) { return (uint32_t)(((uint64_t)c->colids[rp + e] * kIdClasses) / c->n); };
    uint32_t start = 0, cls = cls_of(0);
    cuts.push_back(0);
    for (uint32_t e = 1; e < deg; e++) {
        const uint32_t ce = cls_of(e);
        if (ce != cls || e - start == c->chunk) {
            cuts.push_back(e);
            start = e;
            cls = ce;
        }
    }
    cuts.push_back(deg);
}

// Is row i cut into pieces?  More than `chunk` neighbours.  (Round 3 tried cutting rows of 17 ... 128 neighbours at the id-class
// boundaries as well, so that their pieces get "piece_affinity" too: L2 hit rate 0.513 -> 0.537, +8 ... 11 % time for the extra
// partial sums -- profiles/r03_class_split_min_rejected.txt.)
bool is_split(const f2v_ctx *c, uint32_t i) { return c->chunk != 0 && c->rowptr[i + 1] - c->rowptr[i] > c->chunk; }

// Compulsory bytes of one minibatch: every DISTINCT embedding row it reads (its own rows and their neighbours) once, every
// row it writes once, its neighbour ids and work items once -- what would still cross HBM if everything read twice inside
// the minibatch came from a cache the second time.  (The ns sampled rows, the partial sums of split rows and rowptr are
// left out: a lower bound.)
uint64_t compulsory_bytes(const f2v_ctx *c, SeenScratch &sc, uint32_t row_lo, uint32_t row_hi, bool walk, uint64_t nnz, uint64_t n_items) {
    if (sc.seen.size() != c->n) { sc.seen.assign(c->n, 0u); sc.stamp = 0; }
    if (++sc.stamp == 0) { std::fill(sc.seen.begin(), sc.seen.end(), 0u); sc.stamp = 1; }
    const uint32_t st = sc.stamp;
    uint32_t *seen = sc.seen.data();
    uint64_t distinct = 0;
    const uint32_t *ids = walk ? nullptr : c->colids.data();
    for (uint32_t i = row_lo; i < row_hi; i++) {
        if (seen[i] != st) { seen[i] = st; distinct++; }
        if (!ids) continue;  // walk samples change every epoch: only the rows themselves are counted
        for (uint32_t k = c->rowptr[i]; k < c->rowptr[i + 1]; k++) {
            const uint32_t j = ids[k];
            if (seen[j] != st) { seen[j] = st; distinct++; }
        }
    }
    return distinct * 4ull * c->D + (uint64_t)(row_hi - row_lo) * 4ull * c->D + nnz * 4ull + n_items * sizeof(Item);
}

// The plan cache may hold a few epochs' worth of items (one epoch: every row or piece once -- about nnz / chunk + 8 per split
// row + n); beyond that it is dropped and rebuilt on demand (a run that alternates batch sizes).
size_t plan_cache_limit(const f2v_ctx *c) {
    const size_t epoch = (size_t)(c->nnz / std::max<uint32_t>(c->chunk, 1u)) + 2 * (size_t)c->n + 1024;
    return std::max<size_t>(8 * ((size_t)c->n + 1024), 3 * epoch);
}

// Work items of one launch (rows [row_lo,row_hi), CSR neighbours or walk samples): a whole row, or
// `chunk`-neighbour pieces of a hub row, longest first (the longest waves start first and the four quarters of a
// wave get items of nearly equal length).  Host-side, cached per launch shape.
//
// Who may wait for whom.  Tree nodes WAIT for the partial sums they add (one launch per minibatch), so what they wait
// for must be running or done whatever else the GPU is doing.  Workgroups go to the 8 XCDs round robin (workgroup b
// to XCD b mod 8) and every XCD starts its workgroups in index order; inside one process that is enough.  But a
// second process whose own waiting nodes fill one XCD can keep this launch's first workgroups from starting there
// while the other seven XCDs run ahead into the tree nodes -- two such processes can wait for each other (seen with
// three replays sharing one card: bounded by the time-out, but seconds long).  So the bulk of the waiting -- the
// lowest tree level, one node per group of `fanin` pieces -- is tied to an XCD: a group gets a CLASS 0..7 and its
// pieces and its node are placed in workgroups whose index is congruent to the class modulo 8 (whole rows and, at
// the very end, inert padding items fill the gaps): such a node only ever waits for workgroups of its OWN XCD with
// a smaller index, which that XCD has started before it.  The upper levels are 1/fanin as many nodes -- too few to
// fill an XCD, so they cannot take part in such a ring -- and stay unconstrained.
const Plan &plan_for(f2v_ctx *c, uint32_t row_lo, uint32_t row_hi, bool walk) {
    const auto key = std::make_tuple(row_lo, row_hi, walk ? 1 : 0);
    auto itp = c->plans.find(key);
    if (itp != c->plans.end()) return itp->second;
    if (c->h_items.size() > plan_cache_limit(c)) drop_plans(c);  // bound the cache
    constexpr uint32_t kXcds = 8;
    const uint32_t ipb = items_per_block(c), npb = (uint32_t)c->waves_per_block;
    Plan p;
    p.item_off = c->h_items.size();
    std::vector<Item> whole;  // rows that stay one item
    whole.reserve(row_hi - row_lo);
    struct Node { uint32_t row, in_slot, n, cut; };  // cut: index of the row's first piece boundary in `cuts`
    std::vector<Node> cur, nxt;
    std::vector<uint32_t> cuts;  // per split row: offsets of its pieces' first neighbours, then its degree
    uint32_t slots = 0;
    for (uint32_t i = row_lo; i < row_hi; i++) {
        if (walk) {
            whole.push_back(Item{i, i * (uint32_t)kWalkLength, (uint32_t)kWalkLength, kItemFirst | kItemLast});
            p.nnz += kWalkLength;
            continue;
        }
        const uint32_t rp = c->rowptr[i], deg = c->rowptr[i + 1] - rp;
        p.nnz += deg;
        if (is_split(c, i)) {
            const uint32_t cut0 = (uint32_t)cuts.size();
            piece_cuts(c, i, cuts);
            const uint32_t nc = (uint32_t)cuts.size() - cut0 - 1;
            cur.push_back(Node{i, slots, nc, cut0});
            slots += nc;
        } else {
            whole.push_back(Item{i, rp, deg, kItemFirst | kItemLast});
        }
    }
    p.n_hubs = (uint32_t)cur.size();
    p.n_chunks = slots;
    std::vector<Item> items;
    items.reserve(whole.size() + slots + 64);
    // inert filler: an empty piece of row `row_lo` whose (zero) partial sum goes to a slot of its own that no node reads --
    // the kernels need no special case for it
    Item pad_item{row_lo, 0, 0, kItemPartial};
    bool padded = false;
    const FinItem pad_node{0, 0, kFinToStage, 0};
    std::stable_sort(whole.begin(), whole.end(), [](const Item &x, const Item &y) { return x.cnt > y.cnt; });
    if (!cur.empty()) {
        // Lowest tree level first: one node per GROUP of `fanin` consecutive pieces of a hub.  A group -- its pieces and
        // its node -- gets the class that has been given the fewest neighbours so far (a giant hub thus spreads over all XCDs).
        std::vector<Item> queue[kXcds];
        std::vector<FinItem> nq[kXcds];
        uint64_t load[kXcds] = {}, aload[kXcds] = {};  // neighbours given to every class so far: a group goes to the lightest one
        for (const Node &nd : cur) {
            const uint32_t rp = c->rowptr[nd.row];
            const uint32_t *cut = cuts.data() + nd.cut;
            const uint32_t G = c->fanin < 2 ? nd.n : c->fanin;
            const uint32_t nout = (nd.n + G - 1) / G;
            for (uint32_t o = 0; o < nout; o++) {
                const uint32_t k0 = o * G, k1 = std::min(nd.n, k0 + G);
                uint32_t cls = 0;
                for (uint32_t k = 1; k < kXcds; k++)
                    if (load[k] < load[cls]) cls = k;
                load[cls] += cut[k1] - cut[k0];
                for (uint32_t k = k0; k < k1; k++) {
                    const uint32_t b = cut[k], e = cut[k + 1];
                    // "piece_affinity" (DESIGN.md section 3): a piece goes to the XCD that owns the id range of its (ascending)
                    // neighbours -- the lightest of the XCDs whose ranges it touches: midpoints alone leave the outer ranges short
                    // of work and cost 40 % -- so that every L2 caches an eighth of the matrix instead of all eight the same
                    // hubs: RMAT-20, batch 65536: L2 hit rate 0.31 -> 0.42, L2-miss reads -17 %, epoch time -7...-10 %.
                    // Placement only: bits do not change.  The group's node stays in class `cls`, so it may now wait for other
                    // XCDs' workgroups: fine inside one process (index order), not with a second process on the card that could
                    // fill an XCD with ITS waiting nodes -- ranks that share a card are detected (f2v_push_attach) and keep
                    // the same-XCD groups.
                    uint32_t pc = cls;
                    if (c->piece_affinity && !c->shared_card) {
                        const uint32_t c0 = (uint32_t)(((uint64_t)c->colids[rp + b] * kXcds) / c->n), c1 = (uint32_t)(((uint64_t)c->colids[rp + e - 1] * kXcds) / c->n);
                        pc = c0;
                        for (uint32_t q = c0 + 1; q <= c1; q++)
                            if (aload[q] < aload[pc]) pc = q;
                        aload[pc] += e - b;
                    }
                    queue[pc].push_back(Item{nd.row, rp + b, e - b, kItemPartial | (k == 0 ? kItemFirst : 0u) | (k == nd.n - 1 ? kItemLast : 0u) | (nd.in_slot + k)});
                }
                nq[cls].push_back(FinItem{nd.in_slot + k0, k1 - k0, nout == 1 ? kFinToStage : slots + o, nd.row});
            }
            if (nout > 1) {
                nxt.push_back(Node{nd.row, slots, nout, 0});
                slots += nout;
            }
        }
        for (auto &q : queue) std::stable_sort(q.begin(), q.end(), [](const Item &x, const Item &y) { return x.cnt > y.cnt; });
        // One workgroup per class per round, longest first overall: a class's workgroup takes its own pieces as long as
        // they are at least as long as the longest whole row still waiting (whole rows may run anywhere), else whole rows.
        size_t pos[kXcds] = {}, wpos = 0;
        {
            for (bool more = true; more;) {
                more = false;
                for (uint32_t k = 0; k < kXcds; k++) {
                    for (uint32_t j = 0; j < ipb; j++) {
                        const bool mine = pos[k] < queue[k].size(), any = wpos < whole.size();
                        if (mine && (!any || queue[k][pos[k]].cnt >= whole[wpos].cnt)) items.push_back(queue[k][pos[k]++]);
                        else if (any) items.push_back(whole[wpos++]);
                        else { items.push_back(pad_item); padded = true; }
                    }
                    more = more || pos[k] < queue[k].size();
                }
            }
        }
        items.insert(items.end(), whole.begin() + wpos, whole.end());
        // the nodes' workgroups follow the items': their first one must again be a multiple of 8
        while (items.size() % ((size_t)kXcds * ipb) != 0) { items.push_back(pad_item); padded = true; }
        p.fin_off[0] = c->h_hubs.size();
        size_t npos[kXcds] = {};
        for (bool more = true; more;) {
            more = false;
            for (uint32_t k = 0; k < kXcds; k++) {
                for (uint32_t j = 0; j < npb; j++) c->h_hubs.push_back(npos[k] < nq[k].size() ? nq[k][npos[k]++] : pad_node);
                more = more || npos[k] < nq[k].size();
            }
        }
        p.fin_cnt[0] = (uint32_t)(c->h_hubs.size() - p.fin_off[0]);
        p.n_levels = 1;
        cur.swap(nxt);
    } else {
        items.insert(items.end(), whole.begin(), whole.end());
    }
    const size_t items_at = c->h_items.size();
    c->h_items.insert(c->h_items.end(), items.begin(), items.end());
    p.n_items = (uint32_t)items.size();
    // upper levels: groups of `fanin` sums are added in order until one is left.  Their nodes are few (1/fanin of the
    // level below): wherever they wait, they cannot fill an XCD, so they are packed without regard to classes.
    while (!cur.empty() && p.n_levels < kMaxFinLevels) {
        p.fin_off[p.n_levels] = c->h_hubs.size();
        nxt.clear();
        for (const Node &nd : cur) {
            const uint32_t G = (p.n_levels == kMaxFinLevels - 1 || c->fanin < 2) ? nd.n : c->fanin;
            const uint32_t nout = (nd.n + G - 1) / G;
            if (nout == 1) {
                c->h_hubs.push_back(FinItem{nd.in_slot, nd.n, kFinToStage, nd.row});
            } else {
                for (uint32_t o = 0; o < nout; o++)
                    c->h_hubs.push_back(FinItem{nd.in_slot + o * G, std::min(G, nd.n - o * G), slots + o, nd.row});
                nxt.push_back(Node{nd.row, slots, nout, 0});
                slots += nout;
            }
        }
        p.fin_cnt[p.n_levels] = (uint32_t)(c->h_hubs.size() - p.fin_off[p.n_levels]);
        p.n_levels++;
        cur.swap(nxt);
    }
    if (padded) {  // the fillers' slot: one past every slot a node reads
        for (size_t k = items_at; k < items_at + p.n_items; k++)
            if (c->h_items[k].cnt == 0 && c->h_items[k].flags == kItemPartial) c->h_items[k].flags = kItemPartial | slots;
        slots++;
    }
    p.n_slots = slots;
    if (slots > kItemSlotMask) c->plan_overflow = true;  // the slot index would run into the item's flag bits: the caller fails the call
    if (c->count_compulsory) p.compulsory = compulsory_bytes(c, c->seen_scratch, row_lo, row_hi, walk, p.nnz, p.n_items);
    c->max_slots = std::max<size_t>(c->max_slots, slots);
    return c->plans.emplace(key, p).first->second;
}

// Minibatches per chained launch ("chain_rows" rows per launch, at most 4096 minibatches)
uint32_t chain_len(const f2v_ctx *c, uint32_t batch, bool wide_form = false) {
    uint64_t k = (uint64_t)(wide_form ? c->wide_rows : c->chain_rows) / std::max(batch, 1u);
    // handed-on rows are read through a buffer resource that starts at the launch's first row, at 32-bit byte offsets (load16_agent)
    k = std::min<uint64_t>(k, 0xFFFFFFFFull / ((uint64_t)std::max(batch, 1u) * c->D * sizeof(float)));
    return (uint32_t)std::min<uint64_t>(4096, std::max<uint64_t>(k, 1));
}

// The sub-wave layout the wide form runs a D on.  Narrow rows on a wider layout leave part of every lane group idle, but a
// wavefront then holds fewer items -- and a wavefront of a chained launch is as late as the latest of its items.  Measured
// (tools/wide_min_width.py, tools/cora_configs.py): cora at D = 16, 1200 epochs: 0.098 s on the 16-wide layout (16 items per
// wavefront), 0.086 s on 32, 0.0795 s on 64 (4 items); RMAT-20 at D = 16: 32 wide is 10 % faster than 16 at batch 256 and 17 %
// slower at 2048, 64 wide slower everywhere (throughput counts there).  "wide_min_width" = 0 chooses: 64 for small graphs (up to
// 2 M nonzeros), 32 for minibatches of up to 1024 rows, else the narrowest layout that holds D.  The zero padding of the wider
// tree changes no bit.
uint32_t wide_width(const f2v_ctx *c, uint32_t batch) {
    const uint32_t floor_w = c->wide_min_width ? c->wide_min_width : (c->nnz <= (2ull << 20) ? 64u : batch <= 1024u ? 32u : 16u);
    return std::max(subwave_width(c), floor_w);
}
uint32_t wide_items_per_block(uint32_t width) { return 4u * (width == 16 ? 16u : width == 32 ? 8u : 4u); }

// the wide form of a chained launch: jobs add at most 32 LDS slots, so the fan-in groups must fit
bool wide_usable(const f2v_ctx *c) { return c->wide && c->fanin >= 2 && c->fanin <= 32 && c->waves_per_block == 4; }

bool chain_usable(const f2v_ctx *c, int math, uint32_t batch, int bs_mode, bool sharded) {
    const uint32_t nb = (uint32_t)(((uint64_t)c->n + batch - 1) / batch);
    (void)bs_mode;  // -bs 1 chains too: its per-row sample windows are gathered per item, and those gathers wait for rows like any other
    (void)math;     // option 7 chains too: its five walk samples per row are gathered (and waited for) like CSR neighbours
    // rows narrower than a 128-byte line (D = 16, 8, 4 ...): only in the wide form, whose handed-off rows are read with agent-scope
    // loads alone, and only where no line holds rows of two minibatches (every reader then treats all of a line's rows alike)
    const bool lines_ok = c->D % 32u == 0u || (wide_usable(c) && batch <= c->wide_max_batch && (chain_len(c, batch, true) >= 2 || c->wide_single) && ((uint64_t)batch * c->D) % 32u == 0u);
    return c->chain && !sharded && c->merge_fin && c->xcc_round_robin && !c->capturing && !c->use_graph &&
           subwave_width(c) != 0 && lines_ok && batch <= c->chain_max_batch && nb >= 2 && (chain_len(c, batch) >= 2 || c->wide_single);
}

// The work of minibatches [b0, b0+K) of batch size `batch` as ONE launch: per minibatch its items (rows whose neighbours all
// lie outside the launch's earlier minibatches first: their workgroups never wait), then its combine-tree nodes; a
// descriptor per workgroup with the minibatches it has to wait for.  Same pieces, same fan-in, same slots-in-chunk-order as
// plan_for: the summation order -- and with it every bit of the result -- does not depend on how minibatches are launched.
const ChainPlan &chain_plan_for(f2v_ctx *c, uint32_t b0, uint32_t K, uint32_t batch, bool walk) {
    const auto key = std::make_tuple(b0, K, batch, walk ? 1 : 0);
    auto itp = c->chains.find(key);
    if (itp != c->chains.end()) return itp->second;
    if (c->h_items.size() > plan_cache_limit(c)) drop_plans(c);  // bound the cache
    const uint32_t ipb = items_per_block(c), npb = (uint32_t)c->waves_per_block;
    ChainPlan p;
    p.first_batch = b0;
    p.n_batches = K;
    p.item_off = c->h_items.size();
    p.fin_off = c->h_hubs.size();
    p.wg_off = c->h_wg.size();
    p.lo = (uint32_t)std::min<uint64_t>((uint64_t)b0 * batch, c->n);
    uint32_t slots = 0;
    struct DI { Item it; uint64_t dep; };
    std::vector<DI> items;
    struct Node { uint32_t row, in_slot, n; };
    std::vector<Node> cur, nxt;
    std::vector<uint32_t> cutbuf;
    for (uint32_t k = 0; k < K; k++) {
        const uint32_t lo = (uint32_t)std::min<uint64_t>((uint64_t)(b0 + k) * batch, c->n);
        const uint32_t hi = (uint32_t)std::min<uint64_t>((uint64_t)lo + batch, c->n);
        p.hi = hi;
        items.clear();
        cur.clear();
        uint64_t nnz = 0;
        auto dep_of = [&](uint32_t from, uint32_t to) -> uint64_t {  // 1 + the LAST row of this launch's earlier minibatches among the neighbours [from,to): 0 = independent
            uint64_t m = 0;
            for (uint32_t e = from; e < to; e++) {
                const uint32_t j = c->colids[e];
                if (j >= p.lo && j < lo) m = std::max<uint64_t>(m, (uint64_t)(j - p.lo) + 1);
            }
            return m;
        };
        for (uint32_t i = lo; i < hi; i++) {
            if (walk) {  // option 7: the row's five walk samples of the epoch (they change every epoch: no ordering hint)
                items.push_back(DI{Item{i, i * (uint32_t)kWalkLength, (uint32_t)kWalkLength, kItemFirst | kItemLast}, 0});
                nnz += kWalkLength;
                continue;
            }
            const uint32_t rp = c->rowptr[i], deg = c->rowptr[i + 1] - rp;
            nnz += deg;
            if (is_split(c, i)) {
                cutbuf.clear();
                piece_cuts(c, i, cutbuf);
                const uint32_t nc = (uint32_t)cutbuf.size() - 1;
                for (uint32_t q = 0; q < nc; q++) {
                    const uint32_t b = cutbuf[q], e = cutbuf[q + 1];
                    items.push_back(DI{Item{i, rp + b, e - b, kItemPartial | (q == 0 ? kItemFirst : 0u) | (q == nc - 1 ? kItemLast : 0u) | (slots + q)}, dep_of(rp + b, rp + e)});
                }
                cur.push_back(Node{i, slots, nc});
                slots += nc;
                p.n_hubs++;
                p.n_chunks += nc;
            } else {
                items.push_back(DI{Item{i, rp, deg, kItemFirst | kItemLast}, dep_of(rp, rp + deg)});
            }
        }
        // independent items first (longest first), then the dependent ones, those that wait for the oldest rows first
        std::stable_sort(items.begin(), items.end(), [](const DI &x, const DI &y) {
            if ((x.dep != 0) != (y.dep != 0)) return x.dep == 0;
            if (x.dep != y.dep && x.dep != 0) return x.dep < y.dep;
            return x.it.cnt > y.it.cnt;
        });
        if (items.size() % ipb != 0) {  // inert fillers: empty pieces whose (zero) sum goes to a slot no node reads
            const Item pad{lo, 0, 0, kItemPartial | slots};
            slots++;
            while (items.size() % ipb != 0) items.push_back(DI{pad, 0});
        }
        WgDesc bd{};
        bd.lo = lo;
        p.last_lo = lo;
        bd.item_off = (uint32_t)(c->h_items.size() - p.item_off);
        bd.n_items = (uint32_t)items.size();
        bd.step_blocks = bd.n_items / ipb;
        bd.fin_off = (uint32_t)(c->h_hubs.size() - p.fin_off);
        bd.index = b0 + k;
        const size_t wg_at = c->h_wg.size();  // (fin_n is known only below: the descriptors are patched then)
        for (uint32_t w = 0; w < bd.step_blocks; w++) { bd.blk = w; c->h_wg.push_back(bd); }
        for (const DI &d : items) c->h_items.push_back(d.it);
        // the combine trees of this minibatch's split rows, level by level (fan-in groups in chunk order, as plan_for)
        uint32_t fin_n = 0;
        for (int level = 0; !cur.empty(); level++) {
            nxt.clear();
            for (const Node &nd : cur) {
                const uint32_t G = (level == kMaxFinLevels - 1 || c->fanin < 2) ? nd.n : c->fanin;
                const uint32_t nout = (nd.n + G - 1) / G;
                if (nout == 1) {
                    c->h_hubs.push_back(FinItem{nd.in_slot, nd.n, kFinToStage, nd.row});
                    fin_n++;
                } else {
                    for (uint32_t o = 0; o < nout; o++) {
                        c->h_hubs.push_back(FinItem{nd.in_slot + o * G, std::min(G, nd.n - o * G), slots + o, nd.row});
                        fin_n++;
                    }
                    nxt.push_back(Node{nd.row, slots, nout});
                    slots += nout;
                }
            }
            cur.swap(nxt);
        }
        while (fin_n % npb != 0) { c->h_hubs.push_back(FinItem{0, 0, kFinToStage, 0}); fin_n++; }
        bd.fin_n = fin_n;
        const uint32_t node_blocks = fin_n / npb;
        // (the nodes must directly follow their minibatch's items: later minibatches' items wait for the rows the roots write,
        // and a waiter may only ever wait for a SMALLER workgroup index -- letting the nodes trail by a few minibatches, so that
        // they poll less, was tried: it buys 3-9 % without the row waits and deadlocks with them, caught by the bounded waits)
        for (uint32_t w = 0; w < node_blocks; w++) { bd.blk = bd.step_blocks + w; c->h_wg.push_back(bd); }
        for (size_t w = wg_at; w < c->h_wg.size(); w++) c->h_wg[w].fin_n = fin_n;
        p.n_wgs += bd.step_blocks + node_blocks;
        p.n_fin += fin_n;
        p.nnz += nnz;
        if (c->count_compulsory) p.compulsory += compulsory_bytes(c, c->seen_scratch, lo, hi, walk, nnz, bd.n_items);
    }
    p.n_slots = slots;
    if (slots > kItemSlotMask) c->plan_overflow = true;
    c->max_slots = std::max<size_t>(c->max_slots, slots);
    return c->chains.emplace(key, p).first->second;
}


// Minibatches [b0, b0+K) as ONE launch of qwide_chain_kernel: per minibatch the helper workgroups of its multi-group rows,
// then their finishers, then the workgroups that pack whole low-degree rows and rows of one fan-in group, then the
// combine-tree nodes above the units of rows with more than fanin^2 pieces.  Pieces (piece_cuts), fan-in groups and the
// order of every addition are those of plan_for / chain_plan_for: the result does not depend on which form ran.
// (built into vectors of its own, offsets relative to them: the plans of an epoch are independent of each other and are built by
// several host threads, wide_plans_for_epoch -- one thread took 0.33 s for RMAT-20 at batch 256 and, by the same count, ~5 s for RMAT-24)
struct WideParts {
    std::vector<Item> items;
    std::vector<WJob> jobs;
    std::vector<WideDesc> wide;
    std::vector<FinItem> hubs;
};
WidePlan build_wide_plan(const f2v_ctx *c, uint32_t b0, uint32_t K, uint32_t batch, bool walk, WideParts &out, SeenScratch &seen) {
    const uint32_t layout = wide_width(c, batch);
    const uint32_t ipb = wide_items_per_block(layout);        // lane groups per workgroup = items per round
    const uint32_t pslots = std::max<uint32_t>(ipb, 32u);      // piece slots of a phase (PSLOTS of the kernel)
    const uint32_t F = c->fanin;
    WidePlan p;
    p.width = layout;
    p.first_batch = b0;
    p.n_batches = K;
    p.item_off = p.fin_off = p.wg_off = p.job_off = 0;  // (relative to `out`; wide_plan_append moves them)
    p.lo = (uint32_t)std::min<uint64_t>((uint64_t)b0 * batch, c->n);
    uint32_t slots = 0;  // partial sums in HBM: helpers' group sums, units' sums, upper tree levels

    struct Prog {  // one workgroup: items laid out round by round (ipb per round), jobs in execution order
        std::vector<Item> items;
        std::vector<WJob> jobs;
        uint32_t phases = 0;
    };
    struct Piece { Item it; uint64_t dep; };
    struct Group { uint32_t first, n; uint64_t dep; };  // pieces [first, first+n) of `pieces`
    struct Pack { uint32_t first, n; uint64_t dep; uint8_t kind; uint32_t dst, row; };  // a whole row (n = 1, direct) or a one-group unit
    std::vector<Piece> pieces;
    std::vector<Pack> packs;
    std::vector<Prog> helpers, finishers, packed;
    std::vector<uint32_t> cutbuf;
    struct Node { uint32_t row, in_slot, n; };
    std::vector<Node> cur, nxt;

    // One phase's items: every item has been given its LDS slot (= the order in which the jobs add) when it was appended; they
    // RUN in the order of what they wait for -- independent ones first, the ones that read the most recently written rows in
    // the last round -- padded to whole rounds; the last round carries kItemPhaseEnd.
    auto close_phase = [&](Prog &g, std::vector<Piece> &ph, uint32_t lo_row) {
        if (ph.empty()) return;
        std::stable_sort(ph.begin(), ph.end(), [](const Piece &x, const Piece &y) { return x.dep < y.dep; });
        const size_t at = g.items.size();
        for (const Piece &pc : ph) g.items.push_back(pc.it);
        while ((g.items.size() - at) % ipb != 0) g.items.push_back(Item{lo_row, 0, 0, kItemIdle});
        for (size_t k = g.items.size() - ipb; k < g.items.size(); k++) g.items[k].flags |= kItemPhaseEnd;
        g.phases++;
        ph.clear();
    };
    auto slotted = [](Piece pc, size_t slot) { pc.it.flags |= (uint32_t)slot; return pc; };
    auto mkjob = [](size_t src, uint32_t n, uint8_t kind, size_t phase, uint32_t dst, uint32_t row) {
        WJob j{};
        j.src = (uint8_t)src; j.n = (uint8_t)n; j.kind = kind; j.phase = (uint8_t)phase; j.dst = dst; j.row = row;
        return j;
    };
    // the jobs of one phase as passes of up to 8
    auto add_pass = [&](Prog &g, std::vector<WJob> &js) {
        for (size_t k = 0; k < js.size(); k += 8) {
            const uint8_t len = (uint8_t)std::min<size_t>(8, js.size() - k);
            for (size_t u = 0; u < len; u++) { js[k + u].pass_len = len; g.jobs.push_back(js[k + u]); }
        }
        js.clear();
    };

    for (uint32_t k = 0; k < K; k++) {
        const uint32_t lo = (uint32_t)std::min<uint64_t>((uint64_t)(b0 + k) * batch, c->n);
        const uint32_t hi = (uint32_t)std::min<uint64_t>((uint64_t)lo + batch, c->n);
        p.hi = hi;
        p.last_lo = lo;
        pieces.clear(); packs.clear(); helpers.clear(); finishers.clear(); packed.clear(); cur.clear();
        uint64_t nnz = 0;
        uint32_t mb_items = 0;
        auto dep_of = [&](uint32_t from, uint32_t to) -> uint64_t {  // 1 + the LAST row of this launch's earlier minibatches among the neighbours [from,to): 0 = independent
            uint64_t m = 0;
            for (uint32_t e = from; e < to; e++) {
                const uint32_t j = c->colids[e];
                if (j >= p.lo && j < lo) m = std::max<uint64_t>(m, (uint64_t)(j - p.lo) + 1);
            }
            return m;
        };
        for (uint32_t i = lo; i < hi; i++) {
            if (walk) {  // option 7: the row's five walk samples of the epoch
                pieces.push_back(Piece{Item{i, i * (uint32_t)kWalkLength, (uint32_t)kWalkLength, kItemFirst | kItemLast | kItemDirect}, 0});
                packs.push_back(Pack{(uint32_t)pieces.size() - 1, 1, 0, kJobRow, 0, i});
                nnz += kWalkLength;
                continue;
            }
            const uint32_t rp = c->rowptr[i], deg = c->rowptr[i + 1] - rp;
            nnz += deg;
            if (!is_split(c, i)) {
                pieces.push_back(Piece{Item{i, rp, deg, kItemFirst | kItemLast | kItemDirect}, dep_of(rp, rp + deg)});
                packs.push_back(Pack{(uint32_t)pieces.size() - 1, 1, pieces.back().dep, kJobRow, 0, i});
                continue;
            }
            cutbuf.clear();
            piece_cuts(c, i, cutbuf);
            const uint32_t P = (uint32_t)cutbuf.size() - 1;
            p.n_hubs++;
            p.n_chunks += P;
            const uint32_t first_piece = (uint32_t)pieces.size();
            for (uint32_t q = 0; q < P; q++) {
                const uint32_t b = cutbuf[q], e = cutbuf[q + 1];
                pieces.push_back(Piece{Item{i, rp + b, e - b, (q == 0 ? kItemFirst : 0u) | (q == P - 1 ? kItemLast : 0u)}, dep_of(rp + b, rp + e)});
            }
            // units of F*F pieces (F groups of F): one unit is the whole row unless the row has more pieces than that
            const uint32_t per_unit = F * F;
            const uint32_t n_units = (P + per_unit - 1) / per_unit;
            uint32_t unit_base = 0;
            if (n_units > 1) {
                unit_base = slots;
                slots += n_units;
                cur.push_back(Node{i, unit_base, n_units});
            }
            for (uint32_t u = 0; u < n_units; u++) {
                const uint32_t u0 = u * per_unit, u1 = std::min(P, u0 + per_unit);
                const uint8_t out_kind = n_units > 1 ? kJobPart : kJobRow;
                const uint32_t out_dst = n_units > 1 ? unit_base + u : 0u;
                const uint32_t G = (u1 - u0 + F - 1) / F;
                if (G == 1) {  // one fan-in group: packed with other small rows
                    uint64_t dep = 0;
                    for (uint32_t q = u0; q < u1; q++) dep = std::max(dep, pieces[first_piece + q].dep);
                    packs.push_back(Pack{first_piece + u0, u1 - u0, dep, out_kind, out_dst, i});
                    continue;
                }
                std::vector<Group> groups(G);
                for (uint32_t g = 0; g < G; g++) {
                    const uint32_t q0 = u0 + g * F, q1 = std::min(u1, q0 + F);
                    uint64_t dep = 0;
                    for (uint32_t q = q0; q < q1; q++) dep = std::max(dep, pieces[first_piece + q].dep);
                    groups[g] = Group{first_piece + q0, q1 - q0, dep};
                }
                // the finisher keeps the groups that depend on the most recently written rows (they wait longest)
                std::vector<uint32_t> order(G);
                for (uint32_t g = 0; g < G; g++) order[g] = g;
                std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return groups[x].dep < groups[y].dep; });
                const uint32_t n_own = std::min(G, std::max(1u, c->wide_finish));
                std::vector<uint32_t> own(order.end() - n_own, order.end());   // oldest dependency first, the latest last
                std::vector<uint32_t> rest(order.begin(), order.end() - n_own);
                std::sort(rest.begin(), rest.end());
                const uint32_t l0_base = slots;
                if (!rest.empty()) slots += G;
                // helpers: `wide_span` groups each, group sums written through to HBM
                for (size_t h0 = 0; h0 < rest.size(); h0 += std::max(1u, c->wide_span)) {
                    Prog g;
                    std::vector<Piece> ph;
                    std::vector<WJob> js;
                    for (size_t h = h0; h < std::min(rest.size(), h0 + std::max(1u, c->wide_span)); h++) {
                        const Group &gr = groups[rest[h]];
                        if (ph.size() + gr.n > pslots) { close_phase(g, ph, lo); add_pass(g, js); }
                        js.push_back(mkjob(ph.size(), gr.n, kJobPart, g.phases, l0_base + rest[h], i));
                        for (uint32_t q = 0; q < gr.n; q++) ph.push_back(slotted(pieces[gr.first + q], ph.size()));
                    }
                    close_phase(g, ph, lo);
                    add_pass(g, js);
                    helpers.push_back(std::move(g));
                }
                // finisher: its own groups (sums -> LDS), the helpers' sums imported before its last phase, then the unit's sum
                {
                    Prog g;
                    std::vector<Piece> ph;
                    std::vector<WJob> js, imports;
                    for (uint32_t r : rest) imports.push_back(mkjob(0, 0, kJobImport, 0, pslots + r, l0_base + r));
                    // phases of the own groups, in `own` order; the last phase starts with the last group that does not fit the one before
                    std::vector<std::vector<uint32_t>> phases(1);
                    {
                        uint32_t used = 0;
                        for (uint32_t g2 : own) {
                            if (used + groups[g2].n > pslots) { phases.emplace_back(); used = 0; }
                            phases.back().push_back(g2);
                            used += groups[g2].n;
                        }
                    }
                    for (size_t f = 0; f < phases.size(); f++) {
                        if (f + 1 == phases.size() && !imports.empty()) {  // imports run before the last phase's rounds
                            for (WJob &w : imports) w.phase = f == 0 ? kJobBefore : (uint8_t)(f - 1);
                            add_pass(g, imports);
                        }
                        for (uint32_t g2 : phases[f]) {
                            const Group &gr = groups[g2];
                            js.push_back(mkjob(ph.size(), gr.n, kJobLds, f, pslots + g2, i));
                            for (uint32_t q = 0; q < gr.n; q++) ph.push_back(slotted(pieces[gr.first + q], ph.size()));
                        }
                        close_phase(g, ph, lo);
                        if (f + 1 == phases.size() && js.size() == 1) break;  // the last phase's only group: added by the final job itself
                        add_pass(g, js);
                    }
                    std::vector<WJob> fin{mkjob(pslots, G, out_kind, phases.size() - 1, out_dst, i)};
                    if (js.size() == 1) {  // ... its pieces first (into its sum slot), then the groups' sums: one job, no barrier in between
                        fin[0].src2 = js[0].src; fin[0].n2 = js[0].n; fin[0].dst2 = (uint8_t)js[0].dst;
                        js.clear();
                    }
                    add_pass(g, fin);
                    finishers.push_back(std::move(g));
                }
            }
        }
        // whole rows and one-group units, packed: independent ones first (longest first), then by the age of what they wait for
        {
            std::vector<uint32_t> order(packs.size());
            for (uint32_t x = 0; x < order.size(); x++) order[x] = x;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
                const Pack &a = packs[x], &b = packs[y];
                if ((a.dep != 0) != (b.dep != 0)) return a.dep == 0;
                if (a.dep != b.dep) return a.dep < b.dep;
                return a.n > b.n;
            });
            std::vector<char> taken(packs.size(), 0);
            size_t head = 0;
            // items per phase (a longer one-group unit has a phase to itself): one round where minibatches are small -- they are
            // latency, a second round of whole rows waits behind the first (cora: -15 %); all piece slots where throughput counts
            const uint32_t rounds = c->wide_rounds ? c->wide_rounds : (batch <= 512u ? 1u : 64u);
            const uint32_t cap = (uint32_t)std::min<uint64_t>(pslots, (uint64_t)rounds * ipb);
            while (head < order.size()) {
                Prog g;
                std::vector<Piece> ph;
                std::vector<WJob> js;
                for (uint32_t f = 0; f < std::max(1u, c->wide_phases); f++) {
                    size_t scanned = 0;
                    for (size_t x = head; x < order.size() && ph.size() < cap && scanned < 64; x++) {
                        if (taken[order[x]]) continue;
                        scanned++;
                        const Pack &pk = packs[order[x]];
                        if (ph.size() + pk.n > cap && !(ph.empty() && pk.n <= pslots)) continue;
                        taken[order[x]] = 1;
                        if (!(pieces[pk.first].it.flags & kItemDirect))
                            js.push_back(mkjob(ph.size(), pk.n, pk.kind, f, pk.dst, pk.row));
                        for (uint32_t q = 0; q < pk.n; q++) ph.push_back(slotted(pieces[pk.first + q], ph.size()));
                    }
                    while (head < order.size() && taken[order[head]]) head++;
                    if (ph.empty()) break;
                    close_phase(g, ph, lo);
                    add_pass(g, js);
                }
                if (!g.items.empty()) packed.push_back(std::move(g));
            }
        }
        // descriptors: helpers, finishers, packed
        auto emit = [&](std::vector<Prog> &v) {
            for (Prog &g : v) {
                WideDesc d{};
                d.lo = lo;
                d.index = b0 + k;
                d.kind = 0;
                d.a = (uint32_t)out.items.size();
                d.b = (uint32_t)(g.items.size() / ipb);
                d.c = (uint32_t)out.jobs.size();
                d.d = (uint32_t)g.jobs.size();
                out.items.insert(out.items.end(), g.items.begin(), g.items.end());
                out.jobs.insert(out.jobs.end(), g.jobs.begin(), g.jobs.end());
                out.wide.push_back(d);
                mb_items += (uint32_t)g.items.size();
                p.n_wgs++;
            }
        };
        p.n_helpers += (uint32_t)helpers.size();
        p.n_finishers += (uint32_t)finishers.size();
        p.n_packed += (uint32_t)packed.size();
        // (a unit's helpers come before its finisher -- it waits for their sums; where the whole-row workgroups go is free)
        if (c->wide_order == 2) emit(packed);
        emit(helpers);
        if (c->wide_order == 1) emit(packed);
        emit(finishers);
        if (c->wide_order == 0) emit(packed);
        // the combine trees above the units of rows with more than F*F pieces, level by level
        const uint32_t fin_first = (uint32_t)out.hubs.size();
        uint32_t fin_n = 0;
        for (int level = 0; !cur.empty(); level++) {
            nxt.clear();
            for (const Node &nd : cur) {
                const uint32_t G = (level == kMaxFinLevels - 1) ? nd.n : F;
                const uint32_t nout = (nd.n + G - 1) / G;
                if (nout == 1) {
                    out.hubs.push_back(FinItem{nd.in_slot, nd.n, kFinToStage, nd.row});
                    fin_n++;
                } else {
                    for (uint32_t o = 0; o < nout; o++) {
                        out.hubs.push_back(FinItem{nd.in_slot + o * G, std::min(G, nd.n - o * G), slots + o, nd.row});
                        fin_n++;
                    }
                    nxt.push_back(Node{nd.row, slots, nout});
                    slots += nout;
                }
            }
            cur.swap(nxt);
        }
        while (fin_n % 4u != 0) { out.hubs.push_back(FinItem{0, 0, kFinToStage, 0}); fin_n++; }
        for (uint32_t wn = 0; wn < fin_n / 4u; wn++) {
            WideDesc d{};
            d.lo = lo;
            d.index = b0 + k;
            d.kind = 1;
            d.a = fin_first;
            d.b = fin_n;
            d.c = wn;
            out.wide.push_back(d);
            p.n_wgs++;
            p.n_node_wgs++;
        }
        p.nnz += nnz;
        if (c->count_compulsory) p.compulsory += compulsory_bytes(c, seen, lo, hi, walk, nnz, mb_items);
    }
    p.n_slots = slots;
    return p;
}

// a built plan joins the handle's resident arrays
const WidePlan &wide_plan_append(f2v_ctx *c, uint32_t b0, uint32_t K, uint32_t batch, bool walk, WidePlan p, const WideParts &parts) {
    p.item_off = c->h_items.size();
    p.fin_off = c->h_hubs.size();
    p.wg_off = c->h_wide.size();
    p.job_off = c->h_jobs.size();
    c->h_items.insert(c->h_items.end(), parts.items.begin(), parts.items.end());
    c->h_jobs.insert(c->h_jobs.end(), parts.jobs.begin(), parts.jobs.end());
    c->h_wide.insert(c->h_wide.end(), parts.wide.begin(), parts.wide.end());
    c->h_hubs.insert(c->h_hubs.end(), parts.hubs.begin(), parts.hubs.end());
    // (the helpers' group sums are imported at 32-bit byte offsets: load16_agent)
    if (p.n_slots > kItemSlotMask || (uint64_t)p.n_slots * c->D * sizeof(float) > 0xFFFFFFFFull) c->plan_overflow = true;
    c->max_slots = std::max<size_t>(c->max_slots, p.n_slots);
    return c->wides.emplace(std::make_tuple(b0, K, batch, walk ? 1 : 0), p).first->second;
}

const WidePlan &wide_plan_for(f2v_ctx *c, uint32_t b0, uint32_t K, uint32_t batch, bool walk) {
    auto itp = c->wides.find(std::make_tuple(b0, K, batch, walk ? 1 : 0));
    if (itp != c->wides.end()) return itp->second;
    if (c->h_items.size() > plan_cache_limit(c)) drop_plans(c);
    WideParts parts;
    const WidePlan p = build_wide_plan(c, b0, K, batch, walk, parts, c->seen_scratch);
    return wide_plan_append(c, b0, K, batch, walk, p, parts);
}

// Every wide-form plan of an epoch (launch l covers minibatches [l*K, l*K + K)): the missing ones are built side by side on the
// host's threads -- a plan reads the graph and the handle's parameters only -- and appended in launch order, so that the resident
// arrays are the ones a serial build leaves (F2V_IO_THREADS bounds the threads; a thread that counts compulsory bytes holds 4 N bytes
// of stamps).
void wide_plans_for_epoch(f2v_ctx *c, uint32_t nb, uint32_t K, uint32_t batch, bool walk, unsigned force_threads = 0) {
    std::vector<uint32_t> todo;
    for (uint32_t b0 = 0; b0 < nb; b0 += K)
        if (!c->wides.count(std::make_tuple(b0, std::min(K, nb - b0), batch, walk ? 1 : 0))) todo.push_back(b0);
    if (todo.empty()) return;
    if (c->h_items.size() > plan_cache_limit(c)) {
        drop_plans(c);
        todo.clear();
        for (uint32_t b0 = 0; b0 < nb; b0 += K) todo.push_back(b0);
    }
    unsigned T = std::thread::hardware_concurrency();
    if (const char *e = getenv("F2V_IO_THREADS")) T = (unsigned)atoi(e);
    T = std::max(1u, std::min<unsigned>({T, 32u, (unsigned)todo.size()}));
    if (c->count_compulsory) T = std::min<unsigned>(T, std::max<unsigned>(1u, (unsigned)((1ull << 30) / std::max<uint64_t>(4ull * c->n, 1))));  // <= 1 GiB of stamps
    if (force_threads) T = std::max(2u, std::min<unsigned>(force_threads, (unsigned)todo.size()));  // (the self-test hook: the threaded path on any graph)
    if (T == 1 || (!force_threads && (uint64_t)c->nnz < (1ull << 22))) {  // small graphs: a thread costs more than the plan
        for (uint32_t b0 : todo) (void)wide_plan_for(c, b0, std::min(K, nb - b0), batch, walk);
        return;
    }
    const bool timing = getenv("F2V_PLAN_TIMING") != nullptr;  // (measurements: where the first call of a batch size spends its host time)
    const auto t_0 = std::chrono::steady_clock::now();
    std::vector<WideParts> parts(todo.size());
    std::vector<WidePlan> plans(todo.size());
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++)
        th.emplace_back([&] {
            SeenScratch seen;
            for (size_t k; (k = next.fetch_add(1)) < todo.size();)
                plans[k] = build_wide_plan(c, todo[k], std::min(K, nb - todo[k]), batch, walk, parts[k], seen);
        });
    for (auto &y : th) y.join();
    const auto t_1 = std::chrono::steady_clock::now();
    // the resident arrays get their final size once (no regrowth copies), the parts are moved in by the threads -- side by side: every
    // part has its place (prefix sums of the sizes), and a part is released as soon as it has been copied
    std::vector<size_t> at_items(todo.size() + 1, c->h_items.size()), at_jobs(todo.size() + 1, c->h_jobs.size()), at_wide(todo.size() + 1, c->h_wide.size()),
        at_hubs(todo.size() + 1, c->h_hubs.size());
    for (size_t k = 0; k < todo.size(); k++) {
        at_items[k + 1] = at_items[k] + parts[k].items.size();
        at_jobs[k + 1] = at_jobs[k] + parts[k].jobs.size();
        at_wide[k + 1] = at_wide[k] + parts[k].wide.size();
        at_hubs[k + 1] = at_hubs[k] + parts[k].hubs.size();
    }
    c->h_items.resize(at_items.back());
    c->h_jobs.resize(at_jobs.back());
    c->h_wide.resize(at_wide.back());
    c->h_hubs.resize(at_hubs.back());
    const auto t_2 = std::chrono::steady_clock::now();
    next = 0;
    th.clear();
    for (unsigned t = 0; t < T; t++)
        th.emplace_back([&] {
            for (size_t k; (k = next.fetch_add(1)) < todo.size();) {
                WideParts &q = parts[k];
                if (!q.items.empty()) memcpy(c->h_items.data() + at_items[k], q.items.data(), q.items.size() * sizeof(Item));
                if (!q.jobs.empty()) memcpy(c->h_jobs.data() + at_jobs[k], q.jobs.data(), q.jobs.size() * sizeof(WJob));
                if (!q.wide.empty()) memcpy(c->h_wide.data() + at_wide[k], q.wide.data(), q.wide.size() * sizeof(WideDesc));
                if (!q.hubs.empty()) memcpy(c->h_hubs.data() + at_hubs[k], q.hubs.data(), q.hubs.size() * sizeof(FinItem));
                q = WideParts{};
            }
        });
    for (auto &y : th) y.join();
    for (size_t k = 0; k < todo.size(); k++) {
        WidePlan &p = plans[k];
        p.item_off = at_items[k];
        p.job_off = at_jobs[k];
        p.wg_off = at_wide[k];
        p.fin_off = at_hubs[k];
        if (p.n_slots > kItemSlotMask || (uint64_t)p.n_slots * c->D * sizeof(float) > 0xFFFFFFFFull) c->plan_overflow = true;
        c->max_slots = std::max<size_t>(c->max_slots, p.n_slots);
        c->wides.emplace(std::make_tuple(todo[k], std::min(K, nb - todo[k]), batch, walk ? 1 : 0), p);
    }
    if (timing) {
        const auto t_3 = std::chrono::steady_clock::now();
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "f2v plan timing: %zu plans on %u threads: built %.0f ms, resident arrays sized %.0f ms, parts moved in %.0f ms (%.1f MB of items)\n", todo.size(), T, ms(t_0, t_1), ms(t_1, t_2),
                ms(t_2, t_3), c->h_items.size() * sizeof(Item) / 1e6);
    }
}

// Make every plan built so far resident in HBM (and the partial-sum buffer large enough).
int upload_plans(f2v_ctx *c) {
    if (c->plan_overflow) {
        c->plan_overflow = false;
        drop_plans(c);
        return fail(F2V_EINVAL, "a launch plan needs more than 2^28 partial-sum slots (or, in the wide form, 4 GiB of them): use a larger \"hub_chunk\" or fewer \"chain_rows\" / \"wide_rows\"");
    }
    const size_t need_slots = c->max_slots;
    const bool grow_items = c->h_items.size() > c->d_items_cap, grow_hubs = c->h_hubs.size() > c->d_hubs_cap;
    const bool grow_slots = need_slots > c->partial_slots;
    if (c->h_items.size() == c->d_items_valid && c->h_hubs.size() == c->d_hubs_valid && c->h_wg.size() == c->d_wg_valid &&
        c->h_wide.size() == c->d_wide_valid && c->h_jobs.size() == c->d_jobs_valid && !grow_slots)
        return F2V_OK;  // O(1) steady state
    HIPC(hipStreamSynchronize(c->stream));  // launches in flight read these buffers
    const auto t_up0 = std::chrono::steady_clock::now();
    if (c->h_wide.size() > c->d_wide_cap) {
        if (c->d_wide) (void)hipFree(c->d_wide);
        c->d_wide = nullptr;
        c->d_wide_cap = std::max<size_t>(c->h_wide.size() * 3 / 2, 1024);
        HIPC(hipMalloc((void **)&c->d_wide, c->d_wide_cap * sizeof(WideDesc)));
        c->d_wide_valid = 0;
    }
    if (c->h_wide.size() > c->d_wide_valid)
        HIPC(hipMemcpy(c->d_wide + c->d_wide_valid, c->h_wide.data() + c->d_wide_valid, (c->h_wide.size() - c->d_wide_valid) * sizeof(WideDesc), hipMemcpyHostToDevice));
    c->d_wide_valid = c->h_wide.size();
    if (c->h_jobs.size() > c->d_jobs_cap) {
        if (c->d_jobs) (void)hipFree(c->d_jobs);
        c->d_jobs = nullptr;
        c->d_jobs_cap = std::max<size_t>(c->h_jobs.size() * 3 / 2, 1024);
        HIPC(hipMalloc((void **)&c->d_jobs, c->d_jobs_cap * sizeof(WJob)));
        c->d_jobs_valid = 0;
    }
    if (c->h_jobs.size() > c->d_jobs_valid)
        HIPC(hipMemcpy(c->d_jobs + c->d_jobs_valid, c->h_jobs.data() + c->d_jobs_valid, (c->h_jobs.size() - c->d_jobs_valid) * sizeof(WJob), hipMemcpyHostToDevice));
    c->d_jobs_valid = c->h_jobs.size();
    if (c->h_wg.size() > c->d_wg_cap) {
        if (c->d_wg) (void)hipFree(c->d_wg);
        c->d_wg = nullptr;
        c->d_wg_cap = std::max<size_t>(c->h_wg.size() * 3 / 2, 1024);
        HIPC(hipMalloc((void **)&c->d_wg, c->d_wg_cap * sizeof(WgDesc)));
        c->d_wg_valid = 0;
    }
    if (c->h_wg.size() > c->d_wg_valid)
        HIPC(hipMemcpy(c->d_wg + c->d_wg_valid, c->h_wg.data() + c->d_wg_valid, (c->h_wg.size() - c->d_wg_valid) * sizeof(WgDesc), hipMemcpyHostToDevice));
    c->d_wg_valid = c->h_wg.size();
    if (grow_items) {
        if (c->d_items) (void)hipFree(c->d_items);
        c->d_items = nullptr;
        c->d_items_cap = std::max<size_t>(c->h_items.size() * 3 / 2, 1024);
        HIPC(hipMalloc((void **)&c->d_items, c->d_items_cap * sizeof(Item)));
        c->d_items_valid = 0;
    }
    if (grow_hubs) {
        if (c->d_hubs) (void)hipFree(c->d_hubs);
        c->d_hubs = nullptr;
        c->d_hubs_cap = std::max<size_t>(c->h_hubs.size() * 3 / 2, 256);
        HIPC(hipMalloc((void **)&c->d_hubs, c->d_hubs_cap * sizeof(FinItem)));
        c->d_hubs_valid = 0;
    }
    if (grow_slots) {
        if (c->d_partials) (void)hipFree(c->d_partials);
        c->d_partials = nullptr;
        HIPC(hipMalloc((void **)&c->d_partials, need_slots * (size_t)c->D * sizeof(float)));
        if (c->d_ready) (void)hipFree(c->d_ready);
        c->d_ready = nullptr;
        HIPC(hipMalloc((void **)&c->d_ready, need_slots * sizeof(uint32_t)));
        // hipMemset on device memory may return before the fill has run (it is ordered on the NULL stream, which this
        // engine's non-blocking stream does not wait for): a late fill would wipe the first launch's flags
        HIPC(hipMemsetAsync(c->d_ready, 0, need_slots * sizeof(uint32_t), c->stream));
        HIPC(hipStreamSynchronize(c->stream));
        c->partial_slots = need_slots;
    }
    if (c->h_items.size() > c->d_items_valid)
        HIPC(hipMemcpy(c->d_items + c->d_items_valid, c->h_items.data() + c->d_items_valid,
                       (c->h_items.size() - c->d_items_valid) * sizeof(Item), hipMemcpyHostToDevice));
    if (c->h_hubs.size() > c->d_hubs_valid)
        HIPC(hipMemcpy(c->d_hubs + c->d_hubs_valid, c->h_hubs.data() + c->d_hubs_valid,
                       (c->h_hubs.size() - c->d_hubs_valid) * sizeof(FinItem), hipMemcpyHostToDevice));
    c->d_items_valid = c->h_items.size();
    c->d_hubs_valid = c->h_hubs.size();
    if (getenv("F2V_PLAN_TIMING"))
        fprintf(stderr, "f2v plan timing: upload %.0f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_up0).count());
    return F2V_OK;
}

int reserve_ids(f2v_ctx *c, size_t count) {
    if (count <= c->ids_cap) return F2V_OK;
    HIPC(hipStreamSynchronize(c->stream));
    if (c->d_ids) (void)hipFree(c->d_ids);
    c->d_ids = nullptr;
    HIPC(hipMalloc((void **)&c->d_ids, count * sizeof(uint32_t)));
    c->ids_cap = count;
    return F2V_OK;
}

template <int VEC, bool EXACT>
void launch_commit_t(f2v_ctx *c, uint32_t lo, uint32_t rows) {
    const int wpb = 4;
    const uint32_t blocks = std::min<uint32_t>((rows + wpb - 1) / wpb, 4096u);
    hipLaunchKernelGGL((commit_kernel<VEC, EXACT>), dim3(blocks), dim3(64 * wpb), 0, c->stream, c->d_X[c->cur], c->d_X[c->cur ^ 1], lo, rows, c->D);
}

template <typename F>
int dispatch_layout(f2v_ctx *c, F &&f) {
    // (VEC, EXACT) instantiations: exact vector loads when D == 64*VEC, guarded scalar loads otherwise
    if (c->exact) {
        switch (c->vec) {
            case 1: f(std::integral_constant<int, 1>{}, std::true_type{}); return F2V_OK;
            case 2: f(std::integral_constant<int, 2>{}, std::true_type{}); return F2V_OK;
            case 4: f(std::integral_constant<int, 4>{}, std::true_type{}); return F2V_OK;
            case 8: f(std::integral_constant<int, 8>{}, std::true_type{}); return F2V_OK;
        }
    } else {
        switch (c->vec) {
            case 1: f(std::integral_constant<int, 1>{}, std::false_type{}); return F2V_OK;
            case 2: f(std::integral_constant<int, 2>{}, std::false_type{}); return F2V_OK;
            case 4: f(std::integral_constant<int, 4>{}, std::false_type{}); return F2V_OK;
            case 8: f(std::integral_constant<int, 8>{}, std::false_type{}); return F2V_OK;
        }
    }
    return fail(F2V_EINVAL, "unsupported dimension %u", c->D);
}

// Make d_X[cur] the whole, up-to-date matrix: a completed epoch (all N rows updated) just swaps the two
// matrices; a partial range is folded back with a copy.
int flush_pending(f2v_ctx *c) {
    c->pending = false;
    if (c->upd_hi == c->upd_lo) return F2V_OK;
    if (c->upd_lo == 0 && c->upd_hi == c->n) {
        c->cur ^= 1;
    } else {
        const uint32_t lo = c->upd_lo, rows = c->upd_hi - c->upd_lo;
        int rc = dispatch_layout(c, [&](auto V, auto E) { launch_commit_t<decltype(V)::value, decltype(E)::value>(c, lo, rows); });
        if (rc != F2V_OK) return rc;
        HIPC(hipGetLastError());
    }
    c->upd_lo = c->upd_hi = 0;
    return F2V_OK;
}

// A combine-tree node's wait timed out (the words of d_kerr are in `e`): nodes that give up store nothing, so the rows
// of the affected hub vertices were not updated from that launch on -- the embeddings are invalid and must be set again.
// The handle stays usable: the error words are cleared and it runs one launch per tree level from now on (no in-grid waits).
int kernel_gave_up(f2v_ctx *c, const char *where, const uint32_t *e) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemsetAsync(c->d_kerr, 0, 64, c->stream);
    (void)hipStreamSynchronize(c->stream);
    c->merge_fin = false;  // (chained minibatches need it: off with it)
    c->pending = false;
    c->upd_lo = c->upd_hi = 0;
    c->have_x = false;
    c->x_invalid = true;
    if (e[0] == 4u)  // a finisher's import: err[2] waiting workgroup, [3] partial-sum slot, [4] flag seen, [5] launch, [6] grid, [7] the minibatch's first row
        return fail(F2V_ESTATE, "%s: a chained launch was lost: %u waits for a helper's group sum gave up (first: workgroup %u of %u, minibatch starting at row %u, "
                    "partial-sum slot %u: flag %u, launch %u); the embeddings are invalid from that launch on (set or initialise them again); this handle now runs one "
                    "launch per minibatch and per combine-tree level (\"merge_finalize\" = 0)",
                    where, e[1], e[2], e[6], e[7], e[3], e[4], e[5]);
    if (e[0] == 3u)  // wait_row_slow: err[2] waiting workgroup, [3] row, [4] flag seen, [5] launch, [6] grid, [7] the minibatch's first row
        return fail(F2V_ESTATE, "%s: a chained launch was lost: %u row waits gave up (first: workgroup %u of %u, minibatch starting at row %u, waiting for row %u: "
                    "flag %u, launch %u); the embeddings are invalid from that launch on (set or initialise them again); this handle now runs one launch "
                    "per minibatch and per combine-tree level (\"merge_finalize\" = 0)",
                    where, e[1], e[2], e[6], e[7], e[3], e[4], e[5]);
    return fail(F2V_ESTATE, "%s: %u combine-tree waits gave up (first: node %u of %u [first dependent %u] on slot %u, flag %u, launch %u); "
                "the embeddings are invalid from that minibatch on (set or initialise them again); this handle now runs with \"merge_finalize\" = 0",
                where, e[1], e[2], e[6], e[7], e[3], e[4], e[5]);
}

// after a stream synchronisation: did a kernel give up a bounded wait?
int check_kernel_err(f2v_ctx *c, const char *where) {
    uint32_t e[8] = {};
    HIPC(hipMemcpy(e, c->d_kerr, sizeof e, hipMemcpyDeviceToHost));
    if (e[0]) return kernel_gave_up(c, where, e);
    return F2V_OK;
}

int math_of_option(int option) {
    switch (option) {
        case 5: case 8: case 11: return 5;
        case 6: case 9: return 6;
        case 7: case 10: return 7;
        default: return 0;
    }
}

// Launch one minibatch step (+ hub finalisation) on the handle's stream.  d_ids: device sample ids.
void fill_targets(const f2v_ctx *c, PushTargets &t, int which, uint32_t batch_lo, const uint32_t *d_masks);

// bound of a combine-tree node's wait in ticks of the 100 MHz wall clock ("tree_timeout_ms", default 5 s; the
// environment variable F2V_TREE_TIMEOUT_MS sets the default of new handles)
unsigned long long tree_timeout_ticks(const f2v_ctx *c) { return (unsigned long long)c->tree_timeout_ms * 100000ull; }

// push_masks / push: a sharded run's step -- the kernels also store every finished row into the second matrix of
// the peers that read it (push_masks == nullptr: of every peer).
int launch_step(f2v_ctx *c, int math, uint32_t batch_lo, uint32_t batch_hi, uint32_t row_lo, uint32_t row_hi,
                const uint32_t *d_ids, uint32_t ns, float lr, int bs_mode, bool push = false, const uint32_t *push_masks = nullptr) {
    int rc;
    if (c->upd_hi != c->upd_lo && batch_lo != c->upd_hi) {
        // not the continuation of the updated range (a new epoch, or batches out of order): swap / fold first
        if ((rc = flush_pending(c)) != F2V_OK) return rc;
    }
    const bool walk = (math == 7);
    const Plan plan = plan_for(c, row_lo, row_hi, walk);  // by value: upload_plans may not move it, but keep it simple
    if ((rc = upload_plans(c)) != F2V_OK) return rc;
    StepArgs a{};
    a.X = c->d_X[c->cur];
    a.Xn = c->d_X[c->cur ^ 1];
    a.rowptr = c->d_rowptr;
    a.nbr_ids = walk ? c->d_walks : c->d_colids;
    a.partials = c->d_partials;
    a.sample_ids = d_ids;
    a.items = c->d_items + plan.item_off;
    a.sm_table = c->d_table;
    a.D = c->D;
    a.batch_lo = batch_lo;
    a.n_items = plan.n_items;
    a.upd_lo = c->upd_lo;
    a.upd_rows = c->upd_hi - c->upd_lo;
    a.ns = ns;
    a.bs_mode = bs_mode ? 1u : 0u;
    a.unit_degi = c->unit_degi ? 1u : 0u;
    a.lr = lr;
#ifdef F2V_TEST_HOOKS
    a.xcd_times = c->d_xcd;
#endif
    push = push && c->push.attached && c->push.world > 1;
    if (push) fill_targets(c, a.push, c->cur ^ 1, batch_lo, push_masks);

    const uint32_t wpb = (uint32_t)c->waves_per_block;
    // sub-wave layout when D is a multiple of 4 up to 256: 16, 8 or 4 work items per wavefront
    const uint32_t width = subwave_width(c);
    const bool quarter = width != 0, full = width == c->D;
    const uint32_t per_wave = !quarter ? 1u : (width == 16 ? 16u : width == 32 ? 8u : 4u);
    const uint32_t waves = (plan.n_items + per_wave - 1) / per_wave;
    uint32_t blocks = (waves + wpb - 1) / wpb;  // 0 when this rank has no row of the batch
    a.step_blocks = blocks;
    // sub-wave kernel: the combine trees ride in the same grid (one launch per minibatch)
    const bool fused_tree = quarter && c->merge_fin && !c->capturing && plan.n_levels >= 1 && blocks > 0;
    if (fused_tree) {
        a.fin_items = c->d_hubs + plan.fin_off[0];
        a.fin_n = 0;
        for (int lev = 0; lev < plan.n_levels; lev++) a.fin_n += plan.fin_cnt[lev];  // the levels are stored back to back
        a.ready = c->d_ready;
        a.err = c->d_kerr;
        a.timeout_ticks = tree_timeout_ticks(c);
        a.seq = ++c->launch_seq;
        if (a.seq == 0) a.seq = ++c->launch_seq;  // 0 is what fresh flags hold
#ifdef F2V_TEST_HOOKS
        a.test_withhold_slot = c->test_withhold_slot;
        a.test_withhold_row = kNoSlot;  // (chained launches only)
#endif
        blocks += (a.fin_n + wpb - 1) / wpb;
    }
    if (blocks == 0) {
        // nothing to compute here; the range bookkeeping below still advances
    } else if (quarter) {
#define F2V_Q2(OPT, LPI, NB, U, PUSH, FULL) hipLaunchKernelGGL((qstep_kernel<OPT, LPI, NB, U, PUSH, FULL>), dim3(blocks), dim3(64 * wpb), 0, c->stream, a)
#define F2V_Q(OPT, LPI, NB, U)                                                              \
    do {                                                                                    \
        if (push) { if (full) F2V_Q2(OPT, LPI, NB, U, true, true); else F2V_Q2(OPT, LPI, NB, U, true, false); }    \
        else { if (full) F2V_Q2(OPT, LPI, NB, U, false, true); else F2V_Q2(OPT, LPI, NB, U, false, false); }       \
    } while (0)
        // rows in flight per item: 4 at D = 128 (90 VGPRs, 5 waves/SIMD; 8 is selectable and measured 3-9 % slower
        // on RMAT-20: 116 VGPRs, 4 waves/SIMD) and at D = 256; 8 where a row is a single dwordx4 per lane (D <= 64)
        const bool u8 = (c->rows_in_flight == 8);
        const int o = (math == 5) ? 5 : 6;
        switch (width) {
            case 16: if (o == 5) F2V_Q(5, 4, 1, 8); else F2V_Q(6, 4, 1, 8); break;
            case 32: if (o == 5) F2V_Q(5, 8, 1, 8); else F2V_Q(6, 8, 1, 8); break;
            case 64: if (o == 5) F2V_Q(5, 16, 1, 8); else F2V_Q(6, 16, 1, 8); break;
            case 128:
                if (u8) { if (o == 5) F2V_Q(5, 16, 2, 8); else F2V_Q(6, 16, 2, 8); }
                else { if (o == 5) F2V_Q(5, 16, 2, 4); else F2V_Q(6, 16, 2, 4); }
                break;
            default: if (o == 5) F2V_Q(5, 16, 4, 4); else F2V_Q(6, 16, 4, 4); break;
        }
#undef F2V_Q
#undef F2V_Q2
    } else {
        rc = dispatch_layout(c, [&](auto V, auto E) {
            constexpr int VEC = decltype(V)::value;
            constexpr bool EX = decltype(E)::value;
            if (math == 5)
                hipLaunchKernelGGL((step_kernel<5, VEC, EX>), dim3(blocks), dim3(64 * wpb), 0, c->stream, a);
            else
                hipLaunchKernelGGL((step_kernel<6, VEC, EX>), dim3(blocks), dim3(64 * wpb), 0, c->stream, a);
        });
        if (rc != F2V_OK) return rc;
    }
    HIPC(hipGetLastError());
    const bool tree = !fused_tree && c->merge_fin && !c->capturing && plan.n_levels >= 2;
    if (tree) {
        FinalizeTreeArgs t{};
        t.f.X = a.X;
        t.f.partials = c->d_partials;
        t.f.Xn = a.Xn;
        t.f.items = c->d_hubs + plan.fin_off[0];
        t.f.n_items = 0;
        for (int lev = 0; lev < plan.n_levels; lev++) t.f.n_items += plan.fin_cnt[lev];  // the levels are stored back to back
        t.f.D = c->D;
        t.f.push = a.push;
        t.ready = c->d_ready;
        t.err = c->d_kerr;
        t.timeout_ticks = tree_timeout_ticks(c);
        t.seq = ++c->launch_seq;
        if (t.seq == 0) t.seq = ++c->launch_seq;  // 0 is what fresh flags hold
        t.first_dep = plan.fin_cnt[0];
        const uint32_t fb = (t.f.n_items + wpb - 1) / wpb;  // the node layout counts on `wpb` nodes per workgroup
        rc = dispatch_layout(c, [&](auto V, auto E) {
            constexpr int VEC = decltype(V)::value;
            constexpr bool EX = decltype(E)::value;
            if (math == 5)
                hipLaunchKernelGGL((hub_finalize_tree_kernel<5, VEC, EX>), dim3(fb), dim3(64 * wpb), 0, c->stream, t);
            else
                hipLaunchKernelGGL((hub_finalize_tree_kernel<6, VEC, EX>), dim3(fb), dim3(64 * wpb), 0, c->stream, t);
        });
        if (rc != F2V_OK) return rc;
        HIPC(hipGetLastError());
    }
    for (int lev = 0; lev < plan.n_levels && !tree && !fused_tree; lev++) {
        FinalizeArgs f{};
        f.X = a.X;
        f.partials = c->d_partials;
        f.Xn = a.Xn;
        f.items = c->d_hubs + plan.fin_off[lev];
        f.n_items = plan.fin_cnt[lev];
        f.D = c->D;
        f.push = a.push;
        const uint32_t fb = (f.n_items + 3) / 4;
        rc = dispatch_layout(c, [&](auto V, auto E) {
            constexpr int VEC = decltype(V)::value;
            constexpr bool EX = decltype(E)::value;
            if (math == 5)
                hipLaunchKernelGGL((hub_finalize_kernel<5, VEC, EX>), dim3(fb), dim3(256), 0, c->stream, f);
            else
                hipLaunchKernelGGL((hub_finalize_kernel<6, VEC, EX>), dim3(fb), dim3(256), 0, c->stream, f);
        });
        if (rc != F2V_OK) return rc;
        HIPC(hipGetLastError());
    }
    c->stats.hub_rows += plan.n_hubs;
    c->stats.hub_chunks += plan.n_chunks;
    if (c->upd_hi == c->upd_lo) c->upd_lo = batch_lo;
    c->upd_hi = batch_hi;
    c->pending = true;
    c->p_lo = batch_lo;
    c->p_hi = batch_hi;

    // statistics: algorithmic bytes of SURVEY 8d -- nnz*(4D+4) + rows*(8D+4) + ns*(4D+4) per minibatch
    const uint64_t rows = row_hi - row_lo;
    c->stats.step_launches += 1;
    c->stats.rows += rows;
    c->stats.nnz += plan.nnz;
    c->stats.algorithmic_bytes += plan.nnz * (4ull * c->D + 4) + rows * (8ull * c->D + 4) + (uint64_t)ns * (4ull * c->D + 4);
    c->stats.compulsory_bytes += plan.compulsory;
    return F2V_OK;
}

// One chained launch: minibatches [plan.first_batch, +plan.n_batches) of an epoch whose sample ids (ids_stride per minibatch)
// lie at d_ids_epoch.
int launch_chain(f2v_ctx *c, int math, const ChainPlan &plan, const uint32_t *d_ids_epoch, uint32_t ids_stride, uint32_t ns, float lr, int bs_mode) {
    int rc;
    if (c->upd_hi != c->upd_lo && plan.lo != c->upd_hi) {
        if ((rc = flush_pending(c)) != F2V_OK) return rc;
    }
    ChainArgs ca{};
    StepArgs &a = ca.base;
    a.X = c->d_X[c->cur];
    a.Xn = c->d_X[c->cur ^ 1];
    a.rowptr = c->d_rowptr;
    a.nbr_ids = math == 7 ? c->d_walks : c->d_colids;
    a.partials = c->d_partials;
    a.items = c->d_items + plan.item_off;
    a.sm_table = c->d_table;
    a.D = c->D;
    a.upd_lo = (c->upd_hi == c->upd_lo) ? plan.lo : c->upd_lo;
    a.ns = ns;
    a.bs_mode = bs_mode ? 1u : 0u;
    a.unit_degi = c->unit_degi ? 1u : 0u;
    a.lr = lr;
    a.fin_items = c->d_hubs + plan.fin_off;
    a.ready = c->d_ready;
    a.err = c->d_kerr;
    a.timeout_ticks = (unsigned long long)std::min(c->tree_timeout_ms, c->chain_timeout_ms) * 100000ull;
    a.seq = ++c->launch_seq;
    if (a.seq == 0) a.seq = ++c->launch_seq;
#ifdef F2V_TEST_HOOKS
    a.test_withhold_slot = c->test_withhold_slot;
    a.test_withhold_row = c->test_withhold_row;
#endif
    a.rowflag = c->d_rowflag;
    a.chain_lo = plan.lo;
#ifdef F2V_TEST_HOOKS
    if (c->test_chain_nowait) a.chain_lo = 0xFFFFFFFFu;  // the kernel then treats no row as "written by an earlier minibatch"
    a.stamps = c->d_stamps;
#endif
    ca.wg = c->d_wg + plan.wg_off;
    ca.ids = d_ids_epoch;
    ca.ids_stride = ids_stride;
    const uint32_t width = subwave_width(c), wpb = (uint32_t)c->waves_per_block;
    const bool full = width == c->D;
    const int o = (math == 5) ? 5 : 6;
#define F2V_C2(OPT, LPI, NB, U, FULL) hipLaunchKernelGGL((qstep_chain_kernel<OPT, LPI, NB, U, FULL>), dim3(plan.n_wgs), dim3(64 * wpb), 0, c->stream, ca)
#define F2V_C(OPT, LPI, NB, U) do { if (full) F2V_C2(OPT, LPI, NB, U, true); else F2V_C2(OPT, LPI, NB, U, false); } while (0)
    switch (width) {
        case 32: if (o == 5) F2V_C(5, 8, 1, 8); else F2V_C(6, 8, 1, 8); break;
        case 64: if (o == 5) F2V_C(5, 16, 1, 8); else F2V_C(6, 16, 1, 8); break;
        case 128:
            if (c->rows_in_flight == 8) { if (o == 5) F2V_C(5, 16, 2, 8); else F2V_C(6, 16, 2, 8); }
            else { if (o == 5) F2V_C(5, 16, 2, 4); else F2V_C(6, 16, 2, 4); }
            break;
        default: if (o == 5) F2V_C(5, 16, 4, 4); else F2V_C(6, 16, 4, 4); break;
    }
#undef F2V_C
#undef F2V_C2
    HIPC(hipGetLastError());
    c->last_train_form = 1;
    if (c->upd_hi == c->upd_lo) c->upd_lo = plan.lo;
    c->upd_hi = plan.hi;
    c->pending = true;
    c->p_lo = plan.last_lo;
    c->p_hi = plan.hi;
    c->stats.hub_rows += plan.n_hubs;
    c->stats.hub_chunks += plan.n_chunks;
    c->stats.step_launches += 1;
    c->stats.rows += plan.hi - plan.lo;
    c->stats.nnz += plan.nnz;
    c->stats.algorithmic_bytes += plan.nnz * (4ull * c->D + 4) + (uint64_t)(plan.hi - plan.lo) * (8ull * c->D + 4) + (uint64_t)plan.n_batches * ns * (4ull * c->D + 4);
    c->stats.compulsory_bytes += plan.compulsory;
    return F2V_OK;
}


// One launch of the wide form (qwide_chain_kernel): minibatches [plan.first_batch, +plan.n_batches)
// `epochs` > 1 (the plan covers the whole graph): that many epochs chained in the one launch -- the current matrix is copied into a ring
// of epochs + 1 matrices, epoch e reads matrix e and writes matrix e + 1 behind its own row flags, and the last one is copied back
// (two 1-MB copies per launch on a graph like cora, where the launch boundary they replace is a third of an epoch).
int launch_wide(f2v_ctx *c, int math, const WidePlan &plan, const uint32_t *d_ids_epoch, uint32_t ids_stride, uint32_t ns, float lr, int bs_mode,
                uint32_t *epochs_io = nullptr, uint64_t ids_epoch_stride = 0) {
    int rc;
    uint32_t epochs = epochs_io ? *epochs_io : 1u;
    if (epochs > 1 && (c->ring_epochs < epochs || c->ring_slots < std::max<size_t>(plan.n_slots, 1))) {
        // the ring of matrices (and the per-epoch flags and partial sums): where the device has no room for it, one epoch per launch
        HIPC(hipStreamSynchronize(c->stream));
        for (void *p : {(void *)c->d_ring, (void *)c->d_ring_partials, (void *)c->d_ring_flags, (void *)c->d_ring_ready})
            if (p) (void)hipFree(p);
        c->d_ring = c->d_ring_partials = nullptr;
        c->d_ring_flags = c->d_ring_ready = nullptr;
        const uint32_t cap = std::max(epochs, c->ring_epochs);
        const size_t slots = std::max<size_t>(std::max<size_t>(plan.n_slots, c->ring_slots), 1), mat = (size_t)c->n * c->D;
        c->ring_epochs = 0;
        c->ring_slots = 0;
        bool refuse = false;
#ifdef F2V_TEST_HOOKS
        refuse = getenv("F2V_TEST_RING_REFUSE") != nullptr;  // fault injection: as if the device had no room for the ring
#endif
        if (refuse || hipMalloc((void **)&c->d_ring, (size_t)(cap + 1) * mat * sizeof(float)) != hipSuccess ||
            hipMalloc((void **)&c->d_ring_partials, (size_t)cap * slots * c->D * sizeof(float)) != hipSuccess ||
            hipMalloc((void **)&c->d_ring_flags, (size_t)cap * c->n * sizeof(uint32_t)) != hipSuccess ||
            hipMalloc((void **)&c->d_ring_ready, (size_t)cap * slots * sizeof(uint32_t)) != hipSuccess) {
            (void)hipGetLastError();
            for (void *p : {(void *)c->d_ring, (void *)c->d_ring_partials, (void *)c->d_ring_flags, (void *)c->d_ring_ready})
                if (p) (void)hipFree(p);
            c->d_ring = c->d_ring_partials = nullptr;
            c->d_ring_flags = c->d_ring_ready = nullptr;
            c->ring_refused = true;  // (f2v_train stops asking)
            epochs = 1;
        } else {
            HIPC(hipMemsetAsync(c->d_ring_flags, 0, (size_t)cap * c->n * sizeof(uint32_t), c->stream));  // 0 is no launch's sequence number
            HIPC(hipMemsetAsync(c->d_ring_ready, 0, (size_t)cap * slots * sizeof(uint32_t), c->stream));
            c->ring_epochs = cap;
            c->ring_slots = slots;
        }
    }
    if (epochs_io) *epochs_io = epochs;
    const bool ring = epochs > 1;
    if (ring || (c->upd_hi != c->upd_lo && plan.lo != c->upd_hi)) {
        if ((rc = flush_pending(c)) != F2V_OK) return rc;
    }
    const size_t matrix = (size_t)c->n * c->D;
    if (ring) {
        if (plan.lo != 0 || plan.hi != c->n || plan.n_node_wgs != 0) return fail(F2V_ESTATE, "launch_wide: epochs can only be chained where one launch covers the graph");
        HIPC(hipMemcpyAsync(c->d_ring, c->d_X[c->cur], matrix * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    }
    WideArgs wa{};
    StepArgs &a = wa.base;
    a.X = ring ? c->d_ring : c->d_X[c->cur];
    a.Xn = ring ? c->d_ring + matrix : c->d_X[c->cur ^ 1];
    a.rowptr = c->d_rowptr;
    a.nbr_ids = math == 7 ? c->d_walks : c->d_colids;
    a.partials = ring ? c->d_ring_partials : c->d_partials;
    a.items = c->d_items + plan.item_off;
    a.sm_table = c->d_table;
    a.D = c->D;
    a.upd_lo = (c->upd_hi == c->upd_lo) ? plan.lo : c->upd_lo;
    a.ns = ns;
    a.bs_mode = bs_mode ? 1u : 0u;
    a.unit_degi = c->unit_degi ? 1u : 0u;
    a.lr = lr;
    a.fin_items = c->d_hubs + plan.fin_off;
    a.ready = ring ? c->d_ring_ready : c->d_ready;
    a.err = c->d_kerr;
    a.timeout_ticks = (unsigned long long)std::min(c->tree_timeout_ms, c->chain_timeout_ms) * 100000ull;
    a.seq = ++c->launch_seq;
    if (a.seq == 0) a.seq = ++c->launch_seq;
#ifdef F2V_TEST_HOOKS
    a.test_withhold_slot = c->test_withhold_slot;
    a.test_withhold_row = c->test_withhold_row;
#endif
    a.rowflag = ring ? c->d_ring_flags : c->d_rowflag;
    a.chain_lo = plan.lo;
    if (ring) {
        wa.wgs_per_epoch = plan.n_wgs;
        wa.n_rows = c->n;
        wa.slots_per_epoch = (uint32_t)c->ring_slots;
        wa.ring_stride = matrix;
        wa.ids_epoch_stride = ids_epoch_stride;
    }
#ifdef F2V_TEST_HOOKS
    a.test_nowait = c->test_chain_mode;
    a.stamps = c->d_stamps;
#endif
    wa.wg = c->d_wide + plan.wg_off;
    wa.jobs = c->d_jobs + plan.job_off;
    wa.ids = d_ids_epoch;
    wa.ids_stride = ids_stride;
    const uint32_t width = plan.width;
    const bool full = width == c->D;
    const int o = (math == 5) ? 5 : 6;
    // "wide_samples_early" (-1 = automatic): on for graphs of up to 2 M nonzeros, whose launches are one dependency chain
    const bool early = ring || (c->wide_samples_early >= 0 ? c->wide_samples_early != 0 : c->nnz <= (2ull << 20));
    c->last_wide_early = early;
    c->last_wide_width = width;
    c->last_train_form = 2;
    c->last_wide_epochs = std::max(c->last_wide_epochs, epochs);  // (the most epochs one launch of this f2v_train has carried)
    const uint32_t grid = plan.n_wgs * epochs;
#define F2V_W3(OPT, LPI, NB, U, FULL, MODE) hipLaunchKernelGGL((qwide_chain_kernel<OPT, LPI, NB, U, FULL, MODE>), dim3(grid), dim3(256), 0, c->stream, wa)
#define F2V_W2(OPT, LPI, NB, U, FULL) do { if (ring) F2V_W3(OPT, LPI, NB, U, FULL, 2); else if (early) F2V_W3(OPT, LPI, NB, U, FULL, 1); else F2V_W3(OPT, LPI, NB, U, FULL, 0); } while (0)
#define F2V_W(OPT, LPI, NB, U) do { if (full) F2V_W2(OPT, LPI, NB, U, true); else F2V_W2(OPT, LPI, NB, U, false); } while (0)
    switch (width) {
        case 16: if (o == 5) F2V_W(5, 4, 1, 8); else F2V_W(6, 4, 1, 8); break;
        case 32: if (o == 5) F2V_W(5, 8, 1, 8); else F2V_W(6, 8, 1, 8); break;
        case 64: if (o == 5) F2V_W(5, 16, 1, 8); else F2V_W(6, 16, 1, 8); break;
        case 128: if (o == 5) F2V_W(5, 16, 2, 4); else F2V_W(6, 16, 2, 4); break;
        default: if (o == 5) F2V_W(5, 16, 4, 4); else F2V_W(6, 16, 4, 4); break;
    }
#undef F2V_W
#undef F2V_W2
#undef F2V_W3
    HIPC(hipGetLastError());
    if (ring) {  // the last epoch's matrix becomes the current one: nothing is pending
        HIPC(hipMemcpyAsync(c->d_X[c->cur], c->d_ring + (size_t)epochs * matrix, matrix * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    } else {
        if (c->upd_hi == c->upd_lo) c->upd_lo = plan.lo;
        c->upd_hi = plan.hi;
        c->pending = true;
        c->p_lo = plan.last_lo;
        c->p_hi = plan.hi;
    }
    c->stats.hub_rows += (uint64_t)plan.n_hubs * epochs;
    c->stats.hub_chunks += (uint64_t)plan.n_chunks * epochs;
    c->stats.step_launches += 1;
    c->stats.rows += (uint64_t)(plan.hi - plan.lo) * epochs;
    c->stats.nnz += plan.nnz * epochs;
    c->stats.algorithmic_bytes += (plan.nnz * (4ull * c->D + 4) + (uint64_t)(plan.hi - plan.lo) * (8ull * c->D + 4) + (uint64_t)plan.n_batches * ns * (4ull * c->D + 4)) * epochs;
    c->stats.compulsory_bytes += plan.compulsory * epochs;
    return F2V_OK;
}

void generate_walks_host(f2v_ctx *c, std::vector<uint32_t> &walks) {
    // sample/algorithms.cpp:1097-1118, from the handle's rand() stream (f2v_host.cpp: walks_host)
    walks.resize((size_t)c->n * kWalkLength);
    walks_host(c->rng, c->rowptr.data(), c->colids.data(), c->n, c->nnz, walks.data());
}

// ---- push exchange helpers (include/f2v.h) ---------------------------------------------------------------
// slice of rank r of minibatch [lo,hi): contiguous; work-balanced over the CSR (f2v_shard_bounds), equal row counts
// for the walk-based option 7 (every row has five pairs)
void shard_of(const f2v_ctx *c, bool walk, uint32_t lo, uint32_t hi, uint32_t rank, uint32_t world, uint32_t *my_lo, uint32_t *my_hi) {
    if (walk) {
        const uint32_t per = (hi - lo + world - 1) / world;
        *my_lo = (uint32_t)std::min<uint64_t>((uint64_t)lo + (uint64_t)rank * per, hi);
        *my_hi = (uint32_t)std::min<uint64_t>((uint64_t)*my_lo + per, hi);
        return;
    }
    uint32_t bounds[kMaxRanks + 1];
    (void)f2v_shard_bounds(c->rowptr.data(), lo, hi, world, bounds);
    *my_lo = bounds[rank];
    *my_hi = bounds[rank + 1];
}

// where the peers receive rows of the minibatch that starts at `batch_lo`: their copy of matrix `which`, or the half of
// their landing buffer this exchange uses
void fill_targets(const f2v_ctx *c, PushTargets &t, int which, uint32_t batch_lo, const uint32_t *d_masks) {
    const auto &P = c->push;
    for (uint32_t r = 0; r < P.world; r++)
        t.peer[r] = P.landing ? P.peer_landing[r] + (size_t)(P.round & 1u) * P.landing_cap * c->D : P.peer_X[which][r];
    t.masks = d_masks;
    t.self = P.rank;
    t.world = P.world;
    t.row_base = P.landing ? batch_lo : 0u;
}

// copy my rows [row_lo,row_hi) of local matrix `which` to the peers that read them
int launch_push(f2v_ctx *c, int which, const uint32_t *d_masks, uint32_t batch_lo, uint32_t row_lo, uint32_t row_hi) {
    if (c->push.world < 2 || row_hi <= row_lo) return F2V_OK;
    PushArgs a{};
    a.src = c->d_X[which];
    fill_targets(c, a.to, which, batch_lo, d_masks);
    a.row_lo = row_lo;
    a.rows = row_hi - row_lo;
    a.D = c->D;
    const uint32_t wpb = 4;
    const uint32_t blocks = std::min<uint32_t>((a.rows + wpb - 1) / wpb, 2048u);
    int rc = dispatch_layout(c, [&](auto V, auto E) {
        hipLaunchKernelGGL((push_rows_kernel<decltype(V)::value, decltype(E)::value>), dim3(blocks), dim3(64 * wpb), 0, c->stream, a);
    });
    if (rc != F2V_OK) return rc;
    HIPC(hipGetLastError());
    return F2V_OK;
}

int launch_barrier(f2v_ctx *c);

// End of one exchange: the flag barrier and, in landing-buffer mode, the move of what the peers sent for minibatch
// [lo,hi) (this rank computed [my_lo,my_hi)) from the landing buffer into matrix `which`.
int finish_exchange(f2v_ctx *c, int which, const uint32_t *d_masks, uint32_t lo, uint32_t hi, uint32_t my_lo, uint32_t my_hi) {
    int rc = launch_barrier(c);
    if (rc != F2V_OK) return rc;
    auto &P = c->push;
    if (P.world < 2 || !P.landing) return F2V_OK;
    UnpackArgs u{};
    u.landing = P.landing_buf + (size_t)(P.round & 1u) * P.landing_cap * c->D;
    u.X = c->d_X[which];
    u.masks = d_masks;
    u.lo = lo;
    u.rows = hi - lo;
    u.my_lo = my_lo;
    u.my_hi = my_hi;
    u.D = c->D;
    u.self = P.rank;
    const uint32_t wpb = 4;
    const uint32_t blocks = std::min<uint32_t>((u.rows + wpb - 1) / wpb, 2048u);
    rc = dispatch_layout(c, [&](auto V, auto E) {
        hipLaunchKernelGGL((unpack_rows_kernel<decltype(V)::value, decltype(E)::value>), dim3(blocks), dim3(64 * wpb), 0, c->stream, u);
    });
    if (rc != F2V_OK) return rc;
    HIPC(hipGetLastError());
    P.round++;
    return F2V_OK;
}

int launch_barrier(f2v_ctx *c) {
    if (c->push.world < 2) return F2V_OK;
    BarrierArgs b{};
    b.flags = c->push.flags;
    for (uint32_t r = 0; r < c->push.world; r++) b.peer_flags[r] = c->push.peer_flags[r];
    b.err = c->push.d_err;
    b.seq = ++c->push.seq;
    b.timeout_ticks = (unsigned long long)c->push.timeout_ms * 100000ull;  // wall_clock64 ticks at 100 MHz
    b.self = c->push.rank;
    b.world = c->push.world;
    hipLaunchKernelGGL(xgmi_barrier_kernel, dim3(1), dim3(64), 0, c->stream, b);
    HIPC(hipGetLastError());
    return F2V_OK;
}

// after a stream synchronisation: did every barrier see all peers?
int check_push_err(f2v_ctx *c, const char *where) {
    if (c->push.world < 2) return F2V_OK;
    uint32_t e = 0;
    HIPC(hipMemcpy(&e, c->push.d_err, sizeof e, hipMemcpyDeviceToHost));
    if (e) return fail(F2V_ESTATE, "%s: a peer did not reach the xGMI barrier within %lld ms", where, (long long)c->push.timeout_ms);
    return F2V_OK;
}

int push_detach(f2v_ctx *c) {
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->push.attached && !c->push.local) {
        for (uint32_t r = 0; r < c->push.world; r++) {
            if (r == c->push.rank) continue;
            for (int k = 0; k < 2; k++)
                if (c->push.peer_X[k][r]) (void)hipIpcCloseMemHandle(c->push.peer_X[k][r]);
            if (c->push.peer_flags[r]) (void)hipIpcCloseMemHandle(c->push.peer_flags[r]);
            if (c->push.peer_landing[r]) (void)hipIpcCloseMemHandle(c->push.peer_landing[r]);
        }
    }
    for (int k = 0; k < 2; k++)
        for (int r = 0; r < kMaxRanks; r++) c->push.peer_X[k][r] = nullptr;
    for (int r = 0; r < kMaxRanks; r++) { c->push.peer_flags[r] = nullptr; c->push.peer_landing[r] = nullptr; }
    c->push.attached = false;
    c->push.local = false;
    c->push.rank = 0;
    c->push.world = 1;
    if (c->shared_card) { c->shared_card = false; drop_plans(c); }
    return F2V_OK;
}

// Reader masks of this run in HBM: the neighbour part is static per (batch, world) and cached; the vertices this
// run samples are read by everyone and are patched in (and last run's patched out) by two scatter launches.
int prepare_masks(f2v_ctx *c, uint32_t batch, const std::vector<uint32_t> &ids) {
    auto &P = c->push;
    if (P.mask_batch != batch || P.mask_world != P.world || P.base_masks.size() != c->n || !P.d_masks) {
        P.base_masks.assign(c->n, 0u);
        int rc = f2v_push_masks(c->rowptr.data(), c->colids.data(), c->n, batch, P.world, nullptr, 0, P.base_masks.data());
        if (rc != F2V_OK) return rc;
        HIPC(hipStreamSynchronize(c->stream));
        if (!P.d_masks) HIPC(hipMalloc((void **)&P.d_masks, (size_t)c->n * sizeof(uint32_t)));
        HIPC(hipMemcpy(P.d_masks, P.base_masks.data(), (size_t)c->n * sizeof(uint32_t), hipMemcpyHostToDevice));
        P.patched_ids.clear();
        P.mask_batch = batch;
        P.mask_world = P.world;
        P.pushed_per_epoch = 0;
        for (uint32_t b0 = 0; b0 < c->n; b0 += batch) {
            uint32_t lo, hi;
            shard_of(c, false, b0, (uint32_t)std::min<uint64_t>((uint64_t)b0 + batch, c->n), P.rank, P.world, &lo, &hi);
            for (uint32_t v = lo; v < hi; v++) P.pushed_per_epoch += (uint64_t)__builtin_popcount(P.base_masks[v]);
        }
    }
    // patch list: [restore ids | restore values | set ids | set values]
    std::vector<uint32_t> uniq(ids);
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    const size_t nr = P.patched_ids.size(), ns = uniq.size();
    if (nr + ns == 0) return F2V_OK;
    std::vector<uint32_t> h(2 * (nr + ns));
    for (size_t k = 0; k < nr; k++) { h[k] = P.patched_ids[k]; h[nr + k] = P.base_masks[P.patched_ids[k]]; }
    const uint32_t everyone = (P.world >= 32 ? 0xFFFFFFFFu : ((1u << P.world) - 1u));
    for (size_t k = 0; k < ns; k++) { h[2 * nr + k] = uniq[k]; h[2 * nr + ns + k] = everyone; }
    HIPC(hipStreamSynchronize(c->stream));
    if (h.size() > P.patch_cap) {
        if (P.d_patch) (void)hipFree(P.d_patch);
        P.d_patch = nullptr;
        P.patch_cap = h.size() * 2;
        HIPC(hipMalloc((void **)&P.d_patch, P.patch_cap * sizeof(uint32_t)));
    }
    HIPC(hipMemcpy(P.d_patch, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (nr) hipLaunchKernelGGL(mask_patch_kernel, dim3((uint32_t)((nr + 255) / 256)), dim3(256), 0, c->stream, P.d_masks, P.d_patch, P.d_patch + nr, (uint32_t)nr);
    if (ns) hipLaunchKernelGGL(mask_patch_kernel, dim3((uint32_t)((ns + 255) / 256)), dim3(256), 0, c->stream, P.d_masks, P.d_patch + 2 * nr, P.d_patch + 2 * nr + ns, (uint32_t)ns);
    HIPC(hipGetLastError());
    P.patched_ids.swap(uniq);
    return F2V_OK;
}

int train_impl(f2v_ctx *c, int option, uint32_t iters, uint32_t batch, uint32_t ns, float lr, int bs_mode, double *seconds_out, bool sharded);

}  // namespace

extern "C" {

int f2v_create(const uint32_t *rowptr, const uint32_t *colids, uint32_t n, uint64_t nnz, uint32_t dim, int device,
               f2v_handle *out) {
    if (!rowptr || (!colids && nnz) || !out) return fail(F2V_EINVAL, "f2v_create: null argument");
    if (n < 2) return fail(F2V_EINVAL, "f2v_create: need at least 2 vertices (rand() %% (N-1))");
    if (dim == 0 || dim > 512) return fail(F2V_EINVAL, "f2v_create: dim %u outside 1..512", dim);
    if (nnz >= 0xFFFFFFFFull) return fail(F2V_EINVAL, "f2v_create: nnz exceeds 32-bit row pointers");
    if (rowptr[0] != 0 || rowptr[n] != nnz) return fail(F2V_EINVAL, "f2v_create: rowptr[0] must be 0 and rowptr[n] == nnz");
    for (uint32_t i = 0; i < n; i++)
        if (rowptr[i + 1] < rowptr[i]) return fail(F2V_EINVAL, "f2v_create: rowptr not monotone at row %u", i);
    for (uint64_t k = 0; k < nnz; k++)
        if (colids[k] >= n) return fail(F2V_EINVAL, "f2v_create: colids[%llu] = %u is not a vertex", (unsigned long long)k, colids[k]);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(F2V_ENODEV, "f2v_create: no HIP device visible (libf2v has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(F2V_ENODEV, "f2v_create: device %d not in [0,%d)", device, ndev);
    HIPC(hipSetDevice(device));
    f2v_ctx *c = new f2v_ctx();
    c->device = device;
    c->n = n;
    c->nnz = nnz;
    c->D = dim;
    c->vec = pick_vec(dim);
    c->exact = ((uint32_t)(64 * c->vec) == dim);
    c->rowptr.assign(rowptr, rowptr + n + 1);
    c->colids.assign(colids, colids + nnz);
    c->rng.seed(1);
    auto bail = [&](int rc) { f2v_destroy(c); return rc; };
#define HIPB(expr)                                                                                                   \
    do {                                                                                                             \
        hipError_t e__ = (expr);                                                                                     \
        if (e__ != hipSuccess)                                                                                       \
            return bail(fail(e__ == hipErrorOutOfMemory ? F2V_ENOMEM : F2V_ENODEV, "%s: %s", #expr, hipGetErrorString(e__))); \
    } while (0)
    HIPB(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPB(hipMalloc((void **)&c->d_rowptr, ((size_t)n + 1) * sizeof(uint32_t)));
    HIPB(hipMalloc((void **)&c->d_colids, std::max<size_t>(nnz, 1) * sizeof(uint32_t)));
    for (int k = 0; k < 2; k++) HIPB(hipMalloc((void **)&c->d_X[k], ((size_t)n + kPadRows) * dim * sizeof(float)));
    HIPB(hipMalloc((void **)&c->d_table, kSmTableSize * sizeof(float)));
    HIPB(hipMalloc((void **)&c->d_kerr, 64));
    HIPB(hipMemsetAsync(c->d_kerr, 0, 64, c->stream));
    HIPB(hipStreamSynchronize(c->stream));
    HIPB(hipMemcpy(c->d_rowptr, rowptr, ((size_t)n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (nnz) HIPB(hipMemcpy(c->d_colids, colids, nnz * sizeof(uint32_t), hipMemcpyHostToDevice));
    float table[kSmTableSize];
    sm_table_host(table);
    HIPB(hipMemcpy(c->d_table, table, sizeof table, hipMemcpyHostToDevice));
    HIPB(hipHostMalloc((void **)&c->h_kerr, 4 * 64, hipHostMallocDefault));
    memset(c->h_kerr, 0, 4 * 64);
    if (const char *e = getenv("F2V_TREE_TIMEOUT_MS")) {
        const long ms = atol(e);
        if (ms > 0 && ms <= 600000) c->tree_timeout_ms = ms;
    }
    if (const char *e = getenv("F2V_RECOVER")) c->recover = atoi(e) != 0;  // default of "recover" for new handles
#ifdef F2V_TEST_HOOKS
    if (const char *e = getenv("F2V_TEST_WITHHOLD_SLOT")) c->test_withhold_slot = (uint32_t)strtoul(e, nullptr, 0);  // f2v_test_withhold_flag from outside
    if (const char *e = getenv("F2V_TEST_WITHHOLD_ROW")) c->test_withhold_row = (uint32_t)strtoul(e, nullptr, 0);
#endif
    {
        // Dispatch probe: the one-launch minibatch (combine-tree nodes waiting inside the step kernel's grid) counts on
        // 8 XCDs taking workgroups round robin.  Checked here, on this device as this process sees it; if it does not
        // hold the handle starts with "merge_finalize" = 0 (one launch per tree level: no in-grid waits at all).
        constexpr uint32_t kProbe = 256;
        uint32_t *d_x = nullptr, h_x[kProbe];
        HIPB(hipMalloc((void **)&d_x, kProbe * sizeof(uint32_t)));
        hipLaunchKernelGGL(xcc_probe_kernel, dim3(kProbe), dim3(64), 0, c->stream, d_x);
        HIPB(hipGetLastError());
        HIPB(hipMemcpyAsync(h_x, d_x, sizeof h_x, hipMemcpyDeviceToHost, c->stream));
        HIPB(hipStreamSynchronize(c->stream));
        (void)hipFree(d_x);
        uint32_t seen = 0;
        bool rr = true;
        for (uint32_t b = 0; b < kProbe; b++) {
            seen |= 1u << (h_x[b] & 15u);
            if (h_x[b] != h_x[b % 8u]) rr = false;
        }
        c->xcc_count = (uint32_t)__builtin_popcount(seen);
        for (uint32_t b = 1; b < 8u; b++)
            for (uint32_t k = 0; k < b; k++)
                if (h_x[b] == h_x[k]) rr = false;
        c->xcc_round_robin = rr && c->xcc_count == 8u;
        if (!c->xcc_round_robin) c->merge_fin = false;
    }
#undef HIPB
    *out = c;
    return F2V_OK;
}

int f2v_destroy(f2v_handle c) {
    if (!c) return F2V_OK;
    (void)hipSetDevice(c->device);
    (void)push_detach(c);
    void *ptrs[] = {c->d_rowptr, c->d_colids, c->d_walks, c->d_walks_alt, c->d_ids, c->d_X[0], c->d_X[1],
                    c->d_partials, c->d_table, c->d_items, c->d_hubs, c->d_ready, c->d_kerr, c->d_wg, c->d_rowflag, c->d_snap, c->d_wide, c->d_jobs, c->d_ring, c->d_ring_partials, c->d_ring_flags, c->d_ring_ready, c->push.flags, c->push.d_err, c->push.d_masks, c->push.d_patch, c->push.landing_buf};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
#ifdef F2V_TEST_HOOKS
    if (c->d_stamps) (void)hipFree(c->d_stamps);
    if (c->d_xcd) (void)hipFree(c->d_xcd);
#endif
    if (c->h_kerr) (void)hipHostFree(c->h_kerr);
    for (hipEvent_t e : c->ev_snap)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return F2V_OK;
}

int f2v_srand(f2v_handle c, uint32_t seed) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    c->rng.seed(seed);
    c->fast_seed = seed;
    c->fast_epoch = 0;
    return F2V_OK;
}

int f2v_rand_index(f2v_handle c, uint32_t max_num, uint32_t min_num, uint32_t *out) {
    if (!c || !out || max_num <= min_num) return fail(F2V_EINVAL, "f2v_rand_index: bad argument");
    *out = c->rng.index(max_num, min_num);
    return F2V_OK;
}

int f2v_rand_indices(f2v_handle c, uint32_t max_num, uint32_t min_num, uint64_t count, uint64_t keep, uint32_t *out) {
    if (!c || max_num <= min_num || keep > count || (keep && !out)) return fail(F2V_EINVAL, "f2v_rand_indices: bad argument");
    for (uint64_t k = 0; k < count; k++) {
        const uint32_t r = c->rng.index(max_num, min_num);
        if (k < keep) out[k] = r;
    }
    return F2V_OK;
}

int f2v_init_embeddings(f2v_handle c, int kind) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    if (kind != F2V_INIT_SYMMETRIC && kind != F2V_INIT_UNIT) return fail(F2V_EINVAL, "f2v_init_embeddings: kind %d", kind);
    HIPC(hipSetDevice(c->device));
    const size_t total = (size_t)c->n * c->D;
    if (c->fast_rng) {  // non-parity: hash-based uniform values generated in HBM
        HIPC(hipStreamSynchronize(c->stream));
        c->pending = false;
        c->upd_lo = c->upd_hi = 0;
        hipLaunchKernelGGL(fast_init_kernel, dim3(4096), dim3(256), 0, c->stream, c->d_X[c->cur], (uint64_t)total, kind, c->fast_seed * 0x9E3779B97F4A7C15ull + 17);
        HIPC(hipGetLastError());
        HIPC(hipStreamSynchronize(c->stream));
        c->have_x = true;
        c->x_invalid = false;
        return F2V_OK;
    }
    HIPC(hipStreamSynchronize(c->stream));
    c->pending = false;
    c->upd_lo = c->upd_hi = 0;
    constexpr size_t kPiece = (size_t)64 << 20;  // floats: 256 MiB
    if (total >= 2 * kPiece) {
        // Large matrices (8 GiB at RMAT-24): the one serial rand() stream is filled piece by piece into two pinned staging
        // buffers by the fill threads (jump-ahead states, bit-identical to the serial stream) while the previous piece
        // travels to HBM -- the pageable-memory copy of the whole matrix used to take longer than generating it.
        float *stage[2] = {nullptr, nullptr};
        hipEvent_t done[2];
        bool used[2] = {false, false};
        for (int k = 0; k < 2; k++) {
            HIPC(hipHostMalloc((void **)&stage[k], kPiece * sizeof(float), hipHostMallocDefault));
            HIPC(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
        }
        int k = 0;
        for (size_t off = 0; off < total; off += kPiece, k ^= 1) {
            const size_t len = std::min(kPiece, total - off);
            if (used[k]) HIPC(hipEventSynchronize(done[k]));
            init_embeddings_host(c->rng, stage[k], len, kind);
            HIPC(hipMemcpyAsync(c->d_X[c->cur] + off, stage[k], len * sizeof(float), hipMemcpyHostToDevice, c->stream));
            HIPC(hipEventRecord(done[k], c->stream));
            used[k] = true;
        }
        HIPC(hipStreamSynchronize(c->stream));
        for (int j = 0; j < 2; j++) { (void)hipEventDestroy(done[j]); (void)hipHostFree(stage[j]); }
    } else {
        std::unique_ptr<float[]> x(new float[total]);  // not value-initialised: the fill threads are the first to touch it
        init_embeddings_host(c->rng, x.get(), total, kind);
        HIPC(hipMemcpy(c->d_X[c->cur], x.get(), total * sizeof(float), hipMemcpyHostToDevice));
    }
    c->have_x = true;
    c->x_invalid = false;
    return F2V_OK;
}

int f2v_set_embeddings(f2v_handle c, const float *x) {
    if (!c || !x) return fail(F2V_EINVAL, "f2v_set_embeddings: null argument");
    HIPC(hipSetDevice(c->device));
    HIPC(hipStreamSynchronize(c->stream));
    c->pending = false;
    c->upd_lo = c->upd_hi = 0;
    HIPC(hipMemcpy(c->d_X[c->cur], x, (size_t)c->n * c->D * sizeof(float), hipMemcpyHostToDevice));
    c->have_x = true;
    c->x_invalid = false;
    return F2V_OK;
}

int f2v_get_embeddings(f2v_handle c, float *x_out) {
    if (!c || !x_out) return fail(F2V_EINVAL, "f2v_get_embeddings: null argument");
    if (!c->have_x)
        return fail(F2V_ESTATE, c->x_invalid ? "f2v_get_embeddings: the embeddings are invalid since a launch gave up a bounded wait: set or initialise them again"
                                             : "f2v_get_embeddings: embeddings were never initialised");
    HIPC(hipSetDevice(c->device));
    int rc = flush_pending(c);
    if (rc != F2V_OK) return rc;
    HIPC(hipStreamSynchronize(c->stream));
    if ((rc = check_kernel_err(c, "f2v_get_embeddings")) != F2V_OK) return rc;
    HIPC(hipMemcpy(x_out, c->d_X[c->cur], (size_t)c->n * c->D * sizeof(float), hipMemcpyDeviceToHost));
    return F2V_OK;
}

int f2v_set_param(f2v_handle c, const char *name, int64_t value) {
    if (!c || !name) return fail(F2V_EINVAL, "f2v_set_param: null argument");
    if (!strcmp(name, "hub_chunk")) {
        if (value < 0 || value > 0x7FFFFFFF) return fail(F2V_EINVAL, "hub_chunk out of range");
        HIPC(hipSetDevice(c->device));
        int rc = flush_pending(c);
        if (rc != F2V_OK) return rc;
        HIPC(hipStreamSynchronize(c->stream));
        if (value > (int64_t)kItemSlotMask) return fail(F2V_EINVAL, "hub_chunk out of range");
        // partial-sum slots are 28-bit: pieces plus the nodes of their combine trees must stay below 2^28
        if (value > 0 && c->nnz / (uint64_t)value >= (1ull << 27)) return fail(F2V_EINVAL, "hub_chunk %lld is too small for %llu nonzeros", (long long)value, (unsigned long long)c->nnz);
        c->chunk = (uint32_t)value;
        c->chunk_auto = false;
        drop_plans(c);
        return F2V_OK;
    }
    if (!strcmp(name, "hub_chunk_for_batch")) {  // resolve the automatic chunk for this batch size now
        if (value <= 0 || value > 0xFFFFFFFFll) return fail(F2V_EINVAL, "hub_chunk_for_batch: bad batch size");
        HIPC(hipSetDevice(c->device));
        int rc = flush_pending(c);
        if (rc != F2V_OK) return rc;
        HIPC(hipStreamSynchronize(c->stream));
        const uint32_t ch = auto_chunk(c, (uint32_t)value);
        if (ch != c->chunk) { c->chunk = ch; drop_plans(c); }
        c->chunk_auto = true;
        return F2V_OK;
    }
    if (!strcmp(name, "hub_fanin")) {
        if (value < 0 || value == 1 || value > 0x7FFFFFFF) return fail(F2V_EINVAL, "hub_fanin must be 0 (one sequential pass) or >= 2");
        HIPC(hipSetDevice(c->device));
        int rc = flush_pending(c);
        if (rc != F2V_OK) return rc;
        HIPC(hipStreamSynchronize(c->stream));
        c->fanin = (uint32_t)value;
        drop_plans(c);
        return F2V_OK;
    }
    if (!strcmp(name, "quarter_wave")) {
        if (c->use_quarter != (value != 0)) {  // the item layout follows the kernel's items per workgroup
            HIPC(hipSetDevice(c->device));
            int rc = flush_pending(c);
            if (rc != F2V_OK) return rc;
            HIPC(hipStreamSynchronize(c->stream));
            c->use_quarter = value != 0;
            drop_plans(c);
        }
        return F2V_OK;
    }
    if (!strcmp(name, "fast_rng")) {
        c->fast_rng = value != 0;
        return F2V_OK;
    }
    if (!strcmp(name, "use_graph")) {
        c->use_graph = value != 0;
        return F2V_OK;
    }
    if (!strcmp(name, "class_cut")) {
        HIPC(hipSetDevice(c->device));
        int rc = flush_pending(c);
        if (rc != F2V_OK) return rc;
        HIPC(hipStreamSynchronize(c->stream));
        c->class_cut = value != 0;
        drop_plans(c);
        return F2V_OK;
    }
    if (!strcmp(name, "piece_affinity")) {
        if (c->piece_affinity != (value != 0)) {
            HIPC(hipSetDevice(c->device));
            int rc = flush_pending(c);
            if (rc != F2V_OK) return rc;
            HIPC(hipStreamSynchronize(c->stream));
            c->piece_affinity = value != 0;
            drop_plans(c);
        }
        return F2V_OK;
    }
    if (!strcmp(name, "count_compulsory")) {
        if (c->count_compulsory != (value != 0)) {
            HIPC(hipSetDevice(c->device));
            int rc = flush_pending(c);
            if (rc != F2V_OK) return rc;
            HIPC(hipStreamSynchronize(c->stream));
            c->count_compulsory = value != 0;
            drop_plans(c);
        }
        return F2V_OK;
    }
    if (!strcmp(name, "rows_in_flight")) {
        if (value != 0 && value != 4 && value != 8) return fail(F2V_EINVAL, "rows_in_flight must be 0 (default), 4 or 8");
        c->rows_in_flight = (int)value;
        return F2V_OK;
    }
    if (!strcmp(name, "push_fused")) {
        c->push.fused = value != 0;
        return F2V_OK;
    }
    if (!strcmp(name, "merge_finalize")) {
        c->merge_fin = value != 0;
        c->waits_suspended = false;  // the caller's own choice stands
        return F2V_OK;
    }
    if (!strcmp(name, "chain_batches")) {
        c->chain = value != 0;
        return F2V_OK;
    }
    if (!strcmp(name, "chain_max_batch")) {
        if (value < 0 || value > 0xFFFFFFFFll) return fail(F2V_EINVAL, "chain_max_batch out of range");
        c->chain_max_batch = (uint32_t)value;
        return F2V_OK;
    }
    if (!strcmp(name, "wide_epochs")) {
        if (value < 0 || value > 1024) return fail(F2V_EINVAL, "wide_epochs must be 0 (automatic) ... 1024");
        c->wide_epochs = (uint32_t)value;
        return F2V_OK;
    }
    if (!strcmp(name, "wide_single")) {
        c->wide_single = value != 0;
        return F2V_OK;
    }
    if (!strcmp(name, "replicate_small")) {
        if (value < 0 || value > 2) return fail(F2V_EINVAL, "replicate_small must be 0 (never), 1 (where no peer shares the GPU) or 2 (always)");
        c->replicate_small = (int)value;
        return F2V_OK;
    }
    if (!strcmp(name, "wide_samples_early")) {
        if (value < -1 || value > 1) return fail(F2V_EINVAL, "wide_samples_early must be -1 (automatic), 0 or 1");
        c->wide_samples_early = (int)value;
        return F2V_OK;
    }
    if (!strcmp(name, "wide_min_width")) {
        if (value != 0 && value != 16 && value != 32 && value != 64 && value != 128) return fail(F2V_EINVAL, "wide_min_width must be 0 (automatic), 16, 32, 64 or 128");
        HIPC(hipSetDevice(c->device));
        int rc = flush_pending(c);
        if (rc != F2V_OK) return rc;
        HIPC(hipStreamSynchronize(c->stream));
        c->wide_min_width = (uint32_t)value;
        drop_plans(c);
        return F2V_OK;
    }
    if (!strcmp(name, "wide_max_batch")) {
        if (value < 0 || value > 0xFFFFFFFFll) return fail(F2V_EINVAL, "wide_max_batch out of range");
        c->wide_max_batch = (uint32_t)value;
        return F2V_OK;
    }
    if (!strcmp(name, "wide_rows")) {
        if (value < 2 || value > 0x7FFFFFFFll) return fail(F2V_EINVAL, "wide_rows out of range");
        HIPC(hipSetDevice(c->device));
        int rc = flush_pending(c);
        if (rc != F2V_OK) return rc;
        HIPC(hipStreamSynchronize(c->stream));
        c->wide_rows = (uint32_t)value;
        drop_plans(c);
        return F2V_OK;
    }
    if (!strcmp(name, "chain_rows")) {  // rows one chained launch covers
        if (value < 2 || value > 0x7FFFFFFFll) return fail(F2V_EINVAL, "chain_rows out of range");
        HIPC(hipSetDevice(c->device));
        int rc = flush_pending(c);
        if (rc != F2V_OK) return rc;
        HIPC(hipStreamSynchronize(c->stream));
        c->chain_rows = c->wide_rows = (uint32_t)value;  // an explicit value holds for both forms ("wide_rows" alone: the wide form's)
        drop_plans(c);
        return F2V_OK;
    }
    if (!strcmp(name, "tree_timeout_ms")) {
        if (value < 1 || value > 600000) return fail(F2V_EINVAL, "tree_timeout_ms must be 1..600000");
        c->tree_timeout_ms = value;
        return F2V_OK;
    }
    if (!strcmp(name, "epoch_marks")) {
        if (value < 0 || value > 0x7FFFFFFF) return fail(F2V_EINVAL, "epoch_marks out of range");
        c->mark_every = (uint32_t)value;
        return F2V_OK;
    }
    if (!strcmp(name, "chain_wide") || !strcmp(name, "wide_phases") || !strcmp(name, "wide_span") || !strcmp(name, "wide_finish") || !strcmp(name, "wide_order") || !strcmp(name, "wide_rounds")) {
        if (name[0] == 'w' && strcmp(name, "wide_order") && strcmp(name, "wide_rounds") && (value < 1 || value > 64)) return fail(F2V_EINVAL, "%s must be 1..64", name);
        if (!strcmp(name, "wide_rounds") && (value < 0 || value > 64)) return fail(F2V_EINVAL, "wide_rounds must be 0..64");
        if (!strcmp(name, "wide_order") && (value < 0 || value > 2)) return fail(F2V_EINVAL, "wide_order must be 0, 1 or 2");
        HIPC(hipSetDevice(c->device));
        int rc = flush_pending(c);
        if (rc != F2V_OK) return rc;
        HIPC(hipStreamSynchronize(c->stream));
        if (!strcmp(name, "chain_wide")) c->wide = value != 0;
        else if (!strcmp(name, "wide_phases")) c->wide_phases = (uint32_t)value;
        else if (!strcmp(name, "wide_span")) c->wide_span = (uint32_t)value;
        else if (!strcmp(name, "wide_order")) c->wide_order = (uint32_t)value;
        else if (!strcmp(name, "wide_rounds")) c->wide_rounds = (uint32_t)value;
        else c->wide_finish = (uint32_t)value;
        drop_plans(c);
        return F2V_OK;
    }
    if (!strcmp(name, "chain_timeout_ms")) {
        if (value < 1 || value > 600000) return fail(F2V_EINVAL, "chain_timeout_ms must be 1..600000");
        c->chain_timeout_ms = value;
        return F2V_OK;
    }
    if (!strcmp(name, "recover")) {
        c->recover = value != 0;
        if (!c->recover && c->d_snap) {  // the snapshot matrix goes with the net
            HIPC(hipSetDevice(c->device));
            HIPC(hipStreamSynchronize(c->stream));
            (void)hipFree(c->d_snap);
            c->d_snap = nullptr;
        }
        return F2V_OK;
    }
    if (!strcmp(name, "push_landing")) {  // takes effect at the next f2v_push_export
        c->push.force_landing = value != 0;
        return F2V_OK;
    }
    if (!strcmp(name, "push_timeout_ms")) {
        if (value < 1 || value > 600000) return fail(F2V_EINVAL, "push_timeout_ms must be 1..600000");
        c->push.timeout_ms = value;
        return F2V_OK;
    }
    if (!strcmp(name, "waves_per_block")) {
        if (value != 1 && value != 2 && value != 4) return fail(F2V_EINVAL, "waves_per_block must be 1, 2 or 4");
        if (c->waves_per_block != (int)value) {
            HIPC(hipSetDevice(c->device));
            int rc = flush_pending(c);
            if (rc != F2V_OK) return rc;
            HIPC(hipStreamSynchronize(c->stream));
            c->waves_per_block = (int)value;
            drop_plans(c);
        }
        return F2V_OK;
    }
    return fail(F2V_EINVAL, "f2v_set_param: unknown parameter '%s'", name);
}

int f2v_get_param(f2v_handle c, const char *name, int64_t *out) {
    if (!c || !name || !out) return fail(F2V_EINVAL, "f2v_get_param: null argument");
    if (!strcmp(name, "hub_chunk")) { *out = c->chunk; return F2V_OK; }
    if (!strcmp(name, "waves_per_block")) { *out = c->waves_per_block; return F2V_OK; }
    if (!strcmp(name, "quarter_wave")) { *out = c->use_quarter ? 1 : 0; return F2V_OK; }
    if (!strcmp(name, "hub_fanin")) { *out = c->fanin; return F2V_OK; }
    if (!strcmp(name, "fast_rng")) { *out = c->fast_rng ? 1 : 0; return F2V_OK; }
    if (!strcmp(name, "use_graph")) { *out = c->use_graph ? 1 : 0; return F2V_OK; }
    if (!strcmp(name, "count_compulsory")) { *out = c->count_compulsory ? 1 : 0; return F2V_OK; }
    if (!strcmp(name, "piece_affinity")) { *out = c->piece_affinity ? 1 : 0; return F2V_OK; }
    if (!strcmp(name, "class_cut")) { *out = c->class_cut ? 1 : 0; return F2V_OK; }
    if (!strcmp(name, "shared_card")) { *out = c->shared_card ? 1 : 0; return F2V_OK; }
    if (!strcmp(name, "hub_chunk_auto")) { *out = c->chunk_auto ? 1 : 0; return F2V_OK; }
    if (!strcmp(name, "push_timeout_ms")) { *out = c->push.timeout_ms; return F2V_OK; }
    if (!strcmp(name, "push_fused")) { *out = c->push.fused ? 1 : 0; return F2V_OK; }
    if (!strcmp(name, "merge_finalize")) { *out = c->merge_fin ? 1 : 0; return F2V_OK; }
    if (!strcmp(name, "tree_timeout_ms")) { *out = c->tree_timeout_ms; return F2V_OK; }
    if (!strcmp(name, "chain_batches")) { *out = c->chain ? 1 : 0; return F2V_OK; }
    if (!strcmp(name, "chain_timeout_ms")) { *out = c->chain_timeout_ms; return F2V_OK; }
    if (!strcmp(name, "chain_wide")) { *out = c->wide ? 1 : 0; return F2V_OK; }
    if (!strcmp(name, "epoch_marks")) { *out = c->mark_every; return F2V_OK; }
    if (!strcmp(name, "wide_phases")) { *out = c->wide_phases; return F2V_OK; }
    if (!strcmp(name, "wide_max_batch")) { *out = c->wide_max_batch; return F2V_OK; }
    if (!strcmp(name, "wide_min_width")) { *out = c->wide_min_width; return F2V_OK; }
    if (!strcmp(name, "wide_single")) { *out = c->wide_single ? 1 : 0; return F2V_OK; }
    if (!strcmp(name, "replicate_small")) { *out = c->replicate_small; return F2V_OK; }
    if (!strcmp(name, "last_train_replicated")) { *out = c->last_replicated ? 1 : 0; return F2V_OK; }
    if (!strcmp(name, "plan_resident_bytes")) {  // launch plans resident on the host (and, uploaded, in HBM): items, jobs, workgroup descriptors, tree nodes
        *out = (int64_t)(c->h_items.size() * sizeof(Item) + c->h_jobs.size() * sizeof(WJob) + c->h_wide.size() * sizeof(WideDesc) + c->h_wg.size() * sizeof(WgDesc) + c->h_hubs.size() * sizeof(FinItem));
        return F2V_OK;
    }
    if (!strcmp(name, "wide_samples_early")) { *out = c->wide_samples_early; return F2V_OK; }
    if (!strcmp(name, "wide_epochs")) { *out = c->wide_epochs; return F2V_OK; }
    if (!strcmp(name, "last_wide_epochs")) { *out = c->last_wide_epochs; return F2V_OK; }
    if (!strcmp(name, "wide_rows")) { *out = c->wide_rows; return F2V_OK; }
    if (!strcmp(name, "wide_span")) { *out = c->wide_span; return F2V_OK; }
    if (!strcmp(name, "wide_order")) { *out = c->wide_order; return F2V_OK; }
    if (!strcmp(name, "wide_rounds")) { *out = c->wide_rounds; return F2V_OK; }
    if (!strcmp(name, "wide_finish")) { *out = c->wide_finish; return F2V_OK; }
    if (!strcmp(name, "last_train_form")) { *out = c->last_train_form; return F2V_OK; }
    if (!strcmp(name, "last_wide_width")) { *out = c->last_wide_width; return F2V_OK; }
    if (!strcmp(name, "last_wide_early")) { *out = c->last_wide_early ? 1 : 0; return F2V_OK; }
    if (!strcmp(name, "recover")) { *out = c->recover ? 1 : 0; return F2V_OK; }
    if (!strcmp(name, "recoveries")) { *out = c->recoveries; return F2V_OK; }
    if (!strcmp(name, "chain_max_batch")) { *out = c->chain_max_batch; return F2V_OK; }
    if (!strcmp(name, "chain_rows")) { *out = c->chain_rows; return F2V_OK; }
    if (!strcmp(name, "xcc_count")) { *out = c->xcc_count; return F2V_OK; }
    if (!strcmp(name, "xcc_round_robin")) { *out = c->xcc_round_robin ? 1 : 0; return F2V_OK; }
    if (!strcmp(name, "push_landing")) { *out = (c->push.attached || c->push.exported) ? (c->push.landing ? 1 : 0) : (c->push.force_landing ? 1 : 0); return F2V_OK; }
    if (!strcmp(name, "push_world")) { *out = c->push.attached ? c->push.world : 0; return F2V_OK; }
    if (!strcmp(name, "push_rank")) { *out = c->push.rank; return F2V_OK; }
    if (!strcmp(name, "dim")) { *out = c->D; return F2V_OK; }
    if (!strcmp(name, "n")) { *out = c->n; return F2V_OK; }
    if (!strcmp(name, "nnz")) { *out = (int64_t)c->nnz; return F2V_OK; }
    return fail(F2V_EINVAL, "f2v_get_param: unknown parameter '%s'", name);
}

int f2v_set_walks(f2v_handle c, const uint32_t *walks) {
    if (!c || !walks) return fail(F2V_EINVAL, "f2v_set_walks: null argument");
    HIPC(hipSetDevice(c->device));
    const size_t cnt = (size_t)c->n * kWalkLength;
    for (size_t k = 0; k < cnt; k++)
        if (walks[k] >= c->n) return fail(F2V_EINVAL, "f2v_set_walks: walks[%zu] = %u is not a vertex", k, walks[k]);
    if (!c->d_walks) HIPC(hipMalloc((void **)&c->d_walks, cnt * sizeof(uint32_t)));
    // stream-ordered after every step that still reads the previous epoch's walks
    HIPC(hipMemcpyAsync(c->d_walks, walks, cnt * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    c->have_walks = true;
    return F2V_OK;
}

// non-parity walks generated in HBM (see fast_walks_kernel)
static int fast_walks(f2v_ctx *c) {
    const size_t cnt = (size_t)c->n * kWalkLength;
    if (!c->d_walks) HIPC(hipMalloc((void **)&c->d_walks, cnt * sizeof(uint32_t)));
    hipLaunchKernelGGL(fast_walks_kernel, dim3((c->n + 255) / 256), dim3(256), 0, c->stream, c->d_rowptr, c->d_colids, c->n, c->nnz,
                       c->d_walks, c->fast_seed * 0xD1342543DE82EF95ull + 5, c->fast_epoch++);
    HIPC(hipGetLastError());
    c->have_walks = true;
    return F2V_OK;
}

int f2v_generate_walks(f2v_handle c, uint32_t *walks_out) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    if (c->fast_rng) {
        HIPC(hipSetDevice(c->device));
        int rc = fast_walks(c);
        if (rc != F2V_OK) return rc;
        if (walks_out) {
            HIPC(hipStreamSynchronize(c->stream));
            HIPC(hipMemcpy(walks_out, c->d_walks, (size_t)c->n * kWalkLength * sizeof(uint32_t), hipMemcpyDeviceToHost));
        }
        return F2V_OK;
    }
    std::vector<uint32_t> w;
    generate_walks_host(c, w);
    if (walks_out) memcpy(walks_out, w.data(), w.size() * sizeof(uint32_t));
    return f2v_set_walks(c, w.data());
}

int f2v_minibatch_step(f2v_handle c, int option, uint32_t batch_lo, uint32_t batch_hi, uint32_t row_lo,
                       uint32_t row_hi, const uint32_t *sample_ids, uint32_t n_sample_ids, uint32_t ns, float lr,
                       int bs_mode) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    const int math = math_of_option(option);
    if (!math) return fail(F2V_EINVAL, "f2v_minibatch_step: option %d is outside 5..11", option);
    c->unit_degi = option == 10;
    if (!c->have_x) return fail(F2V_ESTATE, "f2v_minibatch_step: embeddings not initialised");
    if (batch_lo >= batch_hi || batch_hi > c->n) return fail(F2V_EINVAL, "f2v_minibatch_step: bad batch [%u,%u)", batch_lo, batch_hi);
    if (row_lo < batch_lo || row_hi > batch_hi || row_lo > row_hi) return fail(F2V_EINVAL, "f2v_minibatch_step: rows [%u,%u) outside the batch", row_lo, row_hi);
    if (math == 7 && bs_mode) return fail(F2V_EINVAL, "option 7 has no -bs 1 variant");
    if (math == 7 && !c->have_walks) return fail(F2V_ESTATE, "option 7 needs f2v_set_walks / f2v_generate_walks first");
    const uint32_t need = bs_mode ? (batch_hi - batch_lo) + ns - 1 : ns;
    if (ns && (!sample_ids || n_sample_ids < need)) return fail(F2V_EINVAL, "f2v_minibatch_step: %u sample ids needed, %u given", need, n_sample_ids);
    for (uint32_t k = 0; k < need; k++)
        if (sample_ids[k] >= c->n) return fail(F2V_EINVAL, "f2v_minibatch_step: sample id %u is not a vertex", sample_ids[k]);
    HIPC(hipSetDevice(c->device));
    int rc = reserve_ids(c, std::max<size_t>(need, 64));
    if (rc != F2V_OK) return rc;
    // the previous step may still be reading d_ids: order the copy behind it on the stream
    c->ids_valid = 0;
    if (need) {
        HIPC(hipMemcpyAsync(c->d_ids, sample_ids, need * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
    }
    return launch_step(c, math, batch_lo, batch_hi, row_lo, row_hi, c->d_ids, ns, lr, bs_mode);
}

int f2v_upload_sample_ids(f2v_handle c, const uint32_t *ids, uint64_t count) {
    if (!c || (!ids && count)) return fail(F2V_EINVAL, "f2v_upload_sample_ids: null argument");
    for (uint64_t k = 0; k < count; k++)
        if (ids[k] >= c->n) return fail(F2V_EINVAL, "f2v_upload_sample_ids: id %u is not a vertex", ids[k]);
    HIPC(hipSetDevice(c->device));
    int rc = reserve_ids(c, std::max<size_t>(count, 64));
    if (rc != F2V_OK) return rc;
    HIPC(hipStreamSynchronize(c->stream));  // steps in flight still read the previous ids
    if (count) HIPC(hipMemcpy(c->d_ids, ids, count * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->ids_valid = count;
    return F2V_OK;
}

int f2v_minibatch_step_at(f2v_handle c, int option, uint32_t batch_lo, uint32_t batch_hi, uint32_t row_lo,
                          uint32_t row_hi, uint64_t ids_offset, uint32_t ns, float lr, int bs_mode) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    const int math = math_of_option(option);
    if (!math) return fail(F2V_EINVAL, "f2v_minibatch_step_at: option %d is outside 5..11", option);
    c->unit_degi = option == 10;
    if (!c->have_x) return fail(F2V_ESTATE, "f2v_minibatch_step_at: embeddings not initialised");
    if (batch_lo >= batch_hi || batch_hi > c->n) return fail(F2V_EINVAL, "f2v_minibatch_step_at: bad batch [%u,%u)", batch_lo, batch_hi);
    if (row_lo < batch_lo || row_hi > batch_hi || row_lo > row_hi) return fail(F2V_EINVAL, "f2v_minibatch_step_at: rows [%u,%u) outside the batch", row_lo, row_hi);
    if (math == 7 && bs_mode) return fail(F2V_EINVAL, "option 7 has no -bs 1 variant");
    if (math == 7 && !c->have_walks) return fail(F2V_ESTATE, "option 7 needs f2v_set_walks / f2v_generate_walks first");
    const uint64_t need = bs_mode ? (uint64_t)(batch_hi - batch_lo) + ns - 1 : ns;
    if (ids_offset + need > c->ids_valid) return fail(F2V_EINVAL, "f2v_minibatch_step_at: ids [%llu,+%llu) were not uploaded", (unsigned long long)ids_offset, (unsigned long long)need);
    HIPC(hipSetDevice(c->device));
    return launch_step(c, math, batch_lo, batch_hi, row_lo, row_hi, c->d_ids + ids_offset, ns, lr, bs_mode);
}

int f2v_flush(f2v_handle c) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    HIPC(hipSetDevice(c->device));
    int rc = flush_pending(c);
    if (rc != F2V_OK) return rc;
    HIPC(hipStreamSynchronize(c->stream));
    return check_kernel_err(c, "f2v_flush");
}

int f2v_synchronize(f2v_handle c) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    HIPC(hipSetDevice(c->device));
    HIPC(hipStreamSynchronize(c->stream));
    return F2V_OK;
}

int f2v_stage_reserve(f2v_handle c, uint32_t rows) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    // new rows are written in place into the second matrix, which has kPadRows of slack behind row N
    if (rows > c->n + kPadRows) return fail(F2V_EINVAL, "f2v_stage_reserve: %u rows exceed the matrix", rows);
    return F2V_OK;
}

int f2v_stage_device_ptr(f2v_handle c, uint64_t *devptr_out, uint32_t *cap_out) {
    if (!c || !devptr_out) return fail(F2V_EINVAL, "f2v_stage_device_ptr: null argument");
    if (!c->pending) return fail(F2V_ESTATE, "f2v_stage_device_ptr: no minibatch is pending");
    *devptr_out = (uint64_t)(uintptr_t)(c->d_X[c->cur ^ 1] + (size_t)c->p_lo * c->D);
    if (cap_out) *cap_out = c->n + kPadRows - c->p_lo;
    return F2V_OK;
}

int f2v_stage_read(f2v_handle c, uint32_t row_lo, uint32_t row_hi, float *out) {
    if (!c || !out) return fail(F2V_EINVAL, "f2v_stage_read: null argument");
    if (!c->pending || row_lo < c->p_lo || row_hi > c->p_hi || row_lo > row_hi) return fail(F2V_ESTATE, "f2v_stage_read: rows outside the pending minibatch");
    HIPC(hipSetDevice(c->device));
    HIPC(hipStreamSynchronize(c->stream));
    HIPC(hipMemcpy(out, c->d_X[c->cur ^ 1] + (size_t)row_lo * c->D, (size_t)(row_hi - row_lo) * c->D * sizeof(float), hipMemcpyDeviceToHost));
    return F2V_OK;
}

int f2v_stage_write(f2v_handle c, uint32_t row_lo, uint32_t row_hi, const float *in) {
    if (!c || !in) return fail(F2V_EINVAL, "f2v_stage_write: null argument");
    if (!c->pending || row_lo < c->p_lo || row_hi > c->p_hi || row_lo > row_hi) return fail(F2V_ESTATE, "f2v_stage_write: rows outside the pending minibatch");
    HIPC(hipSetDevice(c->device));
    HIPC(hipStreamSynchronize(c->stream));
    HIPC(hipMemcpy(c->d_X[c->cur ^ 1] + (size_t)row_lo * c->D, in, (size_t)(row_hi - row_lo) * c->D * sizeof(float), hipMemcpyHostToDevice));
    return F2V_OK;
}

int f2v_rows_read(f2v_handle c, const uint32_t *ids, uint32_t count, float *out) {
    if (!c || (count && (!ids || !out))) return fail(F2V_EINVAL, "f2v_rows_read: null argument");
    HIPC(hipSetDevice(c->device));
    HIPC(hipStreamSynchronize(c->stream));
    const float *base = c->d_X[c->cur ^ 1];
    for (uint32_t k = 0; k < count; k++) {
        if (ids[k] >= c->n) return fail(F2V_EINVAL, "f2v_rows_read: id %u is not a vertex", ids[k]);
        HIPC(hipMemcpy(out + (size_t)k * c->D, base + (size_t)ids[k] * c->D, (size_t)c->D * sizeof(float), hipMemcpyDeviceToHost));
    }
    return F2V_OK;
}

int f2v_rows_write(f2v_handle c, const uint32_t *ids, uint32_t count, const float *in) {
    if (!c || (count && (!ids || !in))) return fail(F2V_EINVAL, "f2v_rows_write: null argument");
    HIPC(hipSetDevice(c->device));
    HIPC(hipStreamSynchronize(c->stream));
    float *base = c->d_X[c->cur ^ 1];
    for (uint32_t k = 0; k < count; k++) {
        if (ids[k] >= c->n) return fail(F2V_EINVAL, "f2v_rows_write: id %u is not a vertex", ids[k]);
        HIPC(hipMemcpy(base + (size_t)ids[k] * c->D, in + (size_t)k * c->D, (size_t)c->D * sizeof(float), hipMemcpyHostToDevice));
    }
    return F2V_OK;
}

int f2v_embeddings_device_ptr(f2v_handle c, uint64_t *out) {
    if (!c || !out) return fail(F2V_EINVAL, "null argument");
    *out = (uint64_t)(uintptr_t)c->d_X[c->cur];  // the whole matrix only after f2v_flush
    return F2V_OK;
}

int f2v_stream(f2v_handle c, uint64_t *out) {
    if (!c || !out) return fail(F2V_EINVAL, "null argument");
    *out = (uint64_t)(uintptr_t)c->stream;
    return F2V_OK;
}

int f2v_get_stats(f2v_handle c, f2v_stats *out) {
    if (!c || !out) return fail(F2V_EINVAL, "null argument");
    *out = c->stats;
    out->recoveries = c->recoveries;
    out->merge_finalize = c->merge_fin ? 1u : 0u;
    return F2V_OK;
}

int f2v_train(f2v_handle c, int option, uint32_t iters, uint32_t batch, uint32_t ns, float lr, int bs_mode,
              double *seconds_out) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    // A launch whose workgroups wait for each other (combine-tree nodes inside the step kernel's grid, chained minibatches)
    // counts on this process having the GPU to itself: another process that fills the card with ITS waiting workgroups can
    // keep the ones everybody waits for from starting.  Every wait is bounded, and a wait that gives up must not cost the
    // caller its embeddings: the matrix and the rand() state are kept as they were when the call began ("recover", one more
    // matrix of HBM), and a call that loses a launch runs again from there with one launch per minibatch and per tree level
    // -- no in-grid waits, the same bits.
    bool snap = false;
    Rand rng0 = c->rng;
    const uint64_t fast_epoch0 = c->fast_epoch;
    if (c->waits_suspended && !c->merge_fin && ++c->calls_since_give_up > c->waits_backoff) {
        c->merge_fin = true;  // a recovered give-up is taken for a transient: the fast launch forms come back (with the net below)
        c->waits_suspended = false;
    }
    double snap_seconds = 0.0;
    if (c->recover && c->merge_fin && c->have_x && iters > 0) {  // (merge_fin: this handle's launches may hold in-grid waits)
        HIPC(hipSetDevice(c->device));
        int rc = flush_pending(c);
        if (rc != F2V_OK) return rc;
        const size_t bytes = (size_t)c->n * c->D * sizeof(float);
        if (!c->d_snap && hipMalloc((void **)&c->d_snap, bytes) != hipSuccess) {
            (void)hipGetLastError();  // no room for it: the call runs without the net
            c->d_snap = nullptr;
        }
        if (c->d_snap) {
            for (auto &e : c->ev_snap)
                if (!e) HIPC(hipEventCreateWithFlags(&e, hipEventDefault));
            HIPC(hipEventRecord(c->ev_snap[0], c->stream));
            HIPC(hipMemcpyAsync(c->d_snap, c->d_X[c->cur], bytes, hipMemcpyDeviceToDevice, c->stream));
            HIPC(hipEventRecord(c->ev_snap[1], c->stream));
            snap = true;
        }
    }
    const bool waits_before = c->merge_fin;
    int rc = train_impl(c, option, iters, batch, ns, lr, bs_mode, seconds_out, false);
    if (snap && hipEventQuery(c->ev_snap[1]) == hipSuccess) {  // (train_impl has synchronised the stream on every healthy way out)
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev_snap[0], c->ev_snap[1]) == hipSuccess) snap_seconds = ms * 1e-3;
    }
    (void)hipGetLastError();
    bool recovered = false;
    if (rc == F2V_ESTATE && snap && c->x_invalid && !c->merge_fin) {
        const std::string why = f2v_last_error();
        HIPC(hipStreamSynchronize(c->stream));
        HIPC(hipMemcpyAsync(c->d_X[c->cur], c->d_snap, (size_t)c->n * c->D * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        c->rng = rng0;
        c->fast_epoch = fast_epoch0;
        c->have_x = true;
        c->x_invalid = false;
        c->recoveries++;
        rc = train_impl(c, option, iters, batch, ns, lr, bs_mode, seconds_out, false);
        if (rc == F2V_OK) {
            recovered = true;
            (void)fail(F2V_OK, "f2v_train recovered: the call was run again from its start with one launch per minibatch and tree level after: %s", why.c_str());
            if (waits_before) {  // in-grid waits come back after a few healthy calls (doubling: 1, 2, 4 ... 64)
                c->waits_suspended = true;
                c->calls_since_give_up = 0;
                c->waits_backoff = c->recoveries <= 1 ? 1u : std::min(c->waits_backoff * 2u, 64u);
            }
        }
    }
    c->stats.snapshot_seconds = snap_seconds;
    c->stats.recovered = recovered ? 1u : 0u;
    return rc;
}

int f2v_train_marks(f2v_handle c, double *seconds_out, uint32_t cap, uint32_t *count_out) {
    if (!c || (cap && !seconds_out)) return fail(F2V_EINVAL, "f2v_train_marks: null argument");
    if (count_out) *count_out = (uint32_t)c->marks.size();
    for (uint32_t k = 0; k < cap && k < c->marks.size(); k++) seconds_out[k] = c->marks[k];
    return F2V_OK;
}

int f2v_train_sharded(f2v_handle c, int option, uint32_t iters, uint32_t batch, uint32_t ns, float lr, int bs_mode,
                      double *seconds_out) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    if (!c->push.attached) return fail(F2V_ESTATE, "f2v_train_sharded: f2v_push_attach first");
    const int math = math_of_option(option);
    c->last_replicated = false;
    if (math && batch && iters && !(math == 7 && bs_mode) && c->have_x && c->push.all_chain &&
        (c->replicate_small == 2 || (c->replicate_small == 1 && !c->push.any_shared))) {
        // Every rank must decide alike: the graph, the batch, the engine parameters (the callers set them alike on every rank, as they
        // must for the sharded form) and what f2v_push_attach learnt about ALL ranks.  Not this rank's "merge_finalize": a rank whose
        // in-grid waits are suspended after a recovered give-up still runs the whole epoch itself, one launch per minibatch -- same bits.
        const bool waits = c->merge_fin;
        c->merge_fin = true;
        const bool replicate = chain_usable(c, math, batch, bs_mode, false);
        c->merge_fin = waits;
        if (replicate) {
            c->push.rows_pushed = c->push.rows_allgather = 0;
            c->last_replicated = true;
            return f2v_train(c, option, iters, batch, ns, lr, bs_mode, seconds_out);
        }
    }
    return train_impl(c, option, iters, batch, ns, lr, bs_mode, seconds_out, true);
}

}  // extern "C"

namespace {

// f2v_train, and f2v_train_sharded when `sharded`: this rank computes its slice of every minibatch, pushes the new
// rows to the peers that read them and passes the flag barrier before the next minibatch.
int train_impl(f2v_ctx *c, int option, uint32_t iters, uint32_t batch, uint32_t ns, float lr, int bs_mode, double *seconds_out, bool sharded) {
    const int math = math_of_option(option);
    if (!math) return fail(F2V_EINVAL, "f2v_train: option %d is outside 5..11", option);
    c->unit_degi = option == 10;
    if (!c->have_x) return fail(F2V_ESTATE, "f2v_train: embeddings not initialised (f2v_init_embeddings)");
    if (batch == 0) return fail(F2V_EINVAL, "f2v_train: batch must be positive");
    if (math == 7 && bs_mode) return fail(F2V_EINVAL, "option 7 has no -bs 1 variant");
    HIPC(hipSetDevice(c->device));
    int rc;
    const uint32_t n = c->n;
    const uint32_t nb = (uint32_t)(((uint64_t)n + batch - 1) / batch);
    // rand() draws per minibatch: ns, or ns*BATCHSIZE with -bs 1 (algorithms.cpp:686, 966) of which rows+ns-1 are used
    const uint64_t ndraw = bs_mode ? (uint64_t)ns * batch : ns;
    const uint64_t stride = bs_mode ? (uint64_t)std::min(batch, n) + ns : ns;  // ids kept per minibatch
    const uint64_t per_epoch = (uint64_t)nb * stride;
    if (c->chunk_auto) {
        // sharded: a rank's launch covers batch/world rows, and its longest serial stretch (one chunk) has to
        // shrink with it or the step kernel stops getting shorter (measured on RMAT-20, B = 65536, 8 ranks: 87 us
        // per minibatch with the whole batch's chunk, 32 us with the slice's).  The chunk is part of the summation
        // order: results are bit-identical for any world size GIVEN the chunk ("hub_chunk" pins it).
        const uint32_t ch = auto_chunk(c, sharded ? (batch + c->push.world - 1) / c->push.world : batch);
        if (ch != c->chunk) {
            if ((rc = flush_pending(c)) != F2V_OK) return rc;
            HIPC(hipStreamSynchronize(c->stream));
            c->chunk = ch;
            drop_plans(c);
        }
    }
    // Sample ids do not depend on the embeddings: options 5/6 pre-draw every epoch's ids (as long
    // as that stays below 1 GiB); option 7 interleaves walk generation, so it goes epoch by epoch.
    const bool all_upfront = (math != 7 || c->fast_rng) && (per_epoch * iters * 4ull <= (1ull << 30));
    // small minibatches: groups of them in one launch (chain_plan_for), ordered by data dependencies instead of launch boundaries
    const bool chained = iters > 0 && chain_usable(c, math, batch, bs_mode, sharded);
    const bool wide = chained && wide_usable(c) && batch <= c->wide_max_batch && (chain_len(c, batch, true) >= 2 || c->wide_single);
    const uint32_t K = chained ? chain_len(c, batch, wide) : 1;
    uint32_t epochs_max = 1;  // "wide_epochs"
    if (wide && all_upfront && math != 7 && !bs_mode && ns <= 8 && !c->mark_every
#ifdef F2V_TEST_HOOKS
        && !c->d_stamps && !c->test_chain_mode
#endif
    ) {
        epochs_max = c->wide_epochs ? c->wide_epochs : (c->nnz <= (2ull << 20) ? 32u : 1u);
        epochs_max = (uint32_t)std::min<uint64_t>(epochs_max, (8ull << 30) / std::max<uint64_t>((uint64_t)n * c->D * sizeof(float), 1));  // the ring stays below 8 GiB
        if (2ull * n * c->D * sizeof(float) > 0xFFFFFFFFull) epochs_max = 1;  // an epoch's two matrices of the ring are read at 32-bit byte offsets (load16_agent)
    }
    c->last_wide_epochs = 1;
    c->last_train_form = 0;  // (set where a launch is made: launch_step 0 / 3 under hipGraph replay, launch_chain 1, launch_wide 2)
    if (wide) {
        wide_plans_for_epoch(c, nb, K, batch, math == 7);
    } else if (chained) {
        for (uint32_t b0 = 0; b0 < nb; b0 += K) (void)chain_plan_for(c, b0, std::min(K, nb - b0), batch, math == 7);
    } else {
        for (uint32_t b = 0; b < nb; b++) {  // all launch plans up-front: one upload, no syncs inside the timed loop
            uint32_t lo = b * batch, hi = (uint32_t)std::min<uint64_t>((uint64_t)b * batch + batch, n);
            if (sharded) shard_of(c, math == 7, lo, hi, c->push.rank, c->push.world, &lo, &hi);
            (void)plan_for(c, lo, hi, math == 7);
        }
    }
    if ((rc = upload_plans(c)) != F2V_OK) return rc;
    // (epochs whose ids cannot all be drawn up-front alternate between two device buffers: reserved HERE, once -- a second, larger
    // reservation further down would hipFree the first, and hipFree waits for every stream of the device: with the ranks of a push
    // exchange as engines of ONE process (tools/push_world_local.py) that is a peer's spinning barrier kernel, which waits for this rank)
    const uint64_t dev_ids = std::max<uint64_t>(all_upfront ? per_epoch * std::max(iters, 1u) : 2 * per_epoch, 64);
    if ((rc = reserve_ids(c, dev_ids)) != F2V_OK) return rc;
    c->ids_valid = 0;
    std::vector<uint32_t> ids;
    auto draw_epoch = [&](std::vector<uint32_t> &v, size_t off) {
        for (uint32_t b = 0; b < nb; b++) {
            uint32_t maxv = n - 1;
            if (math == 7) {  // algorithms.cpp:1125 (option 10: :2124)
                const uint64_t e = (uint64_t)(b + 1) * batch;
                if (e < maxv) maxv = (uint32_t)e;
            }
            // option 9's own rule (AlgoForce2VecNSRW_SREAL_D128_AVXZ): full minibatches draw from [0, (b+1)*BATCHSIZE)
            // (algorithms.cpp:1700-1704; reaches vertex N-1 when the batch size divides N), the tail from [0, N-1) (:1939-1941)
            if (option == 9 && b < n / batch) maxv = (b + 1) * batch;
            for (uint64_t s = 0; s < ndraw; s++) {
                const uint32_t r = c->rng.index(maxv, 0);
                if (s < stride) v[off + (size_t)b * stride + s] = r;
            }
        }
    };
    if (chained && !c->d_rowflag) {
        HIPC(hipMalloc((void **)&c->d_rowflag, (size_t)c->n * sizeof(uint32_t)));
        HIPC(hipMemsetAsync(c->d_rowflag, 0, (size_t)c->n * sizeof(uint32_t), c->stream));  // 0 is no launch's sequence number
        HIPC(hipStreamSynchronize(c->stream));
    }
    if (all_upfront) {
        ids.assign(per_epoch * iters, 0u);
        for (uint32_t it = 0; it < iters; it++) draw_epoch(ids, (size_t)it * per_epoch);
        if (!ids.empty()) HIPC(hipMemcpy(c->d_ids, ids.data(), ids.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    } else {
        ids.assign(per_epoch, 0u);
    }
    c->stats = f2v_stats{};
    // sharded: who reads which row is static when the neighbours are the CSR's and the run's sample ids are known
    // up-front; otherwise (option 7's walks, -bs 1's per-row sample windows) every new row goes to every peer
    const bool exchanging = sharded && c->push.world > 1;
    if (exchanging && c->push.landing && std::min(batch, n) > c->push.landing_cap)
        return fail(F2V_EINVAL, "f2v_train_sharded: a minibatch of %u rows does not fit the landing buffer (%u rows; matrices of 2 GiB and more are exchanged through it)", std::min(batch, n), c->push.landing_cap);
    const bool need_based = exchanging && math != 7 && !bs_mode && all_upfront;
    const uint32_t *d_masks = nullptr;
    if (need_based) {
        if ((rc = prepare_masks(c, batch, ids)) != F2V_OK) return rc;
        d_masks = c->push.d_masks;
    }
    if (sharded) c->push.rows_pushed = c->push.rows_allgather = 0;
    struct Events {  // destroyed on every way out of this function
        std::vector<hipEvent_t> all;
        int make(hipEvent_t *e, unsigned flags) {
            HIPC(hipEventCreateWithFlags(e, flags));
            all.push_back(*e);
            return F2V_OK;
        }
        ~Events() { for (hipEvent_t e : all) (void)hipEventDestroy(e); }
    } events;
    hipEvent_t ev0, ev1;
    std::vector<hipEvent_t> mark_ev;  // "epoch_marks"
    c->marks.clear();
    if ((rc = events.make(&ev0, hipEventDefault)) != F2V_OK || (rc = events.make(&ev1, hipEventDefault)) != F2V_OK) return rc;
    HIPC(hipEventRecord(ev0, c->stream));
#ifdef F2V_TEST_HOOKS
    // F2V_PUSH_CHAOS=<seed> (self-test build only): every rank stalls at random minibatches (different ones on every rank)
    unsigned long long chaos = 0;
    if (const char *e = getenv("F2V_PUSH_CHAOS")) chaos = (strtoull(e, nullptr, 10) + 1) * 0x9E3779B97F4A7C15ull + c->push.rank * 0xD1B54A32D192ED03ull;
#endif
    // The kernels' error words (a combine-tree wait that gave up) are copied to pinned memory at the end of every epoch,
    // stream-ordered, and looked at without blocking as soon as the copy has run: a lost run fails within an epoch or two
    // of the give-up instead of at its very end.  Four copies may be in flight (ring of events / 64-byte slots of h_kerr).
    constexpr int kErrRing = 4;
    uint32_t err_seq = 0;
    hipEvent_t err_ev[kErrRing];
    bool err_used[kErrRing] = {};
    for (auto &e : err_ev)
        if ((rc = events.make(&e, hipEventDisableTiming)) != F2V_OK) return rc;
    auto poll_errors = [&](bool wait_slot, int slot) -> const uint32_t * {  // -> the error words if some epoch gave up
        for (int k = 0; k < kErrRing; k++) {
            if (!err_used[k]) continue;
            if (wait_slot && k == slot) (void)hipEventSynchronize(err_ev[k]);
            else if (hipEventQuery(err_ev[k]) != hipSuccess) continue;
            err_used[k] = false;
            if (c->h_kerr[16 * k]) return c->h_kerr + 16 * k;
        }
        return nullptr;
    };
    const bool graphed = c->use_graph && math != 7 && all_upfront && iters >= 2 && !sharded;
    if (graphed) {
        c->last_train_form = 3;  // one launch per minibatch and tree level, replayed from two captured graphs
        // hipGraph replay: an epoch's launch chain is identical every epoch except for (a) which of the two matrices
        // is read and which written -- they alternate, hence one graph per epoch parity -- and (b) the sample ids,
        // which each replay finds at a fixed place (its parity's region of d_ids), refreshed by a stream-ordered copy.
        hipGraph_t graph[2] = {nullptr, nullptr};
        hipGraphExec_t exec[2] = {nullptr, nullptr};
        for (int par = 0; par < 2; par++) {
            HIPC(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            c->capturing = true;  // a replayed launch cannot carry a fresh sequence number: one launch per tree level
            for (uint32_t b = 0; b < nb && rc == F2V_OK; b++) {
                const uint32_t lo = b * batch;
                const uint32_t hi = (uint32_t)std::min<uint64_t>((uint64_t)lo + batch, n);
                rc = launch_step(c, math, lo, hi, lo, hi, c->d_ids + (size_t)par * per_epoch + (size_t)b * stride, ns, lr, bs_mode);
            }
            hipError_t e = hipStreamEndCapture(c->stream, &graph[par]);
            c->capturing = false;
            if (rc != F2V_OK) return rc;
            HIPC(e);
            HIPC(hipGraphInstantiate(&exec[par], graph[par], nullptr, nullptr, 0));
        }
        // the host-side bookkeeping now stands where two executed epochs leave it; one epoch's statistics were counted twice
        f2v_stats one = c->stats;
        one.step_launches /= 2; one.rows /= 2; one.nnz /= 2; one.algorithmic_bytes /= 2; one.hub_rows /= 2; one.hub_chunks /= 2; one.compulsory_bytes /= 2;
        HIPC(hipEventRecord(ev0, c->stream));
        for (uint32_t it = 0; it < iters; it++) {
            if (it >= 2)
                HIPC(hipMemcpyAsync(c->d_ids + (size_t)(it & 1) * per_epoch, ids.data() + (size_t)it * per_epoch, per_epoch * sizeof(uint32_t),
                                    hipMemcpyHostToDevice, c->stream));
            HIPC(hipGraphLaunch(exec[it & 1], c->stream));
        }
        if (iters & 1) c->cur ^= 1;  // an odd number of epochs ends on the other matrix than the two captured ones did
        c->stats = one;
        c->stats.step_launches *= iters; c->stats.rows *= iters; c->stats.nnz *= iters; c->stats.algorithmic_bytes *= iters; c->stats.compulsory_bytes *= iters;
        c->stats.hub_rows *= iters; c->stats.hub_chunks *= iters;
        HIPC(hipStreamSynchronize(c->stream));
        for (int par = 0; par < 2; par++) { (void)hipGraphExecDestroy(exec[par]); (void)hipGraphDestroy(graph[par]); }
    }
    // Epochs whose host-side inputs cannot all be drawn up-front (option 7: the walks are 5 N ids per epoch, and the one serial
    // rand() stream interleaves them with the sample ids): a producer thread draws epoch e+1's walks and ids -- they never depend
    // on the embeddings (sample/algorithms.cpp:1097-1132) -- into pinned buffers while the device runs epoch e; the copies are
    // stream-ordered asynchronous copies into alternating device buffers, and nothing in the loop waits for the device.
    // Epoch time = max(host generation, device), instead of their sum plus two synchronisations.
    struct HostEpoch {
        uint32_t *walks = nullptr, *ids = nullptr;  // pinned
        hipEvent_t copied = nullptr;                // the H2D copies out of this slot have completed
        bool full = false, copy_pending = false;
    } slot[2];
    std::mutex mu;
    std::condition_variable cv;
    std::thread producer;
    bool stop_producer = false;
    const bool host_walks = (math == 7 && !c->fast_rng);
    const bool produced = !all_upfront && !graphed && iters > 0;
    uint32_t *d_walk_buf[2] = {nullptr, nullptr};
    auto end_producer = [&] {
        if (producer.joinable()) {
            { std::lock_guard<std::mutex> lk(mu); stop_producer = true; }
            cv.notify_all();
            producer.join();
        }
        if (slot[0].copy_pending || slot[1].copy_pending) (void)hipStreamSynchronize(c->stream);  // no copy still reads the pinned buffers
        for (auto &sl : slot) {
            if (sl.walks) (void)hipHostFree(sl.walks);
            if (sl.ids) (void)hipHostFree(sl.ids);
            if (sl.copied) (void)hipEventDestroy(sl.copied);
            sl = HostEpoch{};
        }
    };
    struct AtExit {
        std::function<void()> f;
        ~AtExit() { f(); }
    } producer_guard{end_producer};  // every return below joins the thread and frees the pinned buffers
    if (produced) {
        const size_t nwalk = (size_t)n * kWalkLength;
        if (host_walks) {
            if (!c->d_walks) HIPC(hipMalloc((void **)&c->d_walks, nwalk * sizeof(uint32_t)));
            if (!c->d_walks_alt) HIPC(hipMalloc((void **)&c->d_walks_alt, nwalk * sizeof(uint32_t)));
            d_walk_buf[0] = c->d_walks;
            d_walk_buf[1] = c->d_walks_alt;
        }
        if ((rc = reserve_ids(c, std::max<uint64_t>(2 * per_epoch, 64))) != F2V_OK) return rc;
        for (auto &sl : slot) {
            if (host_walks) HIPC(hipHostMalloc((void **)&sl.walks, nwalk * sizeof(uint32_t), hipHostMallocDefault));
            HIPC(hipHostMalloc((void **)&sl.ids, std::max<size_t>(per_epoch, 1) * sizeof(uint32_t), hipHostMallocDefault));
            HIPC(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming));
        }
        const int dev = c->device;
        producer = std::thread([&, dev] {
            (void)hipSetDevice(dev);
            std::vector<uint32_t> w, v(per_epoch);
            for (uint32_t it = 0; it < iters; it++) {
                HostEpoch &sl = slot[it & 1];
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop_producer || !sl.full; });
                    if (stop_producer) return;
                }
                if (sl.copy_pending) { (void)hipEventSynchronize(sl.copied); sl.copy_pending = false; }  // the slot's last copies have left it
                if (host_walks) {  // the reference's order: the epoch's walks, then its minibatches' sample ids
                    generate_walks_host(c, w);
                    memcpy(sl.walks, w.data(), w.size() * sizeof(uint32_t));
                }
                draw_epoch(v, 0);
                if (per_epoch) memcpy(sl.ids, v.data(), per_epoch * sizeof(uint32_t));
                { std::lock_guard<std::mutex> lk(mu); sl.full = true; }
                cv.notify_all();
            }
        });
    }
    for (uint32_t it = 0; it < iters && !graphed; it++) {
        if (math == 7 && c->fast_rng) {
            if ((rc = fast_walks(c)) != F2V_OK) return rc;  // stream-ordered: no host work, no synchronisation
        }
        const uint32_t *d_epoch_ids = c->d_ids + (all_upfront ? (size_t)it * per_epoch : (size_t)(it & 1) * per_epoch);
        if (produced) {
            HostEpoch &sl = slot[it & 1];
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return sl.full; });
            }
            // stream-ordered behind epoch it-2's kernels, the last readers of these device buffers
            hipError_t he = hipSuccess;
            if (host_walks) {
                he = hipMemcpyAsync(d_walk_buf[it & 1], sl.walks, (size_t)n * kWalkLength * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
                c->d_walks = d_walk_buf[it & 1];
                c->d_walks_alt = d_walk_buf[(it & 1) ^ 1];
                c->have_walks = true;
            }
            if (he == hipSuccess && per_epoch)
                he = hipMemcpyAsync(c->d_ids + (size_t)(it & 1) * per_epoch, sl.ids, per_epoch * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream);
            if (he == hipSuccess) he = hipEventRecord(sl.copied, c->stream);
            {
                std::lock_guard<std::mutex> lk(mu);
                sl.copy_pending = true;
                sl.full = false;
            }
            cv.notify_all();
            HIPC(he);
        }
        if (wide) {
            for (uint32_t b0 = 0; b0 < nb; b0 += K) {
                const WidePlan plan = wide_plan_for(c, b0, std::min(K, nb - b0), batch, math == 7);
                if ((rc = upload_plans(c)) != F2V_OK) return rc;
                // epochs chained in one launch: the plan covers the graph, every epoch's sample ids are on the device, no walks, no marks
                uint32_t E = 1;
                if (epochs_max > 1 && K >= nb && plan.n_node_wgs == 0 && !c->ring_refused) E = std::min(epochs_max, iters - it);
                if ((rc = launch_wide(c, math, plan, d_epoch_ids, (uint32_t)stride, ns, lr, bs_mode, &E, per_epoch)) != F2V_OK) return rc;
                it += E - 1;
            }
        } else if (chained) {
            for (uint32_t b0 = 0; b0 < nb; b0 += K) {
                const ChainPlan plan = chain_plan_for(c, b0, std::min(K, nb - b0), batch, math == 7);
                if ((rc = upload_plans(c)) != F2V_OK) return rc;  // O(1) unless the plan cache was dropped meanwhile
                if ((rc = launch_chain(c, math, plan, d_epoch_ids, (uint32_t)stride, ns, lr, bs_mode)) != F2V_OK) return rc;
            }
        }
        for (uint32_t b = 0; b < nb && !chained; b++) {
            const uint32_t lo = b * batch;
            const uint32_t hi = (uint32_t)std::min<uint64_t>((uint64_t)lo + batch, n);
            uint32_t my_lo = lo, my_hi = hi;
            if (sharded) shard_of(c, math == 7, lo, hi, c->push.rank, c->push.world, &my_lo, &my_hi);
#ifdef F2V_TEST_HOOKS
            if (exchanging && chaos) {  // random host-side stalls skew the ranks against each other
                chaos = chaos * 6364136223846793005ull + 1442695040888963407ull;
                if (((chaos >> 33) & 7u) == 0u) {
                    HIPC(hipStreamSynchronize(c->stream));
                    std::this_thread::sleep_for(std::chrono::microseconds((chaos >> 40) % 3000u));
                }
            }
#endif
            if ((rc = launch_step(c, math, lo, hi, my_lo, my_hi, d_epoch_ids + (size_t)b * stride, ns, lr, bs_mode, exchanging && c->push.fused, d_masks)) != F2V_OK) return rc;
            if (exchanging) {
                if (!c->push.fused && (rc = launch_push(c, c->cur ^ 1, d_masks, lo, my_lo, my_hi)) != F2V_OK) return rc;
                if ((rc = finish_exchange(c, c->cur ^ 1, d_masks, lo, hi, my_lo, my_hi)) != F2V_OK) return rc;
                c->push.rows_allgather += (uint64_t)(my_hi - my_lo) * (c->push.world - 1);
            }
        }
        if (exchanging) c->push.rows_pushed += need_based ? c->push.pushed_per_epoch : (uint64_t)0;
        if (c->mark_every && (it + 1) % c->mark_every == 0 && mark_ev.size() < 4096) {
            hipEvent_t e;
            if ((rc = events.make(&e, hipEventDefault)) != F2V_OK) return rc;
            HIPC(hipEventRecord(e, c->stream));
            mark_ev.push_back(e);
        }
        if (c->merge_fin) {
            const int slot = (int)(err_seq++ % kErrRing);  // (by copies made, not by epoch: a launch may carry 32 epochs)
            const uint32_t *bad = poll_errors(err_used[slot], slot);  // the slot about to be reused is waited for (4 epochs old)
            if (!bad) {
                HIPC(hipMemcpyAsync(c->h_kerr + 16 * slot, c->d_kerr, 64, hipMemcpyDeviceToHost, c->stream));
                HIPC(hipEventRecord(err_ev[slot], c->stream));
                err_used[slot] = true;
            } else {
                uint32_t e[16];
                memcpy(e, bad, sizeof e);
                char where[96];
                snprintf(where, sizeof where, "%s (noticed after epoch %u of %u)", sharded ? "f2v_train_sharded" : "f2v_train", it + 1, iters);
                return kernel_gave_up(c, where, e);
            }
        }
    }
    if (exchanging && !need_based) c->push.rows_pushed = c->push.rows_allgather;
    if ((rc = flush_pending(c)) != F2V_OK) return rc;
    if (need_based && iters > 0) {
        // replicas are complete only in the rows their rank reads: one full exchange makes them whole
        for (uint32_t b = 0; b < nb; b++) {
            const uint32_t lo = b * batch, hi = (uint32_t)std::min<uint64_t>((uint64_t)lo + batch, n);
            uint32_t my_lo, my_hi;
            shard_of(c, math == 7, lo, hi, c->push.rank, c->push.world, &my_lo, &my_hi);
            if ((rc = launch_push(c, c->cur, nullptr, lo, my_lo, my_hi)) != F2V_OK) return rc;
            // the landing buffer holds one minibatch: an exchange per minibatch; mapped matrices take them all at once
            if (c->push.landing && (rc = finish_exchange(c, c->cur, nullptr, lo, hi, my_lo, my_hi)) != F2V_OK) return rc;
        }
        if (!c->push.landing && (rc = launch_barrier(c)) != F2V_OK) return rc;
    }
    HIPC(hipEventRecord(ev1, c->stream));
    HIPC(hipEventSynchronize(ev1));
    float ms = 0.f;
    HIPC(hipEventElapsedTime(&ms, ev0, ev1));
    c->stats.device_seconds = ms * 1e-3;
    if (seconds_out) *seconds_out = ms * 1e-3;
    for (hipEvent_t e : mark_ev) {
        float mm = 0.f;
        HIPC(hipEventElapsedTime(&mm, ev0, e));
        c->marks.push_back(mm * 1e-3);
    }
    if ((rc = check_kernel_err(c, sharded ? "f2v_train_sharded" : "f2v_train")) != F2V_OK) return rc;
    if (exchanging) return check_push_err(c, "f2v_train_sharded");
    return F2V_OK;
}

}  // namespace

extern "C" {

int f2v_push_export(f2v_handle c, void *handles_out) {
    if (!c || !handles_out) return fail(F2V_EINVAL, "f2v_push_export: null argument");
    HIPC(hipSetDevice(c->device));
    if (c->push.attached) return fail(F2V_ESTATE, "f2v_push_export: detach first");
    int rc = flush_pending(c);
    if (rc != F2V_OK) return rc;
    HIPC(hipStreamSynchronize(c->stream));
    // fresh, zeroed flags for every attachment: no peer can write them before it has seen this export
    if (c->push.flags) (void)hipFree(c->push.flags);
    c->push.flags = nullptr;
    HIPC(hipExtMallocWithFlags((void **)&c->push.flags, 4096, hipDeviceMallocFinegrained));
    HIPC(hipMemset(c->push.flags, 0, 4096));
    if (!c->push.d_err) HIPC(hipMalloc((void **)&c->push.d_err, 64));
    HIPC(hipMemset(c->push.d_err, 0, 64));
    HIPC(hipDeviceSynchronize());
    c->push.seq = 0;
    c->push.round = 0;
    PushExport e{};
    const size_t matrix_bytes = ((size_t)c->n + kPadRows) * c->D * sizeof(float);
    // What HIP IPC can map.  Evidence (tools/ipc_probe.py, ROCm 7.2, dmabuf IPC, HSA_ENABLE_IPC_MODE_LEGACY=0): an allocation one
    // row below 2^31 bytes attaches in a millisecond, one row above 2^31 bytes never returns from hipIpcOpenMemHandle -- the
    // step is at exactly 2^31 BYTES OF ALLOCATION SIZE, whatever the row count, padding or matrix contents, and the 4-KiB
    // fine-grained flag array and the landing buffers (<= 1 GiB) always attach: a signed 32-bit size on the import path of the
    // IPC handle, not a property of peer access or of lazy enabling (the same flag maps the small allocations).  So the
    // threshold IS that number; tests/test_gpu_large.py attaches one matrix just below it directly and one just above it
    // through the landing buffer.  Only ever observed with both processes on one GPU (no multi-GPU box was available).
    c->push.landing = c->push.force_landing || matrix_bytes >= kIpcMaxBytes;
    if (c->push.landing) {
        // two halves of at most 512 MiB; room for the self-test's `world` pattern rows in any case
        const size_t per_row = (size_t)c->D * sizeof(float);
        c->push.landing_cap = (uint32_t)std::max<size_t>(std::min<size_t>(((size_t)512 << 20) / per_row, (size_t)c->n + kPadRows), 64);
        if (c->push.landing_buf) (void)hipFree(c->push.landing_buf);
        c->push.landing_buf = nullptr;
        HIPC(hipMalloc((void **)&c->push.landing_buf, 2 * (size_t)c->push.landing_cap * per_row));
        HIPC(hipIpcGetMemHandle(&e.x[0], c->push.landing_buf));
    } else {
        for (int k = 0; k < 2; k++) HIPC(hipIpcGetMemHandle(&e.x[k], c->d_X[k]));
    }
    HIPC(hipIpcGetMemHandle(&e.flags, c->push.flags));
    e.magic = kPushMagic;
    e.n = c->n;
    e.D = c->D;
    e.cur = (uint32_t)c->cur;
    e.landing = c->push.landing ? 1u : 0u;
    e.landing_cap = c->push.landing_cap;
    memset(e.bus_id, 0, sizeof e.bus_id);
    {
        char bus[64] = {};
        if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, c->device) == hipSuccess) strncpy(e.bus_id, bus, sizeof e.bus_id - 1);
    }
    e.caps = c->xcc_round_robin ? 1u : 0u;
    memcpy(handles_out, &e, sizeof e);
    c->push.exported = true;
    return F2V_OK;
}

int f2v_push_attach(f2v_handle c, uint32_t rank, uint32_t world, const void *all_handles) {
    if (!c || !all_handles) return fail(F2V_EINVAL, "f2v_push_attach: null argument");
    if (world == 0 || world > (uint32_t)kMaxRanks || rank >= world) return fail(F2V_EINVAL, "f2v_push_attach: rank %u of %u (at most %d ranks)", rank, world, kMaxRanks);
    if (!c->push.exported) return fail(F2V_ESTATE, "f2v_push_attach: f2v_push_export first");
    if (c->push.attached) return fail(F2V_ESTATE, "f2v_push_attach: already attached");
    HIPC(hipSetDevice(c->device));
    const PushExport *all = static_cast<const PushExport *>(all_handles);
    for (uint32_t r = 0; r < world; r++) {
        PushExport e;
        memcpy(&e, all + r, sizeof e);
        if (e.magic != kPushMagic) return fail(F2V_EINVAL, "f2v_push_attach: export of rank %u is not an export", r);
        if (e.n != c->n || e.D != c->D || e.cur != (uint32_t)c->cur)
            return fail(F2V_ESTATE, "f2v_push_attach: rank %u holds a different engine state (n %u dim %u matrix %u; here %u %u %d)", r, e.n, e.D, e.cur, c->n, c->D, c->cur);
        if ((e.landing != 0) != c->push.landing || (c->push.landing && e.landing_cap != c->push.landing_cap))
            return fail(F2V_ESTATE, "f2v_push_attach: rank %u exchanges through %s, this rank through %s (\"push_landing\" must agree)", r,
                        e.landing ? "a landing buffer" : "mapped matrices", c->push.landing ? "a landing buffer" : "mapped matrices");
    }
    {
        PushExport me;
        memcpy(&me, all + rank, sizeof me);
        bool shared = false;
        for (uint32_t r = 0; r < world; r++) {
            PushExport e;
            memcpy(&e, all + r, sizeof e);
            if (r != rank && (me.bus_id[0] == 0 || !memcmp(e.bus_id, me.bus_id, sizeof me.bus_id))) shared = true;
        }
        if (shared != c->shared_card) {  // placement of the hub pieces depends on it
            c->shared_card = shared;
            drop_plans(c);
        }
        // what every rank must decide alike ("replicate_small"): do any two ranks share a card, can every rank chain
        c->push.any_shared = false;
        c->push.all_chain = true;
        for (uint32_t r = 0; r < world; r++) {
            PushExport e;
            memcpy(&e, all + r, sizeof e);
            if (!(e.caps & 1u)) c->push.all_chain = false;
            for (uint32_t q = r + 1; q < world; q++) {
                PushExport f;
                memcpy(&f, all + q, sizeof f);
                if (e.bus_id[0] == 0 || !memcmp(e.bus_id, f.bus_id, sizeof e.bus_id)) c->push.any_shared = true;
            }
        }
    }
    c->push.rank = rank;
    c->push.world = world;
    c->push.attached = true;  // from here on push_detach unmaps whatever got mapped
    for (uint32_t r = 0; r < world; r++) {
        if (r == rank) {
            for (int k = 0; k < 2; k++) c->push.peer_X[k][r] = c->d_X[k];
            c->push.peer_flags[r] = c->push.flags;
            c->push.peer_landing[r] = c->push.landing_buf;
            continue;
        }
        PushExport e;
        memcpy(&e, all + r, sizeof e);
        if (c->push.landing) {
            hipError_t he = hipIpcOpenMemHandle((void **)&c->push.peer_landing[r], e.x[0], hipIpcMemLazyEnablePeerAccess);
            if (he != hipSuccess) {
                c->push.peer_landing[r] = nullptr;
                (void)push_detach(c);
                return fail(F2V_ENODEV, "f2v_push_attach: cannot map the landing buffer of rank %u: %s", r, hipGetErrorString(he));
            }
        }
        for (int k = 0; k < 2 && !c->push.landing; k++) {
            hipError_t he = hipIpcOpenMemHandle((void **)&c->push.peer_X[k][r], e.x[k], hipIpcMemLazyEnablePeerAccess);
            if (he != hipSuccess) {
                c->push.peer_X[k][r] = nullptr;
                (void)push_detach(c);
                return fail(F2V_ENODEV, "f2v_push_attach: cannot map the matrix of rank %u: %s", r, hipGetErrorString(he));
            }
        }
        hipError_t he = hipIpcOpenMemHandle((void **)&c->push.peer_flags[r], e.flags, hipIpcMemLazyEnablePeerAccess);
        if (he != hipSuccess) {
            c->push.peer_flags[r] = nullptr;
            (void)push_detach(c);
            return fail(F2V_ENODEV, "f2v_push_attach: cannot map the flags of rank %u: %s", r, hipGetErrorString(he));
        }
    }
    c->push.mask_batch = 0;  // masks depend on (rank, world)
    c->push.exported = false;
    return F2V_OK;
}

#ifdef F2V_TEST_HOOKS
int f2v_test_withhold_flag(f2v_handle c, uint32_t slot) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    c->test_withhold_slot = slot;
    return F2V_OK;
}

int f2v_test_withhold_row(f2v_handle c, uint32_t row) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    c->test_withhold_row = row;
    return F2V_OK;
}

int f2v_test_chain_nowait(f2v_handle c, int on) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    c->test_chain_nowait = (on & 1) != 0;
    c->test_chain_mode = (uint32_t)on;
    return F2V_OK;
}

// Host-only check of the wide form's launch plans (no device is touched): builds the programs of one epoch for a graph and a
// set of tunables exactly as f2v_train would and verifies what the kernel relies on --
//   * every row is finished exactly once (by a whole-row item, a row job, or the root of a combine tree), every neighbour of a
//     split row belongs to exactly one piece, first / last pieces are flagged as such;
//   * a workgroup's items fill whole rounds, the items of a round agree on whether it ends a phase, piece slots and job ranges
//     stay inside the LDS slot space, passes hold at most 8 jobs, a job adds consecutive pieces of ONE row in neighbour order;
//   * WAITS ONLY POINT BACKWARDS: a helper's group sum is produced by a workgroup with a smaller index than the finisher that
//     imports it, every sum a combine-tree node adds by a smaller index than the node's, minibatches appear in order.
// stats_out[0..6]: workgroups, helpers, finishers, packed workgroups, node workgroups, partial-sum slots, a checksum of the plan arrays.
int f2v_test_wide_plan_check(const uint32_t *rowptr, const uint32_t *colids, uint32_t n, uint64_t nnz, uint32_t dim, uint32_t batch, int walk,
                             const char *const *names, const int64_t *values, uint32_t n_params, uint64_t *stats_out) {
    if (!rowptr || (!colids && nnz) || n < 2 || dim == 0 || batch == 0) return fail(F2V_EINVAL, "f2v_test_wide_plan_check: bad argument");
    f2v_ctx ctx;
    f2v_ctx *c = &ctx;
    unsigned plan_threads = 0;
    c->n = n; c->nnz = nnz; c->D = dim;
    c->rowptr.assign(rowptr, rowptr + n + 1);
    c->colids.assign(colids, colids + nnz);
    c->chunk = auto_chunk(c, batch);
    for (uint32_t k = 0; k < n_params; k++) {
        const std::string nm = names[k];
        const int64_t v = values[k];
        if (nm == "hub_chunk") c->chunk = (uint32_t)v;
        else if (nm == "hub_fanin") c->fanin = (uint32_t)v;
        else if (nm == "class_cut") c->class_cut = v != 0;
        else if (nm == "wide_phases") c->wide_phases = (uint32_t)v;
        else if (nm == "wide_rounds") c->wide_rounds = (uint32_t)v;
        else if (nm == "wide_span") c->wide_span = (uint32_t)v;
        else if (nm == "wide_finish") c->wide_finish = (uint32_t)v;
        else if (nm == "wide_order") c->wide_order = (uint32_t)v;
        else if (nm == "wide_rows") c->wide_rows = (uint32_t)v;
        else if (nm == "wide_min_width") c->wide_min_width = (uint32_t)v;
        else if (nm == "plan_threads") plan_threads = (unsigned)v;  // build the epoch's plans on this many host threads (wide_plans_for_epoch)
        else return fail(F2V_EINVAL, "f2v_test_wide_plan_check: unknown parameter '%s'", nm.c_str());
    }
    if (!wide_usable(c) || subwave_width(c) == 0) return fail(F2V_EINVAL, "f2v_test_wide_plan_check: the wide form does not run this shape");
    const uint32_t nb = (uint32_t)(((uint64_t)n + batch - 1) / batch), K = chain_len(c, batch, true);
    const uint32_t width = wide_width(c, batch), ipb = wide_items_per_block(width), pslots = std::max<uint32_t>(ipb, 32u), F = c->fanin;
    std::vector<uint32_t> finished(n, 0), covered(nnz, 0);
    uint64_t st[7] = {};
    if (plan_threads) wide_plans_for_epoch(c, nb, K, batch, walk != 0, plan_threads);
#define F2V_PLAN_FAIL(...) return fail(F2V_ESTATE, "f2v_test_wide_plan_check: " __VA_ARGS__)
    for (uint32_t b0 = 0; b0 < nb; b0 += K) {
        const WidePlan p = wide_plan_for(c, b0, std::min(K, nb - b0), batch, walk != 0);
        if (c->plan_overflow) F2V_PLAN_FAIL("slot overflow");
        st[0] += p.n_wgs; st[1] += p.n_helpers; st[2] += p.n_finishers; st[3] += p.n_packed; st[4] += p.n_node_wgs; st[5] = std::max<uint64_t>(st[5], p.n_slots);
        // partial-sum slot -> who announces it: 4 * workgroup + wavefront for a tree node (one node per wavefront: a node may wait for an
        // earlier wavefront of its own workgroup -- they are resident together), 4 * workgroup + 3 for a job (its consumer must be a LATER workgroup)
        std::vector<int64_t> producer(p.n_slots + 1, -1);
        uint32_t last_lo = 0;
        // pass 1: who produces which slot
        for (uint32_t wgi = 0; wgi < p.n_wgs; wgi++) {
            const WideDesc &d = c->h_wide[p.wg_off + wgi];
            if (d.kind == 0) {
                for (uint32_t j = 0; j < d.d; j++) {
                    const WJob &jb = c->h_jobs[p.job_off + d.c + j];
                    if (jb.kind == kJobPart) {
                        if (jb.dst >= p.n_slots || producer[jb.dst] != -1) F2V_PLAN_FAIL("partial-sum slot %u produced twice or out of range", jb.dst);
                        producer[jb.dst] = 4 * (int64_t)wgi + 3;
                    }
                }
            } else {
                for (uint32_t w = 0; w < 4 && d.c * 4 + w < d.b; w++) {
                    const FinItem &h = c->h_hubs[p.fin_off + d.a + d.c * 4 + w];
                    if (h.n == 0) continue;
                    if (h.out != kFinToStage) {
                        if (h.out >= p.n_slots || producer[h.out] != -1) F2V_PLAN_FAIL("tree node output slot %u produced twice or out of range", h.out);
                        producer[h.out] = 4 * (int64_t)wgi + w;
                    }
                }
            }
        }
        // pass 2: every workgroup
        for (uint32_t wgi = 0; wgi < p.n_wgs; wgi++) {
            const WideDesc &d = c->h_wide[p.wg_off + wgi];
            if (d.lo < last_lo || d.lo < p.lo || d.lo >= p.hi) F2V_PLAN_FAIL("workgroup %u: minibatch at row %u out of order", wgi, d.lo);
            last_lo = d.lo;
            const uint32_t mb_hi = (uint32_t)std::min<uint64_t>((uint64_t)d.lo + batch, n);
            if (d.kind == 1) {
                for (uint32_t w = 0; w < 4 && d.c * 4 + w < d.b; w++) {
                    const FinItem &h = c->h_hubs[p.fin_off + d.a + d.c * 4 + w];
                    if (h.n == 0) continue;
                    for (uint32_t k = 0; k < h.n; k++)
                        if (h.in_slot + k >= p.n_slots || producer[h.in_slot + k] < 0 || producer[h.in_slot + k] >= 4 * (int64_t)wgi + w)
                            F2V_PLAN_FAIL("tree node of row %u (workgroup %u) adds slot %u whose producer is not an earlier workgroup (or an earlier wavefront of its own)", h.row, wgi, h.in_slot + k);
                    if (h.out == kFinToStage) {
                        if (h.row < d.lo || h.row >= mb_hi) F2V_PLAN_FAIL("tree root of row %u outside its minibatch", h.row);
                        finished[h.row]++;
                    }
                }
                continue;
            }
            const Item *items = c->h_items.data() + p.item_off + d.a;
            const WJob *jobs = c->h_jobs.data() + p.job_off + d.c;
            if (d.b == 0) F2V_PLAN_FAIL("workgroup %u has no round", wgi);
            // phases: rounds up to a round whose items carry kItemPhaseEnd
            std::vector<std::vector<const Item *>> phase_items(1);
            for (uint32_t r = 0; r < d.b; r++) {
                const bool end = (items[(size_t)r * ipb].flags & kItemPhaseEnd) != 0;
                for (uint32_t q = 0; q < ipb; q++) {
                    const Item &it = items[(size_t)r * ipb + q];
                    if (((it.flags & kItemPhaseEnd) != 0) != end) F2V_PLAN_FAIL("workgroup %u round %u: items disagree about the end of the phase", wgi, r);
                    if (it.flags & kItemIdle) continue;
                    if (it.row < d.lo || it.row >= mb_hi) F2V_PLAN_FAIL("workgroup %u: row %u outside its minibatch [%u,%u)", wgi, it.row, d.lo, mb_hi);
                    if (!walk) {
                        if (it.nb < rowptr[it.row] || it.nb + it.cnt > rowptr[it.row + 1]) F2V_PLAN_FAIL("row %u: piece outside the row's neighbours", it.row);
                        for (uint32_t e = 0; e < it.cnt; e++) covered[it.nb + e]++;
                        if (((it.flags & kItemFirst) != 0) != (it.nb == rowptr[it.row]) || ((it.flags & kItemLast) != 0) != (it.nb + it.cnt == rowptr[it.row + 1]))
                            F2V_PLAN_FAIL("row %u: first / last flags of a piece are wrong", it.row);
                    }
                    if (it.flags & kItemDirect) finished[it.row]++;
                    else if ((it.flags & kItemPieceSlot) >= pslots) F2V_PLAN_FAIL("row %u: piece slot out of range", it.row);
                    phase_items.back().push_back(&it);
                }
                if (end && r + 1 < d.b) phase_items.emplace_back();
                if (!end && r + 1 == d.b) F2V_PLAN_FAIL("workgroup %u: the last round does not end a phase", wgi);
            }
            uint32_t last_phase = 0;
            bool before_ok = true;
            for (uint32_t j = 0; j < d.d; j++) {
                const WJob &jb = jobs[j];
                if (jb.pass_len == 0 || jb.pass_len > 8) F2V_PLAN_FAIL("workgroup %u: a pass of %u jobs", wgi, jb.pass_len);
                if (jb.phase == kJobBefore) { if (!before_ok) F2V_PLAN_FAIL("workgroup %u: a before-the-first-round job behind a phase job", wgi); }
                else {
                    before_ok = false;
                    if (jb.phase < last_phase || jb.phase >= phase_items.size()) F2V_PLAN_FAIL("workgroup %u: job phases out of order", wgi);
                    last_phase = jb.phase;
                }
                if (jb.kind == kJobImport) {
                    if (jb.dst < pslots || jb.dst >= pslots + kWideSumSlots) F2V_PLAN_FAIL("workgroup %u: an import lands outside the sum slots", wgi);
                    if (jb.row >= p.n_slots || producer[jb.row] < 0 || producer[jb.row] >= 4 * (int64_t)wgi)
                        F2V_PLAN_FAIL("workgroup %u imports slot %u whose producer is not an earlier workgroup", wgi, jb.row);
                    continue;
                }
                if (jb.n == 0 || (uint32_t)jb.src + jb.n > pslots + kWideSumSlots) F2V_PLAN_FAIL("workgroup %u: a job's slots are out of range", wgi);
                // piece-slot ranges: consecutive pieces of ONE row, in neighbour order, all in the job's phase
                auto check_pieces = [&](uint32_t first, uint32_t cnt, uint32_t row) -> bool {
                    if (first + cnt > pslots || jb.phase == kJobBefore) return false;
                    uint32_t next_nb = 0;
                    for (uint32_t k = 0; k < cnt; k++) {
                        const Item *hit = nullptr;
                        for (const Item *it : phase_items[jb.phase])
                            if (!(it->flags & kItemDirect) && (it->flags & kItemPieceSlot) == first + k) { if (hit) return false; hit = it; }
                        if (!hit || hit->row != row || (k && hit->nb != next_nb)) return false;
                        next_nb = hit->nb + hit->cnt;
                    }
                    return true;
                };
                if (jb.n2 != 0) {
                    if (jb.dst2 < jb.src || jb.dst2 >= (uint32_t)jb.src + jb.n || jb.pass_len != 1) F2V_PLAN_FAIL("workgroup %u: a fused job's first sum does not land among its second's slots", wgi);
                    if (!walk && !check_pieces(jb.src2, jb.n2, jb.row)) F2V_PLAN_FAIL("row %u: a fused job's pieces are not consecutive pieces of the row", jb.row);
                }
                if (jb.src < pslots) {
                    if (jb.n > F) F2V_PLAN_FAIL("row %u: a job adds %u pieces, more than the fan-in", jb.row, jb.n);
                    if (!walk && !check_pieces(jb.src, jb.n, jb.row)) F2V_PLAN_FAIL("row %u: a job's pieces are not consecutive pieces of the row", jb.row);
                }
                if (jb.kind == kJobLds && (jb.dst < pslots || jb.dst >= pslots + kWideSumSlots)) F2V_PLAN_FAIL("workgroup %u: a group sum lands outside the sum slots", wgi);
                if (jb.kind == kJobRow) {
                    if (jb.row < d.lo || jb.row >= mb_hi) F2V_PLAN_FAIL("row job of row %u outside its minibatch", jb.row);
                    finished[jb.row]++;
                }
            }
        }
    }
#undef F2V_PLAN_FAIL
    for (uint32_t i = 0; i < n; i++)
        if (finished[i] != 1) return fail(F2V_ESTATE, "f2v_test_wide_plan_check: row %u is finished %u times", i, finished[i]);
    if (!walk)
        for (uint64_t e = 0; e < nnz; e++)
            if (covered[e] != 1) return fail(F2V_ESTATE, "f2v_test_wide_plan_check: neighbour %llu is in %u pieces", (unsigned long long)e, covered[e]);
    {   // [6]: a checksum of the resident plan arrays (items, jobs, descriptors, tree nodes): the threaded build leaves the serial build's
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](const void *p, size_t bytes) {
            const unsigned char *q = static_cast<const unsigned char *>(p);
            for (size_t k = 0; k < bytes; k++) h = (h ^ q[k]) * 1099511628211ull;
        };
        mix(c->h_items.data(), c->h_items.size() * sizeof(Item));
        mix(c->h_jobs.data(), c->h_jobs.size() * sizeof(WJob));
        mix(c->h_wide.data(), c->h_wide.size() * sizeof(WideDesc));
        mix(c->h_hubs.data(), c->h_hubs.size() * sizeof(FinItem));
        st[6] = h;
    }
    if (stats_out) memcpy(stats_out, st, sizeof st);
    return F2V_OK;
}

int f2v_test_stamps(f2v_handle c, int on, unsigned long long *out) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    HIPC(hipSetDevice(c->device));
    HIPC(hipStreamSynchronize(c->stream));
    const size_t bytes = 4 * (size_t)c->n * sizeof(unsigned long long);
    if (out) {
        if (!c->d_stamps) return fail(F2V_ESTATE, "f2v_test_stamps: not switched on");
        HIPC(hipMemcpy(out, c->d_stamps, bytes, hipMemcpyDeviceToHost));
    }
    if (on) {
        if (!c->d_stamps) HIPC(hipMalloc((void **)&c->d_stamps, bytes));
        HIPC(hipMemsetAsync(c->d_stamps, 0, bytes, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
    } else if (c->d_stamps) {
        (void)hipFree(c->d_stamps);
        c->d_stamps = nullptr;
    }
    return F2V_OK;
}

int f2v_test_push_attach_local(f2v_handle c, uint32_t rank, uint32_t world, const f2v_handle *all) {
    if (!c || !all) return fail(F2V_EINVAL, "f2v_test_push_attach_local: null argument");
    if (world == 0 || world > (uint32_t)kMaxRanks || rank >= world || all[rank] != c) return fail(F2V_EINVAL, "f2v_test_push_attach_local: rank %u of %u", rank, world);
    if (!c->push.exported) return fail(F2V_ESTATE, "f2v_test_push_attach_local: f2v_push_export first (it creates the flags)");
    if (c->push.attached) return fail(F2V_ESTATE, "f2v_test_push_attach_local: already attached");
    for (uint32_t r = 0; r < world; r++) {
        if (!all[r] || !all[r]->push.flags || all[r]->n != c->n || all[r]->D != c->D || all[r]->cur != c->cur || all[r]->device != c->device)
            return fail(F2V_ESTATE, "f2v_test_push_attach_local: peer %u is not an exported engine of the same shape on the same device", r);
        for (int k = 0; k < 2; k++) c->push.peer_X[k][r] = all[r]->d_X[k];
        c->push.peer_flags[r] = all[r]->push.flags;
        c->push.peer_landing[r] = all[r]->push.landing_buf;
        if (all[r]->push.landing != c->push.landing) return fail(F2V_ESTATE, "f2v_test_push_attach_local: \"push_landing\" must agree");
    }
    if (!c->shared_card && world > 1) { c->shared_card = true; drop_plans(c); }
    c->push.rank = rank;
    c->push.world = world;
    c->push.attached = c->push.local = true;
    c->push.mask_batch = 0;
    c->push.exported = false;
    return F2V_OK;
}

#endif  // F2V_TEST_HOOKS

int f2v_push_detach(f2v_handle c) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    HIPC(hipSetDevice(c->device));
    return push_detach(c);
}

int f2v_push_selftest(f2v_handle c) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    if (!c->push.attached) return fail(F2V_ESTATE, "f2v_push_selftest: f2v_push_attach first");
    HIPC(hipSetDevice(c->device));
    int rc = flush_pending(c);
    if (rc != F2V_OK) return rc;
    const uint32_t W = c->push.world, me = c->push.rank, D = c->D;
    const int which = c->cur ^ 1;  // the slack rows behind row N of the matrix no epoch is reading
    // a pattern nobody could hold by accident: depends on the rank and on how many barriers have passed (the same
    // number on every rank)
    const unsigned long long nonce = c->push.seq;
    auto value = [&](uint32_t r, uint32_t d) { return (float)(1 + r) * 1000.0f + (float)(nonce % 977) + (float)d / 1024.0f; };
    std::vector<float> row(D);
    for (uint32_t d = 0; d < D; d++) row[d] = value(me, d);
    HIPC(hipMemcpyAsync(c->d_X[which] + (size_t)(c->n + me) * D, row.data(), D * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    if ((rc = launch_barrier(c)) != F2V_OK) return rc;  // nobody pushes before everybody has laid its pattern down
    // as a "minibatch" [n, n+W) of which this rank computed row n+me: lands in the peers' slack rows, or in slot `me` of
    // their landing buffers, from where the exchange's second half moves it to the slack rows
    if ((rc = launch_push(c, which, nullptr, c->n, c->n + me, c->n + me + 1)) != F2V_OK) return rc;
    if ((rc = finish_exchange(c, which, nullptr, c->n, c->n + W, c->n + me, c->n + me + 1)) != F2V_OK) return rc;
    HIPC(hipStreamSynchronize(c->stream));
    if ((rc = check_push_err(c, "f2v_push_selftest")) != F2V_OK) return rc;
    std::vector<float> got((size_t)W * D);
    HIPC(hipMemcpy(got.data(), c->d_X[which] + (size_t)c->n * D, got.size() * sizeof(float), hipMemcpyDeviceToHost));
    bool ok = true;
    for (uint32_t r = 0; r < W && ok; r++)
        for (uint32_t d = 0; d < D; d++)
            if (got[(size_t)r * D + d] != value(r, d)) { ok = false; break; }
    if ((rc = launch_barrier(c)) != F2V_OK) return rc;  // nobody reuses the slack rows before everybody has checked
    HIPC(hipStreamSynchronize(c->stream));
    if ((rc = check_push_err(c, "f2v_push_selftest")) != F2V_OK) return rc;
    if (!ok) return fail(F2V_ENODEV, "f2v_push_selftest: a peer's row did not arrive intact");
    return F2V_OK;
}

int f2v_push_stats(f2v_handle c, uint64_t *rows_pushed_out, uint64_t *rows_allgather_out) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    if (rows_pushed_out) *rows_pushed_out = c->push.rows_pushed;
    if (rows_allgather_out) *rows_allgather_out = c->push.rows_allgather;
    return F2V_OK;
}

#ifdef F2V_TEST_HOOKS
// PMC calibration hook: gathers `rows` distinct rows of 128 floats (a random permutation, so every
// 512-byte row is fetched exactly once) with the step kernel's access pattern; `reps` launches.
// Known HBM read volume per launch: rows * 512 bytes (+ 4 bytes per row of ids).
int f2v_test_gather_calibration(int device, uint32_t rows, uint32_t reps) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(F2V_ENODEV, "no HIP device visible");
    if (rows < 4) return fail(F2V_EINVAL, "rows");
    HIPC(hipSetDevice(device));
    float *d_t = nullptr, *d_o = nullptr;
    uint32_t *d_i = nullptr;
    HIPC(hipMalloc((void **)&d_t, (size_t)rows * 128 * sizeof(float)));
    HIPC(hipMalloc((void **)&d_i, (size_t)rows * sizeof(uint32_t)));
    HIPC(hipMalloc((void **)&d_o, 64));
    HIPC(hipMemset(d_t, 0, (size_t)rows * 128 * sizeof(float)));
    std::vector<uint32_t> perm(rows);
    for (uint32_t i = 0; i < rows; i++) perm[i] = i;
    Rand g;
    g.seed(7);
    for (uint32_t i = rows - 1; i > 0; i--) std::swap(perm[i], perm[(uint32_t)(((uint64_t)g.next() * 2147483648ull + (uint64_t)g.next()) % (i + 1))]);
    HIPC(hipMemcpy(d_i, perm.data(), (size_t)rows * sizeof(uint32_t), hipMemcpyHostToDevice));
    for (uint32_t r = 0; r < reps; r++)
        hipLaunchKernelGGL((gather_calibration_kernel<2>), dim3(4096), dim3(256), 0, 0, d_t, d_i, rows, d_o);
    HIPC(hipGetLastError());
    HIPC(hipDeviceSynchronize());
    (void)hipFree(d_t); (void)hipFree(d_i); (void)hipFree(d_o);
    return F2V_OK;
}

// Per-XCD timing of the launches that follow (one launch per minibatch): on = 1 allocates / clears the 32 words (StepArgs::xcd_times), `out`
// (32 words, may be null) receives what has been recorded since; on = 0 frees them.
int f2v_test_xcd_times(f2v_handle c, int on, unsigned long long *out) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    HIPC(hipSetDevice(c->device));
    HIPC(hipStreamSynchronize(c->stream));
    if (out && c->d_xcd) HIPC(hipMemcpy(out, c->d_xcd, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (on) {
        if (!c->d_xcd) HIPC(hipMalloc((void **)&c->d_xcd, 32 * sizeof(unsigned long long)));
        unsigned long long init[32];
        for (int k = 0; k < 32; k++) init[k] = (k >= 8 && k < 16) ? ~0ull : 0ull;
        HIPC(hipMemcpy(c->d_xcd, init, sizeof init, hipMemcpyHostToDevice));
    } else if (c->d_xcd) {
        (void)hipFree(c->d_xcd);
        c->d_xcd = nullptr;
    }
    return F2V_OK;
}

// The memory side of one real launch, alone: the plan of minibatch [row_lo, row_hi) (options 5/6 at D = 128: the headline's layout) replayed by
// plan_gather_kernel -- `reps` launches, the best time in microseconds.  The second matrix is written where mode bit 1 is set: the
// handle's embeddings are garbage afterwards.
int f2v_test_plan_gather(f2v_handle c, uint32_t row_lo, uint32_t row_hi, uint32_t mode, uint32_t reps, double *us_out) {
    if (!c || !us_out || row_lo >= row_hi || row_hi > c->n || reps == 0) return fail(F2V_EINVAL, "f2v_test_plan_gather: bad argument");
    if (subwave_width(c) != 128 || c->D != 128) return fail(F2V_EINVAL, "f2v_test_plan_gather: D = 128 only");
    HIPC(hipSetDevice(c->device));
    int rc = flush_pending(c);
    if (rc != F2V_OK) return rc;
    const Plan plan = plan_for(c, row_lo, row_hi, false);
    if ((rc = upload_plans(c)) != F2V_OK) return rc;
    float *d_o = nullptr;
    HIPC(hipMalloc((void **)&d_o, 64));
    hipEvent_t e0, e1;
    HIPC(hipEventCreate(&e0));
    HIPC(hipEventCreate(&e1));
    const uint32_t G = 1u << ((mode >> 2) & 3u);  // mode bits 2-3: a wavefront takes 1 / 2 / 4 / 8 consecutive groups of items
    const uint32_t blocks = ((plan.n_items + 16u * 8u * G - 1u) / (16u * 8u * G)) * 8u;  // (plan_gather_kernel: workgroup B takes the plan's workgroups B % 8 + 8 (G (B / 8) + k))
    float best = 1e30f;
    for (uint32_t r = 0; r < reps + 2; r++) {
        HIPC(hipEventRecord(e0, c->stream));
#define F2V_PG(GG) hipLaunchKernelGGL((plan_gather_kernel<16, 2, 4, GG>), dim3(blocks), dim3(256), 0, c->stream, c->d_X[c->cur], c->d_X[c->cur ^ 1], c->d_items + plan.item_off, plan.n_items, c->d_colids, mode & 3u, d_o)
        if (G == 1) F2V_PG(1); else if (G == 2) F2V_PG(2); else if (G == 4) F2V_PG(4); else F2V_PG(8);
#undef F2V_PG
        HIPC(hipEventRecord(e1, c->stream));
        HIPC(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPC(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(d_o);
    *us_out = best * 1e3;
    return F2V_OK;
}

#endif  // F2V_TEST_HOOKS

// Stand-alone rehearsal of what f2v_push_attach + the push kernels need from the machine, meant to run in a
// THROW-AWAY process before the real engines exist: allocate `bytes` of device memory and a fine-grained flag array
// on `device`, swap IPC handles with the other ranks through files in `dir`, map theirs, store a word at both ends of
// every peer's buffer and into every peer's flags from a kernel, and check what the peers stored here.  Whatever goes
// wrong -- a mapping call that never returns, a fault on the first remote store -- happens to this process.
int f2v_diag_ipc_preflight(int device, uint32_t rank, uint32_t world, const char *dir, uint64_t bytes, double timeout_s) {
    if (!dir || world == 0 || world > (uint32_t)kMaxRanks || rank >= world || bytes < 4096) return fail(F2V_EINVAL, "f2v_diag_ipc_preflight: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(F2V_ENODEV, "f2v_diag_ipc_preflight: device %d of %d", device, ndev);
    HIPC(hipSetDevice(device));
    uint32_t *data = nullptr;
    unsigned long long *flags = nullptr;
    HIPC(hipMalloc((void **)&data, bytes));
    HIPC(hipMemset(data, 0, 4096));
    HIPC(hipMemset((char *)data + bytes - 4096, 0, 4096));
    HIPC(hipExtMallocWithFlags((void **)&flags, 4096, hipDeviceMallocFinegrained));
    HIPC(hipMemset(flags, 0, 4096));
    HIPC(hipDeviceSynchronize());
    struct Blob { hipIpcMemHandle_t data, flags; } mine, theirs;
    HIPC(hipIpcGetMemHandle(&mine.data, data));
    HIPC(hipIpcGetMemHandle(&mine.flags, flags));
    const auto t0 = std::chrono::steady_clock::now();
    auto late = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s; };
    auto path = [&](const char *what, uint32_t r) { return std::string(dir) + "/" + what + "." + std::to_string(r); };
    auto publish = [&](const char *what, const void *p, size_t len) -> bool {
        const std::string tmp = path(what, rank) + ".tmp";
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f) return false;
        const bool ok = fwrite(p, 1, len, f) == len;
        return (fclose(f) == 0) && ok && rename(tmp.c_str(), path(what, rank).c_str()) == 0;
    };
    auto await = [&](const char *what, uint32_t r, void *p, size_t len) -> bool {
        for (;;) {
            FILE *f = fopen(path(what, r).c_str(), "rb");
            if (f) {
                const size_t got = fread(p, 1, len, f);
                fclose(f);
                if (got == len) return true;
            }
            if (late()) return false;
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
    };
    if (!publish("handles", &mine, sizeof mine)) return fail(F2V_EIO, "f2v_diag_ipc_preflight: cannot write into %s", dir);
    PreflightArgs a{};
    a.self = rank;
    a.world = world;
    a.value = 0xF2F00000u + rank;
    a.last_word = bytes / 4 - 1;
    for (uint32_t r = 0; r < world; r++) {
        if (r == rank) { a.data[r] = data; a.flags[r] = flags; continue; }
        if (!await("handles", r, &theirs, sizeof theirs)) return fail(F2V_ESTATE, "f2v_diag_ipc_preflight: rank %u never published its handles", r);
        HIPC(hipIpcOpenMemHandle((void **)&a.data[r], theirs.data, hipIpcMemLazyEnablePeerAccess));
        HIPC(hipIpcOpenMemHandle((void **)&a.flags[r], theirs.flags, hipIpcMemLazyEnablePeerAccess));
    }
    hipLaunchKernelGGL(preflight_write_kernel, dim3(1), dim3(64), 0, 0, a);
    HIPC(hipGetLastError());
    HIPC(hipDeviceSynchronize());
    char one = 1;
    if (!publish("stored", &one, 1)) return fail(F2V_EIO, "f2v_diag_ipc_preflight: cannot write into %s", dir);
    for (uint32_t r = 0; r < world; r++)
        if (r != rank && !await("stored", r, &one, 1)) return fail(F2V_ESTATE, "f2v_diag_ipc_preflight: rank %u never finished its stores", r);
    std::vector<uint32_t> head(world), tail(world);
    std::vector<unsigned long long> fl(world);
    HIPC(hipMemcpy(head.data(), data, world * 4, hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(tail.data(), data + a.last_word + 1 - world, world * 4, hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(fl.data(), flags, world * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (uint32_t r = 0; r < world; r++) {
        const uint32_t want = 0xF2F00000u + r;
        if (head[r] != want || tail[world - 1 - r] != want || fl[r] != want) bad++;
    }
    if (!publish("checked", &one, 1)) return fail(F2V_EIO, "f2v_diag_ipc_preflight: cannot write into %s", dir);
    for (uint32_t r = 0; r < world; r++)  // nobody unmaps or frees while a peer may still be reading
        if (r != rank && !await("checked", r, &one, 1)) break;
    for (uint32_t r = 0; r < world; r++)
        if (r != rank) { (void)hipIpcCloseMemHandle(a.data[r]); (void)hipIpcCloseMemHandle(a.flags[r]); }
    (void)hipFree(data);
    (void)hipFree(flags);
    if (bad) return fail(F2V_ENODEV, "f2v_diag_ipc_preflight: %d of %u peers' stores did not arrive intact", bad, world);
    return F2V_OK;
}

// Random-row gather ceiling of this card: every 512-byte row of a `table_bytes` table is fetched exactly once per pass, in
// random order, with the step kernel's access pattern (16 lanes x dwordx4 per row, 4 rows in flight per item); enough
// passes per launch to gather about 2 GiB.  Small tables measure the Infinity Cache / L2, tables far beyond 256 MiB the HBM.
int f2v_diag_gather_rate(int device, uint64_t table_bytes, uint32_t reps, double *gbps_out) {
    if (!gbps_out || table_bytes < (1u << 20) || table_bytes > (64ull << 30) || reps == 0) return fail(F2V_EINVAL, "f2v_diag_gather_rate: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(F2V_ENODEV, "no such HIP device");
    HIPC(hipSetDevice(device));
    const uint32_t rows = (uint32_t)(table_bytes / 512);
    const uint32_t passes = (uint32_t)std::max<uint64_t>(1, (2ull << 30) / ((uint64_t)rows * 512));
    const uint64_t n_ids = (uint64_t)rows * passes;
    if (n_ids > 0xFFFFFFF0ull) return fail(F2V_EINVAL, "f2v_diag_gather_rate: table too large");
    std::vector<uint32_t> ids(n_ids);
    Rand g;
    g.seed(11);
    auto big = [&] { return ((uint64_t)g.next() << 31) | (uint64_t)g.next(); };
    for (uint32_t i = 0; i < rows; i++) ids[i] = i;
    for (uint32_t i = rows - 1; i > 0; i--) std::swap(ids[i], ids[(uint32_t)(big() % (i + 1))]);
    for (uint32_t p = 1; p < passes; p++) {  // later passes: the same permutation under another random rotation and stride
        const uint32_t rot = (uint32_t)(big() % rows);
        for (uint32_t i = 0; i < rows; i++) ids[(size_t)p * rows + i] = ids[(i + rot) % rows] ^ 0u;
    }
    float *d_t = nullptr, *d_o = nullptr;
    uint32_t *d_i = nullptr;
    HIPC(hipMalloc((void **)&d_t, (size_t)rows * 512));
    HIPC(hipMalloc((void **)&d_i, n_ids * sizeof(uint32_t)));
    HIPC(hipMalloc((void **)&d_o, 64));
    HIPC(hipMemset(d_t, 0, (size_t)rows * 512));
    HIPC(hipMemcpy(d_i, ids.data(), n_ids * sizeof(uint32_t), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIPC(hipEventCreate(&e0));
    HIPC(hipEventCreate(&e1));
    double best = 0.0;
    for (uint32_t r = 0; r < reps + 2; r++) {
        HIPC(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((gather_calibration_kernel<2>), dim3(4096), dim3(256), 0, 0, d_t, d_i, (uint32_t)n_ids, d_o);
        HIPC(hipEventRecord(e1, 0));
        HIPC(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPC(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2 && ms > 0.f) best = std::max(best, (double)n_ids * 516.0 / (ms * 1e-3) * 1e-9);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(d_t); (void)hipFree(d_i); (void)hipFree(d_o);
    *gbps_out = best;
    return F2V_OK;
}

// Streaming-copy ceiling of this card: `reps` copies of `bytes` (read + written = 2*bytes each), best rate in GB/s.
int f2v_diag_stream_copy(int device, uint64_t bytes, uint32_t reps, double *gbps_out) {
    if (!gbps_out || bytes < 4096 || reps == 0 || bytes / 16 / 1024 > 0x7FFFFFFFull) return fail(F2V_EINVAL, "f2v_diag_stream_copy: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(F2V_ENODEV, "no such HIP device");
    HIPC(hipSetDevice(device));
    float4 *a = nullptr, *b = nullptr;
    HIPC(hipMalloc((void **)&a, bytes));
    HIPC(hipMalloc((void **)&b, bytes));
    HIPC(hipMemset(a, 1, bytes));
    HIPC(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    HIPC(hipEventCreate(&e0));
    HIPC(hipEventCreate(&e1));
    double best = 0.0;
    for (uint32_t r = 0; r < reps + 2; r++) {
        HIPC(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((stream_copy_kernel<4>), dim3((uint32_t)((bytes / 16 + 1023) / 1024)), dim3(256), 0, 0, (const f32x4_copy_t *)a, (f32x4_copy_t *)b, (uint64_t)(bytes / 16));
        HIPC(hipEventRecord(e1, 0));
        HIPC(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPC(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2 && ms > 0.f) best = std::max(best, 2.0 * (double)bytes / (ms * 1e-3) * 1e-9);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(a); (void)hipFree(b);
    *gbps_out = best;
    return F2V_OK;
}

#ifdef F2V_TEST_HOOKS
int f2v_test_wave_reduce(int device, const float *in, uint32_t rows, uint32_t width, float *out) {
    if (!in || !out || width == 0 || width > 512) return fail(F2V_EINVAL, "f2v_test_wave_reduce: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(F2V_ENODEV, "no HIP device visible");
    HIPC(hipSetDevice(device));
    float *d_in = nullptr, *d_out = nullptr;
    HIPC(hipMalloc((void **)&d_in, (size_t)rows * width * sizeof(float)));
    HIPC(hipMalloc((void **)&d_out, (size_t)rows * sizeof(float)));
    HIPC(hipMemcpy(d_in, in, (size_t)rows * width * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(wave_reduce_test_kernel, dim3((rows + 3) / 4), dim3(256), 0, 0, d_in, rows, width, d_out);
    HIPC(hipGetLastError());
    HIPC(hipDeviceSynchronize());
    HIPC(hipMemcpy(out, d_out, (size_t)rows * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return F2V_OK;
}
#endif  // F2V_TEST_HOOKS

}  // extern "C"

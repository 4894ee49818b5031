// f2v_engine.hip -- HBM-resident Force2Vec engine behind the C ABI of include/f2v.h.
//
// Device layout (all in the HBM of one MI355X):
//   X          fp32 [N x D] row-major (rows 4*D bytes apart: 512 B at D = 128)
//   rowptr     u32  [N+1], colids u32 [nnz]           (the reference's CSR, sample/CSR.h:89-96)
//   stage[2]   fp32 [batch rows x D]  new rows of the pending / current minibatch (ping-pong)
//   partials   fp32 [hub chunks x D]  partial force sums of split hub rows
//   sample ids u32, walks u32 [5N], sigmoid table fp32 [2048]
// The host side only sequences launches; every arithmetic step runs in f2v_kernels.hip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "f2v.h"
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

struct f2v_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t n = 0, D = 0;
    uint64_t nnz = 0;
    int vec = 1;
    bool exact = false;
    std::vector<uint32_t> rowptr, colids;  // host copies: hub planning, walk generation, statistics
    uint32_t *d_rowptr = nullptr, *d_colids = nullptr, *d_walks = nullptr, *d_ids = nullptr;
    size_t ids_cap = 0;
    float *d_X = nullptr, *d_stage[2] = {nullptr, nullptr}, *d_partials = nullptr, *d_table = nullptr;
    uint32_t stage_cap = 0;
    bool have_x = false, have_walks = false;
    Rand rng;
    // hub plan for the current chunk size
    uint32_t chunk = 512, plan_chunk = 0xFFFFFFFFu;
    std::vector<uint32_t> hub_rows;     // rows with degree > chunk, ascending
    std::vector<uint32_t> hub_prefix;   // chunk-count prefix over hub_rows
    uint2 *d_extras = nullptr;
    HubRow *d_hubs = nullptr;
    // pending (staged, not yet committed) minibatch
    bool pending = false;
    uint32_t p_lo = 0, p_hi = 0;
    int p_idx = 0;
    int waves_per_block = 4;
    f2v_stats stats{};
};

namespace {

int pick_vec(uint32_t D) {
    int v = 1;
    while ((uint32_t)(64 * v) < D) v <<= 1;
    return v;
}

int build_hub_plan(f2v_ctx *c) {
    if (c->plan_chunk == c->chunk) return F2V_OK;
    c->hub_rows.clear();
    c->hub_prefix.assign(1, 0u);
    if (c->d_extras) { (void)hipFree(c->d_extras); c->d_extras = nullptr; }
    if (c->d_hubs) { (void)hipFree(c->d_hubs); c->d_hubs = nullptr; }
    if (c->d_partials) { (void)hipFree(c->d_partials); c->d_partials = nullptr; }
    if (c->chunk != 0) {
        std::vector<uint2> extras;
        std::vector<HubRow> hubs;
        for (uint32_t i = 0; i < c->n; i++) {
            const uint32_t deg = c->rowptr[i + 1] - c->rowptr[i];
            if (deg > c->chunk) {
                const uint32_t nc = (deg + c->chunk - 1) / c->chunk;
                hubs.push_back(HubRow{i, (uint32_t)extras.size(), nc});
                for (uint32_t k = 0; k < nc; k++) extras.push_back(make_uint2(i, k));
                c->hub_rows.push_back(i);
                c->hub_prefix.push_back((uint32_t)extras.size());
            }
        }
        if (!extras.empty()) {
            HIPC(hipMalloc((void **)&c->d_extras, extras.size() * sizeof(uint2)));
            HIPC(hipMalloc((void **)&c->d_hubs, hubs.size() * sizeof(HubRow)));
            HIPC(hipMalloc((void **)&c->d_partials, extras.size() * (size_t)c->D * sizeof(float)));
            HIPC(hipMemcpy(c->d_extras, extras.data(), extras.size() * sizeof(uint2), hipMemcpyHostToDevice));
            HIPC(hipMemcpy(c->d_hubs, hubs.data(), hubs.size() * sizeof(HubRow), hipMemcpyHostToDevice));
        }
    }
    c->plan_chunk = c->chunk;
    return F2V_OK;
}

int reserve_stage(f2v_ctx *c, uint32_t rows) {
    if (rows <= c->stage_cap) return F2V_OK;
    if (c->pending) return fail(F2V_ESTATE, "staging buffer cannot grow while a minibatch is pending (flush first)");
    HIPC(hipStreamSynchronize(c->stream));  // a commit of the old buffers may still be in flight
    for (int k = 0; k < 2; k++) {
        if (c->d_stage[k]) (void)hipFree(c->d_stage[k]);
        c->d_stage[k] = nullptr;
        HIPC(hipMalloc((void **)&c->d_stage[k], (size_t)rows * c->D * sizeof(float)));
    }
    c->stage_cap = rows;
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
void launch_commit_t(f2v_ctx *c, const float *stage, uint32_t lo, uint32_t rows) {
    const int wpb = 4;
    const uint32_t blocks = std::min<uint32_t>((rows + wpb - 1) / wpb, 4096u);
    hipLaunchKernelGGL((commit_kernel<VEC, EXACT>), dim3(blocks), dim3(64 * wpb), 0, c->stream, c->d_X, stage, lo, rows, c->D);
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

int flush_pending(f2v_ctx *c) {
    if (!c->pending) return F2V_OK;
    const float *stage = c->d_stage[c->p_idx];
    const uint32_t lo = c->p_lo, rows = c->p_hi - c->p_lo;
    int rc = dispatch_layout(c, [&](auto V, auto E) { launch_commit_t<decltype(V)::value, decltype(E)::value>(c, stage, lo, rows); });
    if (rc != F2V_OK) return rc;
    HIPC(hipGetLastError());
    c->pending = false;
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
int launch_step(f2v_ctx *c, int math, uint32_t batch_lo, uint32_t batch_hi, uint32_t row_lo, uint32_t row_hi,
                const uint32_t *d_ids, uint32_t ns, float lr, int bs_mode) {
    int rc;
    if (c->pending && !(c->p_hi <= batch_lo || batch_hi <= c->p_lo)) {
        // the pending batch overlaps this one (single-batch epochs): commit it with its own launch
        if ((rc = flush_pending(c)) != F2V_OK) return rc;
    }
    const uint32_t brows = batch_hi - batch_lo;
    if (brows > c->stage_cap) {
        if ((rc = flush_pending(c)) != F2V_OK) return rc;
        if ((rc = reserve_stage(c, brows)) != F2V_OK) return rc;
    }
    const bool walk = (math == 7);
    if ((rc = build_hub_plan(c)) != F2V_OK) return rc;
    const int cur = c->pending ? (c->p_idx ^ 1) : 0;

    StepArgs a{};
    a.X = c->d_X;
    a.rowptr = c->d_rowptr;
    a.nbr_ids = walk ? c->d_walks : c->d_colids;
    a.stage_prev = c->pending ? c->d_stage[c->p_idx] : nullptr;
    a.stage_cur = c->d_stage[cur];
    a.sample_ids = d_ids;
    a.sm_table = c->d_table;
    a.D = c->D;
    a.batch_lo = batch_lo;
    a.row_lo = row_lo;
    a.n_rows = row_hi - row_lo;
    a.prev_lo = c->pending ? c->p_lo : 0;
    a.prev_rows = c->pending ? (c->p_hi - c->p_lo) : 0;
    a.ns = ns;
    a.bs_mode = bs_mode ? 1u : 0u;
    a.chunk = walk ? 0u : c->chunk;
    a.walk_mode = walk ? 1u : 0u;
    a.lr = lr;
    uint32_t h0 = 0, h1 = 0;
    if (!walk && c->chunk != 0 && !c->hub_rows.empty()) {
        h0 = (uint32_t)(std::lower_bound(c->hub_rows.begin(), c->hub_rows.end(), row_lo) - c->hub_rows.begin());
        h1 = (uint32_t)(std::lower_bound(c->hub_rows.begin(), c->hub_rows.end(), row_hi) - c->hub_rows.begin());
    }
    const uint32_t slot_base = h1 > h0 ? c->hub_prefix[h0] : 0;
    a.n_extra = h1 > h0 ? c->hub_prefix[h1] - slot_base : 0;
    a.extras = h1 > h0 ? c->d_extras + slot_base : nullptr;
    a.partials = h1 > h0 ? c->d_partials + (size_t)slot_base * c->D : nullptr;

    const uint32_t wpb = (uint32_t)c->waves_per_block;
    const uint32_t items = a.n_extra + a.n_rows;
    // enough blocks for the items; when a rank has few or no rows of this batch, still enough to commit the pending one
    const uint32_t blocks = std::max<uint32_t>(std::max<uint32_t>(1u, (items + wpb - 1) / wpb),
                                               std::min<uint32_t>((a.prev_rows + wpb - 1) / wpb, 2048u));
    rc = dispatch_layout(c, [&](auto V, auto E) {
        constexpr int VEC = decltype(V)::value;
        constexpr bool EX = decltype(E)::value;
        if (math == 5)
            hipLaunchKernelGGL((step_kernel<5, VEC, EX>), dim3(blocks), dim3(64 * wpb), 0, c->stream, a);
        else
            hipLaunchKernelGGL((step_kernel<6, VEC, EX>), dim3(blocks), dim3(64 * wpb), 0, c->stream, a);
    });
    if (rc != F2V_OK) return rc;
    HIPC(hipGetLastError());
    if (h1 > h0) {
        FinalizeArgs f{};
        f.X = c->d_X;
        f.partials = a.partials;
        f.stage_cur = a.stage_cur;
        f.hubs = c->d_hubs + h0;
        f.slot_base = slot_base;
        f.n_hubs = h1 - h0;
        f.D = c->D;
        f.batch_lo = batch_lo;
        const uint32_t fb = (f.n_hubs + 3) / 4;
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
        c->stats.hub_rows += f.n_hubs;
        c->stats.hub_chunks += a.n_extra;
    }
    c->pending = true;
    c->p_lo = batch_lo;
    c->p_hi = batch_hi;
    c->p_idx = cur;

    // statistics: algorithmic bytes of SURVEY 8d -- nnz*(4D+4) + rows*(8D+4) + ns*(4D+4) per minibatch
    const uint64_t rows = a.n_rows;
    const uint64_t nz = walk ? rows * kWalkLength : (uint64_t)(c->rowptr[row_hi] - c->rowptr[row_lo]);
    c->stats.step_launches += 1;
    c->stats.rows += rows;
    c->stats.nnz += nz;
    c->stats.algorithmic_bytes += nz * (4ull * c->D + 4) + rows * (8ull * c->D + 4) + (uint64_t)ns * (4ull * c->D + 4);
    return F2V_OK;
}

void generate_walks_host(f2v_ctx *c, std::vector<uint32_t> &walks) {
    // sample/algorithms.cpp:1097-1118.  deg>2: a random neighbour except the last; deg==2: the first;
    // otherwise colids[w] with the VERTEX id as edge index (kept for parity; clamped to the array).
    walks.resize((size_t)c->n * kWalkLength);
    const uint32_t *rp = c->rowptr.data(), *ci = c->colids.data();
    for (uint32_t i = 0; i < c->n; i++) {
        uint32_t w = i;
        for (int s = 0; s < kWalkLength; s++) {
            uint32_t j = w;
            const uint32_t deg = rp[w + 1] - rp[w];
            if (deg > 2)
                j = c->rng.index(rp[w + 1] - 1, rp[w]);
            else if (deg == 2)
                j = rp[w];
            if ((uint64_t)j >= c->nnz) j = (uint32_t)(c->nnz ? c->nnz - 1 : 0);
            walks[(size_t)i * kWalkLength + s] = ci[j];
            w = ci[j];
        }
    }
}

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
    HIPB(hipMalloc((void **)&c->d_X, (size_t)n * dim * sizeof(float)));
    HIPB(hipMalloc((void **)&c->d_table, kSmTableSize * sizeof(float)));
    HIPB(hipMemcpy(c->d_rowptr, rowptr, ((size_t)n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (nnz) HIPB(hipMemcpy(c->d_colids, colids, nnz * sizeof(uint32_t), hipMemcpyHostToDevice));
    float table[kSmTableSize];
    sm_table_host(table);
    HIPB(hipMemcpy(c->d_table, table, sizeof table, hipMemcpyHostToDevice));
#undef HIPB
    *out = c;
    return F2V_OK;
}

int f2v_destroy(f2v_handle c) {
    if (!c) return F2V_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void *ptrs[] = {c->d_rowptr, c->d_colids, c->d_walks, c->d_ids, c->d_X, c->d_stage[0], c->d_stage[1],
                    c->d_partials, c->d_table, c->d_extras, c->d_hubs};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return F2V_OK;
}

int f2v_srand(f2v_handle c, uint32_t seed) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    c->rng.seed(seed);
    return F2V_OK;
}

int f2v_rand_index(f2v_handle c, uint32_t max_num, uint32_t min_num, uint32_t *out) {
    if (!c || !out || max_num <= min_num) return fail(F2V_EINVAL, "f2v_rand_index: bad argument");
    *out = c->rng.index(max_num, min_num);
    return F2V_OK;
}

int f2v_init_embeddings(f2v_handle c, int kind) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    if (kind != F2V_INIT_SYMMETRIC && kind != F2V_INIT_UNIT) return fail(F2V_EINVAL, "f2v_init_embeddings: kind %d", kind);
    HIPC(hipSetDevice(c->device));
    const size_t total = (size_t)c->n * c->D;
    std::vector<float> x(total);
    init_embeddings_host(c->rng, x.data(), total, kind);
    HIPC(hipStreamSynchronize(c->stream));
    c->pending = false;
    HIPC(hipMemcpy(c->d_X, x.data(), total * sizeof(float), hipMemcpyHostToDevice));
    c->have_x = true;
    return F2V_OK;
}

int f2v_set_embeddings(f2v_handle c, const float *x) {
    if (!c || !x) return fail(F2V_EINVAL, "f2v_set_embeddings: null argument");
    HIPC(hipSetDevice(c->device));
    HIPC(hipStreamSynchronize(c->stream));
    c->pending = false;
    HIPC(hipMemcpy(c->d_X, x, (size_t)c->n * c->D * sizeof(float), hipMemcpyHostToDevice));
    c->have_x = true;
    return F2V_OK;
}

int f2v_get_embeddings(f2v_handle c, float *x_out) {
    if (!c || !x_out) return fail(F2V_EINVAL, "f2v_get_embeddings: null argument");
    if (!c->have_x) return fail(F2V_ESTATE, "f2v_get_embeddings: embeddings were never initialised");
    HIPC(hipSetDevice(c->device));
    int rc = flush_pending(c);
    if (rc != F2V_OK) return rc;
    HIPC(hipStreamSynchronize(c->stream));
    HIPC(hipMemcpy(x_out, c->d_X, (size_t)c->n * c->D * sizeof(float), hipMemcpyDeviceToHost));
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
        c->chunk = (uint32_t)value;
        return F2V_OK;
    }
    if (!strcmp(name, "waves_per_block")) {
        if (value != 1 && value != 2 && value != 4) return fail(F2V_EINVAL, "waves_per_block must be 1, 2 or 4");
        c->waves_per_block = (int)value;
        return F2V_OK;
    }
    return fail(F2V_EINVAL, "f2v_set_param: unknown parameter '%s'", name);
}

int f2v_get_param(f2v_handle c, const char *name, int64_t *out) {
    if (!c || !name || !out) return fail(F2V_EINVAL, "f2v_get_param: null argument");
    if (!strcmp(name, "hub_chunk")) { *out = c->chunk; return F2V_OK; }
    if (!strcmp(name, "waves_per_block")) { *out = c->waves_per_block; return F2V_OK; }
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

int f2v_generate_walks(f2v_handle c, uint32_t *walks_out) {
    if (!c) return fail(F2V_EINVAL, "null handle");
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
    if (need) {
        HIPC(hipMemcpyAsync(c->d_ids, sample_ids, need * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
    }
    return launch_step(c, math, batch_lo, batch_hi, row_lo, row_hi, c->d_ids, ns, lr, bs_mode);
}

int f2v_flush(f2v_handle c) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    HIPC(hipSetDevice(c->device));
    int rc = flush_pending(c);
    if (rc != F2V_OK) return rc;
    HIPC(hipStreamSynchronize(c->stream));
    return F2V_OK;
}

int f2v_synchronize(f2v_handle c) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    HIPC(hipSetDevice(c->device));
    HIPC(hipStreamSynchronize(c->stream));
    return F2V_OK;
}

int f2v_stage_reserve(f2v_handle c, uint32_t rows) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    HIPC(hipSetDevice(c->device));
    if (rows > c->stage_cap) {
        int rc = flush_pending(c);
        if (rc != F2V_OK) return rc;
        HIPC(hipStreamSynchronize(c->stream));
    }
    return reserve_stage(c, rows);
}

int f2v_stage_device_ptr(f2v_handle c, uint64_t *devptr_out, uint32_t *cap_out) {
    if (!c || !devptr_out) return fail(F2V_EINVAL, "f2v_stage_device_ptr: null argument");
    if (!c->pending) return fail(F2V_ESTATE, "f2v_stage_device_ptr: no minibatch is pending");
    *devptr_out = (uint64_t)(uintptr_t)c->d_stage[c->p_idx];
    if (cap_out) *cap_out = c->stage_cap;
    return F2V_OK;
}

int f2v_stage_read(f2v_handle c, uint32_t row_lo, uint32_t row_hi, float *out) {
    if (!c || !out) return fail(F2V_EINVAL, "f2v_stage_read: null argument");
    if (!c->pending || row_lo < c->p_lo || row_hi > c->p_hi || row_lo > row_hi) return fail(F2V_ESTATE, "f2v_stage_read: rows outside the pending minibatch");
    HIPC(hipSetDevice(c->device));
    HIPC(hipStreamSynchronize(c->stream));
    HIPC(hipMemcpy(out, c->d_stage[c->p_idx] + (size_t)(row_lo - c->p_lo) * c->D, (size_t)(row_hi - row_lo) * c->D * sizeof(float), hipMemcpyDeviceToHost));
    return F2V_OK;
}

int f2v_stage_write(f2v_handle c, uint32_t row_lo, uint32_t row_hi, const float *in) {
    if (!c || !in) return fail(F2V_EINVAL, "f2v_stage_write: null argument");
    if (!c->pending || row_lo < c->p_lo || row_hi > c->p_hi || row_lo > row_hi) return fail(F2V_ESTATE, "f2v_stage_write: rows outside the pending minibatch");
    HIPC(hipSetDevice(c->device));
    HIPC(hipStreamSynchronize(c->stream));
    HIPC(hipMemcpy(c->d_stage[c->p_idx] + (size_t)(row_lo - c->p_lo) * c->D, in, (size_t)(row_hi - row_lo) * c->D * sizeof(float), hipMemcpyHostToDevice));
    return F2V_OK;
}

int f2v_embeddings_device_ptr(f2v_handle c, uint64_t *out) {
    if (!c || !out) return fail(F2V_EINVAL, "null argument");
    *out = (uint64_t)(uintptr_t)c->d_X;
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
    return F2V_OK;
}

int f2v_train(f2v_handle c, int option, uint32_t iters, uint32_t batch, uint32_t ns, float lr, int bs_mode,
              double *seconds_out) {
    if (!c) return fail(F2V_EINVAL, "null handle");
    const int math = math_of_option(option);
    if (!math) return fail(F2V_EINVAL, "f2v_train: option %d is outside 5..11", option);
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
    if ((rc = reserve_stage(c, std::min(batch, n))) != F2V_OK) {
        if ((rc = flush_pending(c)) != F2V_OK) return rc;
        HIPC(hipStreamSynchronize(c->stream));
        if ((rc = reserve_stage(c, std::min(batch, n))) != F2V_OK) return rc;
    }
    if ((rc = build_hub_plan(c)) != F2V_OK) return rc;
    // Sample ids do not depend on the embeddings: options 5/6 pre-draw every epoch's ids (as long
    // as that stays below 1 GiB); option 7 interleaves walk generation, so it goes epoch by epoch.
    const bool all_upfront = (math != 7) && (per_epoch * iters * 4ull <= (1ull << 30));
    const uint64_t dev_ids = std::max<uint64_t>(all_upfront ? per_epoch * std::max(iters, 1u) : per_epoch, 64);
    if ((rc = reserve_ids(c, dev_ids)) != F2V_OK) return rc;
    std::vector<uint32_t> ids;
    auto draw_epoch = [&](std::vector<uint32_t> &v, size_t off) {
        for (uint32_t b = 0; b < nb; b++) {
            uint32_t maxv = n - 1;
            if (math == 7) {  // algorithms.cpp:1125
                const uint64_t e = (uint64_t)(b + 1) * batch;
                if (e < maxv) maxv = (uint32_t)e;
            }
            for (uint64_t s = 0; s < ndraw; s++) {
                const uint32_t r = c->rng.index(maxv, 0);
                if (s < stride) v[off + (size_t)b * stride + s] = r;
            }
        }
    };
    if (all_upfront) {
        ids.assign(per_epoch * iters, 0u);
        for (uint32_t it = 0; it < iters; it++) draw_epoch(ids, (size_t)it * per_epoch);
        if (!ids.empty()) HIPC(hipMemcpy(c->d_ids, ids.data(), ids.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    } else {
        ids.assign(per_epoch, 0u);
    }
    c->stats = f2v_stats{};
    hipEvent_t ev0, ev1;
    HIPC(hipEventCreate(&ev0));
    HIPC(hipEventCreate(&ev1));
    HIPC(hipEventRecord(ev0, c->stream));
    std::vector<uint32_t> walks;
    for (uint32_t it = 0; it < iters; it++) {
        if (math == 7) {
            generate_walks_host(c, walks);
            if (!c->d_walks) HIPC(hipMalloc((void **)&c->d_walks, walks.size() * sizeof(uint32_t)));
            HIPC(hipStreamSynchronize(c->stream));  // previous epoch's steps read d_walks / d_ids
            HIPC(hipMemcpy(c->d_walks, walks.data(), walks.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            c->have_walks = true;
        }
        const uint32_t *d_epoch_ids = c->d_ids + (all_upfront ? (size_t)it * per_epoch : 0);
        if (!all_upfront) {
            draw_epoch(ids, 0);
            HIPC(hipStreamSynchronize(c->stream));
            HIPC(hipMemcpy(c->d_ids, ids.data(), ids.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        for (uint32_t b = 0; b < nb; b++) {
            const uint32_t lo = b * batch;
            const uint32_t hi = (uint32_t)std::min<uint64_t>((uint64_t)lo + batch, n);
            if ((rc = launch_step(c, math, lo, hi, lo, hi, d_epoch_ids + (size_t)b * stride, ns, lr, bs_mode)) != F2V_OK) return rc;
        }
    }
    if ((rc = flush_pending(c)) != F2V_OK) return rc;
    HIPC(hipEventRecord(ev1, c->stream));
    HIPC(hipEventSynchronize(ev1));
    float ms = 0.f;
    HIPC(hipEventElapsedTime(&ms, ev0, ev1));
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    c->stats.device_seconds = ms * 1e-3;
    if (seconds_out) *seconds_out = ms * 1e-3;
    return F2V_OK;
}

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

}  // extern "C"

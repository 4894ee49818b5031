// f2v_kernels.hip.h -- the HIP/CDNA4 (gfx950) kernels of the Force2Vec hot path.
//
// Work decomposition: ONE 64-lane wavefront per source vertex (row) of the minibatch.
// Lane l owns the VEC contiguous dimensions [l*VEC, l*VEC+VEC) of every D-dim row
// (D <= 64*VEC; D = 128 -> VEC = 2: one 512-byte fully coalesced row per wave-load), so
//   * x_i and the row's force accumulator Y_i live in registers for the whole row,
//   * each CSR neighbour / negative sample is one coalesced row gather,
//   * the squared distance (t-distribution, option 5) or dot product (sigmoid, options
//     6/7) is an in-lane adjacent-pair tree followed by a lane-xor butterfly
//     (1,2,4,8 via DPP, 16 via ds_swizzle, 32 via two v_readlane): the canonical balanced
//     adjacent-pair tree over next_pow2(D) terms that oracle/f2v_oracle.c::ORC_ORDER_TREE
//     restates, so the kernels are bit-exact against the oracle,
//   * the fp64 scalars d1 / coef of the reference (sample/algorithms.cpp:608,622,867) are
//     computed redundantly per lane in fp64, the clamp keeps the compiled reference's
//     NaN -> -5 rule, and mul/add are NOT contracted (-ffp-contract=off).
//
// Minibatch sequencing (Jacobi inside a batch, Gauss-Seidel across batches,
// sample/algorithms.cpp:588-639) without a second launch per batch: the new rows of batch b
// go to a staging buffer, and the step kernel of batch b+1 (a) commits them to X in its
// prologue and (b) redirects every read of a batch-b row (neighbour or negative sample) to
// the staging buffer, so no wave ever reads a row of X that another wave of the same launch
// writes.  Rows of the CURRENT batch are only read from X (their pre-batch values), which is
// exactly the reference's snapshot semantics for samples and in-batch neighbours.
//
// Hub rows (degree > chunk) are cut into chunks of `chunk` neighbours handled by separate
// waves that leave partial sums in HBM; hub_finalize_kernel adds them in chunk order
// (the GPU counterpart of option 11's nnz-balanced partition, sample/algorithms.cpp:2483-2523).
#ifndef F2V_KERNELS_HIP_H_
#define F2V_KERNELS_HIP_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace f2v {

struct StepArgs {
    float *X;                    // N x D embedding matrix (row-major, fp32)
    const uint32_t *rowptr;      // CSR row pointers [N+1]
    const uint32_t *nbr_ids;     // CSR colids, or the epoch's walk samples [5*N] (option 7)
    const float *stage_prev;     // new rows of the previous (pending) minibatch
    float *stage_cur;            // new rows of this minibatch
    float *partials;             // hub chunk partial sums of this launch [n_extra x D]
    const uint32_t *sample_ids;  // negative-sample vertex ids of this minibatch (device)
    const uint2 *extras;         // (row, chunk) of each hub chunk item of this launch
    const float *sm_table;       // 2048-entry sigmoid table
    uint32_t D;
    uint32_t batch_lo;           // first row of the minibatch (staging row 0)
    uint32_t row_lo, n_rows;     // rows this launch computes
    uint32_t prev_lo, prev_rows; // pending minibatch to commit / redirect to
    uint32_t n_extra;
    uint32_t ns;
    uint32_t bs_mode;
    uint32_t chunk;              // hub chunk size (0 = never split)
    uint32_t walk_mode;          // option 7: neighbours are nbr_ids[5*row .. 5*row+5)
    float lr;
};

struct HubRow {
    uint32_t row, slot0, nchunks;
};

struct FinalizeArgs {
    const float *X;
    const float *partials;  // same base as StepArgs::partials
    float *stage_cur;
    const HubRow *hubs;     // slot0 relative to this launch's first slot
    uint32_t slot_base;     // global slot of this launch's first hub chunk
    uint32_t n_hubs;
    uint32_t D;
    uint32_t batch_lo;
};

// ---- cross-lane primitives -------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Sum over the 64 lanes, every lane receives the total.  Order: lane pairs (l, l^1), then ^2,
// ^4, ^8, ^16, ^32.  quad_perm [1,0,3,2] / [2,3,0,1] are the xor-1 / xor-2 swaps; once quads
// (resp. 8-groups) are uniform row_half_mirror / row_mirror deliver the xor-4 / xor-8
// partner's value; ds_swizzle bit-mode xor 0x10 crosses the 16-lane rows; the two 32-lane
// halves are combined through SGPRs.  fp32 addition is commutative, so both partners of every
// pair compute bit-identical sums.
__device__ __forceinline__ float wave_allreduce_tree(float v) {
    v = v + dpp_mov<0xB1>(v);
    v = v + dpp_mov<0x4E>(v);
    v = v + dpp_mov<0x141>(v);
    v = v + dpp_mov<0x140>(v);
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
    const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    return lo + hi;
}

template <int VEC>
__device__ __forceinline__ float inlane_tree(float (&t)[VEC]) {
#pragma unroll
    for (int s = 1; s < VEC; s <<= 1) {
#pragma unroll
        for (int k = 0; k + s < VEC; k += 2 * s) t[k] = t[k] + t[k + s];
    }
    return t[0];
}

// ---- row load / store ---------------------------------------------------------------------------
template <int VEC, bool EXACT>
__device__ __forceinline__ void load_row(const float *src, uint32_t lane, uint32_t D, float (&out)[VEC]) {
    if constexpr (EXACT) {
        if constexpr (VEC == 1) {
            out[0] = src[lane];
        } else if constexpr (VEC == 2) {
            const float2 t = *reinterpret_cast<const float2 *>(src + lane * 2);
            out[0] = t.x; out[1] = t.y;
        } else {
#pragma unroll
            for (int q = 0; q < VEC / 4; ++q) {
                const float4 t = *reinterpret_cast<const float4 *>(src + lane * VEC + q * 4);
                out[4 * q + 0] = t.x; out[4 * q + 1] = t.y; out[4 * q + 2] = t.z; out[4 * q + 3] = t.w;
            }
        }
    } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const uint32_t d = lane * VEC + v;
            out[v] = d < D ? src[d] : 0.0f;
        }
    }
}

template <int VEC, bool EXACT>
__device__ __forceinline__ void store_row(float *dst, uint32_t lane, uint32_t D, const float (&in)[VEC]) {
    if constexpr (EXACT) {
        if constexpr (VEC == 1) {
            dst[lane] = in[0];
        } else if constexpr (VEC == 2) {
            *reinterpret_cast<float2 *>(dst + lane * 2) = make_float2(in[0], in[1]);
        } else {
#pragma unroll
            for (int q = 0; q < VEC / 4; ++q)
                *reinterpret_cast<float4 *>(dst + lane * VEC + q * 4) =
                    make_float4(in[4 * q + 0], in[4 * q + 1], in[4 * q + 2], in[4 * q + 3]);
        }
    } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const uint32_t d = lane * VEC + v;
            if (d < D) dst[d] = in[v];
        }
    }
}

// ---- scalars of the reference ----------------------------------------------------------------
// scale() as the reference compiles it (maxss then minss): NaN -> -5 (sample/algorithms.cpp:6-10)
__device__ __forceinline__ float clamp_ref(float f) {
    float t = (f > -5.0f) ? f : -5.0f;
    return (t < 5.0f) ? t : 5.0f;
}

// fast_SM (sample/algorithms.cpp:766-770); v == 6.0 reads one past the table there, clamped here
__device__ __forceinline__ float fast_sm(const float *table, float v) {
    if (v > 6.0f) return 1.0f;
    if (v < -6.0f) return 0.0f;
    const float res = (float)(2048 / (2.0 * 6.0));
    int idx = (int)(((double)v + 6.0) * (double)res);
    idx = idx > 2047 ? 2047 : (idx < 0 ? 0 : idx);
    return table[idx];
}

// One (row, other-row) interaction.  OPT 5: t-distribution kernel; OPT 6: sigmoid kernel.
// NEG selects the negative-sample (repulsive) form.
template <int OPT, int VEC, bool NEG>
__device__ __forceinline__ void pair_update(const float (&xi)[VEC], const float (&xj)[VEC], float (&Y)[VEC], float lr,
                                            double c0, const float *table) {
    float t[VEC];
    if constexpr (OPT == 5) {
        float diff[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            diff[v] = xi[v] - xj[v];
            t[v] = diff[v] * diff[v];
        }
        const float a = wave_allreduce_tree(inlane_tree<VEC>(t));
        float d1;
        if constexpr (NEG)
            d1 = (float)(2.0 / ((double)a * (1.0 + (double)a)));  // algorithms.cpp:622
        else
            d1 = (float)(-2.0 / (1.0 + (double)a));               // algorithms.cpp:608
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const float f = clamp_ref(diff[v] * d1);
            const float s = lr * f;
            Y[v] = Y[v] + s;
        }
    } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) t[v] = xi[v] * xj[v];
        const float a = wave_allreduce_tree(inlane_tree<VEC>(t));
        const float sm = fast_sm(table, a);
        if constexpr (!NEG) {
            const double coef = (1.0 - (double)sm) * c0;  // algorithms.cpp:867
#pragma unroll
            for (int v = 0; v < VEC; ++v) Y[v] = (float)((double)xj[v] * coef + (double)Y[v]);
        } else {
            const float w = lr * sm;  // algorithms.cpp:907
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float p = w * xj[v];
                Y[v] = Y[v] - p;
            }
        }
    }
}

// Walk a list of row ids (CSR neighbours, walk samples or negative samples): 64 ids per
// coalesced id load, U row gathers in flight before the first interaction is evaluated.
template <int OPT, int VEC, bool EXACT, bool NEG>
__device__ __forceinline__ void process_list(const StepArgs &a, const uint32_t *ids, uint32_t nb, uint32_t ne,
                                             uint32_t lane, const float (&xi)[VEC], float (&Y)[VEC], double c0) {
    constexpr int U = 8;
    const uint32_t D = a.D;
    for (uint32_t base = nb; base < ne; base += 64) {
        const uint32_t cnt = (ne - base) < 64u ? (ne - base) : 64u;
        const uint32_t idv = (lane < cnt) ? ids[base + lane] : 0u;
        for (uint32_t g = 0; g < cnt; g += U) {
            float xj[U][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t k = (g + u) < cnt ? (g + u) : (cnt - 1);
                const uint32_t j = (uint32_t)__builtin_amdgcn_readlane((int)idv, (int)k);
                const uint32_t pj = j - a.prev_lo;
                const float *src = (pj < a.prev_rows) ? a.stage_prev + (size_t)pj * D : a.X + (size_t)j * D;
                load_row<VEC, EXACT>(src, lane, D, xj[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (g + u < cnt) pair_update<OPT, VEC, NEG>(xi, xj[u], Y, a.lr, c0, a.sm_table);
            }
        }
    }
}

template <int OPT, int VEC, bool EXACT>
__global__ __launch_bounds__(256) void step_kernel(const StepArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wpb = blockDim.x >> 6;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * wpb + (threadIdx.x >> 6)));
    const uint32_t total_waves = gridDim.x * wpb;
    const uint32_t D = a.D;

    // (1) commit the pending minibatch: X[prev rows] = staged rows (K5, algorithms.cpp:629-639)
    for (uint32_t r = w; r < a.prev_rows; r += total_waves) {
        float t[VEC];
        load_row<VEC, EXACT>(a.stage_prev + (size_t)r * D, lane, D, t);
        store_row<VEC, EXACT>(a.X + (size_t)(a.prev_lo + r) * D, lane, D, t);
    }

    // (2) this wave's item: a hub chunk, or a whole row
    uint32_t row, nb, ne;
    bool last_chunk = true, first_chunk = true, partial = false;
    float *out;
    if (w < a.n_extra) {
        const uint2 e = a.extras[w];
        row = e.x;
        const uint32_t rp = a.rowptr[row], rpe = a.rowptr[row + 1];
        nb = rp + e.y * a.chunk;
        ne = (rpe - nb) > a.chunk ? nb + a.chunk : rpe;
        last_chunk = (ne == rpe);
        first_chunk = (e.y == 0);
        partial = true;
        out = a.partials + (size_t)w * D;
    } else {
        const uint32_t r = w - a.n_extra;
        if (r >= a.n_rows) return;
        row = a.row_lo + r;
        if (a.walk_mode) {
            nb = row * 5u;
            ne = nb + 5u;
        } else {
            nb = a.rowptr[row];
            ne = a.rowptr[row + 1];
            if (a.chunk != 0 && (ne - nb) > a.chunk) return;  // hub row: its chunks are extra items
        }
        out = a.stage_cur + (size_t)(row - a.batch_lo) * D;
    }

    float xi[VEC], Y[VEC];
    load_row<VEC, EXACT>(a.X + (size_t)row * D, lane, D, xi);
    double c0 = 0.0;
    if constexpr (OPT == 5) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) Y[v] = 0.0f;
    } else {
        // the sigmoid variants accumulate onto a copy of x_i (algorithms.cpp:824-831)
#pragma unroll
        for (int v = 0; v < VEC; ++v) Y[v] = first_chunk ? xi[v] : 0.0f;
        const uint32_t gdeg = a.rowptr[row + 1] - a.rowptr[row];
        const float degi = (float)(1.0 / (double)(gdeg + 1u));  // algorithms.cpp:854
        c0 = (double)(a.lr * degi);
    }

    process_list<OPT, VEC, EXACT, false>(a, a.nbr_ids, nb, ne, lane, xi, Y, c0);
    if (last_chunk) {
        const uint32_t sbase = a.bs_mode ? (row - a.batch_lo) : 0u;
        process_list<OPT, VEC, EXACT, true>(a, a.sample_ids, sbase, sbase + a.ns, lane, xi, Y, c0);
    }

    if constexpr (OPT == 5) {
        if (!partial) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) Y[v] = xi[v] + Y[v];  // algorithms.cpp:636
        }
    }
    store_row<VEC, EXACT>(out, lane, D, Y);
}

// Adds a hub row's chunk partials in chunk order and stages the row's new embedding.
template <int OPT, int VEC, bool EXACT>
__global__ __launch_bounds__(256) void hub_finalize_kernel(const FinalizeArgs f) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (w >= f.n_hubs) return;
    const HubRow h = f.hubs[w];
    const uint32_t D = f.D;
    const float *p = f.partials + (size_t)(h.slot0 - f.slot_base) * D;
    float Y[VEC], P[VEC];
    load_row<VEC, EXACT>(p, lane, D, Y);
    for (uint32_t c = 1; c < h.nchunks; ++c) {
        load_row<VEC, EXACT>(p + (size_t)c * D, lane, D, P);
#pragma unroll
        for (int v = 0; v < VEC; ++v) Y[v] = Y[v] + P[v];
    }
    if constexpr (OPT == 5) {
        float xi[VEC];
        load_row<VEC, EXACT>(f.X + (size_t)h.row * D, lane, D, xi);
#pragma unroll
        for (int v = 0; v < VEC; ++v) Y[v] = xi[v] + Y[v];
    }
    store_row<VEC, EXACT>(f.stage_cur + (size_t)(h.row - f.batch_lo) * D, lane, D, Y);
}

// X[lo + r] = stage[r] for r < rows (commit without a following step)
template <int VEC, bool EXACT>
__global__ __launch_bounds__(256) void commit_kernel(float *X, const float *stage, uint32_t lo, uint32_t rows, uint32_t D) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wpb = blockDim.x >> 6;
    const uint32_t total = gridDim.x * wpb;
    for (uint32_t r = blockIdx.x * wpb + (threadIdx.x >> 6); r < rows; r += total) {
        float t[VEC];
        load_row<VEC, EXACT>(stage + (size_t)r * D, lane, D, t);
        store_row<VEC, EXACT>(X + (size_t)(lo + r) * D, lane, D, t);
    }
}

// Self-test of the reduction order: out[r] = tree sum of in[r*width .. +width)
__global__ void wave_reduce_test_kernel(const float *in, uint32_t rows, uint32_t width, float *out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    uint32_t vec = 1;
    while (64u * vec < width) vec <<= 1;
    float t[8];
    for (uint32_t v = 0; v < 8; ++v) {
        const uint32_t d = lane * vec + v;
        t[v] = (v < vec && d < width) ? in[(size_t)r * width + d] : 0.0f;
    }
    const float s = wave_allreduce_tree(inlane_tree<8>(t));
    if (lane == 0) out[r] = s;
}

}  // namespace f2v
#endif

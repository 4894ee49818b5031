// f2v_kernels.hip.h -- the HIP/CDNA4 (gfx950) kernels of the Force2Vec hot path.
//
// Work decomposition: a wavefront (or, in the quarter-wave layout, a 16-lane DPP row) owns one
// source vertex (row) of the minibatch for the whole of its neighbour list:
//   * x_i and the row's force accumulator Y_i live in registers for the whole row,
//   * each CSR neighbour is one coalesced row gather from HBM straight into registers (read once per item);
//     the minibatch's negative-sample rows are staged in LDS once per workgroup and shared by its items,
//   * the squared distance (t-distribution, option 5) or dot product (sigmoid, options 6/7)
//     is reduced in ONE canonical order -- the balanced adjacent-pair binary tree over
//     next_pow2(D) zero-padded terms -- built from an in-lane pair tree and DPP lane-xor
//     steps; the test oracle restates that order on the CPU, so the kernels are checked
//     bit for bit,
//   * the fp64 scalars d1 / coef of the reference (sample/algorithms.cpp:608,622,867) are
//     computed per lane in fp64, the clamp keeps the compiled reference's NaN -> -5 rule, and
//     mul/add are NOT contracted (-ffp-contract=off).
// Two layouts share those semantics:
//   step_kernel  <OPT,VEC,EXACT>  any D <= 512: one item per wavefront, lane l owns dims
//                                 [l*VEC, l*VEC+VEC); butterfly = DPP xor 1,2,4,8, ds_swizzle 16,
//                                 v_readlane 32.
//   qstep_kernel <OPT,LPI,NB,U>   width 4*LPI*NB in {16, 32, 64, 128, 256}, D = the width or any smaller multiple
//                                 of 4: 64/LPI items per wavefront on LPI lanes of a DPP row each; every VALU
//                                 instruction serves 64/LPI pairs.
//
// Minibatch sequencing (Jacobi inside a batch, Gauss-Seidel across batches,
// sample/algorithms.cpp:588-639) with ONE launch per batch and no copy: the embedding matrix
// exists twice in HBM.  During an epoch the rows already updated (a growing contiguous range
// [upd_lo, upd_lo+upd_rows)) live in the NEW matrix Xn, all others in the current matrix X; the
// step kernel of a batch writes its rows to Xn and reads every other row (neighbour or negative
// sample) from Xn if it lies in the updated range, from X otherwise.  Rows of the CURRENT batch
// are outside that range, so they are read with their pre-batch values -- exactly the
// reference's snapshot semantics for samples and in-batch neighbours -- and no wave ever reads
// a row another wave of the same launch writes.  When the range covers all N rows the two
// matrices swap roles.  Per row the kernel moves 4D bytes in and 4D bytes out: the algorithmic
// minimum.
//
// Load balance (the GPU counterpart of option 11's nnz-balanced partition,
// sample/algorithms.cpp:2483-2523): the host cuts every minibatch into work ITEMS -- a whole
// row, or one `chunk`-neighbour piece of a hub row (degree > chunk) -- sorted longest first.
// Hub pieces leave partial sums in HBM; combine-tree nodes add them, `fanin` at a time, in chunk order -- in the
// sub-wave kernel as extra workgroups at the end of the same grid, so that a minibatch is one launch.
#ifndef F2V_KERNELS_HIP_H_
#define F2V_KERNELS_HIP_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace f2v {
#ifdef F2V_TEST_HOOKS
// The self-test build (libf2v_selftest.so) has a StepArgs of its own (four more fields) and kernels that read it: they are types
// and symbols of their own, so that no loader, however the two libraries meet in one process, can pair one build's launch stub
// or argument block with the other's kernel.
inline namespace selftest {
#endif

// One unit of work: a whole row, or one chunk of a hub row's neighbour list.
struct Item {
    uint32_t row;    // source vertex
    uint32_t nb;     // first index into nbr_ids
    uint32_t cnt;    // neighbours in this item
    uint32_t flags;  // kItem* bits | partial slot (low 28 bits)
};
constexpr uint32_t kItemPartial = 1u << 31;  // store the force sum to partials[slot] (hub chunk), not the new row
constexpr uint32_t kItemFirst = 1u << 30;    // first chunk of its row (sigmoid: accumulates onto x_i)
constexpr uint32_t kItemLast = 1u << 29;     // last chunk of its row: also takes the negative samples
constexpr uint32_t kItemSlotMask = (1u << 28) - 1;

// One node of a hub row's combine tree: add partial rows [in_slot, in_slot+n) in order.
struct FinItem {
    uint32_t in_slot, n;
    uint32_t out;  // partial slot of the sum, or kFinToStage: it is the row's total -> stage the new row
    uint32_t row;
};
constexpr uint32_t kFinToStage = 0xFFFFFFFFu;

constexpr int kMaxRanks = 8;  // multi-GPU push exchange: ranks of one xGMI hive

// Where a rank's new rows go besides its own second matrix (sharded runs): the same matrix of every peer whose bit
// is set in the row's reader mask.  world <= 1: nothing is pushed.
struct PushTargets {
    float *peer[kMaxRanks];   // where row `row_base` lands on every rank (mapped through HIP IPC; peer[self] is not written):
                              // the second matrix itself (row_base = 0), or -- matrices of 2 GiB and more cannot be mapped --
                              // a landing buffer holding one minibatch (row_base = first row of the minibatch)
    const uint32_t *masks;    // per vertex: bit r = rank r reads the row; nullptr = every rank does
    uint32_t self, world;
    uint32_t row_base;
};

struct StepArgs {
    const float *X;              // current N x D embedding matrix (row-major, fp32): rows not yet updated this epoch
    float *Xn;                   // new matrix: rows of the updated range, and this batch's output
    const uint32_t *rowptr;      // CSR row pointers [N+1]
    const uint32_t *nbr_ids;     // CSR colids, or the epoch's walk samples [5*N] (option 7)
    float *partials;             // hub chunk partial sums of this launch [slots x D]
    const uint32_t *sample_ids;  // negative-sample vertex ids of this minibatch (device)
    const Item *items;           // this launch's work items, longest first
    const float *sm_table;       // 2048-entry sigmoid table
    uint32_t D;
    uint32_t batch_lo;           // first row of the minibatch (-bs 1 sample window base)
    uint32_t n_items;
    uint32_t upd_lo, upd_rows;   // rows [upd_lo, upd_lo+upd_rows) are read from Xn
    uint32_t ns;
    uint32_t bs_mode;
    uint32_t unit_degi;          // option 10: the attraction is NOT divided by deg + 1 (the reference's AVX512 option 10 has `degi = 1.0`, algorithms.cpp:2155, :3793)
    float lr;
    PushTargets push;
    // sub-wave kernel only: the combine trees of this launch's hub rows run in the same grid, behind the step items
    const FinItem *fin_items;         // all levels, lowest first; nullptr: the trees run as launches of their own
    uint32_t fin_n;
    uint32_t step_blocks;             // workgroups [0, step_blocks) step items, the rest run tree nodes
    uint32_t seq;                     // this launch's sequence number for the ready flags
    uint32_t *ready;
    uint32_t *err;
    unsigned long long timeout_ticks;
    // chained minibatches (qstep_chain_kernel): rows [chain_lo, chain_lo + chain_rows) are written by EARLIER minibatches of
    // this very launch; a reader waits until rowflag[row] == seq before it loads one of them
    uint32_t *rowflag;
    uint32_t chain_lo, chain_rows;
#ifdef F2V_TEST_HOOKS
    uint32_t test_withhold_slot;      // f2v_test_withhold_flag: the piece with this partial slot never announces (kNoSlot: none)
    // f2v_test_stamps: per row four 100-MHz wall-clock words -- [0] its last hub piece announced, [1] its last inner tree node
    // announced, [2] its row flag stored, [3] ~(first time a waiter that had to wait saw that flag); nullptr: off
    unsigned long long *stamps;
    uint32_t test_withhold_row;       // f2v_test_withhold_row: chained launches never store this row's flag (kNoSlot: none)
    uint32_t test_nowait;             // f2v_test_chain_nowait (wide form): no row of the launch is waited for
    // f2v_test_xcd_times (one launch per minibatch): per XCD k -- [k] latest end of a workgroup, [8 + k] earliest start, [16 + k] sum of the
    // workgroups' durations, [24 + k] workgroups (100-MHz wall clock); nullptr: off
    unsigned long long *xcd_times;
#endif
};
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;

struct FinalizeArgs {
    const float *X;
    float *partials;
    float *Xn;
    const FinItem *items;
    uint32_t n_items;
    uint32_t D;
    PushTargets push;
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
template <int VEC>
__device__ __forceinline__ void load_vec(const float *p, float (&out)[VEC]) {
    if constexpr (VEC == 1) {
        out[0] = p[0];
    } else if constexpr (VEC == 2) {
        const float2 t = *reinterpret_cast<const float2 *>(p);
        out[0] = t.x; out[1] = t.y;
    } else {
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q) {
            const float4 t = *reinterpret_cast<const float4 *>(p + q * 4);
            out[4 * q + 0] = t.x; out[4 * q + 1] = t.y; out[4 * q + 2] = t.z; out[4 * q + 3] = t.w;
        }
    }
}

template <int VEC>
__device__ __forceinline__ void store_vec(float *p, const float (&in)[VEC]) {
    if constexpr (VEC == 1) {
        p[0] = in[0];
    } else if constexpr (VEC == 2) {
        *reinterpret_cast<float2 *>(p) = make_float2(in[0], in[1]);
    } else {
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q)
            *reinterpret_cast<float4 *>(p + q * 4) = make_float4(in[4 * q + 0], in[4 * q + 1], in[4 * q + 2], in[4 * q + 3]);
    }
}

// EXACT: D == 64*VEC, every lane owns VEC live dims.  Otherwise lanes past D hold zeros; when D is a multiple of
// VEC (rows then stay aligned to the vector width) the live lanes still use one vector access.
template <int VEC, bool EXACT>
__device__ __forceinline__ void load_row(const float *src, uint32_t lane, uint32_t D, float (&out)[VEC]) {
    if constexpr (EXACT) {
        load_vec<VEC>(src + lane * VEC, out);
    } else if (D % VEC == 0) {
        if (lane * VEC < D) {
            load_vec<VEC>(src + lane * VEC, out);
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) out[v] = 0.0f;
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
        store_vec<VEC>(dst + lane * VEC, in);
    } else if (D % VEC == 0) {
        if (lane * VEC < D) store_vec<VEC>(dst + lane * VEC, in);
    } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const uint32_t d = lane * VEC + v;
            if (d < D) dst[d] = in[v];
        }
    }
}

// A row store that is written through to the peer's memory: relaxed system-scope atomic stores carry the sc0 sc1
// bits, so the data does not linger in this GPU's L2 and no cache write-back is needed afterwards.
template <int VEC, bool EXACT>
__device__ __forceinline__ void store_row_system(float *dst, uint32_t lane, uint32_t D, const float (&in)[VEC]) {
    if constexpr (VEC >= 2) {
        if (EXACT || D % VEC == 0) {
            if (EXACT || lane * VEC < D) {
                unsigned long long *q = reinterpret_cast<unsigned long long *>(dst + lane * VEC);
#pragma unroll
                for (int v = 0; v < VEC; v += 2) {
                    const unsigned long long bits = (unsigned long long)__builtin_bit_cast(uint32_t, in[v]) |
                                                    ((unsigned long long)__builtin_bit_cast(uint32_t, in[v + 1]) << 32);
                    __hip_atomic_store(q + v / 2, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            return;
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const uint32_t d = lane * VEC + v;
        if (d < D) __hip_atomic_store(reinterpret_cast<uint32_t *>(dst + d), __builtin_bit_cast(uint32_t, in[v]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));
// 16 bytes per lane, written through at system scope (global_store_dwordx4 sc0 sc1): consecutive lanes write
// consecutive 16-byte pieces, so a 16-lane item sends whole 256-byte runs down the link
__device__ __forceinline__ void store16_system(float *p, const float4 v) {
    const f32x4_t x = {v.x, v.y, v.z, v.w};
    // s_nop 1 inside the statement: hipcc pads no hazards of an asm instruction, and its next instruction may otherwise
    // overwrite the four data registers before the store has read them (cdna_hip_programming.md 5.7 item 1)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
}

// the same written through at agent scope (sc1): visible to the other XCDs' L2s once acknowledged
__device__ __forceinline__ void store16_agent(float *p, const float4 v) {
    const f32x4_t x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
}

// The wave's row regrouped into 16-byte pieces: every written-through store is one fabric write, and an 8-byte one
// costs 2.7x, a 4-byte one ~6x the time per byte of a 16-byte one (MI355X_MICROARCH.md).  VEC = 2: lane pairs (xor 1),
// the even lane holds four consecutive dims; VEC = 1: quads, lane 4k holds them.  Valid when D is a multiple of 4.
template <int VEC>
__device__ __forceinline__ bool gather16(const float (&in)[VEC], uint32_t lane, float4 (&out)[VEC >= 4 ? VEC / 4 : 1], uint32_t &first_dim) {
    if constexpr (VEC >= 4) {
#pragma unroll
        for (int k = 0; k < VEC / 4; ++k) out[k] = make_float4(in[4 * k], in[4 * k + 1], in[4 * k + 2], in[4 * k + 3]);
        first_dim = lane * VEC;
        return true;
    } else if constexpr (VEC == 2) {
        const float o0 = dpp_mov<0xB1>(in[0]), o1 = dpp_mov<0xB1>(in[1]);  // the partner lane's pair
        out[0] = make_float4(in[0], in[1], o0, o1);
        first_dim = lane * 2u;
        return (lane & 1u) == 0u;
    } else {
        const float a = dpp_mov<0x55>(in[0]), b = dpp_mov<0xAA>(in[0]), c = dpp_mov<0xFF>(in[0]);  // quad lanes 1, 2, 3
        out[0] = make_float4(in[0], a, b, c);
        first_dim = lane;
        return (lane & 3u) == 0u;
    }
}

// push one finished row (held by the wave, VEC values per lane) to the peers that read it
template <int VEC, bool EXACT>
__device__ __forceinline__ void push_row(const PushTargets &t, uint32_t row, uint32_t lane, uint32_t D, const float (&in)[VEC]) {
    const uint32_t others = ((1u << t.world) - 1u) & ~(1u << t.self);
    const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)((t.masks ? t.masks[row] : others) & others));
    if (D % 4u == 0u) {  // rows are 16-byte aligned: 16-byte stores
        float4 piece[VEC >= 4 ? VEC / 4 : 1];
        uint32_t d0;
        const bool mine = gather16<VEC>(in, lane, piece, d0) && d0 < D;
#pragma unroll
        for (int q = 0; q < kMaxRanks; ++q) {
            if ((m & (1u << q)) && mine) {
                float *dst = t.peer[q] + (size_t)(row - t.row_base) * D + d0;
#pragma unroll
                for (int k = 0; k < (VEC >= 4 ? VEC / 4 : 1); ++k)
                    if (d0 + 4u * k < D) store16_system(dst + 4 * k, piece[k]);
            }
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < kMaxRanks; ++q) {
        if (m & (1u << q)) store_row_system<VEC, EXACT>(t.peer[q] + (size_t)(row - t.row_base) * D, lane, D, in);
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
                const float *src = ((j - a.upd_lo) < a.upd_rows ? a.Xn : a.X) + (size_t)j * D;
                load_row<VEC, EXACT>(src, lane, D, xj[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (g + u < cnt) pair_update<OPT, VEC, NEG>(xi, xj[u], Y, a.lr, c0, a.sm_table);
            }
        }
    }
}

// Generic layout (any D <= 512): one wavefront per work item.
template <int OPT, int VEC, bool EXACT>
__global__ __launch_bounds__(256) void step_kernel(const StepArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wpb = blockDim.x >> 6;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * wpb + (threadIdx.x >> 6)));
    const uint32_t D = a.D;
    if (w >= a.n_items) return;

    const Item it = a.items[w];
    const uint32_t row = it.row;
    const bool partial = (it.flags & kItemPartial) != 0;
    const bool first_chunk = (it.flags & kItemFirst) != 0;
    const bool last_chunk = (it.flags & kItemLast) != 0;
    float *out = partial ? a.partials + (size_t)(it.flags & kItemSlotMask) * D : a.Xn + (size_t)row * D;

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
        const float degi = a.unit_degi ? 1.0f : (float)(1.0 / (double)(gdeg + 1u));  // algorithms.cpp:854
        c0 = (double)(a.lr * degi);
    }

    process_list<OPT, VEC, EXACT, false>(a, a.nbr_ids, it.nb, it.nb + it.cnt, lane, xi, Y, c0);
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
    if (a.push.world > 1u) {  // sharded run: the new row also goes to the peers that read it
        if (!partial) push_row<VEC, EXACT>(a.push, row, lane, D, Y);
        __builtin_amdgcn_s_waitcnt(0);
    }
}

// ---- all levels of the combine trees in ONE launch ---------------------------------------------------------
// The upper levels of the trees hold a handful of nodes (the rows with more than `fanin` chunks), so a launch per
// level is pure launch latency on the critical path of every minibatch.  hub_finalize_tree_kernel runs every
// level's nodes in one grid, lowest level first: a node of an upper level waits until the nodes it adds have
// announced their sums -- ready[slot] == seq, the launch's sequence number -- which they do after storing them
// written through at agent scope (sc1: visible to every XCD's L2) and waiting for the stores to be acknowledged.
// Forward progress: a node only ever waits for nodes with a SMALLER index, and workgroups are dispatched in index
// order, so whatever a wave waits for is already running or done; every wait is bounded all the same (err = 2).
// The sub-wave step kernel goes one step further and appends the trees to its OWN grid (StepArgs::tree): the hub
// pieces announce their partial sums the same way, and a minibatch is ONE launch.
struct FinalizeTreeArgs {
    FinalizeArgs f;           // items: all levels, lowest first; n_items: all of them
    uint32_t *rowflag;        // chained minibatches: per row, the launch that last wrote it
    uint32_t *ready;          // per partial slot: sequence number of the launch that last produced it
    uint32_t *err;
    unsigned long long timeout_ticks;
    uint32_t seq;
    uint32_t first_dep;       // items from this index on add sums produced INSIDE this launch
#ifdef F2V_TEST_HOOKS
    unsigned long long *stamps;
    uint32_t test_withhold_row;
#endif
};

template <int VEC, bool EXACT>
__device__ __forceinline__ void load_row_agent(const float *src, uint32_t lane, uint32_t D, float (&out)[VEC]) {
    if constexpr (VEC >= 2) {
        if (EXACT || D % VEC == 0) {
            if (EXACT || lane * VEC < D) {
                const unsigned long long *q = reinterpret_cast<const unsigned long long *>(src + lane * VEC);
#pragma unroll
                for (int v = 0; v < VEC; v += 2) {
                    const unsigned long long bits = __hip_atomic_load(q + v / 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    out[v] = __builtin_bit_cast(float, (uint32_t)bits);
                    out[v + 1] = __builtin_bit_cast(float, (uint32_t)(bits >> 32));
                }
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) out[v] = 0.0f;
            }
            return;
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const uint32_t d = lane * VEC + v;
        out[v] = d < D ? __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const uint32_t *>(src + d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.0f;
    }
}

template <int VEC, bool EXACT>
__device__ __forceinline__ void store_row_agent(float *dst, uint32_t lane, uint32_t D, const float (&in)[VEC]) {
    if constexpr (VEC >= 2) {
        if (EXACT || D % VEC == 0) {
            if (EXACT || lane * VEC < D) {
                unsigned long long *q = reinterpret_cast<unsigned long long *>(dst + lane * VEC);
#pragma unroll
                for (int v = 0; v < VEC; v += 2) {
                    const unsigned long long bits = (unsigned long long)__builtin_bit_cast(uint32_t, in[v]) |
                                                    ((unsigned long long)__builtin_bit_cast(uint32_t, in[v + 1]) << 32);
                    __hip_atomic_store(q + v / 2, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            return;
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const uint32_t d = lane * VEC + v;
        if (d < D) __hip_atomic_store(reinterpret_cast<uint32_t *>(dst + d), __builtin_bit_cast(uint32_t, in[v]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Y = p[0] + p[1] + ... + p[n-1] in order, 16 row loads in flight; COHERENT: the rows were produced inside this launch
template <int VEC, bool EXACT, bool COHERENT>
__device__ __forceinline__ void add_partials(const float *p, uint32_t n, uint32_t lane, uint32_t D, float (&Y)[VEC]) {
    if constexpr (COHERENT) load_row_agent<VEC, EXACT>(p, lane, D, Y);
    else load_row<VEC, EXACT>(p, lane, D, Y);
    constexpr int U = 16;
    for (uint32_t c = 1; c < n; c += U) {
        float P[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t k = (c + u) < n ? (c + u) : (n - 1);
            if constexpr (COHERENT) load_row_agent<VEC, EXACT>(p + (size_t)k * D, lane, D, P[u]);
            else load_row<VEC, EXACT>(p + (size_t)k * D, lane, D, P[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c + u < n) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) Y[v] = Y[v] + P[u][v];
            }
        }
    }
}

// one node of the trees: wave `w` of the nodes' part of a grid.  ROW_THROUGH: the row's new embedding is written through at
// agent scope too (chained minibatches: a later minibatch of the SAME launch reads it, possibly on another XCD).
template <int OPT, int VEC, bool EXACT, bool ROW_THROUGH = false>
__device__ __forceinline__ void finalize_tree_node(const FinalizeTreeArgs &a, uint32_t w, uint32_t lane) {
    const FinalizeArgs &f = a.f;
    if (w >= f.n_items) return;
    const FinItem h = f.items[w];
    if (h.n == 0u) return;  // padding
    const uint32_t D = f.D;
    const float *p = f.partials + (size_t)h.in_slot * D;
    float Y[VEC];
    if (w >= a.first_dep) {
        // Wait for the sums this node adds.  A wait that gives up -- its own time-out, or another node's (err is set: the
        // launch is lost) -- makes the node return WITHOUT storing or announcing anything: a sum that was never announced is
        // never added, whoever waits for this node gives up in turn, and the grid drains at once.  The rows such nodes were
        // to produce keep their old contents; the host sees err at the end of the epoch and fails the call (F2V_ESTATE).
        const unsigned long long t0 = wall_clock64();
        bool gave_up = false;
        for (uint32_t c = lane; c < h.n && !gave_up; c += 64u) {
            uint32_t seen, spins = 0;
            while ((seen = __hip_atomic_load(a.ready + h.in_slot + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != a.seq) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 15u) != 0u) continue;
                if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { gave_up = true; break; }
                if (wall_clock64() - t0 > a.timeout_ticks) {
                    // err[0] code, [1] how many waits timed out, [2..7] the first of them: node, slot, flag seen, seq, node count, first dependent
                    if (__hip_atomic_fetch_add(a.err + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                        a.err[2] = w; a.err[3] = h.in_slot + c; a.err[4] = seen; a.err[5] = a.seq; a.err[6] = f.n_items; a.err[7] = a.first_dep;
                    }
                    __hip_atomic_store(a.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gave_up = true;
                    break;
                }
            }
        }
        if (__builtin_amdgcn_ballot_w64(gave_up) != 0ull) return;
        // every poll of this wave has returned its value before the first load of a sum is issued (and the compiler may
        // not move those loads up: relaxed atomics of different addresses are otherwise unordered for it)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        add_partials<VEC, EXACT, true>(p, h.n, lane, D, Y);
    } else {
        add_partials<VEC, EXACT, false>(p, h.n, lane, D, Y);
    }
    if (h.out == kFinToStage) {
        if constexpr (OPT == 5) {
            float xi[VEC];
            load_row<VEC, EXACT>(f.X + (size_t)h.row * D, lane, D, xi);
#pragma unroll
            for (int v = 0; v < VEC; ++v) Y[v] = xi[v] + Y[v];
        }
        if constexpr (ROW_THROUGH) {
            store_row_agent<VEC, EXACT>(f.Xn + (size_t)h.row * D, lane, D, Y);
            __builtin_amdgcn_s_waitcnt(0);  // the row is in memory before it is announced
#ifdef F2V_TEST_HOOKS
            if (h.row == a.test_withhold_row) return;  // fault injection: this row is never announced
#endif
            if (lane == 0) __hip_atomic_store(a.rowflag + h.row, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef F2V_TEST_HOOKS
            if (a.stamps && lane == 0) a.stamps[4 * (size_t)h.row + 2] = wall_clock64();
#endif
        } else {
            store_row<VEC, EXACT>(f.Xn + (size_t)h.row * D, lane, D, Y);
        }
        if (f.push.world > 1u) {
            push_row<VEC, EXACT>(f.push, h.row, lane, D, Y);
            __builtin_amdgcn_s_waitcnt(0);
        }
    } else {
        store_row_agent<VEC, EXACT>(f.partials + (size_t)h.out * D, lane, D, Y);
        __builtin_amdgcn_s_waitcnt(0);  // the sum is in memory before it is announced
        if (lane == 0) __hip_atomic_store(a.ready + h.out, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef F2V_TEST_HOOKS
        if (a.stamps && lane == 0) atomicMax(a.stamps + 4 * (size_t)h.row + 1, wall_clock64());
#endif
    }
}

template <int OPT, int VEC, bool EXACT>
__global__ __launch_bounds__(256) void hub_finalize_tree_kernel(const FinalizeTreeArgs a) {
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    finalize_tree_node<OPT, VEC, EXACT>(a, w, threadIdx.x & 63u);
}

// ---- sub-wave layout: LPI lanes per item, D = 4*LPI*NB ----------------------------------------------
// 64/LPI work items per wavefront, each on LPI consecutive lanes of a 16-lane DPP row: LPI = 16 with NB = 1, 2, 4
// blocks (D = 64, 128, 256: four items per wavefront, the "quarter-wave" layout of the flagship D = 128), LPI = 8
// (D = 32, eight items) and LPI = 4 (D = 16, sixteen items).  Lane t of an item owns, in each block b of 4*LPI dims,
// the four contiguous dims [4*LPI*b + 4t, +4): every row is NB coalesced (LPI lanes x dwordx4) loads, a block sum
// is an in-lane pair tree plus log2(LPI) DPP row steps (xor 1, 2, [4, [8]] -- no LDS crossbar, no SGPR hop), and
// the blocks are added pairwise: the same canonical adjacent-pair tree as the generic layout, so results are
// bit-identical.  Every VALU instruction serves 64/LPI (row, neighbour) pairs instead of one, which takes the
// kernel from VALU-bound (14 fp64 + 20 fp32 ops per pair) to gather-bound.  Items arrive sorted by length, so
// the items of a wave run almost equally long.
template <int LPI>
__device__ __forceinline__ float item_allreduce_tree(float v) {
    v = v + dpp_mov<0xB1>(v);
    v = v + dpp_mov<0x4E>(v);
    if constexpr (LPI >= 8) v = v + dpp_mov<0x141>(v);
    if constexpr (LPI >= 16) v = v + dpp_mov<0x140>(v);
    return v;
}

// One (row, other row) interaction in two parts.  pair_coef_q: the pair's scalar -- the t-distribution's d1 (option 5), the
// attraction's (1 - sigmoid) * c0 in fp64 or the repulsion's lr * sigmoid (options 6, 7) -- which needs x_i and the other row
// only; pair_apply_q: the other row's contribution added onto Y, which must happen in list order.  pair_update_q is one after the
// other; the wide chained kernel computes the scalars of the rows that have arrived while it waits for a late one.
template <int OPT, bool NEG>
struct PairCoef { using type = float; };
template <>
struct PairCoef<6, false> { using type = double; };  // (the kernels' OPT is 5 or 6: option 7 runs option 6's arithmetic on walk samples)

template <int OPT, int LPI, int NB, bool NEG>
__device__ __forceinline__ typename PairCoef<OPT, NEG>::type pair_coef_q(const float (&xi)[NB][4], const float4 (&xj4)[NB], float lr, double c0,
                                                                         const float *table) {
    float xj[NB][4], bs[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        xj[b][0] = xj4[b].x; xj[b][1] = xj4[b].y; xj[b][2] = xj4[b].z; xj[b][3] = xj4[b].w;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float t[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            if constexpr (OPT == 5) {
                const float d = xi[b][v] - xj[b][v];
                t[v] = d * d;
            } else {
                t[v] = xi[b][v] * xj[b][v];
            }
        }
        bs[b] = item_allreduce_tree<LPI>((t[0] + t[1]) + (t[2] + t[3]));
    }
    float a;
    if constexpr (NB == 1) a = bs[0];
    else if constexpr (NB == 2) a = bs[0] + bs[1];
    else a = (bs[0] + bs[1]) + (bs[2] + bs[3]);
    if constexpr (OPT == 5) {
        if constexpr (NEG) return (float)(2.0 / ((double)a * (1.0 + (double)a)));  // algorithms.cpp:622
        else return (float)(-2.0 / (1.0 + (double)a));                              // algorithms.cpp:608
    } else {
        const float sm = fast_sm(table, a);
        if constexpr (!NEG) return (1.0 - (double)sm) * c0;  // algorithms.cpp:867
        else return lr * sm;                                  // algorithms.cpp:907
    }
}

template <int OPT, int LPI, int NB, bool NEG>
__device__ __forceinline__ void pair_apply_q(const float (&xi)[NB][4], const float4 (&xj4)[NB], float (&Y)[NB][4], float lr,
                                             typename PairCoef<OPT, NEG>::type coef) {
    float xj[NB][4];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        xj[b][0] = xj4[b].x; xj[b][1] = xj4[b].y; xj[b][2] = xj4[b].z; xj[b][3] = xj4[b].w;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            if constexpr (OPT == 5) {
                const float f = clamp_ref((xi[b][v] - xj[b][v]) * coef);
                const float s = lr * f;
                Y[b][v] = Y[b][v] + s;
            } else if constexpr (!NEG) {
                Y[b][v] = (float)((double)xj[b][v] * coef + (double)Y[b][v]);
            } else {
                const float p = coef * xj[b][v];
                Y[b][v] = Y[b][v] - p;
            }
        }
    }
}

template <int OPT, int LPI, int NB, bool NEG>
__device__ __forceinline__ void pair_update_q(const float (&xi)[NB][4], const float4 (&xj4)[NB], float (&Y)[NB][4],
                                              float lr, double c0, const float *table) {
    pair_apply_q<OPT, LPI, NB, NEG>(xi, xj4, Y, lr, pair_coef_q<OPT, LPI, NB, NEG>(xi, xj4, lr, c0, table));
}

// Chained minibatches: row j is written by an earlier minibatch of this launch -- wait until it has been announced
// (rowflag[j] == seq, stored by its writer after the written-through row was acknowledged).  -> true: gave up (time-out, or
// the launch is lost already); the caller then stores nothing.
// (forceinline on purpose: as a real call -- __noinline__, to keep the cold path out of the loop -- it made the whole chained
// kernel 3x slower: 12.4 instead of 4.7 ms per epoch without a single wait taken, tools/chain_probe_variants.py)
__device__ __forceinline__ bool wait_row_slow(const StepArgs &a, const uint32_t *flags, uint32_t j) {
    const uint32_t *f = flags + j;
    const unsigned long long t0 = wall_clock64();
    for (uint32_t spins = 1;; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.seq) {
#ifdef F2V_TEST_HOOKS
            if (a.stamps) atomicMax(a.stamps + 4 * (size_t)j + 3, ~wall_clock64());
#endif
            return false;
        }
        if ((spins & 15u) != 0u) continue;
        if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return true;
        if (wall_clock64() - t0 > a.timeout_ticks) {
            // err[0] code 3, [1] how many waits timed out, [2..7] the first: waiting workgroup, row, flag seen, seq, grid, minibatch's first row
            if (__hip_atomic_fetch_add(a.err + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                a.err[2] = blockIdx.x; a.err[3] = j; a.err[4] = *f; a.err[5] = a.seq; a.err[6] = gridDim.x; a.err[7] = a.batch_lo;
            }
            __hip_atomic_store(a.err, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return true;
        }
    }
}

// (`flags`: the array the row's writer announces in -- the launch's own, or, where a launch chains several epochs, an earlier epoch's)
__device__ __forceinline__ bool wait_row_at(const StepArgs &a, const uint32_t *flags, uint32_t j) {
    bool bad = false;
    if (__hip_atomic_load(flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.seq) bad = wait_row_slow(a, flags, j);
    asm volatile("" ::: "memory");  // the row's loads stay behind the poll
    return bad;
}
__device__ __forceinline__ bool wait_row(const StepArgs &a, uint32_t j) { return wait_row_at(a, a.rowflag, j); }

__device__ __forceinline__ const float *row_src(const StepArgs &a, uint32_t j, uint32_t D) {
    return ((j - a.upd_lo) < a.upd_rows ? a.Xn : a.X) + (size_t)j * D;
}

// 16 bytes of a row that another workgroup of THIS launch has written (written through, sc1) and announced: ONE 16-byte load at
// agent scope (buffer_load_dwordx4 ... sc1: past the L1, which no store of another CU ever refreshes) -- the measured hand-off
// recipe of MI355X_MICROARCH.md (table row 1: sc1 stores, the storing wave's vmcnt(0), sc1 flag store, sc1 poll, and EVERY load
// of the handed-off bytes a global_/buffer_ sc1 load of 4, 8 or 16 bytes).  __hip_atomic_load stops at 64 bits -- rounds 2-3 read
// such rows as two global_load_dwordx2 sc1, which the guide prices at 0.54-0.70x the 16-byte rate -- so the load is the raw
// buffer form (the compiler tracks it in vmcnt like any other load; an inline-asm global_load would need hand-kept counters).
// The buffer resource must be wave-uniform while an item's row is not: the resource spans everything a launch can hand on, and
// the row is a 32-bit byte offset (the host keeps every such span below 4 GiB: chain_len, epochs_max, wide_plan_append / wide_plans_for_epoch).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t hand_rsrc(const float *base) {
    // raw buffer: stride 0, no range check (num_records = 2^32 - 1), gfx9 dword 3 with DATA_FORMAT = 32 bits
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, -1, 0x00020000);
}
__device__ __forceinline__ float4 load16_agent(rsrc_t r, uint32_t byte_off) {
    // (bit_cast of the builtin's own vector type: initialising an ext_vector int4 from it compiles to a ONE-dword load + splat)
    const f32x4_t v = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, /*aux: sc1*/ 16));
    return make_float4(v.x, v.y, v.z, v.w);
}
// Where the rows a launch hands on live: row j of the second matrix (written by an earlier minibatch of the launch) at byte
// xn_off + j * row_bytes of `r`, row j of the first matrix (a launch that chains epochs: written by its previous epoch) at x_off + ...
// -- uint32 arithmetic, wrapping: xn_off is "minus the launch's first row" where the resource starts at that row.
struct HandSrc {
    rsrc_t r;
    uint32_t xn_off, x_off, row_bytes;
    __device__ __forceinline__ uint32_t at(uint32_t j, bool second) const { return j * row_bytes + (second ? xn_off : x_off); }
};
// one launch of minibatches [chain_lo, ...) over the two matrices: handed rows are rows of the second matrix from chain_lo on
__device__ __forceinline__ HandSrc hand_src_window(const StepArgs &a, uint32_t D) {
    HandSrc h;
    h.r = hand_rsrc(a.Xn + (size_t)a.chain_lo * D);
    h.row_bytes = D * 4u;
    h.xn_off = 0u - a.chain_lo * h.row_bytes;
    h.x_off = 0u;  // (unused: nothing of the first matrix is handed on)
    return h;
}

// One item's list of row ids; `cnt` is this item's length, `maxcnt` the wave's (uniform).
// U rows per item are in flight before the first interaction is evaluated; the ids of the next group are
// fetched one group ahead.
// FULL: D == 4*LPI*NB.  Otherwise D is any smaller multiple of 4 (rows stay 16-byte aligned): lane t's block b is live
// iff 4*LPI*b + 4t < D, dead pieces read as zero -- the zero padding of the canonical tree -- and are never stored.
template <int OPT, int LPI, int NB, bool NEG, int U, bool FULL, bool CHAIN>
__device__ __forceinline__ void qprocess(const StepArgs &a, const HandSrc &hs, const uint32_t *ids, uint32_t cnt, uint32_t maxcnt, uint32_t t, uint32_t D,
                                         const float (&xi)[NB][4], float (&Y)[NB][4], double c0, const float *table, bool &bad) {
    uint32_t j[U];
#pragma unroll
    for (int u = 0; u < U; ++u) j[u] = ((uint32_t)u < cnt) ? ids[u] : 0u;
    for (uint32_t g = 0; g < maxcnt; g += U) {
        float4 xj[U][NB];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (g + u < cnt) {
                bool handed = false;  // the row was written inside this launch: wait for its flag, then agent-scope loads
                if constexpr (CHAIN) {
                    handed = (j[u] - a.chain_lo) < a.chain_rows;
                    if (handed) bad = wait_row(a, j[u]) || bad;
                }
                const float *src = row_src(a, j[u], D) + t * 4;
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if (!(FULL || 4u * LPI * b + 4u * t < D)) xj[u][b] = make_float4(0.f, 0.f, 0.f, 0.f);
                    else if (CHAIN && handed) xj[u][b] = load16_agent(hs.r, hs.at(j[u], true) + 16u * t + 16u * LPI * b);
                    else xj[u][b] = *reinterpret_cast<const float4 *>(src + 4 * LPI * b);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) j[u] = (g + U + u < cnt) ? ids[g + U + u] : 0u;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (g + u < cnt) pair_update_q<OPT, LPI, NB, NEG>(xi, xj[u], Y, a.lr, c0, table);
        }
    }
}

// The same walk for the rounds of a wide program (always a chained launch, always attracting neighbours): the first U ids of
// the list arrive in `j` (loaded a round ago), and once the list's last gathers have been issued the first U ids of the NEXT
// round's list are requested into `j` -- behind the gathers, so that waiting for the rows does not wait for them.  A round
// then costs one memory latency (the rows) instead of three dependent ones (item, ids, rows).
// `between` runs once per call, when the list's first gathers are out and before anything is waited for (all threads of the
// workgroup reach it: it may hold a barrier; -> true: the workgroup gives up).
template <int OPT, int LPI, int NB, int U, bool FULL, bool SKIP, class Between>
__device__ __forceinline__ void qprocess_pre(const StepArgs &a, const HandSrc &hs, const uint32_t *ids, uint32_t cnt, uint32_t maxcnt, uint32_t t, uint32_t D,
                                             const float (&xi)[NB][4], float (&Y)[NB][4], double c0, const float *table, bool &bad,
                                             uint32_t (&j)[U], const uint32_t *next_ids, uint32_t next_cnt, Between &&between,
                                             const uint32_t *prev_flags) {
    // prev_flags (a launch that chains epochs, from its second epoch on): the rows this epoch has not updated yet were written by
    // the launch's PREVIOUS epoch -- they are awaited too, in that epoch's flag array, and read from the first matrix at agent scope
    uint32_t g = 0;
    do {
        float4 xj[U][NB];
        bool handed[U], inr[U];
        uint32_t fl[U];
        // rows written inside this launch: ALL their flags are requested first (side by side, behind nothing) ...
#pragma unroll
        for (int u = 0; u < U; ++u) {
            inr[u] = (j[u] - a.chain_lo) < a.chain_rows;
            handed[u] = (g + u < cnt) && (inr[u] || prev_flags != nullptr);
            fl[u] = 0u;
            if (handed[u]) fl[u] = __hip_atomic_load((inr[u] ? a.rowflag : prev_flags) + j[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ... then the rows that need no wait ...
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (g + u < cnt && !handed[u]) {
                const float *src = row_src(a, j[u], D) + t * 4;
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if (!(FULL || 4u * LPI * b + 4u * t < D)) xj[u][b] = make_float4(0.f, 0.f, 0.f, 0.f);
                    else xj[u][b] = *reinterpret_cast<const float4 *>(src + 4 * LPI * b);
                }
            }
        }
        uint32_t j0[U];
#pragma unroll
        for (int u = 0; u < U; ++u) j0[u] = j[u];
        const bool last = g + U >= maxcnt;  // (uniform)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (last) j[u] = ((uint32_t)u < next_cnt) ? next_ids[u] : 0u;
            else j[u] = (g + U + u < cnt) ? ids[g + U + u] : 0u;
        }
        // SKIP (the form for large graphs, where most groups of most wavefronts hold no handed row at all): one wave-uniform branch
        // over the handed-row blocks instead of a dozen per-row ones -- batch 1024 / 2048 on RMAT-20 -1 ... -2 %.  Not in the forms for
        // small graphs (a launch is one dependency chain, nearly every group holds handed rows): there the extra branch COST 4 % on
        // cora at D = 16 (0.0676 -> 0.0702 s per 1200 epochs; profiles/r04_cora_regression_found.txt)
        bool anyh = true;
        if constexpr (SKIP) {
            bool anyh_lane = false;
#pragma unroll
            for (int u = 0; u < U; ++u) anyh_lane = anyh_lane || handed[u];
            anyh = __builtin_amdgcn_ballot_w64(anyh_lane) != 0ull;
        }
        // ... then, side by side again, every handed row whose flag was already up (most: they were written minibatches ago)
        if (anyh) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (handed[u] && fl[u] == a.seq) {
                    asm volatile("" ::: "memory");  // the row's loads stay behind the poll
                    const uint32_t off = hs.at(j0[u], inr[u]) + 16u * t;
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        if (!(FULL || 4u * LPI * b + 4u * t < D)) xj[u][b] = make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef F2V_TEST_HOOKS
                        else if (a.test_nowait & 4u) xj[u][b] = *reinterpret_cast<const float4 *>((inr[u] ? a.Xn : a.X) + (size_t)j0[u] * D + t * 4 + 4 * LPI * b);
#endif
                        else xj[u][b] = load16_agent(hs.r, off + 16u * LPI * b);
                    }
                    handed[u] = false;
                }
            }
        }
        if (g == 0u && between()) {
            bad = true;
            return;
        }
        // the late rows' flags once more (in flight while the scalars of the rows that are here are computed: when a late row
        // arrives, its own scalar and the additions in list order are all that is left of the item)
        if (anyh) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (handed[u]) fl[u] = __hip_atomic_load((inr[u] ? a.rowflag : prev_flags) + j0[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (anyh && prev_flags != nullptr) {
            // (a launch that chains epochs: EVERY row is behind a flag, and when a workgroup starts -- epochs ahead of its turn -- none
            // is up; the rows whose flags this second look finds up are requested side by side, not one by one in the loop below)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (handed[u] && fl[u] == a.seq) {
                    asm volatile("" ::: "memory");
                    const uint32_t off = hs.at(j0[u], inr[u]) + 16u * t;
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        if (!(FULL || 4u * LPI * b + 4u * t < D)) xj[u][b] = make_float4(0.f, 0.f, 0.f, 0.f);
                        else xj[u][b] = load16_agent(hs.r, off + 16u * LPI * b);
                    }
                    handed[u] = false;
                }
            }
        }
        typename PairCoef<OPT, false>::type cf[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            cf[u] = 0;
            if (g + u < cnt && !handed[u]) cf[u] = pair_coef_q<OPT, LPI, NB, false>(xi, xj[u], a.lr, c0, table);
        }
        // the interactions in list order; a row that has not been announced yet is awaited just before its own
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (anyh && handed[u]) {
                if (fl[u] != a.seq) bad = wait_row_at(a, inr[u] ? a.rowflag : prev_flags, j0[u]) || bad;  // its flag, then agent-scope loads
                asm volatile("" ::: "memory");
                const uint32_t off = hs.at(j0[u], inr[u]) + 16u * t;
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if (!(FULL || 4u * LPI * b + 4u * t < D)) xj[u][b] = make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef F2V_TEST_HOOKS
                    else if (a.test_nowait & 4u) xj[u][b] = *reinterpret_cast<const float4 *>((inr[u] ? a.Xn : a.X) + (size_t)j0[u] * D + t * 4 + 4 * LPI * b);
#endif
                    else xj[u][b] = load16_agent(hs.r, off + 16u * LPI * b);
                }
                cf[u] = pair_coef_q<OPT, LPI, NB, false>(xi, xj[u], a.lr, c0, table);
            }
            if (g + u < cnt) pair_apply_q<OPT, LPI, NB, false>(xi, xj[u], Y, a.lr, cf[u]);
        }
        g += U;
    } while (g < maxcnt);
}

// the largest value any item of the wave holds (the value is uniform inside an item): a scalar
template <int LPI>
__device__ __forceinline__ uint32_t wave_max_of_items(uint32_t v) {
    uint32_t m = 0;
#pragma unroll
    for (int l = 0; l < 64; l += LPI) {
        const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)v, l);
        m = x > m ? x : m;
    }
    return m;
}

// The body of one workgroup: `blk` is its index in the minibatch's own grid (step items first, then tree nodes).
// CHAIN: the minibatch is one of several in this launch (qstep_chain_kernel): finished rows are written through at agent
// scope, because a later minibatch of the same launch may read them on another XCD.
template <int OPT, int LPI, int NB, int U, bool PUSH, bool FULL, bool CHAIN>
__device__ __forceinline__ void qstep_body(const StepArgs &a, const uint32_t blk) {
    constexpr uint32_t DP = 4u * LPI * NB, IPW = 64u / LPI;  // padded dims (the tree's width); items per wavefront
    const uint32_t D = FULL ? DP : a.D;                       // live dims = row stride
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t t = lane & (LPI - 1u), q = lane / LPI;
    const uint32_t wpb = blockDim.x >> 6;
    if (blk >= a.step_blocks) {
        // the tail of the grid: one wavefront per node of the combine trees of this launch's hub rows; every node
        // waits for the partial sums it adds (hub pieces below announce theirs through the same flags)
        constexpr int FVEC = DP >= 64u ? (int)(DP / 64u) : 1;
        FinalizeTreeArgs ft;
        ft.f.X = a.X; ft.f.partials = a.partials; ft.f.Xn = a.Xn; ft.f.items = a.fin_items; ft.f.n_items = a.fin_n; ft.f.D = D;
        ft.f.push = a.push;
        ft.ready = a.ready; ft.err = a.err; ft.timeout_ticks = a.timeout_ticks; ft.seq = a.seq; ft.first_dep = 0u; ft.rowflag = a.rowflag;
#ifdef F2V_TEST_HOOKS
        ft.stamps = a.stamps;
        ft.test_withhold_row = a.test_withhold_row;
#endif
        const uint32_t node = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blk - a.step_blocks) * wpb + (threadIdx.x >> 6)));
        finalize_tree_node<OPT, FVEC, (FULL && DP % 64u == 0u), CHAIN>(ft, node, lane);
        return;
    }
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blk * wpb + (threadIdx.x >> 6)));

    // The minibatch's negative samples are the same ns rows for every item (except with -bs 1): the workgroup
    // stages them in LDS once -- each CU then fetches them from L2 once per workgroup instead of once per item, and
    // the repulsive interactions read them at LDS latency.  All lane groups of a wave read the same 16-byte
    // slots (broadcast), consecutive lanes consecutive slots: conflict-free.
    constexpr uint32_t kLdsSamples = 8;
    __shared__ float4 smp[kLdsSamples][DP / 4];
    const bool lds_samples = !a.bs_mode && a.ns <= kLdsSamples;
    const HandSrc hs = hand_src_window(a, D);  // (chained launches: where handed-on rows are read, 16 bytes at agent scope)
    // options 6/7: the 8-KiB sigmoid table is looked up once per interaction, in the middle of the dependent chain
    // dot product -> sigma -> update; from LDS that lookup costs ~64 cycles instead of an L1/L2 round trip
    __shared__ float sm_lds[OPT == 5 ? 1 : 2048];
    if constexpr (OPT != 5) {
        for (uint32_t k = threadIdx.x; k < 2048u; k += blockDim.x) sm_lds[k] = a.sm_table[k];
    }
    bool smp_bad = false;
    if (lds_samples) {
        for (uint32_t k = threadIdx.x; k < a.ns * (DP / 4); k += blockDim.x) {
            const uint32_t sidx = k / (DP / 4), c4 = k % (DP / 4);
            bool handed = false;
            if constexpr (CHAIN) {
                const uint32_t sj = a.sample_ids[sidx];
                handed = (sj - a.chain_lo) < a.chain_rows;
                if (handed) smp_bad = wait_row(a, sj) || smp_bad;
            }
            const float *srow = row_src(a, a.sample_ids[sidx], D);
            smp[sidx][c4] = !(FULL || 4u * c4 < D) ? make_float4(0.f, 0.f, 0.f, 0.f)
                            : (CHAIN && handed)    ? load16_agent(hs.r, hs.at(a.sample_ids[sidx], true) + 16u * c4)
                                                   : reinterpret_cast<const float4 *>(srow)[c4];
        }
    }
    if constexpr (CHAIN) {
        // a sample row that never arrived poisons the whole workgroup: nobody stores
        if (__syncthreads_or(smp_bad ? 1 : 0)) return;
    } else {
        if (lds_samples || OPT != 5) __syncthreads();
    }
    const float *table = OPT == 5 ? a.sm_table : sm_lds;
    if (IPW * w >= a.n_items) return;

    // this item's lanes (lane groups past the end of the list idle with cnt = 0)
    const uint32_t idx = IPW * w + q;
    const bool active = idx < a.n_items;
    Item it;
    if (active) it = a.items[idx];
    else { it.row = 0; it.nb = 0; it.cnt = 0; it.flags = 0; }
    const uint32_t row = it.row;
    const bool partial = (it.flags & kItemPartial) != 0;
    const bool first_chunk = (it.flags & kItemFirst) != 0;
    const bool last_chunk = (it.flags & kItemLast) != 0;

    float xi[NB][4], Y[NB][4];
    {
        const float *src = a.X + (size_t)row * D + t * 4;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float4 v = (FULL || 4u * LPI * b + 4u * t < D) ? *reinterpret_cast<const float4 *>(src + 4 * LPI * b) : make_float4(0.f, 0.f, 0.f, 0.f);
            xi[b][0] = v.x; xi[b][1] = v.y; xi[b][2] = v.z; xi[b][3] = v.w;
        }
    }
    double c0 = 0.0;
    if constexpr (OPT == 5) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) Y[b][v] = 0.0f;
    } else {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int v = 0; v < 4; ++v) Y[b][v] = first_chunk ? xi[b][v] : 0.0f;
        const uint32_t gdeg = a.rowptr[row + 1] - a.rowptr[row];
        const float degi = a.unit_degi ? 1.0f : (float)(1.0 / (double)(gdeg + 1u));  // algorithms.cpp:854
        c0 = (double)(a.lr * degi);
    }

    bool bad = false;  // chained minibatches: a wait for an earlier minibatch's row gave up -- this item stores nothing
    qprocess<OPT, LPI, NB, false, U, FULL, CHAIN>(a, hs, a.nbr_ids + it.nb, it.cnt, wave_max_of_items<LPI>(it.cnt), t, D, xi, Y, c0, table, bad);
    if (lds_samples) {
        if (active && last_chunk) {
            for (uint32_t sidx = 0; sidx < a.ns; ++sidx) {
                float4 xs[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b) xs[b] = smp[sidx][LPI * b + t];
                pair_update_q<OPT, LPI, NB, true>(xi, xs, Y, a.lr, c0, table);
            }
        }
    } else {
        const uint32_t scnt = (active && last_chunk) ? a.ns : 0u;
        const uint32_t sbase = a.bs_mode ? (row - a.batch_lo) : 0u;
        qprocess<OPT, LPI, NB, true, U, FULL, CHAIN>(a, hs, a.sample_ids + sbase, scnt, wave_max_of_items<LPI>(scnt), t, D, xi, Y, c0, table, bad);
    }

    if (CHAIN && __builtin_amdgcn_ballot_w64(bad) != 0ull) {
        // (wave-uniform on purpose: the launch is lost anyway, and a wave either announces all its items or none)
        return;
    }
    if (active) {
        float *out = (partial ? a.partials + (size_t)(it.flags & kItemSlotMask) * D : a.Xn + (size_t)row * D) + t * 4;
        float4 v[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (OPT == 5 && !partial)
                v[b] = make_float4(xi[b][0] + Y[b][0], xi[b][1] + Y[b][1], xi[b][2] + Y[b][2], xi[b][3] + Y[b][3]);  // algorithms.cpp:636
            else
                v[b] = make_float4(Y[b][0], Y[b][1], Y[b][2], Y[b][3]);
            if (!FULL && !(4u * LPI * b + 4u * t < D)) continue;
            if ((partial && a.fin_items) || CHAIN) store16_agent(out + 4 * LPI * b, v[b]);  // a tree node (or a later minibatch) of this grid reads it
            else *reinterpret_cast<float4 *>(out + 4 * LPI * b) = v[b];
        }
        if constexpr (PUSH) {
            // sharded run: the finished row goes, from registers, to the peers that read it (each item of the wave
            // has its own row and reader mask: the loop over peers is predicated per lane group)
            if (!partial) {
                const uint32_t others = ((1u << a.push.world) - 1u) & ~(1u << a.push.self);
                const uint32_t m = (a.push.masks ? a.push.masks[row] : others) & others;
#pragma unroll
                for (int q = 0; q < kMaxRanks; ++q) {
                    if (m & (1u << q)) {
                        float *dst = a.push.peer[q] + (size_t)(row - a.push.row_base) * D + t * 4;
#pragma unroll
                        for (int b = 0; b < NB; ++b)
                            if (FULL || 4u * LPI * b + 4u * t < D) store16_system(dst + 4 * LPI * b, v[b]);
                    }
                }
            }
        }
    }
    if constexpr (PUSH) __builtin_amdgcn_s_waitcnt(0);  // the peers' memory has acknowledged this wave's rows
    if constexpr (CHAIN) {
        __builtin_amdgcn_s_waitcnt(0);  // new rows are in memory before they are announced to the later minibatches of the launch
#ifdef F2V_TEST_HOOKS
        if (active && !partial && t == 0u && row != a.test_withhold_row)
#else
        if (active && !partial && t == 0u)
#endif
            __hip_atomic_store(a.rowflag + row, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef F2V_TEST_HOOKS
        if (a.stamps && active && !partial && t == 0u) a.stamps[4 * (size_t)row + 2] = wall_clock64();
#endif
    }
    if (a.fin_items) {
        __builtin_amdgcn_s_waitcnt(0);  // partial sums are in memory before they are announced
#ifdef F2V_TEST_HOOKS
        if ((it.flags & kItemSlotMask) == a.test_withhold_slot) return;  // fault injection: this piece never announces its sum
#endif
        if (active && partial && t == 0u) __hip_atomic_store(a.ready + (it.flags & kItemSlotMask), a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef F2V_TEST_HOOKS
        if (a.stamps && active && partial && it.cnt != 0u && t == 0u) atomicMax(a.stamps + 4 * (size_t)row, wall_clock64());
#endif
    }
}

template <int OPT, int LPI, int NB, int U, bool PUSH = false, bool FULL = true>
__global__ __launch_bounds__(256) void qstep_kernel(const StepArgs a) {
#ifdef F2V_TEST_HOOKS
    const unsigned long long t0 = a.xcd_times ? wall_clock64() : 0ull;
#endif
    qstep_body<OPT, LPI, NB, U, PUSH, FULL, false>(a, blockIdx.x);
#ifdef F2V_TEST_HOOKS
    if (a.xcd_times && threadIdx.x == 0u) {
        uint32_t id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(id));
        const unsigned long long t1 = wall_clock64();
        atomicMax(a.xcd_times + (id & 7u), t1);
        atomicMin(a.xcd_times + 8u + (id & 7u), t0);
        atomicAdd(a.xcd_times + 16u + (id & 7u), t1 - t0);
        atomicAdd(a.xcd_times + 24u + (id & 7u), 1ull);
    }
#endif
}

// ---- chained minibatches: several consecutive minibatches in ONE launch ---------------------------------------------
// A small minibatch (the reference's default is 384 rows) is a few microseconds of work, but as a dependent launch of its
// own it costs ~6 us of launch boundary on top: at batch 256 the engine was bound by that chain.  Here one launch covers many
// consecutive minibatches (workgroups batch-major, in index order) and the Gauss-Seidel order between them is kept by DATA
// dependencies at ROW granularity instead of launch boundaries: an item reads neighbour (or negative-sample) row j
//   * from the first matrix if j has not been updated this epoch (as always),
//   * from the second matrix if an earlier LAUNCH updated it (as always),
//   * and if an earlier minibatch of THIS launch updates it -- j in [chain_lo, its own minibatch's first row) -- from the
//     second matrix AFTER waiting for rowflag[j] == seq, which j's writer (its whole-row item, or the root of its combine
//     tree) stores once the written-through row (sc1) has been acknowledged.
// On RMAT-20 at batch 256 about 1.6 % of the neighbours are such rows; everything else runs ahead of the chain, and the
// chain itself is the row-level dependency chain (hubs read hubs: ~24 deep in 64 minibatches), not one hop per minibatch.
//   no stale lines   a row of the second matrix is never read before its writer has announced it (it is outside the
//                    updated range until then), so no cache of the reading XCD can hold an older copy of its lines; rows are
//                    whole 128-byte lines (D a multiple of 32 -- other D keep one launch per minibatch); plain loads follow
//                    the (L1-bypassing) poll
//   progress         an item only ever waits for rows of workgroups with a smaller index, and every XCD starts its
//                    workgroups in index order (checked by f2v_create's dispatch probe): the lowest unfinished one can
//                    always run; all waits are bounded by "tree_timeout_ms" all the same (err = 3: nothing is stored from
//                    there on, the launch drains, f2v_train fails within an epoch)
struct WgDesc {            // one per workgroup, 32 bytes: everything it needs to know about its minibatch in ONE load
    uint32_t lo;           // first row of the minibatch
    uint32_t item_off;     // its items (offset into the launch's item array), n_items of them
    uint32_t n_items;
    uint32_t step_blocks;  // workgroups of the minibatch that step items; the rest run tree nodes
    uint32_t fin_off, fin_n;
    uint32_t index;        // global minibatch index of the epoch: which sample ids
    uint32_t blk;          // this workgroup's index inside the minibatch's grid (step items first, then tree nodes)
};
struct ChainArgs {
    StepArgs base;            // what all minibatches share; items / fin_items point at the launch's arrays
    const WgDesc *wg;
    const uint32_t *ids;      // the epoch's sample ids, `ids_stride` per minibatch
    uint32_t ids_stride;
};

// (5 waves per SIMD like the plain kernel where the registers allow it: without the bound the D = 128 instance takes 98 VGPRs, two too many)
template <int OPT, int LPI, int NB, int U, bool FULL>
__global__ __launch_bounds__(256, (NB <= 2 && U <= 4) ? 5 : 1) void qstep_chain_kernel(const ChainArgs c) {
    const WgDesc bd = c.wg[blockIdx.x];
    StepArgs a = c.base;
    a.batch_lo = bd.lo;
    a.upd_rows = bd.lo - a.upd_lo;      // rows [upd_lo, this minibatch's first row) are read from the second matrix ...
    a.chain_rows = bd.lo - a.chain_lo;  // ... those from chain_lo on after waiting for their flag
#ifdef F2V_TEST_HOOKS
    if (a.chain_lo == 0xFFFFFFFFu) a.chain_rows = 0u;  // f2v_test_chain_nowait
#endif
    a.sample_ids = c.ids + (size_t)bd.index * c.ids_stride;
    a.items = c.base.items + bd.item_off;
    a.n_items = bd.n_items;
    a.step_blocks = bd.step_blocks;
    a.fin_items = bd.fin_n ? c.base.fin_items + bd.fin_off : nullptr;
    a.fin_n = bd.fin_n;
    qstep_body<OPT, LPI, NB, U, false, FULL, true>(a, bd.blk);
}

// ---- chained minibatches, wide form: a split row's pieces meet in LDS ------------------------------------------------
// qstep_chain_kernel above hands a split row's partial sums from workgroup to workgroup through HBM: piece -> ready flag ->
// tree node -> ready flag -> (tree node ->) row flag -- three or four global hand-offs (written-through store, its
// acknowledgement, a flag store, an L2-missing poll, agent-scope loads) on every hop of the row-to-row dependency chain that
// bounds small minibatches (profiles/r03_chain_hops_baseline.txt).  Here the pieces of a row meet INSIDE one workgroup:
//   * a workgroup runs a PROGRAM of rounds (one piece, or one whole low-degree row, per lane group and round); a piece's
//     force sum goes to an LDS slot instead of HBM; inside a phase the pieces run in the order of what they wait for (the
//     ones that read the most recently written rows in the last round), whatever their slots;
//   * when a PHASE of rounds ends, JOBS add slots in order, 32 lanes per job, 8 jobs side by side: the same chunk order and
//     the same fan-in groups as the combine tree (piece sums of one fan-in group are added first to last, then the groups'
//     sums first to last), so every bit is the one the tree -- and the oracle's restatement of it -- produces;
//   * a row of up to `fanin` pieces is finished by one job (x_i added, ONE written-through row, ONE row flag); several such
//     rows share a workgroup;
//   * a row (or, beyond fanin^2 pieces, each fanin^2-piece unit of it) of several fan-in groups has a FINISHER workgroup that
//     owns the groups whose neighbours were updated most recently -- the ones that will wait longest -- runs them last, and
//     adds all groups' sums itself; the other groups are computed ahead of time by HELPER workgroups (smaller index) whose
//     group sums travel through HBM (written through + ready flag) and are imported into the finisher's LDS before its last
//     phase.  After the last awaited neighbour row has arrived, what is left is one gather, one in-LDS group sum, one in-LDS
//     sum of the groups' sums and the row store: one global hand-off per hop instead of three or four;
//   * rows of more than fanin^2 pieces keep combine-tree nodes for the levels above their units' sums (finalize_tree_node).
// Waits only ever point at smaller workgroup indices (rows of earlier minibatches; a unit's helpers), as before.
struct WJob {            // 16 bytes
    uint8_t src, n;      // LDS slots [src, src + n) are added in order (kJobImport: unused)
    uint8_t kind;        // kJob*
    uint8_t phase;       // the job runs when this phase of the workgroup's program has ended (kJobBefore: before the first round)
    uint8_t pass_len;    // jobs [this, this + pass_len) run side by side (<= 8): set on every job of the pass
    uint8_t src2, n2;    // n2 != 0: first LDS slots [src2, src2 + n2) are added in order into LDS slot `dst2` (one of [src, src + n)):
    uint8_t dst2;        //          the last fan-in group's sum and the sum of the groups' sums in ONE job of one 32-lane team
    uint32_t dst;        // kJobLds, kJobImport: LDS slot; kJobPart: partial-sum slot in HBM
    uint32_t row;        // kJobRow / kJobPart: the row; kJobImport: the partial-sum slot in HBM that is fetched
};
static_assert(sizeof(WJob) == 16, "job descriptor layout");
constexpr uint8_t kJobLds = 0, kJobPart = 1, kJobRow = 2, kJobImport = 3;
constexpr uint8_t kJobBefore = 255;
constexpr uint32_t kItemDirect = 1u << 28;    // wide programs: a whole row -- its new embedding is stored by the item itself
constexpr uint32_t kItemPhaseEnd = 1u << 27;  // ... the round this item belongs to is the last of its phase (on every item of the round)
constexpr uint32_t kItemIdle = 1u << 26;      // ... filler: nothing to compute, nothing to store
constexpr uint32_t kItemPieceSlot = 0xFFu;    // ... a piece's LDS slot (the order in which a job adds): pieces are RUN in the order of what they wait for
constexpr uint32_t kWideSumSlots = 32;        // LDS slots for fan-in group sums, behind the piece slots

struct WideDesc {        // one per workgroup, 32 bytes
    uint32_t lo;         // first row of the minibatch
    uint32_t index;      // global minibatch index of the epoch: which sample ids
    uint32_t kind;       // 0: program of rounds and jobs; 1: combine-tree nodes
    uint32_t a, b, c, d; // program: first item, rounds, first job, jobs; nodes: first node of the minibatch, its nodes, this workgroup's index among their workgroups
    uint32_t pad;
};
struct WideArgs {
    StepArgs base;            // what all minibatches share; items / fin_items point at the launch's arrays
    const WideDesc *wg;
    const WJob *jobs;
    const uint32_t *ids;      // the epoch's sample ids, `ids_stride` per minibatch
    uint32_t ids_stride;    // Several EPOCHS in one launch ("wide_epochs"; wgs_per_epoch = 0: one epoch, the fields below unused): workgroup b belongs to
    // epoch e = b / wgs_per_epoch of the launch and runs the program of workgroup b % wgs_per_epoch.  base.X is a ring of
    // matrices `ring_stride` floats apart: epoch e reads matrix e and writes matrix e + 1 (no buffer is written twice in a launch,
    // so nothing a slower workgroup still reads is ever overwritten); every epoch has its own row flags (n_rows apart), partial-sum
    // slots and their flags (slots_per_epoch apart) and sample ids (ids_epoch_stride apart).  A row of the first matrix is, from
    // the second epoch on, a row the launch's previous epoch wrote: awaited in that epoch's flags, read at agent scope.
    uint32_t wgs_per_epoch, n_rows, slots_per_epoch;
    uint64_t ring_stride, ids_epoch_stride;
};

// a workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global load and store of the wave
// (the next round's prefetched item and ids, written-through rows on their way) -- none of which the other waves need
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr uint32_t kWideJobsLds = 64;  // a workgroup's first jobs are staged in LDS when it starts (more stay in global memory)

// EARLY ("wide_samples_early"): the sample rows this launch writes itself are awaited BEFORE the first round's neighbour waits and the
// samples' scalars are computed ahead of them; otherwise both stay behind the neighbours.  (Two kernels, not a run-time switch: with
// both forms inlined the kernel grew enough to run 7 % slower on RMAT-20.)
// MODE 2: EARLY, and the launch may chain several epochs (WideArgs::wgs_per_epoch; its own kernel for the same reason -- the plain
// form must not carry that code -- and because it needs ~170 registers: 2 waves per SIMD, which costs nothing where a launch is a chain).
template <int OPT, int LPI, int NB, int U, bool FULL, int MODE>
#ifndef F2V_WIDE_WAVES
#define F2V_WIDE_WAVES 3
#endif
__global__ __launch_bounds__(256, (NB <= 2 && U <= 4) ? (MODE == 2 ? 2 : OPT == 5 ? F2V_WIDE_WAVES : 3) : 1) void qwide_chain_kernel(const WideArgs w) {
    constexpr bool EARLY = MODE >= 1, EPOCHS = MODE == 2;
    constexpr uint32_t DP = 4u * LPI * NB, IPW = 64u / LPI, IPB = 4u * IPW;  // padded dims; lane groups per wavefront / workgroup
    constexpr uint32_t PSLOTS = IPB > 32u ? IPB : 32u;                       // piece slots: one phase of rounds
    constexpr uint32_t C4 = DP / 4u;                                          // 16-byte pieces per (padded) row
    uint32_t ep = 0u, wgi = blockIdx.x;  // (uniform)
    if (EPOCHS && w.wgs_per_epoch != 0u) {
        ep = blockIdx.x / w.wgs_per_epoch;
        wgi = blockIdx.x - ep * w.wgs_per_epoch;
    }
    const WideDesc bd = w.wg[wgi];
    StepArgs a = w.base;
    const uint32_t *prev_flags = nullptr;  // the previous epoch's row flags (second epoch of a launch onwards; always null unless EPOCHS)
    if (EPOCHS && w.wgs_per_epoch != 0u) {
        a.X = w.base.X + (size_t)ep * w.ring_stride;
        a.Xn = const_cast<float *>(a.X) + w.ring_stride;
        a.rowflag = w.base.rowflag + (size_t)ep * w.n_rows;
        a.partials = w.base.partials + (size_t)ep * w.slots_per_epoch * w.base.D;
        a.ready = w.base.ready + (size_t)ep * w.slots_per_epoch;
        if (ep != 0u) prev_flags = a.rowflag - w.n_rows;
    }
    a.batch_lo = bd.lo;
    a.upd_rows = bd.lo - a.upd_lo;      // rows [upd_lo, this minibatch's first row) are read from the second matrix ...
    a.chain_rows = bd.lo - a.chain_lo;  // ... those from chain_lo on after waiting for their flag
#ifdef F2V_TEST_HOOKS
    if (a.test_nowait & 1u) a.chain_rows = 0u;  // f2v_test_chain_nowait (further bits: timing experiments, results WRONG)
#endif
    a.sample_ids = w.ids + (EPOCHS ? (size_t)ep * w.ids_epoch_stride : (size_t)0) + (size_t)bd.index * w.ids_stride;
    const uint32_t D = FULL ? DP : a.D;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t t = lane & (LPI - 1u), q = lane / LPI;
    // where handed-on rows are read (16 bytes at agent scope): the launch's window of the second matrix, or -- epochs chained
    // in one launch -- this epoch's two matrices of the ring, which lie ring_stride floats apart
    HandSrc hs = hand_src_window(a, D);
    if (EPOCHS && w.wgs_per_epoch != 0u) {
        hs.r = hand_rsrc(a.X);
        hs.x_off = 0u;
        hs.xn_off = (uint32_t)(w.ring_stride * 4u);
    }
    const rsrc_t part_rsrc = hand_rsrc(a.partials);  // helpers' group sums (partial-sum slot s at byte s * 4D)
    if (bd.kind == 1u) {
        constexpr int FVEC = DP >= 64u ? (int)(DP / 64u) : 1;
        FinalizeTreeArgs ft;
        ft.f.X = a.X; ft.f.partials = a.partials; ft.f.Xn = a.Xn; ft.f.items = a.fin_items + bd.a; ft.f.n_items = bd.b; ft.f.D = D;
        ft.f.push = a.push;
        ft.ready = a.ready; ft.err = a.err; ft.timeout_ticks = a.timeout_ticks; ft.seq = a.seq; ft.first_dep = 0u; ft.rowflag = a.rowflag;
#ifdef F2V_TEST_HOOKS
        ft.stamps = a.stamps;
        ft.test_withhold_row = a.test_withhold_row;
#endif
        const uint32_t node = (uint32_t)__builtin_amdgcn_readfirstlane((int)(bd.c * 4u + wave));
        finalize_tree_node<OPT, FVEC, (FULL && DP % 64u == 0u), true>(ft, node, lane);
        return;
    }

    // requested first, side by side with the sample ids below: the first round's item (its neighbour ids are requested as soon as
    // it is here, behind the sample rows) and this thread's share of the job descriptors -- the workgroup's first gathers are
    // three dependent loads away from its start, not five
    const Item *items = a.items + bd.a;
    const WJob *gjobs = w.jobs + bd.c;
    const uint32_t n_jobs = bd.d;
    Item it = items[wave * IPW + q];
    uint4 job_mine = make_uint4(0u, 0u, 0u, 0u);
    if (threadIdx.x < n_jobs && threadIdx.x < kWideJobsLds) job_mine = *reinterpret_cast<const uint4 *>(&gjobs[threadIdx.x]);

    __shared__ float4 slots[PSLOTS + kWideSumSlots][C4];
    constexpr uint32_t kLdsSamples = 8;
    __shared__ float4 smp[kLdsSamples][C4];
    __shared__ float sm_lds[OPT == 5 ? 1 : 2048];
    __shared__ uint32_t wg_bad;  // a bounded wait of this workgroup gave up: nothing more is stored or announced
    const bool lds_samples = !a.bs_mode && a.ns <= kLdsSamples;
    if (threadIdx.x == 0) wg_bad = 0u;
    if constexpr (OPT != 5) {
        for (uint32_t k = threadIdx.x; k < 2048u; k += 256u) sm_lds[k] = a.sm_table[k];
    }
    // The minibatch's negative-sample rows, staged in LDS once per workgroup.  Those that an earlier minibatch of this launch
    // writes (bit k of `late`) are NOT waited for here: the first round's neighbour gathers go out first, and the wait sits
    // just in front of the first use of the samples (a sample row of the previous minibatch would otherwise hold the whole
    // workgroup before it has requested anything).
    uint32_t late = 0;  // (uniform)
    if (lds_samples) {
        for (uint32_t sidx = 0; sidx < a.ns; ++sidx)
            if ((a.sample_ids[sidx] - a.chain_lo) < a.chain_rows || prev_flags != nullptr) late |= 1u << sidx;
        for (uint32_t k = threadIdx.x; k < a.ns * C4; k += 256u) {
            const uint32_t sidx = k / C4, c4 = k % C4;
            if ((late >> sidx) & 1u) continue;
            smp[sidx][c4] = (FULL || 4u * c4 < D) ? reinterpret_cast<const float4 *>(row_src(a, a.sample_ids[sidx], D))[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    auto stage_late_samples = [&]() {  // all threads; -> true: a wait gave up
        bool bad = false;
        for (uint32_t k = threadIdx.x; k < a.ns * C4; k += 256u) {
            const uint32_t sidx = k / C4, c4 = k % C4;
            if (!((late >> sidx) & 1u)) continue;
            const uint32_t sj = a.sample_ids[sidx];
            if (!(FULL || 4u * c4 < D)) {
                smp[sidx][c4] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                bad = wait_row_at(a, (sj - a.chain_lo) < a.chain_rows ? a.rowflag : prev_flags, sj) || bad;
                smp[sidx][c4] = load16_agent(hs.r, hs.at(sj, (sj - a.chain_lo) < a.chain_rows) + 16u * c4);
            }
        }
        return bad;
    };
    const float *table = OPT == 5 ? a.sm_table : sm_lds;

    uint32_t jc = 0;  // next job (uniform)
    uint32_t jpre[U];  // the first ids of the current round's list, requested a round ago
    {
        const uint32_t cnt0 = (it.flags & kItemIdle) ? 0u : it.cnt;
#pragma unroll
        for (int u = 0; u < U; ++u) jpre[u] = ((uint32_t)u < cnt0) ? a.nbr_ids[it.nb + u] : 0u;
    }
    // (the jobs run at the end of the dependency chain's hops: their descriptors wait in LDS, not behind two more global loads)
    __shared__ __attribute__((aligned(16))) WJob ljobs[kWideJobsLds];
    if (threadIdx.x < n_jobs && threadIdx.x < kWideJobsLds) *reinterpret_cast<uint4 *>(&ljobs[threadIdx.x]) = job_mine;
    auto job_at = [&](uint32_t k) -> WJob {  // (decoded from one 16-byte load: a struct picked by a condition ends up in scratch)
        const uint4 raw = k < kWideJobsLds ? *reinterpret_cast<const uint4 *>(&ljobs[k]) : *reinterpret_cast<const uint4 *>(&gjobs[k]);
        WJob j;
        j.src = (uint8_t)raw.x; j.n = (uint8_t)(raw.x >> 8); j.kind = (uint8_t)(raw.x >> 16); j.phase = (uint8_t)(raw.x >> 24);
        j.pass_len = (uint8_t)raw.y; j.src2 = (uint8_t)(raw.y >> 8); j.n2 = (uint8_t)(raw.y >> 16); j.dst2 = (uint8_t)(raw.y >> 24);
        j.dst = raw.z; j.row = raw.w;
        return j;
    };
    __syncthreads();
    // the workgroup's last job, when it finishes a row alone in its pass (a finisher's): x_i is requested now, not at the end of the hop
    float xi_pre = 0.f;
    uint32_t xi_pre_for = 0xFFFFFFFFu;
    if (OPT == 5 && n_jobs != 0u) {
        const WJob jl = job_at(n_jobs - 1u);
        if (jl.kind == kJobRow && jl.pass_len == 1u) {
            xi_pre_for = n_jobs - 1u;
            if (prev_flags != nullptr) {  // (the row's own last value is the previous epoch's: awaited, agent scope)
                if (wait_row_at(a, prev_flags, jl.row)) wg_bad = 1u;
                lds_barrier();
                if (wg_bad) return;  // (uniform: read behind the barrier)
                if (threadIdx.x < D)
                    xi_pre = __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const uint32_t *>(a.X + (size_t)jl.row * D + threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            } else if (threadIdx.x < D) xi_pre = a.X[(size_t)jl.row * D + threadIdx.x];
        }
    }

    // the jobs of one phase: passes of up to 8 jobs, 32 lanes each; lane tl of a job owns the 16-byte pieces tl, tl + 32, ... of the row
    auto run_jobs = [&](const uint32_t tag) {
        const uint32_t team = threadIdx.x >> 5, tl = threadIdx.x & 31u;
        while (jc < n_jobs) {
            const WJob head = job_at(jc);
            if (head.phase != tag) break;
            const uint32_t len = head.pass_len;
            if (len == 1u && head.kind != kJobImport) {
                // A job alone in its pass (a whole fan-in group's sum, a unit's sum: the long ones, at the end of a hop of the
                // dependency chain) runs on ALL threads, one dim each: every slot read is issued before the first addition, the
                // additions are one dependent chain of n instead of four interleaved ones (MI355X: ~0.5 us -> ~0.2 us for 32 slots).
                const WJob jb = head;
                const uint32_t d = threadIdx.x;
                float *S = reinterpret_cast<float *>(&slots[0][0]);
                // slots [first, first + n) of this thread's dim, added in order: 8 reads at a time off one base address (n is uniform:
                // the guards are scalar branches), so that the job costs a dozen registers, not 64
                auto sum32 = [&](uint32_t first, uint32_t n) -> float {
                    const float *base = S + (size_t)first * DP + d;
                    float acc = base[0];
                    for (uint32_t k0 = 1; k0 < n; k0 += 8u) {
                        float v[8];
#pragma unroll
                        for (uint32_t u = 0; u < 8u; ++u)
                            if (k0 + u < n) v[u] = base[(size_t)(k0 + u) * DP];
#pragma unroll
                        for (uint32_t u = 0; u < 8u; ++u)
                            if (k0 + u < n) acc = acc + v[u];
                    }
                    return acc;
                };
                if (d < D) {  // (D is a multiple of 4 wherever the sub-wave layouts run -- subwave_width() -- so the quads that gather16<1> regroups below are live or idle as a whole)
                    float xi1 = 0.f;
                    if (OPT == 5 && jb.kind == kJobRow) {
                        // (a launch that chains epochs: the row's pieces in this workgroup have waited for its previous value already)
                        if (xi_pre_for == jc) xi1 = xi_pre;
                        else if (prev_flags != nullptr) xi1 = __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const uint32_t *>(a.X + (size_t)jb.row * D + d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        else xi1 = a.X[(size_t)jb.row * D + d];
                    }
#ifdef F2V_TEST_HOOKS
                    const bool skip_sums = (a.test_nowait & 8u) != 0u;
#else
                    constexpr bool skip_sums = false;
#endif
                    if (jb.n2 != 0u && !skip_sums) S[(size_t)jb.dst2 * DP + d] = sum32(jb.src2, jb.n2);  // (read back by this very lane below)
                    float acc = sum32(jb.src, skip_sums ? 1u : jb.n);
                    if (jb.kind == kJobLds) {
                        S[(size_t)jb.dst * DP + d] = acc;
                    } else {
                        if (OPT == 5 && jb.kind == kJobRow) acc = xi1 + acc;  // algorithms.cpp:636
                        const float one[1] = {acc};
                        float4 piece[1];
                        uint32_t fd;
                        const bool mine = gather16<1>(one, lane, piece, fd);  // quads -> 16-byte written-through stores
                        float *out = jb.kind == kJobRow ? a.Xn + (size_t)jb.row * D : a.partials + (size_t)jb.dst * D;
                        if (mine) store16_agent(out + d, piece[0]);
#ifdef F2V_TEST_HOOKS
                        if (!(a.test_nowait & 2u))
#endif
                        __builtin_amdgcn_s_waitcnt(0);  // this wavefront's written-through bytes have been acknowledged ...
                    }
                }
                jc += 1u;
                lds_barrier();  // ... by every wavefront that stored, before one lane announces them
                if (threadIdx.x == 0u && (jb.kind == kJobPart || jb.kind == kJobRow)) {
                    if (jb.kind == kJobRow) {
#ifdef F2V_TEST_HOOKS
                        if (jb.row != a.test_withhold_row)  // fault injection: this row is never announced
#endif
                        __hip_atomic_store(a.rowflag + jb.row, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
#ifdef F2V_TEST_HOOKS
                        if (a.stamps) atomicMax(a.stamps + 4 * (size_t)jb.row + 1, wall_clock64());
                        if (jb.dst != a.test_withhold_slot)  // fault injection: this sum is never announced
#endif
                        __hip_atomic_store(a.ready + jb.dst, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
#ifdef F2V_TEST_HOOKS
                if (jb.kind == kJobRow && a.stamps && threadIdx.x == 0u) a.stamps[4 * (size_t)jb.row + 2] = wall_clock64();
#endif
                continue;
            }
            if (team < len) {
                const WJob jb = job_at(jc + team);
                if (jb.kind == kJobImport) {
                    // a helper's group sum: wait for its flag (bounded), then agent-scope loads into the LDS slot
                    const unsigned long long t0 = wall_clock64();
                    bool gave_up = false;
                    uint32_t spins = 0;
                    while (__hip_atomic_load(a.ready + jb.row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.seq) {
                        __builtin_amdgcn_s_sleep(1);
                        if ((++spins & 15u) != 0u) continue;
                        if (__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { gave_up = true; break; }
                        if (wall_clock64() - t0 > a.timeout_ticks) {
                            if (__hip_atomic_fetch_add(a.err + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                                a.err[2] = blockIdx.x; a.err[3] = jb.row; a.err[4] = a.ready[jb.row]; a.err[5] = a.seq; a.err[6] = gridDim.x; a.err[7] = a.batch_lo;
                            }
                            __hip_atomic_store(a.err, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            gave_up = true;
                            break;
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the poll has returned before the sum's loads are issued
                    if (gave_up) {
                        wg_bad = 1u;
                    } else {
                        for (uint32_t c = tl; 4u * c < D; c += 32u) slots[jb.dst][c] = load16_agent(part_rsrc, jb.row * (D * 4u) + 16u * c);
                    }
                } else {
                    // slots [first, first + n) added in order, 8 LDS reads in flight
                    auto add_slots = [&](uint32_t first, uint32_t n, uint32_t c) -> float4 {
                        float4 acc = slots[first][c];
                        for (uint32_t k = 1; k < n; k += 8u) {
                            float4 p[8];
#pragma unroll
                            for (uint32_t u = 0; u < 8u; ++u) p[u] = slots[first + (k + u < n ? k + u : n - 1u)][c];
#pragma unroll
                            for (uint32_t u = 0; u < 8u; ++u) {
                                if (k + u < n) { acc.x = acc.x + p[u].x; acc.y = acc.y + p[u].y; acc.z = acc.z + p[u].z; acc.w = acc.w + p[u].w; }
                            }
                        }
                        return acc;
                    };
                    for (uint32_t c = tl; 4u * c < D; c += 32u) {
                        float4 xi4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (OPT == 5 && jb.kind == kJobRow) xi4 = prev_flags != nullptr ? load16_agent(hs.r, hs.at(jb.row, false) + 16u * c) : *reinterpret_cast<const float4 *>(a.X + (size_t)jb.row * D + 4u * c);
#ifdef F2V_TEST_HOOKS
                        const bool skip_sums = (a.test_nowait & 8u) != 0u;
#else
                        constexpr bool skip_sums = false;
#endif
                        if (jb.n2 != 0u && !skip_sums) slots[jb.dst2][c] = add_slots(jb.src2, jb.n2, c);  // (read back by this very lane below)
                        float4 acc = add_slots(jb.src, skip_sums ? 1u : jb.n, c);
                        if (jb.kind == kJobLds) {
                            slots[jb.dst][c] = acc;
                        } else if (jb.kind == kJobPart) {
                            store16_agent(a.partials + (size_t)jb.dst * D + 4u * c, acc);
                        } else {
                            if constexpr (OPT == 5) acc = make_float4(xi4.x + acc.x, xi4.y + acc.y, xi4.z + acc.z, xi4.w + acc.w);  // algorithms.cpp:636
                            store16_agent(a.Xn + (size_t)jb.row * D + 4u * c, acc);
                        }
                    }
                    if (jb.kind == kJobPart || jb.kind == kJobRow) {
#ifdef F2V_TEST_HOOKS
                        if (!(a.test_nowait & 2u))
#endif
                        __builtin_amdgcn_s_waitcnt(0);  // the written-through bytes have been acknowledged before they are announced
                        if (tl == 0u) {
                            if (jb.kind == kJobRow) {
#ifdef F2V_TEST_HOOKS
                                if (jb.row != a.test_withhold_row)  // fault injection: this row is never announced
#endif
                                __hip_atomic_store(a.rowflag + jb.row, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            } else {
#ifdef F2V_TEST_HOOKS
                                if (a.stamps) atomicMax(a.stamps + 4 * (size_t)jb.row + 1, wall_clock64());
                                if (jb.dst != a.test_withhold_slot)  // fault injection: this sum is never announced
#endif
                                __hip_atomic_store(a.ready + jb.dst, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                    }
#ifdef F2V_TEST_HOOKS
                    if (jb.kind == kJobRow && a.stamps && tl == 0u) a.stamps[4 * (size_t)jb.row + 2] = wall_clock64();
#endif
                }
            }
            jc += len;
            lds_barrier();  // the pass's sums are in LDS (or on their way to memory); its source slots may be reused
        }
    };

    if (n_jobs && job_at(0).phase == kJobBefore) {
        run_jobs(kJobBefore);
        if (wg_bad) return;
    }
    uint32_t phase = 0;
    for (uint32_t r = 0; r < bd.b; ++r) {
        Item itn;  // the next round's item: requested now, needed when this round's gathers are in flight
        if (r + 1 < bd.b) itn = items[(r + 1) * IPB + wave * IPW + q];
        else { itn.row = it.row; itn.nb = 0; itn.cnt = 0; itn.flags = kItemIdle; }
        const bool idle = (it.flags & kItemIdle) != 0;
        const bool direct = (it.flags & kItemDirect) != 0;
        const bool first_chunk = (it.flags & kItemFirst) != 0;
        const bool last_chunk = (it.flags & kItemLast) != 0;
        const uint32_t row = it.row;
        const uint32_t cnt = idle ? 0u : it.cnt;

        float xi[NB][4], Y[NB][4];
        bool xi_bad = false;
        {
            const float *src = a.X + (size_t)row * D + t * 4;
            if (prev_flags != nullptr && !idle) xi_bad = wait_row_at(a, prev_flags, row);  // (its own last value: the previous epoch's)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (FULL || 4u * LPI * b + 4u * t < D) v = prev_flags != nullptr ? load16_agent(hs.r, hs.at(row, false) + 16u * t + 16u * LPI * b) : *reinterpret_cast<const float4 *>(src + 4 * LPI * b);
                xi[b][0] = v.x; xi[b][1] = v.y; xi[b][2] = v.z; xi[b][3] = v.w;
            }
        }
        double c0 = 0.0;
        if constexpr (OPT == 5) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int v = 0; v < 4; ++v) Y[b][v] = 0.0f;
        } else {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int v = 0; v < 4; ++v) Y[b][v] = first_chunk ? xi[b][v] : 0.0f;
            const uint32_t gdeg = a.rowptr[row + 1] - a.rowptr[row];
            const float degi = a.unit_degi ? 1.0f : (float)(1.0 / (double)(gdeg + 1u));  // algorithms.cpp:854
            c0 = (double)(a.lr * degi);
        }
        bool bad = xi_bad, quit = false;
        float cs[EARLY ? kLdsSamples : 1];  // EARLY: the samples' scalars (pair_coef_q), computed while the neighbours are still on their way
        // (First round only) the sample rows this launch writes are awaited and staged.  EARLY -- where the minibatches of a launch
        // are one dependency chain (a small graph) -- this happens behind the round's gathers and IN FRONT of its neighbour waits: a
        // sample row and a neighbour row of the previous minibatch arrive at about the same time, and waiting for one after the other
        // made every hop one flag + one row round trip longer (cora, D = 128: -6 %); where most workgroups wait for nothing (RMAT-20)
        // the early barrier costs 2 %, and the staging stays behind the neighbours.
        auto stage_late = [&]() -> bool {
            if (late != 0u) {
                if (stage_late_samples()) wg_bad = 1u;
                late = 0u;
                lds_barrier();
                if (wg_bad) { quit = true; return true; }  // a sample row that never arrived: nobody stores
            }
            return false;
        };
        auto between = [&]() -> bool {
            if constexpr (EARLY) {
                if (stage_late()) return true;
                if (lds_samples && !idle && last_chunk) {
#pragma unroll
                    for (int sidx = 0; sidx < (int)kLdsSamples; ++sidx) {
                        cs[sidx] = 0.f;
                        if ((uint32_t)sidx < a.ns) {
                            float4 xs[NB];
#pragma unroll
                            for (int b = 0; b < NB; ++b) xs[b] = smp[sidx][LPI * b + t];
                            cs[sidx] = pair_coef_q<OPT, LPI, NB, true>(xi, xs, a.lr, c0, table);
                        }
                    }
                }
            }
            return false;
        };
        qprocess_pre<OPT, LPI, NB, U, FULL, MODE == 0>(a, hs, a.nbr_ids + it.nb, cnt, wave_max_of_items<LPI>(cnt), t, D, xi, Y, c0, table, bad, jpre,
                                            a.nbr_ids + itn.nb, (itn.flags & kItemIdle) ? 0u : itn.cnt, between, prev_flags);
        if constexpr (!EARLY) (void)stage_late();
        if (quit) return;
        if (lds_samples) {
            if (!idle && last_chunk) {
                if constexpr (EARLY) {
#pragma unroll
                    for (int sidx = 0; sidx < (int)kLdsSamples; ++sidx) {
                        if ((uint32_t)sidx < a.ns) {
                            float4 xs[NB];
#pragma unroll
                            for (int b = 0; b < NB; ++b) xs[b] = smp[sidx][LPI * b + t];
                            pair_apply_q<OPT, LPI, NB, true>(xi, xs, Y, a.lr, cs[sidx]);
                        }
                    }
                } else {
                    for (uint32_t sidx = 0; sidx < a.ns; ++sidx) {
                        float4 xs[NB];
#pragma unroll
                        for (int b = 0; b < NB; ++b) xs[b] = smp[sidx][LPI * b + t];
                        pair_update_q<OPT, LPI, NB, true>(xi, xs, Y, a.lr, c0, table);
                    }
                }
            }
        } else {
            const uint32_t scnt = (!idle && last_chunk) ? a.ns : 0u;
            const uint32_t sbase = a.bs_mode ? (row - a.batch_lo) : 0u;
            qprocess<OPT, LPI, NB, true, U, FULL, true>(a, hs, a.sample_ids + sbase, scnt, wave_max_of_items<LPI>(scnt), t, D, xi, Y, c0, table, bad);
        }
        const bool wave_bad = __builtin_amdgcn_ballot_w64(bad) != 0ull;  // (wave-uniform: a wave stores all its items or none)
        if (wave_bad) wg_bad = 1u;
        if (!idle && !wave_bad) {
            if (direct) {
                float *out = a.Xn + (size_t)row * D + t * 4;
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if (!FULL && !(4u * LPI * b + 4u * t < D)) continue;
                    const float4 v = OPT == 5 ? make_float4(xi[b][0] + Y[b][0], xi[b][1] + Y[b][1], xi[b][2] + Y[b][2], xi[b][3] + Y[b][3])  // algorithms.cpp:636
                                              : make_float4(Y[b][0], Y[b][1], Y[b][2], Y[b][3]);
                    store16_agent(out + 4 * LPI * b, v);
                }
            } else {
#pragma unroll
                for (int b = 0; b < NB; ++b) slots[it.flags & kItemPieceSlot][LPI * b + t] = make_float4(Y[b][0], Y[b][1], Y[b][2], Y[b][3]);
#ifdef F2V_TEST_HOOKS
                if (a.stamps && t == 0u) atomicMax(a.stamps + 4 * (size_t)row, wall_clock64());
#endif
            }
        }
        if (__builtin_amdgcn_ballot_w64(!idle && direct) != 0ull) {
            __builtin_amdgcn_s_waitcnt(0);  // the wave's new rows are in memory before they are announced
            if (!idle && direct && !wave_bad && t == 0u) {
#ifdef F2V_TEST_HOOKS
                if (row != a.test_withhold_row)
#endif
                __hip_atomic_store(a.rowflag + row, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef F2V_TEST_HOOKS
                if (a.stamps) a.stamps[4 * (size_t)row + 2] = wall_clock64();
#endif
            }
        }
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)(it.flags & kItemPhaseEnd)) != 0u) {
            if (jc < n_jobs && job_at(jc).phase == phase) {
                lds_barrier();  // the phase's piece sums are in LDS
                if (wg_bad) return;
                run_jobs(phase);
                if (wg_bad) return;
            }
            phase++;
        }
        it = itn;
    }
}

// One level of the hub combine trees of a launch: every item adds up to `fanin` partial rows in
// order (8 row loads in flight); the root of a row's tree stages the row's new embedding.
template <int OPT, int VEC, bool EXACT>
__global__ __launch_bounds__(256) void hub_finalize_kernel(const FinalizeArgs f) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (w >= f.n_items) return;
    const FinItem h = f.items[w];
    if (h.n == 0u) return;  // padding (the node layout of the one-launch variants)
    const uint32_t D = f.D;
    const float *p = f.partials + (size_t)h.in_slot * D;
    float Y[VEC];
    load_row<VEC, EXACT>(p, lane, D, Y);
    constexpr int U = 16;
    for (uint32_t c = 1; c < h.n; c += U) {
        float P[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t k = (c + u) < h.n ? (c + u) : (h.n - 1);
            load_row<VEC, EXACT>(p + (size_t)k * D, lane, D, P[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c + u < h.n) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) Y[v] = Y[v] + P[u][v];
            }
        }
    }
    if (h.out == kFinToStage) {
        if constexpr (OPT == 5) {
            float xi[VEC];
            load_row<VEC, EXACT>(f.X + (size_t)h.row * D, lane, D, xi);
#pragma unroll
            for (int v = 0; v < VEC; ++v) Y[v] = xi[v] + Y[v];
        }
        store_row<VEC, EXACT>(f.Xn + (size_t)h.row * D, lane, D, Y);
        if (f.push.world > 1u) {
            push_row<VEC, EXACT>(f.push, h.row, lane, D, Y);
            __builtin_amdgcn_s_waitcnt(0);
        }
    } else {
        store_row<VEC, EXACT>(f.partials + (size_t)h.out * D, lane, D, Y);
    }
}

// X[lo + r] = Xn[lo + r] for r < rows: folds a PARTIAL updated range back (out-of-order batches, reading the
// matrix in the middle of an epoch); a completed epoch swaps the matrices instead.
template <int VEC, bool EXACT>
__global__ __launch_bounds__(256) void commit_kernel(float *X, const float *Xn, uint32_t lo, uint32_t rows, uint32_t D) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wpb = blockDim.x >> 6;
    const uint32_t total = gridDim.x * wpb;
    for (uint32_t r = blockIdx.x * wpb + (threadIdx.x >> 6); r < rows; r += total) {
        float t[VEC];
        load_row<VEC, EXACT>(Xn + (size_t)(lo + r) * D, lane, D, t);
        store_row<VEC, EXACT>(X + (size_t)(lo + r) * D, lane, D, t);
    }
}

// ---- multi-GPU: push exchange over xGMI (include/f2v.h) ------------------------------------------------
// Every rank holds both matrices of every peer mapped into its address space (HIP IPC).  After a minibatch's step
// kernels a rank copies each of its new rows into the same row of the second matrix of every peer whose bit is set
// in the row's reader mask: one wavefront per row, one coalesced row load (the row was written a moment ago: L2 /
// MALL), one coalesced row store per reading peer.  The stores are posted writes on the direct xGMI link to that
// peer, so all 7 links of a GPU carry traffic at once and nothing waits for a reply.  The stores are written
// through at system scope and every wave waits for its acknowledgements: when the kernel has finished, its rows
// are in the peers' memory.  (A release fence per wave instead -- buffer_wbl2 -- walks the whole L2 each time.)
// The step and finalize kernels of a sharded run do the same from registers, while the row is still there (PUSH):
// the transfer then overlaps the rest of the launch instead of following it.

struct PushArgs {
    const float *src;         // local matrix that holds the new rows
    PushTargets to;
    uint32_t row_lo, rows;    // this rank's rows of the minibatch
    uint32_t D;
};

template <int VEC, bool EXACT>
__global__ __launch_bounds__(256) void push_rows_kernel(const PushArgs p) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wpb = blockDim.x >> 6;
    const uint32_t total = gridDim.x * wpb;
    for (uint32_t r = blockIdx.x * wpb + (threadIdx.x >> 6); r < p.rows; r += total) {
        const uint32_t row = p.row_lo + r;
        float t[VEC];
        load_row<VEC, EXACT>(p.src + (size_t)row * p.D, lane, p.D, t);
        push_row<VEC, EXACT>(p.to, row, lane, p.D, t);
    }
    __builtin_amdgcn_s_waitcnt(0);  // this wave's stores have been acknowledged by the peers' memory before it ends
}

// Landing-buffer mode, after the barrier: rows of minibatch [lo, lo+rows) that peers pushed into this rank's landing
// buffer (slot = row - lo) move to their place in the matrix.  Rows of this rank's own slice and rows nobody pushed
// here (their mask lacks this rank's bit) are skipped.
struct UnpackArgs {
    const float *landing;
    float *X;
    const uint32_t *masks;
    uint32_t lo, rows, my_lo, my_hi, D, self;
};

template <int VEC, bool EXACT>
__global__ __launch_bounds__(256) void unpack_rows_kernel(const UnpackArgs u) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wpb = blockDim.x >> 6;
    const uint32_t total = gridDim.x * wpb;
    for (uint32_t r = blockIdx.x * wpb + (threadIdx.x >> 6); r < u.rows; r += total) {
        const uint32_t row = u.lo + r;
        if (row >= u.my_lo && row < u.my_hi) continue;
        if (u.masks && !((u.masks[row] >> u.self) & 1u)) continue;
        float t[VEC];
        load_row<VEC, EXACT>(u.landing + (size_t)r * u.D, lane, u.D, t);
        store_row<VEC, EXACT>(u.X + (size_t)row * u.D, lane, u.D, t);
    }
}

// Flag barrier between minibatches, one 64-lane workgroup: lane r tells rank r "I have finished step `seq`" (a
// system-scope release store into r's flag array, which lives in fine-grained memory and is mapped like the
// matrices) and waits until rank r has told us the same.  The kernel runs on the engine's stream behind the push
// kernel, so "finished" covers this rank's pushes; the next minibatch's step kernel runs behind it, so it starts only
// when every peer's pushes have landed here.  Every wait is bounded by `timeout_ticks` of the 100 MHz wall clock: a
// missing peer sets *err and the grid drains.
struct BarrierArgs {
    unsigned long long *flags;                  // local: flags[r] = last step rank r has announced
    unsigned long long *peer_flags[kMaxRanks];  // the flag array of every rank
    uint32_t *err;
    unsigned long long seq, timeout_ticks;
    uint32_t self, world;
};

__global__ __launch_bounds__(64) void xgmi_barrier_kernel(const BarrierArgs b) {
    const uint32_t r = threadIdx.x;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __builtin_amdgcn_s_waitcnt(0);  // the write-back above has completed before any flag leaves
    if (r < b.world && r != b.self) {
        unsigned long long *theirs = b.peer_flags[0];
#pragma unroll
        for (int q = 1; q < kMaxRanks; ++q)
            if ((uint32_t)q == r) theirs = b.peer_flags[q];
        __hip_atomic_store(theirs + b.self, b.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = wall_clock64();
        // once a wait has given up the run is lost anyway: later barriers do not wait again (one time-out, not one per minibatch)
        const bool lost = __hip_atomic_load(b.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        // relaxed system-scope loads go to memory every time without invalidating this XCD's L2 on every poll
        while (!lost && __hip_atomic_load(b.flags + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < b.seq) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > b.timeout_ticks) {
                __hip_atomic_store(b.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // once: what the peers pushed is read from memory from here on
}

// masks[idx[k]] = val[k] in order of k is not needed: indices are distinct within one patch list
__global__ void mask_patch_kernel(uint32_t *masks, const uint32_t *idx, const uint32_t *val, uint32_t count) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < count) masks[idx[k]] = val[k];
}

// ---- non-parity fast mode: counter-based RNG on the device ---------------------------
// The reference's libc rand() stream is inherently serial (and, for the option-7 walks, its consumption is
// data dependent), so the parity path draws it on the host.  For very large N that costs seconds per run
// (init) or bounds the epoch (walks).  The fast mode replaces both with a stateless hash of
// (seed, stream, index): same distributions, different numbers -- it is NOT bit-comparable with the
// reference and is off unless "fast_rng" is set.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// X[k] = U[-1,1) (kind 0) or U[0,1) (kind 1) with 24 random bits, as randInitF / randInit distribute
__global__ void fast_init_kernel(float *X, uint64_t total, int kind, uint64_t seed) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += stride) {
        const float u = (float)(mix64(seed ^ (k * 0xD6E8FEB86659FD93ull)) >> 40) * (1.0f / 16777216.0f);
        X[k] = kind == 0 ? -1.0f + 2.0f * u : u;
    }
}

// One thread per vertex: the 5-step semi-random walk of sample/algorithms.cpp:1097-1118 -- a uniformly random
// neighbour except the last one if deg > 2, the first neighbour if deg == 2, otherwise colids[w] with the
// vertex id as edge index (the reference's quirk, clamped to the array).
__global__ void fast_walks_kernel(const uint32_t *rowptr, const uint32_t *colids, uint32_t n, uint64_t nnz, uint32_t *walks,
                                  uint64_t seed, uint64_t epoch) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w = i;
    for (uint32_t s = 0; s < 5; ++s) {
        const uint32_t rp = rowptr[w], deg = rowptr[w + 1] - rp;
        uint64_t j = w;
        if (deg > 2) j = rp + mix64(seed ^ mix64(epoch * 0x100000001B3ull + (uint64_t)i * 5u + s)) % (deg - 1);
        else if (deg == 2) j = rp;
        if (j >= nnz) j = nnz ? nnz - 1 : 0;
        w = colids[j];
        walks[(size_t)i * 5u + s] = w;
    }
}

// PMC calibration and on-box gather ceiling (f2v_diag_gather_rate): the step kernel's access pattern with a KNOWN byte count.  Every quarter-wave
// gathers whole 64*NB-float rows (NB x 16 lanes x dwordx4) named by `ids`, each row exactly once,
// and folds them into a checksum so that the loads stay live.
template <int NB>
__global__ __launch_bounds__(256) void gather_calibration_kernel(const float *table, const uint32_t *ids, uint32_t n_ids, float *out) {
    constexpr uint32_t D = 64u * NB;
    const uint32_t lane = threadIdx.x & 63u, t = lane & 15u;
    const uint32_t qid = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4u + (lane >> 4);
    const uint32_t total_q = gridDim.x * (blockDim.x >> 6) * 4u;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (uint32_t k = qid * 4u; k < n_ids; k += total_q * 4u) {
        float4 v[4][NB];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t j = ids[(k + u) < n_ids ? (k + u) : (n_ids - 1)];
#pragma unroll
            for (int b = 0; b < NB; ++b) v[u][b] = *reinterpret_cast<const float4 *>(table + (size_t)j * D + 64 * b + t * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int b = 0; b < NB; ++b) { acc.x += v[u][b].x; acc.y += v[u][b].y; acc.z += v[u][b].z; acc.w += v[u][b].w; }
    }
    const float s = (acc.x + acc.y) + (acc.z + acc.w);
    if (s == 12345.678f) out[0] = s;  // practically never: keeps the loads from being optimised away
}

// IPC preflight (f2v_diag_ipc_preflight): thread q stores a word into the mapped buffer and the mapped flag array of rank q
struct PreflightArgs {
    uint32_t *data[kMaxRanks];
    unsigned long long *flags[kMaxRanks];
    uint32_t self, world, value;
    uint64_t last_word;  // index of the last 32-bit word of the data buffers (the far end of the mapping is exercised too)
};

#ifdef F2V_TEST_HOOKS
// Self-test build only (f2v_test_plan_gather): the MEMORY side of a real launch and nothing else -- the plan's items in the plan's order, their
// neighbour rows gathered with the step kernel's pattern (lane groups in lockstep up to the longest item of the wavefront, U rows in
// flight), no interaction computed, no samples.  mode bit 0: the item's own row is read as well; bit 1: whole-row items store a row.
template <int LPI, int NB, int U, int G>
__global__ __launch_bounds__(256, 5) void plan_gather_kernel(const float *X, float *Xn, const Item *items, uint32_t n_items, const uint32_t *nbr_ids, uint32_t mode, float *out) {
    // G > 1 (experiment): a wavefront takes G consecutive groups of items, one after the other; all their items are requested at once and
    // the first ids of group k + 1 while group k's rows are gathered -- the start-up chain item -> ids -> rows is paid once per wavefront
    constexpr uint32_t D = 4u * LPI * NB, IPW = 64u / LPI;
    const uint32_t lane = threadIdx.x & 63u, t = lane & (LPI - 1u), q = lane / LPI;
    // (the plan's workgroups go to the XCDs round robin and hub pieces sit in the workgroups of "their" XCD: workgroup B of this grid takes
    // the plan's workgroups B % 8 + 8 (G (B / 8) + k), k < G -- the same XCD)
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), wpb = blockDim.x >> 6;
    auto group_of = [&](int k) { return ((blockIdx.x & 7u) + 8u * (G * (blockIdx.x >> 3) + (uint32_t)k)) * wpb + wv; };
    if (IPW * group_of(0) >= n_items) return;
    Item it[G];
#pragma unroll
    for (int k = 0; k < G; ++k) {
        const uint32_t idx = IPW * group_of(k) + q;
        if (idx < n_items) it[k] = items[idx];
        else { it[k].row = 0; it[k].nb = 0; it[k].cnt = 0; it[k].flags = 0; }
    }
    uint32_t j[U];
#pragma unroll
    for (int u = 0; u < U; ++u) j[u] = ((uint32_t)u < it[0].cnt) ? nbr_ids[it[0].nb + u] : 0u;
    float keep = 0.f;
#pragma unroll
    for (int k = 0; k < G; ++k) {
        const uint32_t idx = IPW * group_of(k) + q;
        float4 acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = (mode & 1u) ? *reinterpret_cast<const float4 *>(X + (size_t)it[k].row * D + t * 4 + 4 * LPI * b) : make_float4(0.f, 0.f, 0.f, 0.f);
        const uint32_t *ids = nbr_ids + it[k].nb;
        const uint32_t cnt = it[k].cnt, maxcnt = wave_max_of_items<LPI>(cnt);
        uint32_t g = 0;
        do {
            float4 xj[U][NB];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (g + u < cnt) {
#pragma unroll
                    for (int b = 0; b < NB; ++b) xj[u][b] = *reinterpret_cast<const float4 *>(X + (size_t)j[u] * D + t * 4 + 4 * LPI * b);
                }
            const bool last = g + U >= maxcnt;  // (uniform)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (last) j[u] = (k + 1 < G && (uint32_t)u < it[k + 1 < G ? k + 1 : k].cnt) ? nbr_ids[it[k + 1 < G ? k + 1 : k].nb + u] : 0u;
                else j[u] = (g + U + u < cnt) ? ids[g + U + u] : 0u;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (g + u < cnt) {
#pragma unroll
                    for (int b = 0; b < NB; ++b) { acc[b].x += xj[u][b].x; acc[b].y += xj[u][b].y; acc[b].z += xj[u][b].z; acc[b].w += xj[u][b].w; }
                }
            g += U;
        } while (g < maxcnt);
        if ((mode & 2u) && idx < n_items && !(it[k].flags & kItemPartial)) {
#pragma unroll
            for (int b = 0; b < NB; ++b) *reinterpret_cast<float4 *>(Xn + (size_t)it[k].row * D + t * 4 + 4 * LPI * b) = acc[b];
        } else {
            keep += acc[0].x + acc[NB - 1].y;
        }
    }
    if (keep == 12345.678f) out[0] = keep;
}
#endif

__global__ void preflight_write_kernel(const PreflightArgs a) {
    const uint32_t q = threadIdx.x;
    if (q >= a.world) return;
    uint32_t *d = a.data[0];
    unsigned long long *f = a.flags[0];
#pragma unroll
    for (int k = 1; k < kMaxRanks; ++k)
        if ((uint32_t)k == q) { d = a.data[k]; f = a.flags[k]; }
    __hip_atomic_store(d + a.self, a.value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(d + a.last_word - a.self, a.value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(f + a.self, (unsigned long long)a.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Which XCD runs which workgroup?  The one-launch minibatch ties a combine-tree node and the pieces it waits for to
// workgroups of equal index modulo 8, counting on the dispatcher dealing workgroups round robin over 8 XCDs
// (MI355X_MICROARCH.md, "Workgroup dispatch": observed, not promised).  f2v_create checks it on the device it got:
// out[b] = XCC_ID of workgroup b.  Anything else (a partitioned GPU, another XCD count) selects one launch per tree level.
__global__ void xcc_probe_kernel(uint32_t *out) {
    uint32_t id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(id));
    if (threadIdx.x == 0) out[blockIdx.x] = id;
}

// On-box streaming ceiling: dst = src, 16 bytes per lane.  Every workgroup copies ONE contiguous tile of
// U x 256 x 16 bytes, all U loads in flight before the first store, non-temporal both ways (nothing is re-read): measured
// 6.3-6.4 TB/s read + written on a 1-GiB copy, the guide's figure for the part (a grid-stride loop over 8192 workgroups,
// round 1's kernel, reaches 4.9; tools/src/stream_copy_sweep.hip has the sweep).
typedef float f32x4_copy_t __attribute__((ext_vector_type(4)));
template <int U>
__global__ __launch_bounds__(256) void stream_copy_kernel(const f32x4_copy_t *src, f32x4_copy_t *dst, uint64_t n4) {
    const uint64_t base = (uint64_t)blockIdx.x * (U * 256u) + threadIdx.x;
    f32x4_copy_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (base + (uint64_t)u * 256u < n4) v[u] = __builtin_nontemporal_load(src + base + (uint64_t)u * 256u);
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (base + (uint64_t)u * 256u < n4) __builtin_nontemporal_store(v[u], dst + base + (uint64_t)u * 256u);
}

#ifdef F2V_TEST_HOOKS
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
#endif  // F2V_TEST_HOOKS

#ifdef F2V_TEST_HOOKS
}  // inline namespace selftest
#endif
}  // namespace f2v
#endif

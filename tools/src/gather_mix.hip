// Do L2 hits and L2 misses of a row gather share one resource?  The step kernel's access pattern (16 lanes x dwordx4 x 2 per 512-byte row, 4 rows in
// flight per lane group, 4 groups per wavefront) over an id list that mixes a HOT table (stays in every XCD's L2) and a COLD one (uniformly random rows of
// a table that only the Infinity Cache or HBM holds): time against the share of hot reads.  t(f) = (1-f) t(0) + f t(1) says "one resource", max() says two.
//   hipcc --offload-arch=gfx950 -O3 -o bin/gather_mix tools/src/gather_mix.hip ;  bin/gather_mix [cold MiB = 150] [hot rows = 2048] [reads = 4 Mi]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <random>
#include <algorithm>
#include <cmath>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int U>
__global__ __launch_bounds__(256, 5) void gather_kernel(const float *table, const uint32_t *ids, uint32_t per_group, float *out) {
    const uint32_t lane = threadIdx.x & 63u, t = lane & 15u, q = lane >> 4;
    const uint32_t group = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4u + q;
    const uint32_t *my = ids + (size_t)group * per_group;
    float acc = 0.f;
    uint32_t j[U];
#pragma unroll
    for (int u = 0; u < U; ++u) j[u] = my[u];
    for (uint32_t g = 0; g < per_group; g += U) {
        float4 x[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float *src = table + (size_t)j[u] * 128u + t * 4u;
            x[u][0] = *reinterpret_cast<const float4 *>(src);
            x[u][1] = *reinterpret_cast<const float4 *>(src + 64);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) j[u] = (g + U + u < per_group) ? my[g + U + u] : 0u;
#pragma unroll
        for (int u = 0; u < U; ++u) acc += ((x[u][0].x + x[u][0].y) + (x[u][0].z + x[u][0].w)) + ((x[u][1].x + x[u][1].y) + (x[u][1].z + x[u][1].w));  // every component: the loads stay 16 bytes wide
    }
    if (acc == 12345.678f) out[0] = acc;
}

// the same gather with the ids taken grid-strided (consecutive lane groups take consecutive groups of U ids: the calibration kernel's way)
template <int U>
__global__ __launch_bounds__(256, 5) void gather_strided_ids_kernel(const float *table, const uint32_t *ids, uint32_t n_ids, float *out) {
    const uint32_t lane = threadIdx.x & 63u, t = lane & 15u;
    const uint32_t qid = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4u + (lane >> 4);
    const uint32_t total_q = gridDim.x * (blockDim.x >> 6) * 4u;
    float acc = 0.f;
    for (uint32_t k = qid * U; k < n_ids; k += total_q * U) {
        float4 x[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float *src = table + (size_t)ids[k + u] * 128u + t * 4u;
            x[u][0] = *reinterpret_cast<const float4 *>(src);
            x[u][1] = *reinterpret_cast<const float4 *>(src + 64);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += ((x[u][0].x + x[u][0].y) + (x[u][0].z + x[u][0].w)) + ((x[u][1].x + x[u][1].y) + (x[u][1].z + x[u][1].w));  // every component: the loads stay 16 bytes wide
    }
    if (acc == 12345.678f) out[0] = acc;
}

// per-group lists as in the step kernel, but a lane group fetches 16 ids with ONE load (lane t takes id g + t) and hands them round by DPP-free shuffles
__global__ __launch_bounds__(256, 5) void gather_wide_ids_kernel(const float *table, const uint32_t *ids, uint32_t per_group, float *out) {
    const uint32_t lane = threadIdx.x & 63u, t = lane & 15u, q = lane >> 4;
    const uint32_t group = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4u + q;
    const uint32_t *my = ids + (size_t)group * per_group;
    float acc = 0.f;
    uint32_t mine = my[t];
    for (uint32_t g = 0; g < per_group; g += 16) {
        const uint32_t next = (g + 16 + t < per_group) ? my[g + 16 + t] : 0u;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            float4 x[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t j = (uint32_t)__shfl((int)mine, (int)(q * 16 + h * 4 + u), 64);
                const float *src = table + (size_t)j * 128u + t * 4u;
                x[u][0] = *reinterpret_cast<const float4 *>(src);
                x[u][1] = *reinterpret_cast<const float4 *>(src + 64);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += ((x[u][0].x + x[u][0].y) + (x[u][0].z + x[u][0].w)) + ((x[u][1].x + x[u][1].y) + (x[u][1].z + x[u][1].w));  // every component: the loads stay 16 bytes wide
        }
        mine = next;
    }
    if (acc == 12345.678f) out[0] = acc;
}

// the same gather with the whole wavefront on ONE row (64 lanes x dwordx2 = 512 contiguous bytes per instruction), U rows in flight
template <int U>
__global__ __launch_bounds__(256, 5) void gather_row_per_wave_kernel(const float *table, const uint32_t *ids, uint32_t per_group, float *out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t *my = ids + (size_t)wave * per_group * 4u;  // a wavefront takes the lists of its four lane groups, one after the other
    const uint32_t total = per_group * 4u;
    float acc = 0.f;
    for (uint32_t g = 0; g < total; g += U) {
        float2 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = *reinterpret_cast<const float2 *>(table + (size_t)my[g + u] * 128u + lane * 2u);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += x[u].x + x[u].y;
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char **argv) {
    const size_t cold_mib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 150;
    const uint32_t hot_rows = argc > 2 ? (uint32_t)strtoul(argv[2], nullptr, 10) : 2048;
    const uint64_t reads = argc > 3 ? strtoull(argv[3], nullptr, 10) : (4ull << 20);
    const uint32_t cold_rows = (uint32_t)(cold_mib * 2048);
    const uint32_t rows = hot_rows + cold_rows;  // hot rows first
    const uint32_t blocks = 4096, groups = blocks * 4 * 4;
    const uint32_t per_group = (uint32_t)((reads / groups + 15) / 16 * 16);
    const uint64_t n = (uint64_t)per_group * groups;
    float *d_t, *d_o;
    uint32_t *d_i;
    CK(hipMalloc((void **)&d_t, (size_t)rows * 512));
    CK(hipMemset(d_t, 0, (size_t)rows * 512));
    CK(hipMalloc((void **)&d_i, n * 4));
    CK(hipMalloc((void **)&d_o, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("cold table %zu MiB (%u rows), hot table %u rows (%.1f MiB), %llu row reads of 512 B per launch (%.2f GB), 4096 x 256 threads, 5 waves/SIMD\n", cold_mib, cold_rows,
           hot_rows, hot_rows * 512.0 / (1 << 20), (unsigned long long)n, n * 512.0 * 1e-9);
    std::mt19937_64 mt(5);
    std::vector<uint32_t> ids(n);
    double t_cold = 0, t_hot = 0;
    const int variant = argc > 4 ? atoi(argv[4]) : 0;       // 0: the step kernel's pattern; 8: 8 rows in flight; 1: one row per wavefront instruction (16 in flight); 2: ids grid-strided; 3: 16 ids per load, shuffled
    const size_t lds_pad = argc > 5 ? strtoull(argv[5], nullptr, 10) : 0;  // dynamic LDS per workgroup: bounds the workgroups per CU (160 KiB / pad)
    printf("variant %d, %zu bytes of LDS padding per workgroup\n", variant, lds_pad);
    if (getenv("GATHER_MIX_ZIPF")) {
        // one table, rows drawn with power-law popularity (weight of the r-th most popular row: (r + 1)^-s, popular rows scattered over the table):
        // hits and misses as a graph's neighbour stream produces them, not as two uniform populations
        for (const char *tok = strtok(getenv("GATHER_MIX_ZIPF"), ","); tok; tok = strtok(nullptr, ",")) {
            const double sexp = atof(tok);
            std::vector<double> cdf(cold_rows);
            double acc = 0;
            for (uint32_t r = 0; r < cold_rows; r++) { acc += pow((double)r + 1.0, -sexp); cdf[r] = acc; }
            std::vector<uint32_t> perm(cold_rows);
            for (uint32_t r = 0; r < cold_rows; r++) perm[r] = r;
            for (uint32_t r = cold_rows - 1; r > 0; r--) std::swap(perm[r], perm[mt() % (r + 1)]);
            for (auto &v : ids) {
                const double u = (double)(mt() >> 11) * (1.0 / 9007199254740992.0) * acc;
                v = hot_rows + perm[(uint32_t)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin())];
            }
            CK(hipMemcpy(d_i, ids.data(), n * 4, hipMemcpyHostToDevice));
            float best = 1e30f;
            for (int r = -400; r < 7; r++) {
                if (r >= 0) CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL((gather_kernel<4>), dim3(blocks), dim3(256), 0, 0, d_t, d_i, per_group, d_o);
                if (r < 0) continue;
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            std::vector<uint32_t> sorted(ids);
            std::sort(sorted.begin(), sorted.end());
            const size_t distinct = std::unique(sorted.begin(), sorted.end()) - sorted.begin();
            printf("power law s = %.2f over %u rows: %zu distinct rows in %llu reads; %.1f us = %.2f TB/s of rows\n", sexp, cold_rows, distinct, (unsigned long long)n, best * 1e3,
                   n * 512.0 / (best * 1e-3) * 1e-12);
            fflush(stdout);
        }
        return 0;
    }
    const double fr[] = {0.0, 1.0, 0.25, 0.5, 0.75, 0.9};
    for (double f : fr) {
        for (auto &v : ids) {
            const bool hot = (double)(mt() >> 11) * (1.0 / 9007199254740992.0) < f;
            v = (hot && hot_rows) ? (uint32_t)(mt() % hot_rows) : hot_rows + (uint32_t)(mt() % cold_rows);
        }
        CK(hipMemcpy(d_i, ids.data(), n * 4, hipMemcpyHostToDevice));
        float best = 1e30f;
        const int warm = getenv("GATHER_MIX_WARM") ? atoi(getenv("GATHER_MIX_WARM")) : 400;  // untimed launches first: clocks settle ~15 ms after a change of load
        for (int r = -warm; r < 7; r++) {
            if (r >= 0) CK(hipEventRecord(e0, 0));
            if (variant == 8) hipLaunchKernelGGL((gather_kernel<8>), dim3(blocks), dim3(256), lds_pad, 0, d_t, d_i, per_group, d_o);
            else if (variant == 2) hipLaunchKernelGGL((gather_strided_ids_kernel<4>), dim3(blocks), dim3(256), lds_pad, 0, d_t, d_i, (uint32_t)n, d_o);
            else if (variant == 3) hipLaunchKernelGGL((gather_wide_ids_kernel), dim3(blocks), dim3(256), lds_pad, 0, d_t, d_i, per_group, d_o);
            else if (variant == 1) hipLaunchKernelGGL((gather_row_per_wave_kernel<16>), dim3(blocks), dim3(256), lds_pad, 0, d_t, d_i, per_group, d_o);
            else hipLaunchKernelGGL((gather_kernel<4>), dim3(blocks), dim3(256), lds_pad, 0, d_t, d_i, per_group, d_o);
            if (r < 0) continue;
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        if (f == 0.0) t_cold = best;
        if (f == 1.0) t_hot = best;
        const double add = (1 - f) * t_cold + f * t_hot, mx = std::max((1 - f) * t_cold, f * t_hot);
        printf("hot share %.2f: %.1f us  = %.2f TB/s of rows", f, best * 1e3, n * 512.0 / (best * 1e-3) * 1e-12);
        if (f != 0.0 && f != 1.0) printf("   (one resource: %.1f us, two: %.1f us)", add * 1e3, mx * 1e3);
        printf("\n");
        fflush(stdout);
    }
    return 0;
}

// Which copy kernel reaches the part's streaming ceiling?  (MI355X_MICROARCH.md: 6.29 TB/s for a float4 copy.)
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/stream_copy_sweep tools/src/stream_copy_sweep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// v0: grid-stride, one float4 per iteration (round 1's kernel)
__global__ __launch_bounds__(256) void copy_v0(const float4 *src, float4 *dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += stride) dst[k] = src[k];
}

// v1: every workgroup owns contiguous tiles of U*256 float4; U loads in flight before the first store
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_tile(const f32x4 *src, f32x4 *dst, size_t n4) {
    const size_t tile = (size_t)U * 256;
    for (size_t base = (size_t)blockIdx.x * tile; base < n4; base += (size_t)gridDim.x * tile) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t k = base + (size_t)u * 256 + threadIdx.x;
            if (k < n4) v[u] = NT ? __builtin_nontemporal_load(src + k) : src[k];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t k = base + (size_t)u * 256 + threadIdx.x;
            if (k < n4) { if (NT) __builtin_nontemporal_store(v[u], dst + k); else dst[k] = v[u]; }
        }
    }
}

template <typename F>
double time_it(F &&launch, size_t bytes, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double best = 0;
    for (int r = 0; r < reps + 2; r++) {
        hipEventRecord(e0, 0);
        launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2) { const double g = 2.0 * bytes / (ms * 1e-3) * 1e-9; if (g > best) best = g; }
    }
    return best;
}

int main(int argc, char **argv) {
    const size_t bytes = (argc > 1 ? atoll(argv[1]) : 1024ll) << 20;
    f32x4 *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipDeviceSynchronize();
    const size_t n4 = bytes / 16;
    const int reps = 10;
    printf("copy of %zu MiB (read + written bytes per second)\n", bytes >> 20);
    for (int g : {2048, 8192, 32768})
        printf("v0 grid-stride float4, %6d blocks: %7.0f GB/s\n", g, time_it([&] { copy_v0<<<g, 256>>>((const float4 *)a, (float4 *)b, n4); }, bytes, reps));
#define RUN(U, NT)                                                                                                          \
    for (int g : {1024, 2048, 4096, 16384, (int)((n4 + (size_t)U * 256 - 1) / ((size_t)U * 256))})                            \
        printf("tile U=%d %s, %7d blocks: %7.0f GB/s\n", U, NT ? "nt" : "  ", g, time_it([&] { copy_tile<U, NT><<<g, 256>>>(a, b, n4); }, bytes, reps));
    RUN(1, false) RUN(4, false) RUN(8, false) RUN(4, true) RUN(8, true) RUN(16, true)
    printf("hipMemcpyAsync D2D: %7.0f GB/s\n", time_it([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, bytes, reps));
    return 0;
}

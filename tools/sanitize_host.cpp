#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "f2v.h"
int main(int argc, char **argv) {
    uint32_t n; uint64_t nnz; uint32_t *rp, *ci;
    for (int a = 1; a < argc; a++) {
        if (f2v_read_mtx(argv[a], &n, &nnz, &rp, &ci)) { printf("read failed: %s\n", f2v_last_error()); return 1; }
        printf("%s: n=%u nnz=%llu\n", argv[a], n, (unsigned long long)nnz);
        for (uint32_t world : {1u, 2u, 3u, 8u}) for (uint32_t batch : {1u, 7u, 256u, 100000u}) {
            std::vector<uint32_t> m(n), ids = {0, n - 1, n / 2};
            if (f2v_push_masks(rp, ci, n, batch, world, ids.data(), ids.size(), m.data())) { printf("masks failed %s\n", f2v_last_error()); return 1; }
            std::vector<uint32_t> b(world + 1);
            for (uint32_t lo = 0; lo < n; lo += batch) {
                uint32_t hi = lo + batch < n ? lo + batch : n;
                if (f2v_shard_bounds(rp, lo, hi, world, b.data())) return 1;
                if (b[0] != lo || b[world] != hi) { printf("bounds wrong\n"); return 1; }
            }
        }
        std::vector<float> x((size_t)n * 16);
        f2v_rng *g = f2v_rng_create(1); f2v_rng_fill(g, x.data(), x.size(), 0); f2v_rng_jump(g, 12345); f2v_rng_destroy(g);
        {   // option 7's walks, two epochs from one stream (blocks of predicted stream positions, draws given back)
            std::vector<uint32_t> w((size_t)n * 5);
            f2v_rng *gw = f2v_rng_create(1);
            for (int e = 0; e < 2; e++)
                if (f2v_rng_walks(gw, rp, ci, n, nnz, w.data())) { printf("walks failed %s\n", f2v_last_error()); return 1; }
            for (uint32_t v : w) if (v >= n) { printf("walk sample outside the graph\n"); return 1; }
            f2v_rng_destroy(gw);
        }
        char name[512]; f2v_output_name(argv[a], "/tmp/asan/", 5, 0, 256, 16, 3, 5, name, sizeof name);
        if (f2v_write_embd(name, x.data(), n, 16)) return 1;
        if (f2v_write_csr_bin("/tmp/asan/g.f2vcsr", rp, ci, n, nnz)) return 1;
        uint32_t n2; uint64_t z2; uint32_t *r2, *c2;
        if (f2v_read_csr_bin("/tmp/asan/g.f2vcsr", &n2, &z2, &r2, &c2) || n2 != n || z2 != nnz) return 1;
        f2v_free(r2); f2v_free(c2); f2v_free(rp); f2v_free(ci);
    }
    float t[2048]; f2v_sm_table(t);
    printf("ok\n");
    return 0;
}

#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cstring>
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
        {   // the text reader (threaded above 4 MB: a wide matrix makes the file big enough) on what the writer wrote, then on damaged copies
            uint32_t rn, rd; float *rx = nullptr;
            if (f2v_read_embd(name, &rn, &rd, &rx) || rn != n || rd != 16) { printf("read_embd failed %s\n", f2v_last_error()); return 1; }
            f2v_free(rx);
            const uint32_t wn = 3000, wd = 200;
            std::vector<float> wx((size_t)wn * wd);
            std::mt19937 mt(7);
            for (auto &v : wx) v = (float)((int)(mt() % 2000001) - 1000000) * 1e-5f;
            wx[5] = 1e-42f; wx[6] = 3e38f; wx[7] = -0.0f; wx[8] = 1.0f / 0.0f;
            if (f2v_write_embd("/tmp/asan/wide.embd", wx.data(), wn, wd)) return 1;
            for (const char *thr : {"1", "3", "16"}) {
                setenv("F2V_IO_THREADS", thr, 1);
                if (f2v_read_embd("/tmp/asan/wide.embd", &rn, &rd, &rx) || rn != wn || rd != wd) { printf("wide read failed %s\n", f2v_last_error()); return 1; }
                f2v_free(rx);
            }
            FILE *fp = fopen("/tmp/asan/wide.embd", "rb");
            std::vector<char> txt; int ch;
            while ((ch = fgetc(fp)) != EOF) txt.push_back((char)ch);
            fclose(fp);
            auto damaged = [&](size_t len, long poke, char with) {
                FILE *o = fopen("/tmp/asan/bad.embd", "wb");
                std::vector<char> t2(txt.begin(), txt.begin() + len);
                if (poke >= 0 && (size_t)poke < len) t2[poke] = with;
                if (!t2.empty()) fwrite(t2.data(), 1, t2.size(), o);
                fclose(o);
                float *bx = nullptr; uint32_t a2, b2;
                const int rc = f2v_read_embd("/tmp/asan/bad.embd", &a2, &b2, &bx);
                if (rc == 0) f2v_free(bx);
                return rc;
            };
            for (size_t len : {(size_t)0, (size_t)3, (size_t)9, txt.size() / 3, txt.size() / 2 + 1, txt.size() - 2, txt.size() - 1})
                if (damaged(len, -1, 0) == 0 && len < txt.size() - 2) { printf("a truncated file was accepted (%zu bytes)\n", len); return 1; }
            for (long poke : {10L, (long)txt.size() / 4, (long)txt.size() / 2, (long)txt.size() - 5})
                for (char with : {'x', '-', ' ', '\0'}) (void)damaged(txt.size(), poke, with);
            unsetenv("F2V_IO_THREADS");
        }
        if (f2v_write_csr_bin("/tmp/asan/g.f2vcsr", rp, ci, n, nnz)) return 1;
        uint32_t n2; uint64_t z2; uint32_t *r2, *c2;
        if (f2v_read_csr_bin("/tmp/asan/g.f2vcsr", &n2, &z2, &r2, &c2) || n2 != n || z2 != nnz) return 1;
        f2v_free(r2); f2v_free(c2); f2v_free(rp); f2v_free(ci);
    }
    float t[2048]; f2v_sm_table(t);
    printf("ok\n");
    return 0;
}

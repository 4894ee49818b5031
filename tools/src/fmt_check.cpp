// Host only: f2v_host.cpp's fmt_g against sprintf("%g ") on EVERY float bit pattern (2^32 of them, threads), or on a sample.
//   g++ -O2 -pthread -I include -I force2vec_amd/csrc tools/src/fmt_check.cpp -o /tmp/fmt_check && /tmp/fmt_check [stride]
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define F2V_FMT_CHECK 1
#include "../../force2vec_amd/csrc/f2v_host.cpp"
int main(int argc, char **argv) {
    const unsigned long long stride = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    unsigned T = std::thread::hardware_concurrency();
    std::atomic<unsigned long long> bad{0}, fast{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++)
        th.emplace_back([&, t] {
            char a[64], b[64];
            unsigned long long nb = 0, nf = 0;
            for (unsigned long long bits = t * stride; bits < (1ull << 32); bits += (unsigned long long)T * stride) {
                const uint32_t u = (uint32_t)bits;
                float v;
                memcpy(&v, &u, 4);
                char *e = fmt_g(a, v);
                *e = 0;
                sprintf(b, "%g ", (double)v);
                if (strcmp(a, b) != 0) { if (nb++ < 4) fprintf(stderr, "MISMATCH bits %08x: fast '%s' sprintf '%s'\n", u, a, b); }
                const double x = v < 0 ? -(double)v : (double)v;
                nf += x >= 1e-4 && x < 1e6;
            }
            bad += nb;
            fast += nf;
        });
    for (auto &y : th) y.join();
    printf("stride %llu: %llu mismatches; %llu values in the fast range\n", stride, bad.load(), fast.load());
    return bad.load() ? 1 : 0;
}

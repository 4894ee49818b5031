// f2v_internal.h -- shared between the host-side boundary code and the HBM engine.
#ifndef F2V_INTERNAL_H_
#define F2V_INTERNAL_H_
#include <cstddef>
#include <cstdint>

namespace f2v {

constexpr int kSmTableSize = 2048;   // SM_TABLE_SIZE, sample/algorithms.h:43
constexpr double kSmBound = 6.0;     // SM_BOUND, sample/algorithms.h:44
constexpr int kWalkLength = 5;       // WALKLENGTH, sample/algorithms.cpp:1073
constexpr float kMaxBound = 5.0f;    // MAXBOUND, sample/algorithms.h:41

int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// glibc rand() stream (TYPE_3 additive feedback), see f2v_host.cpp
struct Rand {
    int32_t r[31];
    int f, b;
    void seed(uint32_t s);
    inline int next() {
        uint32_t v = (uint32_t)r[f] + (uint32_t)r[b];
        r[f] = (int32_t)v;
        if (++f >= 31) f = 0;
        if (++b >= 31) b = 0;
        return (int)(v >> 1);
    }
    // randIndex(max,min), sample/algorithms.cpp:55-58
    inline uint32_t index(uint32_t max_num, uint32_t min_num) { return ((uint32_t)next() % (max_num - min_num)) + min_num; }
    // Advance the stream by `k` draws in O(31^3 log k): x[i] = x[i-31] + x[i-3] (mod 2^32) is linear, so k steps are
    // one 31x31 matrix over Z/2^32 applied to the 31-word state.  This is what lets N*D draws be made in parallel
    // (and bit-identically to the serial stream).
    void jump(uint64_t k);
    // Take the last draw back (the recurrence is invertible: the word it overwrote is the sum minus the other tap).
    inline void back() {
        if (--f < 0) f = 30;
        if (--b < 0) b = 30;
        r[f] = (int32_t)((uint32_t)r[f] - (uint32_t)r[b]);
    }
};

void init_embeddings_host(Rand &g, float *x, size_t total, int kind);
void sm_table_host(float *t);
// One epoch's option-7 walk samples [5*n] drawn from `g` exactly as the reference's serial loop draws them (f2v_host.cpp)
void walks_host(Rand &g, const uint32_t *rowptr, const uint32_t *colids, uint32_t n, uint64_t nnz, uint32_t *walks);

}  // namespace f2v
#endif

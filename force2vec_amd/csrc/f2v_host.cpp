// f2v_host.cpp -- host-side pieces of the drop-in boundary: the libc rand() stream the
// reference draws from, MatrixMarket ingest into CSR, the .embd writer and file naming,
// the sigmoid table.  No device code here; see f2v_engine.hip for the HBM side.
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "f2v.h"
#include "f2v_internal.h"

namespace f2v {

static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

// ---- libc rand(): glibc's TYPE_3 additive-feedback generator -------------------------------
// r[i] = r[i-3] + r[i-31] (mod 2^32), output = r[i] >> 1; srandom_r seeds r[0..30] with the
// Lehmer sequence 16807*x mod (2^31-1) and discards the first 310 outputs.  Restated so that
// results do not depend on the platform libc; the reference seeds with srand(1)
// (Test/Force2Vec.cpp:126) and draws with rand() (sample/algorithms.cpp:41,50,56).
void Rand::seed(uint32_t s) {
    if (s == 0) s = 1;
    r[0] = (int32_t)s;
    for (int i = 1; i < 31; i++) {
        long hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
        long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        r[i] = (int32_t)w;
    }
    f = 3;
    b = 0;
    for (int i = 0; i < 310; i++) next();
}

namespace {

using Mat = std::vector<uint32_t>;  // 31 x 31, row-major, arithmetic mod 2^32

Mat mat_mul(const Mat &a, const Mat &b) {
    Mat c(31 * 31, 0u);
    for (int i = 0; i < 31; i++)
        for (int k = 0; k < 31; k++) {
            const uint32_t aik = a[i * 31 + k];
            if (!aik) continue;
            for (int j = 0; j < 31; j++) c[i * 31 + j] += aik * b[k * 31 + j];
        }
    return c;
}

// v = (x[i-31], ..., x[i-1]);  one draw: v' = (v[1..30], v[0] + v[28])
Mat step_matrix_pow(uint64_t k) {
    Mat result(31 * 31, 0u), base(31 * 31, 0u);
    for (int i = 0; i < 31; i++) result[i * 31 + i] = 1u;
    for (int i = 0; i < 30; i++) base[i * 31 + i + 1] = 1u;
    base[30 * 31 + 0] = 1u;
    base[30 * 31 + 28] = 1u;
    while (k) {
        if (k & 1) result = mat_mul(base, result);
        base = mat_mul(base, base);
        k >>= 1;
    }
    return result;
}

void apply_matrix(const Mat &m, Rand &g) {
    uint32_t v[31], w[31];
    for (int k = 0; k < 31; k++) v[k] = (uint32_t)g.r[(g.f + k) % 31];  // oldest first
    for (int i = 0; i < 31; i++) {
        uint32_t acc = 0;
        for (int j = 0; j < 31; j++) acc += m[i * 31 + j] * v[j];
        w[i] = acc;
    }
    for (int k = 0; k < 31; k++) g.r[k] = (int32_t)w[k];
    g.f = 0;
    g.b = 28;
}

inline void fill(Rand &g, float *x, size_t count, int kind) {
    // double arithmetic narrowed on store, as `-1.0 + 2.0 * rand()/(RAND_MAX+1.0)` compiles
    if (kind == F2V_INIT_SYMMETRIC)
        for (size_t k = 0; k < count; k++) x[k] = (float)(-1.0 + 2.0 * (double)g.next() / 2147483648.0);
    else
        for (size_t k = 0; k < count; k++) x[k] = (float)((double)g.next() / 2147483648.0);
}

}  // namespace

void Rand::jump(uint64_t k) {
    if (k) apply_matrix(step_matrix_pow(k), *this);
}

// randInitF / randInit: N*D draws of the ONE serial rand() stream.  The stream is cut into equal chunks whose start
// states come from Rand::jump, the chunks are filled by threads, and `g` leaves exactly as after `total` serial draws.
void init_embeddings_host(Rand &g, float *x, size_t total, int kind) {
    unsigned T = std::thread::hardware_concurrency();
    if (const char *e = getenv("F2V_IO_THREADS")) T = (unsigned)atoi(e);
    T = std::max(1u, std::min(T, 64u));
    if (total < (size_t)(1u << 20) || T == 1) {
        fill(g, x, total, kind);
        return;
    }
    const size_t chunk = (total + T - 1) / T;
    const Mat m = step_matrix_pow(chunk);
    std::vector<Rand> start(T, g);
    for (unsigned t = 1; t < T; t++) {
        start[t] = start[t - 1];
        apply_matrix(m, start[t]);
    }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++) {
        const size_t lo = std::min(total, (size_t)t * chunk), hi = std::min(total, lo + chunk);
        th.emplace_back([&, t, lo, hi] { fill(start[t], x + lo, hi - lo, kind); });
    }
    for (auto &y : th) y.join();
    g.jump(total);
}

void sm_table_host(float *t) {
    // init_SM_TABLE (sample/algorithms.cpp:757-764): float x, float exp, double reciprocal
    for (int i = 0; i < kSmTableSize; i++) {
        float x = (float)(2.0 * kSmBound * i / kSmTableSize - kSmBound);
        t[i] = (float)(1.0 / (double)(1.0f + expf(-x)));
    }
}

// Option 7's walks (sample/algorithms.cpp:1097-1118): from every vertex five steps -- degree > 2: a random neighbour except the
// last (ONE rand() draw); degree 2: the first neighbour; otherwise colids[w] with the VERTEX id as edge index (kept for parity,
// clamped to the array) -- each step's target is a sample and the next step's start.  rand() is drawn only at vertices of
// degree > 2, so where a walk's draws lie in the stream depends on every walk before it: the reference's loop is one chain of
// dependent cache misses (35 ns a step on RMAT-20).
//
// Here up to 64 consecutive walks are IN FLIGHT, each at its own step, from stream positions PREDICTED for them (a draw at every step
// but the first of a walk that starts at a vertex of degree <= 2, which is known beforehand); one pass over the walks in flight
// computes every one's next edge index (and prefetches it), a second pass reads the targets (and prefetches their row pointers).  A
// walk that reaches a vertex of degree <= 2 where a draw was predicted has drawn one number fewer than the younger walks in flight
// assumed: they -- and only they -- are dropped and issued again behind it, from the corrected position.  On RMAT-20 that happens to
// one walk in 32 (tools/walk_deficits.py), and the younger walks have by then taken a step or two each: round 3's blocks of 32
// (all walks step by step together, validated when the block was done, everything behind the first mispredicted walk thrown away and
// the block boundary drained) lost ~38 % of their work, this loses ~12 %.  Same samples, same stream position afterwards (draws taken
// ahead are given back, Rand::back), whatever the graph; where more than one walk in three is mispredicted (cora: half the steps draw
// nothing, and the graph sits in the cache) the reference's own serial loop runs, 64 walks at a time.
void walks_host(Rand &g, const uint32_t *rp, const uint32_t *ci, uint32_t n, uint64_t nnz, uint32_t *walks) {
    constexpr uint32_t L = (uint32_t)kWalkLength, SMAX = 64, RING = 1024, MASK = RING - 1;
    static_assert(RING >= 2 * SMAX * L && (RING & MASK) == 0, "the ring holds every draw a walk in flight may ask for");
    if (nnz == 0) {  // no vertex has a neighbour: nothing is drawn (the reference would read colids[0] of an empty array)
        for (size_t k = 0; k < (size_t)n * L; k++) walks[k] = 0;
        return;
    }
    const uint32_t last = (uint32_t)(nnz - 1);
    uint32_t ring[RING];
    uint64_t gen = 0;       // draws taken from g so far: ring[k & MASK] is draw k for gen - RING <= k < gen
    uint64_t next_off = 0;  // the stream position predicted for the next walk to be issued (exact once nothing is in flight)
    struct Slot { uint64_t base; uint32_t i, v, j; uint8_t s, used, pred; };  // base: the stream position of the walk's first draw
    Slot sl[SMAX];
    uint32_t head = 0, count = 0, next_i = 0;
    // how many walks are in flight follows how often the predictions have been failing
    // (at most a quarter of them are issued per pass, so that the walks in flight stay staggered over the five steps: a walk that
    // mispredicts at its step s then finds the younger ones at steps below s, not all at s.  RMAT-20, build container, ms per epoch:
    // round 3's blocks 175-183; 32 in flight 156-170; 48 / 64 in flight, 12 ... 20 issued per pass 141-152; with the true
    // positions given (nothing ever dropped: the bound of any one-thread scheme) 96-107)
    uint32_t width = 32, issue_cap = 8, seen = 0, missed = 0;
    const char *fixed = getenv("F2V_WALKS_IN_FLIGHT");  // (measurements: a fixed number of walks in flight, F2V_WALKS_ISSUE per pass)
    if (fixed) {
        width = std::max(1u, std::min<uint32_t>((uint32_t)atoi(fixed), SMAX));
        issue_cap = std::max(1u, width / 4u);
        if (const char *e = getenv("F2V_WALKS_ISSUE")) issue_cap = std::max(1u, (uint32_t)atoi(e));
    }
    auto adapt = [&] {
        if (seen >= 512u) { seen >>= 1; missed >>= 1; }
        if (fixed) return;
        // fewer than one walk in 24 mispredicted -> 64 in flight (the memory latency is what there is to hide); up to one in three ->
        // about one run's worth; beyond that the serial loop
        width = 24u * missed <= seen ? SMAX : 3u * missed <= seen ? std::max(4u, std::min(32u, seen / std::max(missed, 1u))) : 1u;
        issue_cap = std::max(2u, width / 4u);
    };
    while (next_i < n || count) {
        if (width == 1u && count == 0u) {
            // the reference's serial loop, 64 walks at a time (draws straight from the generator: give back what was taken ahead)
            while (gen > next_off) { g.back(); gen--; }
            const uint32_t stop = std::min(n, next_i + 64u);
            for (; next_i < stop; next_i++) {
                uint32_t v = next_i, draws = 0;
                for (uint32_t s = 0; s < L; s++) {
                    const uint32_t lo = rp[v], deg = rp[v + 1] - lo;
                    uint32_t j = v;
                    if (deg > 2) { j = g.index(rp[v + 1] - 1, lo); draws++; }
                    else if (deg == 2) j = lo;
                    if (j > last) j = last;
                    v = ci[j];
                    walks[(size_t)next_i * L + s] = v;
                }
                gen += draws;
                next_off += draws;
                seen++;
                missed += draws < L - 1u ? 1u : 0u;  // (what would have been mispredicted, more or less: a step past the first drew nothing)
            }
            adapt();
            continue;
        }
        // issue walks up to the width
        uint32_t issued = 0;
        while (count < width && next_i < n && issued++ < issue_cap) {
            Slot &x = sl[(head + count) % SMAX];
            x.i = x.v = next_i;
            x.s = x.used = 0;
            x.base = next_off;
            x.pred = (uint8_t)(rp[next_i + 1] - rp[next_i] > 2 ? L : L - 1u);
            next_off += x.pred;
            next_i++;
            count++;
        }
        while (gen < next_off) { ring[gen & MASK] = (uint32_t)g.next(); gen++; }
        // pass 1: every walk's next edge index
        for (uint32_t k = 0; k < count; k++) {
            Slot &x = sl[(head + k) % SMAX];
            const uint32_t v = x.v, lo = rp[v], deg = rp[v + 1] - lo;
            uint32_t j = v;
            if (deg > 2) {
                j = ring[(x.base + x.used) & MASK] % (deg - 1) + lo;  // randIndex(rowptr[v+1] - 1, rowptr[v])
                x.used++;
            } else {
                if (deg == 2) j = lo;
                if (x.s != 0) {
                    // a draw was predicted here: this walk takes one number fewer than the younger walks in flight assumed -- they go
                    // back to the queue, behind the corrected position
                    x.pred--;
                    missed++;
                    next_i = x.i + 1;
                    next_off = x.base + x.pred;
                    count = k + 1;
                }
            }
            if (j > last) j = last;
            x.j = j;
            __builtin_prefetch(ci + j);
        }
        // pass 2: the targets
        for (uint32_t k = 0; k < count; k++) {
            Slot &x = sl[(head + k) % SMAX];
            const uint32_t v = ci[x.j];
            walks[(size_t)x.i * L + x.s] = v;
            x.v = v;
            x.s++;
            __builtin_prefetch(rp + v);
        }
        while (count && sl[head].s == L) {  // (walks finish in the order they were issued)
            head = (head + 1) % SMAX;
            count--;
            seen++;
        }
        adapt();
    }
    while (gen > next_off) { g.back(); gen--; }
}

}  // namespace f2v

using namespace f2v;

extern "C" {

const char *f2v_last_error(void) { return g_err; }
const char *f2v_version(void) { return "f2v-mi355x 0.5 (gfx950)"; }

f2v_rng *f2v_rng_create(uint32_t seed) {
    Rand *g = new Rand();
    g->seed(seed);
    return reinterpret_cast<f2v_rng *>(g);
}
void f2v_rng_destroy(f2v_rng *g) { delete reinterpret_cast<Rand *>(g); }
int f2v_rng_next(f2v_rng *g) { return reinterpret_cast<Rand *>(g)->next(); }
void f2v_rng_jump(f2v_rng *g, uint64_t k) { reinterpret_cast<Rand *>(g)->jump(k); }
int f2v_rng_fill(f2v_rng *g, float *out, uint64_t count, int kind) {
    if (!g || (!out && count) || (kind != F2V_INIT_SYMMETRIC && kind != F2V_INIT_UNIT)) return fail(F2V_EINVAL, "f2v_rng_fill: bad argument");
    init_embeddings_host(*reinterpret_cast<Rand *>(g), out, (size_t)count, kind);
    return F2V_OK;
}

int f2v_rng_walks(f2v_rng *g, const uint32_t *rowptr, const uint32_t *colids, uint32_t n, uint64_t nnz, uint32_t *walks_out) {
    if (!g || !rowptr || (!colids && nnz) || !walks_out) return fail(F2V_EINVAL, "f2v_rng_walks: null argument");
    if (rowptr[0] != 0 || rowptr[n] != nnz) return fail(F2V_EINVAL, "f2v_rng_walks: rowptr[0] must be 0 and rowptr[n] == nnz");
    for (uint64_t k = 0; k < nnz; k++)
        if (colids[k] >= n) return fail(F2V_EINVAL, "f2v_rng_walks: column id %u outside the graph", colids[k]);
    walks_host(*reinterpret_cast<Rand *>(g), rowptr, colids, n, nnz, walks_out);
    return F2V_OK;
}

int f2v_sm_table(float *t) {
    if (!t) return fail(F2V_EINVAL, "f2v_sm_table: null output");
    sm_table_host(t);
    return F2V_OK;
}

void f2v_free(void *p) { free(p); }

// MatrixMarket coordinate text -> CSR with the reference's semantics (sample/IO.h:59-156):
//  * leading lines starting with '%' are header/comments; the word "symmetric" in any of
//    them mirrors every off-diagonal entry and DROPS diagonal entries (IO.h:122-134);
//    a general matrix keeps its diagonal;
//  * entries are "row col [value]" (1-based); the value is ignored by options 5-11;
//  * duplicates are kept; column ids end up ascending inside each row
//    (CSC per-column sort CSC.h:173-186, then transpose CSR.h:172-182).
int f2v_read_mtx(const char *path, uint32_t *n_out, uint64_t *nnz_out, uint32_t **rowptr_out, uint32_t **colids_out) {
    if (!path || !n_out || !nnz_out || !rowptr_out || !colids_out) return fail(F2V_EINVAL, "f2v_read_mtx: null argument");
    FILE *fp = fopen(path, "rb");
    if (!fp) return fail(F2V_EIO, "f2v_read_mtx: cannot open %s: %s", path, strerror(errno));
    // slurp: the parse below works on memory (fast path for 10^8-edge files)
    const bool trace = getenv("F2V_IO_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char *what) {
        if (trace) { const double t = now(); fprintf(stderr, "f2v_read_mtx: %-22s %.3f s\n", what, t - t_prev); t_prev = t; }
    };
    fseek(fp, 0, SEEK_END);
    long fsz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    std::unique_ptr<char[]> buf(new char[(size_t)fsz + 1]);
    size_t got = fread(buf.get(), 1, (size_t)fsz, fp);
    fclose(fp);
    buf[got] = 0;
    lap("read file");
    const char *p = buf.get(), *end = buf.get() + got;
    static const char kSym[] = "symmetric";
    bool symmetric = false;
    while (p < end && *p == '%') {
        const char *eol = (const char *)memchr(p, '\n', (size_t)(end - p));
        if (!eol) eol = end;
        if (std::search(p, eol, kSym, kSym + 9) != eol) symmetric = true;
        p = eol < end ? eol + 1 : end;
    }
    auto parse_uint = [&](uint64_t &v) -> bool {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) p++;
        if (p >= end || *p < '0' || *p > '9') return false;
        uint64_t x = 0;
        while (p < end && *p >= '0' && *p <= '9') x = x * 10 + (uint64_t)(*p++ - '0');
        v = x;
        return true;
    };
    auto skip_line = [&]() {
        const char *eol = (const char *)memchr(p, '\n', (size_t)(end - p));
        p = eol ? eol + 1 : end;
    };
    uint64_t m = 0, n = 0, nz = 0;
    if (!parse_uint(m) || !parse_uint(n) || !parse_uint(nz)) return fail(F2V_EIO, "f2v_read_mtx: %s: bad size line", path);
    skip_line();
    if (m >= 0xFFFFFFFFull) return fail(F2V_EINVAL, "f2v_read_mtx: %llu rows exceed 32-bit vertex ids", (unsigned long long)m);

    // ---- parallel parse: the entry lines are cut into T byte ranges at line boundaries; every thread turns its
    // lines into (row, col) pairs (kSkip for a line that does not start with two integers, as the serial rule).
    const char *body = p;
    const size_t body_len = (size_t)(end - body);
    unsigned T = std::thread::hardware_concurrency();
    if (const char *e = getenv("F2V_IO_THREADS")) T = (unsigned)atoi(e);
    T = std::max(1u, std::min(T, 64u));
    if (body_len < (8u << 20)) T = 1;
    constexpr uint32_t kSkip = 0xFFFFFFFFu;
    struct Part { std::vector<uint32_t> r, c; uint64_t bad_line = ~0ull; };
    std::vector<Part> parts(T);
    std::vector<const char *> cut(T + 1);
    cut[0] = body;
    cut[T] = end;
    for (unsigned t = 1; t < T; t++) {
        const char *q = body + body_len * t / T;
        const char *eol = (const char *)memchr(q, '\n', (size_t)(end - q));
        cut[t] = eol ? eol + 1 : end;
    }
    for (unsigned t = 1; t <= T; t++) cut[t] = std::max(cut[t], cut[t - 1]);
    auto parse_range = [&](unsigned t) {
        const char *q = cut[t], *qe = cut[t + 1];
        Part &P = parts[t];
        P.r.reserve((size_t)(qe - q) / 12 + 16);
        P.c.reserve((size_t)(qe - q) / 12 + 16);
        auto num = [&](uint64_t &v) -> bool {
            while (q < qe && (*q == ' ' || *q == '\t' || *q == '\r')) q++;
            if (q >= qe || *q < '0' || *q > '9') return false;
            uint64_t x = 0;
            while (q < qe && *q >= '0' && *q <= '9') x = x * 10 + (uint64_t)(*q++ - '0');
            v = x;
            return true;
        };
        while (q < qe) {
            uint64_t r = 0, c = 0;
            const bool ok = num(r) && num(c);
            const char *eol = (const char *)memchr(q, '\n', (size_t)(qe - q));
            q = eol ? eol + 1 : qe;
            if (!ok) { P.r.push_back(kSkip); P.c.push_back(kSkip); continue; }
            if (r == 0 || c == 0 || r > m || c > m) { if (P.bad_line == ~0ull) P.bad_line = P.r.size(); P.r.push_back(kSkip); P.c.push_back(kSkip); continue; }
            P.r.push_back((uint32_t)(r - 1));
            P.c.push_back((uint32_t)(c - 1));
        }
    };
    auto run_parallel = [&](unsigned nthreads, const std::function<void(unsigned)> &fn) {
        if (nthreads <= 1) { fn(0); return; }
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nthreads; t++) th.emplace_back(fn, t);
        fn(0);
        for (auto &x : th) x.join();
    };
    run_parallel(T, parse_range);
    lap("parse");
    // only the first nz lines count (IO.h:108: `while (!eof && cnz < nnz)`); an out-of-range id among them is an error
    std::vector<uint64_t> line0(T + 1, 0);
    for (unsigned t = 0; t < T; t++) line0[t + 1] = line0[t] + parts[t].r.size();
    for (unsigned t = 0; t < T; t++)
        if (parts[t].bad_line != ~0ull && line0[t] + parts[t].bad_line < nz)
            return fail(F2V_EIO, "f2v_read_mtx: %s: entry %llu out of range", path, (unsigned long long)(line0[t] + parts[t].bad_line));
    // ---- CSR: per-row counts, scatter into the row buckets, sort every row -- all over row-disjoint or atomic slots
    uint32_t *rowptr = (uint32_t *)calloc((size_t)m + 1, sizeof(uint32_t));
    if (!rowptr) return fail(F2V_ENOMEM, "f2v_read_mtx: out of memory");
    std::vector<std::atomic<uint32_t>> fill((size_t)m);
    for (auto &f : fill) f.store(0, std::memory_order_relaxed);
    auto for_entries = [&](unsigned t, auto &&emit) {
        const Part &P = parts[t];
        const uint64_t lim = nz > line0[t] ? std::min<uint64_t>(nz - line0[t], P.r.size()) : 0;
        for (uint64_t k = 0; k < lim; k++) {
            const uint32_t r = P.r[k], c = P.c[k];
            if (r == kSkip) continue;
            if (symmetric) {
                if (r == c) continue;  // IO.h:131-134: diagonal entries of a symmetric file are dropped
                emit(r, c);
                emit(c, r);
            } else {
                emit(r, c);
            }
        }
    };
    run_parallel(T, [&](unsigned t) { for_entries(t, [&](uint32_t r, uint32_t) { fill[r].fetch_add(1, std::memory_order_relaxed); }); });
    uint64_t cnt = 0;
    for (size_t i = 0; i < m; i++) {
        rowptr[i] = (uint32_t)cnt;
        cnt += fill[i].load(std::memory_order_relaxed);
        if (cnt >= 0xFFFFFFFFull) { free(rowptr); return fail(F2V_EINVAL, "f2v_read_mtx: nnz exceeds 32-bit row pointers"); }
        fill[i].store(rowptr[i], std::memory_order_relaxed);
    }
    rowptr[m] = (uint32_t)cnt;
    lap("count rows");
    uint32_t *colids = (uint32_t *)malloc((cnt ? cnt : 1) * sizeof(uint32_t));
    if (!colids) { free(rowptr); return fail(F2V_ENOMEM, "f2v_read_mtx: out of memory"); }
    run_parallel(T, [&](unsigned t) { for_entries(t, [&](uint32_t r, uint32_t c) { colids[fill[r].fetch_add(1, std::memory_order_relaxed)] = c; }); });
    lap("scatter");
    // ascending column ids inside each row (duplicates kept): CSC sort + transpose of the reference (CSC.h:173-186, CSR.h:172-182)
    std::atomic<uint64_t> next_row{0};
    run_parallel(T, [&](unsigned) {
        for (;;) {
            const uint64_t i0 = next_row.fetch_add(4096, std::memory_order_relaxed);
            if (i0 >= m) break;
            const uint64_t i1 = std::min<uint64_t>(i0 + 4096, m);
            for (uint64_t i = i0; i < i1; i++) std::sort(colids + rowptr[i], colids + rowptr[i + 1]);
        }
    });
    lap("sort rows");
    *n_out = (uint32_t)m;
    *nnz_out = cnt;
    *rowptr_out = rowptr;
    *colids_out = colids;
    return F2V_OK;
}

// writeToFile (sample/algorithms.h:118-136): ostream << float is "%g" with 6 significant digits.
// "%g " of a float as glibc prints it (what `out << value << " "` writes, six significant digits) -- exactly, without printf: for
// 1e-4 <= |v| < 1e6 (where %g uses fixed notation) the value scaled to six significant digits, |v| * 10^(5-k), is an EXACT double (a
// float has 24 significant bits, 10^e for e <= 10 is 2^e times a 24-bit integer: 48 bits), so the round-half-even decision is exact
// too; everything else -- zeros, tiny and huge values, a carry into 1e+06, inf, nan -- goes through sprintf.  ~8x faster than
// sprintf("%g"), and the 134 M values of an RMAT-20 run at D = 128 are what the CLI spends most of its time on after training.
// Byte-identical to sprintf on every float checked (tests/test_host_boundary.py: edge cases + random bit patterns; tools/fmt_check.cpp: 2^32).
static inline char *fmt_g(char *q, float v) {
    const double x = (double)v;
    const double a = x < 0 ? -x : x;
    if (!(a >= 1e-4 && a < 1e6)) return q + sprintf(q, "%g ", x);  // (also nan; 1e-4 as a double is above the true 1e-4: the exact test follows)
    const double t = a * 1e4;  // exact; t in [1, 1e10) once a >= 1e-4 for real
    if (t < 1.0) return q + sprintf(q, "%g ", x);
    static const double p10[11] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10};
    int k = 0;  // 10^k <= t < 10^(k+1)
    while (t >= p10[k + 1]) k++;
    int X = k - 4;                      // the decimal exponent of a
    const double s = a * p10[5 - X];    // exact: a * 10^(5 - X) in [1e5, 1e6)
    uint32_t d = (uint32_t)s;           // floor
    const double frac = s - (double)d;  // exact
    if (frac > 0.5 || (frac == 0.5 && (d & 1u))) d++;
    if (d == 1000000u) {
        d = 100000u;
        if (++X == 6) return q + sprintf(q, "%g ", x);
    }
    if (x < 0) *q++ = '-';
    char dig[6];
    for (int i = 5; i >= 0; i--) { dig[i] = (char)('0' + d % 10u); d /= 10u; }
    int last = 5;  // trailing zeros go
    while (last > 0 && dig[last] == '0') last--;
    if (X >= 0) {
        for (int i = 0; i <= X; i++) *q++ = dig[i];
        if (last > X) {
            *q++ = '.';
            for (int i = X + 1; i <= last; i++) *q++ = dig[i];
        }
    } else {
        *q++ = '0';
        *q++ = '.';
        for (int i = 0; i < -X - 1; i++) *q++ = '0';
        for (int i = 0; i <= last; i++) *q++ = dig[i];
    }
    *q++ = ' ';
    return q;
}

int f2v_write_embd(const char *path, const float *x, uint32_t n, uint32_t dim) {
    if (!path || !x) return fail(F2V_EINVAL, "f2v_write_embd: null argument");
    FILE *fp = fopen(path, "wb");
    if (!fp) return fail(F2V_EIO, "f2v_write_embd: cannot open %s: %s", path, strerror(errno));
    fprintf(fp, "%u %u\n", n, dim);
    // The text is what `out << value << " "` writes (six significant digits: %g), row by row -- and formatting IS the cost: 134 M values
    // for RMAT-20 at D = 128, ~20 s on one thread, more than 1200 epochs of training take.  Rows are formatted by the host's threads
    // (F2V_IO_THREADS) in slices of a few thousand rows, a round of slices at a time, and written in order: the same bytes.
    unsigned T = std::thread::hardware_concurrency();
    if (const char *e = getenv("F2V_IO_THREADS")) T = (unsigned)atoi(e);
    T = std::max(1u, std::min(T, 64u));
    if ((uint64_t)n * dim < (1u << 20)) T = 1;
    const uint32_t slice = std::max<uint32_t>(1u, (uint32_t)std::min<uint64_t>(4096, (4ull << 20) / std::max<uint64_t>((uint64_t)dim * 12, 1)));  // ~4 MB of text per slice
    auto format = [&](uint32_t lo, uint32_t hi, std::vector<char> &buf) {
        buf.resize((size_t)(hi - lo) * ((size_t)dim * 16 + 16));
        char *q = buf.data();
        for (uint32_t i = lo; i < hi; i++) {
            q += sprintf(q, "%u ", i + 1);
            const float *row = x + (size_t)i * dim;
            for (uint32_t d = 0; d < dim; d++) q = fmt_g(q, row[d]);
            *q++ = '\n';
        }
        buf.resize((size_t)(q - buf.data()));
    };
    std::vector<std::vector<char>> bufs(T);
    bool ok = true;
    for (uint64_t base = 0; base < n && ok; base += (uint64_t)T * slice) {
        const unsigned live = (unsigned)std::min<uint64_t>(T, (n - base + slice - 1) / slice);
        std::vector<std::thread> th;
        for (unsigned t = 1; t < live; t++)
            th.emplace_back([&, t] { format((uint32_t)(base + (uint64_t)t * slice), (uint32_t)std::min<uint64_t>(n, base + (uint64_t)(t + 1) * slice), bufs[t]); });
        format((uint32_t)base, (uint32_t)std::min<uint64_t>(n, base + slice), bufs[0]);
        for (auto &y : th) y.join();
        for (unsigned t = 0; t < live && ok; t++) ok = fwrite(bufs[t].data(), 1, bufs[t].size(), fp) == bufs[t].size();
    }
    if (!ok) {
        fclose(fp);
        return fail(F2V_EIO, "f2v_write_embd: short write to %s", path);
    }
    if (fclose(fp) != 0) return fail(F2V_EIO, "f2v_write_embd: close failed for %s", path);
    return F2V_OK;
}

// The reader of what f2v_write_embd / writeToFile wrote ("<N> <D>" then "<id> v0 ... " per row, ids 1-based, any row order):
// performancescores/runnodeclassclust.py:57-79 reads the same.  *x_out: malloc'ed N x D floats (f2v_free).
int f2v_read_embd(const char *path, uint32_t *n_out, uint32_t *dim_out, float **x_out) {
    if (!path || !n_out || !dim_out || !x_out) return fail(F2V_EINVAL, "f2v_read_embd: null argument");
    FILE *fp = fopen(path, "rb");
    if (!fp) return fail(F2V_EIO, "f2v_read_embd: cannot open %s: %s", path, strerror(errno));
    // slurp, then parse on the host's threads: the file is white-space separated tokens -- "<N> <D>", then per row its 1-based id and D
    // values, rows in any order (what fscanf("%f") read one token at a time through round 3: ~15 s for an RMAT-20 run at D = 128)
    fseek(fp, 0, SEEK_END);
    const long fsz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    if (fsz < 0) { fclose(fp); return fail(F2V_EIO, "f2v_read_embd: cannot size %s", path); }
    std::unique_ptr<char[]> buf(new (std::nothrow) char[(size_t)fsz + 1]);
    if (!buf) { fclose(fp); return fail(F2V_ENOMEM, "f2v_read_embd: out of memory"); }
    const size_t got = fread(buf.get(), 1, (size_t)fsz, fp);
    fclose(fp);
    buf[got] = 0;
    const char *p = buf.get(), *end = buf.get() + got;
    auto is_ws = [](char ch) { return ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r' || ch == '\v' || ch == '\f'; };
    auto header_uint = [&](unsigned long long &v) -> bool {
        while (p < end && is_ws(*p)) p++;
        if (p >= end || *p < '0' || *p > '9') return false;
        unsigned long long x = 0;
        while (p < end && *p >= '0' && *p <= '9') { x = x * 10 + (unsigned long long)(*p - '0'); if (x > 0xFFFFFFFFull) return false; p++; }
        if (p < end && !is_ws(*p)) return false;
        v = x;
        return true;
    };
    unsigned long long n = 0, dim = 0;
    // (no bound on D beyond the header's own fields: the reference's writeToFile has none, and the engine's generic kernel takes any D)
    if (!header_uint(n) || !header_uint(dim) || n == 0 || dim == 0 || n > (~(size_t)0) / sizeof(float) / dim)
        return fail(F2V_EIO, "f2v_read_embd: %s does not start with '<N> <D>'", path);
    float *x = static_cast<float *>(malloc((size_t)n * dim * sizeof(float)));
    if (!x) return fail(F2V_ENOMEM, "f2v_read_embd: out of memory");
    const char *body = p;
    const size_t body_len = (size_t)(end - body);
    unsigned T = std::thread::hardware_concurrency();
    if (const char *e = getenv("F2V_IO_THREADS")) T = (unsigned)atoi(e);
    T = std::max(1u, std::min(T, 64u));
    if (body_len < (4u << 20)) T = 1;
    // byte ranges cut at white space; pass 1 counts each range's tokens, pass 2 parses them knowing where in the N x (D + 1) grid they lie
    std::vector<const char *> cut(T + 1);
    cut[0] = body;
    cut[T] = end;
    for (unsigned t = 1; t < T; t++) {
        const char *q = body + body_len * t / T;
        while (q < end && !is_ws(*q)) q++;
        cut[t] = q;
    }
    for (unsigned t = 1; t <= T; t++) cut[t] = std::max(cut[t], cut[t - 1]);
    std::vector<unsigned long long> first(T + 1, 0);
    auto parallel = [&](auto &&fn) {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < T; t++) th.emplace_back(fn, t);
        fn(0u);
        for (auto &y : th) y.join();
    };
    parallel([&](unsigned t) {
        unsigned long long c = 0;
        for (const char *q = cut[t], *qe = cut[t + 1]; q < qe;) {
            while (q < qe && is_ws(*q)) q++;
            if (q >= qe) break;
            c++;
            while (q < qe && !is_ws(*q)) q++;
        }
        first[t + 1] = c;
    });
    for (unsigned t = 0; t < T; t++) first[t + 1] += first[t];
    const unsigned long long per_row = dim + 1, want = n * per_row;
    if (first[T] != want) {
        free(x);
        if (first[T] < want) return fail(F2V_EIO, "f2v_read_embd: %s: row %llu of %llu is malformed, out of range or repeated", path, first[T] / per_row + 1, n);
        return fail(F2V_EIO, "f2v_read_embd: %s holds more than the %llu rows its header announces", path, n);
    }
    std::vector<std::atomic<uint8_t>> seen((size_t)n);
    for (auto &f : seen) f.store(0, std::memory_order_relaxed);
    std::atomic<unsigned long long> bad_row{~0ull};
    auto flag_bad = [&](unsigned long long row) {
        unsigned long long cur = bad_row.load();
        while (row < cur && !bad_row.compare_exchange_weak(cur, row)) {}
    };
    parallel([&](unsigned t) {
        unsigned long long k = first[t];  // global token index
        unsigned long long id = 0;        // the row the current tokens belong to (0: its id token lies in an earlier range)
        const char *q = cut[t], *qe = cut[t + 1];
        if (k % per_row != 0) {
            // this range starts inside a row whose id token belongs to an earlier range: find it by walking back from the range's start
            unsigned long long back = k % per_row;
            const char *r = q;
            while (back) {
                while (r > body && is_ws(r[-1])) r--;
                while (r > body && !is_ws(r[-1])) r--;
                back--;
            }
            char *e2 = nullptr;
            const unsigned long long v = strtoull(r, &e2, 10);
            id = (e2 != r && (e2 >= end || is_ws(*e2)) && v >= 1 && v <= n) ? v : 0;  // (a bad id is reported by the range that owns the token)
        }
        while (q < qe) {
            while (q < qe && is_ws(*q)) q++;
            if (q >= qe) break;
            const char *tok = q;
            while (q < qe && !is_ws(*q)) q++;
            const unsigned long long row = k / per_row, col = k % per_row;
            char *e2 = nullptr;
            if (col == 0) {
                const unsigned long long v = (*tok >= '0' && *tok <= '9') ? strtoull(tok, &e2, 10) : 0;
                if (e2 != q || v < 1 || v > n || seen[(size_t)(v - 1)].exchange(1)) { flag_bad(row); id = 0; }
                else id = v;
            } else {
                const float f = strtof(tok, &e2);
                if (e2 != q) flag_bad(row);
                else if (id) x[(size_t)(id - 1) * dim + (col - 1)] = f;
            }
            k++;
        }
    });
    if (bad_row.load() != ~0ull) {
        free(x);
        return fail(F2V_EIO, "f2v_read_embd: %s: row %llu of %llu is malformed, out of range or repeated", path, bad_row.load() + 1, n);
    }
    *n_out = (uint32_t)n;
    *dim_out = (uint32_t)dim;
    *x_out = x;
    return F2V_OK;
}

// Raw fp32 N x D (f2v_write_embd_bin, the scorers' readBinEmbeddings format): the file must hold exactly n * dim floats.
int f2v_read_embd_bin(const char *path, uint32_t n, uint32_t dim, float *x_out) {
    if (!path || !x_out) return fail(F2V_EINVAL, "f2v_read_embd_bin: null argument");
    FILE *fp = fopen(path, "rb");
    if (!fp) return fail(F2V_EIO, "f2v_read_embd_bin: cannot open %s: %s", path, strerror(errno));
    const size_t total = (size_t)n * dim;
    const bool ok = fread(x_out, sizeof(float), total, fp) == total && fgetc(fp) == EOF;
    fclose(fp);
    return ok ? F2V_OK : fail(F2V_EIO, "f2v_read_embd_bin: %s does not hold exactly %u x %u floats", path, n, dim);
}

static const char kCsrMagic[8] = {'F', '2', 'V', 'C', 'S', 'R', '1', 0};

int f2v_write_csr_bin(const char *path, const uint32_t *rowptr, const uint32_t *colids, uint32_t n, uint64_t nnz) {
    if (!path || !rowptr || (!colids && nnz)) return fail(F2V_EINVAL, "f2v_write_csr_bin: null argument");
    FILE *fp = fopen(path, "wb");
    if (!fp) return fail(F2V_EIO, "f2v_write_csr_bin: cannot open %s: %s", path, strerror(errno));
    const uint32_t hdr[2] = {n, 0};
    bool ok = fwrite(kCsrMagic, 1, 8, fp) == 8 && fwrite(hdr, 4, 2, fp) == 2 && fwrite(&nnz, 8, 1, fp) == 1 &&
              fwrite(rowptr, 4, (size_t)n + 1, fp) == (size_t)n + 1 && (nnz == 0 || fwrite(colids, 4, nnz, fp) == nnz);
    ok = (fclose(fp) == 0) && ok;
    return ok ? F2V_OK : fail(F2V_EIO, "f2v_write_csr_bin: short write to %s", path);
}

int f2v_read_csr_bin(const char *path, uint32_t *n_out, uint64_t *nnz_out, uint32_t **rowptr_out, uint32_t **colids_out) {
    if (!path || !n_out || !nnz_out || !rowptr_out || !colids_out) return fail(F2V_EINVAL, "f2v_read_csr_bin: null argument");
    FILE *fp = fopen(path, "rb");
    if (!fp) return fail(F2V_EIO, "f2v_read_csr_bin: cannot open %s: %s", path, strerror(errno));
    char magic[8];
    uint32_t hdr[2];
    uint64_t nnz = 0;
    if (fread(magic, 1, 8, fp) != 8 || memcmp(magic, kCsrMagic, 8) != 0 || fread(hdr, 4, 2, fp) != 2 || fread(&nnz, 8, 1, fp) != 1) {
        fclose(fp);
        return fail(F2V_EIO, "f2v_read_csr_bin: %s is not an F2VCSR1 file", path);
    }
    const uint32_t n = hdr[0];
    if (nnz >= 0xFFFFFFFFull) { fclose(fp); return fail(F2V_EINVAL, "f2v_read_csr_bin: nnz exceeds 32-bit row pointers"); }
    uint32_t *rowptr = (uint32_t *)malloc(((size_t)n + 1) * 4), *colids = (uint32_t *)malloc((nnz ? nnz : 1) * 4);
    if (!rowptr || !colids) { free(rowptr); free(colids); fclose(fp); return fail(F2V_ENOMEM, "f2v_read_csr_bin: out of memory"); }
    bool ok = fread(rowptr, 4, (size_t)n + 1, fp) == (size_t)n + 1 && (nnz == 0 || fread(colids, 4, nnz, fp) == nnz);
    fclose(fp);
    ok = ok && rowptr[0] == 0 && rowptr[n] == nnz;
    for (uint32_t i = 0; ok && i < n; i++) ok = rowptr[i] <= rowptr[i + 1];
    for (uint64_t k = 0; ok && k < nnz; k++) ok = colids[k] < n;
    if (!ok) { free(rowptr); free(colids); return fail(F2V_EIO, "f2v_read_csr_bin: %s is truncated or inconsistent", path); }
    *n_out = n; *nnz_out = nnz; *rowptr_out = rowptr; *colids_out = colids;
    return F2V_OK;
}

int f2v_write_embd_bin(const char *path, const float *x, uint32_t n, uint32_t dim) {
    if (!path || !x) return fail(F2V_EINVAL, "f2v_write_embd_bin: null argument");
    FILE *fp = fopen(path, "wb");
    if (!fp) return fail(F2V_EIO, "f2v_write_embd_bin: cannot open %s: %s", path, strerror(errno));
    const size_t total = (size_t)n * dim;
    bool ok = fwrite(x, sizeof(float), total, fp) == total;
    ok = (fclose(fp) == 0) && ok;
    return ok ? F2V_OK : fail(F2V_EIO, "f2v_write_embd_bin: short write to %s", path);
}

int f2v_output_name(const char *input, const char *outdir, int option, int bs_mode, uint32_t batch, uint32_t dim,
                    uint32_t iters, uint32_t ns, char *out, size_t out_len) {
    if (!input || !outdir || !out) return fail(F2V_EINVAL, "f2v_output_name: null argument");
    // basename = last '/'-separated token (algorithms.h:119-121)
    std::string in(input), base;
    size_t pos = in.find_last_of('/');
    base = pos == std::string::npos ? in : in.substr(pos + 1);
    const char *tag;
    switch (option) {
        case 5: tag = "F2VNS"; break;             // algorithms.cpp:650 (and :752 for -bs 1)
        case 6: tag = "F2VWNS"; break;            // :930 (:1059)
        case 7: tag = "F2VWNSF"; break;           // :1201
        case 8: tag = "F2VNS_AVXZ"; break;        // :1635
        case 9: tag = dim == 64 ? "F2VWNSLB64_AVXZ" : "F2VWNS_AVXZ"; break;        // :3235 / :2047
        case 10: tag = dim == 64 ? "F2VWEFFNS_AVXZ64" : "F2VWEFFNS_AVXZ"; break;   // :4047 / :2409
        case 11: tag = dim == 64 ? "F2VNSLB_AVXZ64" : "F2VNSLB_AVXZ"; break;       // :3681 / :2860
        default: return fail(F2V_EINVAL, "f2v_output_name: option %d is outside 5..11", option);
    }
    (void)bs_mode;
    int w = snprintf(out, out_len, "%s%s%s%uD%uIT%uNS%u.embd", outdir, base.c_str(), tag, batch, dim, iters, ns);
    if (w < 0 || (size_t)w >= out_len) return fail(F2V_EINVAL, "f2v_output_name: buffer too small");
    return F2V_OK;
}

// Slices of one minibatch for the push exchange: contiguous, balanced by WORK -- a row costs its neighbour gathers
// plus a fixed share (own row in and out, negative samples), weight = degree + 4 -- not by row count: on a power-law
// graph the rank that draws the minibatch's biggest hub would otherwise keep everybody waiting at the barrier
// (RMAT-20, B = 65536, 8 ranks: heaviest slice 1.13x the mean with equal row counts).
int f2v_shard_bounds(const uint32_t *rowptr, uint32_t lo, uint32_t hi, uint32_t world, uint32_t *bounds) {
    if (!rowptr || !bounds || world == 0 || lo > hi) return fail(F2V_EINVAL, "f2v_shard_bounds: bad argument");
    auto W = [&](uint32_t i) -> uint64_t { return (uint64_t)(rowptr[i] - rowptr[lo]) + 4ull * (i - lo); };  // strictly increasing
    const uint64_t total = W(hi);
    bounds[0] = lo;
    for (uint32_t k = 1; k < world; k++) {
        const uint64_t target = (total * k + world - 1) / world;
        uint32_t a = bounds[k - 1], b = hi;  // smallest i in [a, hi] with W(i) >= target
        while (a < b) {
            const uint32_t m = a + (b - a) / 2;
            if (W(m) >= target) b = m; else a = m + 1;
        }
        bounds[k] = a;
    }
    bounds[world] = hi;
    return F2V_OK;
}

// Who reads which row in the sharded run (include/f2v.h: the push exchange).  One pass over the CSR: the owner of
// row u reads every neighbour of u; threads take row ranges and OR their rank's bit in atomically.
int f2v_push_masks(const uint32_t *rowptr, const uint32_t *colids, uint32_t n, uint32_t batch, uint32_t world,
                   const uint32_t *sample_ids, uint64_t n_ids, uint32_t *masks) {
    if (!rowptr || !masks || (!colids && rowptr[n]) || (!sample_ids && n_ids)) return fail(F2V_EINVAL, "f2v_push_masks: null argument");
    if (batch == 0 || world == 0 || world > 32) return fail(F2V_EINVAL, "f2v_push_masks: batch %u / world %u", batch, world);
    const uint32_t nb = (uint32_t)(((uint64_t)n + batch - 1) / batch);
    std::vector<uint32_t> bounds((size_t)nb * (world + 1));
    for (uint32_t b = 0; b < nb; b++)
        (void)f2v_shard_bounds(rowptr, b * batch, (uint32_t)std::min<uint64_t>((uint64_t)b * batch + batch, n), world, bounds.data() + (size_t)b * (world + 1));
    auto owner = [&](uint32_t u) -> uint32_t {
        const uint32_t *bd = bounds.data() + (size_t)(u / batch) * (world + 1);
        return (uint32_t)(std::upper_bound(bd + 1, bd + world, u) - (bd + 1));  // slice k = [bd[k], bd[k+1])
    };
    memset(masks, 0, (size_t)n * sizeof(uint32_t));
    unsigned T = std::thread::hardware_concurrency();
    if (const char *e = getenv("F2V_IO_THREADS")) T = (unsigned)atoi(e);
    T = std::max(1u, std::min(T, 64u));
    if (rowptr[n] < (1u << 20)) T = 1;
    auto work = [&](uint32_t r0, uint32_t r1) {
        for (uint32_t u = r0; u < r1; u++) {
            const uint32_t bit = 1u << owner(u);
            for (uint32_t k = rowptr[u]; k < rowptr[u + 1]; k++) {
                uint32_t *m = masks + colids[k];
                if (!(__atomic_load_n(m, __ATOMIC_RELAXED) & bit)) __atomic_fetch_or(m, bit, __ATOMIC_RELAXED);
            }
        }
    };
    if (T == 1) {
        work(0, n);
    } else {
        std::vector<std::thread> th;
        const uint64_t nnz = rowptr[n];
        uint32_t r0 = 0;
        for (unsigned t = 0; t < T; t++) {  // equal shares of the nonzeros
            const uint64_t target = nnz * (t + 1) / T;
            uint32_t r1 = (t == T - 1) ? n : (uint32_t)(std::upper_bound(rowptr, rowptr + n + 1, (uint32_t)target) - rowptr - 1);
            r1 = std::max(r1, r0);
            if (t == T - 1) r1 = n;
            th.emplace_back(work, r0, r1);
            r0 = r1;
        }
        for (auto &y : th) y.join();
    }
    const uint32_t everyone = world >= 32 ? 0xFFFFFFFFu : ((1u << world) - 1u);
    for (uint64_t k = 0; k < n_ids; k++) {
        if (sample_ids[k] >= n) return fail(F2V_EINVAL, "f2v_push_masks: sample id %u is not a vertex", sample_ids[k]);
        masks[sample_ids[k]] = everyone;
    }
    for (uint32_t v = 0; v < n; v++) masks[v] &= ~(1u << owner(v));  // the owner has the row already
    return F2V_OK;
}


}  // extern "C"

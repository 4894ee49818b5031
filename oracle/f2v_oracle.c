/*
 * f2v_oracle.c -- TEST INFRASTRUCTURE ONLY (parity oracle).
 *
 * A plain-C, single-threaded restatement of the Force2Vec hot path of the reference
 * (HipGraph/Force2Vec, /root/reference): options 5/6/7 and their "-bs 1" variants.
 * Nothing under force2vec_amd/ (the product) includes, links or calls this file; only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the
 * checker.
 *
 * Two arithmetic orders are implemented:
 *   ORC_ORDER_REF  (0): the reference's order as compiled by g++ -O3 -ffast-math on
 *       x86-64: squared-distance / dot product summed SEQUENTIALLY over
 *       d ascending, separate mul+add, fp64 scalar d1.  Pinned bit-for-bit (at the 6
 *       printed digits of the .embd format) against oracle/_ref (the genuine reference
 *       built from /root/reference by oracle/build_ref.sh) -- tests/golden/ holds the
 *       outputs of that binary, tests/test_oracle_golden.py does the comparison.
 *   ORC_ORDER_TREE (1): identical except that the per-pair reduction over d is the
 *       balanced adjacent-pair binary tree over next_pow2(D) zero-padded terms -- the
 *       canonical order of the HIP wavefront reduction (in-lane pairs, then lane xor
 *       1,2,4,8,16,32).  The HIP kernels are bit-exact against this order.
 * and, orthogonally, hub chunking (chunk > 0): a row with more than `chunk` neighbours
 * is cut into pieces of at most `chunk` neighbours -- after every `chunk` neighbours and
 * (orc_set_class_cut, default 8) wherever the ascending neighbour ids cross from one eighth
 * of the id range into the next: piece_cuts below, the engine's rule of the same name;
 * each piece accumulates from zero (the last
 * one also takes the negative samples) and the pieces' partials are combined by a
 * `fanin`-ary tree: consecutive groups of `fanin` partials are added sequentially, level
 * by level, until one is left (fanin == 0: one sequential pass over all partials).
 * chunk == 0 is the reference's single sequential accumulation.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (NO -ffast-math).  See oracle/Makefile.
 *
 * Reference citations are file:line under /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_ORDER_REF 0
#define ORC_ORDER_TREE 1

#define ORC_MAXBOUND 5.0f        /* sample/algorithms.h:41  MAXBOUND */
#define ORC_SM_TABLE_SIZE 2048   /* sample/algorithms.h:43 */
#define ORC_SM_BOUND 6.0         /* sample/algorithms.h:44 */
#define ORC_WALKLENGTH 5         /* sample/algorithms.cpp:1073 */
#define ORC_MAXDIM 4096

/* ------------------------------------------------------------------------------------
 * libc rand(): glibc TYPE_3 additive feedback generator (r[i] = r[i-3] + r[i-31],
 * output >> 1), seeded as srandom_r does.  The reference calls srand(1)
 * (Test/Force2Vec.cpp:126) and then only rand() (sample/algorithms.cpp:41,50,56).
 * glibc is not part of /root/reference; this restates stdlib/random_r.c of glibc 2.35
 * and is checked against the C library's own rand() in tests/test_rng.py.
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int32_t r[31];
    int f, b; /* front / rear indices */
} orc_rng;

void orc_srand(orc_rng *g, unsigned int seed) {
    if (seed == 0) seed = 1;
    g->r[0] = (int32_t)seed;
    for (int i = 1; i < 31; i++) {
        long hi = g->r[i - 1] / 127773;
        long lo = g->r[i - 1] % 127773;
        long word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        g->r[i] = (int32_t)word;
    }
    g->f = 3;
    g->b = 0;
    for (int i = 0; i < 310; i++) {
        g->r[g->f] = (int32_t)((uint32_t)g->r[g->f] + (uint32_t)g->r[g->b]);
        if (++g->f >= 31) g->f = 0;
        if (++g->b >= 31) g->b = 0;
    }
}

int orc_rand(orc_rng *g) {
    uint32_t v = (uint32_t)g->r[g->f] + (uint32_t)g->r[g->b];
    g->r[g->f] = (int32_t)v;
    if (++g->f >= 31) g->f = 0;
    if (++g->b >= 31) g->b = 0;
    return (int)(v >> 1);
}

orc_rng *orc_rng_new(unsigned int seed) {
    orc_rng *g = (orc_rng *)malloc(sizeof(orc_rng));
    orc_srand(g, seed);
    return g;
}
void orc_rng_free(orc_rng *g) { free(g); }

/* randIndex(max,min), sample/algorithms.cpp:55-58 */
static uint32_t rand_index(orc_rng *g, uint32_t max_num, uint32_t min_num) {
    return ((uint32_t)orc_rand(g) % (max_num - min_num)) + min_num;
}
uint32_t orc_rand_index(orc_rng *g, uint32_t max_num, uint32_t min_num) {
    return rand_index(g, max_num, min_num);
}

/* randInitF (kind 0, sample/algorithms.cpp:47-53): X = -1 + 2*rand()/(RAND_MAX+1.0)
 * randInit  (kind 1, sample/algorithms.cpp:38-45): X = rand()/(RAND_MAX+1.0)
 * double arithmetic, narrowed to float on store. */
void orc_init_embeddings(orc_rng *g, float *X, uint32_t n, uint32_t d, int kind) {
    size_t total = (size_t)n * d;
    for (size_t k = 0; k < total; k++) {
        double r = (double)orc_rand(g);
        if (kind == 0)
            X[k] = (float)(-1.0 + 2.0 * r / 2147483648.0);
        else
            X[k] = (float)(r / 2147483648.0);
    }
}

/* ------------------------------------------------------------------------------------
 * MatrixMarket reader + CSR (sample/IO.h:59-156 ReadASCII, sample/CSC.h:146-188,
 * sample/CSR.h:154-186): header lines start with '%', "symmetric" anywhere in a header
 * line mirrors every off-diagonal entry and DROPS self-loops (IO.h:122-134; a general
 * matrix keeps them); duplicates are kept; the CSC sort + transpose leaves colids
 * ascending inside each row.  Values are parsed but never read by options 5-11.
 * Returns 0 on success.  Caller frees with orc_free().
 * ---------------------------------------------------------------------------------- */
static int cmp_pair(const void *a, const void *b) {
    const uint32_t *x = (const uint32_t *)a, *y = (const uint32_t *)b;
    if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
    if (x[1] != y[1]) return x[1] < y[1] ? -1 : 1;
    return 0;
}

int orc_read_mtx(const char *path, uint32_t *n_out, uint64_t *nnz_out, uint32_t **rowptr_out,
                 uint32_t **colids_out) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    char line[512];
    int symmetric = 0;
    long m = 0, n = 0, nnz = 0;
    for (;;) {
        if (!fgets(line, sizeof line, f)) { fclose(f); return -2; }
        if (line[0] == '%') {
            if (strstr(line, "symmetric")) symmetric = 1;
            continue;
        }
        if (sscanf(line, "%ld %ld %ld", &m, &n, &nnz) != 3) { fclose(f); return -3; }
        break;
    }
    size_t cap = (size_t)nnz * (symmetric ? 2 : 1);
    uint32_t *pairs = (uint32_t *)malloc((cap ? cap : 1) * 2 * sizeof(uint32_t));
    size_t cnt = 0;
    for (long k = 0; k < nnz; k++) {
        if (!fgets(line, sizeof line, f)) break;
        long r = atol(line);
        char *sp = strchr(line, ' ');
        if (!sp) break;
        long c = atol(sp + 1);
        r--; c--;
        if (symmetric) {
            if (r == c) continue; /* IO.h:131-134 */
            pairs[2 * cnt] = (uint32_t)r; pairs[2 * cnt + 1] = (uint32_t)c; cnt++;
            pairs[2 * cnt] = (uint32_t)c; pairs[2 * cnt + 1] = (uint32_t)r; cnt++;
        } else {
            pairs[2 * cnt] = (uint32_t)r; pairs[2 * cnt + 1] = (uint32_t)c; cnt++;
        }
    }
    fclose(f);
    qsort(pairs, cnt, 2 * sizeof(uint32_t), cmp_pair);
    uint32_t *rowptr = (uint32_t *)calloc((size_t)m + 1, sizeof(uint32_t));
    uint32_t *colids = (uint32_t *)malloc((cnt ? cnt : 1) * sizeof(uint32_t));
    for (size_t k = 0; k < cnt; k++) {
        rowptr[pairs[2 * k] + 1]++;
        colids[k] = pairs[2 * k + 1];
    }
    for (long i = 0; i < m; i++) rowptr[i + 1] += rowptr[i];
    free(pairs);
    *n_out = (uint32_t)m;
    *nnz_out = cnt;
    *rowptr_out = rowptr;
    *colids_out = colids;
    return 0;
}
void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------------------------------
 * Scalars
 * ---------------------------------------------------------------------------------- */
/* scale(), sample/algorithms.cpp:6-10, as compiled with -ffast-math: maxss then minss,
 * so a NaN (0 * inf when a negative sample is the row's own vertex) becomes -5. */
static inline float scale_ref(float v) {
    float t = (v > -ORC_MAXBOUND) ? v : -ORC_MAXBOUND;
    return (t < ORC_MAXBOUND) ? t : ORC_MAXBOUND;
}

/* init_SM_TABLE, sample/algorithms.cpp:757-764: float x, float exp, 1.0/(float) in
 * double, narrowed. */
void orc_sm_table(float *table) {
    for (int i = 0; i < ORC_SM_TABLE_SIZE; i++) {
        float x = (float)(2.0 * ORC_SM_BOUND * i / ORC_SM_TABLE_SIZE - ORC_SM_BOUND);
        table[i] = (float)(1.0 / (double)(1.0f + expf(-x)));
    }
}

/* init_SM_TABLE AS COMPILED by g++ 11.4 -O3 -ffast-math on x86-64 (disassembly of the
 * oracle/_ref binary): entries 0 and 2045..2047 are compile-time constants equal to the
 * source-level values above; entries 1..2044 are computed four at a time as
 *   v = _ZGVbN4v_expf(float(6.0 - i*12/2048)) + 1.0f;  r = rcpps(v);  out = (r+r) - (v*r)*r
 * i.e. libmvec's vector expf and the hardware reciprocal ESTIMATE plus one Newton step.
 * rcpps is not specified bit-for-bit across CPU vendors, so the reference's option 6/7
 * results depend on the host CPU at the 1e-7 level per table entry.  This variant exists
 * only so that the rest of the option 6/7 restatement can be pinned bit-for-bit against
 * oracle/_ref ON THE SAME HOST; the product and the oracle's default use orc_sm_table(). */
#if defined(__x86_64__)
#include <xmmintrin.h>
extern __m128 _ZGVbN4v_expf(__m128);
int orc_sm_table_as_compiled(float *table) {
    orc_sm_table(table);
    for (int i = 1; i <= 2041; i += 4) {
        float nx[4];
        for (int k = 0; k < 4; k++) nx[k] = (float)(6.0 - (double)(i + k) * 0.005859375);
        __m128 v = _mm_add_ps(_ZGVbN4v_expf(_mm_loadu_ps(nx)), _mm_set1_ps(1.0f));
        __m128 r = _mm_rcp_ps(v);
        __m128 t = _mm_mul_ps(_mm_mul_ps(v, r), r);
        _mm_storeu_ps(table + i, _mm_sub_ps(_mm_add_ps(r, r), t));
    }
    return 0;
}
#else
int orc_sm_table_as_compiled(float *table) { orc_sm_table(table); return -1; }
#endif

/* Test hook: the sigmoid table every option-6/7 routine below uses (default: orc_sm_table). */
static float g_table[ORC_SM_TABLE_SIZE];
static int g_table_ready = 0;
void orc_set_sm_table(const float *table) {
    if (table) { memcpy(g_table, table, sizeof g_table); g_table_ready = 1; }
    else { orc_sm_table(g_table); g_table_ready = 1; }
}
static const float *current_table(void) {
    if (!g_table_ready) orc_set_sm_table(NULL);
    return g_table;
}

/* fast_SM, sample/algorithms.cpp:766-770; SM_RESOLUTION is a float constant
 * (algorithms.h:49).  v == 6.0 indexes one past the table in the reference; the
 * restatement clamps that one case to the last entry. */
static inline float fast_sm(const float *table, float v) {
    const float res = (float)(ORC_SM_TABLE_SIZE / (2.0 * ORC_SM_BOUND));
    if (v > (float)ORC_SM_BOUND) return 1.0f;
    if (v < -(float)ORC_SM_BOUND) return 0.0f;
    int idx = (int)(((double)v + ORC_SM_BOUND) * (double)res);
    if (idx > ORC_SM_TABLE_SIZE - 1) idx = ORC_SM_TABLE_SIZE - 1;
    if (idx < 0) idx = 0;
    return table[idx];
}
float orc_fast_sm(const float *table, float v) { return fast_sm(table, v); }

static float reduce_terms(float *t, uint32_t D, int order) {
    if (order == ORC_ORDER_REF) {
        float a = 0.0f;
        for (uint32_t d = 0; d < D; d++) a += t[d];
        return a;
    }
    uint32_t P = 1;
    while (P < D) P <<= 1;
    for (uint32_t d = D; d < P; d++) t[d] = 0.0f;
    while (P > 1) {
        P >>= 1;
        for (uint32_t k = 0; k < P; k++) t[k] = t[2 * k] + t[2 * k + 1];
    }
    return t[0];
}

/* Hub combine fan-in (see the header); set through orc_set_fanin, default 32. */
static uint32_t g_fanin = 32;
void orc_set_fanin(uint32_t fanin) { g_fanin = fanin; }

/* Where a row of more than `chunk` neighbours is cut into pieces (the engine's piece_cuts, force2vec_amd/csrc/f2v_engine.hip):
 * after every `chunk` neighbours and -- unless switched off with orc_set_class_cut(0) -- wherever the ascending neighbour ids
 * cross from one of g_classes equal parts of the id range [0, n) into the next.  Returns the number of pieces; cuts[0..pieces]
 * are the offsets of the pieces' first neighbours and, last, deg (cuts has room for deg + 1 entries). */
static uint32_t g_classes = 8;
void orc_set_class_cut(uint32_t classes) { g_classes = classes; }

static uint32_t piece_cuts(const uint32_t *nbrs, uint32_t deg, uint32_t chunk, uint32_t n, uint32_t *cuts) {
    uint32_t np = 0;
    if (chunk == 0 || deg <= chunk) {
        cuts[0] = 0;
        cuts[1] = deg;
        return 1;
    }
    if (g_classes == 0 || n == 0) {
        for (uint32_t b = 0; b < deg; b += chunk) cuts[np++] = b;
        cuts[np] = deg;
        return np;
    }
    uint32_t start = 0, cls = (uint32_t)(((uint64_t)nbrs[0] * g_classes) / n);
    cuts[np++] = 0;
    for (uint32_t e = 1; e < deg; e++) {
        uint32_t ce = (uint32_t)(((uint64_t)nbrs[e] * g_classes) / n);
        if (ce != cls || e - start == chunk) {
            cuts[np++] = e;
            start = e;
            cls = ce;
        }
    }
    cuts[np] = deg;
    return np;
}

/* parts: n partial vectors of D floats, contiguous; combined in place into parts[0..D). */
static void combine_partials(float *parts, uint32_t n, uint32_t D) {
    uint32_t G = g_fanin ? g_fanin : n;
    if (G < 2) G = 2;
    while (n > 1) {
        uint32_t nout = (n + G - 1) / G;
        for (uint32_t o = 0; o < nout; o++) {
            uint32_t lo = o * G, hi = lo + G < n ? lo + G : n;
            float *acc = parts + (size_t)o * D; /* o <= lo: writing group o never clobbers unread input */
            if (o != lo) memcpy(acc, parts + (size_t)lo * D, D * sizeof(float));
            for (uint32_t k = lo + 1; k < hi; k++)
                for (uint32_t d = 0; d < D; d++) acc[d] = acc[d] + parts[(size_t)k * D + d];
        }
        n = nout;
    }
}

/* ------------------------------------------------------------------------------------
 * One row of one minibatch.  X is the embedding matrix BEFORE the minibatch (Jacobi
 * inside a batch), S[k] points at the k-th negative-sample row as it was before the
 * minibatch (the reference copies them first: algorithms.cpp:577-586).
 * `nbrs`/`deg` is the neighbour list the forces run over (CSR row for options 5/6,
 * the 5 walk samples for option 7); `graph_deg` is the CSR degree (for degi).
 * Writes the row's NEW embedding to out[0..D).
 * ---------------------------------------------------------------------------------- */
static void tdist_accumulate(const float *xi, const float *xj, uint32_t D, float lr, int order,
                             int negative, float *Y) {
    float diff[ORC_MAXDIM], t[ORC_MAXDIM];
    for (uint32_t d = 0; d < D; d++) {
        diff[d] = xi[d] - xj[d];
        t[d] = diff[d] * diff[d];
    }
    float a = reduce_terms(t, D, order);
    float d1;
    if (!negative)
        d1 = (float)(-2.0 / (1.0 + (double)a)); /* algorithms.cpp:608 */
    else
        d1 = (float)(2.0 / ((double)a * (1.0 + (double)a))); /* algorithms.cpp:622 */
    for (uint32_t d = 0; d < D; d++) {
        float f = scale_ref(diff[d] * d1);
        float s = lr * f;
        Y[d] = Y[d] + s; /* algorithms.cpp:610-611 / 624-625 */
    }
}

/* option 5 row: sample/algorithms.cpp:588-639 */
static void row_tdist(const float *X, uint32_t D, uint32_t n, uint32_t i, const uint32_t *nbrs, uint32_t deg,
                      const float *const *S, uint32_t ns, float lr, int order, uint32_t chunk,
                      float *out) {
    const float *xi = X + (size_t)i * D;
    uint32_t *cuts = (uint32_t *)malloc(((size_t)deg + 2) * sizeof(uint32_t));
    uint32_t nchunks = piece_cuts(nbrs, deg, chunk, n, cuts);
    float *parts = (float *)malloc((size_t)nchunks * D * sizeof(float));
    for (uint32_t c = 0; c < nchunks; c++) {
        uint32_t lo = cuts[c], hi = cuts[c + 1];
        float *P = parts + (size_t)c * D;
        for (uint32_t d = 0; d < D; d++) P[d] = 0.0f;
        for (uint32_t k = lo; k < hi; k++)
            tdist_accumulate(xi, X + (size_t)nbrs[k] * D, D, lr, order, 0, P);
        if (c == nchunks - 1)
            for (uint32_t s = 0; s < ns; s++) tdist_accumulate(xi, S[s], D, lr, order, 1, P);
    }
    combine_partials(parts, nchunks, D);
    for (uint32_t d = 0; d < D; d++) out[d] = xi[d] + parts[d]; /* algorithms.cpp:636 */
    free(parts);
    free(cuts);
}

/* options 6/7 row: sample/algorithms.cpp:833-921 (6), 1142-1193 (7).
 * The accumulator starts as a COPY of x_i (algorithms.cpp:824-831) and replaces x_i. */
static void row_sigmoid(const float *X, uint32_t D, uint32_t n, uint32_t i, const uint32_t *nbrs, uint32_t deg,
                        uint32_t graph_deg, const float *const *S, uint32_t ns, float lr,
                        const float *table, int order, uint32_t chunk, float *out) {
    const float *xi = X + (size_t)i * D;
    float t[ORC_MAXDIM];
    float degi = (float)(1.0 / (double)(graph_deg + 1u)); /* algorithms.cpp:854 */
    double c0 = (double)(lr * degi);
    uint32_t *cuts = (uint32_t *)malloc(((size_t)deg + 2) * sizeof(uint32_t));
    uint32_t nchunks = piece_cuts(nbrs, deg, chunk, n, cuts);
    float *parts = (float *)malloc((size_t)nchunks * D * sizeof(float));
    for (uint32_t c = 0; c < nchunks; c++) {
        uint32_t lo = cuts[c], hi = cuts[c + 1];
        float *P = parts + (size_t)c * D;
        /* chunk 0 starts from x_i (the reference's copy-in); later chunks from zero */
        for (uint32_t d = 0; d < D; d++) P[d] = (c == 0) ? xi[d] : 0.0f;
        for (uint32_t k = lo; k < hi; k++) {
            const float *xj = X + (size_t)nbrs[k] * D;
            for (uint32_t d = 0; d < D; d++) t[d] = xi[d] * xj[d];
            float a = reduce_terms(t, D, order);
            float sm = fast_sm(table, a);
            double coef = (1.0 - (double)sm) * c0; /* algorithms.cpp:867 */
            for (uint32_t d = 0; d < D; d++) P[d] = (float)((double)xj[d] * coef + (double)P[d]);
        }
        if (c == nchunks - 1) {
            for (uint32_t s = 0; s < ns; s++) {
                const float *sj = S[s];
                for (uint32_t d = 0; d < D; d++) t[d] = xi[d] * sj[d];
                float r = reduce_terms(t, D, order);
                float sm = fast_sm(table, r);
                float w = lr * sm; /* algorithms.cpp:907 */
                for (uint32_t d = 0; d < D; d++) {
                    float p = w * sj[d];
                    P[d] = P[d] - p;
                }
            }
        }
    }
    combine_partials(parts, nchunks, D);
    memcpy(out, parts, D * sizeof(float)); /* algorithms.cpp:918 */
    free(parts);
    free(cuts);
}

/* ------------------------------------------------------------------------------------
 * One minibatch [lo,hi) updated IN PLACE with the reference's Jacobi-within-batch
 * semantics (forces from the pre-batch X, all rows committed afterwards).
 *   option    5 | 6 | 7
 *   bs_mode   0: all rows use sample_ids[0..ns);  1: row i uses sample_ids[i-lo .. i-lo+ns)
 *             (algorithms.cpp:719-720, 1030-1031)
 *   walks     option 7 only: uint32[5*N] walk samples of this epoch
 * Rows outside [row_lo,row_hi) are left untouched (multi-GPU shard check).
 * ---------------------------------------------------------------------------------- */
int orc_minibatch(int option, int bs_mode, const uint32_t *rowptr, const uint32_t *colids,
                  uint32_t n, uint32_t D, float *X, uint32_t lo, uint32_t hi, uint32_t row_lo,
                  uint32_t row_hi, const uint32_t *sample_ids, uint32_t ns, float lr,
                  const uint32_t *walks, int order, uint32_t chunk) {
    if (D > ORC_MAXDIM || hi > n || lo > hi) return -1;
    if (row_lo < lo) row_lo = lo;
    if (row_hi > hi) row_hi = hi;
    uint32_t nsid = bs_mode ? (hi - lo) + ns - 1 : ns;
    if (hi == lo) return 0;
    /* snapshot of the sample rows before the batch */
    float *snap = (float *)malloc((size_t)(nsid ? nsid : 1) * D * sizeof(float));
    for (uint32_t s = 0; s < nsid; s++)
        memcpy(snap + (size_t)s * D, X + (size_t)sample_ids[s] * D, D * sizeof(float));
    float *newrows = (float *)malloc((size_t)(hi - lo) * D * sizeof(float));
    const float **S = (const float **)malloc((ns ? ns : 1) * sizeof(float *));
    const float *table = current_table();
    for (uint32_t i = row_lo; i < row_hi; i++) {
        uint32_t base = bs_mode ? (i - lo) : 0;
        for (uint32_t s = 0; s < ns; s++) S[s] = snap + (size_t)(base + s) * D;
        uint32_t gdeg = rowptr[i + 1] - rowptr[i];
        float *out = newrows + (size_t)(i - lo) * D;
        if (option == 5)
            row_tdist(X, D, n, i, colids + rowptr[i], gdeg, S, ns, lr, order, chunk, out);
        else if (option == 6)
            row_sigmoid(X, D, n, i, colids + rowptr[i], gdeg, gdeg, S, ns, lr, table, order, chunk, out);
        else if (option == 7 || option == 10)
            /* option 10 (AlgoForce2VecNSRWEFF_SREAL_D128/D64_AVXZ) never divides by the degree: `degi = 1.0` is all it has
             * (algorithms.cpp:2155, :2345, :3793, :3983) where option 7 sets 1/(deg+1) (:1159) -- a "degree" of 0 here */
            row_sigmoid(X, D, n, i, walks + (size_t)i * ORC_WALKLENGTH, ORC_WALKLENGTH, option == 10 ? 0u : gdeg, S, ns, lr,
                        table, order, 0 /* the 5 walk samples are never split */, out);
        else { free(snap); free(newrows); free((void *)S); return -2; }
    }
    for (uint32_t i = row_lo; i < row_hi; i++)
        memcpy(X + (size_t)i * D, newrows + (size_t)(i - lo) * D, D * sizeof(float));
    free(snap); free(newrows); free((void *)S);
    return 0;
}

/* Walk generation, option 7, once per epoch: sample/algorithms.cpp:1097-1118.
 * deg>2: colids[rowptr[w] + rand()%(deg-1)] (never the last neighbour); deg==2: the
 * first neighbour; otherwise colids[w] -- the VERTEX id used as an edge index, as the
 * reference does (reads past the array when w >= nnz there; clamped here). */
void orc_generate_walks(orc_rng *g, const uint32_t *rowptr, const uint32_t *colids, uint32_t n,
                        uint64_t nnz, uint32_t *walks) {
    for (uint32_t i = 0; i < n; i++) {
        uint32_t w = i;
        for (int step = 0; step < ORC_WALKLENGTH; step++) {
            uint32_t j = w;
            uint32_t deg = rowptr[w + 1] - rowptr[w];
            if (deg > 2)
                j = rand_index(g, rowptr[w + 1] - 1, rowptr[w]);
            else if (deg == 2)
                j = rowptr[w];
            if ((uint64_t)j >= nnz) j = (uint32_t)(nnz ? nnz - 1 : 0);
            walks[(size_t)i * ORC_WALKLENGTH + step] = colids[j];
            w = colids[j];
        }
    }
}

/* Number of sample ids one minibatch consumes from rand(): ns, or ns*BATCH in -bs 1 mode
 * (algorithms.cpp:686, 966) of which only the first rows+ns-1 are ever read. */

/* ------------------------------------------------------------------------------------
 * Whole training runs: AlgoForce2VecNS / NSBS (option 5, algorithms.cpp:544-753),
 * NSRW / NSRWBS (option 6, :778-1060), NSRWEFF (option 7, :1063-1203).
 * The caller has seeded `g` (srand(1) in the reference, Test/Force2Vec.cpp:126); this
 * draws the initial embeddings, then per epoch/minibatch the sample ids, exactly in the
 * reference's rand() order.  X (N*D floats) receives the final embeddings.
 * If sample_log != NULL it receives every drawn sample id in order (test hook).
 * ---------------------------------------------------------------------------------- */
int orc_train(int option, int bs_mode, const uint32_t *rowptr, const uint32_t *colids, uint32_t n,
              uint64_t nnz, uint32_t D, float *X, orc_rng *g, uint32_t iters, uint32_t batch,
              uint32_t ns, float lr, int order, uint32_t chunk, int do_init) {
    if (n < 2 || batch == 0) return -1;
    /* The AVX512 twins (Test/Force2Vec.cpp:152-183) run the maths of options 5/6/7; what differs in their sources besides the
     * arithmetic details (rcp14, FMA, 4 partial dot sums) is option 9's negative-sample range and option 10's missing degree
     * normalisation (orc_minibatch), kept here as `cli_option`. */
    const int cli_option = option;
    if (option == 8 || option == 11) option = 5;
    else if (option == 9) option = 6;
    else if (option == 10) option = 7;
    if (option == 7 && bs_mode) return -2;
    if (do_init) orc_init_embeddings(g, X, n, D, option == 5 ? 0 : 1);
    uint32_t nb = (n + batch - 1) / batch;
    uint32_t ndraw = bs_mode ? ns * batch : ns;
    uint32_t *ids = (uint32_t *)malloc((size_t)(ndraw ? ndraw : 1) * sizeof(uint32_t));
    uint32_t *walks = NULL;
    if (option == 7) walks = (uint32_t *)malloc((size_t)n * ORC_WALKLENGTH * sizeof(uint32_t));
    int rc = 0;
    for (uint32_t it = 0; it < iters && rc == 0; it++) {
        if (option == 7) orc_generate_walks(g, rowptr, colids, n, nnz, walks);
        for (uint32_t b = 0; b < nb && rc == 0; b++) {
            uint32_t lo = b * batch;
            uint32_t hi = lo + batch < n ? lo + batch : n;
            uint32_t maxv = n - 1;
            if (option == 7) { /* algorithms.cpp:1125 (option 10: :2124) */
                uint64_t e = (uint64_t)(b + 1) * batch;
                if (e < maxv) maxv = (uint32_t)e;
            }
            /* option 9 (AlgoForce2VecNSRW_SREAL_D128_AVXZ): full minibatches draw from [0, (b+1)*BATCHSIZE) -- algorithms.cpp:1700-1704,
             * which reaches vertex N-1 when the batch size divides N -- the tail minibatch from [0, N-1) (:1939-1941) */
            if (cli_option == 9 && b < n / batch) maxv = (b + 1) * batch;
            for (uint32_t s = 0; s < ndraw; s++) ids[s] = rand_index(g, maxv, 0);
            rc = orc_minibatch(cli_option == 10 ? 10 : option, bs_mode, rowptr, colids, n, D, X, lo, hi, lo, hi, ids, ns, lr,
                               walks, order, chunk);
        }
    }
    free(ids);
    free(walks);
    return rc;
}

/* New embedding of ONE row given the pre-batch X and explicit sample ids (full-size
 * sampled-row checks: the test downloads X before a step and compares chosen rows). */
int orc_row(int option, const uint32_t *rowptr, const uint32_t *colids, uint32_t n, uint32_t D, const float *X,
            uint32_t i, const uint32_t *sample_ids, uint32_t ns, float lr, const uint32_t *walks,
            int order, uint32_t chunk, float *out) {
    if (D > ORC_MAXDIM) return -1;
    const float **S = (const float **)malloc((ns ? ns : 1) * sizeof(float *));
    for (uint32_t s = 0; s < ns; s++) S[s] = X + (size_t)sample_ids[s] * D;
    const float *table = current_table();
    uint32_t gdeg = rowptr[i + 1] - rowptr[i];
    int rc = 0;
    if (option == 5)
        row_tdist(X, D, n, i, colids + rowptr[i], gdeg, S, ns, lr, order, chunk, out);
    else if (option == 6)
        row_sigmoid(X, D, n, i, colids + rowptr[i], gdeg, gdeg, S, ns, lr, table, order, chunk, out);
    else if (option == 7 || option == 10)
        row_sigmoid(X, D, n, i, walks + (size_t)i * ORC_WALKLENGTH, ORC_WALKLENGTH, option == 10 ? 0u : gdeg, S, ns, lr, table,
                    order, 0, out);
    else
        rc = -2;
    free((void *)S);
    return rc;
}

/* writeToFile, sample/algorithms.h:118-136: "<N> <D>\n" then "<i+1> v0 v1 ... \n" with
 * ostream's default float format (%g, 6 significant digits) and a trailing space. */
int orc_write_embd(const char *path, const float *X, uint32_t n, uint32_t D) {
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    fprintf(f, "%u %u\n", n, D);
    for (uint32_t i = 0; i < n; i++) {
        fprintf(f, "%u ", i + 1);
        for (uint32_t d = 0; d < D; d++) fprintf(f, "%g ", (double)X[(size_t)i * D + d]);
        fputc('\n', f);
    }
    fclose(f);
    return 0;
}

// algorithms.hpp -- host-side mirror of the reference's `class algorithms`
// (sample/algorithms.h:51-137) over the C ABI of libf2v: same constructor arguments, same
// method names and argument meaning for the options 5-11 entry points, same side effects
// (the "... Wall time required:" line, the .embd file, result = {seconds}).  The embedding
// matrix lives in HBM; nothing here computes forces.
#ifndef F2V_ALGORITHMS_HPP_
#define F2V_ALGORITHMS_HPP_
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "f2v.h"

#define VALUETYPE float
#define INDEXTYPE unsigned int

namespace f2v_host {

struct CSRGraph {  // CSR<INDEXTYPE, VALUETYPE> of sample/CSR.h:89-96 (values are never read by options 5-11)
    INDEXTYPE rows = 0;
    uint64_t nnz = 0;
    INDEXTYPE *rowptr = nullptr;
    INDEXTYPE *colids = nullptr;
    ~CSRGraph() { f2v_free(rowptr); f2v_free(colids); }
    CSRGraph() = default;
    CSRGraph(const CSRGraph &) = delete;
    CSRGraph &operator=(const CSRGraph &) = delete;
};

// SetInputMatricesAsCSR, sample/commonutility.h:44-54.  `use_cache`: keep / reuse "<input>.f2vcsr", the binary
// CSR of the parsed file (the same arrays, so training is bit-identical); an input that already ends in
// ".f2vcsr" is read as such.
inline void SetInputMatricesAsCSR(CSRGraph &A, const std::string &inputfile, bool use_cache = false) {
    const std::string ext = ".f2vcsr";
    const bool is_bin = inputfile.size() > ext.size() && inputfile.compare(inputfile.size() - ext.size(), ext.size(), ext) == 0;
    const std::string cache = is_bin ? inputfile : inputfile + ext;
    if (is_bin || use_cache) {
        if (f2v_read_csr_bin(cache.c_str(), &A.rows, &A.nnz, &A.rowptr, &A.colids) == F2V_OK) {
            std::cout << "Reading binary CSR cache:" << cache << std::endl;
            std::cout << "Input Matrix: Rows = " << A.rows << ", nnz = " << A.nnz << std::endl;
            return;
        }
        if (is_bin) throw std::runtime_error(f2v_last_error());
    }
    std::cout << "Reading input matrices in text (ascii)... " << std::endl;
    std::cout << "Input File Directory:" << inputfile << std::endl;
    if (f2v_read_mtx(inputfile.c_str(), &A.rows, &A.nnz, &A.rowptr, &A.colids) != F2V_OK) throw std::runtime_error(f2v_last_error());
    std::cout << "Input Matrix: Rows = " << A.rows << ", nnz = " << A.nnz << std::endl;
    if (use_cache && f2v_write_csr_bin(cache.c_str(), A.rowptr, A.colids, A.rows, A.nnz) != F2V_OK)
        std::cerr << "warning: " << f2v_last_error() << std::endl;
}

class algorithms {
   public:
    f2v_handle h = nullptr;
    INDEXTYPE DIM, rows;
    std::string filename, outputdir;
    double gpu_train_seconds = 0.0;  // device time of the epoch loop alone
    bool binary_output = false;      // also write "<name>.bin": raw fp32 N x D (readBinEmbeddings format)
    bool text_output = true;         // the reference's text .embd (19 GB at 16 M x 128: switch off with -notext 1)
    std::string init_path;           // warm start (-init): a text .embd or, ending in ".bin", a raw fp32 N x D file, instead of randInit
    f2v_stats stats{};
    int rank = 0, world = 1;         // > 1 after join_ranks: one process per GPU, minibatch rows sharded (f2v_train_sharded)

    algorithms(CSRGraph &A_csr, std::string input, std::string outputd, INDEXTYPE dim, VALUETYPE /*gamma*/, INDEXTYPE /*bsize*/, int device = 0)
        : DIM(dim), rows(A_csr.rows), filename(input), outputdir(outputd) {
        if (f2v_create(A_csr.rowptr, A_csr.colids, A_csr.rows, A_csr.nnz, dim, device, &h) != F2V_OK) throw std::runtime_error(f2v_last_error());
    }
    ~algorithms() { f2v_destroy(h); }
    algorithms(const algorithms &) = delete;

    void srand(unsigned seed) {  // Test/Force2Vec.cpp:126
        check(f2v_srand(h, seed));
        last_seed = seed;
        seeded = true;
    }

    // options 5 / 5 -bs 1 / 6 / 6 -bs 1 / 7 (sample/algorithms.h:86-91)
    std::vector<VALUETYPE> AlgoForce2VecNS(INDEXTYPE IT, INDEXTYPE TH, INDEXTYPE B, INDEXTYPE ns, VALUETYPE lr) { return run(5, 0, IT, B, ns, lr, "Force2Vec Parallel Wall time required:"); }
    std::vector<VALUETYPE> AlgoForce2VecNSBS(INDEXTYPE IT, INDEXTYPE TH, INDEXTYPE B, INDEXTYPE ns, VALUETYPE lr) { return run(5, 1, IT, B, ns, lr, "Force2Vec Parallel Wall time required (with BS negative samples):"); }
    std::vector<VALUETYPE> AlgoForce2VecNSRW(INDEXTYPE IT, INDEXTYPE TH, INDEXTYPE B, INDEXTYPE ns, VALUETYPE lr) { return run(6, 0, IT, B, ns, lr, "Force2Vec Parallel Wall time required:"); }
    std::vector<VALUETYPE> AlgoForce2VecNSRWBS(INDEXTYPE IT, INDEXTYPE TH, INDEXTYPE B, INDEXTYPE ns, VALUETYPE lr) { return run(6, 1, IT, B, ns, lr, "Force2Vec Parallel Wall time required (with BS negative samples):"); }
    std::vector<VALUETYPE> AlgoForce2VecNSRWEFF(INDEXTYPE IT, INDEXTYPE TH, INDEXTYPE B, INDEXTYPE ns, VALUETYPE lr) { return run(7, 0, IT, B, ns, lr, "Force2VecWNSEFF Parallel Wall time required:"); }
    // the AVX512 entry points (sample/algorithms.h:93-102): same maths on the GPU, hub rows load-balanced
    std::vector<VALUETYPE> AlgoForce2VecNS_SREAL_D128_AVXZ(INDEXTYPE IT, INDEXTYPE TH, INDEXTYPE B, INDEXTYPE ns, VALUETYPE lr) { return run(8, 0, IT, B, ns, lr, "Force2Vec Parallel Wall time required:"); }
    std::vector<VALUETYPE> AlgoForce2VecNSRW_SREAL_D128_AVXZ(INDEXTYPE IT, INDEXTYPE TH, INDEXTYPE B, INDEXTYPE ns, VALUETYPE lr) { return run(9, 0, IT, B, ns, lr, "Force2Vec Parallel Wall time required:"); }
    std::vector<VALUETYPE> AlgoForce2VecNSRWEFF_SREAL_D128_AVXZ(INDEXTYPE IT, INDEXTYPE TH, INDEXTYPE B, INDEXTYPE ns, VALUETYPE lr) { return run(10, 0, IT, B, ns, lr, "Force2VecWNSEFF Parallel Wall time required:"); }
    std::vector<VALUETYPE> AlgoForce2VecNSLB_SREAL_D128_AVXZ(INDEXTYPE IT, INDEXTYPE TH, INDEXTYPE B, INDEXTYPE ns, VALUETYPE lr) { return run(11, 0, IT, B, ns, lr, "Force2Vec Parallel Wall time required:"); }

    // Multi-GPU: this process is rank `r` of `w` (one per GPU, all constructed on the same graph, seeded alike).
    // The ranks swap the IPC handles of their matrices through files in `dir` (any directory they all see), map each
    // other (f2v_push_attach) and run the self-test together; the option methods then train sharded and only
    // rank 0 reports and writes the embedding.  No MPI, no torch: the exchange itself runs inside libf2v.
    void join_ranks(int r, int w, const std::string &dir, double timeout_s = 120.0) {
        if (w < 1 || w > F2V_PUSH_MAX_RANKS || r < 0 || r >= w) throw std::runtime_error("join_ranks: bad rank / world");
        unsigned char mine[F2V_PUSH_EXPORT_BYTES];
        check(f2v_push_export(h, mine));
        auto path = [&](int k) { return dir + "/f2v_export." + std::to_string(k); };
        {
            const std::string tmp = path(r) + ".tmp";
            FILE *f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(mine, 1, sizeof mine, f) != sizeof mine || fclose(f) != 0 || rename(tmp.c_str(), path(r).c_str()) != 0)
                throw std::runtime_error("join_ranks: cannot publish " + path(r));
        }
        std::vector<unsigned char> all((size_t)w * F2V_PUSH_EXPORT_BYTES);
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < w; k++) {
            for (;;) {
                FILE *f = fopen(path(k).c_str(), "rb");
                if (f) {
                    const size_t got = fread(all.data() + (size_t)k * F2V_PUSH_EXPORT_BYTES, 1, F2V_PUSH_EXPORT_BYTES, f);
                    fclose(f);
                    if (got == F2V_PUSH_EXPORT_BYTES) break;
                }
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
                    throw std::runtime_error("join_ranks: rank " + std::to_string(k) + " never published its handles in " + dir);
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
            }
        }
        check(f2v_push_attach(h, (uint32_t)r, (uint32_t)w, all.data()));
        check(f2v_push_selftest(h));
        rank = r;
        world = w;
    }

    // writeToFile, sample/algorithms.h:118-136 (file name rule in f2v_output_name)
    void writeToFile(int option, int bs, INDEXTYPE B, INDEXTYPE IT, INDEXTYPE ns) {
        char name[4096];
        check(f2v_output_name(filename.c_str(), outputdir.c_str(), option, bs, B, DIM, IT, ns, name, sizeof name));
        std::cout << "Creating output file in following directory:" << name << std::endl;
        std::vector<float> x((size_t)rows * DIM);
        check(f2v_get_embeddings(h, x.data()));
        if (text_output) check(f2v_write_embd(name, x.data(), rows, DIM));
        if (binary_output) check(f2v_write_embd_bin((std::string(name) + ".bin").c_str(), x.data(), rows, DIM));
    }

   private:
    unsigned last_seed = 1;
    bool seeded = false;  // srand() was the last thing to touch the handle's rand() stream: a lost run can be repeated from it
    static void check(int rc) {
        if (rc != F2V_OK) throw std::runtime_error(f2v_last_error());
    }
    int64_t param(const char *name) {
        int64_t v = 0;
        check(f2v_get_param(h, name, &v));
        return v;
    }
    // randInitF / randInit (algorithms.cpp:38-53) -- or the embedding file of a warm start
    void init(int math) {
        if (init_path.empty()) {
            check(f2v_init_embeddings(h, math == 5 ? F2V_INIT_SYMMETRIC : F2V_INIT_UNIT));
            return;
        }
        const std::string ext = ".bin";
        if (init_path.size() > ext.size() && init_path.compare(init_path.size() - ext.size(), ext.size(), ext) == 0) {
            std::vector<float> x((size_t)rows * DIM);
            check(f2v_read_embd_bin(init_path.c_str(), rows, DIM, x.data()));
            check(f2v_set_embeddings(h, x.data()));
        } else {
            uint32_t n = 0, d = 0;
            float *x = nullptr;
            check(f2v_read_embd(init_path.c_str(), &n, &d, &x));
            const bool fits = n == rows && d == DIM;
            if (fits) check(f2v_set_embeddings(h, x));
            f2v_free(x);
            if (!fits) throw std::runtime_error("-init: " + init_path + " holds " + std::to_string(n) + " x " + std::to_string(d) + " values, the run needs " +
                                                std::to_string(rows) + " x " + std::to_string(DIM));
        }
    }
    std::vector<VALUETYPE> run(int option, int bs, INDEXTYPE IT, INDEXTYPE B, INDEXTYPE ns, VALUETYPE lr, const char *msg) {
        // the reference's timer spans randInit + the epoch loop (algorithms.cpp:557-558, 647)
        auto t0 = std::chrono::steady_clock::now();
        const int math = (option == 5 || option == 8 || option == 11) ? 5 : 6;
        init(math);
        if (world > 1) {
            check(f2v_train_sharded(h, option, IT, B, ns, lr, bs, &gpu_train_seconds));
        } else {
            // A launch whose in-grid waits gave up (another tenant of the GPU kept its workgroups from starting) is not the end
            // of the run: f2v_train repeats the call by itself from its snapshot ("recover"); where it could not (no room for
            // the snapshot, "recover" = 0) the handle has already switched to launches without in-grid waits, and the run is
            // repeated here from its seed -- same process, same bytes as a healthy run.
            const int64_t before = param("recoveries");
            const bool waits_before = param("merge_finalize") != 0;
            const bool repeatable = seeded;
            seeded = false;
            int rc = f2v_train(h, option, IT, B, ns, lr, bs, &gpu_train_seconds);
            // (a lost launch is the one failure that switches "merge_finalize" from 1 to 0; any other F2V_ESTATE -- and a handle
            // whose in-grid waits the caller had switched off -- is reported, not retried)
            if (rc == F2V_ESTATE && repeatable && waits_before && param("merge_finalize") == 0) {
                std::cerr << "Force2Vec: " << f2v_last_error() << "\nForce2Vec: running again from seed " << last_seed << std::endl;
                check(f2v_srand(h, last_seed));
                init(math);
                rc = f2v_train(h, option, IT, B, ns, lr, bs, &gpu_train_seconds);
            } else if (rc == F2V_OK && param("recoveries") != before) {
                std::cerr << "Force2Vec: " << f2v_last_error() << std::endl;
            }
            check(rc);
        }
        auto t1 = std::chrono::steady_clock::now();
        const double sec = std::chrono::duration<double>(t1 - t0).count();
        f2v_get_stats(h, &stats);
        if (rank == 0) {  // every replica is complete; one of them reports
            std::cout << msg << sec << " seconds" << std::endl;
            writeToFile(option, bs, B, IT, ns);
        }
        return std::vector<VALUETYPE>{(VALUETYPE)sec};
    }
};

}  // namespace f2v_host
#endif

// cli_main.cpp -- ./bin/Force2Vec: the reference's process boundary (Test/Force2Vec.cpp:49-199) over libf2v.
// Same flags, defaults, exit codes, output file names and Results.txt line, so shell scripts and the
// reference's Python scorers keep working; the parsing itself is table driven.  `-threads` and `-gamma` are
// accepted and unused (the force kernels run on the MI355X).
// Extra flags: -device <int>, -seed <int> (default 1 = the reference's srand(1)), -cache 1 (keep / reuse a
// binary CSR "<input>.f2vcsr"), -binout 1 (also write "<output>.embd.bin", raw fp32 N x D), -notext 1, -fastrng 1
// (NON-parity fast mode: device-side initial embeddings and option-7 walks), -gpus <n> (one forked process per GPU,
// devices -device .. -device+n-1: every minibatch's rows sharded over them, new rows pushed over xGMI --
// f2v_train_sharded; the ranks meet through files in a private temporary directory; same output, written by rank 0).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <thread>

#include <sys/wait.h>
#include <unistd.h>

#include "algorithms.hpp"

using namespace f2v_host;

namespace {

struct Settings {
    std::string input, output, init;
    long batch = 384, iter = 1200, threads = (long)std::thread::hardware_concurrency(), dim = 128, nsamples = 5, option = 5, bs = 0;
    long device = 0, seed = 1, cache = 0, binout = 0, fastrng = 0, notext = 0, gpus = 1, samegpu = 0;
    double gamma = 1.0, lr = 0.02;
};

enum class Kind { Text, Integer, Real };
struct Flag {
    const char *name;
    Kind kind;
    void *target;
    const char *help;
};

// option number -> the name the reference prints and logs (Test/Force2Vec.cpp:80-102) and the method that runs it
struct Variant {
    int option;
    const char *name;
    std::vector<VALUETYPE> (algorithms::*plain)(INDEXTYPE, INDEXTYPE, INDEXTYPE, INDEXTYPE, VALUETYPE);
    std::vector<VALUETYPE> (algorithms::*with_bs)(INDEXTYPE, INDEXTYPE, INDEXTYPE, INDEXTYPE, VALUETYPE);
};
const Variant kVariants[] = {
    {5, "Force2Vec:t-distribution with negative sampling", &algorithms::AlgoForce2VecNS, &algorithms::AlgoForce2VecNSBS},
    {6, "Force2Vec:sigmoid with negative sampling", &algorithms::AlgoForce2VecNSRW, &algorithms::AlgoForce2VecNSRWBS},
    {7, "Force2Vec:sigmoid based random-walk", &algorithms::AlgoForce2VecNSRWEFF, nullptr},
    {8, "Force2Vec:AVX512 support for t-distribution with negative sampling", &algorithms::AlgoForce2VecNS_SREAL_D128_AVXZ, nullptr},
    {9, "Force2Vec:AVX512 support for sigmoid with negative sampling", &algorithms::AlgoForce2VecNSRW_SREAL_D128_AVXZ, nullptr},
    {10, "Force2Vec:AVX512 support for sigmoid based random-walk", &algorithms::AlgoForce2VecNSRWEFF_SREAL_D128_AVXZ, nullptr},
    {11, "Force2Vec:Load-balancing with AVX512 support for t-distribution with negative sampling", &algorithms::AlgoForce2VecNSLB_SREAL_D128_AVXZ, nullptr},
};

void usage(const Flag *flags, size_t count) {
    printf("\nUsage of Force2Vec tool:\n");
    for (size_t k = 0; k < count; k++) printf("%s %s\n", flags[k].name, flags[k].help);
    printf("-h, show help message.\n");
}

}  // namespace

int main(int argc, char *argv[]) {
    Settings s;
    const Flag flags[] = {
        {"-input", Kind::Text, &s.input, "<string>, full path of input file (required); a name ending in .f2vcsr is read as a binary CSR."},
        {"-output", Kind::Text, &s.output, "<string>, directory (with trailing /) where the output file will be stored. (default: current directory)"},
        {"-batch", Kind::Integer, &s.batch, "<int>, size of minibatch. (default:384)"},
        {"-iter", Kind::Integer, &s.iter, "<int>, number of iteration. (default:1200)"},
        {"-threads", Kind::Integer, &s.threads, "<int>, accepted for compatibility (the force kernels run on the GPU)."},
        {"-dim", Kind::Integer, &s.dim, "<int>, size of embedding dimension, 1..512. (default:128)"},
        {"-nsamples", Kind::Integer, &s.nsamples, "<int>, number of negative samples. (default:5)"},
        {"-lr", Kind::Real, &s.lr, "<float>, learning rate of SGD. (default:0.02)"},
        {"-gamma", Kind::Real, &s.gamma, "<float>, accepted for compatibility (unused by options 5-11)."},
        {"-bs", Kind::Integer, &s.bs, "<int>, 1 = draw nsamples*batch negative samples per minibatch (options 5 and 6)."},
        {"-option", Kind::Integer, &s.option,
         "<int>, 5 tForce2Vec (t-distribution + negative sampling), 6 sForce2Vec (sigmoid), 7 rForce2Vec (semi-random walk);\n"
         "        8..11 run the same three with the reference's AVX512 output names (8,11 -> 5; 9 -> 6; 10 -> 7). (default:5)"},
        {"-device", Kind::Integer, &s.device, "<int>, HIP device ordinal. (default:0)"},
        {"-seed", Kind::Integer, &s.seed, "<int>, srand() seed. (default:1)"},
        {"-cache", Kind::Integer, &s.cache, "<int>, 1 = keep / reuse the binary CSR <input>.f2vcsr."},
        {"-binout", Kind::Integer, &s.binout, "<int>, 1 = also write <output file>.bin, raw fp32 N x D (the scorers' binary embedding format)."},
        {"-init", Kind::Text, &s.init, "<string>, warm start: a text .embd (or, ending in .bin, raw fp32 N x D) of the run's N and -dim instead of the random initial embedding."},
        {"-notext", Kind::Integer, &s.notext, "<int>, 1 = skip the text .embd (use with -binout 1 for very large graphs)."},
        {"-gpus", Kind::Integer, &s.gpus, "<int>, number of GPUs (1..8): one process per GPU from -device on, minibatch rows sharded, rows exchanged over xGMI. (default:1)"},
        {"-samegpu", Kind::Integer, &s.samegpu, "<int>, 1 = all ranks of a -gpus run on device -device (self-test on a one-GPU machine)."},
        {"-fastrng", Kind::Integer, &s.fastrng, "<int>, 1 = NON-PARITY fast mode: initial embeddings and option-7 walks from a device-side RNG."},
    };
    const size_t nflags = sizeof flags / sizeof flags[0];
    for (int p = 1; p < argc; p++) {
        if (!strcmp(argv[p], "-h")) {
            usage(flags, nflags);
            return 1;  // Test/Force2Vec.cpp:112-115
        }
        for (size_t k = 0; k < nflags && p + 1 < argc; k++) {
            if (strcmp(argv[p], flags[k].name)) continue;
            const char *v = argv[++p];
            if (flags[k].kind == Kind::Text) *static_cast<std::string *>(flags[k].target) = v;
            else if (flags[k].kind == Kind::Integer) *static_cast<long *>(flags[k].target) = atol(v);
            else *static_cast<double *>(flags[k].target) = atof(v);
            break;
        }
    }
    if (s.input.empty()) {
        printf("Valid input file needed!...\n");  // Test/Force2Vec.cpp:117-120
        return 1;
    }
    const Variant *variant = nullptr;
    for (const Variant &v : kVariants)
        if (v.option == s.option) variant = &v;
    if (!variant) {
        printf("This build implements options 5 to 11 (the negative-sampling force kernels); option %ld is out of scope.\n", s.option);
        return 1;
    }
    if (s.batch <= 0 || s.dim <= 0 || s.iter < 0 || s.nsamples < 0) {
        printf("-batch and -dim must be positive, -iter and -nsamples non-negative.\n");
        return 1;
    }
    if (s.gpus < 1 || s.gpus > F2V_PUSH_MAX_RANKS) {
        printf("-gpus must be 1..%d.\n", F2V_PUSH_MAX_RANKS);
        return 1;
    }
    std::vector<VALUETYPE> seconds;
    int rank = 0;
    std::string meet;  // directory the ranks of a -gpus run meet in
    try {
        CSRGraph graph;
        SetInputMatricesAsCSR(graph, s.input, s.cache != 0);  // host only: read once, inherited by the forked ranks
        std::vector<pid_t> kids;
        if (s.gpus > 1) {
            char tmpl[] = "/tmp/f2v_ranks_XXXXXX";
            if (!mkdtemp(tmpl)) throw std::runtime_error("cannot create a meeting directory under /tmp");
            meet = tmpl;
            fflush(nullptr);
            // no HIP call has been made yet: every rank initialises its own GPU after the fork
            for (int r = 1; r < s.gpus; r++) {
                const pid_t pid = fork();
                if (pid < 0) throw std::runtime_error("fork failed");
                if (pid == 0) { rank = r; kids.clear(); break; }
                kids.push_back(pid);
            }
        }
        int rc_mine = 0;
        try {
            algorithms algo(graph, s.input, s.output, (INDEXTYPE)s.dim, (VALUETYPE)s.gamma, (INDEXTYPE)s.batch, (int)s.device + (s.samegpu ? 0 : rank));
            algo.binary_output = s.binout != 0;
            algo.text_output = s.notext == 0;
            algo.init_path = s.init;
            if (s.fastrng && f2v_set_param(algo.h, "fast_rng", 1) != F2V_OK) throw std::runtime_error(f2v_last_error());
            algo.srand((unsigned)s.seed);
            if (s.gpus > 1) algo.join_ranks(rank, (int)s.gpus, meet);
            if (rank == 0) std::cout << "Running: " << variant->name << std::endl;
            auto method = (s.bs != 0 && variant->with_bs) ? variant->with_bs : variant->plain;
            seconds = (algo.*method)((INDEXTYPE)s.iter, (INDEXTYPE)s.threads, (INDEXTYPE)s.batch, (INDEXTYPE)s.nsamples, (VALUETYPE)s.lr);
            const double t = algo.gpu_train_seconds;
            if (rank == 0 && s.gpus == 1)
                printf("GPU epoch loop: %.6f s, %.4g nnz/s, %.1f GB/s algorithmic\n", t, t > 0 ? algo.stats.nnz / t : 0.0,
                       t > 0 ? algo.stats.algorithmic_bytes / t * 1e-9 : 0.0);
            else if (rank == 0)
                printf("GPU epoch loop: %.6f s, %.4g nnz/s on %ld GPUs\n", t, t > 0 ? (double)graph.nnz * (double)s.iter / t : 0.0, s.gpus);
        } catch (const std::exception &e) {
            fprintf(stderr, "Force2Vec%s: %s\n", s.gpus > 1 ? (" [rank " + std::to_string(rank) + "]").c_str() : "", e.what());
            rc_mine = 2;
        }
        if (rank != 0) _exit(rc_mine);  // a forked rank: nothing of the parent's to unwind
        for (pid_t pid : kids) {
            int st = 0;
            if (waitpid(pid, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) rc_mine = 2;
        }
        if (!meet.empty()) {
            for (int r = 0; r < s.gpus; r++) (void)remove((meet + "/f2v_export." + std::to_string(r)).c_str());
            (void)rmdir(meet.c_str());
        }
        if (rc_mine) return rc_mine;
    } catch (const std::exception &e) {
        fprintf(stderr, "Force2Vec: %s\n", e.what());
        return 2;
    }
    // one line per run appended to ./Results.txt, in the reference's format (Test/Force2Vec.cpp:191-198)
    std::ofstream log("Results.txt", std::ofstream::app);
    log << "Algo:" << variant->name << "\tInit:RAND\tIteration:" << s.iter << "\tNumofthreads:" << s.threads << "\tBatchSize:" << s.batch
        << "\tDimension:" << s.dim << "\tTime(sec.):" << seconds[0] << "\t" << std::endl;
    return 0;
}

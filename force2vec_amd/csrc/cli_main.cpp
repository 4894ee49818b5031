// cli_main.cpp -- ./bin/Force2Vec: the reference's process boundary (Test/Force2Vec.cpp) over
// libf2v.  Same flags, defaults, messages, exit codes, output file names and Results.txt line;
// `-threads` and `-gamma` are accepted and unused (the force kernels run on the MI355X).
// Additional flags: -device <int>, -seed <int> (default 1, the reference's srand(1)), -cache 1 (keep/reuse a
// binary CSR "<input>.f2vcsr"), -binout 1 (also write "<output>.embd.bin", raw fp32 N x D).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <thread>

#include "algorithms.hpp"

using namespace std;
using namespace f2v_host;

static void helpmessage() {
    printf("\n");
    printf("Usage of Force2Vec tool:\n");
    printf("-input <string>, full path of input file (required).\n");
    printf("-output <string>, directory where output file will be stored. (default: current directory)\n");
    printf("-batch <int>, size of minibatch. (default:384)\n");
    printf("-iter <int>, number of iteration. (default:1200)\n");
    printf("-threads <int>, accepted for compatibility (the force kernels run on the GPU).\n");
    printf("-dim <int>, size of embedding dimension. (default:128) \n");
    printf("-nsamples <int>, number of negative samples. (default:5) \n");
    printf("-lr <float>, learning rate of SGD. (default:0.02)\n");
    printf("-option <int>, an integer among 5 to 11. (default:5)\n");
    printf("        -option 5 - for t-distribution + negative sampling (tForce2Vec).\n");
    printf("        -option 6 - for sigmoid + negative sampling (sForce2Vec).\n");
    printf("        -option 7 - for sigmoid + semi-random walk (rForce2Vec).\n");
    printf("        -option 8..11 - the same three on the GPU with hub rows load-balanced (8,11 -> 5; 9 -> 6; 10 -> 7).\n");
    printf("-bs <int>, 1 = draw nsamples*batch negative samples per minibatch (options 5 and 6).\n");
    printf("-device <int>, HIP device ordinal. (default:0)\n");
    printf("-seed <int>, srand() seed. (default:1)\n");
    printf("-fastrng <int>, 1 = NON-PARITY fast mode: initial embeddings and option-7 walks from a device-side RNG.\n");
    printf("-cache <int>, 1 = keep / reuse the binary CSR <input>.f2vcsr (an input ending in .f2vcsr is read directly).\n");
    printf("-binout <int>, 1 = also write <output file>.bin, raw fp32 N x D (the scorers' binary embedding format).\n");
    printf("-h, show help message.\n");
}

static int TestAlgorithms(int argc, char *argv[]) {
    VALUETYPE gamma = 1.0, lr = 0.02;
    INDEXTYPE batchsize = 384, iterations = 1200, numberOfThreads = std::thread::hardware_concurrency(), dim = 128, option = 5, nsamples = 5;
    string inputfile = "", outputfile = "", algoname = "Force2Vec:t-distribution with negative sampling", initname = "RAND";
    INDEXTYPE bs = 0;
    int device = 0, cache = 0, binout = 0, fastrng = 0;
    unsigned seed = 1;
    for (int p = 0; p < argc; p++) {
        const bool has_val = p + 1 < argc;
        if (strcmp(argv[p], "-h") == 0) {
            helpmessage();
            exit(1);
        }
        if (!has_val) continue;
        if (strcmp(argv[p], "-input") == 0) inputfile = argv[p + 1];
        else if (strcmp(argv[p], "-output") == 0) outputfile = argv[p + 1];
        else if (strcmp(argv[p], "-batch") == 0) batchsize = atoi(argv[p + 1]);
        else if (strcmp(argv[p], "-iter") == 0) iterations = atoi(argv[p + 1]);
        else if (strcmp(argv[p], "-threads") == 0) numberOfThreads = atoi(argv[p + 1]);
        else if (strcmp(argv[p], "-dim") == 0) dim = atoi(argv[p + 1]);
        else if (strcmp(argv[p], "-gamma") == 0) gamma = atof(argv[p + 1]);
        else if (strcmp(argv[p], "-bs") == 0) bs = atoi(argv[p + 1]);
        else if (strcmp(argv[p], "-device") == 0) device = atoi(argv[p + 1]);
        else if (strcmp(argv[p], "-seed") == 0) seed = (unsigned)atoi(argv[p + 1]);
        else if (strcmp(argv[p], "-cache") == 0) cache = atoi(argv[p + 1]);
        else if (strcmp(argv[p], "-fastrng") == 0) fastrng = atoi(argv[p + 1]);
        else if (strcmp(argv[p], "-binout") == 0) binout = atoi(argv[p + 1]);
        else if (strcmp(argv[p], "-lr") == 0) lr = atof(argv[p + 1]);
        else if (strcmp(argv[p], "-nsamples") == 0) nsamples = atoi(argv[p + 1]);
        else if (strcmp(argv[p], "-option") == 0) {
            option = atoi(argv[p + 1]);
            if (option == 5) algoname = "Force2Vec:t-distribution with negative sampling";
            else if (option == 6) algoname = "Force2Vec:sigmoid with negative sampling";
            else if (option == 7) algoname = "Force2Vec:sigmoid based random-walk";
            else if (option == 8) algoname = "Force2Vec:AVX512 support for t-distribution with negative sampling";
            else if (option == 9) algoname = "Force2Vec:AVX512 support for sigmoid with negative sampling";
            else if (option == 10) algoname = "Force2Vec:AVX512 support for sigmoid based random-walk";
            else if (option == 11) algoname = "Force2Vec:Load-balancing with AVX512 support for t-distribution with negative sampling";
        }
    }
    if (inputfile.size() == 0) {
        printf("Valid input file needed!...\n");
        exit(1);
    }
    if (option < 5 || option > 11) {
        printf("This build implements options 5 to 11 (the negative-sampling force kernels); option %u is out of scope.\n", option);
        exit(1);
    }
    if (batchsize == 0 || dim == 0) {
        printf("-batch and -dim must be positive.\n");
        exit(1);
    }
    vector<VALUETYPE> outputvec;
    try {
        CSRGraph A_csr;
        SetInputMatricesAsCSR(A_csr, inputfile, cache != 0);
        algorithms algo(A_csr, inputfile, outputfile, dim, gamma, batchsize, device);
        algo.binary_output = binout != 0;
        if (fastrng && f2v_set_param(algo.h, "fast_rng", 1) != F2V_OK) throw std::runtime_error(f2v_last_error());
        algo.srand(seed);
        cout << "Running: " << algoname << endl;
        if (option == 5) outputvec = bs == 0 ? algo.AlgoForce2VecNS(iterations, numberOfThreads, batchsize, nsamples, lr)
                                             : algo.AlgoForce2VecNSBS(iterations, numberOfThreads, batchsize, nsamples, lr);
        else if (option == 6) outputvec = bs == 0 ? algo.AlgoForce2VecNSRW(iterations, numberOfThreads, batchsize, nsamples, lr)
                                                  : algo.AlgoForce2VecNSRWBS(iterations, numberOfThreads, batchsize, nsamples, lr);
        else if (option == 7) outputvec = algo.AlgoForce2VecNSRWEFF(iterations, numberOfThreads, batchsize, nsamples, lr);
        else if (option == 8) outputvec = algo.AlgoForce2VecNS_SREAL_D128_AVXZ(iterations, numberOfThreads, batchsize, nsamples, lr);
        else if (option == 9) outputvec = algo.AlgoForce2VecNSRW_SREAL_D128_AVXZ(iterations, numberOfThreads, batchsize, nsamples, lr);
        else if (option == 10) outputvec = algo.AlgoForce2VecNSRWEFF_SREAL_D128_AVXZ(iterations, numberOfThreads, batchsize, nsamples, lr);
        else outputvec = algo.AlgoForce2VecNSLB_SREAL_D128_AVXZ(iterations, numberOfThreads, batchsize, nsamples, lr);
        const double esec = algo.gpu_train_seconds > 0 ? (double)algo.stats.nnz / algo.gpu_train_seconds : 0.0;
        printf("GPU epoch loop: %.6f s, %.4g nnz/s, %.1f GB/s algorithmic\n", algo.gpu_train_seconds, esec,
               algo.gpu_train_seconds > 0 ? algo.stats.algorithmic_bytes / algo.gpu_train_seconds * 1e-9 : 0.0);
    } catch (const std::exception &e) {
        fprintf(stderr, "Force2Vec: %s\n", e.what());
        return 2;
    }
    ofstream output;
    output.open("Results.txt", ofstream::app);  // Test/Force2Vec.cpp:191-198
    output << "Algo:" << algoname << "\tInit:" << initname << "\tIteration:";
    output << iterations << "\tNumofthreads:" << numberOfThreads << "\tBatchSize:" << batchsize << "\tDimension:" << dim << "\tTime(sec.):";
    output << outputvec[0] << "\t";
    output << endl;
    output.close();
    return 0;
}

int main(int argc, char *argv[]) { return TestAlgorithms(argc, argv); }

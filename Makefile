# Builds the MI355X (gfx950) Force2Vec engine: libf2v.so (C ABI, include/f2v.h) and the
# drop-in CLI bin/Force2Vec.  hipcc cross-compiles without a GPU.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC = force2vec_amd/csrc
# -ffp-contract=off: the reference's mul/add pairs are separate roundings (x86-64 without FMA);
# parity with the oracle depends on it.
CXXFLAGS = -O3 -std=c++17 -fPIC -pthread -Iinclude -I$(CSRC) -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result
# The libraries export the C ABI of include/f2v.h (F2V_API) and nothing else: hidden visibility for every other symbol -- the
# kernels' host-side launch stubs included -- and -Bsymbolic, so that a second build of the library in the same process (the
# self-test build preloaded under the CLI, an older libf2v of another package) can neither interpose on this one's internals
# nor be interposed on.  (Round 3's "memory access fault ... address 0x1000": DESIGN section 3.)
LIBFLAGS = -fvisibility=hidden -fvisibility-inlines-hidden -Wl,-Bsymbolic -Wl,--exclude-libs,ALL -Wl,--version-script=$(CSRC)/libf2v.map
HIPFLAGS = --offload-arch=$(ARCH) $(CXXFLAGS) $(LIBFLAGS)
LIB = force2vec_amd/libf2v.so
# the same sources with the self-test hooks of include/f2v_test.h compiled in (tests/, tools/ only)
TESTLIB = force2vec_amd/libf2v_selftest.so
SRCS = $(CSRC)/f2v_engine.hip $(CSRC)/f2v_kernels.hip.h $(CSRC)/f2v_host.cpp $(CSRC)/f2v_internal.h include/f2v.h $(CSRC)/libf2v.map

all: $(LIB) $(TESTLIB) bin/Force2Vec bin/Force2Vec_selftest

$(LIB): $(SRCS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/f2v_engine.hip $(CSRC)/f2v_host.cpp

$(TESTLIB): $(SRCS) include/f2v_test.h
	$(HIPCC) $(HIPFLAGS) -DF2V_TEST_HOOKS -shared -o $@ $(CSRC)/f2v_engine.hip $(CSRC)/f2v_host.cpp

bin/Force2Vec: $(CSRC)/cli_main.cpp $(CSRC)/algorithms.hpp include/f2v.h $(LIB)
	mkdir -p bin
	$(HIPCC) $(CXXFLAGS) -o $@ $(CSRC)/cli_main.cpp -Lforce2vec_amd -lf2v -Wl,-rpath,'$$ORIGIN/../force2vec_amd'

# the same CLI over the self-test build of the library (fault injection through F2V_TEST_* environment variables: tests/ only)
bin/Force2Vec_selftest: $(CSRC)/cli_main.cpp $(CSRC)/algorithms.hpp include/f2v.h $(TESTLIB)
	mkdir -p bin
	$(HIPCC) $(CXXFLAGS) -o $@ $(CSRC)/cli_main.cpp -Lforce2vec_amd -lf2v_selftest -Wl,-rpath,'$$ORIGIN/../force2vec_amd'

clean:
	rm -f $(LIB) $(TESTLIB) bin/Force2Vec bin/Force2Vec_selftest
.PHONY: all clean

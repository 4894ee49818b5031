"""A/B timing of several builds of the library on ONE box (force2vec_amd/libf2v*.so except the self-test build, selected through
F2V_LIBRARY): device time per epoch of chained minibatches (batch 256, 384, 1024, or the sizes given as arguments) on RMAT-20,
interleaved, three rounds.   usage: ab_compare_small.py [BATCH ...]"""
import glob, os, subprocess, sys
batches = [int(x) for x in sys.argv[1:]] or [256, 384, 1024]
code = r'''
import sys; sys.path.insert(0, ".")
BATCHES = %r
import bench, force2vec_amd as F
rowptr, colids = bench.load_graph(20, 16, 1)
eng = F.Engine(rowptr, colids, 128)
eng.srand(1); eng.init_embeddings(0)
out = []
for b in BATCHES:
    eng.train(5, 3, b)
    out.append(min(eng.train(5, 3, b) / 3 for _ in range(3)) * 1e3)
print(" ".join("%%.3f" %% x for x in out))
''' % (batches,)
libs = [x for x in sorted(glob.glob("force2vec_amd/libf2v*.so")) if "selftest" not in x]
for rep in range(3):
    for lib in libs:
        env = dict(os.environ, F2V_LIBRARY=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        print("%-14s batch %s: %s ms/epoch" % (os.path.basename(lib)[3:-3], " / ".join(map(str, batches)), out), flush=True)

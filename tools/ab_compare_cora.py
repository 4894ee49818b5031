import glob, os, subprocess, sys
code = r'''
import sys, os; sys.path.insert(0, ".")
import force2vec_amd as F
rowptr, colids = F.read_mtx("tests/golden/cora.mtx")
out = []
for dim in (16, 128):
    eng = F.Engine(rowptr, colids, dim)
    eng.srand(1); eng.init_embeddings(0)
    eng.train(5, 50, 256)
    out.append(min(eng.train(5, 1200, 256) for _ in range(3)))
    eng.close()
print("D=16 %.4f s  D=128 %.4f s" % tuple(out))
'''
libs = [x for x in sorted(glob.glob("force2vec_amd/libf2v*.so")) if "selftest" not in x]
for rep in range(2):
    for lib in libs:
        env = dict(os.environ, F2V_LIBRARY=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        print("%-14s cora 1200 epochs: %s" % (os.path.basename(lib)[3:-3], out), flush=True)

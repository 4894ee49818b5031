"""A/B timing of several builds of the library on ONE box (force2vec_amd/libf2v*.so, selected through F2V_LIBRARY): device
time per epoch at batch 65536 and 262144, RMAT-20."""
import os, subprocess, sys
code = r'''
import sys; sys.path.insert(0, ".")
import bench, force2vec_amd as F
rowptr, colids = bench.load_graph(20, 16, 1)
eng = F.Engine(rowptr, colids, 128)
eng.srand(1); eng.init_embeddings(0)
eng.train(5, 25, 65536)
best = min(eng.train(5, 10, 65536) / 10 for _ in range(4))
b2 = 0
eng.train(5, 5, 262144)
b2 = min(eng.train(5, 8, 262144) / 8 for _ in range(3))
print("%.4f %.4f" % (best * 1e3, b2 * 1e3))
'''
for rep in range(3):
    for tag, lib in [(os.path.basename(x)[6:-3] or "head", x) for x in sorted(__import__("glob").glob("force2vec_amd/libf2v*.so"))]:
        env = dict(os.environ, F2V_LIBRARY=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        print(tag, out, flush=True)

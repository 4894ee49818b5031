#!/usr/bin/env python3
"""BASELINE configs[0] / [1] (the reference's own dataset): cora, batch 256 (and the CLI default 384), 1200 epochs -- device time of
f2v_train in its three launch forms (wide chained, round-2 chained, one launch per minibatch), options 5 / 6 / 7; then engine
parameters swept on option 5, batch 256.   Usage: cora_configs.py [key=v1,v2,... ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import force2vec_amd as F

rowptr, colids = F.read_mtx(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cora.mtx"))
FORMS = {2: "wide chained", 1: "round-2 chained", 0: "one launch per minibatch"}


def run(dim, option, batch, params, iters=1200):
    eng = F.Engine(rowptr, colids, dim)
    for k, v in params.items():
        eng.set_param(k, v)
    eng.srand(1)
    eng.init_embeddings(0 if option in (5, 8, 11) else 1)
    eng.train(option, 50, batch)
    sec = min(eng.train(option, iters, batch) for _ in range(2))
    form = eng.get_param("last_train_form")
    eng.close()
    return sec, form


sweeps = sys.argv[1:]
if not sweeps:
    for dim in (16, 128):
        for option in (5, 6, 7):
            for batch in (256, 384):
                out = []
                for params in ({}, {"chain_wide": 0}, {"chain_batches": 0}):
                    sec, form = run(dim, option, batch, params)
                    out.append("%.3f s %s" % (sec, FORMS[form]))
                print("cora D=%3d option %d batch %d, 1200 epochs: %s" % (dim, option, batch, " | ".join(out)), flush=True)
for sweep in sweeps:
    key, vals = sweep.split("=")
    for v in vals.split(","):
        print("  %s=%s: " % (key, v) + "  ".join("D=%d %.4f s" % (dim, run(dim, 5, 256, {key: int(v)})[0]) for dim in (16, 128)), flush=True)

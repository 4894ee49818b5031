#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc CSVs of tools/profile_traffic.sh into traffic.json: per launch of the dominant kernel, the
bytes that left the XCDs' L2s (FETCH_SIZE corrected by the factor measured on the calibration kernel -- known byte count,
same access pattern -- plus WRITE_SIZE), the L2 hit rate, the fabric read-request mix and average read latency next to
the same figures of the calibration kernel reading from HBM and from the Infinity Cache, and the wave-level stall split."""
import csv
import glob
import json
import os
import sys


def counters(d, kernel_substr, skip_first=0):
    """-> {counter: mean over dispatches}, number of dispatches; the first `skip_first` dispatches of the kernel are left out."""
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel_substr in row["Kernel_Name"]:
                key = (row["Counter_Name"], int(row["Dispatch_Id"]))
                per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
    names = sorted({k[0] for k in per})
    out, nd = {}, 0
    for nm in names:
        ids = sorted(i for (c, i) in per if c == nm)[skip_first:]
        if ids:
            out[nm] = sum(per[(nm, i)] for i in ids) / len(ids)
            nd = len(ids)
    return out, nd


def main():
    prof, out = sys.argv[1], sys.argv[2]
    batch, scale, dim = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    option = int(sys.argv[6]) if len(sys.argv) > 6 else 5
    # the step kernel of the run: the name the bench line reports (one launch per minibatch, chained, or the wide form) up to its
    # template list; without a bench line any of the three
    K = None
    try:
        K = json.loads([l for l in open(os.path.join(prof, "pmc_fetch.json")) if l.startswith("{")][-1])["roofline"]["kernel"].split("::")[-1].split("<")[0]
    except Exception:
        pass
    if not K:
        K = "_kernel"  # (qstep_kernel / qstep_chain_kernel / qwide_chain_kernel are the only kernels of the timed region)
    step = {}
    for name in ("fetch", "write", "l2", "ea_rd", "ea_wr", "sq"):
        c, nd = counters(os.path.join(prof, "pmc_" + name), K)
        step.update(c)
        step["launches_" + name] = nd
    if "FETCH_SIZE" not in step or "WRITE_SIZE" not in step:
        sys.exit("parse_pmc: no counters found for a kernel named *%s* under %s (is the kernel name right?)" % (K, prof))
    bench_line = {}
    try:
        bench_line = json.loads([l for l in open(os.path.join(prof, "pmc_fetch.json")) if l.startswith("{")][-1])
    except Exception:
        pass
    cal = {}
    for where, rows in (("hbm", 4194304), ("mall", 393216)):
        c = {}
        for kind in ("fetch", "ea", "l2"):
            cc, nd = counters(os.path.join(prof, "cal_%s_%s" % (where, kind)), "gather_calibration_kernel", skip_first=2)
            c.update(cc)
        c["known_bytes_per_launch"] = rows * 512 + rows * 4
        if c.get("TCC_EA0_RDREQ_sum"):
            c["avg_ea_read_latency_cycles"] = c["TCC_EA0_RDREQ_LEVEL_sum"] / c["TCC_EA0_RDREQ_sum"]
        if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum"):
            c["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        cal[where] = c
    factor = cal["hbm"]["known_bytes_per_launch"] / (cal["hbm"]["FETCH_SIZE"] * 1024.0)
    l2_miss = step["FETCH_SIZE"] * 1024.0 * factor + step["WRITE_SIZE"] * 1024.0
    import time
    res = {"batch": batch, "scale": scale, "dim": dim, "option": option, "collected": time.strftime("%Y-%m-%d"),
           "hub_chunk": (bench_line.get("config") or {}).get("hub_chunk"), "kernel": (bench_line.get("roofline") or {}).get("kernel"),
           "library": None,
           "l2_miss_bytes_per_launch": l2_miss,
           "l2_hit_rate": step["TCC_HIT_sum"] / (step["TCC_HIT_sum"] + step["TCC_MISS_sum"]) if step.get("TCC_MISS_sum") else None,
           "avg_ea_read_latency_cycles": step["TCC_EA0_RDREQ_LEVEL_sum"] / step["TCC_EA0_RDREQ_sum"] if step.get("TCC_EA0_RDREQ_sum") else None,
           "fetch_correction_factor": factor,
           "step_kernel": step, "calibration": cal,
           "note": "l2_miss_bytes_per_launch = FETCH_SIZE x fetch_correction_factor (measured on the calibration kernel: the same 16-lane dwordx4 row "
                   "gathers, every byte fetched once from a 2-GiB table) + WRITE_SIZE (exact for 16-B/lane stores), KiB -> bytes, mean over the step "
                   "kernel's launches.  These are requests of the L2s to the fabric: Infinity-Cache hits are included (MI355X_MICROARCH.md, HBM); "
                   "rocprofv3 -L lists no counter beyond the TCC on this part, so HBM and Infinity Cache cannot be split by counters.  "
                   "avg_ea_read_latency_cycles (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ) next to calibration.hbm / calibration.mall shows which of the two the "
                   "kernel's misses resemble."}
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from force2vec_amd import _lib
        res["library"] = _lib.lib().f2v_version().decode()
    except Exception:
        pass
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k not in ("step_kernel", "calibration", "note")}))


if __name__ == "__main__":
    main()

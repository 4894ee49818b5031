#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc CSVs of tools/profile_traffic.sh into profiles/traffic.json:
HBM bytes per launch of the dominant kernel, FETCH_SIZE corrected by the factor measured on the
calibration kernel (known byte count, same access pattern), WRITE_SIZE taken as is."""
import csv
import glob
import json
import os
import sys


def mean_counter(d, kernel_substr, counter):
    vals = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel_substr in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.setdefault(row["Dispatch_Id"], 0.0)
                vals[row["Dispatch_Id"]] += float(row["Counter_Value"])
    v = list(vals.values())
    return (sum(v) / len(v), len(v)) if v else (None, 0)


def main():
    prof, out, batch, scale, dim = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    cal_rows = 4 * 1024 * 1024
    known = cal_rows * 512 + cal_rows * 4
    f_cal, n1 = mean_counter(os.path.join(prof, "cal_FETCH_SIZE"), "gather_calibration_kernel", "FETCH_SIZE")
    f_step, n2 = mean_counter(os.path.join(prof, "pmc_FETCH_SIZE"), "qstep_kernel", "FETCH_SIZE")
    w_step, n3 = mean_counter(os.path.join(prof, "pmc_WRITE_SIZE"), "qstep_kernel", "WRITE_SIZE")
    f_fin, _ = mean_counter(os.path.join(prof, "pmc_FETCH_SIZE"), "hub_finalize_kernel", "FETCH_SIZE")
    w_fin, _ = mean_counter(os.path.join(prof, "pmc_WRITE_SIZE"), "hub_finalize_kernel", "WRITE_SIZE")
    factor = known / (f_cal * 1024.0)
    res = {"batch": batch, "scale": scale, "dim": dim,
           "calibration": {"kernel": "gather_calibration_kernel<2>", "known_bytes_per_launch": known, "FETCH_SIZE_KiB": f_cal,
                           "bytes_per_FETCH_SIZE_KiB": factor * 1024.0, "correction_factor": factor, "launches": n1},
           "step_kernel": {"FETCH_SIZE_KiB": f_step, "WRITE_SIZE_KiB": w_step, "launches": n2},
           "hub_finalize_kernel": {"FETCH_SIZE_KiB": f_fin, "WRITE_SIZE_KiB": w_fin},
           "hbm_bytes_per_launch": f_step * 1024.0 * factor + w_step * 1024.0,
           "note": "FETCH_SIZE x correction_factor (measured on the calibration kernel: same 16-lane dwordx4 row gathers, every byte "
                   "fetched once from a 2 GiB table) + WRITE_SIZE (exact for 16-B/lane stores), KiB -> bytes; per step-kernel launch"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

#!/usr/bin/env bash
# Run ON THE GPU BOX (gpurun): what the CUs do during chained minibatches -- SQ wave-cycle split and L2 counters of the step kernel at a
# small batch, separate rocprofv3 --pmc passes (never combined with trace domains other than --kernel-trace).
#   usage: tools/pmc_small_batch.sh [BATCH]
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
BATCH=${1:-384}
O=$R/gpurun_out/pmc_small
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --extra-batches= --config5-scale 0 --config4 0 --cora 0 --sustained-s 0 --verify-rows 0 --steps 3 --warmup 1 --settle-ms 0 --batch $BATCH"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- $B > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --kernel-trace --output-format csv -d $O/l2 -- $B > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B > /dev/null 2>&1
python3 - <<PY
import sys
sys.path.insert(0, "$R/tools")
from parse_pmc import counters
k = "_chain_kernel<"
sq, n = counters("$O/sq", k)
l2, _ = counters("$O/l2", k)
f, _ = counters("$O/fetch", k)
w, _ = counters("$O/write", k)
wc = sq["SQ_WAVE_CYCLES"]
print("batch $BATCH, per launch of the chained step kernel (%d launches):" % n)
print("  wave-cycles split: SQ_WAIT_ANY %.1f %%, SQ_ACTIVE_INST_ANY %.1f %%, SQ_WAIT_INST_ANY %.1f %%, VALU %.1f %%; waves %.0f" % (
    100 * sq["SQ_WAIT_ANY"] / wc, 100 * sq["SQ_ACTIVE_INST_ANY"] / wc, 100 * sq["SQ_WAIT_INST_ANY"] / wc, 100 * sq["SQ_ACTIVE_INST_VALU"] / wc, sq["SQ_WAVES"]))
print("  L2: hit rate %.3f, requests %.0f; FETCH_SIZE %.0f KiB (x2 on gfx950 for wide reads), WRITE_SIZE %.0f KiB" % (
    l2["TCC_HIT_sum"] / (l2["TCC_HIT_sum"] + l2["TCC_MISS_sum"]), l2["TCC_REQ_sum"], f["FETCH_SIZE"], w["WRITE_SIZE"]))
PY

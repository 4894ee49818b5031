#!/usr/bin/env bash
# Run ON THE GPU BOX (gpurun): A/B of the "piece_affinity" experiment (hub pieces placed on the XCD that owns the id range of
# their neighbours) on the bench workload -- epoch time, then L2 hit rate and L2-miss bytes from separate PMC passes.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
BATCH=${1:-65536}; PARAM=${2:-piece_affinity}
O=$R/gpurun_out/affinity
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --extra-batches= --config5-scale 0 --config4 0 --verify-rows 0 --batch $BATCH"
for rep in 1 2; do for a in 0 1; do
  $B --steps 30 --warmup 5 --param $PARAM=$a 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$PARAM $a run $rep: %.4f ms/epoch  %.2f G edges/s' % (r['ms_per_step'], r['value']/1e9))" | tee -a $O/times.txt
done; done
for a in 0 1; do
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/l2_$a -- $B --steps 3 --warmup 1 --settle-ms 0 --param $PARAM=$a > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$a -- $B --steps 3 --warmup 1 --settle-ms 0 --param $PARAM=$a > /dev/null 2>&1
done
python3 - <<PY | tee -a $O/times.txt
import sys
sys.path.insert(0, "$R/tools")
from parse_pmc import counters
for a in (0, 1):
    c, n = counters("$O/l2_%d" % a, "qstep_kernel")
    f, _ = counters("$O/fetch_%d" % a, "qstep_kernel")
    print("'$PARAM' %d: L2 hit rate %.4f (%d launches), FETCH_SIZE %.0f KiB per launch" % (a, c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), n, f["FETCH_SIZE"]))
PY

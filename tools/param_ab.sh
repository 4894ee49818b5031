#!/usr/bin/env bash
# Run ON THE GPU BOX (gpurun): one engine parameter over several values on the bench workload -- epoch time (two runs each, interleaved),
# then L2 hit rate and FETCH_SIZE per launch of the step kernel from separate PMC passes.
#   usage: tools/param_ab.sh BATCH PARAM VALUE [VALUE ...]
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
BATCH=$1; PARAM=$2; shift 2
O=$R/gpurun_out/ab_$PARAM
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --extra-batches= --config5-scale 0 --config4 0 --cora 0 --sustained-s 0 --verify-rows 0 --batch $BATCH"
for rep in 1 2; do for a in "$@"; do
  $B --steps 30 --warmup 5 --param $PARAM=$a 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('$PARAM $a run $rep: %.4f ms/epoch  %.2f G edges/s  (hub chunk %s)' % (r['ms_per_step'], r['value']/1e9, r['config']['hub_chunk']))" | tee -a $O/times.txt
done; done
for a in "$@"; do
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/l2_$a -- $B --steps 3 --warmup 1 --settle-ms 0 --param $PARAM=$a > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$a -- $B --steps 3 --warmup 1 --settle-ms 0 --param $PARAM=$a > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_$a -- $B --steps 3 --warmup 1 --settle-ms 0 --param $PARAM=$a > /dev/null 2>&1
done
python3 - "$@" <<PY | tee -a $O/times.txt
import sys
sys.path.insert(0, "$R/tools")
from parse_pmc import counters
for a in sys.argv[1:]:
    c, n = counters("$O/l2_%s" % a, "qstep_kernel<")
    f, _ = counters("$O/fetch_%s" % a, "qstep_kernel<")
    w, _ = counters("$O/write_%s" % a, "qstep_kernel<")
    print("'$PARAM' %s: L2 hit rate %.4f (%d launches), FETCH_SIZE %.0f KiB, WRITE_SIZE %.0f KiB per launch" % (a, c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), n, f["FETCH_SIZE"], w["WRITE_SIZE"]))
PY

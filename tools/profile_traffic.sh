#!/usr/bin/env bash
# Run ON THE GPU BOX from the repo root (gpurun): separate rocprofv3 --pmc passes for FETCH_SIZE and
# WRITE_SIZE over the bench command and over the calibration kernel, then profiles/traffic.json.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
BATCH=${1:-65536}; SCALE=${2:-20}; DIM=${3:-128}
P=$R/gpurun_out/prof
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/pmc_$C -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --settle-ms 0 --extra-batches "" --batch $BATCH --scale $SCALE --dim $DIM > $P/pmc_$C.json 2> $P/pmc_$C.err
  echo "pass $C done"
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/cal_FETCH_SIZE -- python3 $R/tools/calibrate_gather.py > $P/cal.log 2>&1
echo "calibration done"
python3 $R/tools/parse_pmc.py $P $R/gpurun_out/traffic.json $BATCH $SCALE $DIM

#!/usr/bin/env bash
# Run ON THE GPU BOX from the repo root (gpurun): separate rocprofv3 --pmc passes (never combined with the trace domains
# gpurun refuses) over the bench command and over the calibration kernel, then gpurun_out/traffic.json -- copy it to
# profiles/traffic.json (bench.py reads it) and to profiles/rNN_pmc_*.json.
#   usage: tools/profile_traffic.sh [BATCH [SCALE [DIM [OPTION]]]]
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
BATCH=${1:-65536}; SCALE=${2:-20}; DIM=${3:-128}; OPTION=${4:-5}
P=$R/gpurun_out/prof
rm -rf $P && mkdir -p $P
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --settle-ms 0 --extra-batches= --config5-scale 0 --config4 0 --cora 0 --sustained-s 0 --live-pmc 0 --option7 0 --verify-rows 0 --batch $BATCH --scale $SCALE --dim $DIM --option $OPTION"
pass() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $P/pmc_$name -- $BENCH > $P/pmc_$name.json 2> $P/pmc_$name.err
  echo "pass $name done"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass ea_rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_LEVEL_sum
pass ea_wr TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
# calibration: the same access pattern with a known byte count, from HBM (2-GiB table) and from the Infinity Cache (192 MiB)
for T in hbm:4194304 mall:393216; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/cal_${T%%:*}_fetch -- python3 $R/tools/calibrate_gather.py ${T##*:} > $P/cal_${T%%:*}.log 2>&1
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_LEVEL_sum --kernel-trace --output-format csv -d $P/cal_${T%%:*}_ea -- python3 $R/tools/calibrate_gather.py ${T##*:} >> $P/cal_${T%%:*}.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --kernel-trace --output-format csv -d $P/cal_${T%%:*}_l2 -- python3 $R/tools/calibrate_gather.py ${T##*:} >> $P/cal_${T%%:*}.log 2>&1
  echo "calibration ${T%%:*} done"
done
python3 $R/tools/parse_pmc.py $P $R/gpurun_out/traffic.json $BATCH $SCALE $DIM $OPTION

#!/usr/bin/env bash
# World-size rehearsal of the multi-GPU path on ONE card (run on the GPU box from the repo root).  The boxes of this pool admit at most
# 6 processes on a card: (1) the drop-in CLI with one PROCESS per rank through real HIP IPC at the most ranks the box takes,
# `bin/Force2Vec -gpus 6 -samegpu 1`, output bytes against the 1-GPU run's; (2) world 8 (= F2V_PUSH_MAX_RANKS, the driver's scaling run)
# as 8 ENGINES of one process, tools/push_world_local.py.  One run each, no loop.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/world; rm -rf $O; mkdir -p $O/one $O/six $O/one6 $O/six6
export HSA_ENABLE_IPC_MODE_LEGACY=0
G=$R/tests/golden
[ -f /tmp/f2v_golden_pubmed.mtx ] || python3 -c "import gzip,shutil; shutil.copyfileobj(gzip.open('$G/pubmed.mtx.gz','rb'), open('/tmp/f2v_golden_pubmed.mtx','wb'))"
run() { # dir gpus input option batch
  ( cd $1 && timeout -k 10 300 $R/bin/Force2Vec -input $3 -output $1/ -iter 30 -batch $5 -dim 128 -option $4 -gpus $2 -samegpu 1 > $1/log.txt 2>&1; echo "exit $?" >> $1/log.txt )
}
run $O/one 1 $G/cora.mtx 5 256;  run $O/six 6 $G/cora.mtx 5 256
run $O/one6 1 /tmp/f2v_golden_pubmed.mtx 6 4096;  run $O/six6 6 /tmp/f2v_golden_pubmed.mtx 6 4096
for pair in "one six cora_option5_batch256" "one6 six6 pubmed_option6_batch4096"; do
  set -- $pair
  a=$(ls $O/$1/*.embd 2>/dev/null | head -1); b=$(ls $O/$2/*.embd 2>/dev/null | head -1)
  if [ -n "$a" ] && [ -n "$b" ] && cmp -s "$a" "$b"; then verdict="output bytes identical to the 1-GPU run's ($(md5sum < $b | cut -c1-12))"; else verdict="OUTPUT DIFFERS OR IS MISSING"; fi
  echo "bin/Force2Vec -gpus 6 -samegpu 1, $3, 30 iterations: $(tail -1 $O/$2/log.txt); $verdict; $(grep -h 'GPU epoch loop' $O/$2/log.txt | head -1)"
done
timeout -k 10 600 python3 $R/tools/push_world_local.py 8

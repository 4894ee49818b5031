#!/usr/bin/env bash
# Run ON THE GPU BOX: the drop-in CLI and the reference binary on BASELINE configs[0] / [1] (cora, batch 256, 1200 iterations), whole-process
# wall time (graph read + training + .embd written) and the training time each prints.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=/tmp/cli_cora; rm -rf $O; mkdir -p $O/ours $O/ref
M=$R/tests/golden/cora.mtx
for D in 16 128; do
  s=$(date +%s%N); $R/bin/Force2Vec -input $M -output $O/ours/ -iter 1200 -batch 256 -dim $D -nsamples 5 -lr 0.02 -option 5 > $O/ours_$D.log 2>&1; e=$(date +%s%N)
  echo "ours       D=$D: wall $(( (e - s) / 1000000 )) ms; $(grep -i -m1 'time' $O/ours_$D.log)"
  for T in 1 32; do
    if [ -x $R/oracle/_ref/Force2Vec ]; then
      s=$(date +%s%N); $R/oracle/_ref/Force2Vec -input $M -output $O/ref/ -iter 1200 -batch 256 -dim $D -nsamples 5 -lr 0.02 -option 5 -threads $T > $O/ref_${D}_$T.log 2>&1; e=$(date +%s%N)
      echo "reference  D=$D -threads $T: wall $(( (e - s) / 1000000 )) ms; $(grep -i -m1 'time' $O/ref_${D}_$T.log)"
    fi
  done
done

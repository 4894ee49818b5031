set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k384 -- python3 $R/bench.py --batch 384 --steps 10 --warmup 2 --no-cpu-baseline --extra-batches= --config5-scale 0 --config4 0 --cora 0 --sustained-s 0 --live-pmc 0 --option7 0 --verify-rows 0 > $O/bench_384_profiled.json 2> $O/bench_384.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k65536 -- python3 $R/bench.py --no-cpu-baseline --extra-batches= --config5-scale 0 --config4 0 --cora 0 --sustained-s 0 --live-pmc 0 --option7 0 --verify-rows 0 > $O/bench_65536_profiled.json 2> $O/bench_65536.err
cd $R
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.json

# usage (on the GPU box, from the repo root): bash tools/gather_mix_sweep.sh   -- bin/gather_mix built by: hipcc --offload-arch=gfx950 -O3 -o bin/gather_mix tools/src/gather_mix.hip
for v in 0 8 1 2 3; do timeout -k 10 100 bin/gather_mix 150 2048 4194304 $v 0 || exit 1; done
timeout -k 10 100 bin/gather_mix 2048 2048 4194304 0 0 || exit 1
timeout -k 10 100 bin/gather_mix 150 16384 4194304 0 0 || exit 1
for pad in 140000 70000 50000; do timeout -k 10 100 bin/gather_mix 150 2048 4194304 0 $pad || exit 1; done

#!/usr/bin/env bash
# AddressSanitizer + UBSan over the host side of libf2v (CPU only: GPU ASan is not available on this pool):
# MatrixMarket reader (threaded), binary CSR, .embd writer, rand() jump-ahead fill, push masks / shard bounds (threaded).
#   tools/sanitize_host.sh graph.mtx [more.mtx ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I$R/include -I$R/force2vec_amd/csrc -pthread \
    $R/tools/sanitize_host.cpp $R/force2vec_amd/csrc/f2v_host.cpp -o $T/drv
sed -i "s#/tmp/asan/#$T/#g" /dev/null 2>/dev/null || true
(cd $T && mkdir -p /tmp/asan && ASAN_OPTIONS=detect_leaks=1 ./drv "$@")
rm -rf $T

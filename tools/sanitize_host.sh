#!/usr/bin/env bash
# AddressSanitizer + UBSan over the host side of libf2v (CPU only: GPU ASan is not available on this pool):
# MatrixMarket reader (threaded), binary CSR, .embd writer, rand() jump-ahead fill, option 7's walk generation, push masks / shard bounds (threaded).
#   tools/sanitize_host.sh /abs/path/graph.mtx [more.mtx ...]      (scratch files go to /tmp/asan)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/asan
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I$R/include -I$R/force2vec_amd/csrc -pthread \
    $R/tools/sanitize_host.cpp $R/force2vec_amd/csrc/f2v_host.cpp -o /tmp/asan/drv
ASAN_OPTIONS=detect_leaks=1 /tmp/asan/drv "$@"

#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.
# Compiles the genuine reference (HipGraph/Force2Vec) from its sources where they lie
# under /root/reference into oracle/_ref/ (git-ignored, travels with gpurun).  It does
# NOT run the reference's Makefile; the flags are the ones that Makefile documents
# (Makefile:9-13): -g -fomit-frame-pointer -ffast-math -fopenmp -O3 -std=c++11 -DCPP,
# plus -mavx512f -mavx512dq -DAVX512=1 for the options 8-11 build.
# No reference source is copied into this repository.
set -euo pipefail
REF="${F2V_REFERENCE_DIR:-/root/reference}"
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
if [ ! -f "$REF/sample/algorithms.cpp" ]; then
  echo "build_ref: $REF not present; keeping whatever is already in $OUT" >&2
  exit 0
fi
mkdir -p "$OUT"
FLAGS="-g -fomit-frame-pointer -ffast-math -fopenmp -O3 -std=c++11 -DCPP -w"
INC="-I$REF -I$REF/sample -I$REF/Test"
build() {  # $1 = output name, $2.. = extra flags
  local name="$1"; shift
  if [ "$OUT/$name" -nt "$REF/sample/algorithms.cpp" ] && [ "$OUT/$name" -nt "${BASH_SOURCE[0]}" ]; then return; fi
  g++ $INC $FLAGS "$@" -o "$OUT/$name" "$REF/sample/algorithms.cpp" "$REF/Test/Force2Vec.cpp"
  echo "build_ref: built $OUT/$name"
}
build Force2Vec &
build Force2Vec_avx512 -mavx512f -mavx512dq -DAVX512=1 &
wait

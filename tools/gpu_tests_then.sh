#!/bin/bash
# usage: tools/gpu_tests_then.sh LOG CMD...   -- the -m gpu suite (bounded), then CMD unless the suite was killed at its bound
# (a GPU step that timed out is followed by no further GPU step)
log=$1; shift
timeout -k 10 1000 python -m pytest tests -m gpu -q > "$log" 2>&1
rc=$?
tail -4 "$log"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "suite killed at its bound: stopping"; exit $rc; fi
"$@"
rc2=$?
[ $rc -ne 0 ] && exit $rc
exit $rc2

#!/usr/bin/env python3
"""PMC calibration run: launches libf2v's gather_calibration_kernel (the step kernel's access pattern, every 512-byte row
of a table fetched exactly once per launch, in random order) so that the counters can be compared with a known byte
count (MI355X_MICROARCH.md, HBM section).  argv[1] = rows (default 4 Mi = a 2-GiB table, far beyond the 256-MiB
Infinity Cache; 393216 = 192 MiB: resident in it from the second launch on).  Uses the self-test build."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from force2vec_amd import _lib

ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 4 * 1024 * 1024
T = _lib.selftest_lib()
_lib.check(T.f2v_test_gather_calibration(0, ROWS, 6), T)
print("calibration: %d rows x 512 B = %d bytes per launch (+%d bytes of ids)" % (ROWS, ROWS * 512, ROWS * 4))

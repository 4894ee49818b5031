#!/usr/bin/env python3
"""PMC calibration run: launches libf2v's gather_calibration_kernel (the step kernel's access
pattern, every 512-byte row of a 2 GiB table fetched exactly once per launch) so that
FETCH_SIZE can be compared with a known byte count (MI355X_MICROARCH.md, HBM section)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from force2vec_amd import _lib

ROWS = 4 * 1024 * 1024
_lib.check(_lib.selftest_lib().f2v_test_gather_calibration(0, ROWS, 3), _lib.selftest_lib())
print("calibration: %d rows x 512 B = %d bytes per launch (+%d bytes of ids)" % (ROWS, ROWS * 512, ROWS * 4))

#!/usr/bin/env python3
"""Which sort paths does the GPU parity suite reach?  Runs tests/test_gpu_parity.py in this process on the
-DSMHIP_PHASE_PROF library (tools/phase_prof.sh build; SMHIP_LIB must point at it) and prints the part counters:
[20] workgroup LDS path, [21] workgroup in-memory bitonic path, [22] wave path 32-bit keys, [23] wave path
64-bit keys with <= 512 anchors (wide dense ranges), [24] wave path 513..1024 anchors."""
import ctypes
import os
import sys

import pytest

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
rc = pytest.main(["-q", "-x", "-m", "gpu", os.path.join(root, "tests", "test_gpu_parity.py")] + sys.argv[1:])
from sigmap_amd import _abi  # noqa: E402

buf = (ctypes.c_ulonglong * 32)()
_abi.hip_lib().smhip_debug_phase_prof(buf, 0)
print("parts sorted per path:", {k: int(buf[k]) for k in range(20, 25)}, "pytest rc", int(rc))

"""tools/fuzz_parity.py as a test: 40 fixed-seed rounds of randomised genomes (one to four contigs, repeats), noisy /
flat / stuck reads and random device parameters (hit caps down to 3, fast slots down to 4 or sized from the index
census, radius, walker occupancy, every search mode, the speculative route and its repeat path) -- records, PAF rows
and sDTW refinement against the oracle, zero mismatches.  The log of a run goes to gpurun_out/ when that exists."""
import io
import os
import sys
from contextlib import redirect_stdout

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_fuzz_parity_40_fixed_seed_rounds():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity
    buf = io.StringIO()
    with redirect_stdout(buf):
        rc = fuzz_parity.main(40, 1)
    log = buf.getvalue()
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "fuzz_test_40_rounds.log"), "w") as f:
            f.write(log)
    assert rc == 0 and "TOTAL mismatches: 0" in log, log[-4000:]
    assert log.count("seed ") == 40
    # the rounds did exercise both walks and the repeat path of the speculative route
    assert "walked=2" in log and "walked=0" in log

"""FAST5 ingestion (signal_batch.cc:69-151).  No GPU needed.  The reference's own
reader cannot be built here (signal_batch.cc includes slow5.h), so these tests pin the loader to
the behaviour read off the reference source: both layouts, group order, attribute-as-string parsing
(six significant digits for float attributes), pA conversion and the (30, 200) pA filter."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from sigmap_amd import _abi, pipeline, synth

pytestmark = pytest.mark.skipif(not _abi.host_lib().smhost_have_fast5(), reason="host library built without HDF5")


def _reads(n=5, seed=3):
    lm = synth.make_pore_model()
    genome = synth.make_genome(21, [60000])
    raw, _ = synth.simulate_reads(genome, lm, n, seed)
    return [np.ascontiguousarray(r) for r in raw]


@pytest.mark.parametrize("multi", [True, False])
def test_fast5_roundtrip_equals_sgr(tmp_path, multi):
    raws = _reads()
    raws[1] = np.concatenate([raws[1], np.array([3000, -2000, 100], np.int16)])  # samples outside (30, 200) pA
    ids = ["r%02d-%s" % (i, "abc"[i % 3]) for i in range(len(raws))]
    d5, ds = tmp_path / "f5", tmp_path / "sg"
    os.makedirs(d5 / "sub"); os.makedirs(ds)
    synth.write_fast5(str(d5 / "sub" / "batch_0.fast5"), raws, ids, multi_read=multi)   # found in a sub-directory
    synth.write_sgr(str(ds / "batch_0.sgr"), raws, ids)
    a, b = synth.load_signals(str(d5)), synth.load_signals(str(ds))
    assert len(a) == len(b) == len(raws)
    # groups are listed by name (H5_INDEX_NAME): "read_r00-a" < "read_r01-b" < ... / "Read_1" < "Read_2" ...
    assert [x[0] for x in a] == ids == [x[0] for x in b]
    for (ia, pa, dg, rg, of), (ib, pb, _, _, _), r in zip(a, b, raws):
        assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32))
        exp = po.add_signal(r, synth.DIGITISATION, synth.RANGE, synth.OFFSET)
        assert np.array_equal(pa.view(np.uint32), exp.view(np.uint32))
        assert (dg, rg, of) == (np.float32(synth.DIGITISATION), np.float32(synth.RANGE), np.float32(synth.OFFSET))
    assert len(a[1][1]) == len(a[0][1])  # the three out-of-range samples were dropped


def test_fast5_float_attributes_keep_six_significant_digits(tmp_path):
    """hdf5_tools.hpp:1124-1141 prints a float attribute with operator<< (precision 6) before
    signal_batch.cc:125-131 parses it with atof: range 1402.882324 is used as 1402.88."""
    raws = _reads(2)
    p = str(tmp_path / "x.fast5")
    synth.write_fast5(p, raws, multi_read=True, digitisation=8192.0, rng=1402.882324, offset=6.0)
    got = synth.load_signals(p)     # a single file instead of a directory
    assert got[0][3] == np.float32(1402.88) and got[0][2] == np.float32(8192.0) and got[0][4] == np.float32(6.0)
    exp = po.add_signal(raws[0], 8192.0, float(np.float32(1402.88)), 6.0)
    assert np.array_equal(got[0][1].view(np.uint32), exp.view(np.uint32))
    other = po.add_signal(raws[0], 8192.0, 1402.882324, 6.0)
    assert not np.array_equal(exp.view(np.uint32), other.view(np.uint32))   # the rounding is observable


def test_fast5_mixed_directory_order(tmp_path):
    raws = _reads(4)
    synth.write_fast5(str(tmp_path / "b.fast5"), raws[:2], ["b0", "b1"], multi_read=True)
    synth.write_fast5(str(tmp_path / "a.fast5"), raws[2:3], ["a0"], multi_read=False)
    synth.write_sgr(str(tmp_path / "c.sgr"), raws[3:], ["c0"])
    assert [x[0] for x in synth.load_signals(str(tmp_path))] == ["a0", "b0", "b1", "c0"]   # FAST5 (sorted paths), then .sgr

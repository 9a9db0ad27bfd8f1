import os
import subprocess
import sys

import numpy as np
import pytest

# the tests drive rarely taken routes through development knobs (smhip_dev_knob), which a context only accepts
# when it was created under SMHIP_DEV=1
os.environ["SMHIP_DEV"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # make sure the CPU-side libraries exist (cheap no-op when already built)
    need = [os.path.join(ROOT, "sigmap_amd", "lib", "libsigmap_host.so"),
            os.path.join(ROOT, "oracle", "libsigmap_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.run(["make", "-C", ROOT, "sigmap_amd/lib/libsigmap_host.so"], check=True,
                       stdout=subprocess.DEVNULL)
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libsigmap_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="session")
def small_world():
    """Seeded 3-contig 300 kb genome, pore model, index (host build), 48 reads in pA (the ORACLE's
    AddSignal conversion, so_add_signal: the checker's inputs never pass through product code)."""
    from oracle import pyoracle as po
    from sigmap_amd import pipeline, synth
    genome = synth.make_genome(7, [150000, 100000, 50000])
    lm = synth.make_pore_model()
    raw, truth = synth.simulate_reads(genome, lm, 48, 11)
    ix = pipeline.HostIndex.build(genome, lm, threads=4)
    pa = [po.add_signal(r, synth.DIGITISATION, synth.RANGE, synth.OFFSET) for r in raw]
    return dict(genome=genome, lm=lm, raw=raw, truth=truth, ix=ix, pa=pa,
                ref_lens=np.array([len(g) for g in genome], dtype=np.uint32))


@pytest.fixture(scope="session")
def oracle_tree(small_world):
    from oracle import pyoracle as po
    return po.OracleTree(small_world["ix"].values.copy())


@pytest.fixture(scope="session")
def device(small_world):
    from sigmap_amd import pipeline
    dev = pipeline.DeviceMapper(0)
    dev.upload_index(small_world["ix"])
    return dev

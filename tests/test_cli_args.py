"""The `sigmap` CLI's option parser (no GPU needed: parsing happens before any device call)."""
import os
import subprocess

import pytest

from conftest import ROOT

BIN = os.path.join(ROOT, "sigmap_amd", "bin", "sigmap")


def _run(*args):
    if not os.path.exists(BIN):
        pytest.fail("sigmap_amd/bin/sigmap missing: run `make cli`")
    return subprocess.run([BIN, *args], capture_output=True, text=True, timeout=60)


def test_unknown_option_is_an_error_like_cxxopts():
    """The reference parses with cxxopts (sigmap.cc:1378), which throws on an option it does not know;
    a typo such as --serach-radius must not silently map with the default."""
    for bad in (["--serach-radius", "0.1", "-m"], ["-m", "--no-such-flag"], ["--step-sizes=3", "-m"], ["-q"]):
        r = _run(*bad)
        assert r.returncode != 0, bad
        assert "does not exist" in r.stderr or "Unknown argument" in r.stderr, (bad, r.stderr)


def test_missing_value_and_missing_inputs_are_errors():
    r = _run("-m", "--gpus")
    assert r.returncode != 0 and "missing an argument" in r.stderr
    r = _run("-m", "--search-radius", "0.1")          # known option, but -m needs -r -p -x -s -o (sigmap.cc:1456-1492)
    assert r.returncode != 0 and "No reference file specified" in r.stderr
    r = _run("-i", "-d", "7", "-r", "x", "-p", "y", "-o", "z")
    assert r.returncode != 0 and "dimension 6 only" in r.stderr


def test_help_lists_the_reference_flags():
    r = _run("-h")
    assert r.returncode == 0
    for flag in ("--build-index", "--map", "--step-size", "--search-radius", "--max-num-chunks", "--min-num-anchors",
                 "--min-num-anchors-output", "--stop-mapping", "--stop-mapping-output", "--stop-mapping-mean",
                 "--stop-mapping-mean-output", "--ref", "--pore-model", "--ref-index", "--sig-dir", "--output", "--no-warmup"):
        assert flag in r.stderr + r.stdout, flag

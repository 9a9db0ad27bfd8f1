"""The C-ABI libraries load and export every symbol include/*.h declares (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from sigmap_amd import _abi


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s_[a-z0-9_]+)\s*\(" % prefix, text)))


def test_host_library_exports_every_declared_symbol():
    L = C.CDLL(_abi.HOST_LIB_PATH)
    names = _declared("sigmap_host.h", "smhost")
    assert set(names) == set(_abi.HOST_SYMBOLS)
    for n in names:
        assert hasattr(L, n), n


def test_hip_library_exports_every_declared_symbol():
    if not os.path.exists(_abi.HIP_LIB_PATH):
        pytest.fail("libsigmap_hip.so missing: build it with `make` (hipcc cross-compiles without a GPU)")
    L = C.CDLL(_abi.HIP_LIB_PATH)
    names = _declared("sigmap_hip.h", "smhip")
    assert set(names) == set(_abi.HIP_SYMBOLS)
    for n in names:
        assert hasattr(L, n), n


def test_default_params_match_reference_constants():
    p = _abi.Params()
    _abi.hip_lib().smhip_default_params(C.byref(p))
    assert (p.window_length1, p.window_length2) == (3, 6)                      # event.h:31-37
    assert abs(p.threshold1 - 4.30265) < 1e-6 and abs(p.threshold2 - 2.57058) < 1e-6
    assert p.compress_delta == 0.1 and p.min_num_events == 50                  # sigmap.cc:1075, :660
    assert abs(p.search_radius - 0.08) < 1e-7 and p.step_size == 2             # sigmap.cc:1380-1384
    assert (p.max_gap_length, p.max_target_gap_length, p.chaining_band_length) == (2000, 5000, 5000)
    assert (p.max_num_skips, p.min_num_anchors, p.num_best_chains, p.num_nearest_points) == (25, 2, 3, 5000)
    assert p.min_chaining_score == 10.0                                        # spatial_index.cc:284-291


def test_no_gpu_fails_loudly():
    """Without a GPU the product must raise, not fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sigmap_amd import pipeline
    with pytest.raises(RuntimeError):
        pipeline.DeviceMapper(0)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sigmap_amd")):
        for f in files:
            if f.endswith((".py", ".cc", ".h", ".hip")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in src and "sigmap_oracle" not in src and "oracle/" not in src, f

"""The host-side exhaustive checks of the two division-free forms the device code uses (tools/micro/): compiled and run
here on a part of their range that takes a few seconds -- the chain classifier's start-score factor for every
non-negative distance below 8 radii at the default radius, the t-statistics' x / w for the two default window lengths over
every float of magnitude 2^-100 .. 2^100.  (The device's own code is swept over every float by the GPU test
test_division_free_forms_on_the_device_over_every_float.)"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, src, extra=()):
    exe = str(tmp_path / (os.path.basename(src)[:-2]))
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", *extra, os.path.join(ROOT, "tools", "micro", src), "-o", exe, "-lm"],
                   check=True)
    return exe


def test_start_coef_form_matches_the_division_for_every_distance(tmp_path):
    r = subprocess.run([_build(tmp_path, "start_coef_check.c"), "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches 0" in r.stdout


def test_reciprocal_division_by_the_default_windows_matches(tmp_path):
    exe = _build(tmp_path, "div_by_window_check.c", ("-include", "stdlib.h"))   # (libm fmaf: no -mfma, any host)
    for w in ("3", "6"):
        r = subprocess.run([exe, w, w], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "mismatches 0" in r.stdout


def test_blocking_sim_reproduces_the_quad_walks_record_count(tmp_path):
    """tools/micro/blocking_sim.cc (how many tree records a set-walk query fetches under other record packings: the
    experiment behind the round-5 decision on oct / pack records) builds and runs on a small real tree: the set walk's entry
    rule on the nanoflann-identical tree of 20 000 points, heads at even depth = what BuildQuads marks; a packing with
    more slots per record never fetches more records than one with fewer (the DP's optimum is monotone in K)."""
    import re
    import numpy as np
    from sigmap_amd import pipeline
    rng = np.random.default_rng(3)
    vals = rng.normal(size=20005).astype(np.float32)
    ix = pipeline.HostIndex.from_points(np.arange(len(vals), dtype=np.uint64) << 1, vals, threads=1)
    nodes = np.ascontiguousarray(ix.nodes)
    with open(tmp_path / "nodes.bin", "wb") as f:
        f.write(np.int32(len(nodes)).tobytes()); f.write(nodes.tobytes())
    for name, arr, dt in (("values", ix.values, np.float32), ("vind", ix.vind, np.uint32)):
        a = np.ascontiguousarray(arr, dtype=dt)
        with open(tmp_path / (name + ".bin"), "wb") as f:
            f.write(np.int64(len(a)).tobytes()); f.write(a.tobytes())
    qs = np.array([vals[i:i + 6] + rng.normal(scale=0.05, size=6).astype(np.float32) for i in range(0, 20000, 10)], np.float32)
    with open(tmp_path / "queries.bin", "wb") as f:
        f.write(np.int64(len(qs)).tobytes()); f.write(qs.tobytes())
    exe = str(tmp_path / "blocking_sim")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tools", "micro", "blocking_sim.cc"), "-o", exe], check=True)
    r = subprocess.run([exe, str(tmp_path / "nodes.bin"), str(tmp_path / "values.bin"), str(tmp_path / "vind.bin"),
                        str(tmp_path / "queries.bin"), "0.08"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rec = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"^(.*?)\s+records/query\s+([0-9.]+)", r.stdout, re.M)}
    assert rec["even depth (current quads)"] > 1.0
    for w in ("geo", "emp"):
        ks = [rec["dp-%s K=%d" % (w, k)] for k in (4, 6, 7, 8)]
        assert ks == sorted(ks, reverse=True), ks
    assert rec["dp-geo K=8"] < rec["even depth (current quads)"]

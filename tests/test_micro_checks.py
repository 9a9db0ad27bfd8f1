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

"""D5: fast_dtw.cc::DTW() + generate_path + print_alignment (fast_dtw.cc:51-140,160-209).

CPU part: the C oracle (oracle/sigmap_oracle.c so_dtw_window / so_print_alignment) against a direct
pure-Python transcription of the reference's data flow (window as lists of coordinates, a dict from
coordinate to window index, std::min_element's first minimum) on small cases, known answers for the
CIGAR printer, and the survey's observation that on a planted match the windowed DTW() equals
Sigmap::sDTW at full window and at bands 8 / 32 / 64.  GPU part: smhip_dtw_path bit-exact against the
oracle.  (The oracle itself is parity-unpinned for this row: fast_dtw.cc cannot be compiled here.)"""
import numpy as np
import pytest

from conftest import bits
from oracle import pyoracle as po

F32 = np.float32
FMAX = np.finfo(np.float32).max


def _py_dtw(target, query, window):
    """fast_dtw.cc:74-140 with a window given as rows of (target_coord, query_coord)."""
    T, Q = len(target), len(query)
    index = {}
    for i, row in enumerate(window):
        for j, c in enumerate(row):
            index[c] = (i, j)
    codes = [[0] * len(r) for r in window]
    prev = [F32(FMAX)] * (Q + 1)
    prev[0] = F32(0)
    best, end = F32(FMAX), -1
    qpos = None
    for i, row in enumerate(window):
        cur = [F32(FMAX)] * (Q + 1)
        cur[0] = F32(0)
        for j, (tc, qc) in enumerate(row):
            tpos, qpos = tc + 1, qc + 1
            cost = F32(abs(F32(target[tc]) - F32(query[qc])))
            with np.errstate(over="ignore"):
                cand = [F32(prev[qpos - 1] + cost), F32(cur[qpos - 1] + cost), F32(cur[qpos - 1] + F32(2)),
                        F32(prev[qpos] + F32(2))]
            cand = [c if np.isfinite(c) else F32(FMAX) for c in cand]   # FLT_MAX + small == FLT_MAX in f32
            arg = min(range(4), key=lambda k: (cand[k], k))
            cur[qpos] = cand[arg]
            codes[i][j] = arg
        if qpos == Q and cur[qpos] < best:
            best, end = cur[qpos], i
        prev = cur
    if end < 0:
        return best, -1, [], -1
    row, col = end, len(window[end]) - 1
    coord = window[row][col]
    path = []
    qshift, tshift = [-1, -1, -1, 0], [-1, 0, 0, -1]
    while coord[1] != 0:
        coord = window[row][col]
        c = codes[row][col]
        path.append(c)
        coord = (coord[0] + tshift[c], coord[1] + qshift[c])
        row, col = index[coord]
    path.append(codes[row][col])
    start_t = window[row][col][0]
    path.reverse()
    return best, window[end][0][0], path, start_t


def _py_print_alignment(path):
    """fast_dtw.cc:160-209."""
    flags = "MMID"
    if not path:
        return ""
    per_base = []
    same = 1
    if path[0] == 3:
        per_base.append("1D")
        last = 3
    else:
        last = 1 if path[0] == 0 else 2
    per_base.append("")
    for i in range(1, len(path)):
        f = path[i]
        if f in (1, 2):
            if last == f:
                same += 1
            else:
                per_base[-1] += str(same) + flags[last]
                same = 1
                last = f
        else:
            per_base[-1] += str(same) + flags[last]
            if f == 0:
                last = 1
            elif f == 3:
                last = 3
            if i != len(path) - 1:
                per_base.append("")
                same = 1
    return "".join("(" + b + ")" for b in per_base)


def _band(T, Q, band, slope=None):
    """per-row 1-based inclusive column intervals of a band around the line q = 1 + slope * t."""
    slope = (Q - 1) / max(T - 1, 1) if slope is None else slope
    centre = 1 + np.floor(np.arange(T) * slope).astype(np.int64)
    lo = np.clip(centre - band // 2, 1, Q).astype(np.uint32)
    hi = np.clip(centre + band // 2, 1, Q).astype(np.uint32)
    return lo, hi


def _planted(rng, T=700, Q=420, start=130, noise=0.05):
    """z-scored-looking target; the query is a noisy copy of target[start : start + Q] in which a few
    positions are repeated (one base, several events: the over-segmentation real reads show).  No
    target position is dropped: a dropped base is where the two recurrences differ by design (DTW()
    charges a flat 2 for it, Sigmap::sDTW the cell cost)."""
    target = rng.standard_normal(T).astype(np.float32)
    q = target[start:start + Q].copy()
    dup = rng.choice(len(q), 12, replace=False)
    q = np.insert(q, dup, q[dup])[:Q]
    return target, (q + noise * rng.standard_normal(len(q))).astype(np.float32)


def test_print_alignment_known_answers():
    cases = {
        (): "",
        (0,): "()",
        (3,): "(1D)()",
        (0, 0): "(1M)",                # the second base is closed without being opened (reference quirk)
        (0, 1, 1, 0, 2, 0): "(3M)(1M1I)",
        (0, 1, 2, 2, 1): "(2M2I)",     # the run in progress at the end is not written
        (3, 0, 1, 3, 0, 0): "(1D)(1D)(2M)(1D)(1M)",
        (2, 2, 0, 1): "(2I)()",
    }
    for path, want in cases.items():
        assert _py_print_alignment(list(path)) == want, path
        assert po.print_alignment(np.array(path, np.uint8)) == want, path
    rng = np.random.default_rng(0)
    for _ in range(300):
        path = rng.integers(0, 4, size=rng.integers(1, 40)).astype(np.uint8)
        assert po.print_alignment(path) == _py_print_alignment(list(path))


def test_host_print_alignment_matches_oracle():
    from sigmap_amd import pipeline
    rng = np.random.default_rng(1)
    for _ in range(300):
        path = rng.integers(0, 4, size=rng.integers(0, 60)).astype(np.uint8)
        assert pipeline.print_alignment(path) == po.print_alignment(path)


def test_oracle_dtw_window_matches_direct_transcription():
    rng = np.random.default_rng(2)
    for case in range(40):
        T, Q = int(rng.integers(3, 40)), int(rng.integers(1, 25))
        target = rng.standard_normal(T).astype(np.float32) * F32(1.5)
        query = rng.standard_normal(Q).astype(np.float32) * F32(1.5)
        if case % 3 == 0:
            lo, hi = np.ones(T, np.uint32), np.full(T, Q, np.uint32)   # full window
        else:
            lo, hi = _band(T, Q, int(rng.integers(2, 12)))
        window = [[(t, q - 1) for q in range(int(lo[t]), int(hi[t]) + 1)] for t in range(T)]
        d, e, path, st = _py_dtw(target, query, window)
        od, oe, opath, ost = po.dtw_window(target, query, lo, hi)
        assert bits(od) == bits(d) and oe == e, case
        if e >= 0:
            assert list(opath) == path and ost == st, case


def test_planted_match_full_window_and_bands():
    """on a planted match DTW() with the full window and with bands 8 / 32 / 64
    around the true diagonal returns Sigmap::sDTW's distance.  Exactly so for a noise-free planted copy
    (distance 0, path of one-to-one and one-base-to-several-signals moves only, starting at the planted
    target position); with noise the two recurrences may part in the last digits (sDTW may drop a base
    at the cell's cost where DTW() pays 2), so there the check is closeness plus band-independence."""
    rng = np.random.default_rng(3)
    for noise in (0.0, 0.05):
        target, query = _planted(rng, noise=noise)
        T, Q = len(target), len(query)
        want, want_end = po.sdtw(target, query)
        full = po.dtw_window(target, query, np.ones(T, np.uint32), np.full(T, Q, np.uint32))
        if noise == 0.0:
            assert want == 0.0 and full[0] == 0.0 and full[1] == want_end - 1   # sDTW's end is 1-based
            assert set(full[2].tolist()) <= {0, 1} and full[3] == 130
        else:
            assert abs(full[0] - want) <= 1e-3 * want and abs(full[1] - (want_end - 1)) <= 2
        for band in (8, 32, 64):
            # band around the planted diagonal: row t maps to column t - 130 (+ the repeats so far)
            centre = np.arange(T) - 130 + 1 + 6
            lo = np.clip(centre - band // 2, 1, Q).astype(np.uint32)
            hi = np.clip(centre + band // 2, 1, Q).astype(np.uint32)
            d, e, path, st = po.dtw_window(target, query, lo, hi)
            if band >= 32:   # the 12 repeats shift the diagonal by up to 12 columns: inside these bands
                assert bits(d) == bits(full[0]) and e == full[1] and st == full[3], (noise, band)
                assert np.array_equal(path, full[2])
                cig = po.print_alignment(path)
                assert cig.startswith("(") and cig.count("(") >= Q - 30
            else:
                assert d >= full[0]   # a narrower window can only cost more


# ------------------------------------------------------------------ GPU

@pytest.mark.gpu
def test_gpu_dtw_path_matches_oracle(device):
    from sigmap_amd import pipeline
    rng = np.random.default_rng(4)
    problems = []
    target, query = _planted(rng)
    T, Q = len(target), len(query)
    problems.append((target, query, np.ones(T, np.uint32), np.full(T, Q, np.uint32)))
    for band in (8, 32, 64):
        centre = np.arange(T) - 130 + 1 + 6
        problems.append((target, query, np.clip(centre - band // 2, 1, Q).astype(np.uint32),
                         np.clip(centre + band // 2, 1, Q).astype(np.uint32)))
    for case in range(60):     # ragged random problems, incl. rows whose window misses the last column
        T, Q = int(rng.integers(1, 300)), int(rng.integers(1, 120))
        t = (rng.standard_normal(T) * 1.5).astype(np.float32)
        q = (rng.standard_normal(Q) * 1.5).astype(np.float32)
        if case % 4 == 0:
            lo, hi = np.ones(T, np.uint32), np.full(T, Q, np.uint32)
        else:
            lo, hi = _band(T, Q, int(rng.integers(2, 40)), slope=float(rng.uniform(0.2, 1.2)))
        problems.append((t, q, lo, hi))
    t_off = np.concatenate([[0], np.cumsum([len(p[0]) for p in problems])]).astype(np.uint64)
    q_off = np.concatenate([[0], np.cumsum([len(p[1]) for p in problems])]).astype(np.uint64)
    dist, end, paths, start = device.dtw_path(np.concatenate([p[0] for p in problems]), t_off,
                                              np.concatenate([p[1] for p in problems]), q_off,
                                              np.concatenate([p[2] for p in problems]),
                                              np.concatenate([p[3] for p in problems]))
    finite = 0
    for i, (t, q, lo, hi) in enumerate(problems):
        od, oe, opath, ost = po.dtw_window(t, q, lo, hi)
        assert bits(dist[i]) == bits(od) and end[i] == oe, i
        if oe >= 0:
            finite += 1
            assert np.array_equal(paths[i], opath) and start[i] == ost, i
            assert pipeline.print_alignment(paths[i]) == po.print_alignment(opath)
        else:
            assert len(paths[i]) == 0 and start[i] == -1
    assert finite >= 40
    # the planted match: the bands that contain the optimal path give the full window's answer
    assert bits(dist[2]) == bits(dist[0]) and bits(dist[3]) == bits(dist[0]) and dist[1] >= dist[0]


@pytest.mark.gpu
@pytest.mark.parametrize("band", [8, 32, 64])
def test_gpu_refine_path_matches_oracle(device, small_world, band):
    """smhip_refine_path_run: DTW() with path over chain 0's window (window and band exactly as
    smhip_refine_run states them) == the oracle on the same window: distance, end, start, path codes,
    CIGAR.  The mapping records are not touched."""
    from sigmap_amd import pipeline
    genome, lm, pa = small_world["genome"], small_world["lm"], small_world["pa"][:24]
    sig, offs = pipeline.reference_signals(genome, lm)
    device.upload_reference(sig, offs)
    offsets = np.concatenate([[0], np.cumsum([len(x) for x in pa])]).astype(np.uint32)
    samples = np.concatenate(pa)
    device.upload_batch(samples, offsets)
    device.run(3); device.sync()
    before = device.download()
    pad = 32
    device.refine_path(band=band, pad=pad)
    dist, end, start, paths = device.refine_path_download()
    assert before.tobytes() == device.download().tobytes()
    feat, _, nf, _, _ = device.events(samples, offsets)
    checked = 0
    for i in range(len(pa)):
        r = before[i]
        if r["n_chains"] == 0:
            assert end[i] == -1 and len(paths[i]) == 0
            continue
        c0 = r["chains"][0]
        strand = 0 if c0["direction"] == 1 else 1
        so, se = int(offs[2 * c0["ref"] + strand]), int(offs[2 * c0["ref"] + strand + 1])
        s0, e0 = int(c0["start"]), int(c0["end"])
        t0 = max(0, s0 - pad); t1 = min(se - so, e0 + 6 + pad)
        target = sig[so + t0:so + t1]
        q = feat[i, :nf[i]]
        qa, qb = int(r["c0_last_qpos"]), int(r["c0_first_qpos"])
        den = max(1, e0 - s0)
        ta = np.arange(t0, t1, dtype=np.int64)
        centre = 1 + qa + ((ta - s0) * (qb - qa)) // den       # floor division
        lo = np.clip(centre - band // 2, 1, None)
        hi = np.minimum(centre + band // 2, len(q))
        hi = np.where(hi < lo, lo - 1, hi)                      # empty rows
        d, e, opath, ost = po.dtw_window(target, q, lo.astype(np.uint32), np.maximum(hi, 0).astype(np.uint32))
        if e < 0:
            assert end[i] == -1
            continue
        assert bits(np.float32(dist[i])) == bits(np.float32(d)), (i, dist[i], d)
        assert end[i] == t0 + e and start[i] == t0 + ost
        assert np.array_equal(paths[i], opath)
        assert pipeline.print_alignment(paths[i]) == po.print_alignment(opath)
        checked += 1
    assert checked >= (12 if band >= 32 else 1)


@pytest.mark.gpu
def test_gpu_refine_path_beyond_2_32_window_cells(device, small_world):
    """One batch whose windows hold more than 2^32 cells (1 250 chunks x ~8 000 rows x ~450 columns: a
    large pad and a band wider than the query, i.e. what long carried chains or --sdtw-band 512 reach at
    the CLI's batch sizes).  The per-row cell offsets are 64-bit: the chunks whose cells lie beyond the
    4 GiB mark -- where 32-bit offsets would alias the first chunks' cells -- give the oracle's distance,
    end, start and path like the first ones."""
    from sigmap_amd import pipeline
    genome, lm = small_world["genome"], small_world["lm"]
    base = small_world["pa"][:25]
    reps = 50
    pa = base * reps
    sig, offs = pipeline.reference_signals(genome, lm)
    device.upload_reference(sig, offs)
    offsets = np.concatenate([[0], np.cumsum([len(x) for x in pa])]).astype(np.uint32)
    samples = np.concatenate(pa)
    device.upload_batch(samples, offsets)
    device.run(3); device.sync()
    res = device.download()
    pad, band = 3800, 20000
    device.refine_path(band=band, pad=pad)
    dist, end, start, paths = device.refine_path_download()
    feat, _, nf, _, _ = device.events(samples[:offsets[len(base)]], offsets[:len(base) + 1])

    def window(i):
        r = res[i]
        c0 = r["chains"][0]
        strand = 0 if c0["direction"] == 1 else 1
        so, se = int(offs[2 * c0["ref"] + strand]), int(offs[2 * c0["ref"] + strand + 1])
        t0 = max(0, int(c0["start"]) - pad); t1 = min(se - so, int(c0["end"]) + 6 + pad)
        return so, t0, t1

    cells, cells_before = 0, {}
    for i in range(len(pa)):
        cells_before[i] = cells
        if res[i]["n_chains"] and end[i] != -1:
            so, t0, t1 = window(i)
            cells += (t1 - t0) * int(nf[i % len(base)])
    assert cells > 2 ** 32, cells
    beyond = [i for i in range(len(pa)) if cells_before[i] > 2 ** 32 and res[i]["n_chains"] and end[i] >= 0]
    assert len(beyond) >= 3
    for i in [0, 1, len(pa) // 2] + beyond[:2] + beyond[-2:]:
        if res[i]["n_chains"] == 0:
            continue
        so, t0, t1 = window(i)
        if t1 - t0 > 8192:
            assert end[i] == -1
            continue
        q = feat[i % len(base), :nf[i % len(base)]]
        T = t1 - t0
        d, e, opath, ost = po.dtw_window(sig[so + t0:so + t1], q, np.ones(T, np.uint32), np.full(T, len(q), np.uint32))
        assert bits(np.float32(dist[i])) == bits(np.float32(d)), (i, dist[i], d)
        assert end[i] == t0 + e and start[i] == t0 + ost and np.array_equal(paths[i], opath), i
    # replicas of one read give one answer wherever their cells lie
    for i in range(len(base), len(pa)):
        assert bits(np.float32(dist[i])) == bits(np.float32(dist[i % len(base)])) and end[i] == end[i % len(base)]
        assert np.array_equal(paths[i], paths[i % len(base)])

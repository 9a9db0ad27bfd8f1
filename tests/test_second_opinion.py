"""Second opinions for the oracle functions whose reference translation units cannot be compiled in this image
(sigmap.cc includes <slow5/slow5.h>): direct Python transcriptions, written from the reference source and not from
oracle/sigmap_oracle.c, of

  * Sigmap::sDTW                        (sigmap.cc:1259-1291)   vs  so_sdtw   (and so_sdtw_banded with a full band),
  * Sigmap::GenerateMADNormalizedSignal (sigmap.cc:1100-1129)   vs  so_mad_normalize,
  * Sigmap::GeneratePeaks               (sigmap.cc:1176-1196)   vs  so_generate_peaks.

They do NOT pin the oracle (both are restatements by the same project); they catch transcription slips: a row /
column swap, an off-by-one in the end position, an averaged median, a `>` for a `>=`.  Every arithmetic step is
done in numpy float32, in the reference's order.  CPU only."""
import numpy as np

from conftest import bits
from oracle import pyoracle as po

F32 = np.float32
FLT_MAX = np.finfo(np.float32).max


def _py_sdtw(target, query):
    """sigmap.cc:1259-1291, line by line."""
    tl, ql = len(target), len(query)
    min_dtw_distance = FLT_MAX
    mapping_end_position = -1
    previous_row = [FLT_MAX] * (ql + 1)
    previous_row[0] = F32(0)
    current_row = [F32(0)] * (ql + 1)
    for target_position in range(1, tl + 1):
        current_row[0] = F32(0)
        for query_position in range(1, ql + 1):
            cost = F32(abs(F32(target[target_position - 1] - query[query_position - 1])))
            m = min(previous_row[query_position - 1], previous_row[query_position], current_row[query_position - 1])
            with np.errstate(over="ignore"):
                current_row[query_position] = F32(cost + m)      # FLT_MAX + cost rounds back to FLT_MAX or to inf
        if current_row[ql] < min_dtw_distance:
            min_dtw_distance = current_row[ql]
            mapping_end_position = target_position
        current_row, previous_row = previous_row, current_row
    return F32(min_dtw_distance), mapping_end_position


def test_sdtw_matches_direct_transcription():
    rng = np.random.default_rng(11)
    for case in range(60):
        T, Q = int(rng.integers(1, 70)), int(rng.integers(1, 40))
        target = (rng.standard_normal(T) * 1.5).astype(np.float32)
        query = (rng.standard_normal(Q) * 1.5).astype(np.float32)
        if case % 4 == 0 and T > Q + 2:                          # a planted copy: the minimum is an interior row
            s = int(rng.integers(0, T - Q))
            target[s:s + Q] = query + (rng.standard_normal(Q) * 0.01).astype(np.float32)
        if case % 7 == 0:                                        # ties: quantised values, the FIRST minimal row wins
            target = np.round(target).astype(np.float32)
            query = np.round(query).astype(np.float32)
        d, e = _py_sdtw(target, query)
        od, oe = po.sdtw(target, query)
        assert bits(od) == bits(d) and oe == e, (case, T, Q, od, d, oe, e)
        # the banded form with every row's band = the whole query is the same recurrence
        bd, be = po.sdtw_banded(target, query, np.ones(T, np.uint32), np.full(T, Q, np.uint32))
        assert bits(bd) == bits(d) and be == e, case


def test_sdtw_edge_cases():
    # the reference reports row 1 when every row ties, -1 and FLT_MAX when nothing beats FLT_MAX (empty target)
    q = np.array([1.0, 2.0], np.float32)
    d, e = _py_sdtw(np.zeros(0, np.float32), q)
    od, oe = po.sdtw(np.zeros(0, np.float32), q)
    assert (bits(od), oe) == (bits(d), e) == (bits(FLT_MAX), -1)
    t = np.full(5, 3.0, np.float32)
    d, e = _py_sdtw(t, np.array([3.0], np.float32))
    od, oe = po.sdtw(t, np.array([3.0], np.float32))
    assert (bits(od), oe) == (bits(d), e) and e == 1 and d == 0
    # one target value against a longer query: only the left neighbour is finite
    d, e = _py_sdtw(np.array([0.5], np.float32), np.array([0.25, 1.0, -2.0], np.float32))
    od, oe = po.sdtw(np.array([0.5], np.float32), np.array([0.25, 1.0, -2.0], np.float32))
    assert (bits(od), oe) == (bits(d), e)


def _py_mad(x):
    """sigmap.cc:1100-1129: nth_element at n / 2 is the n/2-th order statistic (0-based) -- the UPPER median for even
    n, never an average --, the same for the absolute deviations; then (x - median) / MAD in float."""
    n = len(x)
    median = np.sort(x)[n // 2]
    dev = np.abs((x - median).astype(np.float32)).astype(np.float32)
    mad = np.sort(dev)[n // 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        out = ((x - median).astype(np.float32) / mad).astype(np.float32)
    return out, F32(mad)


def test_mad_normalize_matches_direct_transcription():
    rng = np.random.default_rng(12)
    for case in range(40):
        n = int(rng.integers(1, 400))
        x = (90 + 12 * rng.standard_normal(n)).astype(np.float32)
        if case % 5 == 0:
            x = np.round(x).astype(np.float32)                   # repeated values around the median
        if case % 9 == 0 and n > 4:
            x[: n // 2 + 1] = x[0]                               # MAD = 0: the reference divides by zero (inf / nan)
        want, wmad = _py_mad(x)
        got, mad = po.mad_normalize(x)
        assert bits(F32(mad)) == bits(wmad), (case, n)
        assert np.array_equal(bits(got), bits(want)), (case, n)
    for n in (1, 2, 3, 4):                                       # even n: element n / 2 of the sorted values
        x = np.arange(n, dtype=np.float32)[::-1].copy() * F32(1.5)
        want, wmad = _py_mad(x)
        got, mad = po.mad_normalize(x)
        assert bits(F32(mad)) == bits(wmad) and np.array_equal(bits(got), bits(want)), n


def _py_generate_peaks(x, selective):
    """sigmap.cc:1176-1196."""
    peaks, pos = [], []
    previous_valley = x[0]
    previous_peak = x[0]
    for i in range(1, len(x) - 1):
        if x[i] > x[i - 1] and x[i] >= x[i + 1] and x[i] >= F32(previous_valley + selective):
            peaks.append(x[i]); pos.append(i)
            previous_peak = x[i]
        elif x[i] < x[i - 1] and x[i] <= x[i + 1] and x[i] <= F32(previous_peak - selective):
            peaks.append(x[i]); pos.append(i)
            previous_valley = x[i]
    return np.array(peaks, np.float32), np.array(pos, np.uint64)


def test_generate_peaks_matches_direct_transcription():
    rng = np.random.default_rng(13)
    for case in range(60):
        n = int(rng.integers(2, 600))
        x = np.cumsum(rng.standard_normal(n)).astype(np.float32)
        if case % 3 == 0:
            x = np.round(x * 2).astype(np.float32) / F32(2)      # plateaus: `>` on the left, `>=` on the right
        selective = F32(rng.choice([0.0, 0.25, 1.0, 3.0]))
        wpk, wpos = _py_generate_peaks(x, selective)
        pk, pos = po.generate_peaks(x, selective)
        assert np.array_equal(np.asarray(pos, np.uint64), wpos), (case, n)
        assert np.array_equal(bits(np.asarray(pk, np.float32)), bits(wpk)), (case, n)
    # a zig-zag: every interior sample is an extremum; with a hysteresis above its amplitude nothing after the first
    z = np.tile(np.array([0.0, 1.0], np.float32), 20)
    for sel in (F32(0.5), F32(1.0), F32(1.5)):
        wpk, wpos = _py_generate_peaks(z, sel)
        pk, pos = po.generate_peaks(z, sel)
        assert np.array_equal(np.asarray(pos, np.uint64), wpos) and np.array_equal(bits(np.asarray(pk, np.float32)), bits(wpk))

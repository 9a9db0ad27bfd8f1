"""Parity tests proper: the HIP path (through the C ABI) against the oracle on the same seeded
inputs, against the reference goldens, and through size-independent properties at larger
sizes.  Bit-exact for every integer/index/f32 result.  Run with `-m gpu` on an MI355X."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, bits
from oracle import pyoracle as po
from sigmap_amd import _abi, pipeline, synth

pytestmark = pytest.mark.gpu


def _batch(pa_list):
    offsets = np.concatenate([[0], np.cumsum([len(x) for x in pa_list])]).astype(np.uint32)
    return np.concatenate(pa_list), offsets


# ------------------------------------------------------------------ events (A2-A8)

def test_events_kernel_matches_oracle(device, small_world):
    pa = small_world["pa"]
    samples, offsets = _batch(pa)
    feat, stdv, nf, ne, st = device.events(samples, offsets, feat_stride=2736)
    for i, x in enumerate(pa):
        f, s = po.generate_events(x)
        ev, _ = po.detect_events(x)
        assert st[i] == 0 and ne[i] == len(ev) and nf[i] == len(f)
        assert np.array_equal(bits(feat[i, :nf[i]]), bits(f))
        assert np.array_equal(bits(stdv[i, :nf[i]]), bits(s))


def test_events_kernel_matches_reference_golden(device):
    g = np.load(os.path.join(GOLDEN, "events.npz"))
    pa = [g[f"x{r}"] for r in range(6)]  # includes a ragged 1777-sample chunk
    samples, offsets = _batch(pa)
    feat, stdv, nf, ne, st = device.events(samples, offsets, feat_stride=2736)
    raw = device.events_raw(6)
    for r in range(6):
        n, E = len(pa[r]), len(g[f"mean{r}"])
        assert ne[r] == E
        # every stage of DetectEvents, bit for bit against the REAL event.h (tests/golden/make_golden.py)
        for k in ("ps", "pss"):
            assert np.array_equal(bits(raw[k][r, :n + 1]), bits(g[f"{k}{r}"])), (k, r)
        assert np.array_equal(bits(raw["t1"][r, :n + 1]), bits(g[f"t1_{r}"]))
        assert np.array_equal(bits(raw["t2"][r, :n + 1]), bits(g[f"t2_{r}"]))
        assert raw["n_peaks"][r] == len(g[f"peaks{r}"])
        assert np.array_equal(raw["peaks"][r, :raw["n_peaks"][r]], g[f"peaks{r}"].astype(np.uint32))
        assert np.array_equal(raw["start"][r, :E], g[f"start{r}"].astype(np.uint32))
        assert np.array_equal(raw["length"][r, :E], g[f"length{r}"].astype(np.uint32))
        assert np.array_equal(bits(raw["mean"][r, :E]), bits(g[f"mean{r}"]))
        assert np.array_equal(bits(raw["stdv"][r, :E]), bits(g[f"stdv{r}"]))
        # features are the z-scored, compressed event means: recompute from the golden events
        f, s = po.generate_events(pa[r])
        assert np.array_equal(bits(feat[r, :nf[r]]), bits(f))


def test_events_dense_peaks_beyond_the_lds_peak_lists(device):
    """Square waves of period 6 and 8 samples give a peak every three or four samples: 84 / 63 events in a segment of
    250 samples -- more than the 63 entries of a lane's peak list in LDS (the rest goes to the list in memory),
    resp. exactly as many.  Every stage against the oracle (its event detection is pinned to the real event.h)."""
    rng = np.random.default_rng(5)
    k = np.arange(4000)
    pa = []
    for period, amp in ((6, 10.0), (8, 30.0), (6, 30.0)):
        pa.append((90 + amp * (((k // (period // 2)) % 2) * 2 - 1) + 0.5 * rng.standard_normal(4000)).astype(np.float32))
    mixed = pa[0].copy()
    mixed[2000:] = (90 + 12 * rng.standard_normal(2000)).astype(np.float32)   # dense first half, ordinary second half
    pa.append(mixed)
    samples, offsets = _batch(pa)
    feat, stdv, nf, ne, st = device.events(samples, offsets, feat_stride=2736)
    raw = device.events_raw(len(pa))
    dense = 0
    for i, x in enumerate(pa):
        f, s = po.generate_events(x)
        ev, _ = po.detect_events(x)
        assert st[i] == 0 and ne[i] == len(ev) and nf[i] == len(f), i
        assert np.array_equal(bits(feat[i, :nf[i]]), bits(f)) and np.array_equal(bits(stdv[i, :nf[i]]), bits(s)), i
        starts = np.array([e[0] for e in ev]) if not hasattr(ev, "dtype") else ev["start"]
        assert np.array_equal(raw["start"][i, :len(ev)], np.asarray(starts, dtype=np.uint32)), i
        dense = max(dense, int(np.histogram(starts, bins=np.arange(0, 4001, 250))[0].max()))
    assert dense > 70


def test_events_of_scaled_signals_across_the_fast_division_range(device):
    """The t-statistics divide by the window length with a reciprocal form inside 2^-100 <= |x| <= 2^100 and with the
    IEEE division outside (events.hip, DivW).  Signals scaled so that the window sums / sums of squares lie inside the
    range, straddle its lower end (scale 1e-17: sums of squares around 2^-100 = 7.9e-31) and fall below it: features bit
    for bit against the oracle in every case (the thresholds on the t-statistic do not see the scale)."""
    rng = np.random.default_rng(3)
    base = (90 + 12 * rng.standard_normal(4000)).astype(np.float32)
    pa = [(base * np.float32(sc)).astype(np.float32) for sc in (1.0, 1e-14, 1e-17, 1e-19, 1e9)]
    samples, offsets = _batch(pa)
    feat, stdv, nf, ne, st = device.events(samples, offsets, feat_stride=2736)
    for i, x in enumerate(pa):
        f, s = po.generate_events(x)
        ev, _ = po.detect_events(x)
        assert st[i] == 0 and ne[i] == len(ev) and nf[i] == len(f) and nf[i] > 100, i
        assert np.array_equal(bits(feat[i, :nf[i]]), bits(f)) and np.array_equal(bits(stdv[i, :nf[i]]), bits(s)), i


def test_division_free_forms_on_the_device_over_every_float(device):
    """The two arithmetic shortcuts of the device code, swept ON THE DEVICE against the expressions they replace:
    the classifier's start-score factor for every non-negative distance below 8 radii (four radii), and the
    t-statistics' x / w for EVERY float bit pattern (both signs, denormals, infinities, NaNs) and several window lengths
    incl. the defaults and one beyond the fast form's range.  No value may differ in any bit."""
    for radius in (0.08, 0.15, 0.05, 1.0):
        hi = int(np.float32(8.0 * radius).view(np.uint32))
        assert device.start_coef_sweep(0, hi, radius) == 0, radius
    for w in (3, 6, 2, 5, 7, 64, 65, 1000):
        assert device.div_window_sweep(0, 0x80000000, w) == 0, w            # +0 .. +NaN
        assert device.div_window_sweep(0x80000000, 0xFFFFFFFF, w) == 0, w   # -0 .. -NaN


def test_events_edge_cases(device):
    flat = np.full(4000, 90.0, np.float32)                       # no peaks at all
    short = np.linspace(60, 120, 10).astype(np.float32)         # shorter than two windows
    empty = np.zeros(0, np.float32)
    rng = np.random.default_rng(1)
    noisy = (90 + 25 * rng.standard_normal(4096)).astype(np.float32)  # maximum chunk size, many events
    pa = [flat, short, empty, noisy]
    samples, offsets = _batch(pa)
    feat, stdv, nf, ne, st = device.events(samples, offsets, feat_stride=2736)
    assert nf[0] == 0 and st[0] & 1 and nf[1] == 0 and nf[2] == 0
    f, s = po.generate_events(noisy)
    assert nf[3] == len(f) and np.array_equal(bits(feat[3, :nf[3]]), bits(f))
    with pytest.raises(RuntimeError):  # > 4096 samples in one chunk is rejected, not truncated
        device.events(np.zeros(5000, np.float32), np.array([0, 5000], np.uint32))


# ------------------------------------------------------------------ ADC -> pA on the device (A1)

def test_raw_conversion_matches_oracle_addsignal(device):
    """SignalBatch::AddSignal (signal_batch.cc:139-151): scale, (30, 200) pA filter, in-order
    compaction -- the device's resident-read path against the ORACLE's restatement (so_add_signal), bit
    for bit, and the product's host loop (smhost_raw_to_pa) against the same oracle: reads with
    samples below 30 and above 200 pA, ragged lengths (0, 1, a segment boundary, several segments), a
    read longer than its slot keeps, per-read calibration constants."""
    rng = np.random.default_rng(5)
    lens = [0, 1, 4000, 4095, 4096, 4097, 12000, 30011]
    dg = np.array([8192, 8192, 8192, 2048, 8192, 8192, 4096, 8192], np.float32)
    rg = np.array([1467.61, 1400.5, 1467.61, 748.58, 1467.61, 1211.3, 1467.61, 1467.61], np.float32)
    of = np.array([10, -3, 10, 6, 10, 21, 10, 10], np.float32)
    raws = []
    for k, n in enumerate(lens):
        r = rng.integers(-200, 1400, size=n).astype(np.int16)   # pA from about -30 to 250: both cuts are hit
        if n >= 4000:
            r[100:140] = 32767
            r[-50:] = -32768
        raws.append(r)
    ro = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    pin = pipeline.PinnedArray((int(ro[-1]),), np.int16)
    pin.array[:] = np.concatenate(raws)
    keep = 12000
    device.configure_reads(16, keep)
    slots = np.array([3, 0, 5, 7, 1, 2, 9, 15], np.uint32)
    got_len = device.upload_raw(pin.array, ro, dg, rg, of, slots=slots)
    some_dropped = False
    for k, r in enumerate(raws):
        want = po.add_signal(r, dg[k], rg[k], of[k])
        host = pipeline.raw_to_pa(r, dg[k], rg[k], of[k])       # SignalBatch::AddSignal mirror of the host library
        assert np.array_equal(bits(host), bits(want)), k
        some_dropped |= 0 < len(want) < len(r)
        assert got_len[k] == len(want), (k, got_len[k], len(want))
        pa, kept, total = device.download_pa(int(slots[k]), keep)
        assert total == len(want) and kept == min(len(want), keep)
        assert np.array_equal(bits(pa), bits(want[:keep])), k
    assert some_dropped
    pin.close()


def test_pieces_pulled_from_pinned_memory_match_addsignal(device):
    """smhip_reads_upload_pieces: pieces at every 2-byte phase of a 16-byte unit, of every length around the
    kernel's unit and segment boundaries (0, 1, 7, 8, 9, ... 4096 +- 1, 9000), with samples on both sides of the
    (30, 200) pA cuts, end up in their slots exactly as the oracle's AddSignal leaves them (signal_batch.cc:139-151);
    the samples around a piece never leak into it; and a wrong expect_len makes the next collecting call fail."""
    rng = np.random.default_rng(31)
    total = 400000
    pin = pipeline.PinnedArray((total,), np.int16)
    pin.array[:] = rng.integers(-400, 1600, size=total).astype(np.int16)
    lens = [0, 1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, 255, 256, 257, 4000, 4095, 4096, 4097, 8191, 8192, 9000]
    begins, ln = [], []
    at = 3
    for k, n in enumerate(lens * 2):
        at += int(rng.integers(0, 40))            # any phase
        begins.append(at)
        ln.append(n)
        at += n
    assert at < total
    dg, rg, of = synth.DIGITISATION, synth.RANGE, synth.OFFSET
    n = len(begins)
    device.configure_reads(n + 3, 9000)
    want = [po.add_signal(pin.array[b:b + l], dg, rg, of) for b, l in zip(begins, ln)]
    slots = (np.arange(n, dtype=np.uint32)[::-1] + 2).copy()    # not the identity
    device.upload_pieces(pin, begins, ln, dg, rg, of, slots=slots, expect=[len(w) for w in want])
    for r in range(n):
        got, kept, tot = device.download_pa(int(slots[r]), 9000)
        assert kept == tot == len(want[r]), (r, ln[r], kept, len(want[r]))
        assert np.array_equal(bits(got), bits(want[r])), (r, ln[r])
    # the check against the host's count
    bad = [len(w) for w in want]
    bad[5] += 1
    device.upload_pieces(pin, begins, ln, dg, rg, of, slots=slots, expect=bad)
    with pytest.raises(RuntimeError):
        device.sync()
    # ... sticky: a caller that ignores the error gets it again from every call that would hand out results or build a
    # batch from these slots, until the pieces are uploaded again
    with pytest.raises(RuntimeError):
        device.sync()
    with pytest.raises(RuntimeError, match="upload the pieces again"):
        device.select(np.arange(3), np.zeros(3))
    with pytest.raises(RuntimeError, match="expect_len"):
        device.download()
    device.upload_pieces(pin, begins, ln, dg, rg, of, slots=slots, expect=[len(w) for w in want])
    device.sync()                                 # ... and the context goes on
    device.select(np.arange(3), np.zeros(3))
    # EVERY piece is range-checked on the host, not only the ends of the list: one piece in the middle that runs past
    # the end of its pinned block (or starts in front of it) is an error code, never an address handed to the device
    mid = len(begins) // 2
    for bad_begin, bad_len in ((total - 100, 4000), (-50000, 4000)):
        bb, bl = list(begins), list(ln)
        bb[mid], bl[mid] = bad_begin, bad_len
        with pytest.raises(RuntimeError, match="pinned"):
            device.upload_pieces(pin, bb, bl, dg, rg, of, slots=slots)

    class Pageable:                               # ordinary host memory: refused with an error code, not a GPU fault
        array = np.zeros(20000, np.int16)
    with pytest.raises(RuntimeError, match="pinned"):
        device.upload_pieces(Pageable, [5, 9000], [4000, 4000], dg, rg, of)
    pin.close()


def test_resident_reads_give_the_same_records_as_uploaded_pa(device, small_world):
    """The two ways of feeding a batch -- pA chunks uploaded with the batch, or chunks cut on the device
    out of resident reads converted from raw ADC counts -- give identical result records, for one-chunk
    reads and for chunk 0 / chunk 1 of two-chunk reads."""
    raw = small_world["raw"]
    pa = small_world["pa"]
    samples, offsets = _batch(pa)
    base = device.map_chunks(samples, offsets, n_refs=3)
    n = len(raw)
    flat = np.ascontiguousarray(np.concatenate(raw).astype(np.int16))
    ro = (np.arange(n + 1) * raw.shape[1]).astype(np.uint64)
    device.configure_reads(n, 8000)
    lens = device.upload_raw(flat, ro, synth.DIGITISATION, synth.RANGE, synth.OFFSET)
    assert np.array_equal(lens, [len(x) for x in pa])
    device.select(np.arange(n), np.zeros(n))
    device.run(3); device.sync()
    assert device.download().tobytes() == base.tobytes()
    # two-chunk reads: slot k holds read 2k followed by read 2k+1
    m = n // 2
    ro2 = (np.arange(m + 1) * 2 * raw.shape[1]).astype(np.uint64)
    lens2 = device.upload_raw(flat[:int(ro2[-1])], ro2, synth.DIGITISATION, synth.RANGE, synth.OFFSET)
    assert np.array_equal(lens2, [len(pa[2 * k]) + len(pa[2 * k + 1]) for k in range(m)])
    assert all(len(x) == 4000 for x in pa)   # (the generator's reads lose nothing to the filter)
    sl = np.repeat(np.arange(m), 2)
    ci = np.tile([0, 1], m)
    device.select(sl, ci)
    device.run(3); device.sync()
    assert device.download().tobytes() == base[:2 * m].tobytes()
    with pytest.raises(RuntimeError):   # a chunk beyond what a slot keeps is refused, not clipped silently
        device.select([0], [2])


def test_select_before_the_previous_batch_was_collected(device, small_world):
    """`smhip_batch_select` copies its arguments through pinned staging of the context without waiting for the stream;
    a caller that selects the NEXT batch before it has collected the previous one must get a wait (an event behind the
    copies that read the staging), never a torn copy: batches selected back to back -- other slots, other sizes, with
    and without carried anchors, so that the staging is rewritten and regrown while the previous copies may still be
    queued -- give the records of the same batches mapped one at a time."""
    raw = small_world["raw"]
    n = len(raw)
    flat = np.ascontiguousarray(np.concatenate(raw).astype(np.int16))
    ro = (np.arange(n + 1) * raw.shape[1]).astype(np.uint64)
    device.configure_reads(n, 8000)
    device.upload_raw(flat, ro, synth.DIGITISATION, synth.RANGE, synth.OFFSET)
    rng = np.random.default_rng(77)
    picks = [rng.permutation(n)[:k] for k in (n, 3, n - 1, 1, n // 2)]

    def carried_for(k):   # a few synthetic carried anchors per chunk: the large staging block is used and regrown
        if k % 2 == 0:
            return None, None
        per = 5 + 40 * k
        ca = np.zeros(len(picks[k]) * per, _abi.CARRIED_DTYPE)
        ca["ref_strand"] = rng.integers(0, 6, len(ca)); ca["target_position"] = rng.integers(0, 90000, len(ca))
        ca["query_position"] = rng.integers(0, 400, len(ca)); ca["distance"] = rng.random(len(ca)).astype(np.float32) * 0.05
        return ca, (np.arange(len(picks[k]) + 1) * per).astype(np.uint32)

    carried = [carried_for(k) for k in range(len(picks))]
    want = []
    for sl, (ca, co) in zip(picks, carried):   # one at a time: select, run, collect
        device.select(sl, np.zeros(len(sl)), carried=ca, carried_offsets=co)
        device.run(3); device.sync()
        want.append(device.download().tobytes())
    got = []
    for rep in range(3):                        # back to back: the next select right behind the run, nothing collected
        for k, (sl, (ca, co)) in enumerate(zip(picks, carried)):
            device.select(sl, np.zeros(len(sl)), carried=ca, carried_offsets=co)
            device.run(3)
            if k + 1 < len(picks):
                nsl, (nca, nco) = picks[k + 1], carried[k + 1]
                device.select(nsl, np.zeros(len(nsl)), carried=nca, carried_offsets=nco)   # replaces the batch: never run
                device.select(sl, np.zeros(len(sl)), carried=ca, carried_offsets=co)       # ... and back again
                device.run(3)
            device.sync()
            got.append(device.download().tobytes())
    assert got == want * 3


# ------------------------------------------------------------------ seed lookup (A10)

@pytest.mark.parametrize("blocks,strict", [(0, 0), (0, 1), (1, 0), (8, 1), (8, 0)])
def test_seed_lookup_matches_oracle(device, small_world, oracle_tree, blocks, strict):
    qs = []
    for x in small_world["pa"][:12]:
        f, _ = po.generate_events(x)
        qs += [f[p:p + 6] for p in range(2, len(f) - 5, 2)]
    qs = np.array(qs, np.float32)
    device.params.walk_blocks_per_cu = blocks   # occupancy of the walker must not change results
    device.params.reference_walk = strict       # nor may the box pruning: same hits, same order
    try:
        counts, idx, dist, nodes, pts = device.seed_lookup(qs, cap=1024)
    finally:
        device.params.walk_blocks_per_cu = 0
        device.params.reference_walk = 0
    total = 0
    for k in range(len(qs)):
        n, oi, od, stt = oracle_tree.radius_search(qs[k], 0.08)
        assert counts[k] == n
        assert np.array_equal(idx[k, :n], oi)                      # traversal order, point ids
        assert np.array_equal(bits(dist[k, :n]), bits(od))
        if strict:
            assert nodes[k] == stt[0] and pts[k] == stt[2]         # same nodes / points visited
        else:
            assert nodes[k] <= stt[0] and pts[k] <= stt[2]         # a subset of the reference's work
        total += n
    assert total > 1000


@pytest.mark.parametrize("npts", [6, 7, 15, 21, 45, 300])
def test_seed_lookup_tiny_trees(npts):
    """Degenerate trees: a single leaf (the root), a handful of nodes; every query of the cloud."""
    rng = np.random.default_rng(5)
    vals = rng.normal(size=npts).astype(np.float32)
    ix = pipeline.HostIndex.from_points(np.arange(npts, dtype=np.uint64) << 1, vals, threads=1)
    dev = pipeline.DeviceMapper(0)
    dev.upload_index(ix)
    tree = po.OracleTree(ix.values.copy())
    qs = np.array([vals[i:i + 6] + rng.normal(scale=0.05, size=6).astype(np.float32)
                   for i in range(0, max(1, npts - 5))], np.float32)
    for strict in (0, 1):
        dev.params.reference_walk = strict
        counts, idx, dist, nodes, pts = dev.seed_lookup(qs, cap=64)
        for k in range(len(qs)):
            n, oi, od, stt = tree.radius_search(qs[k], 0.08)
            assert counts[k] == n and np.array_equal(idx[k, :n], oi) and np.array_equal(bits(dist[k, :n]), bits(od))
            if strict:
                assert nodes[k] == stt[0] and pts[k] == stt[2]
    assert counts.sum() >= len(qs)
    dev.close(); tree.close()


def test_upload_rejects_leaves_the_scan_cannot_hold():
    """The scan phase tests one point per lane, 21 lanes per leaf: a tree built with a larger
    max_leaf is refused at upload with a message, never searched wrongly."""
    rng = np.random.default_rng(9)
    vals = rng.normal(size=600).astype(np.float32)
    ok = pipeline.HostIndex.from_points(np.arange(600, dtype=np.uint64) << 1, vals, max_leaf=21, threads=1)
    big = pipeline.HostIndex.from_points(np.arange(600, dtype=np.uint64) << 1, vals, max_leaf=40, threads=1)
    dev = pipeline.DeviceMapper(0)
    dev.upload_index(ok)
    tree = po.OracleTree(ok.values.copy(), leaf_max=21)
    qs = np.array([vals[i:i + 6] for i in range(0, 590, 7)], np.float32)
    counts, idx, dist, _, _ = dev.seed_lookup(qs, cap=64)
    for k in range(len(qs)):
        n, oi, od, _ = tree.radius_search(qs[k], 0.08)
        assert counts[k] == n and np.array_equal(idx[k, :n], oi) and np.array_equal(bits(dist[k, :n]), bits(od))
    with pytest.raises(RuntimeError, match="more than 21 points"):
        dev.upload_index(big)
    dev.close(); tree.close()


def test_seed_lookup_matches_reference_golden():
    g = np.load(os.path.join(GOLDEN, "kdtree.npz"))
    ix = pipeline.HostIndex.from_points(g["positions"], g["values"], threads=2)
    dev = pipeline.DeviceMapper(0)
    dev.upload_index(ix)
    counts, idx, dist, _, _ = dev.seed_lookup(g["queries"], cap=256)
    off = g["hit_off"]
    for k in range(len(g["queries"])):
        n = off[k + 1] - off[k]
        assert counts[k] == n
        assert np.array_equal(idx[k, :n], g["hit_idx"][off[k]:off[k + 1]])
        assert np.array_equal(bits(dist[k, :n]), bits(g["hit_dist"][off[k]:off[k + 1]]))
    dev.close()


def _hit_sets_equal(idx, dist, n, ref_idx, ref_dist):
    """The hits of one query as a SET: (point id, distance bits) pairs, whatever the order."""
    a = sorted(zip(idx[:n].tolist(), bits(dist[:n]).tolist()))
    b = sorted(zip(np.asarray(ref_idx).tolist(), bits(ref_dist).tolist()))
    return a == b


def test_set_walk_matches_reference_golden():
    """The PRODUCTION search of the mapping path -- the order-free set walk over quad records (integer box sums on the
    one-step grid, four-slot visits, 4-byte stack with spill) -- through the per-stage entry point
    (search_mode = SMHIP_SEARCH_SET): per query the same count and the same SET of (point id, distance bits) as the
    REAL nanoflann (tests/golden/kdtree.npz, nanoflann.hpp:1278-1410), queries outside the root box and exact data
    points included; at every walker occupancy."""
    g = np.load(os.path.join(GOLDEN, "kdtree.npz"))
    ix = pipeline.HostIndex.from_points(g["positions"], g["values"], threads=2)
    dev = pipeline.DeviceMapper(0)
    dev.upload_index(ix)
    dev.params.search_mode = _abi.SEARCH_SET
    off = g["hit_off"]
    # ... and with the walk of a query shared between the lanes of its wave (the default), not shared (knob walk_share
    # 0: round 4's walk), and shared with two queries per wave at a time (62 lanes of every wave only ever help)
    for blocks, share, grab in ((0, -1, 0), (1, -1, 0), (8, -1, 0), (0, 0, 0), (0, 1, 2), (1, 1, 64)):
        dev.params.walk_blocks_per_cu = blocks
        dev.set_knob("walk_share", share)
        dev.set_knob("walk_grab", grab)
        dev.set_knob("walk_grab_upto", 2040 if grab else 0)
        counts, idx, dist, _, _ = dev.seed_lookup(g["queries"], cap=256)
        assert dev.debug_counters()[6] == 2          # the quad walk ran
        for k in range(len(g["queries"])):
            n = off[k + 1] - off[k]
            assert counts[k] == n
            assert _hit_sets_equal(idx[k], dist[k], n, g["hit_idx"][off[k]:off[k + 1]], g["hit_dist"][off[k]:off[k + 1]]), k
    dev.params.walk_blocks_per_cu = 0
    for name in ("walk_share", "walk_grab", "walk_grab_upto"):
        dev.set_knob(name, -1 if name == "walk_share" else 0)
    assert int(counts.sum()) == int(off[-1]) and int(off[-1]) > 150 and int(counts.max()) >= 3
    # a cap below a query's hit count: the count is still the total, the stored hits are a subset of the set
    counts2, idx2, dist2, _, _ = dev.seed_lookup(g["queries"], cap=4)
    assert np.array_equal(counts2, counts)
    for k in range(len(g["queries"])):
        ref = set(zip(g["hit_idx"][off[k]:off[k + 1]].tolist(), bits(g["hit_dist"][off[k]:off[k + 1]]).tolist()))
        got = list(zip(idx2[k, :min(4, counts[k])].tolist(), bits(dist2[k, :min(4, counts[k])]).tolist()))
        assert len(set(got)) == len(got) and set(got) <= ref
    dev.close()


@pytest.mark.parametrize("npts,max_leaf", [(6, 20), (21, 20), (26, 21), (45, 20), (300, 20), (2000, 21), (5000, 3)])
def test_set_walk_small_trees_and_full_leaves(npts, max_leaf):
    """Set walk on degenerate shapes: a tree that is one leaf (no quad records: the entry point falls back to the
    reference's order and says so), a root with two leaves (a quad with two used slots), 21-point leaves, deep narrow
    trees (max_leaf 3): the hit sets of the pinned oracle search, every query of the cloud plus far-away ones."""
    rng = np.random.default_rng(npts)
    vals = rng.normal(size=npts).astype(np.float32)
    ix = pipeline.HostIndex.from_points(np.arange(npts, dtype=np.uint64) << 1, vals, max_leaf=max_leaf, threads=1)
    dev = pipeline.DeviceMapper(0)
    dev.upload_index(ix)
    tree = po.OracleTree(ix.values.copy(), leaf_max=max_leaf)
    qs = [vals[i:i + 6] + rng.normal(scale=0.05, size=6).astype(np.float32) for i in range(0, max(1, npts - 5), max(1, npts // 400))]
    qs += [np.full(6, 7.0, np.float32), np.array([-9, 0, 0, 0, 0, 9], np.float32)]
    qs = np.array(qs, np.float32)
    dev.params.search_mode = _abi.SEARCH_SET
    counts, idx, dist, _, _ = dev.seed_lookup(qs, cap=512)
    assert dev.debug_counters()[6] == (2 if ix.n_nodes > 1 else 0)
    for k in range(len(qs)):
        n, oi, od, _ = tree.radius_search(qs[k], 0.08)
        assert counts[k] == n and _hit_sets_equal(idx[k], dist[k], n, oi, od), k
    assert counts.sum() >= len(qs) - 2
    dev.close(); tree.close()


# ------------------------------------------------------------------ whole path (A1-A18)

def _paf_lines(res, pa_list, names, ref_lens):
    L = _abi.host_lib()
    mp = _abi.MapParams(); L.smhost_default_map_params(C.byref(mp))
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    out = []
    for i in range(len(pa_list)):
        r = res[i:i + 1].copy()
        nev = int(r["n_features"][0]) if r["n_features"][0] > 50 else 0
        buf = C.create_string_buffer(2048)
        L.smhost_format_paf(_abi.ptr(r), b"read%d" % i, len(pa_list[i]), 0, nev, arr, _abi.ptr(ref_lens),
                            len(names), C.byref(mp), buf, 2048, None)
        out.append(buf.value.decode())
    return out


@pytest.mark.parametrize("blocks,strict", [(0, 0), (2, 1), (2, 0)])
def test_pipeline_paf_identical_to_oracle(device, small_world, oracle_tree, blocks, strict):
    pa, ix, ref_lens = small_world["pa"], small_world["ix"], small_world["ref_lens"]
    samples, offsets = _batch(pa)
    device.params.walk_blocks_per_cu = blocks
    device.params.reference_walk = strict
    try:
        res = device.map_chunks(samples, offsets, n_refs=3)
    finally:
        device.params.walk_blocks_per_cu = 0
        device.params.reference_walk = 0
    if blocks == 0 and strict == 0:
        # the sort's whole-line partition route took every chunk (no part outgrew its slice); the exact two-pass
        # route (knob sort_lines = 0) gives the same records
        st_sort = device.sort_stats()
        assert st_sort[3] == 0 and st_sort[0] + st_sort[1] > len(pa), st_sort
        try:
            for lines in (0, 4):   # 4: the route's tiles of 4096 records (what batches of large chunks take)
                device.set_knob("sort_lines", lines)
                assert device.map_chunks(samples, offsets, n_refs=3).tobytes() == res.tobytes()
                assert device.sort_stats()[3] == 0
        finally:
            device.set_knob("sort_lines", -1)
        # the chain classifier's lockstep predecessor scans for every tile (default: dense pairs where the windows allow)
        for knob in ("classify_dense", "classify_resolve"):   # (... / nothing chained inside the classifier)
            device.set_knob(knob, 0)
            try:
                assert device.map_chunks(samples, offsets, n_refs=3).tobytes() == res.tobytes(), knob
            finally:
                device.set_knob(knob, -1)
    names = ["contig0", "contig1", "contig2"]
    lines = _paf_lines(res, pa, names, ref_lens)
    mapped = 0
    for i, x in enumerate(pa):
        m = po.map_read(oracle_tree, ix.positions, x, 3, ref_lens)
        assert res[i]["status"] == 0
        assert lines[i] == po.format_paf(m, "read%d" % i, names, ref_lens)   # cols 1-12 + every tag
        assert res[i]["n_anchors"] == m.stats.anchors and res[i]["hits_total"] == m.stats.hits
        if strict:   # the census mode counts exactly the reference's node visits / point tests
            assert res[i]["nodes_visited"] == m.stats.search.nodes and res[i]["points_tested"] == m.stats.search.points
        else:
            assert res[i]["nodes_visited"] <= m.stats.search.nodes and res[i]["points_tested"] <= m.stats.search.points
        mapped += m.mapped
    assert mapped >= len(pa) - 2
    # mapped locus agrees with the generator's truth (sanity of the whole setup)
    t = small_world["truth"]
    good = 0
    for i in range(len(pa)):
        c0 = res[i]["chains"][0]
        if res[i]["n_chains"] and c0["ref"] == t["contig"][i] and (c0["direction"] == 1) == (t["strand"][i] == 0):
            good += 1
    assert good >= len(pa) - 3


def test_pipeline_degenerate_signals_match_oracle(device, small_world, oracle_tree):
    """Reads that are not clean copies of the reference -- pure noise, an almost flat signal, reads
    with a stuck stretch, wide noise -- give the oracle's records and PAF rows too (the kinds of input
    tools/fuzz_parity.py draws)."""
    rng = np.random.default_rng(77)
    ix, ref_lens = small_world["ix"], small_world["ref_lens"]
    base = small_world["pa"]
    pa = [rng.normal(90, 12, size=4000).astype(np.float32),
          np.full(4000, 85.0, np.float32) + rng.normal(0, 0.01, size=4000).astype(np.float32),
          base[0].copy(), base[1].copy(), rng.normal(120, 30, size=4000).astype(np.float32).clip(31, 199)]
    pa[2][1000:1500] = pa[2][1000]
    pa[3][:700] = pa[3][700]
    samples, offsets = _batch(pa)
    res = device.map_chunks(samples, offsets, n_refs=3)
    names = ["contig0", "contig1", "contig2"]
    lines = _paf_lines(res, pa, names, ref_lens)
    for i, x in enumerate(pa):
        m = po.map_read(oracle_tree, ix.positions, x, 3, ref_lens)
        assert lines[i] == po.format_paf(m, "read%d" % i, names, ref_lens)
        assert res[i]["n_anchors"] == m.stats.anchors and res[i]["hits_total"] == m.stats.hits


def test_many_contigs_more_than_128_candidate_chains():
    """The reference keeps up to 3 chains per (strand, contig) bucket with no bound on their number
    (spatial_index.cc:555-568).  140 contigs that share a 6 kb segment: a read from the segment chains
    in every contig, so a chunk has > 128 candidate (and primary) chains -- beyond what the finalize
    kernel ranks in LDS.  Records and PAF rows must still be the oracle's."""
    NC = 140
    shared = synth.make_genome(101, [6000])[0]
    uniq = synth.make_genome(102, [2000] * (2 * NC))
    genome = [uniq[2 * i] + shared + uniq[2 * i + 1] for i in range(NC)]
    lm = synth.make_pore_model()
    raw, _ = synth.simulate_reads([shared], lm, 12, seed=5)
    pa = [po.add_signal(r, synth.DIGITISATION, synth.RANGE, synth.OFFSET) for r in raw]
    ix = pipeline.HostIndex.build(genome, lm, threads=4)
    dev = pipeline.DeviceMapper(0)
    dev.upload_index(ix)
    tree = po.OracleTree(ix.values.copy())
    ref_lens = np.array([len(g) for g in genome], np.uint32)
    names = ["contig%d" % i for i in range(NC)]
    samples, offsets = _batch(pa)
    res = dev.map_chunks(samples, offsets, n_refs=NC)
    lines = _paf_lines(res, pa, names, ref_lens)
    most = 0
    for i, x in enumerate(pa):
        m = po.map_read(tree, ix.positions, x, NC, ref_lens)
        most = max(most, m.n_chains)
        assert res[i]["status"] == 0
        assert res[i]["n_chains"] == m.n_chains and res[i]["n_anchors"] == m.stats.anchors
        assert lines[i] == po.format_paf(m, "read%d" % i, names, ref_lens)
    assert most > 128, most   # the case this test exists for
    tree.close()
    dev.close()


def test_more_contigs_than_the_partition_kernel_counts_groups_for():
    """600 contigs = 1200 (reference, strand) groups: beyond the 1024 group counters the sort's partition
    kernel keeps in LDS, so the bucket boundaries come from GroupStartKernel's pass over the sorted group
    column.  Reads from a segment that 40 of the contigs share chain in all of those; records and PAF rows
    against the oracle."""
    NC, NSH = 600, 40
    shared = synth.make_genome(201, [5000])[0]
    uniq = synth.make_genome(202, [1500] * (2 * NC))
    genome = [uniq[2 * i] + (shared if i % (NC // NSH) == 0 else shared[:0]) + uniq[2 * i + 1] for i in range(NC)]
    lm = synth.make_pore_model()
    raw, _ = synth.simulate_reads([shared], lm, 6, seed=9)
    raw2, _ = synth.simulate_reads(genome, lm, 6, seed=10)
    pa = [po.add_signal(r, synth.DIGITISATION, synth.RANGE, synth.OFFSET) for r in list(raw) + list(raw2)]
    ix = pipeline.HostIndex.build(genome, lm, threads=4)
    dev = pipeline.DeviceMapper(0)
    dev.upload_index(ix)
    tree = po.OracleTree(ix.values.copy())
    ref_lens = np.array([len(g) for g in genome], np.uint32)
    names = ["contig%d" % i for i in range(NC)]
    samples, offsets = _batch(pa)
    res = dev.map_chunks(samples, offsets, n_refs=NC)
    lines = _paf_lines(res, pa, names, ref_lens)
    for i, x in enumerate(pa):
        m = po.map_read(tree, ix.positions, x, NC, ref_lens)
        assert res[i]["status"] == 0
        assert res[i]["n_chains"] == m.n_chains and res[i]["n_anchors"] == m.stats.anchors
        assert lines[i] == po.format_paf(m, "read%d" % i, names, ref_lens)
    tree.close()
    dev.close()


def test_pipeline_small_fast_cap_uses_overflow_pass(device, small_world):
    """Queries with more hits than the fast slots are re-walked into exact-size segments:
    results must not depend on the capacity."""
    pa = small_world["pa"][:16]
    samples, offsets = _batch(pa)
    base = device.map_chunks(samples, offsets, n_refs=3)
    device.params.hits_per_query_fast = 2
    try:
        tiny = device.map_chunks(samples, offsets, n_refs=3)
    finally:
        device.params.hits_per_query_fast = pipeline.default_params().hits_per_query_fast
    assert np.array_equal(base.tobytes(), tiny.tobytes())


def test_index_export_copy_attach_gives_identical_results(device, small_world):
    """What the multi-GPU replication does on the device side (multigpu.replicate_index): the exported
    walker arrays are copied byte for byte into fresh device buffers and a second context attaches the
    copies."""
    from sigmap_amd import multigpu
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    pa = small_world["pa"][:24]
    samples, offsets = _batch(pa)
    base = device.map_chunks(samples, offsets, n_refs=3)
    d = device.export_index()
    assert d.n_nodes + 1 == d.n_leaves and d.bbox_bytes == 96
    nd = _abi.IndexDev()
    for f in multigpu._SCALARS:
        setattr(nd, f, getattr(d, f))
    keep = []
    for k, a in enumerate(multigpu._ARRAYS):
        nbytes = int(getattr(d, a + "_bytes"))
        t = C.c_void_p()
        assert hip.hipMalloc(C.byref(t), nbytes) == 0
        device.export_copy(k, t.value, nbytes)   # smhip_index_export_copy: what replicate_index uses on rank 0
        setattr(nd, a, t.value)
        setattr(nd, a + "_bytes", nbytes)
        keep.append(t)
    peer = pipeline.DeviceMapper(0)
    peer.attach_index(nd)
    got = peer.map_chunks(samples, offsets, n_refs=3)
    assert np.array_equal(base.tobytes(), got.tobytes())
    bad = _abi.IndexDev()
    C.memmove(C.byref(bad), C.byref(nd), C.sizeof(nd))
    bad.n_leaves += 1
    with pytest.raises(RuntimeError):
        peer.attach_index(bad)
    peer.close()
    for t in keep:
        hip.hipFree(t)
    # the C-ABI broadcast with one GPU is a no-op that leaves the index usable; two contexts of one GPU are refused
    L = _abi.hip_lib()
    arr = (C.c_void_p * 1)(device._h)
    assert L.smhip_index_bcast(arr, 1, 0) == 0
    assert np.array_equal(device.map_chunks(samples, offsets, n_refs=3).tobytes(), base.tobytes())
    other = pipeline.DeviceMapper(0)
    arr2 = (C.c_void_p * 2)(device._h, other._h)
    assert L.smhip_index_bcast(arr2, 2, 0) != 0 and b"one context per GPU" in L.smhip_last_error(device._h)
    other.close()


def test_index_bcast_loopback_runs_the_rccl_calls_on_one_gpu(device, small_world):
    """smhip_index_bcast_loopback: librccl is loaded, a one-rank communicator is created
    (ncclCommInitAll) and the six walker arrays go through one grouped ncclBroadcast each -- the call
    sequence of smhip_index_bcast -- into buffers the second context owns.  The copy is byte-identical
    (export_copy of both sides) and maps to identical records, also after the source context is gone."""
    from sigmap_amd import multigpu
    pa = small_world["pa"][:24]
    samples, offsets = _batch(pa)
    src = pipeline.DeviceMapper(0)
    src.upload_index(small_world["ix"])
    base = src.map_chunks(samples, offsets, n_refs=3)
    assert np.array_equal(base.tobytes(), device.map_chunks(samples, offsets, n_refs=3).tobytes())
    peer = pipeline.DeviceMapper(0)
    src.bcast_loopback_to(peer)
    ds, dp = src.export_index(), peer.export_index()
    for f in multigpu._SCALARS:
        assert getattr(ds, f) == getattr(dp, f)
    for k, a in enumerate(multigpu._ARRAYS):
        n = int(getattr(ds, a + "_bytes"))
        assert int(getattr(dp, a + "_bytes")) == n and getattr(dp, a) != getattr(ds, a)
        x, y = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        src.export_copy(k, x.ctypes.data, n)      # (hipMemcpyDefault: host destinations are fine)
        peer.export_copy(k, y.ctypes.data, n)
        assert np.array_equal(x, y), a
    assert np.array_equal(peer.map_chunks(samples, offsets, n_refs=3).tobytes(), base.tobytes())
    src.close()                                   # the peer owns its copy
    assert np.array_equal(peer.map_chunks(samples, offsets, n_refs=3).tobytes(), base.tobytes())
    # argument checks
    L = _abi.hip_lib()
    empty = pipeline.DeviceMapper(0)
    assert L.smhip_index_bcast_loopback(empty._h, peer._h) != 0 and b"no index" in L.smhip_last_error(empty._h)
    assert L.smhip_index_bcast_loopback(peer._h, peer._h) != 0
    empty.close()
    peer.close()


def test_mapper_pool_batches_in_flight_give_serial_results(device, small_world):
    """Batches mapped concurrently on three contexts / streams that share one resident index
    (pipeline.MapperPool) give byte for byte what one context gives one batch after the other."""
    pa = small_world["pa"]
    batches = [_batch(pa[i::5]) for i in range(5)] + [_batch(pa[:7])]
    want = [device.map_chunks(s, o, n_refs=3).tobytes() for s, o in batches]
    pool = pipeline.MapperPool(0, depth=3)
    pool.first.upload_index(small_world["ix"])
    pool.share_index()
    for _ in range(2):  # second round: contexts reuse their scratch with other batch sizes
        got = pool.run_steps(len(batches), lambda dv, i: dv.map_chunks(*batches[i], n_refs=3).tobytes())
        assert got == want
    def failing(dv, i):
        raise ValueError("step %d" % i)
    with pytest.raises(ValueError):  # an exception on a worker thread reaches the caller
        pool.run_steps(2, failing)
    pool.close()


def test_pipeline_hit_cap_truncation_matches_oracle(small_world):
    """spatial_index.cc:371-372: only the first num_nearest_points hits IN TRAVERSAL ORDER become
    anchors.  A point cloud that repeats one signal 12 times gives every seed >= 12 hits; with the
    cap lowered to 5 on both sides the chains only agree if the device keeps exactly the hits the
    reference's traversal order would keep (which hits survive changes targets and scores)."""
    lm = small_world["lm"]
    rng = np.random.default_rng(8)
    unit = bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 3000)])
    host = _abi.host_lib()
    sig = np.zeros(len(unit), np.float32)
    m = host.smhost_reference_signal(_abi.ptr(np.ascontiguousarray(lm)), 6, unit, len(unit), 0, _abi.ptr(sig))
    sig = sig[:m]
    values = np.concatenate([sig + np.float32(1e-4 * k) for k in range(12)]).astype(np.float32)
    positions = (np.arange(len(values), dtype=np.uint64) << np.uint64(1))  # one contig, + strand
    ix = pipeline.HostIndex.from_points(positions, values, threads=2)
    raw, _ = synth.simulate_reads([unit], lm, 6, 2)
    pa = [po.add_signal(r, synth.DIGITISATION, synth.RANGE, synth.OFFSET) for r in raw]
    samples, offsets = _batch(pa)
    tree = po.OracleTree(ix.values.copy())
    for cap in (5, 5000):
        for fast in (3, 128):   # truncation inside the fast slots and inside the overflow pass
            # The order-free set walk, then the queries beyond the cap -- here, with cap 5, all of them -- settled (a)
            # by ranking the hits found in the reference's traversal order and keeping the first `cap` (the rule), (b)
            # by a second search in the reference's order (knob second_search: round 3's route; AUTO then lets the
            # census of the index pick the reference's order for everything), (c) EXACT from the start.  Same records.
            dev = pipeline.DeviceMapper(0)
            dev.upload_index(ix)
            dev.params.num_nearest_points = cap
            dev.params.hits_per_query_fast = fast
            res = dev.map_chunks(samples, offsets, n_refs=1)          # AUTO: set walk + ranking
            c = dev.debug_counters()
            assert c[6] == 2 and (c[7] > 100) == (cap == 5)
            same = [k for k in res.dtype.names if k not in ("nodes_visited", "points_tested")]
            dev.set_knob("second_search", 1)
            dev.params.search_mode = _abi.SEARCH_SET
            again = dev.map_chunks(samples, offsets, n_refs=1)
            assert dev.debug_counters()[6] == 2 and all(np.array_equal(res[k], again[k]) for k in same)
            dev.params.search_mode = _abi.SEARCH_AUTO
            again = dev.map_chunks(samples, offsets, n_refs=1)
            assert dev.debug_counters()[6] == (0 if cap == 5 else 2) and all(np.array_equal(res[k], again[k]) for k in same)
            dev.set_knob("second_search", 0)
            dev.params.search_mode = _abi.SEARCH_EXACT
            again = dev.map_chunks(samples, offsets, n_refs=1)
            assert dev.debug_counters()[6] == 0 and all(np.array_equal(res[k], again[k]) for k in same)
            dev.close()
            assert (res["hits_total"] > res["n_queries"] * 4).sum() >= 2   # repeats: seeds with 12 hits each
            for i, x in enumerate(pa):
                f, _ = po.generate_events(x)
                chains, st = po.generate_chains(tree, ix.positions, f, 1, num_nearest_points=cap)
                assert res[i]["n_anchors"] == st.anchors and res[i]["hits_total"] == st.hits
                assert res[i]["n_chains"] == len(chains)
                for k, c in enumerate(chains[:4]):
                    g = res[i]["chains"][k]
                    assert g["score"] == np.float32(c["score"]) and g["start"] == c["start"] and g["end"] == c["end"]
                    assert g["n_anchors"] == c["n_anchors"]
    tree.close()


def test_repeat_injected_cloud_second_search_at_the_default_cap(small_world):
    """A repeat-injected reference: one 600-base unit copied 5600 times at offsets of 1e-6, so that every seed
    of a read of the unit has 5600 hits -- beyond the DEFAULT cap of 5000 (spatial_index.cc:371-372), with the fast
    slots at 512: the set walk writes four levels of overflow segments per query, OverCapCollectKernel lists every
    hitting query, and either the hits are ranked in the reference's traversal order and the first 5000 kept
    (OverCapSelectKernel: records moved between slots and overflow segments), or -- knob second_search, round 3's
    route -- a second search in the reference's order (stop_at_cap) allocates and writes the segments again.  A
    fresh context per route, incl. AUTO and EXACT.  Chains against the oracle's so_generate_chains."""
    lm = small_world["lm"]
    rng = np.random.default_rng(8)
    unit = bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 600)])
    host = _abi.host_lib()
    sig = np.zeros(len(unit), np.float32)
    m = host.smhost_reference_signal(_abi.ptr(np.ascontiguousarray(lm)), 6, unit, len(unit), 0, _abi.ptr(sig))
    sig = sig[:m]
    copies = 5600
    values = np.concatenate([sig + np.float32(1e-6 * k) for k in range(copies)]).astype(np.float32)
    positions = (np.arange(len(values), dtype=np.uint64) << np.uint64(1))  # one contig, + strand
    ix = pipeline.HostIndex.from_points(positions, values, threads=4)
    raw, _ = synth.simulate_reads([unit], lm, 8, 2)
    pa = [po.add_signal(r, synth.DIGITISATION, synth.RANGE, synth.OFFSET) for r in raw]
    samples, offsets = _batch(pa)
    tree = po.OracleTree(ix.values.copy())
    want = []
    for x in pa:
        f, _ = po.generate_events(x)
        want.append(po.generate_chains(tree, ix.positions, f, 1))
    tree.close()
    assert sum(1 for _, st in want if st.hits >= 100 * copies) >= 3     # reads of the + strand: ~100 seeds x 5600 hits
    same = None
    # (share: the set walk's lanes help each other -- the default --, do not (0), or two queries per wave at a time with 62
    # helpers (2): the overflow segments of one query then fill from many lanes' leaves at once)
    for mode, fast, second, walked, share in ((_abi.SEARCH_AUTO, 512, 0, 2, -1), (_abi.SEARCH_AUTO, 0, 0, 2, -1),
                                              (_abi.SEARCH_SET, 512, 1, 2, -1), (_abi.SEARCH_SET, 0, 1, 2, -1),
                                              (_abi.SEARCH_EXACT, 512, 0, 0, -1), (_abi.SEARCH_AUTO, 0, 1, 0, -1),
                                              (_abi.SEARCH_AUTO, 512, 0, 2, 0), (_abi.SEARCH_AUTO, 512, 0, 2, 2),
                                              (_abi.SEARCH_SET, 0, 1, 2, 2)):
        dev = pipeline.DeviceMapper(0)       # fresh: nothing learnt from earlier batches
        dev.upload_index(ix)
        dev.set_knob("walk_share", -1 if share == 2 else share)
        dev.set_knob("walk_grab", 2 if share == 2 else 0)
        dev.set_knob("walk_grab_upto", 2040 if share == 2 else 0)
        if mode == _abi.SEARCH_AUTO:
            census = dev.index_census()
            assert len(census) == 4096 and census[len(census) // 100] >= copies   # (nearly) every sampled point: 5600+ hits
        dev.params.search_mode = mode
        dev.params.hits_per_query_fast = fast
        dev.set_knob("second_search", second)
        res = dev.map_chunks(samples, offsets, n_refs=1)
        c = dev.debug_counters()
        assert c[6] == walked and (res["status"] == 0).all(), (mode, c[:8])
        if walked == 2:
            assert c[7] >= 300                    # queries beyond the cap: ranked, or handed to the second search
        for i, (chains, st) in enumerate(want):
            assert res[i]["n_anchors"] == st.anchors and res[i]["hits_total"] == st.hits and res[i]["n_chains"] == len(chains)
            for k, ch in enumerate(chains[:4]):
                g = res[i]["chains"][k]
                assert g["score"] == np.float32(ch["score"]) and g["start"] == ch["start"] and g["end"] == ch["end"]
                assert g["n_anchors"] == ch["n_anchors"] and (k > 0 or g["mapq"] == ch["mapq"])
        keep = [k for k in res.dtype.names if k not in ("nodes_visited", "points_tested")]
        cur = b"".join(np.ascontiguousarray(res[k]).tobytes() for k in keep)
        assert same is None or cur == same
        same = cur
        dev.close()
    ix.close()


def test_pipeline_heavy_ties_large_chunks(small_world):
    """160 copies of contig 0's + strand point cloud at the SAME reference positions: a read from
    that strand gets tens of thousands of anchors per chunk, every (group, target) value repeated a
    hundred times (a part of the sort far beyond what a wave or the workgroup LDS sort holds: the
    in-memory bitonic fallback; tie runs of a hundred ordered by (query, distance)), and one bucket
    holding every anchor of the chunk for the chaining kernels."""
    base_ix, truth = small_world["ix"], small_world["truth"]
    ref_of = np.asarray(base_ix.positions) >> np.uint64(33)
    n0 = int(np.argmax(ref_of != 0))  # the cloud starts with contig 0's + strand
    vals0 = np.ascontiguousarray(base_ix.values[:n0]).copy()
    pos0 = np.ascontiguousarray(base_ix.positions[:n0]).copy()
    assert n0 > 100000 and ((pos0 & np.uint64(1)) == 0).all()
    copies = 160
    values = np.concatenate([vals0 + np.float32(2e-5 * k) for k in range(copies)]).astype(np.float32)
    positions = np.tile(pos0, copies)
    ix = pipeline.HostIndex.from_points(positions, values, threads=4)
    pick = [i for i in range(len(truth)) if truth["contig"][i] == 0 and truth["strand"][i] == 0][:3]
    assert len(pick) >= 2
    pa = [small_world["pa"][i] for i in pick]
    dev = pipeline.DeviceMapper(0)
    dev.upload_index(ix)
    samples, offsets = _batch(pa)
    res = dev.map_chunks(samples, offsets, n_refs=1)
    assert (res["status"] == 0).all() and res["n_anchors"].min() > 16384
    # a hundred equal targets in a row outgrow the fixed slices of the whole-line partition route: every chunk
    # takes the exact route, and parts go to the workgroup sorter
    st_sort = dev.sort_stats()
    assert st_sort[3] == len(pa) and st_sort[2] > 0, st_sort
    for lines in (0, 4):                    # the exact route for every chunk from the start / tiles of 4096: the same records
        dev.set_knob("sort_lines", lines)
        res0 = dev.map_chunks(samples, offsets, n_refs=1)
        assert res0.tobytes() == res.tobytes()
        assert lines == 0 or dev.sort_stats()[3] == len(pa)
    dev.set_knob("sort_lines", -1)
    tree = po.OracleTree(ix.values.copy())
    for i, x in enumerate(pa):
        f, _ = po.generate_events(x)
        chains, st = po.generate_chains(tree, ix.positions, f, 1)
        assert res[i]["n_anchors"] == st.anchors and res[i]["hits_total"] == st.hits
        assert res[i]["n_chains"] == len(chains)
        for k, c in enumerate(chains[:4]):
            g = res[i]["chains"][k]
            assert g["score"] == np.float32(c["score"]) and g["start"] == c["start"] and g["end"] == c["end"]
            assert g["n_anchors"] == c["n_anchors"]
    tree.close(); dev.close()


def test_pipeline_wide_key_ranges_and_medium_parts(small_world):
    """The two remaining routes of the part sort (tools/sort_path_coverage.py counts them): target positions
    stretched by 997 make a part's dense-key range wider than 2^23, so its keys do not fit the 32-bit form
    (the 64-bit wave network for parts of <= 512 anchors); 12 copies of contig 0's + strand at the same
    positions put a few thousand anchors of a read from that strand into one part (the workgroup LDS sort:
    1 025 .. 8 192 keys).  Chunk records and chains against the oracle."""
    base_ix, truth = small_world["ix"], small_world["truth"]
    pos = np.asarray(base_ix.positions).copy()
    ref_of = pos >> np.uint64(33)
    n0 = int(np.argmax(ref_of != 0))  # the cloud starts with contig 0's + strand
    stretch = lambda p: (p & ~np.uint64((1 << 33) - 1)) | ((((p >> np.uint64(1)) & np.uint64(0xFFFFFFFF)) * np.uint64(997)) << np.uint64(1)) | (p & np.uint64(1))
    pos_w = stretch(pos)
    assert int(((pos_w >> np.uint64(1)) & np.uint64(0xFFFFFFFF)).max()) < 2**32 - 1
    copies = 12
    vals0 = np.ascontiguousarray(base_ix.values[:n0])
    values = np.concatenate([np.asarray(base_ix.values)] + [vals0 + np.float32(2e-5 * k) for k in range(1, copies)]).astype(np.float32)
    positions = np.concatenate([pos_w] + [pos_w[:n0]] * (copies - 1))
    ix = pipeline.HostIndex.from_points(positions, values, threads=4)
    n_refs = int(ref_of.max()) + 1
    pick = [i for i in range(len(truth)) if truth["contig"][i] == 0 and truth["strand"][i] == 0][:2]
    pick += [i for i in range(len(truth)) if truth["contig"][i] != 0][:2]
    pa = [small_world["pa"][i] for i in pick]
    dev = pipeline.DeviceMapper(0)
    dev.upload_index(ix)
    samples, offsets = _batch(pa)
    res = dev.map_chunks(samples, offsets, n_refs=n_refs)
    assert (res["status"] == 0).all()
    tree = po.OracleTree(ix.values.copy())
    for i, x in enumerate(pa):
        f, _ = po.generate_events(x)
        chains, st = po.generate_chains(tree, ix.positions, f, n_refs)
        assert res[i]["n_anchors"] == st.anchors and res[i]["hits_total"] == st.hits
        assert res[i]["n_chains"] == len(chains)
        for k, c in enumerate(chains[:4]):
            g = res[i]["chains"][k]
            assert g["score"] == np.float32(c["score"]) and g["start"] == c["start"] and g["end"] == c["end"]
            assert g["n_anchors"] == c["n_anchors"] and g["ref"] == c["ref"] and g["direction"] == c["direction"]
    tree.close(); dev.close()


def test_pipeline_many_anchors_large_sort_and_pool_growth(small_world, oracle_tree):
    """A wide search radius gives every chunk tens of thousands of anchors (dozens of parts for the
    sort) and, with 32 fast slots per query, far more overflow hits than the
    initial pool (the pool is grown to the measured demand and the search repeated).  Results must
    still equal the oracle run with the same radius."""
    pa, ix, ref_lens = small_world["pa"][:16], small_world["ix"], small_world["ref_lens"]
    samples, offsets = _batch(pa)
    dev = pipeline.DeviceMapper(0)
    dev.upload_index(ix)
    radius = 0.5
    dev.params.search_radius = radius
    dev.params.hits_per_query_fast = 32
    dev.set_knob("pool_min", 4096)   # first pool far too small: forces the grow-and-repeat path
    res = dev.map_chunks(samples, offsets, n_refs=3)
    dev.set_knob("pool_min", -1)
    assert (res["status"] == 0).all()
    assert res["n_anchors"].min() > 16384 and dev.debug_counters()[1] > 1000
    names = ["contig0", "contig1", "contig2"]
    lines = _paf_lines(res, pa, names, ref_lens)
    for i, x in enumerate(pa):
        m = po.map_read(oracle_tree, ix.positions, x, 3, ref_lens, params={"search_radius": radius})
        assert res[i]["n_anchors"] == m.stats.anchors and res[i]["hits_total"] == m.stats.hits
        assert lines[i] == po.format_paf(m, "read%d" % i, names, ref_lens)
    dev.close()


def test_speculative_batches_equal_the_reading_route(small_world, oracle_tree):
    """smhip_map_run sizes its later stages from two totals only the device knows.  The first batch of a
    context reads them back (two stream synchronisations); later batches are enqueued whole against
    capacities learnt from earlier ones, with guard kernels on the device.  Same records either way --
    also when a guard fires (the spec_tight knob halves a capacity: the batch is emptied on the device,
    flagged, and repeated on the reading route by the next call that collects results, sDTW refinement
    included) and when a batch is several times larger per chunk than anything seen before."""
    pa, ix, ref_lens = small_world["pa"], small_world["ix"], small_world["ref_lens"]
    genome, lm = small_world["genome"], small_world["lm"]
    dev = pipeline.DeviceMapper(0)
    dev.upload_index(ix)
    sig, offs = pipeline.reference_signals(genome, lm)
    dev.upload_reference(sig, offs)
    samples, offsets = _batch(pa[:24])

    def run(with_refine=True):
        dev.upload_batch(samples, offsets)
        dev.run(3)
        if with_refine:
            dev.refine(band=64, pad=32)
        dev.sync()
        res = dev.download()
        return res.tobytes(), (tuple(x.tobytes() for x in dev.refine_download()) if with_refine else None)

    base = run()                                         # reading route: learns the totals
    c = dev.debug_counters()
    assert c[5] == 0 and c[4] == 0
    assert run() == base                                 # speculative, capacities hold
    c = dev.debug_counters()
    assert c[5] == 1 and c[4] == 0
    for k, tight in enumerate(("1", "2")):               # query guard, then anchor guard: repeated, identical
        dev.set_knob("spec_tight", int(tight))
        try:
            assert run() == base
        finally:
            dev.set_knob("spec_tight", 0)
        c = dev.debug_counters()
        assert c[5] == 2 + k and c[4] == 1 + k, (tight, c[:8])
    assert run(with_refine=False)[0] == base[0]
    # a batch with ~20 x the anchors per chunk of anything seen so far (wide radius): the anchor guard fires by
    # itself; the result is the oracle's
    dev.params.search_radius = 0.5
    s2, o2 = _batch(pa)                                  # 48 chunks x > 16 384 anchors: beyond the 2^20 slack
    dev.upload_batch(s2, o2)
    dev.run(3)
    dev.sync()
    res = dev.download()
    c = dev.debug_counters()
    assert c[4] == 3 and (res["status"] == 0).all() and int(res["n_anchors"].sum()) > (1 << 20) + 48 * 2000
    names = ["contig0", "contig1", "contig2"]
    lines = _paf_lines(res, pa, names, ref_lens)
    for i, x in enumerate(pa[:6]):
        m = po.map_read(oracle_tree, ix.positions, x, 3, ref_lens, params={"search_radius": 0.5})
        assert res[i]["n_anchors"] == m.stats.anchors and lines[i] == po.format_paf(m, "read%d" % i, names, ref_lens)
    # ... and the next wide batch fits what was learnt from it
    dev.upload_batch(s2, o2)
    dev.run(3)
    dev.sync()
    assert dev.download().tobytes() == res.tobytes() and dev.debug_counters()[4] == 3
    # a collected-by-nobody batch is settled by the next upload (no stale flag, no hang)
    dev.params.search_radius = pipeline.default_params().search_radius
    dev.upload_batch(samples, offsets)
    dev.run(3)
    dev.upload_batch(samples, offsets)
    dev.run(3)
    dev.sync()
    assert dev.download().tobytes() == base[0]
    dev.close()


def test_repeated_speculative_batch_does_not_read_refilled_slots(small_world, oracle_tree):
    """A speculative batch that outgrows a capacity is mapped again when its results are collected.  By then the
    caller may have refilled the batch's read slots with the NEXT batch's reads (smhip_reads_upload_raw queues its
    conversion behind the batch's kernels: bench.py's PCIe-inclusive region does exactly that): the repeat starts from
    the feature signals of the first pass, so the records are those of the reads the batch was selected from --
    compared with the oracle's -- and the next batch then maps the new reads."""
    genome, lm, ix, ref_lens = small_world["genome"], small_world["lm"], small_world["ix"], small_world["ref_lens"]
    raw_a = small_world["raw"][:24]
    raw_b, _ = synth.simulate_reads(genome, lm, 24, 505)
    dev = pipeline.DeviceMapper(0)
    dev.upload_index(ix)
    dev.configure_reads(24, 4000)

    def upload(raw):
        flat = np.ascontiguousarray(np.concatenate(raw), dtype=np.int16)
        off = np.concatenate([[0], np.cumsum([len(r) for r in raw])]).astype(np.uint64)
        dev.upload_raw(flat, off, synth.DIGITISATION, synth.RANGE, synth.OFFSET, want_len=False)
        return flat   # (kept alive by the caller until the next sync)

    def expect(raw):
        out = []
        for r in raw:
            x = po.add_signal(r, synth.DIGITISATION, synth.RANGE, synth.OFFSET)
            out.append(po.map_read(oracle_tree, ix.positions, x, 3, ref_lens))
        return out

    slots, chunk0 = np.arange(24, dtype=np.uint32), np.zeros(24, np.uint32)
    keep = upload(raw_a)
    dev.select(slots, chunk0, 4000); dev.run(3); dev.sync()
    base_a = dev.download()                                   # reading route: learns the totals
    for tight in (1, 2, 0):                                    # query guard, anchor guard, no guard
        keep = upload(raw_a)
        dev.sync()
        dev.set_knob("spec_tight", tight)
        dev.select(slots, chunk0, 4000)
        dev.run(3)                                            # speculative: everything enqueued, nothing waited for
        keep2 = upload(raw_b)                                 # the next batch's reads into the SAME slots, right away
        dev.sync()                                            # finds the guard's flag and maps the batch again
        res = dev.download()
        dev.set_knob("spec_tight", 0)
        assert res.tobytes() == base_a.tobytes(), tight
        dev.select(slots, chunk0, 4000); dev.run(3); dev.sync()
        res_b = dev.download()
        for i, m in enumerate(expect(raw_b)[:8]):
            assert res_b[i]["n_chains"] == m.n_chains and res_b[i]["n_anchors"] == m.stats.anchors, (tight, i)
            if m.n_chains:
                assert res_b[i]["chains"][0]["score"] == np.float32(m.s1)
    c = dev.debug_counters()
    assert c[4] >= 2 and c[5] >= 3, c[:8]                     # two repeats, at least three speculative runs
    for i, m in enumerate(expect(raw_a)[:8]):
        assert base_a[i]["n_chains"] == m.n_chains and base_a[i]["n_anchors"] == m.stats.anchors
    del keep, keep2
    dev.close()


def test_contexts_of_a_gpu_share_one_reference_signal(device, small_world):
    """smhip_reference_share: a second context refines against the reference signal the first one uploaded
    (one copy per GPU, like the index) and gets the same distances; smhip_mem_info reports the device's memory."""
    genome, lm = small_world["genome"], small_world["lm"]
    sig, offs = pipeline.reference_signals(genome, lm)
    device.upload_reference(sig, offs)
    samples, offsets = _batch(small_world["pa"][:16])

    def refined(dv):
        dv.upload_batch(samples, offsets)
        dv.run(3)
        dv.refine(band=64, pad=32)
        dv.sync()
        return dv.download().tobytes(), tuple(x.tobytes() for x in dv.refine_download())

    want = refined(device)
    peer = pipeline.DeviceMapper(0)
    peer.attach_index(device.export_index())
    with pytest.raises(RuntimeError):
        peer.refine(band=64, pad=32)              # no batch, no reference yet
    empty = pipeline.DeviceMapper(0)
    with pytest.raises(RuntimeError):
        peer.share_reference(empty)               # the owner holds none
    empty.close()
    peer.share_reference(device)
    assert refined(peer) == want
    free, total = peer.mem_info()
    assert 0 < free <= total and total > (64 << 30)
    peer.close()
    assert refined(device) == want                # the owner is unaffected by the borrower's end


def test_integer_gap_tests_equal_the_float_form(device, small_world):
    """The chaining kernels evaluate `0.75 < (float)qd / (float)td < 5` (spatial_index.cc:489-497) as
    4 qd > 3 td and qd < 5 td, exact for target gaps below 2^21 (chain.hip GapTestsPass; exhaustive host
    check: tools/micro/gap_tests.c).  the float_gap knob selects the division form: identical records."""
    samples, offsets = _batch(small_world["pa"])
    base = device.map_chunks(samples, offsets, n_refs=3)
    device.set_knob("float_gap", 1)
    try:
        flt = device.map_chunks(samples, offsets, n_refs=3)
    finally:
        device.set_knob("float_gap", 0)
    assert base.tobytes() == flt.tobytes() and (base["n_chains"] > 0).sum() > 40


def test_pipeline_empty_and_degenerate_batches(device):
    res = device.map_chunks(np.zeros(0, np.float32), np.array([0], np.uint32), n_refs=3)
    assert len(res) == 0
    flat = np.full(4000, 90.0, np.float32)
    res = device.map_chunks(flat, np.array([0, 4000], np.uint32), n_refs=3)
    assert res[0]["n_chains"] == 0 and res[0]["n_queries"] == 0 and res[0]["status"] & 1


def test_chain_anchor_carry_over_roundtrip(device, small_world, oracle_tree):
    """Chain anchors returned for chunk c can be fed back as `carried` for chunk c+1
    (spatial_index.cc:303-322): two-chunk reads must equal the oracle's streaming result."""
    genome, lm = small_world["genome"], small_world["lm"]
    raw, _ = synth.simulate_reads(genome, lm, 12, 77, n_chunks=2)
    pa = [po.add_signal(r, synth.DIGITISATION, synth.RANGE, synth.OFFSET) for r in raw]
    ix, ref_lens = small_world["ix"], small_world["ref_lens"]
    first = [x[:4000] for x in pa]
    samples, offsets = _batch(first)
    device.upload_batch(samples, offsets)
    device.run(3); device.sync()
    r1, anchors, aoff = device.download(with_anchors=True)
    # the anchors of SOME chunks (what a streaming round asks for: the reads that go on), any order, a chunk twice, none
    for pick in ([7, 2, 2, 11, 0], list(range(12)), [5], []):
        got, goff = device.download_anchors_of(pick)
        assert len(goff) == len(pick) + 1 and goff[0] == 0
        for k, c in enumerate(pick):
            want = anchors[int(aoff[c]):int(aoff[c + 1])]
            assert got[int(goff[k]):int(goff[k + 1])].tobytes() == want.tobytes(), (pick, k)
    assert len(anchors) > 0
    with pytest.raises(RuntimeError):
        device.download_anchors_of([12])
    second = [x[4000:8000] for x in pa]
    samples2, offsets2 = _batch(second)
    qstart = np.array([r["n_features"] if r["n_features"] > 50 else 0 for r in r1], np.uint32)
    device.upload_batch(samples2, offsets2, query_start_offsets=qstart, carried=anchors,
                        carried_offsets=aoff.astype(np.uint32))
    device.run(3); device.sync()
    r2 = device.download()
    p = {"stop_ratio": 1e30, "stop_mean_ratio": 1e30, "stop_min_anchors": 1 << 30}  # never stop early
    for i, x in enumerate(pa):
        m = po.map_read(oracle_tree, ix.positions, x[:8000], 3, ref_lens, params=p)
        assert m.chunks_used == 2
        assert r2[i]["n_chains"] == m.n_chains
        if m.n_chains:
            assert r2[i]["chains"][0]["score"] == np.float32(m.s1)
            assert r2[i]["chains"][0]["n_anchors"] == m.cm
            ad = np.float32(r2[i]["c0_sum_dist"]) / np.float32(m.cm)
            assert np.float32(ad) == np.float32(m.ad)


# ------------------------------------------------------------------ properties at BASELINE size

def test_full_size_properties():
    """10 000 x 4000-sample chunks against a 12 Mb (16 x 750 kb) index: too big for the oracle in
    a test, so check size-independent properties: (1) determinism (two runs bit-identical),
    (2) scheduling independence (walker occupancy 5 vs 2 workgroups per CU bit-identical), (3) a 64-read sample
    equals the oracle, (4) accuracy vs the generator's truth."""
    genome = synth.make_genome(7, [750000] * 16)
    lm = synth.make_pore_model()
    n = 10000
    raw, truth = synth.simulate_reads(genome, lm, n, 1000)
    ix = pipeline.HostIndex.build(genome, lm, threads=0)
    dev = pipeline.DeviceMapper(0)
    dev.upload_index(ix)
    pa = [po.add_signal(r, synth.DIGITISATION, synth.RANGE, synth.OFFSET) for r in raw]
    samples, offsets = _batch(pa)
    a = dev.map_chunks(samples, offsets, n_refs=16)
    b = dev.map_chunks(samples, offsets, n_refs=16)
    assert a.tobytes() == b.tobytes()
    dev.params.walk_blocks_per_cu = 2
    c = dev.map_chunks(samples, offsets, n_refs=16)
    assert a.tobytes() == c.tobytes()
    dev.params.walk_blocks_per_cu = 0
    dev.params.reference_walk = 1   # nanoflann's exact node set: same records, larger work counters
    r = dev.map_chunks(samples, offsets, n_refs=16)
    dev.params.reference_walk = 0
    for k in a.dtype.names:
        if k in ("nodes_visited", "points_tested"):
            assert (r[k] >= a[k]).all() and r[k].sum() > a[k].sum()
        else:
            assert np.array_equal(r[k], a[k]), k
    assert (a["status"] == 0).all()
    c0 = a["chains"][:, 0]
    ok = (a["n_chains"] > 0) & (c0["ref"] == truth["contig"]) & ((c0["direction"] == 1) == (truth["strand"] == 0))
    assert ok.mean() > 0.97
    tree = po.OracleTree(ix.values.copy())
    ref_lens = np.full(16, 750000, np.uint32)
    names = ["contig%d" % i for i in range(16)]
    idx = np.arange(0, n, n // 64)[:64]
    lines = _paf_lines(a[idx], [pa[i] for i in idx], names, ref_lens)
    for k, i in enumerate(idx):
        m = po.map_read(tree, ix.positions, pa[i], 16, ref_lens)
        assert lines[k] == po.format_paf(m, "read%d" % k, names, ref_lens)
    tree.close(); dev.close()


def test_band_sweep_properties():
    """BASELINE configs[3] shape (one 4.6 Mb contig, band 32/64/128) on 4000 reads: a wider band
    admits a superset of warping paths, so the refined distance can only shrink, and the mapping
    records do not depend on the refinement at all."""
    genome = synth.make_genome(11, [4600000])
    lm = synth.make_pore_model()
    raw, truth = synth.simulate_reads(genome, lm, 4000, 77)
    ix = pipeline.HostIndex.build(genome, lm, threads=0)
    dev = pipeline.DeviceMapper(0)
    dev.upload_index(ix)
    ref_sig, ref_off = pipeline.reference_signals(genome, lm)
    dev.upload_reference(ref_sig, ref_off)
    pa = [po.add_signal(x, synth.DIGITISATION, synth.RANGE, synth.OFFSET) for x in raw]
    samples, offsets = _batch(pa)
    dev.upload_batch(samples, offsets)
    dev.run(1); dev.sync()
    res = dev.download()
    assert (res["status"] == 0).all()
    c0 = res["chains"][:, 0]
    ok = (res["n_chains"] > 0) & ((c0["direction"] == 1) == (truth["strand"] == 0))
    assert ok.mean() > 0.95
    d = {}
    for band in (32, 64, 128):
        dev.refine(band=band, pad=32); dev.sync()
        d[band] = dev.refine_download()
        assert np.array_equal(dev.download().tobytes(), res.tobytes())
    refined = d[32][1] >= 0   # a path that fits the narrow band also fits the wider ones
    assert refined.mean() > 0.9 and (d[64][1] >= 0)[refined].all() and (d[128][1] >= 0)[d[64][1] >= 0].all()
    assert (d[64][0][refined] <= d[32][0][refined]).all() and (d[128][0][refined] <= d[64][0][refined]).all()
    assert (d[128][0][refined] < d[32][0][refined]).any()
    dev.close()


# ------------------------------------------------------------------ CWT feature path (D1-D3)

@pytest.mark.parametrize("scale0", [8.0, 5.0])
def test_cwt_features_match_oracle_and_reference(device, small_world, scale0):
    """MAD normalisation and the peak picker are exact; the DoG CWT is the same linear map as the
    reference's f32-FFT evaluation, so it is held to the FFT's rounding: 1e-5 of the signal's maximum
    against the oracle restatement AND against the real cwt.cc (oracle/_ref) -- measured 2e-7.  A
    peak that hinges on the last bit of a CWT value may move, so peak lists are compared through the
    oracle's peak picker run on the DEVICE's CWT signal (exact), and against the oracle's own
    end-to-end list with a small allowance."""
    pa = small_world["pa"][:12] + [small_world["pa"][0][:1500], np.full(4000, 90.0, np.float32) + np.arange(4000, dtype=np.float32) % 7]
    samples, offsets = _batch(pa)
    feat, pos, cnt, rows = device.cwt_features(samples, offsets, scale0=scale0, want_cwt=True)
    total = same = 0
    for i, x in enumerate(pa):
        n = len(x)
        a, _ = po.mad_normalize(x)
        ocw = po.cwt_dog(a, scale0)
        tol = 1e-5 * float(np.abs(ocw).max())
        assert np.abs(rows[i, :n] - ocw).max() <= tol
        if po.have_ref():
            assert np.abs(rows[i, :n] - po.ref_cwt_dog(a, scale0)).max() <= tol
        # D3 exactly, given the device's own CWT signal
        z = np.zeros(n, np.float32)
        po.lib().so_zscore(po._p(np.ascontiguousarray(rows[i, :n])), C.c_size_t(n), po._p(z))
        # (SD accumulated sequentially in f64 like the reference; numpy's pairwise sums can differ)
        pk, pp = po.generate_peaks(z, _oracle_sd(rows[i, :n]) / np.float32(4))
        assert cnt[i] == len(pk)
        assert np.array_equal(pos[i, :cnt[i]], pp) and np.array_equal(bits(feat[i, :cnt[i]]), bits(pk))
        # end to end against the oracle (its own CWT): the same peaks but for last-bit cases
        opk, opp, _ = po.cwt_features(x, scale0)
        total += len(opp)
        same += len(np.intersect1d(opp, pos[i, :cnt[i]]))
    assert total > 2000 and same >= 0.995 * total


def test_cli_features_mode_dumps_device_features(tmp_path, small_world, device):
    """`sigmap --features events|cwt -s DIR -o FILE` (development mode; the reachable form of the reference's
    disabled EventsToText dump and of its dormant CWT feature path): every whole 4000-sample chunk's feature
    signal as text.  events == the oracle's GenerateEvents (exact, printed %f); cwt == what smhip_cwt_features
    returns for the same chunks (that stage's own parity: test_cwt_features_match_oracle_and_reference)."""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "sigmap_amd", "bin", "sigmap")
    genome, lm = small_world["genome"], small_world["lm"]
    raw2, _ = synth.simulate_reads(genome, lm, 3, 21, n_chunks=2)
    raw1, _ = synth.simulate_reads(genome, lm, 2, 22, n_chunks=1)
    raws = [r for r in raw2] + [r for r in raw1] + [np.asarray(raw1[0][:3000])]     # (the last: no whole chunk)
    os.makedirs(tmp_path / "reads")
    synth.write_sgr(str(tmp_path / "reads" / "a.sgr"), raws)
    chunks = []   # (read name, chunk, pA samples)
    for i, rw in enumerate(raws):
        x = po.add_signal(rw, synth.DIGITISATION, synth.RANGE, synth.OFFSET)
        for c in range(len(x) // 4000):
            chunks.append(("read%d" % i, c, x[c * 4000:(c + 1) * 4000]))
    assert len(chunks) == 8
    for kind in ("events", "cwt"):
        out = tmp_path / (kind + ".txt")
        r = subprocess.run([exe, "--features", kind, "-s", str(tmp_path / "reads"), "-o", str(out)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        got = [l.split("\t") for l in open(out).read().splitlines()]
        want = []
        if kind == "events":
            for name, c, x in chunks:
                f, _ = po.generate_events(x)
                want += [[name, str(c), str(k), "0", "%f" % v] for k, v in enumerate(f)]
        else:
            samples, offsets = _batch([x for _, _, x in chunks])
            feat, pos, cnt, _ = device.cwt_features(samples, offsets, scale0=8.0)
            for j, (name, c, _) in enumerate(chunks):
                want += [[name, str(c), str(k), str(int(pos[j, k])), "%f" % feat[j, k]] for k in range(int(cnt[j]))]
        assert got == want and len(got) > 1000
    r = subprocess.run([exe, "--features", "mad", "-s", str(tmp_path / "reads"), "-o", str(tmp_path / "x")], capture_output=True, text=True)
    assert r.returncode != 0 and "events" in r.stderr


def _oracle_sd(x):
    """SD exactly as GenerateZscoreNormalizedSignal accumulates it (sequential f64), narrowed to f32."""
    mean = 0.0
    for v in x:
        mean += float(v)
    mean /= len(x)
    sd = 0.0
    for v in x:
        sd += (float(v) - mean) * (float(v) - mean)
    sd /= (len(x) - 1)
    return np.float32(np.sqrt(sd))


# ------------------------------------------------------------------ banded sDTW (D4/D5)

def test_sdtw_kernel_matches_oracle(device):
    rng = np.random.default_rng(3)
    targets, queries, los, his, exp = [], [], [], [], []
    for case in range(12):
        T = int(rng.integers(50, 700)); Q = int(rng.integers(20, 200))
        t = rng.normal(size=T).astype(np.float32)
        s = int(rng.integers(0, max(1, T - Q)))
        q = (t[s:s + Q] + rng.normal(scale=0.1, size=min(Q, T - s))).astype(np.float32)
        Q = len(q)
        if case % 3 == 0:      # full band == unbanded Sigmap::sDTW
            lo = np.ones(T, np.uint32); hi = np.full(T, Q, np.uint32)
            d, e = po.sdtw(t, q)
            d2, e2 = po.sdtw_banded(t, q, lo, hi)
            assert d == d2 and e == e2
        else:                  # diagonal band of half-width w around the planted start
            w = [16, 32, 64][case % 3]
            centre = np.arange(1, T + 1) - s
            lo = np.clip(centre - w, 1, Q).astype(np.uint32); hi = np.clip(centre + w, 0, Q).astype(np.uint32)
            d, e = po.sdtw_banded(t, q, lo, hi)
        targets.append(t); queries.append(q); los.append(lo); his.append(hi); exp.append((d, e))
    t_off = np.concatenate([[0], np.cumsum([len(t) for t in targets])]).astype(np.uint64)
    q_off = np.concatenate([[0], np.cumsum([len(q) for q in queries])]).astype(np.uint64)
    dist, end = device.sdtw_banded(np.concatenate(targets), t_off, np.concatenate(queries), q_off,
                                   np.concatenate(los), np.concatenate(his))
    for k, (d, e) in enumerate(exp):
        # f32 adds in the same order as the reference recurrence: bit-exact, well inside the
        # 1e-4 relative tolerance north_star allows for DTW distance
        assert bits(np.float32(dist[k])) == bits(np.float32(d)), (k, dist[k], d)
        assert end[k] == e


# ------------------------------------------------------------------ the CLI end to end

def test_cli_index_and_streaming_map_match_oracle(tmp_path, small_world):
    """`sigmap -i` then `sigmap -m` (sigmap_amd/bin/sigmap): multi-chunk reads with early stop and
    chain carry-over; PAF (minus the wall-clock mt tag) must equal the oracle's streaming result,
    in the reference's output order (reference index, then read order)."""
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "sigmap_amd", "bin", "sigmap")
    assert os.path.exists(exe), "CLI not built (make cli)"
    genome, lm = small_world["genome"], small_world["lm"]
    names = ["contig0", "contig1", "contig2"]
    synth.write_fasta(str(tmp_path / "ref.fa"), genome, names)
    synth.write_pore_model(str(tmp_path / "model.txt"), lm)
    raw3, _ = synth.simulate_reads(genome, lm, 10, 5, n_chunks=3)
    raw1, _ = synth.simulate_reads(genome, lm, 10, 6, n_chunks=1)
    raws = [r for r in raw3] + [r for r in raw1] + [np.full(4000, 600, np.int16)]  # + a flat read
    # + reads with ADC samples outside (30, 200) pA: the device drops them (signal_batch.cc:141-147) and the
    # chunk boundaries move with the FILTERED array; the second one is a sample short of a third chunk
    raws.append(np.insert(raw3[0], 500, np.full(100, 32767, np.int16)))
    raws.append(np.concatenate([raw3[1][:6000], np.full(37, -32768, np.int16), raw3[1][6000:11999]]))
    os.makedirs(tmp_path / "reads")
    synth.write_sgr(str(tmp_path / "reads" / "a.sgr"), raws)
    r = subprocess.run([exe, "-i", "-r", str(tmp_path / "ref.fa"), "-p", str(tmp_path / "model.txt"),
                        "-o", str(tmp_path / "idx")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "-m", "-r", str(tmp_path / "ref.fa"), "-p", str(tmp_path / "model.txt"),
                        "-x", str(tmp_path / "idx"), "-s", str(tmp_path / "reads"), "-o", str(tmp_path / "out.paf")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = [l for l in open(tmp_path / "out.paf").read().splitlines()]
    strip = lambda l: "\t".join(f for f in l.split("\t") if not f.startswith("mt:f:"))
    # oracle: index files written by the CLI are read by the oracle's own reader
    dim, ml, pos, val, n = C.c_int(), C.c_int(), C.c_void_p(), C.c_void_p(), C.c_uint64()
    assert po.lib().so_points_load(str(tmp_path / "idx.pt").encode(), C.byref(dim), C.byref(ml), C.byref(pos),
                                   C.byref(val), C.byref(n)) == 0
    values = np.ctypeslib.as_array(C.cast(val, C.POINTER(C.c_float)), shape=(n.value,)).copy()
    positions = np.ctypeslib.as_array(C.cast(pos, C.POINTER(C.c_uint64)), shape=(n.value,)).copy()
    tree = po.OracleTree.load(str(tmp_path / "idx.si"), values)
    ref_lens = small_world["ref_lens"]
    rows = []
    for i, rw in enumerate(raws):
        pa = po.add_signal(rw, synth.DIGITISATION, synth.RANGE, synth.OFFSET)
        m = po.map_read(tree, positions, pa, 3, ref_lens)
        rows.append((m.ref, i, po.format_paf(m, "read%d" % i, names, ref_lens).rstrip("\n")))
    rows.sort(key=lambda t: (t[0], t[1]))
    exp = [t[2] for t in rows]
    assert [strip(l) for l in got] == exp
    assert sum(1 for l in exp if "\t*\t" not in l) >= 18
    tree.close()
    # the same reads as one multi-read FAST5 file (signal_batch.cc:69-151): identical PAF
    if _abi.host_lib().smhost_have_fast5():
        os.makedirs(tmp_path / "reads5")
        synth.write_fast5(str(tmp_path / "reads5" / "a.fast5"), raws, ["read%03d" % i for i in range(len(raws))])
        r = subprocess.run([exe, "-m", "-r", str(tmp_path / "ref.fa"), "-p", str(tmp_path / "model.txt"),
                            "-x", str(tmp_path / "idx"), "-s", str(tmp_path / "reads5"), "-o", str(tmp_path / "out5.paf")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        got5 = [strip(l) for l in open(tmp_path / "out5.paf").read().splitlines()]
        rename = lambda l: "\t".join(["read%d" % int(l.split("\t")[0][4:])] + l.split("\t")[1:])
        assert [rename(l) for l in got5] == exp
    # the same reads as BLOW5 files (signal_batch.cc:153-210; one plain, one with zlib records): identical PAF
    os.makedirs(tmp_path / "readsb")
    half = len(raws) // 2
    synth.write_blow5(str(tmp_path / "readsb" / "a.blow5"), raws[:half], ["read%d" % i for i in range(half)])
    synth.write_blow5(str(tmp_path / "readsb" / "b.blow5"), raws[half:], ["read%d" % i for i in range(half, len(raws))],
                      record_press=1)
    r = subprocess.run([exe, "-m", "-r", str(tmp_path / "ref.fa"), "-p", str(tmp_path / "model.txt"),
                        "-x", str(tmp_path / "idx"), "-s", str(tmp_path / "readsb"), "-o", str(tmp_path / "outb.paf")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert [strip(l) for l in open(tmp_path / "outb.paf").read().splitlines()] == exp
    # --sdtw-band adds a dt:f: tag to rows that have a chain and changes nothing else
    r = subprocess.run([exe, "-m", "-r", str(tmp_path / "ref.fa"), "-p", str(tmp_path / "model.txt"),
                        "-x", str(tmp_path / "idx"), "-s", str(tmp_path / "reads"), "-o", str(tmp_path / "out2.paf"),
                        "--sdtw-band", "64"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got2 = open(tmp_path / "out2.paf").read().splitlines()
    nodt = lambda l: "\t".join(f for f in l.split("\t") if not f.startswith(("mt:f:", "dt:f:")))
    assert [nodt(l) for l in got2] == exp
    assert sum(1 for l in got2 if "\tdt:f:" in l) >= 18
    # --sdtw-cigar adds the banded DTW() path's per-base CIGAR (fast_dtw.cc print_alignment) and nothing else
    r = subprocess.run([exe, "-m", "-r", str(tmp_path / "ref.fa"), "-p", str(tmp_path / "model.txt"),
                        "-x", str(tmp_path / "idx"), "-s", str(tmp_path / "reads"), "-o", str(tmp_path / "out3.paf"),
                        "--sdtw-band", "64", "--sdtw-cigar"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got3 = open(tmp_path / "out3.paf").read().splitlines()
    nocg = lambda l: "\t".join(f for f in l.split("\t") if not f.startswith(("mt:f:", "cg:Z:")))
    assert [nocg(l) for l in got3] == ["\t".join(f for f in l.split("\t") if not f.startswith("mt:f:")) for l in got2]
    cgs = [f for l in got3 for f in l.split("\t") if f.startswith("cg:Z:")]
    assert len(cgs) >= 18 and all(c.startswith("cg:Z:(") and c.endswith(")") and c.count("M") > 50 for c in cgs)
    # the number of GPUs, of device contexts / streams and the chunk budget never show in the output
    # (default above: 1 GPU, 2 streams; here one stream, three streams with 3-chunk batches each, and
    # the multi-GPU code path with one GPU)
    for k, extra in enumerate((["--streams", "1"], ["--streams", "3", "--batch-chunks", "9", "--sdtw-band", "64"],
                               ["--gpus", "1", "--streams", "2"])):
        r = subprocess.run([exe, "-m", "-r", str(tmp_path / "ref.fa"), "-p", str(tmp_path / "model.txt"),
                            "-x", str(tmp_path / "idx"), "-s", str(tmp_path / "reads"),
                            "-o", str(tmp_path / ("out_s%d.paf" % k))] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        gots = open(tmp_path / ("out_s%d.paf" % k)).read().splitlines()
        assert [nodt(l) for l in gots] == exp
        if "--sdtw-band" in extra:
            dt = lambda l: [f for f in l.split("\t") if f.startswith("dt:f:")]
            assert [dt(l) for l in gots] == [dt(l) for l in got2]
    # a chunk budget the device's memory cannot hold (the 250 Mb reference with the default --batch-chunks 32768: 20 MB
    # per chunk, tests/test_host_index.py has the arithmetic): the CLI shrinks the rounds, says so, maps on -- here with
    # the free memory it may count on faked down to 64 MiB, which leaves a context a handful of chunks per round
    r = subprocess.run([exe, "-m", "-r", str(tmp_path / "ref.fa"), "-p", str(tmp_path / "model.txt"),
                        "-x", str(tmp_path / "idx"), "-s", str(tmp_path / "reads"), "-o", str(tmp_path / "out_m.paf"),
                        "--batch-chunks", "32768"], capture_output=True, text=True,
                       env=dict(os.environ, SIGMAP_DEV_FREE_MB="64"))
    assert r.returncode == 0, r.stderr
    assert "Chunks in flight per context limited to" in r.stderr and "by the device's free memory (64 MiB)" in r.stderr
    assert [strip(l) for l in open(tmp_path / "out_m.paf").read().splitlines()] == exp
    # more GPUs than the box has: a loud failure, no partial output
    n_gpu = _abi.hip_lib().smhip_device_count()
    r = subprocess.run([exe, "-m", "-r", str(tmp_path / "ref.fa"), "-p", str(tmp_path / "model.txt"),
                        "-x", str(tmp_path / "idx"), "-s", str(tmp_path / "reads"), "-o", str(tmp_path / "out_g.paf"),
                        "--gpus", str(n_gpu + 1)], capture_output=True, text=True)
    assert r.returncode != 0 and "GPU" in r.stderr and not os.path.exists(tmp_path / "out_g.paf")
    if n_gpu >= 2:   # (never on the 1-GPU test boxes) index broadcast over RCCL + read sharding
        r = subprocess.run([exe, "-m", "-r", str(tmp_path / "ref.fa"), "-p", str(tmp_path / "model.txt"),
                            "-x", str(tmp_path / "idx"), "-s", str(tmp_path / "reads"), "-o", str(tmp_path / "out_g2.paf"),
                            "--gpus", "2"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert [nodt(l) for l in open(tmp_path / "out_g2.paf").read().splitlines()] == exp


# ------------------------------------------------------------------ sDTW refinement of chain 0 (D4/D5)

@pytest.mark.parametrize("band", [32, 64, 128])
def test_refine_matches_oracle(device, small_world, band):
    """smhip_refine_run: banded sDTW of the chunk's features against the z-scored reference signal
    around chain 0 (window + band exactly as include/sigmap_hip.h states) == the oracle's banded
    sDTW on the same window.  Also: refinement does not touch the mapping results."""
    genome, lm, pa = small_world["genome"], small_world["lm"], small_world["pa"][:24]
    sig, offs = pipeline.reference_signals(genome, lm)
    device.upload_reference(sig, offs)
    samples, offsets = _batch(pa)
    device.upload_batch(samples, offsets)
    device.run(3); device.sync()
    before = device.download()
    pad = 32
    device.refine(band=band, pad=pad); device.sync()
    dist, end = device.refine_download()
    assert before.tobytes() == device.download().tobytes()
    feat, _, nf, _, _ = device.events(samples, offsets)
    checked = 0
    for i in range(len(pa)):
        r = before[i]
        if r["n_chains"] == 0:
            assert end[i] == -1
            continue
        c0 = r["chains"][0]
        strand = 0 if c0["direction"] == 1 else 1
        so, se = int(offs[2 * c0["ref"] + strand]), int(offs[2 * c0["ref"] + strand + 1])
        start, endp = int(c0["start"]), int(c0["end"])
        t0 = max(0, start - pad); t1 = min(se - so, endp + 6 + pad)
        target = sig[so + t0:so + t1]
        q = feat[i, :nf[i]]
        qa, qb = int(r["c0_last_qpos"]), int(r["c0_first_qpos"])
        den = max(1, endp - start)
        ta = np.arange(t0, t1, dtype=np.int64)
        centre = 1 + qa + ((ta - start) * (qb - qa)) // den       # floor division
        lo = np.clip(centre - band // 2, 1, None).astype(np.int64)
        hi = np.clip(centre + band // 2, None, len(q)).astype(np.int64)
        hi = np.maximum(hi, 0)
        d, e = po.sdtw_banded(target, q, lo.astype(np.uint32), hi.astype(np.uint32))
        if e < 0:
            assert end[i] == -1
        else:
            assert bits(np.float32(dist[i])) == bits(np.float32(d)), (i, dist[i], d)
            assert end[i] == t0 + e
            # the refined end lies near the chain's end on the reference
            assert abs(int(end[i]) - (endp + 6)) < 64
        checked += 1
    assert checked >= 20


@pytest.mark.gpu
def test_quad_records_follow_the_binary_records(device):
    """The order-free search's node array (seed_lookup.hip BuildQuads) against the binary records it is derived from:
    the heads are exactly the inner nodes at even depth, numbered in pre-order; a head's slots are the nodes two levels
    below it (a leaf child takes a slot itself) with the boxes the parents' binary records hold and references to
    quads / leaves; unused slots carry no reference and a box nothing is close to; the box grid has one step."""
    d = device.export_index()
    assert d.n_quads > 0 and d.quads_bytes == d.n_quads * 128
    nodes = np.empty(d.nodes_bytes // 4, dtype=np.uint32)
    quads = np.empty(d.quads_bytes // 4, dtype=np.uint32)
    bbox = np.empty(d.bbox_bytes // 4, dtype=np.float32)
    device.export_copy(0, nodes.ctypes.data, nodes.nbytes)   # (smhip_index_export_copy also copies to host memory)
    device.export_copy(5, quads.ctypes.data, quads.nbytes)
    device.export_copy(4, bbox.ctypes.data, bbox.nbytes)
    nodes = nodes.reshape(-1, 16)
    quads = quads.reshape(-1, 32)
    assert len(set(bbox[18:24].tolist())) == 1 and bbox[18] > 0           # one grid step for all dimensions
    LEAF, MASK = 0x80000000, 0x87FFFFFF

    def child(rec, which):   # (reference, six box words {lo01, lo23, lo45, hi01, hi23, hi45}) of child 0 / 1
        ref = int(rec[2 + which]) & MASK
        w = rec[4:16]
        return ref, (w[0:6] if which == 0 else w[6:12])

    # heads in pre-order = increasing inner index (inner records are numbered in pre-order)
    depth = np.full(len(nodes), -1, dtype=np.int64)
    depth[0] = 0
    for x in range(len(nodes)):          # a node's children have larger indices: one forward pass
        assert depth[x] >= 0
        for which in (0, 1):
            ref, _ = child(nodes[x], which)
            if not ref & LEAF:
                depth[ref] = depth[x] + 1
    heads = np.flatnonzero(depth % 2 == 0)
    assert len(heads) == d.n_quads
    quad_of = {int(x): k for k, x in enumerate(heads)}
    for k, x in enumerate(heads):
        want = []
        for which in (0, 1):
            ref, box = child(nodes[x], which)
            if ref & LEAF:
                want.append((ref, box))
            else:
                for w2 in (0, 1):
                    g, gbox = child(nodes[ref], w2)
                    want.append((g if g & LEAF else quad_of[g], gbox))
        q = quads[k]
        for s in range(4):
            if s < len(want):
                assert int(q[s]) == want[s][0] and np.array_equal(q[4 + 6 * s:10 + 6 * s], want[s][1]), (k, s)
            else:
                assert int(q[s]) == 0xFFFFFFFF
                assert (q[4 + 6 * s:7 + 6 * s] == 0xFFFFFFFF).all() and (q[7 + 6 * s:10 + 6 * s] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("n_bases", [480, 700, 1500])
def test_pipeline_small_trees_match_oracle(small_world, n_bases):
    """Whole path on references of a few hundred bases: trees of three to seven levels, where the quad records of
    the order-free search have leaves in their first-level slots, unused slots and heads with a single inner
    grandchild; anchors, hit totals and chains against the oracle."""
    lm = small_world["lm"]
    rng = np.random.default_rng(n_bases)
    unit = bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n_bases)])
    host = _abi.host_lib()
    sig = np.zeros(len(unit), np.float32)
    m = host.smhost_reference_signal(_abi.ptr(np.ascontiguousarray(lm)), 6, unit, len(unit), 0, _abi.ptr(sig))
    values = sig[:m].astype(np.float32)
    positions = (np.arange(len(values), dtype=np.uint64) << np.uint64(1))  # one contig, + strand
    ix = pipeline.HostIndex.from_points(positions, values, threads=1)
    raw, _ = synth.simulate_reads([unit], lm, 5, 3)
    pa = [po.add_signal(r, synth.DIGITISATION, synth.RANGE, synth.OFFSET) for r in raw]
    samples, offsets = _batch(pa)
    tree = po.OracleTree(ix.values.copy())
    dev = pipeline.DeviceMapper(0)
    dev.upload_index(ix)
    d = dev.export_index()
    assert d.n_quads > 0
    res = dev.map_chunks(samples, offsets, n_refs=1)
    assert int(dev.debug_counters()[6]) == 2          # searched order-free, over the quad records
    dev.close()
    assert res["n_anchors"].sum() > 0
    for i, x in enumerate(pa):
        f, _ = po.generate_events(x)
        chains, st = po.generate_chains(tree, ix.positions, f, 1)
        assert res[i]["n_anchors"] == st.anchors and res[i]["hits_total"] == st.hits and res[i]["n_chains"] == len(chains)
        for k, c in enumerate(chains[:4]):
            g = res[i]["chains"][k]
            assert g["score"] == np.float32(c["score"]) and g["start"] == c["start"] and g["end"] == c["end"]
    tree.close()

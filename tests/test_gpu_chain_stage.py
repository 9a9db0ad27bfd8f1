"""The chaining stage in isolation (smhip_chain_anchors, and smhip_generate_chains from features): anchors
handed in directly -- not engineered through point clouds -- against the oracle's so_chain_sorted_anchors
(spatial_index.cc:411-576), on constructions that sit exactly on the DP's exits and tie rules, plus tangled random
ones.  Everything is compared: every primary chain's score bits, bucket, span, anchor count, chain 0's MAPQ and
statistics, and the anchor lists of all primary chains."""
import numpy as np
import pytest

from conftest import bits
from oracle import pyoracle as po
from sigmap_amd import _abi, pipeline

pytestmark = pytest.mark.gpu

D0 = np.float32(0.01)


def _diag(t0, q0, n, step=10, qstep=None, rs=0, dist=D0):
    """n collinear anchors: targets t0, t0 + step, ...; queries q0, q0 + qstep, ..."""
    qstep = step if qstep is None else qstep
    k = np.arange(n)
    return [np.full(n, rs, np.uint32), (t0 + step * k).astype(np.uint32), (q0 + qstep * k).astype(np.uint32),
            np.full(n, dist, np.float32)]


def _cat(*parts):
    return [np.concatenate([p[i] for p in parts]) for i in range(4)]


def _one(rs, t, q, d=D0):
    return [np.array([rs], np.uint32), np.array([t], np.uint32), np.array([q], np.uint32), np.array([d], np.float32)]


def _run_and_compare(dev, chunks, n_refs, shuffle_seed=0):
    """chunks: list of [ref_strand, target, query, distance]; device (anchors shuffled: the entry point sorts) vs oracle.
    Returns the oracle's chains per chunk."""
    rng = np.random.default_rng(shuffle_seed)
    flat, offs = [], [0]
    for c in chunks:
        perm = rng.permutation(len(c[0]))
        a = np.zeros(len(perm), dtype=_abi.CARRIED_DTYPE)
        a["ref_strand"], a["target_position"], a["query_position"], a["distance"] = (x[perm] for x in c)
        flat.append(a)
        offs.append(offs[-1] + len(a))
    res, anchors, aoff = dev.chain_anchors(np.concatenate(flat) if flat else np.zeros(0, _abi.CARRIED_DTYPE), offs, n_refs,
                                           with_anchors=True)
    allc, coff = dev.download_chains()      # every primary chain, not only the record's first four
    # the classifier's other routes -- lockstep predecessor scans for every tile instead of dense pairs; every anchor
    # with a passing predecessor left to the DP kernels instead of chained in the classifier where that is possible --:
    # the same bytes
    for knob in ("classify_dense", "classify_resolve"):
        dev.set_knob(knob, 0)
        try:
            res0, anchors0, aoff0 = dev.chain_anchors(np.concatenate(flat) if flat else np.zeros(0, _abi.CARRIED_DTYPE), offs,
                                                      n_refs, with_anchors=True)
            allc0, coff0 = dev.download_chains()
        finally:
            dev.set_knob(knob, -1)
        assert res0.tobytes() == res.tobytes() and anchors0.tobytes() == anchors.tobytes() and np.array_equal(aoff0, aoff), knob
        assert allc0.tobytes() == allc.tobytes() and np.array_equal(coff0, coff), knob
    out = []
    for i, c in enumerate(chunks):
        want = po.chain_anchors(c[0], c[1], c[2], c[3], n_refs)
        r = res[i]
        assert int(coff[i + 1] - coff[i]) == len(want), i
        for k, ch in enumerate(want):
            g = allc[int(coff[i]) + k]
            assert bits(np.float32(g["score"])) == bits(np.float32(ch["score"])), (i, k)
            assert (g["ref"], g["start"], g["end"], g["n_anchors"], g["direction"]) == \
                   (ch["ref"], ch["start"], ch["end"], ch["n_anchors"], ch["direction"]), (i, k)
            assert g["mapq"] == (ch["mapq"] if k == 0 else 0), (i, k)
        assert r["status"] == 0 and r["n_anchors"] == len(c[0]) and r["n_queries"] == 0, (i, r["status"])
        assert r["n_chains"] == len(want), (i, r["n_chains"], len(want))
        got_anchors = anchors[int(aoff[i]):int(aoff[i + 1])]
        pos = 0
        for k, ch in enumerate(want):
            if k < 4:
                g = r["chains"][k]
                assert bits(np.float32(g["score"])) == bits(np.float32(ch["score"])), (i, k, g["score"], ch["score"])
                assert (g["ref"], g["start"], g["end"], g["n_anchors"], g["direction"]) == \
                       (ch["ref"], ch["start"], ch["end"], ch["n_anchors"], ch["direction"]), (i, k)
                if k == 0:
                    assert g["mapq"] == ch["mapq"], (i, g["mapq"], ch["mapq"])
            # anchors of every primary chain, chain-major, end -> start
            n = ch["n_anchors"]
            ga = got_anchors[pos:pos + n]
            assert len(ga) == n, (i, k)
            assert np.array_equal(ga["target_position"], ch["anchors"]["target_position"]), (i, k)
            assert np.array_equal(ga["query_position"], ch["anchors"]["query_position"]), (i, k)
            assert np.array_equal(bits(ga["distance"]), bits(ch["anchors"]["distance"])), (i, k)
            assert (ga["ref_strand"] == ((ch["ref"] << 1) | (0 if ch["direction"] == 1 else 1))).all(), (i, k)
            pos += n
        assert pos == len(got_anchors), i
        if want:   # chain 0's statistics in the reference's accumulation order (sigmap.cc:714-729)
            a = want[0]["anchors"]
            sd = np.float32(0); st = np.float32(0); sq = np.float32(0)
            for j in range(len(a)):
                sd = np.float32(sd + a["distance"][j])
                if j + 1 < len(a):
                    st = np.float32(st + np.float32(int(a["target_position"][j]) - int(a["target_position"][j + 1])))
                    sq = np.float32(sq + np.float32(int(a["query_position"][j]) - int(a["query_position"][j + 1])))
            assert bits(r["c0_sum_dist"]) == bits(sd) and bits(r["c0_sum_tgap"]) == bits(st) and bits(r["c0_sum_qgap"]) == bits(sq), i
            assert r["c0_first_qpos"] == a["query_position"][0] and r["c0_last_qpos"] == a["query_position"][-1]
            ssum = np.float32(0)
            for ch in want:
                ssum = np.float32(ssum + np.float32(ch["score"]))
            assert bits(r["sum_score"]) == bits(ssum), i
        out.append(want)
    return out


@pytest.fixture(scope="module")
def cdev(small_world):
    dev = pipeline.DeviceMapper(0)
    dev.upload_index(small_world["ix"])      # (the stage needs no index; the context does)
    yield dev
    dev.close()


def test_skip_exit_at_exactly_25_and_26_skips(cdev):
    """num_skips > max_num_skips (25) ends a predecessor scan (spatial_index.cc:527-531): with 25 failing predecessors
    between an anchor and the one it chains to, the chain goes through; with 26 it does not."""
    chunks = []
    for k in (24, 25, 26, 27):
        trunk = _diag(9890, 9890, 11)                       # ..., (9990, 9990)
        # k anchors that reach the score comparison with a score of 0 (gap scale 600+, gap length >= 2995)
        bad = [np.zeros(k, np.uint32), np.full(k, 9995, np.uint32), (10000 - 3000 - np.arange(k)).astype(np.uint32),
               np.full(k, D0, np.float32)]
        chunks.append(_cat(trunk, bad, _one(0, 10000, 10000)))
    want = _run_and_compare(cdev, chunks, 1)
    n0 = [w[0]["n_anchors"] for w in want]
    assert n0[0] == 12 and n0[1] == 12 and n0[2] == 11 and n0[3] == 11, n0     # X joins the trunk iff <= 25 skips


def test_band_of_5000_predecessors(cdev):
    """A scan looks at most chaining_band_length = 5000 anchors back (spatial_index.cc:452-455): predecessors that only
    `continue` (same query position) do not count as skips, so the anchor 4999 / 5000 / 5001 places back is reached /
    reached / not reached."""
    chunks = []
    for nmid in (4998, 4999, 5000, 5001):
        trunk = _diag(1000, 1000, 11, step=100)             # ..., (2000, 2000)
        mid = [np.zeros(nmid, np.uint32), (2001 + np.arange(nmid) % 2400).astype(np.uint32), np.full(nmid, 4500, np.uint32),
               (D0 + np.float32(1e-6) * (np.arange(nmid) // 2400)).astype(np.float32)]
        chunks.append(_cat(trunk, mid, _one(0, 4500, 4500)))   # X: 2500 after the trunk's end on both axes
    want = _run_and_compare(cdev, chunks, 1)
    best = [max(ch["n_anchors"] for ch in w if ch["end"] == 4500) if any(ch["end"] == 4500 for ch in w) else 0 for w in want]
    # X at index 11 + nmid looks back to index 11 + nmid - 5000: the trunk's end (index 10) is seen iff nmid <= 4999
    assert best[0] == 12 and best[1] == 12 and best[2] < 12 and best[3] < 12, best


def test_target_gap_of_5000_and_5001(cdev):
    """previous.target + 5000 < current.target ends the scan (spatial_index.cc:476-479): a gap of exactly 5000 chains.
    (gap length |td - qd| must stay below 2000 and the scale inside (0.75, 5): qd = td - 1000.)"""
    chunks = []
    for gap in (4999, 5000, 5001, 5002):
        chunks.append(_cat(_diag(1000, 1000, 11), _one(0, 1100 + gap, 1100 + gap - 1000)))
    want = _run_and_compare(cdev, chunks, 1)
    n0 = [w[0]["n_anchors"] for w in want]
    assert n0 == [12, 12, 11, 11], n0


def test_gap_scale_and_gap_length_bounds(cdev):
    """0.75 < (float)qd / td < 5 and |td - qd| < 2000, all strict (spatial_index.cc:489-497): pairs on and next to
    every bound, including 4 qd == 3 td and qd == 5 td at several magnitudes (the kernels test them in integers)."""
    pairs = [(4, 3), (400, 300), (4000, 3000), (4001, 3001), (3999, 3000), (1, 5), (100, 500), (100, 499), (101, 500),
             (399, 1995), (400, 1999), (400, 2000), (1000, 2999), (1000, 3000), (2999, 1000), (3000, 1000), (2665, 666),
             (2664, 666), (8, 6), (8, 7), (7, 5), (4999, 3750), (5000, 3750), (5000, 3751), (1333, 1000), (1334, 1000),
             (1, 1), (2, 10), (2, 9), (6, 30), (6, 29), (1200, 6000), (1201, 6000), (800, 2799), (800, 2800)]
    chunks, expect = [], []
    for td, qd in pairs:
        # P -> G is a chain of two (scale 0.8, gap length 1000); X = G + (td, qd).  P lies exactly 5000 before G on the
        # target, i.e. more than 5000 before X: X's scan ends there, (td, qd) against G is the only test that counts
        chunks.append(_cat(_one(0, 20000, 20000), _one(0, 25000, 24000), _one(0, 25000 + td, 24000 + qd)))
        scale = np.float32(qd) / np.float32(td)
        expect.append(3 if (abs(td - qd) < 2000 and scale < np.float32(5) and scale > np.float32(0.75)) else 2)
    want = _run_and_compare(cdev, chunks, 1)
    assert [w[0]["n_anchors"] for w in want] == expect
    assert expect.count(2) >= 12 and expect.count(3) >= 12


def test_equal_target_query_runs_and_exact_duplicates(cdev):
    """Anchors with the same (target, query) and different distances (ordered by distance, SignalAnchor::operator<),
    exact duplicates, the same target with different queries and vice versa -- the `continue` rules at
    spatial_index.cc:468-475 and the sort's tie order."""
    rng = np.random.default_rng(4)
    chunks = []
    for rep in range(6):
        base = _diag(5000, 500, 30, step=7)
        dup = [x.copy() for x in base]
        dup[3] = (dup[3] + rng.uniform(0, 0.05, size=30).astype(np.float32)).astype(np.float32)   # same (t, q), other distance
        same_t = [base[0][:10], base[1][:10], (base[2][:10] + rng.integers(1, 40, size=10)).astype(np.uint32), base[3][:10]]
        same_q = [base[0][:10], (base[1][:10] + rng.integers(1, 40, size=10)).astype(np.uint32), base[2][:10], base[3][:10]]
        chunks.append(_cat(base, dup, base, same_t, same_q))
    want = _run_and_compare(cdev, chunks, 1, shuffle_seed=5)
    assert all(len(w) >= 1 and w[0]["n_anchors"] >= 20 for w in want)


def test_forks_used_predecessors_and_more_than_three_ends(cdev):
    """A trunk with two continuations that cannot chain to each other: the second end's traceback stops at an anchor
    the first one used (spatial_index.cc:165-215).  Six separate loci in one bucket: only num_best_chains = 3 ends are
    traced, in (score, index) order; the running maximum couples the buckets' end tests."""
    chunks = []
    # fork: trunk (t, q) 0..19 step 10; branch A continues on the diagonal, branch B has the same targets shifted by 3
    # and queries BELOW A's (qd < 0 against A's anchors: `continue`), so both only see the trunk
    trunk = _diag(20000, 2000, 20)
    a = _diag(20200, 2200, 8)
    b = _diag(20203, 2190, 8, step=10, qstep=1, dist=np.float32(0.012))
    chunks.append(_cat(trunk, a, b))
    # six loci, more than 5000 apart on the target, lengths 12, 12, 11, 10, 12, 9
    loci = [_diag(10000 + 9000 * k, 100 + 5 * k, n, dist=np.float32(0.01 + 0.001 * k)) for k, n in enumerate((12, 12, 11, 10, 12, 9))]
    chunks.append(_cat(*loci))
    # the same over three buckets of two references: 2 + 2 + 2 loci, and a weak bucket that fails max / 2
    chunks.append(_cat(_diag(3000, 30, 14, rs=0), _diag(20000, 30, 13, rs=0), _diag(3000, 30, 14, rs=1), _diag(3000, 31, 12, rs=2),
                       _diag(30000, 31, 12, rs=2), _diag(500, 30, 3, rs=3)))
    want = _run_and_compare(cdev, chunks, 2)
    assert len(want[0]) == 1 and want[0][0]["n_anchors"] == 28     # branch B overlaps the first chain's span: not primary
    assert len(want[1]) == 3                                       # three ends traced out of six


def test_ties_of_the_chain_order_across_buckets(cdev):
    """Identical loci on both strands of three references: scores and anchor counts tie, the chains are ranked by the
    remaining fields of SignalAnchorChain::operator> (direction, reference, start, end; spatial_index.h:36-44), the
    primary-chain filter and MAPQ see them in that order."""
    chunks = []
    for n in (12, 20):
        parts = [_diag(4000, 40, n, rs=rs) for rs in (0, 1, 2, 3, 4, 5)]
        parts.append(_diag(9000, 40, n, rs=2))       # same score and count inside one bucket too: start / end decide
        chunks.append(_cat(*parts))
    want = _run_and_compare(cdev, chunks, 3)
    # (chains of the two strands of one reference overlap on the target: GeneratePrimaryChains keeps one per locus)
    assert all(len(w) == 4 for w in want)
    assert len({bits(np.float32(ch["score"])).item() for ch in want[0]}) == 1


def test_tangled_random_anchor_sets(cdev):
    """200 random chunks: several noisy diagonals per bucket (jitter on both axes, dropped and doubled anchors, query
    positions shared between loci), random strays, one to three references, sizes from 0 to ~6000 anchors."""
    rng = np.random.default_rng(17)
    chunks = []
    for c in range(200):
        n_refs = 3
        parts = []
        for _ in range(int(rng.integers(0, 7))):
            rs = int(rng.integers(0, 2 * n_refs))
            n = int(rng.integers(2, 120))
            t0, q0 = int(rng.integers(0, 60000)), int(rng.integers(0, 2000))
            step = int(rng.integers(1, 30))
            ratio = float(rng.choice([1.0, 1.0, 0.8, 1.3, 3.0, 0.7, 5.2]))
            k = np.arange(n)
            t = t0 + step * k + rng.integers(-3, 4, size=n)
            q = q0 + (step * k * ratio).astype(np.int64) + rng.integers(-3, 4, size=n)
            keep = rng.random(n) > 0.15
            t, q = np.clip(t[keep], 0, None), np.clip(q[keep], 0, None)
            d = rng.uniform(0, 0.0799, size=len(t)).astype(np.float32)
            parts.append([np.full(len(t), rs, np.uint32), t.astype(np.uint32), q.astype(np.uint32), d])
            if rng.random() < 0.3:      # doubled anchors
                parts.append([x[: len(t) // 3].copy() for x in parts[-1]])
        m = int(rng.integers(0, 60)) if c % 10 else int(rng.integers(3000, 6000))     # strays (every tenth chunk: thousands)
        parts.append([rng.integers(0, 2 * n_refs, size=m).astype(np.uint32), rng.integers(0, 70000, size=m).astype(np.uint32),
                      rng.integers(0, 2200, size=m).astype(np.uint32), rng.uniform(0, 0.0799, size=m).astype(np.float32)])
        chunks.append(_cat(*parts))
    want = _run_and_compare(cdev, chunks, 3, shuffle_seed=9)
    assert sum(1 for w in want if len(w) >= 2) > 50 and sum(1 for w in want if len(w) == 0) >= 1


def test_routes_agree_under_other_skip_band_and_gap_limits(cdev):
    """The oracle has the reference's constants (25 skips, band 5000, target gap 5000).  With other limits the
    classifier's window rule changes (dense pairs and its own chaining only inside windows of min(skips + 1, band)
    predecessors), so its three routes -- dense pairs + chaining, dense pairs only, lockstep scans -- and the DP kernels
    behind them must still produce the same bytes."""
    rng = np.random.default_rng(23)
    chunks = []
    for c in range(60):
        parts = []
        for _ in range(int(rng.integers(1, 6))):
            rs = int(rng.integers(0, 4))
            n = int(rng.integers(2, 150))
            t0, q0 = int(rng.integers(0, 30000)), int(rng.integers(0, 1500))
            step = int(rng.integers(1, 12))
            k = np.arange(n)
            t = t0 + step * k + rng.integers(-2, 3, size=n)
            q = q0 + step * k + rng.integers(-2, 3, size=n)
            d = rng.uniform(0, 0.0799, size=n).astype(np.float32)
            parts.append([np.full(n, rs, np.uint32), np.clip(t, 0, None).astype(np.uint32), np.clip(q, 0, None).astype(np.uint32), d])
        m = int(rng.integers(0, 400))
        parts.append([rng.integers(0, 4, size=m).astype(np.uint32), rng.integers(0, 32000, size=m).astype(np.uint32),
                      rng.integers(0, 1700, size=m).astype(np.uint32), rng.uniform(0, 0.0799, size=m).astype(np.float32)])
        chunks.append(_cat(*parts))
    # ... and an anchor that chains to the trunk only past k failing predecessors, k around the 31 at which the
    # classifier's window count saturates (with a skip limit of 40 every one of these chains goes through)
    # (the anchor is made the FIRST of a 64-anchor tile -- isolated fillers in an earlier bucket -- so that its tile holds
    # few pairs and takes the dense route: a tile with all those predecessors in it exceeds the route's pair budget)
    for k in (29, 30, 31, 32, 35):
        fill = 64 - 11 - k
        fillers = [np.zeros(fill, np.uint32), (7000 * np.arange(fill)).astype(np.uint32), np.full(fill, 5, np.uint32),
                   np.full(fill, D0, np.float32)]
        trunk = _diag(9890, 9890, 11, rs=1)
        bad = [np.ones(k, np.uint32), np.full(k, 9995, np.uint32), (10000 - 3000 - np.arange(k)).astype(np.uint32),
               np.full(k, D0, np.float32)]
        chunks.append(_cat(fillers, trunk, bad, _one(1, 10000, 10000)))
    n_long = 5
    flat, offs = [], [0]
    for c in chunks:
        a = np.zeros(len(c[0]), dtype=_abi.CARRIED_DTYPE)
        a["ref_strand"], a["target_position"], a["query_position"], a["distance"] = c
        flat.append(a)
        offs.append(offs[-1] + len(a))
    flat = np.concatenate(flat)
    saved = (cdev.params.max_num_skips, cdev.params.chaining_band_length, cdev.params.max_target_gap_length)
    try:
        # ((40, 5000, 5000): windows of 31 and more predecessors with a skip limit above the dense route's 30)
        for skips, band, gap in ((3, 5000, 5000), (0, 5000, 5000), (25, 4, 5000), (40, 5000, 300), (7, 9, 60), (40, 5000, 5000),
                                 (31, 5000, 5000), (29, 5000, 5000)):
            cdev.params.max_num_skips, cdev.params.chaining_band_length, cdev.params.max_target_gap_length = skips, band, gap
            got = []
            for knobs in ({}, {"classify_resolve": 0}, {"classify_dense": 0}):
                for k, v in knobs.items():
                    cdev.set_knob(k, v)
                try:
                    res, anchors, aoff = cdev.chain_anchors(flat, offs, 2, with_anchors=True)
                    allc, coff = cdev.download_chains()
                finally:
                    for k in knobs:
                        cdev.set_knob(k, -1)
                got.append((res.tobytes(), anchors.tobytes(), aoff.tobytes(), allc.tobytes(), coff.tobytes()))
            assert got[0] == got[1] == got[2], (skips, band, gap)
            assert (res["status"] == 0).all() and res["n_chains"].sum() > 20, (skips, band, gap)
            if (skips, band, gap) == (40, 5000, 5000):   # (12 anchors: the trunk's 11 + the anchor behind the failing ones)
                assert [int(r["chains"][0]["n_anchors"]) for r in res[-n_long:]] == [12] * n_long
    finally:
        cdev.params.max_num_skips, cdev.params.chaining_band_length, cdev.params.max_target_gap_length = saved


def test_generate_chains_from_features_equals_the_sample_route(cdev, small_world, oracle_tree):
    """smhip_generate_chains (features in: SpatialIndex::GenerateChains' own inputs) gives the records the whole path
    gives from the samples of the same chunks, and the oracle's so_generate_chains -- also with a query start offset
    and with chains carried in from a previous chunk (spatial_index.cc:303-322), and for feature signals shorter than
    StreamingMap's 50-event minimum, which GenerateChains itself does not know."""
    pa, ix = small_world["pa"][:16], small_world["ix"]
    feats = [po.generate_events(x)[0] for x in pa]
    feats += [feats[0][:40], feats[1][:7], feats[2][:6], feats[3][:5], np.zeros(0, np.float32)]   # 40, 7, 6 (one window: no seed), 5, 0
    offs = np.concatenate([[0], np.cumsum([len(f) for f in feats])]).astype(np.uint32)
    res, anchors, aoff = cdev.generate_chains(np.concatenate(feats), offs, 3, with_anchors=True)
    samples = np.concatenate(pa)
    soff = np.concatenate([[0], np.cumsum([len(x) for x in pa])]).astype(np.uint32)
    full = cdev.map_chunks(samples, soff, n_refs=3)
    for k in full.dtype.names:
        if k not in ("n_events", "nodes_visited", "points_tested"):
            assert np.array_equal(full[k], res[:16][k]), k
    for i, f in enumerate(feats):
        if len(f) < 6:
            assert res[i]["n_queries"] == 0 and res[i]["n_chains"] == 0
            continue
        chains, st = po.generate_chains(oracle_tree, ix.positions, f, 3)
        assert res[i]["n_queries"] == st.queries and res[i]["n_anchors"] == st.anchors and res[i]["n_chains"] == len(chains), i
        for k, ch in enumerate(chains[:4]):
            g = res[i]["chains"][k]
            assert bits(np.float32(g["score"])) == bits(np.float32(ch["score"])) and g["start"] == ch["start"] and g["end"] == ch["end"]
    assert res[16]["n_queries"] == 17 and res[17]["n_queries"] == 0 and res[18]["n_queries"] == 0
    # second chunk of the same reads: features of the next 4000 samples do not exist here, so chain the SAME features
    # again with the first round's chains carried in and a query start offset -- against the oracle's map of that call
    carried, coff = anchors[: int(aoff[16])], aoff[:17].astype(np.uint32)
    qs = np.array([len(f) for f in feats[:16]], np.uint32)
    res2 = cdev.generate_chains(np.concatenate(feats[:16]), offs[:17], 3, query_start_offsets=qs, carried=carried,
                                carried_offsets=coff)
    cdev.upload_batch(samples, soff, query_start_offsets=qs, carried=carried, carried_offsets=coff)
    cdev.run(3); cdev.sync()
    full2 = cdev.download()
    for k in full2.dtype.names:
        if k not in ("n_events", "nodes_visited", "points_tested"):
            assert np.array_equal(full2[k], res2[k]), k
    assert (res2["n_anchors"] > res[:16]["n_anchors"]).any()


def test_class_methods_forward_to_the_device(small_world, oracle_tree):
    """The reference's own per-chunk class surface on the mirror classes (sigmap_amd/host/device_methods.cc):
    Sigmap::GenerateEvents (sigmap.h:76; clears the features, APPENDS to the deviations) and
    SpatialIndex::GenerateChains (spatial_index.h:87-92; chains in = the previous chunk's, out = the new ones with
    their anchors), called one chunk at a time the way the reference's StreamingMap loop calls them
    (sigmap.cc:647-666) -- against the oracle's generate_events / generate_chains, two chunks deep."""
    import ctypes as C
    H = _abi.host_lib()
    ix = pipeline.HostIndex.build(small_world["genome"], small_world["lm"], threads=4)   # an index of its own: it gets a device context
    assert H.smhost_index_bind_device(ix._h, None) == 0
    positions = small_world["ix"].positions
    for r in range(6):
        x = small_world["pa"][r]
        # GenerateEvents on [start, end) of the read
        for (a, b) in ((0, 4000), (500, 2500)):
            feat = np.zeros(4096, np.float32); stdv = np.zeros(4096, np.float32)
            nf = H.smhost_generate_events(None, _abi.ptr(x), len(x), a, b, _abi.ptr(feat), _abi.ptr(stdv), 4096, 3)
            f, s = po.generate_events(x[a:b])
            assert nf == len(f) and np.array_equal(bits(feat[:nf]), bits(f))
            assert (stdv[:3] == -1).all() and np.array_equal(bits(stdv[3:3 + nf]), bits(s))    # appended, not cleared
        f, _ = po.generate_events(x)
        # first chunk: no previous chains
        want, _ = po.generate_chains(oracle_tree, positions, f, 3)
        oc = np.zeros(64, dtype=_abi.CHAIN_DTYPE); oa = np.zeros(1 << 16, dtype=_abi.CARRIED_DTYPE); na = C.c_uint64()
        n = H.smhost_generate_chains(ix._h, _abi.ptr(f), len(f), 0, 2, 0.08, 3, None, 0, None, _abi.ptr(oc), 64, _abi.ptr(oa),
                                     len(oa), C.byref(na))
        assert n == len(want)
        at = 0
        for k, ch in enumerate(want):
            g = oc[k]
            assert bits(np.float32(g["score"])) == bits(np.float32(ch["score"]))
            assert (g["ref"], g["start"], g["end"], g["n_anchors"], g["direction"]) == \
                   (ch["ref"], ch["start"], ch["end"], ch["n_anchors"], ch["direction"])
            assert g["mapq"] == (ch["mapq"] if k == 0 else 0)
            ga = oa[at:at + ch["n_anchors"]]
            assert np.array_equal(ga["target_position"], ch["anchors"]["target_position"])
            assert np.array_equal(ga["query_position"], ch["anchors"]["query_position"])
            assert np.array_equal(bits(ga["distance"]), bits(ch["anchors"]["distance"]))
            at += ch["n_anchors"]
        assert at == na.value
        # second chunk of the read loop: the same features again at a query start offset, the chains carried in --
        # against the device's own batched entry point fed with the same inputs (itself held to the oracle above)
        oc2 = np.zeros(64, dtype=_abi.CHAIN_DTYPE); oa2 = np.zeros(1 << 16, dtype=_abi.CARRIED_DTYPE); na2 = C.c_uint64()
        n2 = H.smhost_generate_chains(ix._h, _abi.ptr(f), len(f), len(f), 2, 0.08, 3, _abi.ptr(oc), n, _abi.ptr(oa), _abi.ptr(oc2), 64,
                                      _abi.ptr(oa2), len(oa2), C.byref(na2))
        dev = pipeline.DeviceMapper(0)
        dev.upload_index(small_world["ix"])
        res = dev.generate_chains(f, [0, len(f)], 3, query_start_offsets=[len(f)], carried=oa[:at], carried_offsets=[0, at])
        allc, coff = dev.download_chains()
        assert n2 == res[0]["n_chains"] == coff[1] and n2 >= n
        for k in range(n2):
            for fld in ("ref", "start", "end", "n_anchors", "direction", "mapq"):
                assert oc2[k][fld] == allc[k][fld], (k, fld)
            assert bits(np.float32(oc2[k]["score"])) == bits(np.float32(allc[k]["score"]))
        assert na2.value == sum(int(c["n_anchors"]) for c in allc)
        dev.close()
    ix.close()


@pytest.mark.parametrize("sort_lines", [-1, 0])
def test_anchor_outside_the_batchs_groups_flags_its_chunk_only(cdev, sort_lines):
    """A caller-supplied anchor whose ref_strand names a (reference, strand) group the batch does not have
    (>= 2^ceil(log2(2 n_refs))) has no place in the sort's partition: on the whole-line route its part would lie
    beyond the chunk's slices of the record scratch.  The chunk is flagged SMHIP_ST_SORT_CAP and yields no anchors
    and no chains; its neighbours -- before and after it in the record scratch -- are mapped as if it were not
    there.  Both partition routes."""
    n_refs = 3          # groups 0..5, group field of 3 bits: 8 .. are invalid
    good = _cat(_diag(1000, 10, 40), _diag(90000, 5, 30, rs=3))
    bad = _cat(_diag(5000, 10, 700), _one(9, 7000, 50), _one(0xFFFFFFFF, 1, 2), _diag(400000, 20, 600, rs=5))
    chunks = [good, bad, _cat(_diag(300, 7, 25, rs=2)), good]
    flat, offs = [], [0]
    for c in chunks:
        a = np.zeros(len(c[0]), dtype=_abi.CARRIED_DTYPE)
        a["ref_strand"], a["target_position"], a["query_position"], a["distance"] = c
        flat.append(a)
        offs.append(offs[-1] + len(a))
    cdev.set_knob("sort_lines", sort_lines)
    try:
        res, anchors, aoff = cdev.chain_anchors(np.concatenate(flat), offs, n_refs, with_anchors=True)
    finally:
        cdev.set_knob("sort_lines", -1)
    assert res[1]["status"] & 16 and res[1]["n_anchors"] == 0 and res[1]["n_chains"] == 0
    assert int(aoff[2] - aoff[1]) == 0
    for i in (0, 2, 3):
        c = chunks[i]
        want = po.chain_anchors(c[0], c[1], c[2], c[3], n_refs)
        assert res[i]["status"] == 0 and res[i]["n_anchors"] == len(c[0]) and res[i]["n_chains"] == len(want) > 0
        for k, ch in enumerate(want[:4]):
            g = res[i]["chains"][k]
            assert bits(np.float32(g["score"])) == bits(np.float32(ch["score"])) and g["n_anchors"] == ch["n_anchors"]
    assert res[0]["chains"].tobytes() == res[3]["chains"].tobytes()

#!/usr/bin/env python3
"""Randomised parity run (development aid, beyond the fixed seeds of tests/): for a range of seeds,
genomes of random shapes (incl. repeats), noisy / truncated reads, and random device parameters,
the GPU path's per-chunk records + refinement must equal the oracle's.  Usage:
  python tools/fuzz_parity.py [n_rounds] [first_seed]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402

os.environ.setdefault("SMHIP_DEV", "1")   # the development knobs (smhip_dev_knob) drive the rarely taken routes

from oracle import pyoracle as po  # noqa: E402
from sigmap_amd import _abi, pipeline, synth  # noqa: E402
from test_gpu_parity import _paf_lines  # noqa: E402


def main(rounds, first):
    bad = 0
    for seed in range(first, first + rounds):
        rng = np.random.default_rng(seed)
        n_contigs = int(rng.integers(1, 5))
        lens = [int(rng.integers(20000, 120000)) for _ in range(n_contigs)]
        genome = synth.make_genome(seed, lens)
        if rng.random() < 0.5:  # a repeat: the second half of contig 0 copies its first half
            g0 = genome[0]
            half = len(g0) // 2
            genome[0] = g0[:half] + g0[:len(g0) - half]
        lm = synth.make_pore_model()
        n_reads = int(rng.integers(8, 40))
        raw, _ = synth.simulate_reads(genome, lm, n_reads, seed + 1000)
        pa = [po.add_signal(r, synth.DIGITISATION, synth.RANGE, synth.OFFSET) for r in raw]
        kind = ["read"] * len(pa)
        for i in range(len(pa)):  # pure-noise reads, flat reads (few events), reads with a stuck stretch
            u = rng.random()
            if u < 0.10:
                pa[i] = rng.normal(90, 12, size=len(pa[i])).astype(np.float32)
                kind[i] = "noise"
            elif u < 0.15:
                pa[i] = np.full(len(pa[i]), 85.0, np.float32) + rng.normal(0, 0.01, size=len(pa[i])).astype(np.float32)
                kind[i] = "flat"
            elif u < 0.25:
                a = int(rng.integers(0, len(pa[i]) - 600))
                pa[i] = pa[i].copy()
                pa[i][a:a + 500] = pa[i][a]
                kind[i] = "stuck"
        ix = pipeline.HostIndex.build(genome, lm, threads=4)
        tree = po.OracleTree(ix.values.copy())
        ref_lens = np.array([len(g) for g in genome], dtype=np.uint32)
        dev = pipeline.DeviceMapper(0)
        dev.params.walk_blocks_per_cu = int(rng.choice([0, 1, 2, 8]))
        dev.params.reference_walk = int(rng.integers(0, 2))
        dev.params.hits_per_query_fast = int(rng.choice([512, 64, 4, 0]))      # (0: sized from the index census)
        dev.params.search_mode = int(rng.choice([_abi.SEARCH_AUTO, _abi.SEARCH_AUTO, _abi.SEARCH_EXACT, _abi.SEARCH_SET]))
        dev.set_knob("sort_lines", int(rng.choice([-1, -1, 0, 4])))   # the sort's whole-line partition route, or the exact one
        dev.set_knob("classify_dense", int(rng.choice([-1, -1, 0])))   # dense pairs or lockstep scans in the chain classifier
        dev.set_knob("classify_resolve", int(rng.choice([-1, -1, 0])))  # ... chaining what it can itself, or nothing
        dev.set_knob("second_search", int(rng.choice([0, 0, 1])))  # over-cap queries: ranked, or searched a second time
        # the set walk: lanes of a wave share a query's walk (default), do not, or few queries per wave and many helpers
        share = int(rng.choice([-1, -1, 0, 3]))
        dev.set_knob("walk_share", -1 if share == 3 else share)
        dev.set_knob("walk_grab", int(rng.choice([1, 2, 4, 8])) if share == 3 else 0)
        dev.set_knob("walk_grab_upto", 2040 if share == 3 else 0)
        # the hit cap: lowered caps put (nearly) every query beyond it -- the set walk's second search in the
        # reference's order, or, once the context has seen such a batch, the exact walk for everything
        nnp = int(rng.choice([5000, 5000, 40, 7]))
        dev.params.num_nearest_points = nnp
        radius = float(rng.choice([0.08, 0.08, 0.15]))
        dev.params.search_radius = radius
        dev.upload_index(ix)
        samples = np.concatenate(pa)
        offsets = np.concatenate([[0], np.cumsum([len(x) for x in pa])]).astype(np.uint32)
        res = dev.map_chunks(samples, offsets, n_refs=n_contigs)
        # the same batch again: the speculative route of smhip_map_run (no read-backs), then with one of its
        # capacities halved (guard kernel fires, batch repeated) -- identical records whatever the route
        same = [k for k in res.dtype.names if k not in ("nodes_visited", "points_tested")]
        mism = 0
        for tight in (None, str(int(rng.integers(1, 3)))):
            dev.set_knob("spec_tight", int(tight) if tight else 0)
            try:
                again = dev.map_chunks(samples, offsets, n_refs=n_contigs)
            finally:
                dev.set_knob("spec_tight", 0)
            if not all(np.array_equal(res[k], again[k]) for k in same):
                mism += 1
                print("   ROUTE MISMATCH (spec_tight=%s)" % tight)
        names = ["contig%d" % i for i in range(n_contigs)]
        lines = _paf_lines(res, pa, names, ref_lens)
        for i, x in enumerate(pa):
            if nnp != 5000:  # the oracle's map_read has the reference's constant: compare chunk-level chains
                f, _ = po.generate_events(x)
                if len(f) <= 50:
                    continue
                chains, st = po.generate_chains(tree, ix.positions, f, n_contigs, radius=radius, num_nearest_points=nnp)
                ok = res[i]["n_anchors"] == st.anchors and res[i]["hits_total"] == st.hits and res[i]["n_chains"] == len(chains)
                for k, c in enumerate(chains[:4]):
                    g = res[i]["chains"][k]
                    ok = ok and g["score"] == np.float32(c["score"]) and g["start"] == c["start"] and g["end"] == c["end"] \
                        and g["n_anchors"] == c["n_anchors"]
                if not ok:
                    mism += 1
                    print("   MISMATCH read %d (%s) with cap %d" % (i, kind[i], nnp))
                continue
            m = po.map_read(tree, ix.positions, x, n_contigs, ref_lens, params={"search_radius": radius})
            ok = lines[i] == po.format_paf(m, "read%d" % i, names, ref_lens)
            ok = ok and res[i]["n_anchors"] == m.stats.anchors and res[i]["hits_total"] == m.stats.hits
            if dev.params.reference_walk:
                ok = ok and res[i]["nodes_visited"] == m.stats.search.nodes and res[i]["points_tested"] == m.stats.search.points
            if not ok:
                mism += 1
                print("   MISMATCH read %d (%s): status %d\n     gpu: %s\n     ora: %s" % (
                    i, kind[i], res[i]["status"], lines[i][:200], po.format_paf(m, "read%d" % i, names, ref_lens)[:200]))
        # refinement (register sweep / ring sweep, whichever takes the problem) against the oracle's
        # banded sDTW on the same window and band
        band, pad = int(rng.choice([16, 32, 64, 128])), int(rng.choice([0, 32, 100]))
        sig, offs = pipeline.reference_signals(genome, lm)
        dev.upload_reference(sig, offs)
        dev.refine(band=band, pad=pad); dev.sync()
        dist, end = dev.refine_download()
        feat, _, nf, _, _ = dev.events(samples, offsets)
        rmis = 0
        for i in range(len(pa)):
            r = res[i]
            if r["n_chains"] == 0:
                rmis += end[i] != -1
                continue
            c0 = r["chains"][0]
            strand = 0 if c0["direction"] == 1 else 1
            so, se = int(offs[2 * c0["ref"] + strand]), int(offs[2 * c0["ref"] + strand + 1])
            start, endp = int(c0["start"]), int(c0["end"])
            t0 = max(0, start - pad); t1 = min(se - so, endp + 6 + pad)
            q = feat[i, :nf[i]]
            qa, qb = int(r["c0_last_qpos"]), int(r["c0_first_qpos"])
            den = max(1, endp - start)
            ta = np.arange(t0, t1, dtype=np.int64)
            centre = 1 + qa + ((ta - start) * (qb - qa)) // den
            lo = np.clip(centre - band // 2, 1, None).astype(np.int64)
            hi = np.maximum(np.clip(centre + band // 2, None, len(q)).astype(np.int64), 0)
            d, e = po.sdtw_banded(sig[so + t0:so + t1], q, lo.astype(np.uint32), hi.astype(np.uint32))
            okr = (end[i] == -1) if e < 0 else (np.float32(dist[i]).tobytes() == np.float32(d).tobytes() and end[i] == t0 + e)
            if not okr:
                rmis += 1
                print("   REFINE MISMATCH read %d (%s): band %d pad %d gpu (%r, %d) oracle (%r, %d)" % (
                    i, kind[i], band, pad, dist[i], end[i], d, t0 + e if e >= 0 else -1))
        mism += rmis
        # other chaining limits than the reference's constants (no oracle for those): the chain classifier's routes --
        # dense pairs with its own chaining, dense pairs only, lockstep scans -- and the DP kernels behind them must agree
        saved = (dev.params.max_num_skips, dev.params.chaining_band_length, dev.params.max_target_gap_length, dev.params.max_gap_length)
        dev.params.max_num_skips = int(rng.choice([0, 2, 7, 25, 31, 40]))
        dev.params.chaining_band_length = int(rng.choice([3, 20, 5000]))
        dev.params.max_target_gap_length = int(rng.choice([200, 1000, 5000, 20000]))
        dev.params.max_gap_length = int(rng.choice([50, 2000]))
        got = []
        for knobs in ({"classify_dense": -1, "classify_resolve": -1}, {"classify_resolve": 0}, {"classify_dense": 0}):
            for k, v in knobs.items():
                dev.set_knob(k, v)
            try:
                r2 = dev.map_chunks(samples, offsets, n_refs=n_contigs)
            finally:
                dev.set_knob("classify_dense", -1)
                dev.set_knob("classify_resolve", -1)
            got.append(tuple(r2[k].tobytes() for k in same))
        if not (got[0] == got[1] == got[2]):
            mism += 1
            print("   CLASSIFIER ROUTES DISAGREE (skips %d band %d target gap %d gap %d)" % (
                dev.params.max_num_skips, dev.params.chaining_band_length, dev.params.max_target_gap_length, dev.params.max_gap_length))
        dev.params.max_num_skips, dev.params.chaining_band_length, dev.params.max_target_gap_length, dev.params.max_gap_length = saved
        print("seed %d: %d contigs %s, %d reads, params (G=%d strict=%d fast=%d cap=%d r=%.2f search=%d walked=%d): %d mismatches, %d mapped"
              % (seed, n_contigs, lens, n_reads, dev.params.walk_blocks_per_cu, dev.params.reference_walk,
                 dev.params.hits_per_query_fast, nnp, radius, dev.params.search_mode, int(dev.debug_counters()[6]), mism,
                 int((res["n_chains"] > 0).sum())), flush=True)
        bad += mism
        dev.close(); tree.close(); ix.close()
    print("TOTAL mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(sys.argv[2]) if len(sys.argv) > 2 else 1))

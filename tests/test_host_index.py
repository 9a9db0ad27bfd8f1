"""Host-side product code (sigmap_amd/host via include/sigmap_host.h) against the oracle and
the reference goldens: index construction, `.pt`/`.si` files, pA conversion, PAF logic."""
import ctypes as C
import os

import numpy as np

from conftest import GOLDEN, bits
from oracle import pyoracle as po
from sigmap_amd import _abi, pipeline, synth


def _flat_from_golden(g):
    leaf = g["is_leaf"] == 1
    return leaf, g["a"], g["b"], g["lo"], g["hi"]


def test_host_tree_equals_reference_golden():
    g = np.load(os.path.join(GOLDEN, "kdtree.npz"))
    ix = pipeline.HostIndex.from_points(g["positions"], g["values"], threads=3)
    leaf, a, b, lo, hi = _flat_from_golden(g)
    n = ix.nodes
    assert len(n) == len(leaf)
    assert np.array_equal(n["feat"] < 0, leaf)
    assert np.array_equal(n["a"][leaf], a[leaf]) and np.array_equal(n["b"][leaf], b[leaf])
    assert np.array_equal(n["feat"][~leaf], a[~leaf].astype(np.int32))
    assert np.array_equal(n["a"][~leaf], lo[~leaf].view(np.uint32))
    assert np.array_equal(n["b"][~leaf], hi[~leaf].view(np.uint32))
    assert np.array_equal(ix.vind, g["vind"])
    assert np.array_equal(bits(ix.bbox), bits(g["bbox"]))
    # child2 = first pre-order slot after child1's subtree
    size = np.ones(len(n), np.int64)
    for i in range(len(n) - 1, -1, -1):
        if n["feat"][i] >= 0:
            c2 = n["child2"][i]
            assert c2 == i + 1 + size[i + 1]
            size[i] = 1 + size[i + 1] + size[c2]
    assert size[0] == len(n)


def test_parallel_build_is_deterministic(small_world):
    ix = small_world["ix"]
    one = pipeline.HostIndex.from_points(ix.positions, ix.values, threads=1)
    many = pipeline.HostIndex.from_points(ix.positions, ix.values, threads=8)
    assert np.array_equal(one.nodes, many.nodes) and np.array_equal(one.vind, many.vind)
    assert np.array_equal(one.nodes, ix.nodes)


def test_index_files_roundtrip_and_cross_read(small_world, tmp_path):
    """PREFIX.pt/.si written by the product are read back identically by the product AND by the
    oracle's reader (reference format), and vice versa."""
    ix = small_world["ix"]
    prefix = str(tmp_path / "idx")
    ix.save(prefix)
    assert os.path.getsize(prefix + ".pt") == 16 + 16 * ix.n_points                     # T8
    assert os.path.getsize(prefix + ".si") == 84 + 8 * ix.m_size + 32 * ix.n_nodes     # T9
    back = pipeline.HostIndex.load(prefix)
    assert np.array_equal(back.nodes, ix.nodes) and np.array_equal(back.vind, ix.vind)
    assert np.array_equal(back.positions, ix.positions) and np.array_equal(bits(back.values), bits(ix.values))
    ot = po.OracleTree.load(prefix + ".si", ix.values.copy())
    pa = small_world["pa"][0]
    f, _ = po.generate_events(pa)
    full = po.OracleTree(ix.values.copy())
    for p in range(2, len(f) - 5, 40):
        a = ot.radius_search(f[p:p + 6], 0.08)
        b = full.radius_search(f[p:p + 6], 0.08)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and a[3] == b[3]
    # oracle-written files load in the product
    p2 = str(tmp_path / "o")
    full.save(p2 + ".si")
    assert po.lib().so_points_save((p2 + ".pt").encode(), 6, 20, po._p(ix.positions.copy()),
                                   po._p(ix.values.copy()), ix.n_points) == 0
    again = pipeline.HostIndex.load(p2)
    assert np.array_equal(again.nodes, ix.nodes)
    ot.close(); full.close()


def test_index_pipeline_matches_oracle(small_world):
    """sigmap.cc:999-1046 end to end: masks, expected signal, z-score, dedupe, strand order."""
    genome, lm, ix = small_world["genome"], small_world["lm"], small_world["ix"]
    L = po.lib()
    k, dim = 6, 6
    n = len(genome)
    neg = []
    for g in genome:
        out = C.create_string_buffer(len(g) + 1)
        L.so_reverse_complement(g, len(g), out)
        neg.append(out.raw[:len(g)])
    lens = np.array([len(g) for g in genome], dtype=np.uint32)
    pm = [np.zeros(len(g) - (dim + k - 1) + 1, np.uint8) for g in genome]
    nm = [np.zeros(len(g) - (dim + k - 1) + 1, np.uint8) for g in genome]
    seqs = (C.c_char_p * n)(*genome); nseqs = (C.c_char_p * n)(*neg)
    pmp = (C.c_void_p * n)(*[a.ctypes.data for a in pm]); nmp = (C.c_void_p * n)(*[a.ctypes.data for a in nm])
    L.so_masked_positions(dim + k - 1, 0.0002, n, seqs, nseqs, po._p(lens), pmp, nmp)
    cap = 2 * int(lens.sum())
    positions = np.zeros(cap, np.uint64); values = np.zeros(cap, np.float32)
    count = 0
    lmc = np.ascontiguousarray(lm)
    sigs = []
    for strand_seqs in (genome, neg):
        row = []
        for s in strand_seqs:
            lv = np.zeros(len(s), np.float32)
            m = L.so_level_means(po._p(lmc), k, s, 0, len(s), po._p(lv))
            z = np.zeros(m, np.float32)
            L.so_zscore(po._p(lv), m, po._p(z))
            row.append(z)
        sigs.append(row)
    for i in range(n):
        count = L.so_point_cloud_one_direction(1, i, po._p(pm[i]), po._p(sigs[0][i]), len(sigs[0][i]), dim,
                                               po._p(positions), po._p(values), count)
    for i in range(n):
        count = L.so_point_cloud_one_direction(0, i, po._p(nm[i]), po._p(sigs[1][i]), len(sigs[1][i]), dim,
                                               po._p(positions), po._p(values), count)
    assert count == ix.n_points
    assert np.array_equal(positions[:count], ix.positions)
    assert np.array_equal(bits(values[:count]), bits(ix.values))
    # the reference z-scored strand signal exported for the sDTW target is the same array
    out = np.zeros(len(genome[1]), np.float32)
    m = _abi.host_lib().smhost_reference_signal(_abi.ptr(lmc), k, genome[1], len(genome[1]), 1, _abi.ptr(out))
    assert np.array_equal(bits(out[:m]), bits(sigs[1][1]))


def test_masking_with_repeats_and_ambiguous_bases():
    """High-frequency 11-mers (> 2e-4) and windows ending at an N are masked (sigmap.cc:101-130)."""
    rng = np.random.default_rng(5)
    base = bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 4000)])
    rep = b"ACGTTGCAAGGCTTAACG"
    seq = base[:1000] + rep * 40 + base[1000:2000] + b"NNNN" + base[2000:]
    lm = synth.make_pore_model()
    ix = pipeline.HostIndex.build([seq], lm, threads=1)
    L = po.lib()
    neg = C.create_string_buffer(len(seq) + 1)
    L.so_reverse_complement(seq, len(seq), neg)
    m = len(seq) - 11 + 1
    pmask = np.zeros(m, np.uint8); nmask = np.zeros(m, np.uint8)
    lens = np.array([len(seq)], np.uint32)
    L.so_masked_positions(11, 0.0002, 1, (C.c_char_p * 1)(seq), (C.c_char_p * 1)(neg.raw[:len(seq)]), po._p(lens),
                          (C.c_void_p * 1)(pmask.ctypes.data), (C.c_void_p * 1)(nmask.ctypes.data))
    assert pmask.sum() > 300 and nmask.sum() > 300
    strand_pos = ix.positions[(ix.positions & 1) == 0]
    kept = ((strand_pos >> 1) & 0xFFFFFFFF).astype(np.int64)
    assert not pmask[kept].any()  # no masked position made it into the + strand cloud


def test_raw_to_pa_filter():
    raw = np.array([0, 100, 500, 1000, 2000, -5], dtype=np.int16)
    out = pipeline.raw_to_pa(raw, 8192.0, 1467.61, 10.0)
    scale = np.float32(1467.61) / np.float32(8192.0)
    exp = [(np.float32(v) + np.float32(10.0)) * scale for v in raw]
    exp = np.array([v for v in exp if 30 < v < 200], dtype=np.float32)
    assert np.array_equal(bits(out), bits(exp))


def test_host_addsignal_matches_oracle_addsignal():
    """A1 on the host (SignalBatch::AddSignal mirror behind smhost_raw_to_pa) against the oracle's
    so_add_signal (signal_batch.cc:139-151): ragged lengths, both cuts hit, values at the strict
    boundaries, per-read calibration constants incl. 6-significant-digit attribute values."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(17)
    cal = [(8192.0, 1467.61, 10.0), (8192.0, 1400.5, -3.0), (2048.0, 748.58, 6.0), (4096.0, 1211.3, 21.0),
           (8192.0, 1402.88, 6.0), (1.0, 1.0, 0.0)]
    for dg, rg, of in cal:
        for n in (0, 1, 7, 4000, 4097, 30011):
            raw = rng.integers(-300, 1500, size=n).astype(np.int16)
            if n >= 7:
                raw[:4] = [32767, -32768, 0, 1]
            if (dg, rg, of) == (1.0, 1.0, 0.0) and n >= 7:
                raw[:7] = [29, 30, 31, 199, 200, 201, 100]      # scale 1: the strict bounds themselves
            want = po.add_signal(raw, dg, rg, of)
            got = pipeline.raw_to_pa(raw, dg, rg, of)
            assert np.array_equal(bits(got), bits(want)), (dg, rg, of, n)
            if (dg, rg, of) == (1.0, 1.0, 0.0) and n >= 7:
                assert list(want[:3]) == [31.0, 199.0, 100.0]


def test_chunk_bounds_cut_the_filtered_array_like_the_reference():
    """The CLI's load phase (SignalBatch raw mode, smhost_chunk_bounds): the filtered length is the oracle's
    AddSignal length, and chunk c of the FILTERED array (sigmap.cc:643-652) is exactly what AddSignal keeps of
    raw[bounds[c] .. bounds[c + 1]) -- the piece the mapper sends to the device for that chunk.  Reads that lose
    samples to both cuts (so that chunk boundaries move in the raw array), chunk sizes that do and do not divide the
    block the counter works in, a max_chunks that ends the table early, reads shorter than a chunk."""
    from oracle import pyoracle as po
    L = _abi.host_lib()
    rng = np.random.default_rng(23)
    for case in range(60):
        n = int(rng.integers(0, 70000)) if case % 5 else int(rng.choice([0, 1, 511, 512, 513, 3999, 4000, 4001]))
        cs = int(rng.choice([4000, 512, 100, 1, 4096]))
        max_chunks = int(rng.choice([0, 1, 3, 30, 1000]))
        dg, rg, of = (8192.0, 1467.61, 10.0) if case % 3 else (2048.0, 748.58, 6.0)
        raw = rng.integers(300, 1000, size=n).astype(np.int16)
        if n:   # runs and single samples outside (30, 200) pA
            for _ in range(int(rng.integers(0, 12))):
                at = int(rng.integers(0, n))
                raw[at:at + int(rng.integers(1, 700))] = 32767 if rng.random() < 0.5 else -3000
        filt = po.add_signal(raw, dg, rg, of)
        bounds = np.zeros(max_chunks + 2, np.uint32)
        flen = C.c_uint64()
        nb = L.smhost_chunk_bounds(_abi.ptr(raw), n, dg, rg, of, cs, max_chunks, _abi.ptr(bounds), len(bounds), C.byref(flen))
        assert flen.value == len(filt), case
        assert nb == min(len(filt) // cs, max_chunks) + 1 and bounds[0] == 0, (case, nb, len(filt), cs, max_chunks)
        for c in range(nb - 1):
            piece = po.add_signal(raw[bounds[c]:bounds[c + 1]], dg, rg, of)
            assert len(piece) == cs and np.array_equal(bits(piece), bits(filt[c * cs:(c + 1) * cs])), (case, c)


def test_chunk_budget_for_the_baseline_reference_sizes():
    """What `sigmap -m` lets a context have in flight (host/mapping_logic.h ChunkBudgetForMemory): the default request
    of 32768 chunks over two contexts stands on a 12 Mb reference and is cut to a few thousand on a 250 Mb one
    (4.98e8 points: ~850 hits per query, 1024 fast slots: 32 MB per chunk reserved) instead of failing in hipMalloc; sixteen
    contexts on one device (eight GPUs' lanes folded onto one) get a couple of hundred each, not thousands; never 0."""
    L = _abi.host_lib()
    free = 250 << 30
    small = L.smhost_chunk_budget(free, 4000, 512, 41.0, 2)
    big = L.smhost_chunk_budget(free, 4000, 1024, 850.0, 2)
    assert small >= 16384                      # 32768 / 2 contexts fits
    assert 2000 <= big <= 4000                 # (the 250 Mb shard test maps 8 000-chunk batches in ~170 GB with ONE context)
    assert big * 2 * 20.5e6 < free * 0.55
    many = L.smhost_chunk_budget(free, 4000, 1024, 1100.0, 16)
    assert 100 <= many <= 260 and 16 * (1.5e9 + many * 40e6) < free * 0.65
    assert L.smhost_chunk_budget(1 << 20, 4000, 2048, 5000.0, 8) == 1
    assert L.smhost_chunk_budget(free, 4000, 1024, 1e9, 2) == L.smhost_chunk_budget(free, 4000, 1024, 5000.0, 2)


def test_contexts_for_reads_one_when_they_fit_one_context():
    """The default number of device contexts per GPU of `sigmap -m` (host/mapping_logic.h ContextsForReads): one when
    the GPU's block of reads fits a context's read slots, as many as the reads fill otherwise, never more than
    --streams allows, never 0."""
    f = _abi.host_lib().smhost_contexts_for_reads
    assert f(10000, 16384, 2) == 1 and f(16384, 16384, 2) == 1 and f(16385, 16384, 2) == 2
    assert f(50000, 16384, 2) == 2 and f(50000, 16384, 4) == 4 and f(40000, 16384, 4) == 3
    assert f(0, 16384, 2) == 1 and f(5, 0, 2) == 2 and f(5, 1, 0) == 1
    assert f(12500, 216, 2) == 2          # a GPU's block of configs[4] with sixteen contexts on one device


def test_paf_logic_matches_oracle(small_world, oracle_tree):
    """smhost_format_paf on records assembled from the ORACLE's chains reproduces the oracle's
    own PAF text: pins the host-side decision/format code independently of the GPU."""
    L = _abi.host_lib()
    mp = _abi.MapParams(); L.smhost_default_map_params(C.byref(mp))
    names = [b"contig0", b"contig1", b"contig2"]
    arr = (C.c_char_p * 3)(*names)
    ix, ref_lens = small_world["ix"], small_world["ref_lens"]
    checked = 0
    for i, pa in enumerate(small_world["pa"][:24]):
        feat, _ = po.generate_events(pa)
        chains, st = po.generate_chains(oracle_tree, ix.positions, feat, 3)
        m = po.map_read(oracle_tree, ix.positions, pa, 3, ref_lens)
        r = np.zeros(1, dtype=_abi.RESULT_DTYPE)
        r["n_features"] = len(feat); r["n_chains"] = len(chains)
        s = np.float32(0)
        for c in chains:
            s = np.float32(s + np.float32(c["score"]))
        r["sum_score"] = s
        for k, c in enumerate(chains[:4]):
            for f in ("score", "ref", "start", "end", "n_anchors", "mapq", "direction"):
                r["chains"][0][k][f] = c[f]
        if chains:
            a = chains[0]["anchors"]
            ad = at = aq = np.float32(0)
            for j in range(len(a)):
                ad = np.float32(ad + a["distance"][j])
                if j + 1 < len(a):
                    at = np.float32(at + np.float32(np.uint32(a["target_position"][j] - a["target_position"][j + 1])))
                    aq = np.float32(aq + np.float32(np.uint32(a["query_position"][j] - a["query_position"][j + 1])))
            r["c0_sum_dist"], r["c0_sum_tgap"], r["c0_sum_qgap"] = ad, at, aq
            r["c0_first_qpos"], r["c0_last_qpos"] = a["query_position"][0], a["query_position"][-1]
        buf = C.create_string_buffer(2048)
        L.smhost_format_paf(_abi.ptr(r), b"read%d" % i, len(pa), 0, len(feat) if len(feat) > 50 else 0, arr,
                            _abi.ptr(ref_lens), 3, C.byref(mp), buf, 2048, None)
        assert buf.value.decode() == po.format_paf(m, "read%d" % i, [n.decode() for n in names], ref_lens)
        assert L.smhost_should_stop(_abi.ptr(r), C.byref(mp)) in (0, 1)
        checked += m.mapped
    assert checked >= 20


def test_oracle_tree_from_host_index_arrays_equals_oracle_build():
    """so_kdtree_from_preorder (the .si content handed to the oracle in memory, used by the full-size GPU
    tests) gives the tree the oracle builds itself: same radius-search results in the same order, same
    node / leaf / point counts."""
    from oracle import pyoracle as po
    from sigmap_amd import pipeline, synth
    genome = synth.make_genome(23, [60000, 30000])
    lm = synth.make_pore_model()
    ix = pipeline.HostIndex.build(genome, lm, threads=3)
    built = po.OracleTree(ix.values.copy())
    handed = po.OracleTree.from_host_index(ix)
    assert built.depth() == handed.depth()
    rng = np.random.default_rng(1)
    for _ in range(200):
        i = int(rng.integers(0, ix.m_size))
        q = ix.values[i:i + 6] + rng.normal(0, 0.05, 6).astype(np.float32)
        a = built.radius_search(q, 0.08)
        b = handed.radius_search(q, 0.08)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
        assert a[3] == b[3]
    built.close(); handed.close()

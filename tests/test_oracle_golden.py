"""The oracle (CPU restatement, oracle/sigmap_oracle.c) against golden vectors captured
from the REAL reference code (tests/golden/make_golden.py via oracle/_ref), and, when the
prebuilt oracle/_ref/libsigmap_ref.so is present, against the reference itself on fresh
seeded inputs.  Bit-exact everywhere (integer/index work and IEEE f32 with contraction off)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN, bits
from oracle import pyoracle as po


@pytest.fixture(scope="module")
def ev_gold():
    return np.load(os.path.join(GOLDEN, "events.npz"))


@pytest.fixture(scope="module")
def kd_gold():
    return np.load(os.path.join(GOLDEN, "kdtree.npz"))


@pytest.mark.parametrize("r", range(6))
def test_events_against_reference_golden(ev_gold, r):
    x = ev_gold[f"x{r}"]
    n = len(x)
    L = po.lib()
    ps = np.zeros(n + 1, np.float32); pss = np.zeros(n + 1, np.float32); t = np.zeros(n + 1, np.float32)
    L.so_prefix_sums(po._p(x), n, po._p(ps), po._p(pss))
    assert np.array_equal(bits(ps), bits(ev_gold[f"ps{r}"]))
    assert np.array_equal(bits(pss), bits(ev_gold[f"pss{r}"]))
    L.so_tstat(po._p(ps), po._p(pss), n, 3, po._p(t))
    assert np.array_equal(bits(t), bits(ev_gold[f"t1_{r}"]))
    L.so_tstat(po._p(ps), po._p(pss), n, 6, po._p(t))
    assert np.array_equal(bits(t), bits(ev_gold[f"t2_{r}"]))
    ev, peaks = po.detect_events(x)
    assert np.array_equal(peaks, ev_gold[f"peaks{r}"])
    assert np.array_equal(ev["start"], ev_gold[f"start{r}"])
    assert np.array_equal(ev["length"], ev_gold[f"length{r}"])
    assert np.array_equal(bits(ev["mean"]), bits(ev_gold[f"mean{r}"]))
    assert np.array_equal(bits(ev["stdv"]), bits(ev_gold[f"stdv{r}"]))


def test_kdtree_build_against_reference_golden(kd_gold, tmp_path):
    values = kd_gold["values"]
    t = po.OracleTree(values)
    # write the reference's .si format and compare byte-for-byte with what nanoflann wrote
    p = str(tmp_path / "o.si")
    t.save(p)
    si = np.fromfile(p, dtype=np.uint8)
    hdr = len(kd_gold["si_header"])
    assert np.array_equal(si[:hdr], kd_gold["si_header"])  # m_size, dim, bbox, leaf_max, vind
    nodes = si[hdr:].reshape(-1, 32).copy()
    nodes[:, 16:].view(np.uint64)[nodes[:, 16:].view(np.uint64) != 0] = 1
    assert np.array_equal(nodes, kd_gold["si_nodes"])
    # and the reload of that file gives identical searches
    t2 = po.OracleTree.load(p, values)
    q = kd_gold["queries"][3]
    assert t.radius_search(q, 0.08)[1].tolist() == t2.radius_search(q, 0.08)[1].tolist()
    t.close(); t2.close()


def test_radius_search_against_reference_golden(kd_gold):
    t = po.OracleTree(kd_gold["values"])
    off = kd_gold["hit_off"]
    for k, q in enumerate(kd_gold["queries"]):
        n, idx, dist, _ = t.radius_search(q, 0.08)
        assert n == off[k + 1] - off[k]
        assert np.array_equal(idx, kd_gold["hit_idx"][off[k]:off[k + 1]])  # traversal order
        assert np.array_equal(bits(dist), bits(kd_gold["hit_dist"][off[k]:off[k + 1]]))
    assert off[-1] > 100  # the fixture really contains hits
    t.close()


def test_radius_search_equals_brute_force(kd_gold):
    """Size-independent property: the hit SET is every window with squared L2 < radius."""
    values = kd_gold["values"]
    t = po.OracleTree(values)
    win = np.lib.stride_tricks.sliding_window_view(values, 6)
    for q in kd_gold["queries"][::7]:
        n, idx, dist, _ = t.radius_search(q, 0.08)
        d = ((win.astype(np.float32) - q) ** 2)
        # same association as nanoflann: ((d0+d1)+d2)+d3, then +d4, +d5
        r = ((d[:, 0] + d[:, 1]) + d[:, 2]) + d[:, 3]
        r = (r + d[:, 4]) + d[:, 5]
        assert set(np.nonzero(r < np.float32(0.08))[0].tolist()) == set(idx.tolist())
    t.close()


def test_level_means_against_reference_golden():
    g = np.load(os.path.join(GOLDEN, "levels.npz"))
    lm = np.ascontiguousarray(g["level_means"])
    for i in range(3):
        s = g[f"seq{i}"].tobytes()
        out = np.zeros(len(s), np.float32)
        n = po.lib().so_level_means(po._p(lm), 6, s, 0, len(s), po._p(out))
        assert n == len(g[f"lv{i}"])
        assert np.array_equal(bits(out[:n]), bits(g[f"lv{i}"]))


@pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_vs_live_reference_fresh_inputs():
    from sigmap_amd import pipeline, synth
    R = po.ref()
    genome = synth.make_genome(99, [30000])
    lm = synth.make_pore_model(seed=4)
    raw, _ = synth.simulate_reads(genome, lm, 4, 3)
    ix = pipeline.HostIndex.build(genome, lm, threads=2)
    values = ix.values.copy()
    ot = po.OracleTree(values)
    rt = R.ref_kdtree_build(po._p(values), None, len(values), 6, 20)
    for r in range(4):
        pa = po.add_signal(raw[r], synth.DIGITISATION, synth.RANGE, synth.OFFSET)
        em = np.zeros(len(pa) + 1, np.float32); sd = np.zeros(len(pa) + 1, np.float32)
        ne = R.ref_detect_events_entry(po._p(pa), len(pa), po._p(em), po._p(sd))
        ev, _ = po.detect_events(pa)
        assert ne == len(ev) and np.array_equal(bits(em[:ne]), bits(ev["mean"]))
        f, _ = po.generate_events(pa)
        for p in range(2, len(f) - 5, 16):
            q = np.ascontiguousarray(f[p:p + 6])
            n, idx, dist, _ = ot.radius_search(q, 0.08)
            ridx = np.zeros(4096, np.uint64); rd = np.zeros(4096, np.float32)
            rn = R.ref_radius_search(rt, po._p(q), 0.08, po._p(ridx), po._p(rd), 4096)
            assert rn == n and np.array_equal(ridx[:n], idx) and np.array_equal(bits(rd[:n]), bits(dist))
    R.ref_kdtree_free(rt)
    ot.close()


def test_event_edge_cases():
    # constant signal: no peaks -> reference would index out of bounds; oracle yields nothing
    x = np.full(4000, 90.0, np.float32)
    ev, peaks = po.detect_events(x)
    assert len(peaks) == 0 and len(ev) == 0
    f, s = po.generate_events(x)
    assert len(f) == 0
    # shorter than two long windows: t-stat is all zero (event.h:78-83)
    x = np.linspace(60, 120, 10).astype(np.float32)
    ev, peaks = po.detect_events(x)
    assert len(ev) == 0


def test_seed_positions_degenerate_sort():
    """spatial_index.cc:326-356: positions step, 2*step, ... (0 and 1 never queried)."""
    L = po.lib()
    for nq, step in [(450, 2), (51, 2), (56, 3), (7, 2), (6, 2), (100, 1)]:
        pos = np.zeros(nq + 1, np.uint32)
        n = L.so_seed_positions(nq, 6, step, po._p(pos))
        nwin = nq - 6 + 1
        assert pos[:n].tolist() == [(k + 1) * step for k in range((nwin - 1) // step)]


def test_sdtw_full_band_equals_unbanded():
    rng = np.random.default_rng(0)
    t = rng.normal(size=300).astype(np.float32)
    q = (t[100:160] + rng.normal(scale=0.05, size=60)).astype(np.float32)
    d, end = po.sdtw(t, q)
    lo = np.ones(300, np.uint32); hi = np.full(300, 60, np.uint32)
    d2, end2 = po.sdtw_banded(t, q, lo, hi)
    assert d == d2 and end == end2
    assert 155 <= end <= 165


# ------------------------------------------------------------------ CWT feature path (D1-D3)

def test_cwt_against_reference_golden():
    """The oracle's restatement of the one-scale DoG CWT vs the output of the REAL cwt.cc (golden):
    same linear map, evaluated in double instead of with an f32 FFT -> agreement to the FFT's
    rounding (2e-7 of the maximum measured; 1e-5 required)."""
    g = np.load(os.path.join(GOLDEN, "cwt.npz"))
    for r in range(4):
        for sc in (8, 5, 2):
            want = g["out%d_s%d" % (r, sc)]
            got = po.cwt_dog(g["in%d" % r], float(sc))
            assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
            if po.have_ref():
                assert np.array_equal(bits(po.ref_cwt_dog(g["in%d" % r], float(sc))), bits(want))


def test_mad_and_peaks_known_answers():
    x = np.array([5, 1, 9, 3, 7, 2, 8], np.float32)          # sorted: 1 2 3 5 7 8 9 -> [n/2] = 5
    out, mad = po.mad_normalize(x)                           # |x-5| sorted: 0 2 2 3 3 4 4 -> [3] = 3
    assert mad == 3.0 and np.array_equal(out, (x - 5) / np.float32(3))
    even = np.array([4, 1, 3, 2], np.float32)                # "fake median": sorted[2] = 3 (upper of the two)
    out, mad = po.mad_normalize(even)                        # |x-3| = 1 2 0 1 -> sorted 0 1 1 2 -> [2] = 1
    assert mad == 1.0 and np.array_equal(out, even - 3)
    sig = np.array([0, 1, 0, 2, 0, 0.1, 0, -2, 0, 3, 3, 0], np.float32)
    pk, pos = po.generate_peaks(sig, 0.5)
    # 1@1 peak (>= valley 0 + .5); 0@2 valley (<= peak 1 - .5); 2@3 peak; 0@4 valley; .1@5 no (too small);
    # -2@7 valley; 3@9 peak (3 > 0, 3 >= 3); 3@10 is not > its left neighbour
    assert pos.tolist() == [1, 2, 3, 4, 7, 9] and pk.tolist() == [1, 0, 2, 0, -2, 3]
    f, p, cw = po.cwt_features(np.linspace(60, 120, 50).astype(np.float32) + np.float32(5) * np.sin(np.arange(50, dtype=np.float32)), 4.0)
    assert len(f) == len(p) and (np.diff(p.astype(np.int64)) > 0).all()


# ------------------------------------------------------------------ A18 + reference FASTA loading, pinned to the
# REAL output_tools.h / sequence_batch.cc (tests/golden/paf.json, made by make_golden.py from oracle/_ref)

@pytest.fixture(scope="module")
def paf_gold():
    import json
    with open(os.path.join(GOLDEN, "paf.json")) as f:
        return json.load(f)


def _bits_f32(b):
    import struct
    return struct.unpack("<f", struct.pack("<I", b))[0]


def test_paf_rows_against_reference_golden(paf_gold):
    """Every golden row -- 26 records of the oracle's streaming map, 8 hand-made ones (mapq 0 / 60 / 61 / 63, '-' strand,
    nan / -nan / inf / huge tag values, start + length wrapping 32 bits, a row without chains) -- as printed by the real
    PAFOutputTools<PAFMapping>::AppendMapping / AppendUnmappedRead: the oracle's so_format_paf and the product's row
    printer (smhost_format_paf_record -> host/output_tools.h FormatPAF) give the same bytes."""
    from sigmap_amd import _abi
    H = _abi.host_lib()
    contigs = paf_gold["contigs"]
    kinds = set()
    for rec in paf_gold["rows"]:
        ref = contigs[rec["rid"]]
        m = po.Mapping()
        m.mapped, m.ref, m.read_length = rec["mapped"], rec["rid"], rec["read_length"]
        m.read_start, m.read_end, m.frag_start, m.frag_length = rec["read_start"], rec["read_end"], rec["frag_start"], rec["frag_length"]
        m.mapq, m.direction, m.chunks_used, m.n_chains, m.cm = rec["mapq"], rec["direction"], rec["ci"], rec["n_chains"], rec["cm"]
        m.s1, m.s2, m.sm, m.ad, m.at, m.aq = [_bits_f32(b) for b in rec["fbits"]]
        buf = C.create_string_buffer(4096)
        po.lib().so_format_paf(C.byref(m), rec["read_name"].encode(), ref["name"].encode(), ref["length"], buf, 4096)
        assert buf.value.decode() == rec["line"], rec["read_name"]
        buf2 = C.create_string_buffer(4096)
        k = H.smhost_format_paf_record(rec["read_name"].encode(), rec["read_length"], rec["read_start"], rec["read_end"],
                                       rec["frag_start"], rec["frag_length"], rec["mapq"], rec["direction"],
                                       rec["tags"].encode(), ref["name"].encode(), ref["length"], buf2, 4096)
        assert k == len(rec["line"]) and buf2.value.decode() == rec["line"], rec["read_name"]
        kinds.add((rec["kind"], rec["mapq"] <= 60, rec["direction"]))
    assert {("oracle", True, 1), ("oracle", True, 0), ("oracle", False, 0), ("hand", True, 0), ("hand", False, 1)} <= kinds
    # the tag values: std::to_string(float) of the reference == the "%f" both restatements print
    libc = C.CDLL(None)
    libc.snprintf.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_double]
    for b, text in paf_gold["to_string"]:
        out = C.create_string_buffer(128)
        libc.snprintf(out, 128, b"%f", C.c_double(_bits_f32(b)))   # a float promoted to double, as in a variadic call
        assert out.value.decode() == text, hex(b)


@pytest.mark.parametrize("gz", [False, True])
def test_fasta_loader_against_reference_golden(paf_gold, tmp_path, gz):
    """host/sequence_batch.cc (own reader on zlib) against the real SequenceBatch::LoadAllSequences (kseq) +
    PrepareNegativeSequenceAt on a FASTA text with comments, blank / CRLF lines, lower case, N and IUPAC codes."""
    import gzip
    from sigmap_amd import _abi
    H = _abi.host_lib()
    p = tmp_path / ("ref.fa.gz" if gz else "ref.fa")
    data = paf_gold["fasta"].encode()
    p.write_bytes(gzip.compress(data) if gz else data)
    h = H.smhost_sequences_load(str(p).encode())
    assert h
    try:
        assert H.smhost_sequences_count(h) == len(paf_gold["contigs"])
        assert H.smhost_sequences_num_bases(h) == paf_gold["num_bases"]
        for i, c in enumerate(paf_gold["contigs"]):
            name, seq, neg, ln = C.c_char_p(), C.c_char_p(), C.c_char_p(), C.c_uint32()
            assert H.smhost_sequences_get(h, i, C.byref(name), C.byref(seq), C.byref(ln), C.byref(neg)) == 0
            assert name.value.decode() == c["name"] and ln.value == c["length"]
            assert seq.value.decode("latin1") == c["seq"] and neg.value.decode("latin1") == c["neg"]
            # the oracle's reverse complement (sequence_batch.h:66-78)
            out = C.create_string_buffer(c["length"] + 1)
            po.lib().so_reverse_complement(c["seq"].encode("latin1"), c["length"], out)
            assert out.value.decode("latin1") == c["neg"]
    finally:
        H.smhost_sequences_free(h)


@pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_paf_rows_vs_live_reference_random_records(paf_gold, tmp_path):
    """300 random PAFMapping records (random float bit patterns in the tags, all mapq values of the 6-bit field)
    through the real output_tools.h here and now, against both restatements."""
    from sigmap_amd import _abi
    H, R = _abi.host_lib(), po.ref()
    fa = tmp_path / "ref.fa"
    fa.write_bytes(paf_gold["fasta"].encode())
    h = R.ref_fasta_load(str(fa).encode())
    contigs = paf_gold["contigs"]
    rng = np.random.default_rng(8)
    tmp = str(tmp_path / "row.paf").encode()
    for k in range(300):
        rid = int(rng.integers(len(contigs)))
        u = [int(x) for x in rng.integers(0, 2 ** 32, size=6, dtype=np.uint64)]
        mapq, direction, n_chains = int(rng.integers(64)), int(rng.integers(2)), int(rng.integers(0, 4))
        fb = [int(x) for x in rng.integers(0, 2 ** 32, size=6, dtype=np.uint64)]
        m = po.Mapping()
        m.mapped, m.ref, m.read_length, m.read_start, m.read_end, m.frag_start, m.frag_length = 1, rid, *u[:5]
        m.mapq, m.direction, m.chunks_used, m.n_chains, m.cm = mapq, direction, 1 + k % 30, n_chains, u[5]
        m.s1, m.s2, m.sm, m.ad, m.at, m.aq = [_bits_f32(b) for b in fb]
        name = "r%d" % k
        a = C.create_string_buffer(8192)
        po.lib().so_format_paf(C.byref(m), name.encode(), contigs[rid]["name"].encode(), contigs[rid]["length"], a, 8192)
        line = a.value.decode()
        tags = line.rstrip("\n").split("\t", 12)[12]
        b = C.create_string_buffer(8192)
        if mapq <= 60:   # OutputMappingsInVector, sigmap.cc:205-213
            n = R.ref_paf_mapped(h, rid, k, name.encode(), u[0], u[1], u[2], u[3], u[4], mapq, direction, tags.encode(), tmp, b, 8192)
        else:
            n = R.ref_paf_unmapped(h, rid, k, name.encode(), u[0], mapq, tags.encode(), tmp, b, 8192)
        assert n == len(line) and b.value.decode() == line
        # the tag values themselves through the real std::to_string
        if n_chains >= 1:
            vals = dict(t.split(":", 2)[::2] for t in tags.split("\t"))
            for key, bitsv in zip(("s1", "s2", "sm", "ad", "at", "aq"), fb):
                s = C.create_string_buffer(256)
                R.ref_to_string_float(C.c_float(_bits_f32(bitsv)), s, 256)
                assert vals[key] == s.value.decode()
        c = C.create_string_buffer(8192)
        H.smhost_format_paf_record(name.encode(), u[0], u[1], u[2], u[3], u[4], mapq, direction, tags.encode(),
                                   contigs[rid]["name"].encode(), contigs[rid]["length"], c, 8192)
        assert c.value.decode() == line
    R.ref_fasta_free(h)

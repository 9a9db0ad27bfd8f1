"""Generates tests/golden/*.npz by running the REAL reference code (oracle/_ref, compiled
from /root/reference/src by oracle/Makefile) on seeded synthetic inputs.  Run in the build
container only (needs /root/reference); the .npz files are committed, the reference is not.

  events.npz : 6 signals (pA) -> prefix sums, t-stats, peaks, events        (event.h:226)
  kdtree.npz : 40 kb point cloud -> pre-order tree dump + vind + bbox       (nanoflann.hpp:858-1004)
               + 300 queries -> hit lists in traversal order                 (nanoflann.hpp:1278-1410)
  levels.npz : pore-model table + sequences -> level means                  (pore_model.cc:57-81)
  cwt.npz    : 4 MAD-normalised signals x scales {8, 5, 2} -> DoG CWT signal (cwt.cc:1633-1841)
  paf.json   : a FASTA text -> names / bases / reverse complements as the real SequenceBatch loads them
               (sequence_batch.cc:8-15,82-122, sequence_batch.h:66-78); PAFMapping records -> the rows the
               real PAFOutputTools<PAFMapping> prints (output_tools.h:200-210,336-354), tag values by the
               real std::to_string                                    (`python make_golden.py paf` makes it alone)
"""
import ctypes as C
import json
import os
import struct
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from sigmap_amd import pipeline, synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def _f32_bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def _bits_f32(b):
    return struct.unpack("<f", struct.pack("<I", b))[0]


FASTA_TEXT = (">chrA first contig\twith a comment\n"
              "ACGTACGTTTGACNACGATCGATCGGGATCNNNNACGT\n"
              "acgtnacgtRYKMacgt\n"
              "\n"
              "GGGTTTAAACCC\n"
              ">chrB\r\n"
              "TTGACCAGTNNAC\r\n"
              "GATTACA\r\n"
              ">chrC|x=1 \n"
              "A\n"
              ">chrD\n"
              "CCCCCCCCCCGGGGGGGGGGAAAAAAAAAATTTTTTTTTTACGTACGTACGTNACGT\n")


def paf_golden():
    """The REAL output_tools.h / sequence_batch.cc (oracle/_ref) on a FASTA text and on PAFMapping records: those of
    the oracle's so_map_read for 24 seeded reads (fields as sigmap.cc:745-768 / :862-865 fill them) and hand-made
    ones (mapq 0 / 60 / 61 / 63, '-' strand, nan / inf / huge / tiny tag values, a 32-bit wrap of start + length).
    The tag string is composed as sigmap.cc:730-741 / :830-859 composes it (minus the wall-clock mt:f: tag), every
    float through the real std::to_string."""
    R = po.ref()

    def to_s(bits):
        buf = C.create_string_buffer(512)
        R.ref_to_string_float(C.c_float(_bits_f32(bits)), buf, 512)
        return buf.value.decode()

    def tags_of(ci, sl, n_chains, cm, fbits):
        t = "ci:i:%d\tsl:i:%d" % (ci, sl)
        if n_chains >= 1:
            t += "\tcm:i:%d\tnc:i:%d" % (cm, n_chains)
            for key, b in zip(("s1", "s2", "sm", "ad", "at", "aq"), fbits):
                t += "\t%s:f:%s" % (key, to_s(b))
        return t

    gold = {"fasta": FASTA_TEXT}
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "ref.fa")
        with open(fa, "wb") as f:
            f.write(FASTA_TEXT.encode())
        h = R.ref_fasta_load(fa.encode())
        n = R.ref_fasta_count(h)
        gold["num_bases"] = int(R.ref_fasta_num_bases(h))
        contigs = []
        for i in range(n):
            name, seq, neg, ln = C.c_char_p(), C.c_char_p(), C.c_char_p(), C.c_uint32()
            assert R.ref_fasta_get(h, i, C.byref(name), C.byref(seq), C.byref(ln), C.byref(neg)) == 0
            contigs.append(dict(name=name.value.decode(), length=int(ln.value), seq=seq.value.decode("latin1"),
                                neg=neg.value.decode("latin1")))
        gold["contigs"] = contigs
        rows = []
        # ---- records of real reads (the oracle's streaming map on the small world of tests/conftest.py)
        genome = synth.make_genome(7, [150000, 100000, 50000])
        lm = synth.make_pore_model()
        raw, _ = synth.simulate_reads(genome, lm, 24, 11)
        ix = pipeline.HostIndex.build(genome, lm, threads=4)
        tree = po.OracleTree(ix.values.copy())
        ref_lens = np.array([len(g) for g in genome], dtype=np.uint32)
        rng = np.random.default_rng(3)
        sigs = [po.add_signal(r, synth.DIGITISATION, synth.RANGE, synth.OFFSET) for r in raw]
        sigs += [rng.normal(90, 12, size=4000).astype(np.float32), rng.normal(90, 12, size=12000).astype(np.float32)]
        for i, x in enumerate(sigs):
            m = po.map_read(tree, ix.positions, x, 3, ref_lens)
            rows.append(dict(kind="oracle", read_name="read%d" % i, rid=int(m.ref) % n, mapped=int(m.mapped), read_length=int(m.read_length),
                             read_start=int(m.read_start), read_end=int(m.read_end), frag_start=int(m.frag_start),
                             frag_length=int(m.frag_length), mapq=int(m.mapq), direction=int(m.direction),
                             ci=int(m.chunks_used), n_chains=int(m.n_chains), cm=int(m.cm),
                             fbits=[_f32_bits(v) for v in (m.s1, m.s2, m.sm, m.ad, m.at, m.aq)]))
        tree.close()
        # ---- hand-made records
        NAN, NNAN, INF, NINF = 0x7fc00000, 0xffc00000, 0x7f800000, 0xff800000
        one = _f32_bits(1.0)
        hand = [
            dict(mapq=0, direction=1, fbits=[_f32_bits(33.25), 0, _f32_bits(33.25), _f32_bits(0.0123456), _f32_bits(7.5), _f32_bits(8.999999)]),
            dict(mapq=60, direction=0, fbits=[_f32_bits(1e30), _f32_bits(1e-30), _f32_bits(-0.0), _f32_bits(123456.789), one, one]),
            dict(mapq=61, direction=1, fbits=[NAN, NNAN, INF, NINF, _f32_bits(0.5), _f32_bits(2.5e-7)]),
            dict(mapq=63, direction=0, fbits=[one] * 6),
            dict(mapq=17, direction=0, frag_start=4294967000, frag_length=1000, fbits=[one] * 6),
            dict(mapq=61, direction=0, n_chains=0, read_start=0, read_end=0, frag_start=0, frag_length=0, fbits=[0] * 6),
            dict(mapq=1, direction=1, n_chains=1, cm=10, rid=2, read_length=1, fbits=[_f32_bits(10.0), 0, _f32_bits(10.0), NAN, NAN, NAN]),
            dict(mapq=59, direction=1, rid=3, read_name="a read name with spaces", fbits=[_f32_bits(3.4028235e38)] * 6),
        ]
        for k, hrec in enumerate(hand):
            rec = dict(kind="hand", read_name="hand%d" % k, rid=k % n, read_length=123456 + k, read_start=17 * k, read_end=4000 + k,
                       frag_start=1000 * k + 1, frag_length=3999, ci=1 + k, n_chains=3, cm=11 + k)
            rec.update(hrec)
            rec["mapped"] = 1 if rec["mapq"] <= 60 else 0
            rows.append(rec)
        tmp = os.path.join(td, "row.paf").encode()
        for rec in rows:
            tags = tags_of(rec["ci"], rec["read_length"], rec["n_chains"], rec["cm"], rec["fbits"])
            buf = C.create_string_buffer(4096)
            # OutputMappingsInVector, sigmap.cc:205-213, at the CLI's mapq threshold 0
            if rec["mapq"] <= 60:
                k = R.ref_paf_mapped(h, rec["rid"], 0, rec["read_name"].encode(), rec["read_length"], rec["read_start"],
                                     rec["read_end"], rec["frag_start"], rec["frag_length"], rec["mapq"], rec["direction"],
                                     tags.encode(), tmp, buf, 4096)
            else:
                k = R.ref_paf_unmapped(h, rec["rid"], 0, rec["read_name"].encode(), rec["read_length"], rec["mapq"],
                                       tags.encode(), tmp, buf, 4096)
            assert k > 0
            rec["tags"] = tags
            rec["line"] = buf.value.decode()
        gold["rows"] = rows
        gold["to_string"] = [[b, to_s(b)] for b in (0, 0x80000000, one, NAN, NNAN, INF, NINF, _f32_bits(1e30), _f32_bits(1e-30),
                                                     _f32_bits(0.1), _f32_bits(2.5), _f32_bits(3.4028235e38), 1, _f32_bits(16777217.0))]
        R.ref_fasta_free(h)
    with open(os.path.join(OUT, "paf.json"), "w") as f:
        json.dump(gold, f, indent=1)
    print("paf.json:", len(gold["rows"]), "rows,", len(gold["contigs"]), "contigs")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "paf":
        paf_golden()
        return
    R = po.ref()
    genome = synth.make_genome(21, [25000, 15000])
    lm = synth.make_pore_model()
    raw, _ = synth.simulate_reads(genome, lm, 6, 5)
    # ---- events
    ev = {}
    for r in range(6):
        pa = pipeline.raw_to_pa(raw[r], synth.DIGITISATION, synth.RANGE, synth.OFFSET)
        if r == 5:
            pa = pa[:1777]  # a ragged (non-4000) chunk
        n = len(pa)
        ps = np.zeros(n + 1, np.float32); pss = np.zeros(n + 1, np.float32)
        t1 = np.zeros(n + 1, np.float32); t2 = np.zeros(n + 1, np.float32)
        peaks = np.zeros(n + 1, np.uint64); es = np.zeros(n + 1, np.uint64); el = np.zeros(n + 1, np.uint64)
        em = np.zeros(n + 1, np.float32); sd = np.zeros(n + 1, np.float32)
        npk = C.c_size_t()
        ne = R.ref_detect_events(po._p(pa), n, po._p(ps), po._p(pss), po._p(t1), po._p(t2), po._p(peaks),
                                 C.byref(npk), po._p(es), po._p(el), po._p(em), po._p(sd))
        assert ne > 0
        ev[f"x{r}"] = pa; ev[f"ps{r}"] = ps; ev[f"pss{r}"] = pss; ev[f"t1_{r}"] = t1; ev[f"t2_{r}"] = t2
        ev[f"peaks{r}"] = peaks[:npk.value]; ev[f"start{r}"] = es[:ne]; ev[f"length{r}"] = el[:ne]
        ev[f"mean{r}"] = em[:ne]; ev[f"stdv{r}"] = sd[:ne]
    np.savez_compressed(os.path.join(OUT, "events.npz"), **ev)
    # ---- k-d tree + radius search
    ix = pipeline.HostIndex.build(genome, lm, threads=2)
    values = ix.values.copy(); positions = ix.positions.copy()
    rt = R.ref_kdtree_build(po._p(values), po._p(positions), len(values), 6, 20)
    nn = R.ref_kdtree_dump(rt, None, None, None, None, None, None, None)
    il = np.zeros(nn, np.int32); a = np.zeros(nn, np.uint64); b = np.zeros(nn, np.uint64)
    lo = np.zeros(nn, np.float32); hi = np.zeros(nn, np.float32)
    vind = np.zeros(len(values) - 5, np.uint64); bbox = np.zeros(12, np.float32)
    R.ref_kdtree_dump(rt, po._p(il), po._p(a), po._p(b), po._p(lo), po._p(hi), po._p(vind), po._p(bbox))
    queries = []
    for r in range(6):
        f, _ = po.generate_events(ev[f"x{r}"])
        for p in range(2, len(f) - 5, 8):
            queries.append(f[p:p + 6])
    # plus queries outside the root bounding box and exact data points
    queries.append(np.array([9, 9, 9, 9, 9, 9], np.float32))
    queries.append(np.array([-9, 0, 0, 0, 0, 9], np.float32))
    for k in (0, 1234, 20000):
        queries.append(values[k:k + 6].copy())
    queries = np.array(queries, dtype=np.float32)
    hit_off = [0]; hit_idx = []; hit_dist = []
    for q in queries:
        ridx = np.zeros(1 << 16, np.uint64); rd = np.zeros(1 << 16, np.float32)
        qq = np.ascontiguousarray(q)
        n = R.ref_radius_search(rt, po._p(qq), 0.08, po._p(ridx), po._p(rd), 1 << 16)
        hit_idx.append(ridx[:n].copy()); hit_dist.append(rd[:n].copy()); hit_off.append(hit_off[-1] + n)
    # .si bytes as the reference writes them (pointer fields masked to 0/1: they are heap addresses)
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "g.si")
        R.ref_kdtree_save(rt, p.encode())
        si = np.fromfile(p, dtype=np.uint8)
    hdr = 8 + 4 + 8 + 48 + 8 + 8 + 8 * len(vind)
    nodes_raw = si[hdr:].reshape(-1, 32).copy()
    ptrs = nodes_raw[:, 16:].view(np.uint64)
    ptrs[ptrs != 0] = 1
    leaf_rows = (ptrs[:, 0] == 0) & (ptrs[:, 1] == 0)
    nodes_raw[~leaf_rows, 12:16] = 0  # uninitialised union tail of inner nodes
    np.savez_compressed(os.path.join(OUT, "kdtree.npz"), values=values, positions=positions, is_leaf=il, a=a, b=b,
                        lo=lo, hi=hi, vind=vind, bbox=bbox, queries=queries, hit_off=np.array(hit_off, np.int64),
                        hit_idx=np.concatenate(hit_idx), hit_dist=np.concatenate(hit_dist),
                        si_header=si[:hdr].copy(), si_nodes=nodes_raw)
    R.ref_kdtree_free(rt)
    # ---- level means
    with tempfile.TemporaryDirectory() as td:
        mp = os.path.join(td, "model.txt")
        synth.write_pore_model(mp, lm)
        seqs = [genome[0][:500], genome[1][100:333], b"ACGTNNACGTACGTTTGACNACGATCGATCGGGATCN" * 3]
        lv = {}
        for i, s in enumerate(seqs):
            out = np.zeros(len(s), np.float32)
            k = C.c_int()
            n = R.ref_level_means(mp.encode(), s, 0, len(s), po._p(out), C.byref(k))
            lv[f"seq{i}"] = np.frombuffer(s, dtype=np.uint8); lv[f"lv{i}"] = out[:n].copy()
        lv["level_means"] = lm
    np.savez_compressed(os.path.join(OUT, "levels.npz"), **lv)
    # ---- CWT (the real cwt.cc on MAD-normalised reads; one ragged)
    cw = {}
    for r in range(4):
        pa = pipeline.raw_to_pa(raw[r], synth.DIGITISATION, synth.RANGE, synth.OFFSET)
        if r == 3:
            pa = pa[:2501]
        a, _ = po.mad_normalize(pa)
        cw["in%d" % r] = a
        for sc in (8, 5, 2):
            cw["out%d_s%d" % (r, sc)] = po.ref_cwt_dog(a, float(sc))
    np.savez_compressed(os.path.join(OUT, "cwt.npz"), **cw)
    paf_golden()
    print("golden written:", os.listdir(OUT))


if __name__ == "__main__":
    main()

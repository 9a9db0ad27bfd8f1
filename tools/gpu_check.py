"""Stage-by-stage GPU vs oracle comparison with verbose diagnostics (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sigmap_amd import pipeline, synth, _abi
from oracle import pyoracle as po

def main():
    nreads = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    genome = synth.make_genome(7, [150000, 100000, 50000])
    lm = synth.make_pore_model()
    raw, truth = synth.simulate_reads(genome, lm, nreads, 11)
    t0 = time.time()
    ix = pipeline.HostIndex.build(genome, lm, threads=0)
    print("index", ix.n_points, ix.n_nodes, ix.depth, "%.2fs" % (time.time() - t0))
    dev = pipeline.DeviceMapper(0)
    dev.params.walk_blocks_per_cu = G
    dev.upload_index(ix)
    pa = [po.add_signal(r, synth.DIGITISATION, synth.RANGE, synth.OFFSET) for r in raw]
    offsets = np.concatenate([[0], np.cumsum([len(x) for x in pa])]).astype(np.uint32)
    samples = np.concatenate(pa)
    # --- events
    feat, stdv, nf, ne, st = dev.events(samples, offsets, feat_stride=2736)
    bad = 0
    ofeat = []
    for i in range(nreads):
        f, s = po.generate_events(pa[i])
        ofeat.append(f)
        ev, pk = po.detect_events(pa[i])
        if nf[i] != len(f) or ne[i] != len(ev) or not np.array_equal(feat[i, :nf[i]].view(np.uint32), f.view(np.uint32)) \
           or not np.array_equal(stdv[i, :nf[i]].view(np.uint32), s.view(np.uint32)):
            bad += 1
            if bad < 4:
                print("EVENT MISMATCH read", i, nf[i], len(f), ne[i], len(ev))
                k = min(nf[i], len(f))
                d = np.nonzero(feat[i, :k].view(np.uint32) != f[:k].view(np.uint32))[0]
                print("  first diff at", d[:5], feat[i, d[:3]], f[d[:3]])
    print("events: mismatching reads", bad, "of", nreads, "; features", nf[:8], "events", ne[:8])
    # --- seed lookup (raw queries)
    tree = po.OracleTree(ix.values.copy())
    qs = []
    for i in range(min(nreads, 16)):
        f = ofeat[i]
        for p in range(2, len(f) - 5, 2):
            qs.append(f[p:p + 6])
    qs = np.array(qs, dtype=np.float32)
    ora = [tree.radius_search(q, 0.08) for q in qs]
    for strict in (1, 0):
        dev.params.reference_walk = strict
        counts, idx, dist, nodes, pts = dev.seed_lookup(qs, cap=1024)
        badq = 0
        tot = 0
        for k in range(len(qs)):
            n, oi, od, stt = ora[k]
            tot += n
            ok = counts[k] == n and np.array_equal(idx[k, :n], oi) and np.array_equal(dist[k, :n].view(np.uint32), od.view(np.uint32))
            if strict:
                ok = ok and nodes[k] == stt[0] and pts[k] == stt[2]
            if not ok:
                badq += 1
                if badq < 4:
                    print("QUERY MISMATCH", k, counts[k], n, nodes[k], stt, pts[k])
                    print("   gpu", idx[k, :min(n, 6)], "ora", oi[:6])
        print("seed lookup (reference_walk=%d): mismatching queries" % strict, badq, "of", len(qs), "hits", tot,
              "mean nodes", nodes.mean(), "mean pts", pts.mean())
    # --- full pipeline
    t0 = time.time()
    res = dev.map_chunks(samples, offsets, n_refs=len(genome))
    print("map_chunks %.3fs" % (time.time() - t0))
    ref_lens = np.array([len(g) for g in genome], dtype=np.uint32)
    L = _abi.host_lib()
    import ctypes as C
    mp = _abi.MapParams(); L.smhost_default_map_params(C.byref(mp))
    names = [b"contig%d" % i for i in range(len(genome))]
    arr = (C.c_char_p * len(names))(*names)
    badm = 0
    nm = 0
    for i in range(nreads):
        m = po.map_read(tree, ix.positions, pa[i], len(genome), ref_lens)
        oline = po.format_paf(m, "read%d" % i, [n.decode() for n in names], ref_lens)
        buf = C.create_string_buffer(2048)
        r = res[i:i + 1].copy()
        nev = int(r["n_features"][0]) if r["n_features"][0] > 50 else 0
        L.smhost_format_paf(_abi.ptr(r), b"read%d" % i, len(pa[i]), 0, nev, arr, _abi.ptr(ref_lens), len(genome), C.byref(mp), buf, 2048, None)
        gline = buf.value.decode()
        nm += m.mapped
        if gline != oline:
            badm += 1
            if badm < 6:
                print("PAF MISMATCH", i, "status", r["status"][0])
                print("  gpu:", gline.strip())
                print("  ora:", oline.strip())
                print("  gpu n_anchors", r["n_anchors"][0], "ora anchors", m.stats.anchors, "nq", r["n_queries"][0], m.stats.queries, "hits", r["hits_total"][0], m.stats.hits)
    print("pipeline: PAF mismatches", badm, "of", nreads, "mapped", nm)
    for name in ["events", "seed_lookup", "seed_overflow", "sort", "chain"]:
        pass
    return 0 if (bad == 0 and badq == 0 and badm == 0) else 1

if __name__ == "__main__":
    sys.exit(main())

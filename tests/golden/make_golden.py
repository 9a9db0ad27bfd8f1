"""Generates tests/golden/*.npz by running the REAL reference code (oracle/_ref, compiled
from /root/reference/src by oracle/Makefile) on seeded synthetic inputs.  Run in the build
container only (needs /root/reference); the .npz files are committed, the reference is not.

  events.npz : 6 signals (pA) -> prefix sums, t-stats, peaks, events        (event.h:226)
  kdtree.npz : 40 kb point cloud -> pre-order tree dump + vind + bbox       (nanoflann.hpp:858-1004)
               + 300 queries -> hit lists in traversal order                 (nanoflann.hpp:1278-1410)
  levels.npz : pore-model table + sequences -> level means                  (pore_model.cc:57-81)
  cwt.npz    : 4 MAD-normalised signals x scales {8, 5, 2} -> DoG CWT signal (cwt.cc:1633-1841)
"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from sigmap_amd import pipeline, synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
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
    print("golden written:", os.listdir(OUT))


if __name__ == "__main__":
    main()

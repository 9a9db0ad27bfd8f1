"""BASELINE.json configs at their full sizes on one GPU (`-m gpu`): configs[3] -- 50 000 reads vs one
4.6 Mb contig, sDTW band sweep 32 / 64 / 128 -- and configs[4]'s per-GPU shard -- 25 000 of the 200 000
reads vs one 250 Mb contig (4.98e8 points, 27 GB resident), where more than half of the reads hold a
query with more than 5000 hits, i.e. where the ORDER of the radius search decides which hits survive
the cap (spatial_index.cc:371-372) and with it the PAF.  Every read goes through the device; a sample
is re-mapped by the oracle (PAF row, anchor and hit counts, refined distance bit for bit); the rest is
held to size-independent properties."""
import ctypes as C
import time

import numpy as np
import pytest

from conftest import bits
from oracle import pyoracle as po
from sigmap_amd import _abi, pipeline, synth

pytestmark = pytest.mark.gpu


def _paf_line(res_row, name, read_len, names, ref_lens):
    L = _abi.host_lib()
    mp = _abi.MapParams(); L.smhost_default_map_params(C.byref(mp))
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    r = np.array([res_row], dtype=_abi.RESULT_DTYPE)
    nev = int(r["n_features"][0]) if r["n_features"][0] > 50 else 0
    buf = C.create_string_buffer(2048)
    L.smhost_format_paf(_abi.ptr(r), name.encode(), read_len, 0, nev, arr, _abi.ptr(ref_lens), len(names), C.byref(mp),
                        buf, 2048, None)
    return buf.value.decode()


def _map_all(dev, raw, n_refs, batch, bands=(), pad=32):
    """Every read through the device in batches of `batch` resident reads; returns the result records and,
    per band, (dist, end) of the refinement."""
    n = len(raw)
    dev.configure_reads(batch, 4000)
    res = np.zeros(n, dtype=_abi.RESULT_DTYPE)
    refined = {b: (np.zeros(n, np.float32), np.zeros(n, np.int64)) for b in bands}
    for lo in range(0, n, batch):
        hi = min(n, lo + batch)
        flat = np.ascontiguousarray(raw[lo:hi].reshape(-1))
        ro = (np.arange(hi - lo + 1, dtype=np.uint64) * raw.shape[1])
        lens = dev.upload_raw(flat, ro, synth.DIGITISATION, synth.RANGE, synth.OFFSET)
        assert (lens == raw.shape[1]).all()
        dev.select(np.arange(hi - lo), np.zeros(hi - lo))
        dev.run(n_refs); dev.sync()
        res[lo:hi] = dev.download()
        for b in bands:
            dev.refine(band=b, pad=pad); dev.sync()
            d, e = dev.refine_download()
            refined[b][0][lo:hi] = d
            refined[b][1][lo:hi] = e
            assert dev.download().tobytes() == res[lo:hi].tobytes()   # the refinement never touches the records
    return res, refined


def _oracle_refine(r, x, sig, offs, band, pad):
    """smhip_refine_run's window and band (include/sigmap_hip.h) evaluated by the oracle."""
    c0 = r["chains"][0]
    strand = 0 if c0["direction"] == 1 else 1
    so, se = int(offs[2 * c0["ref"] + strand]), int(offs[2 * c0["ref"] + strand + 1])
    start, endp = int(c0["start"]), int(c0["end"])
    t0 = max(0, start - pad); t1 = min(se - so, endp + 6 + pad)
    q, _ = po.generate_events(x)
    qa, qb = int(r["c0_last_qpos"]), int(r["c0_first_qpos"])
    den = max(1, endp - start)
    ta = np.arange(t0, t1, dtype=np.int64)
    centre = 1 + qa + ((ta - start) * (qb - qa)) // den
    lo = np.clip(centre - band // 2, 1, None).astype(np.int64)
    hi = np.maximum(np.clip(centre + band // 2, None, len(q)), 0).astype(np.int64)
    d, e = po.sdtw_banded(sig[so + t0:so + t1], q, lo.astype(np.uint32), hi.astype(np.uint32))
    return d, (t0 + e if e >= 0 else -1)


def test_config3_ecoli_band_sweep_full_size():
    """BASELINE configs[3]: 50 000 reads vs 1 x 4.6 Mb, bands 32 / 64 / 128."""
    genome = synth.make_genome(11, [4600000])
    lm = synth.make_pore_model()
    n = 50000
    raw, truth = synth.simulate_reads(genome, lm, n, 77)
    ix = pipeline.HostIndex.build(genome, lm, threads=0)
    dev = pipeline.DeviceMapper(0)
    dev.upload_index(ix)
    sig, offs = pipeline.reference_signals(genome, lm)
    dev.upload_reference(sig, offs)
    res, refined = _map_all(dev, raw, 1, 12500, bands=(32, 64, 128))
    assert (res["status"] == 0).all()
    c0 = res["chains"][:, 0]
    ok = (res["n_chains"] > 0) & ((c0["direction"] == 1) == (truth["strand"] == 0))
    assert ok.mean() > 0.95
    # properties over all 50 000: a wider band admits a superset of warping paths
    d32, d64, d128 = (refined[b][0] for b in (32, 64, 128))
    r32 = refined[32][1] >= 0
    assert r32.mean() > 0.9 and (refined[64][1] >= 0)[r32].all() and (refined[128][1] >= 0)[refined[64][1] >= 0].all()
    assert (d64[r32] <= d32[r32]).all() and (d128[r32] <= d64[r32]).all() and (d128[r32] < d32[r32]).any()
    # oracle sample: 64 reads per band (disjoint thirds of 192), PAF + dt bit-exact
    tree = po.OracleTree.from_host_index(ix)
    ref_lens = np.array([len(genome[0])], np.uint32)
    names = ["contig0"]
    rng = np.random.default_rng(5)
    sample = rng.choice(n, 192, replace=False)
    refined_checked = 0
    for k, i in enumerate(sample):
        x = po.add_signal(raw[i], synth.DIGITISATION, synth.RANGE, synth.OFFSET)
        m = po.map_read(tree, ix.positions, x, 1, ref_lens)
        assert _paf_line(res[i], "read%d" % i, len(x), names, ref_lens) == po.format_paf(m, "read%d" % i, names, ref_lens)
        assert res[i]["n_anchors"] == m.stats.anchors and res[i]["hits_total"] == m.stats.hits
        band = (32, 64, 128)[k % 3]
        if res[i]["n_chains"] == 0:
            assert refined[band][1][i] == -1
            continue
        d, e = _oracle_refine(res[i], x, sig, offs, band, 32)
        if e < 0:
            assert refined[band][1][i] == -1
        else:
            assert bits(refined[band][0][i]) == bits(np.float32(d)) and refined[band][1][i] == e, (i, band)
            refined_checked += 1
    assert refined_checked >= 150
    tree.close()
    dev.close()


def test_config4_chr1_250mb_shard():
    """BASELINE configs[4], one GPU's shard: 25 000 reads vs 1 x 250 Mb."""
    t0 = time.time()
    genome = synth.make_genome(13, [250000000])
    lm = synth.make_pore_model()
    n = 25000
    raw, truth = synth.simulate_reads(genome, lm, n, 99)
    ix = pipeline.HostIndex.build(genome, lm, threads=0)
    assert ix.n_points > 4.9e8
    dev = pipeline.DeviceMapper(0)
    dev.upload_index(ix)
    res, _ = _map_all(dev, raw, 1, 10000)          # batches of 10 000 chunks: 1.8 G anchors, ~170 GB of device buffers each
    assert (res["status"] == 0).all()               # no capacity flag: nothing was truncated on the device
    c0 = res["chains"][:, 0]
    ok = (res["n_chains"] > 0) & ((c0["direction"] == 1) == (truth["strand"] == 0)) & \
         (np.abs(c0["start"].astype(np.int64) - np.where(truth["strand"] == 0, truth["pos"],
                                                         len(genome[0]) - truth["pos"] - 450)) < 2000)
    assert ok.mean() > 0.9, ok.mean()
    # a chunk holds a query beyond the 5000-hit cap iff it lost hits: hits_total > anchors (no carry-over here)
    over = res["hits_total"] > res["n_anchors"]
    assert over.mean() > 0.3, over.mean()
    sample = np.concatenate([np.flatnonzero(over)[:24], np.flatnonzero(~over)[:8]])
    assert over[sample].sum() >= 16
    tree = po.OracleTree.from_host_index(ix)
    ref_lens = np.array([len(genome[0])], np.uint32)
    names = ["chr1"]
    proved = 0
    for i in sample:
        x = po.add_signal(raw[i], synth.DIGITISATION, synth.RANGE, synth.OFFSET)
        m = po.map_read(tree, ix.positions, x, 1, ref_lens)
        assert _paf_line(res[i], "read%d" % i, len(x), names, ref_lens) == po.format_paf(m, "read%d" % i, names, ref_lens), i
        assert res[i]["n_anchors"] == m.stats.anchors and res[i]["hits_total"] == m.stats.hits
        proved += int(m.stats.hits > m.stats.anchors)   # the oracle, too, cut a query at 5000 hits
    assert proved >= 16
    # idempotence / order independence at full size: the second batch again, in reverse read order
    back = np.ascontiguousarray(raw[15000:][::-1])
    again, _ = _map_all(dev, back, 1, 10000)
    assert again[::-1].tobytes() == res[15000:].tobytes()
    # (this index puts 0.6 % of the queries beyond the cap, and the census of the index says so.  AUTO searched with
    # the set walk and RANKED those queries' hits in the reference's order)
    c = dev.debug_counters()
    assert c[6] == 2 and c[7] > 1000 and (dev.index_census() > 5000).mean() > 0.0005
    # the same records from the search in the reference's order for every query, and from round 3's route (set
    # walk + second search of the over-cap queries), on 2 000 chunks
    for mode, second in ((_abi.SEARCH_EXACT, 0), (_abi.SEARCH_SET, 1)):
        dev.params.search_mode = mode
        dev.set_knob("second_search", second)
        try:
            other, _ = _map_all(dev, raw[:2000], 1, 2000)
            assert dev.debug_counters()[6] == (0 if mode == _abi.SEARCH_EXACT else 2)
        finally:
            dev.params.search_mode = _abi.SEARCH_AUTO
            dev.set_knob("second_search", 0)
        for k in other.dtype.names:
            if k not in ("nodes_visited", "points_tested"):
                assert np.array_equal(other[k], res[:2000][k]), (mode, k)
    # the census walk (exactly nanoflann's node set) returns the same records on 2 000 chunks
    dev.params.reference_walk = 1
    try:
        census, _ = _map_all(dev, raw[:2000], 1, 2000)
    finally:
        dev.params.reference_walk = 0
    for k in census.dtype.names:
        if k not in ("nodes_visited", "points_tested"):
            assert np.array_equal(census[k], res[:2000][k]), k
    assert (census["nodes_visited"] >= res[:2000]["nodes_visited"]).all()
    tree.close()
    dev.close()
    print("config4 shard test: %.0f s" % (time.time() - t0))


def test_cli_streaming_600_multichunk_reads(tmp_path, small_world):
    """Multi-chunk streaming at a size where every branch is taken many times: 600 reads of 2-4 chunks
    through `sigmap -m`, a third of them starting with 4000-9000 samples of junk (chains of early chunks
    that do not stop the read are carried into the next chunk, spatial_index.cc:303-322), some with ADC
    samples the (30,200) pA filter removes (chunk boundaries follow the FILTERED array), against the
    oracle's streaming result, whole PAF in the reference's order."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "sigmap_amd", "bin", "sigmap")
    genome, lm = small_world["genome"], small_world["lm"]
    names = ["contig0", "contig1", "contig2"]
    synth.write_fasta(str(tmp_path / "ref.fa"), genome, names)
    synth.write_pore_model(str(tmp_path / "model.txt"), lm)
    rng = np.random.default_rng(21)
    raws = []
    for k, nch in enumerate((2, 3, 4)):
        raw, _ = synth.simulate_reads(genome, lm, 200, 300 + k, n_chunks=nch)
        for i, r in enumerate(raw):
            if i % 3 == 0:      # junk first: the read only maps from a later chunk on
                junk = rng.integers(350, 1000, size=int(rng.integers(4000, 9000))).astype(np.int16)
                r = np.concatenate([junk, r])[:len(r)]
            if i % 7 == 0:      # samples outside (30, 200) pA somewhere in the read
                at = int(rng.integers(0, len(r) - 200))
                r = np.insert(r, at, np.full(int(rng.integers(1, 150)), 32767 if i % 2 else -2000, np.int16))
            raws.append(np.ascontiguousarray(r, dtype=np.int16))
    os.makedirs(tmp_path / "reads")
    synth.write_sgr(str(tmp_path / "reads" / "a.sgr"), raws)
    r = subprocess.run([exe, "-i", "-r", str(tmp_path / "ref.fa"), "-p", str(tmp_path / "model.txt"),
                        "-o", str(tmp_path / "idx")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "-m", "-r", str(tmp_path / "ref.fa"), "-p", str(tmp_path / "model.txt"),
                        "-x", str(tmp_path / "idx"), "-s", str(tmp_path / "reads"), "-o", str(tmp_path / "out.paf"),
                        "--batch-chunks", "500", "--streams", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    strip = lambda l: "\t".join(f for f in l.split("\t") if not f.startswith("mt:f:"))
    got = [strip(l) for l in open(tmp_path / "out.paf").read().splitlines()]
    # the multi-GPU code path of the CLI -- contiguous read blocks per "GPU", lanes per GPU, rows merged
    # by (reference, read) -- with three "GPUs" that are all this box's one GPU: the same PAF
    r = subprocess.run([exe, "-m", "-r", str(tmp_path / "ref.fa"), "-p", str(tmp_path / "model.txt"),
                        "-x", str(tmp_path / "idx"), "-s", str(tmp_path / "reads"), "-o", str(tmp_path / "out3.paf"),
                        "--gpus", "3", "--gpus-share-device", "--batch-chunks", "400", "--streams", "2"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert [strip(l) for l in open(tmp_path / "out3.paf").read().splitlines()] == got
    # the default number of contexts: up to two, as the reads need -- two here (600 reads for 250 read slots each), one
    # when they all fit one context's share of --batch-chunks (the CLI says how many lanes mapped); the same PAF
    for k, (budget, lanes) in enumerate((("500", 2), ("4000", 1))):
        env = dict(os.environ, SIGMAP_STREAM_STATS="1")
        r = subprocess.run([exe, "-m", "-r", str(tmp_path / "ref.fa"), "-p", str(tmp_path / "model.txt"),
                            "-x", str(tmp_path / "idx"), "-s", str(tmp_path / "reads"), "-o", str(tmp_path / ("outd%d.paf" % k)),
                            "--batch-chunks", budget], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        assert '"lanes": %d,' % lanes in r.stderr, r.stderr[-600:]
        assert [strip(l) for l in open(tmp_path / ("outd%d.paf" % k)).read().splitlines()] == got
    ix = small_world["ix"]
    tree = po.OracleTree.from_host_index(ix)
    ref_lens = small_world["ref_lens"]
    rows, used_chunks = [], []
    for i, rw in enumerate(raws):
        pa = po.add_signal(rw, synth.DIGITISATION, synth.RANGE, synth.OFFSET)
        m = po.map_read(tree, ix.positions, pa, 3, ref_lens)
        used_chunks.append(m.chunks_used)
        rows.append((m.ref, i, po.format_paf(m, "read%d" % i, names, ref_lens).rstrip("\n")))
    rows.sort(key=lambda t: (t[0], t[1]))
    assert got == [t[2] for t in rows]
    used_chunks = np.array(used_chunks)
    assert (used_chunks >= 2).sum() >= 150 and (used_chunks >= 3).sum() >= 40   # carry-over really happened
    tree.close()


def test_bench_two_ranks_flow_on_one_gpu():
    """bench.py's N > 1 flow end to end with two ranks (launched like the driver launches it): rank 0
    builds the index, the six walker arrays are broadcast, the peer attaches what it received, every
    rank maps its own shard, barrier, max-over-ranks time, one JSON line.  RCCL refuses two ranks on one
    GPU, so on a one-GPU box the arrays travel through `--backend gloo`; everything else is the N > 1
    path the driver runs."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--steps", "2", "--warmup", "1", "--reads", "1500", "--contigs", "4", "--contig-len", "300000", "--no-cpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints ONE line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "weak"
    assert j["config"]["contexts_agree"] and j["config"]["census_results_identical"]
    assert j["config"]["chunks_with_capacity_flags"] == 0 and j["config"]["mapped_fraction"] > 0.95
    assert j["pcie_inclusive"]["results_identical"]
    assert abs(j["value"] - 2 * 1500 * 2 / (j["ms_per_step"] * 2 * 1e-3)) < 1e-3 * j["value"]


def test_bench_one_rank_over_rccl_forced_broadcast():
    """bench.py launched like the driver launches it, with ONE rank, backend nccl (= RCCL) and
    --force-bcast: init_process_group(backend="nccl", device_id=...), the meta + six dist.broadcast
    calls of multigpu.replicate_index, barrier and the max-over-ranks all_reduce all run over RCCL in a
    process that also holds libsigmap_hip.so; two of the three contexts map from the broadcast copy and
    agree byte for byte with the one that maps from the uploaded arrays."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "nccl",
           "--force-bcast", "--steps", "3", "--warmup", "1", "--reads", "1500", "--contigs", "4", "--contig-len", "300000",
           "--depth", "3", "--no-cpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["steps"] == 3
    fb = j["config"]["forced_broadcast"]
    assert fb == {"backend": "rccl", "ranks": 1, "contexts_on_broadcast_copy": 2}
    assert j["config"]["contexts_agree"] and j["config"]["census_results_identical"]
    assert j["config"]["chunks_with_capacity_flags"] == 0 and j["config"]["mapped_fraction"] > 0.95
    assert j["pcie_inclusive"]["results_identical"]


def test_stream_bench_reads_that_use_every_chunk():
    """tools/stream_bench.py end to end at a size a test can afford: 600 reads of 30 chunks through `sigmap -m` -- junk-
    prefixed reads that map from a later chunk on with the junk's chains carried along, junk-only reads that use all
    max_num_chunks = 30 chunks (sigmap.cc:647: the loop's own limit) --, one chunk per read and round pulled from pinned
    memory by the device, 48 sampled reads of every kind against the oracle's streaming map (the script asserts 0
    mismatches), the CLI's per-phase statistics present."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stream_bench.py"), "--reads", "600", "--chunks-per-read", "30",
                        "--contigs", "4", "--contig-len", "300000", "--check", "48", "--runs", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["oracle_check"]["mismatches"] == 0 and j["oracle_check"]["reads"] >= 40
    assert j["oracle_check"]["max_chunks_used"] >= 10 and j["read_kinds"]["junk_only"] >= 4
    assert j["mean_chunks_used"] > 1.5 and j["rounds"] >= 30 and j["batches_mapped_twice"] == 0
    assert set(j["lane_seconds"]) == {"admit_upload", "build_batch", "device", "download_results", "download_anchors", "decide"}

#!/usr/bin/env python3
"""Throughput of the product's real mode: multi-chunk STREAMING reads through `sigmap -m`'s chunk-round loop
(sigmap.cc:647-696 of the reference: chunk after chunk until a stop rule fires; spatial_index.cc:303-322: the chains'
anchors are carried into the next chunk), SURVEY.md 8(f) N2.

    tools/stream_bench.py --reads 10000 --chunks-per-read 8 [--streams N] [--batch-chunks 32768] [--check 96]

Reads (seeded): C chunks of 4000 samples each; a third start with junk (uniform random ADC counts inside the pA filter's
range, 1 .. C / 3 chunks of it) so that the read only maps from a later chunk on and earlier chains are carried over;
2 % are junk throughout (they never map and use every chunk up to --max-num-chunks); the rest map in their first
chunk or two.  The reference is bench.py's (16 x 750 kb).  The CLI runs with SIGMAP_STREAM_STATS=1 and reports where
its lanes' wall time went; --check N compares the PAF rows of N sampled reads with the oracle's streaming map (the
checker, never the thing measured).  One JSON line on stdout."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sigmap_amd import synth  # noqa: E402

CHUNK = 4000


def _make_block(args):
    genome_lens, seed, n, n_chunks, junk_max_chunks, noise_share = args
    genome = synth.make_genome(7, genome_lens)
    lm = synth.make_pore_model()
    raw, _ = synth.simulate_reads(genome, lm, n, seed, n_chunks=n_chunks)
    rng = np.random.default_rng(seed + 1)
    kinds = np.zeros(n, np.uint8)
    out = []
    for i in range(n):
        r = raw[i]
        u = rng.random()
        if u < noise_share:                       # junk throughout
            r = rng.integers(350, 1000, size=len(r)).astype(np.int16)
            kinds[i] = 2
        elif i % 3 == 0:                          # junk first
            j = int(rng.integers(CHUNK, junk_max_chunks * CHUNK + 1)) + int(rng.integers(0, CHUNK))
            j = min(j, len(r) - 2 * CHUNK)
            r = np.concatenate([rng.integers(350, 1000, size=j).astype(np.int16), r])[:len(r)]
            kinds[i] = 1
        out.append(np.ascontiguousarray(r, dtype=np.int16))
    return out, kinds


def make_reads(genome_lens, n_reads, n_chunks, seed, workers):
    junk_max = max(1, n_chunks // 3)
    per = max(1, (n_reads + workers - 1) // workers)
    jobs = [(genome_lens, seed + 101 * k, min(per, n_reads - k * per), n_chunks, junk_max, 0.02)
            for k in range((n_reads + per - 1) // per)]
    reads, kinds = [], []
    with ProcessPoolExecutor(max_workers=workers) as ex:
        for r, k in ex.map(_make_block, jobs):
            reads.extend(r)
            kinds.append(k)
    return reads, np.concatenate(kinds)


def strip_mt(line):
    return "\t".join(f for f in line.rstrip("\n").split("\t") if not f.startswith("mt:f:"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10000)
    ap.add_argument("--chunks-per-read", type=int, default=8)
    ap.add_argument("--contigs", type=int, default=16)
    ap.add_argument("--contig-len", type=int, default=750000)
    ap.add_argument("--streams", type=int, default=0, help="device contexts (0: the CLI's default -- up to two, as the reads need)")
    ap.add_argument("--batch-chunks", type=int, default=32768)
    ap.add_argument("--max-num-chunks", type=int, default=30)
    ap.add_argument("--sdtw-band", type=int, default=0)
    ap.add_argument("--check", type=int, default=96, help="reads compared with the oracle's streaming map")
    ap.add_argument("--runs", type=int, default=3, help="CLI mapping runs (the median span is reported)")
    ap.add_argument("--skip-cold", action="store_true", help="no extra run with the CLI's --no-warmup")
    ap.add_argument("--workdir", default=None, help="keep / reuse the generated files here")
    ap.add_argument("--rocprof-dir", default=None, help="also one run under rocprofv3 --kernel-trace into this directory")
    a = ap.parse_args()
    exe = os.path.join(ROOT, "sigmap_amd", "bin", "sigmap")
    genome_lens = [a.contig_len] * a.contigs
    tmp = None
    d = a.workdir
    if d is None:
        tmp = tempfile.TemporaryDirectory()
        d = tmp.name
    os.makedirs(d, exist_ok=True)
    tag = "r%d_c%d" % (a.reads, a.chunks_per_read)
    rd = os.path.join(d, "reads_" + tag)
    t0 = time.perf_counter()
    workers = max(1, min(32, (os.cpu_count() or 1) // 2))
    reads, kinds = make_reads(genome_lens, a.reads, a.chunks_per_read, 4242, workers)
    if not os.path.exists(os.path.join(rd, "a.sgr")):
        os.makedirs(rd, exist_ok=True)
        synth.write_sgr(os.path.join(rd, "a.sgr"), reads)
    t_gen = time.perf_counter() - t0
    genome = synth.make_genome(7, genome_lens)
    lm = synth.make_pore_model()
    names = ["contig%d" % i for i in range(a.contigs)]
    if not os.path.exists(os.path.join(d, "idx.pt")):
        synth.write_fasta(os.path.join(d, "ref.fa"), genome, names)
        synth.write_pore_model(os.path.join(d, "model.txt"), lm)
        r = subprocess.run([exe, "-i", "-r", d + "/ref.fa", "-p", d + "/model.txt", "-o", d + "/idx"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    cmd = [exe, "-m", "-r", d + "/ref.fa", "-p", d + "/model.txt", "-x", d + "/idx", "-s", rd, "-o", d + "/out.paf",
           "--batch-chunks", str(a.batch_chunks), "--max-num-chunks", str(a.max_num_chunks)] + \
          (["--streams", str(a.streams)] if a.streams > 0 else [])
    if a.sdtw_band:
        cmd += ["--sdtw-band", str(a.sdtw_band)]
    env = dict(os.environ, SIGMAP_STREAM_STATS="1")
    runs = []
    for k in range(max(1, a.runs)):
        time.sleep(float(os.environ.get("CLI_E2E_PAUSE", "4")))     # (the driver scrubs a finished process's memory)
        t1 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, env=env)
        wall = time.perf_counter() - t1
        assert r.returncode == 0, r.stderr[-3000:]
        st = [l for l in r.stderr.splitlines() if l.startswith("STREAM_STATS ")]
        j = json.loads(st[-1][len("STREAM_STATS "):])
        j["process_wall_s"] = round(wall, 3)
        runs.append(j)
    # ... and ONE run without the CLI's warm-up round (--no-warmup): its span holds the contexts' first-batch costs, like
    # the reference's own "Finished mapping in" span, which starts right after loading (sigmap.cc:617)
    cold = None
    if not a.skip_cold:
        time.sleep(float(os.environ.get("CLI_E2E_PAUSE", "4")))
        r = subprocess.run(cmd + ["--no-warmup"], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        st = [l for l in r.stderr.splitlines() if l.startswith("STREAM_STATS ")]
        cold = json.loads(st[-1][len("STREAM_STATS "):])
        cold_paf = {}
        for line in open(d + "/out.paf"):
            cold_paf[line.split("\t", 1)[0]] = strip_mt(line)
    if os.environ.get("STREAM_ROUNDS_OUT"):    # one more run that logs every chunk round of every lane
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(env, SIGMAP_STREAM_STATS="2"))
        assert r.returncode == 0, r.stderr[-3000:]
        with open(os.environ["STREAM_ROUNDS_OUT"], "w") as fh:
            fh.write("\n".join(l for l in r.stderr.splitlines() if l.startswith("STREAM_")) + "\n")
    if a.rocprof_dir:
        os.makedirs(a.rocprof_dir, exist_ok=True)
        subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", a.rocprof_dir, "--"] + cmd,
                       capture_output=True, text=True, env=env)
    runs.sort(key=lambda j: j["span_s"])
    j = runs[len(runs) // 2]
    paf = {}
    for line in open(d + "/out.paf"):
        paf[line.split("\t", 1)[0]] = strip_mt(line)
    assert len(paf) == a.reads
    if cold is not None:
        assert cold_paf == paf, "the PAF of the run without warm-up differs"
    mapped = sum(1 for l in paf.values() if l.split("\t")[4] != "*")
    ls = j["lane_seconds"]
    lane_total = sum(ls.values())
    host_between = lane_total - ls["device"]
    out = {
        "metric": "streaming reads/s through sigmap -m (multi-chunk reads, early stop, chain carry-over)",
        "reads_per_s": round(a.reads / j["span_s"], 1),
        "chunks_per_s": round(j["chunks"] / j["span_s"], 1),
        "span_s": j["span_s"], "all_span_s": [r_["span_s"] for r_ in runs],
        # set-up the span does NOT hold: the CLI's discarded round of the job's own first chunks on every context
        "warmup_s": j.get("warmup_s"),
        "no_warmup": None if cold is None else {
            "span_s": cold["span_s"], "reads_per_s": round(a.reads / cold["span_s"], 1),
            "what": "one run with --no-warmup: the span starts right after loading like the reference's (sigmap.cc:617) and "
                    "holds every context's first-batch costs; same PAF"},
        "reads": a.reads, "chunks_per_read": a.chunks_per_read, "max_num_chunks": a.max_num_chunks,
        "read_kinds": {"clean": int((kinds == 0).sum()), "junk_prefixed": int((kinds == 1).sum()), "junk_only": int((kinds == 2).sum())},
        "mean_chunks_used": round(j["chunks"] / a.reads, 3),
        "mapped_fraction": round(mapped / a.reads, 4),
        "rounds": j["rounds"], "lanes": j["lanes"], "mean_batch": round(j["chunks"] / max(j["rounds"], 1), 1), "max_batch": j["max_batch"],
        "chunks_in_flight_per_lane": j["chunks_in_flight_per_lane"],
        "carried_anchors_uploaded": j["carried_anchors_uploaded"], "anchors_downloaded": j["anchors_downloaded"],
        "speculative_batches": j.get("speculative_batches"), "batches_mapped_twice": j.get("batches_mapped_twice"),
        "lane_seconds": ls,
        # a lane's wall time outside [first enqueue of a round, return of its sync]: stop rules, carry-over, admission
        "host_share_of_lane_time": round(host_between / max(lane_total, 1e-9), 4),
        "host_ms_per_round": round(1e3 * host_between / max(j["rounds"], 1), 3),
        "device_ms_per_round": round(1e3 * ls["device"] / max(j["rounds"], 1), 3),
        "process_wall_s": j["process_wall_s"], "generate_s": round(t_gen, 1),
        "cli": " ".join(os.path.basename(c) if c.startswith("/") else c for c in cmd[:1]) + " " + " ".join(cmd[11:]),
    }
    if a.check > 0:
        from oracle import pyoracle as po
        from sigmap_amd import pipeline
        ix = pipeline.HostIndex.build(genome, lm, threads=0)
        tree = po.OracleTree.from_host_index(ix)
        ref_lens = np.array(genome_lens, dtype=np.uint32)
        rng = np.random.default_rng(5)
        # every kind of read in the sample
        pick = np.concatenate([rng.choice(np.flatnonzero(kinds == k), size=min(int((kinds == k).sum()), n), replace=False)
                               for k, n in ((0, a.check // 2), (1, a.check // 2 - 4), (2, 4))])
        bad, used = 0, []
        for i in pick:
            pa = po.add_signal(reads[i], synth.DIGITISATION, synth.RANGE, synth.OFFSET)
            m = po.map_read(tree, ix.positions, pa, a.contigs, ref_lens, params={"max_num_chunks": a.max_num_chunks})
            used.append(m.chunks_used)
            want = strip_mt(po.format_paf(m, "read%d" % i, names, ref_lens))
            bad += int(paf["read%d" % i] != want)
        tree.close()
        out["oracle_check"] = {"reads": int(len(pick)), "mismatches": bad, "mean_chunks_used": round(float(np.mean(used)), 2),
                               "max_chunks_used": int(max(used))}
        assert bad == 0, out["oracle_check"]
    print(json.dumps(out), flush=True)
    if tmp:
        tmp.cleanup()


if __name__ == "__main__":
    main()

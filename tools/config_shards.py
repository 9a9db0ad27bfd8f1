#!/usr/bin/env python3
"""BASELINE configs[2] / configs[4] with the eight GPUs folded onto ONE: N one-chunk reads through `sigmap -m`
(a) as one GPU maps them (`--gpus 1`) and (b) through the CLI's multi-GPU path with eight "GPUs" that are all this
box's one device (`--gpus 8 --gpus-share-device`: contiguous read blocks per GPU; a lane set per GPU on
its own contexts; rows merged by (reference, read)).  The two PAFs must be identical -- reads are independent, so this
is what eight real GPUs produce, less the RCCL broadcast of the index (a one-GPU box cannot run that: the contexts
share the resident index by export / attach instead).  A sample of the reads is also mapped by the oracle.

    tools/config_shards.py [--reads 100000] [--contigs 16 --contig-len 750000] [--check 200]
    tools/config_shards.py --reads 200000 --contigs 1 --contig-len 250000000 --check 0      # configs[4]'s shape

Prints one JSON object.  Documentation, not a test: it needs minutes and a GPU."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sigmap_amd import synth  # noqa: E402


def strip_mt(line):
    return "\t".join(f for f in line.rstrip("\n").split("\t") if not f.startswith("mt:f:"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100000)
    ap.add_argument("--contigs", type=int, default=16)
    ap.add_argument("--contig-len", type=int, default=750000)
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--check", type=int, default=200, help="reads also mapped by the oracle (0: none)")
    ap.add_argument("--threads", type=int, default=32, help="host threads of the index build (sigmap -i -t)")
    ap.add_argument("--unshared-check", action="store_true",
                    help="a third run with the set walk's lane sharing switched off (development knob): the same PAF")
    a = ap.parse_args()
    exe = os.path.join(ROOT, "sigmap_amd", "bin", "sigmap")
    lm = synth.make_pore_model()
    genome = synth.make_genome(7, [a.contig_len] * a.contigs)
    names = ["contig%d" % i for i in range(a.contigs)]
    t0 = time.perf_counter()
    raw, _ = synth.simulate_reads(genome, lm, a.reads, 1000, n_chunks=1)
    t_gen = time.perf_counter() - t0
    out = {"workload": "%d one-chunk reads of 4000 samples vs %d x %d bases" % (a.reads, a.contigs, a.contig_len),
           "generate_s": round(t_gen, 1)}
    with tempfile.TemporaryDirectory() as d:
        synth.write_fasta(os.path.join(d, "ref.fa"), genome, names)
        synth.write_pore_model(os.path.join(d, "model.txt"), lm)
        os.makedirs(os.path.join(d, "reads"))
        synth.write_sgr(os.path.join(d, "reads", "a.sgr"), [r for r in raw])
        t0 = time.perf_counter()
        r = subprocess.run([exe, "-i", "-r", d + "/ref.fa", "-p", d + "/model.txt", "-o", d + "/idx", "-t", str(a.threads)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        out["index_s"] = round(time.perf_counter() - t0, 1)
        pafs = {}
        runs = [("gpus_1", ["--gpus", "1"], {}), ("gpus_%d_on_one_device" % a.gpus, ["--gpus", str(a.gpus), "--gpus-share-device"], {})]
        if a.unshared_check:  # the batches of these runs are small enough for the set walk to share a query's walk between lanes
            runs.append(("gpus_1_walks_not_shared", ["--gpus", "1"], {"SMHIP_DEV": "1", "SMHIP_WALK_SHARE": "0"}))
        for tag, extra, more_env in runs:
            time.sleep(float(os.environ.get("CLI_E2E_PAUSE", "8")))   # (the driver scrubs a finished process's memory)
            t1 = time.perf_counter()
            r = subprocess.run([exe, "-m", "-r", d + "/ref.fa", "-p", d + "/model.txt", "-x", d + "/idx", "-s", d + "/reads",
                                "-o", d + "/out_%s.paf" % tag] + extra, capture_output=True, text=True,
                               env=dict(os.environ, SIGMAP_STREAM_STATS="1", **more_env))
            assert r.returncode == 0, r.stderr[-2000:]
            wall = time.perf_counter() - t1
            span = [l for l in r.stderr.splitlines() if l.startswith("Finished mapping in")]
            stats = [l for l in r.stderr.splitlines() if l.startswith("STREAM_STATS ")]
            j = json.loads(stats[-1][len("STREAM_STATS "):]) if stats else {}
            pafs[tag] = [strip_mt(l) for l in open(d + "/out_%s.paf" % tag)]
            out[tag] = {"span_s": j.get("span_s"), "reads_per_s": round(a.reads / j["span_s"], 1) if j.get("span_s") else None,
                        "lanes": j.get("lanes"), "chunks_in_flight_per_lane": j.get("chunks_in_flight_per_lane"),
                        "process_wall_s": round(wall, 2), "cli": span[-1] if span else None,
                        "rows": len(pafs[tag]), "mapped": sum(1 for l in pafs[tag] if l.split("\t")[4] != "*")}
        tags = list(pafs)
        out["paf_identical"] = all(pafs[t] == pafs[tags[0]] for t in tags[1:])
        if a.check > 0:
            from oracle import pyoracle as po
            from sigmap_amd import pipeline
            ix = pipeline.HostIndex.load(d + "/idx")   # (the CLI's own index files)
            tree = po.OracleTree.from_host_index(ix)
            ref_lens = np.array([len(g) for g in genome], dtype=np.uint32)
            by_read = {l.split("\t", 1)[0]: l for l in pafs[tags[1]]}
            pick = np.linspace(0, a.reads - 1, a.check).astype(int)
            bad = 0
            for i in pick:
                pa = po.add_signal(raw[i], synth.DIGITISATION, synth.RANGE, synth.OFFSET)
                m = po.map_read(tree, ix.positions, pa, a.contigs, ref_lens)
                want = strip_mt(po.format_paf(m, "read%d" % i, names, ref_lens))
                bad += by_read.get("read%d" % i) != want
            tree.close()
            out["oracle_check"] = {"reads": int(len(pick)), "mismatches": int(bad)}
    print(json.dumps(out))
    return 0 if out["paf_identical"] and not (out.get("oracle_check") or {}).get("mismatches") else 1


if __name__ == "__main__":
    sys.exit(main())

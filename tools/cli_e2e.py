#!/usr/bin/env python3
"""End-to-end wall time of the CLI (`sigmap -i`, then `sigmap -m` over a .sgr read directory) on
the BASELINE configs[1] workload.  Documentation only; bench.py measures the
device path with inputs resident."""
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigmap_amd import synth  # noqa: E402


def main(reads=10000, n_chunks=1):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "sigmap_amd", "bin", "sigmap")
    lm = synth.make_pore_model()
    genome = synth.make_genome(7, [750000] * 16)
    raw, _ = synth.simulate_reads(genome, lm, reads, 1000, n_chunks=n_chunks)
    with tempfile.TemporaryDirectory() as d:
        synth.write_fasta(os.path.join(d, "ref.fa"), genome, ["contig%d" % i for i in range(16)])
        synth.write_pore_model(os.path.join(d, "model.txt"), lm)
        os.makedirs(os.path.join(d, "reads"))
        synth.write_sgr(os.path.join(d, "reads", "a.sgr"), [r for r in raw])
        t0 = time.perf_counter()
        r = subprocess.run([exe, "-i", "-r", d + "/ref.fa", "-p", d + "/model.txt", "-o", d + "/idx"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        print("index %.2f s" % (time.perf_counter() - t0))
        ref_paf = None
        for streams in (1, 2, 3):
            # the driver scrubs the memory a process frees, and the next process's first large hipMalloc waits for
            # that (tools/micro/malloc_cumulative.hip: one call of several seconds right after a 200 GB process):
            # leave it time, so that a run is not charged for its predecessor
            time.sleep(float(os.environ.get("CLI_E2E_PAUSE", "8")))
            t1 = time.perf_counter()
            r = subprocess.run([exe, "-m", "-r", d + "/ref.fa", "-p", d + "/model.txt", "-x", d + "/idx", "-s", d + "/reads",
                                "-o", d + "/out.paf", "--streams", str(streams)], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            t2 = time.perf_counter()
            paf = ["\t".join(f for f in l.split("\t") if not f.startswith("mt:f:")) for l in open(d + "/out.paf")]
            ref_paf = ref_paf or paf
            mapped = sum(1 for l in paf if "\t*\t" not in l)
            span = [l for l in r.stderr.splitlines() if l.startswith("Finished mapping in")]
            print("--streams %d: map %d reads x %d chunk(s): %.2f s wall (process start, index load, upload, map, PAF) "
                  "= %.0f reads/s; mapped %d; PAF identical to --streams 1: %s; %s"
                  % (streams, reads, n_chunks, t2 - t1, reads / (t2 - t1), mapped, paf == ref_paf, span[-1] if span else ""))
            print("  " + " | ".join(l for l in r.stderr.splitlines() if l.endswith("s.")))
            allocs = [l for l in r.stderr.splitlines() if l.startswith("[smhip] hipMalloc")]
            if allocs:  # SMHIP_TRACE_ALLOC=1
                ms = [float(l.split(":")[1].split()[0]) for l in allocs]
                slow = sorted(zip(ms, allocs), reverse=True)[:4]
                print("  %d device allocations, %.0f ms in total; slowest: %s" % (len(ms), sum(ms), "; ".join(a for _, a in slow)))

if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10000, int(sys.argv[2]) if len(sys.argv) > 2 else 1)

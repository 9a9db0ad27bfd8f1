#!/usr/bin/env python3
"""Thread scaling of the CPU baseline (oracle so_map_reads, OpenMP) on this box: chunks/s at 1..N threads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sigmap_amd import pipeline, synth
from oracle import pyoracle as po
genome = synth.make_genome(7, [750000] * 16); lm = synth.make_pore_model()
raw, _ = synth.simulate_reads(genome, lm, 2048, 1000)
ix = pipeline.HostIndex.build(genome, lm, threads=0)
pa = [po.add_signal(r, synth.DIGITISATION, synth.RANGE, synth.OFFSET) for r in raw]
tree = po.OracleTree.from_host_index(ix); rl = np.array([750000] * 16, np.uint32)
for nt in (1, 8, 16, 32, 64, 128, 256):
    if nt > (os.cpu_count() or 1): break
    n = min(len(pa), max(32, nt * 8))
    t0 = time.perf_counter(); po.map_reads(tree, ix.positions, pa[:n], 16, rl, nt); dt = time.perf_counter() - t0
    print("threads %3d: %7.1f chunks/s  (%5.1f per thread)" % (nt, n / dt, n / dt / nt), flush=True)

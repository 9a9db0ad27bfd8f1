#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path: every step uploads the batch's
samples from host memory (smhip_batch_upload) before mapping it.  Never bench.py's `value`."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigmap_amd import pipeline, synth  # noqa: E402


def main(reads=10000, steps=5):
    lm = synth.make_pore_model()
    genome = synth.make_genome(7, [750000] * 16)
    ix = pipeline.HostIndex.build(genome, lm, threads=0)
    dev = pipeline.DeviceMapper(0)
    dev.upload_index(ix)
    raw, _ = synth.simulate_reads(genome, lm, reads, seed=1000)
    pa = [pipeline.raw_to_pa(r, synth.DIGITISATION, synth.RANGE, synth.OFFSET) for r in raw]
    offsets = np.concatenate([[0], np.cumsum([len(x) for x in pa])]).astype(np.uint32)
    samples = np.concatenate(pa)
    ref_sig, ref_off = pipeline.reference_signals(genome, lm)
    dev.upload_reference(ref_sig, ref_off)

    def step(upload):
        if upload:
            dev.upload_batch(samples, offsets)
        dev.run(16)
        dev.refine(band=64, pad=32)
        dev.sync()
        dev.download()
        dev.refine_download()

    dev.upload_batch(samples, offsets)
    step(False)
    for upload in (False, True):
        t0 = time.perf_counter()
        for _ in range(steps):
            step(upload)
        dt = (time.perf_counter() - t0) / steps
        print("upload each step" if upload else "inputs resident ", "%.2f ms/step  %.0f chunks/s" % (dt * 1e3, reads / dt),
              "(%.0f MB of f32 samples per step)" % (samples.nbytes / 1e6))


if __name__ == "__main__":
    main()

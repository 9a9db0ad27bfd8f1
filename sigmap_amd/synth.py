"""Seeded synthetic inputs (SURVEY.md section 8d recipe; own code, numpy).

genome  : i.i.d. uniform ACGT contigs
model   : 4^k level means ~ N(90, 12.5) clipped to [55, 130], rounded to 4 decimals
reads   : uniform contig/position/strand; dwell 1+Poisson(7.9) samples per base;
          sample = level_mean + N(0, 1.5 pA); ADC = rint(pA/scale - offset) as int16 with
          digitisation 8192, range 1467.61, offset 10; exactly n_samples samples.
"""
import numpy as np

DIGITISATION = 8192.0
RANGE = 1467.61
OFFSET = 10.0

_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def make_genome(seed, contig_lens):
    rng = np.random.default_rng(seed)
    return [np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(n))].tobytes()
            for n in contig_lens]


def make_pore_model(seed=12345, k=6):
    rng = np.random.default_rng(seed)
    lv = np.clip(rng.normal(90.0, 12.5, size=4 ** k), 55.0, 130.0)
    return np.round(lv, 4).astype(np.float32)


def write_pore_model(path, level_means, k=6):
    with open(path, "w") as f:
        f.write("kmer\tlevel_mean\tlevel_stdv\tsd_mean\tsd_stdv\n")
        for h, lv in enumerate(level_means):
            kmer = "".join("ACGT"[(h >> (2 * (k - 1 - i))) & 3] for i in range(k))
            f.write(f"{kmer}\t{lv:.4f}\t1.5\t1.0\t0.3\n")


def write_fasta(path, genome, names=None):
    with open(path, "w") as f:
        for i, s in enumerate(genome):
            f.write(f">{names[i] if names else 'contig%d' % i}\n")
            t = s.decode()
            for j in range(0, len(t), 80):
                f.write(t[j:j + 80] + "\n")


def revcomp(seq: bytes) -> bytes:
    return _COMP[np.frombuffer(seq, dtype=np.uint8)][::-1].tobytes()


def _kmer_hashes(codes, k):
    h = np.zeros(len(codes) - k + 1, dtype=np.int64)
    for i in range(k):
        h = (h << 2) | codes[i:len(codes) - k + 1 + i]
    return h


_CODE = np.full(256, 0, dtype=np.int64)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i


def simulate_reads(genome, level_means, n_reads, seed, n_samples=4000, k=6, dwell_mean=7.9,
                   noise_sd=1.5, n_chunks=1):
    """Returns (raw int16 [n_reads, n_samples*n_chunks], truth structured array)."""
    rng = np.random.default_rng(seed)
    total = n_samples * n_chunks
    span = int(total / (1.0 + dwell_mean) * 1.6) + 64  # bases drawn per read (ample)
    lens = np.array([len(g) for g in genome], dtype=np.int64)
    usable = np.maximum(lens - span, 1)
    contig = rng.choice(len(genome), size=n_reads, p=usable / usable.sum())
    pos = (rng.random(n_reads) * usable[contig]).astype(np.int64)
    strand = rng.integers(0, 2, size=n_reads)  # 0 = +, 1 = -
    raw = np.empty((n_reads, total), dtype=np.int16)
    scale = np.float32(RANGE / DIGITISATION)
    garr = [np.frombuffer(g, dtype=np.uint8) for g in genome]
    for r in range(n_reads):
        sub = garr[contig[r]][pos[r]:pos[r] + span]
        if strand[r]:
            sub = _COMP[sub][::-1]
        lv = level_means[_kmer_hashes(_CODE[sub], k)]
        dwell = 1 + rng.poisson(dwell_mean, size=len(lv))
        sig = np.repeat(lv, dwell)[:total]
        if len(sig) < total:  # pathological short dwell draw: pad with the last level
            sig = np.concatenate([sig, np.full(total - len(sig), sig[-1], dtype=sig.dtype)])
        pa = sig.astype(np.float64) + rng.normal(0.0, noise_sd, size=total)
        raw[r] = np.rint(pa / float(scale) - OFFSET).astype(np.int16)
    truth = np.zeros(n_reads, dtype=[("contig", "<i4"), ("pos", "<i8"), ("strand", "<i1"), ("span", "<i4")])
    truth["contig"], truth["pos"], truth["strand"], truth["span"] = contig, pos, strand, span
    return raw, truth


def write_sgr(path, raw, ids=None, digitisation=DIGITISATION, rng=RANGE, offset=OFFSET):
    """This build's flat read container (sigmap_amd/host/signal_batch.h): "SGR1" | u32 n | per read
    { u32 id_len | id | f32 digitisation | f32 range | f32 offset | u32 n_raw | i16 raw[] }."""
    import struct
    with open(path, "wb") as f:
        f.write(b"SGR1")
        f.write(struct.pack("<I", len(raw)))
        for i, r in enumerate(raw):
            rid = (ids[i] if ids else "read%d" % i).encode()
            r = np.ascontiguousarray(r, dtype=np.int16)
            f.write(struct.pack("<I", len(rid)) + rid)
            f.write(struct.pack("<fffI", digitisation, rng, offset, len(r)))
            f.write(r.tobytes())


def write_fast5(path, raw, ids=None, multi_read=True, digitisation=DIGITISATION, rng=RANGE, offset=OFFSET):
    """One FAST5 file (multi- or single-read layout) through the host library's HDF5 writer."""
    import ctypes as C
    from . import _abi
    L = _abi.host_lib()
    if not L.smhost_have_fast5():
        raise RuntimeError("host library was built without HDF5")
    raws = [np.ascontiguousarray(r, dtype=np.int16) for r in raw]
    names = [(ids[i] if ids else "read%d" % i).encode() for i in range(len(raws))]
    idp = (C.c_char_p * len(raws))(*names)
    rp = (C.c_void_p * len(raws))(*[r.ctypes.data for r in raws])
    lens = np.array([len(r) for r in raws], dtype=np.uint32)
    if L.smhost_write_fast5(path.encode(), 1 if multi_read else 0, len(raws), idp, rp, _abi.ptr(lens),
                            digitisation, rng, offset) != 0:
        raise RuntimeError("smhost_write_fast5 failed for " + path)


def write_blow5(path, raw, ids=None, record_press=0, digitisation=DIGITISATION, rng=RANGE, offset=OFFSET, signal_press=0):
    """One BLOW5 file (binary SLOW5, file version 0.2.0, one read group, primary fields only) through the
    host library's writer.  record_press: 0 none, 1 zlib, 2 zstd; signal_press: 0 none, 1 svb-zd."""
    import ctypes as C
    from . import _abi
    L = _abi.host_lib()
    raws = [np.ascontiguousarray(r, dtype=np.int16) for r in raw]
    names = [(ids[i] if ids else "read%d" % i).encode() for i in range(len(raws))]
    idp = (C.c_char_p * len(raws))(*names)
    rp = (C.c_void_p * len(raws))(*[r.ctypes.data for r in raws])
    lens = np.array([len(r) for r in raws], dtype=np.uint32)
    if L.smhost_write_blow5(path.encode(), record_press, signal_press, len(raws), idp, rp, _abi.ptr(lens), digitisation, rng, offset) != 0:
        raise RuntimeError("smhost_write_blow5 failed for " + path)


def load_signals(path):
    """[(id, pA float32 array, digitisation, range, offset)] of every read under `path` (host loader)."""
    import ctypes as C
    from . import _abi
    L = _abi.host_lib()
    h = L.smhost_signals_load(path.encode())
    out = []
    try:
        for i in range(L.smhost_signals_count(h)):
            rid, pa, n = C.c_char_p(), C.POINTER(C.c_float)(), C.c_uint64()
            dg, rg, of = C.c_float(), C.c_float(), C.c_float()
            assert L.smhost_signals_get(h, i, C.byref(rid), C.byref(pa), C.byref(n), C.byref(dg), C.byref(rg), C.byref(of)) == 0
            vals = np.ctypeslib.as_array(pa, shape=(n.value,)).copy() if n.value else np.zeros(0, np.float32)
            out.append((rid.value.decode(), vals, dg.value, rg.value, of.value))
    finally:
        L.smhost_signals_free(h)
    return out

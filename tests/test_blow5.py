"""BLOW5 ingestion (signal_batch.cc:153-210).  No GPU needed.  slow5lib is not in the
image: the reader is checked against this build's own writers only -- the host library's, and an
independent few-line Python one below written from the same description of the layout -- never
against a file slow5lib produced."""
import os
import struct
import subprocess
import sys
import zlib

import numpy as np
import pytest

from conftest import ROOT
from oracle import pyoracle as po
from sigmap_amd import _abi, synth

COLUMNS = "#read_id\tread_group\tdigitisation\toffset\trange\tsampling_rate\tlen_raw_signal\traw_signal"


def _reads(n=5, seed=3):
    lm = synth.make_pore_model()
    genome = synth.make_genome(21, [60000])
    raw, _ = synth.simulate_reads(genome, lm, n, seed)
    return [np.ascontiguousarray(r) for r in raw]


def _py_svb_zd(r):
    """svb-zd of one read, written independently of the host library's encoder: u32 n | control bytes | data bytes;
    32-bit StreamVByte over the zig-zag codes of the sample differences (first against 0)."""
    x = np.ascontiguousarray(r, np.int16).astype(np.int64)
    d = np.diff(np.concatenate([[0], x]))
    v = ((d << 1) ^ (d >> 63)) & 0xFFFFFFFF
    nbytes = np.where(v < 1 << 8, 1, np.where(v < 1 << 16, 2, np.where(v < 1 << 24, 3, 4)))
    ctrl = bytearray((len(v) + 3) // 4)
    data = bytearray()
    for i, (val, nb) in enumerate(zip(v.tolist(), nbytes.tolist())):
        ctrl[i >> 2] |= (nb - 1) << (2 * (i & 3))
        data += int(val).to_bytes(4, "little")[:nb]
    return struct.pack("<I", len(v)) + bytes(ctrl) + bytes(data)


def _py_blow5(path, raws, ids, cal, press=0, version=(0, 2, 0), aux=b"", eof=True, nul_in_id=False, svb=False, groups=1):
    """A BLOW5 file written field by field (the layout in sigmap_amd/host/blow5_reader.cc's header): record
    compression at byte 9, the number of read groups in bytes 10-13, signal compression (file version >= 0.2.0) at
    byte 14."""
    head = b"BLOW5\x01" + bytes(version) + bytes([press]) + struct.pack("<I", groups)
    if version >= (0, 2, 0):
        head += bytes([1 if svb else 0])                  # signal compression
    head += b"\x00" * (64 - len(head))
    text = ("#slow5_version\t%d.%d.%d\n#num_read_groups\t1\n@run_id\tx\n" % version +
            "#char*\tuint32_t\tdouble\tdouble\tdouble\tdouble\tuint64_t\tint16_t*" + ("\tuint8_t" if aux else "") + "\n" +
            COLUMNS + ("\tchannel" if aux else "") + "\n").encode()
    with open(path, "wb") as f:
        f.write(head + struct.pack("<I", len(text)) + text)
        for rid, r in zip(ids, raws):
            rid_b = rid.encode() + (b"\x00" if nul_in_id else b"")
            rec = struct.pack("<H", len(rid_b)) + rid_b + struct.pack("<IddddQ", 0, cal[0], cal[2], cal[1], 4000.0, len(r))
            rec += (_py_svb_zd(r) if svb else np.ascontiguousarray(r, np.int16).tobytes()) + aux
            if press == 1:
                rec = zlib.compress(rec)
            f.write(struct.pack("<Q", len(rec)) + rec)
        if eof:
            f.write(b"5WOLB")


@pytest.mark.parametrize("press,svb", [(0, 0), (1, 0), (2, 0), (0, 1), (1, 1), (2, 1)])
def test_blow5_roundtrip_equals_sgr_and_oracle(tmp_path, press, svb):
    if press == 2 and not _abi.host_lib().smhost_have_zstd():
        pytest.skip("host library built without zstd")
    raws = _reads()
    raws[1] = np.concatenate([raws[1], np.array([3000, -2000, 100], np.int16)])  # samples outside (30, 200) pA
    raws.append(np.zeros(0, np.int16))                                           # an empty read
    ids = ["r%02d-%s" % (i, "abc"[i % 3]) for i in range(len(raws))]
    db, ds = tmp_path / "b5", tmp_path / "sg"
    os.makedirs(db / "sub"); os.makedirs(ds)
    raws[2] = np.array([-32768, 32767, -32768, 0, 1, -1, 255, -256, 127, 128] * 50, np.int16)   # extreme deltas (3-byte codes)
    synth.write_blow5(str(db / "sub" / "batch_0.blow5"), raws, ids, record_press=press, signal_press=svb)   # found in a sub-directory
    synth.write_sgr(str(ds / "batch_0.sgr"), raws, ids)
    a, b = synth.load_signals(str(db)), synth.load_signals(str(ds))
    assert [x[0] for x in a] == ids == [x[0] for x in b]                         # file order
    for (_, pa, dg, rg, of), (_, pb, _, _, _), r in zip(a, b, raws):
        assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32))
        exp = po.add_signal(r, synth.DIGITISATION, synth.RANGE, synth.OFFSET)    # signal_batch.cc:195-207
        assert np.array_equal(pa.view(np.uint32), exp.view(np.uint32))
        assert (dg, rg, of) == (np.float32(synth.DIGITISATION), np.float32(synth.RANGE), np.float32(synth.OFFSET))
    assert len(a[1][1]) == len(a[0][1]) and len(a[-1][1]) == 0
    if svb and press == 0:   # the delta coding pays off on a real-looking signal: about one byte and a quarter per sample
        assert os.path.getsize(db / "sub" / "batch_0.blow5") < 0.75 * os.path.getsize(ds / "batch_0.sgr")


@pytest.mark.parametrize("kw", [dict(), dict(press=1), dict(version=(0, 1, 0)), dict(aux=b"\x07"), dict(nul_in_id=True),
                                dict(version=(1, 0, 0), press=1, aux=b"\x01\x02"), dict(svb=True), dict(svb=True, press=1, aux=b"\x09\x08"),
                                dict(groups=256), dict(groups=1, version=(0, 1, 0), press=1), dict(svb=True, groups=3)])
def test_reader_on_files_written_field_by_field(tmp_path, kw):
    """The same reads through the independent Python writer: file versions 0.1.0 (no signal-compression
    byte) / 0.2.0 / 1.0.0, zlib records, svb-zd signals (slow5tools' default: zlib + svb-zd) with auxiliary fields
    behind the compressed signal, auxiliary fields after raw_signal (ignored, like the reference,
    which reads only the primary fields), a read_id length that counts the NUL, read-group counts whose low byte
    is 0 or 1 (the count sits where an earlier version of this reader looked for the signal compression)."""
    raws = _reads(3)
    ids = ["a", "bb", "ccc"]
    cal = (8192.0, 1402.882324, 6.0)                    # doubles in the file; narrowed to float by the reader
    p = str(tmp_path / "x.blow5")
    _py_blow5(p, raws, ids, cal, **kw)
    got = synth.load_signals(p)                         # a single file instead of a directory
    assert [g[0] for g in got] == ids
    for g, r in zip(got, raws):
        # signal_batch.cc:187-189: float digitisation = rec->digitisation ... (NOT FAST5's 6-digit text round trip)
        assert (g[2], g[3], g[4]) == (np.float32(8192.0), np.float32(1402.882324), np.float32(6.0))
        exp = po.add_signal(r, 8192.0, float(np.float32(1402.882324)), 6.0)
        assert np.array_equal(g[1].view(np.uint32), exp.view(np.uint32))


@pytest.mark.parametrize("svb", [0, 1])
def test_host_writer_and_python_writer_agree_byte_for_byte(tmp_path, svb):
    raws = _reads(2)
    a, b = str(tmp_path / "a.blow5"), str(tmp_path / "b.blow5")
    synth.write_blow5(a, raws, ["x", "y"], record_press=0, digitisation=8192.0, rng=1402.882324, offset=6.0, signal_press=svb)
    _py_blow5(b, raws, ["x", "y"], (8192.0, 1402.882324, 6.0), svb=bool(svb))
    A, B = open(a, "rb").read(), open(b, "rb").read()
    # identical 64-byte binary header, records and end marker; only the free-text attribute lines differ
    assert A[:64] == B[:64] and A[-5:] == B[-5:] == b"5WOLB"
    ha, hb = struct.unpack("<I", A[64:68])[0], struct.unpack("<I", B[64:68])[0]
    assert A[68 + ha:] == B[68 + hb:]
    assert A[68:68 + ha].decode().splitlines()[-1] == COLUMNS == B[68:68 + hb].decode().splitlines()[-1]


def _load_in_subprocess(path):
    code = "import sys; sys.path.insert(0, %r); from sigmap_amd import synth; print(len(synth.load_signals(%r)))" % (ROOT, path)
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)


def test_malformed_files_fail_loudly(tmp_path):
    """slow5_get_next returning anything but SLOW5_ERR_EOF is fatal in the reference
    (signal_batch.cc:170-173); unsupported compression is refused, not guessed at."""
    raws = _reads(2)
    ok = str(tmp_path / "ok.blow5")
    _py_blow5(ok, raws, ["a", "b"], (8192.0, 1467.61, 10.0))
    assert _load_in_subprocess(ok).stdout.strip() == "2"
    data = open(ok, "rb").read()
    cases = {
        "noeof.blow5": (data[:-5], "end-of-file marker"),
        "trunc.blow5": (data[:-400], "Truncated"),
        "magic.blow5": (b"SLOW5\x01" + data[6:], "Not a BLOW5"),
        "sigpress.blow5": (data[:14] + b"\x02" + data[15:], "signal compression"),
        "svbflag.blow5": (data[:14] + b"\x01" + data[15:], "Corrupt BLOW5 record"),   # raw samples are not an svb-zd stream
        "press9.blow5": (data[:9] + b"\x09" + data[10:], "unknown record compression"),
        "cols.blow5": (data.replace(b"digitisation\toffset", b"offset\tdigitisation"), "column line"),
    }
    for name, (blob, msg) in cases.items():
        p = str(tmp_path / name)
        open(p, "wb").write(blob)
        r = _load_in_subprocess(p)
        assert r.returncode != 0 and msg in r.stderr, (name, r.stderr)


def test_mixed_directory_order_fast5_then_blow5(tmp_path):
    """signal_batch.cc:46-66: a path containing ".fast5" is FAST5, else one containing ".blow5" is BLOW5;
    all FAST5 files are loaded before the BLOW5 files."""
    raws = _reads(4)
    if _abi.host_lib().smhost_have_fast5():
        synth.write_fast5(str(tmp_path / "z.fast5"), raws[:1], ["f0"], multi_read=True)
    synth.write_blow5(str(tmp_path / "b.blow5"), raws[1:3], ["b0", "b1"], record_press=1)
    synth.write_blow5(str(tmp_path / "a.blow5"), raws[3:], ["a0"])
    synth.write_sgr(str(tmp_path / "0.sgr"), raws[:1], ["s0"])
    want = (["f0"] if _abi.host_lib().smhost_have_fast5() else []) + ["a0", "b0", "b1", "s0"]
    assert [x[0] for x in synth.load_signals(str(tmp_path))] == want

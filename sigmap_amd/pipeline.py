"""Host-side driver objects over the C ABI: HostIndex (CPU index build / files) and
DeviceMapper (one GPU context: index upload, batch map).  Plumbing only."""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import ptr


class HostIndex:
    """smhost_index wrapper (Sigmap::ConstructIndex / SpatialIndex::{Load,Save})."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("host index construction failed")
        self._h = C.c_void_p(handle)
        v = _abi.IndexView()
        if _abi.host_lib().smhost_index_get(self._h, C.byref(v)) != 0:
            raise RuntimeError("smhost_index_get failed")
        self.dim, self.max_leaf, self.depth = v.dim, v.max_leaf, v.depth
        self.n_points, self.m_size, self.n_nodes = v.n_points, v.m_size, v.n_nodes
        as_np = np.ctypeslib.as_array
        self.values = as_np(v.values, shape=(self.n_points,))
        self.positions = as_np(v.positions, shape=(self.n_points,))
        self.vind = as_np(v.vind, shape=(self.m_size,))
        self.nodes = np.frombuffer(
            (C.c_char * (16 * self.n_nodes)).from_address(C.addressof(v.nodes.contents)),
            dtype=_abi.NODE_DTYPE)
        self.bbox = as_np(v.bbox, shape=(2 * self.dim,))

    @classmethod
    def build(cls, genome, level_means, k=6, dim=6, max_leaf=20, threads=0):
        L = _abi.host_lib()
        arr = (C.c_char_p * len(genome))(*genome)
        lens = np.array([len(g) for g in genome], dtype=np.uint32)
        lm = np.ascontiguousarray(level_means, dtype=np.float32)
        return cls(L.smhost_index_build(ptr(lm), k, arr, ptr(lens), len(genome), dim, max_leaf, threads))

    @classmethod
    def from_points(cls, positions, values, dim=6, max_leaf=20, threads=0):
        p = np.ascontiguousarray(positions, dtype=np.uint64)
        v = np.ascontiguousarray(values, dtype=np.float32)
        return cls(_abi.host_lib().smhost_index_from_points(ptr(p), ptr(v), len(v), dim, max_leaf, threads))

    @classmethod
    def load(cls, prefix):
        return cls(_abi.host_lib().smhost_index_load(prefix.encode()))

    def save(self, prefix):
        if _abi.host_lib().smhost_index_save(self._h, prefix.encode()) != 0:
            raise RuntimeError("smhost_index_save failed")

    def close(self):
        if self._h:
            _abi.host_lib().smhost_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def raw_to_pa(raw, digitisation, rng, offset):
    """SignalBatch::AddSignal conversion + (30,200) pA filter for one read."""
    raw = np.ascontiguousarray(raw, dtype=np.int16)
    out = np.empty(len(raw), dtype=np.float32)
    n = _abi.host_lib().smhost_raw_to_pa(ptr(raw), len(raw), digitisation, rng, offset, ptr(out))
    return out[:n]


def reference_signals(genome, level_means, k=6):
    """Z-scored expected signal of every contig strand (the sDTW target), concatenated in the order
    contig0 +, contig0 -, contig1 +, ... plus the 2*n+1 element offsets."""
    L = _abi.host_lib()
    lm = np.ascontiguousarray(level_means, dtype=np.float32)
    parts, offs = [], [0]
    for g in genome:
        for strand in (0, 1):
            out = np.zeros(len(g), dtype=np.float32)
            m = L.smhost_reference_signal(ptr(lm), k, g, len(g), strand, ptr(out))
            parts.append(out[:m].copy())
            offs.append(offs[-1] + int(m))
    return np.concatenate(parts), np.array(offs, dtype=np.uint64)


class PinnedArray:
    """numpy view of page-locked host memory (smhip_host_alloc): H2D copies from it are DMA at link
    rate and asynchronous.  Keep the object alive while a copy may be in flight."""

    def __init__(self, shape, dtype):
        self.L = _abi.hip_lib()
        self.dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * self.dtype.itemsize
        p = C.c_void_p()
        if self.L.smhip_host_alloc(C.byref(p), max(n, 1)) != 0 or not p.value:
            raise MemoryError("smhip_host_alloc failed")
        self._p = p
        buf = (C.c_char * max(n, 1)).from_address(p.value)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        if self._p is not None:
            self.array = None
            self.L.smhip_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def print_alignment(codes):
    """The reference's per-base CIGAR of a DTW path (host library, print_alignment fast_dtw.cc:160-209)."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    L = _abi.host_lib()
    n = L.smhost_print_alignment(ptr(codes), len(codes), None, 0)
    buf = C.create_string_buffer(int(n) + 1)
    L.smhost_print_alignment(ptr(codes), len(codes), buf, int(n) + 1)
    return buf.value.decode()


def default_params():
    p = _abi.Params()
    _abi.hip_lib().smhip_default_params(C.byref(p))
    return p


class DeviceMapper:
    """One GPU context (smhip_ctx).  Raises if the HIP library or a GPU is missing."""

    def __init__(self, device=0):
        self.L = _abi.hip_lib()
        h = C.c_void_p()
        rc = self.L.smhip_init(device, C.byref(h))
        if rc != 0:
            raise RuntimeError(f"smhip_init(device={device}) failed with {rc}: no usable MI355X GPU")
        self._h = h
        self.params = default_params()
        self.n_chunks = 0

    def _check(self, rc, what):
        if rc != 0:
            msg = self.L.smhip_last_error(self._h)
            raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def upload_index(self, ix: HostIndex):
        self._check(self.L.smhip_index_upload(self._h, ix.dim, ptr(ix.values), ptr(ix.positions),
                                              ix.n_points, ptr(ix.vind), ix.m_size, ptr(ix.nodes),
                                              ix.n_nodes, ptr(ix.bbox)), "smhip_index_upload")

    def export_index(self):
        d = _abi.IndexDev()
        self._check(self.L.smhip_index_export(self._h, C.byref(d)), "smhip_index_export")
        return d

    def export_copy(self, array_id, dst_ptr, nbytes):
        """Copy walker array `array_id` (order of multigpu._ARRAYS) into caller-owned device memory."""
        self._check(self.L.smhip_index_export_copy(self._h, array_id, C.c_void_p(dst_ptr), nbytes),
                    "smhip_index_export_copy")

    def bcast_loopback_to(self, other):
        """RCCL self-test on one GPU (smhip_index_bcast_loopback): `other` gets its own copy of this
        context's index through a one-rank communicator and the grouped ncclBroadcast calls."""
        self._check(self.L.smhip_index_bcast_loopback(self._h, other._h), "smhip_index_bcast_loopback")

    def attach_index(self, d):
        self._check(self.L.smhip_index_attach(self._h, C.byref(d)), "smhip_index_attach")

    def upload_batch(self, samples, offsets, query_start_offsets=None, carried=None, carried_offsets=None):
        samples = np.ascontiguousarray(samples, dtype=np.float32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        self.n_chunks = len(offsets) - 1
        qs = None if query_start_offsets is None else np.ascontiguousarray(query_start_offsets, dtype=np.uint32)
        ca = None if carried is None else np.ascontiguousarray(carried, dtype=_abi.CARRIED_DTYPE)
        co = None if carried_offsets is None else np.ascontiguousarray(carried_offsets, dtype=np.uint32)
        self._check(self.L.smhip_batch_upload(self._h, ptr(samples), ptr(offsets), self.n_chunks,
                                              ptr(qs), ptr(ca), ptr(co)), "smhip_batch_upload")

    def upload_features(self, features, offsets, query_start_offsets=None, carried=None, carried_offsets=None):
        """A batch of FEATURE signals (smhip_features_upload: GenerateChains' inputs); then run / sync / download."""
        features = np.ascontiguousarray(features, dtype=np.float32)
        if len(features) == 0:
            features = np.zeros(1, np.float32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        self.n_chunks = len(offsets) - 1
        qs = None if query_start_offsets is None else np.ascontiguousarray(query_start_offsets, dtype=np.uint32)
        ca = None if carried is None else np.ascontiguousarray(carried, dtype=_abi.CARRIED_DTYPE)
        co = None if carried_offsets is None else np.ascontiguousarray(carried_offsets, dtype=np.uint32)
        self._check(self.L.smhip_features_upload(self._h, ptr(features), ptr(offsets), self.n_chunks,
                                                 ptr(qs), ptr(ca), ptr(co)), "smhip_features_upload")

    def generate_chains(self, features, offsets, n_refs, with_anchors=False, **kw):
        """SpatialIndex::GenerateChains for a batch of feature signals (smhip_generate_chains' steps)."""
        self.upload_features(features, offsets, **kw)
        self.run(n_refs)
        self.sync()
        return self.download(with_anchors=with_anchors)

    def chain_anchors(self, anchors, anchor_offsets, n_refs, with_anchors=False):
        """The chaining stage alone on caller-provided anchors (smhip_chain_anchors)."""
        a = np.ascontiguousarray(anchors, dtype=_abi.CARRIED_DTYPE)
        if len(a) == 0:
            a = np.zeros(1, dtype=_abi.CARRIED_DTYPE)
        ao = np.ascontiguousarray(anchor_offsets, dtype=np.uint32)
        self.n_chunks = len(ao) - 1
        res = np.zeros(self.n_chunks, dtype=_abi.RESULT_DTYPE)
        self._check(self.L.smhip_chain_anchors(self._h, C.byref(self.params), n_refs, ptr(a), ptr(ao), self.n_chunks,
                                               ptr(res)), "smhip_chain_anchors")
        return self.download(with_anchors=True) if with_anchors else res

    def configure_reads(self, n_slots, keep_samples):
        """Resident read slots (SignalBatch::AddSignal on the device)."""
        self._check(self.L.smhip_reads_configure(self._h, n_slots, keep_samples), "smhip_reads_configure")

    def upload_raw(self, raw, raw_offsets, digitisation, rng, offset, slots=None, want_len=True):
        """raw: int16 ADC samples of all reads, concatenated (numpy array, ideally a PinnedArray's);
        per-read digitisation / range / offset arrays.  Returns the filtered length of every read
        (None with want_len=False: the call then only enqueues work)."""
        assert raw.dtype == np.int16 and raw.flags["C_CONTIGUOUS"]
        ro = np.ascontiguousarray(raw_offsets, dtype=np.uint64)
        n = len(ro) - 1
        dg = np.ascontiguousarray(np.broadcast_to(np.float32(digitisation), (n,)), dtype=np.float32)
        rg = np.ascontiguousarray(np.broadcast_to(np.float32(rng), (n,)), dtype=np.float32)
        of = np.ascontiguousarray(np.broadcast_to(np.float32(offset), (n,)), dtype=np.float32)
        sl = None if slots is None else np.ascontiguousarray(slots, dtype=np.uint32)
        out = np.zeros(n, np.uint32) if want_len else None
        self._check(self.L.smhip_reads_upload_raw(self._h, ptr(raw), ptr(ro), ptr(dg), ptr(rg), ptr(of), ptr(sl), n,
                                                  ptr(out)), "smhip_reads_upload_raw")
        return out

    def select(self, slots, chunk_index, chunk_size=4000, query_start_offsets=None, carried=None,
               carried_offsets=None):
        sl = np.ascontiguousarray(slots, dtype=np.uint32)
        ci = np.ascontiguousarray(chunk_index, dtype=np.uint32)
        self.n_chunks = len(sl)
        qs = None if query_start_offsets is None else np.ascontiguousarray(query_start_offsets, dtype=np.uint32)
        ca = None if carried is None else np.ascontiguousarray(carried, dtype=_abi.CARRIED_DTYPE)
        co = None if carried_offsets is None else np.ascontiguousarray(carried_offsets, dtype=np.uint32)
        self._check(self.L.smhip_batch_select(self._h, ptr(sl), ptr(ci), chunk_size, self.n_chunks, ptr(qs), ptr(ca),
                                              ptr(co)), "smhip_batch_select")

    def upload_pieces(self, pinned, begins, lens, digitisation, rng, offset, slots=None, expect=None):
        """Pieces pinned[begins[r] : begins[r] + lens[r]] of a PinnedArray's int16 samples, pulled over PCIe by the
        device, converted, filtered, piece r into slot slots[r] (smhip_reads_upload_pieces).  Asynchronous."""
        arr = pinned.array
        assert arr.dtype == np.int16 and arr.ndim == 1
        n = len(begins)
        base = arr.ctypes.data
        ptrs = (C.c_void_p * max(n, 1))(*[base + 2 * int(b) for b in begins])
        ln = np.ascontiguousarray(lens, dtype=np.uint32)
        dg = np.ascontiguousarray(np.broadcast_to(np.float32(digitisation), (n,)), dtype=np.float32)
        rg = np.ascontiguousarray(np.broadcast_to(np.float32(rng), (n,)), dtype=np.float32)
        of = np.ascontiguousarray(np.broadcast_to(np.float32(offset), (n,)), dtype=np.float32)
        sl = None if slots is None else np.ascontiguousarray(slots, dtype=np.uint32)
        ex = None if expect is None else np.ascontiguousarray(expect, dtype=np.uint32)
        self._check(self.L.smhip_reads_upload_pieces(self._h, ptrs, ptr(ln), ptr(dg), ptr(rg), ptr(of), ptr(sl), ptr(ex), n),
                    "smhip_reads_upload_pieces")

    def download_pa(self, slot, cap):
        pa = np.zeros(cap, np.float32)
        kept, total = C.c_uint32(), C.c_uint32()
        self._check(self.L.smhip_reads_download_pa(self._h, slot, ptr(pa), cap, C.byref(kept), C.byref(total)),
                    "smhip_reads_download_pa")
        return pa[:min(kept.value, cap)], kept.value, total.value

    def run(self, n_refs):
        self._check(self.L.smhip_map_run(self._h, C.byref(self.params), n_refs), "smhip_map_run")

    def sync(self):
        self._check(self.L.smhip_map_sync(self._h), "smhip_map_sync")

    def download(self, with_anchors=False):
        res = np.zeros(self.n_chunks, dtype=_abi.RESULT_DTYPE)
        if not with_anchors:
            self._check(self.L.smhip_map_download(self._h, ptr(res), None, 0, None), "smhip_map_download")
            return res
        cap = int(1 << 22)
        while True:
            anchors = np.zeros(cap, dtype=_abi.CARRIED_DTYPE)
            offs = np.zeros(self.n_chunks + 1, dtype=np.uint64)
            rc = self.L.smhip_map_download(self._h, ptr(res), ptr(anchors), cap, ptr(offs))
            if rc == -5 and cap < (1 << 30):
                cap *= 4
                continue
            self._check(rc, "smhip_map_download")
            return res, anchors[:int(offs[-1])], offs

    def download_anchors_of(self, chunks):
        """The chain anchors of the listed chunks of the batch (smhip_map_download_anchors_of): (anchors, offsets)."""
        ch = np.ascontiguousarray(chunks, dtype=np.uint32)
        offs = np.zeros(len(ch) + 1, dtype=np.uint64)
        anchors = np.zeros(1, dtype=_abi.CARRIED_DTYPE)
        rc = self.L.smhip_map_download_anchors_of(self._h, ptr(ch), len(ch), ptr(anchors), 0, ptr(offs))
        if rc == -5:
            anchors = np.zeros(int(offs[-1]), dtype=_abi.CARRIED_DTYPE)
            rc = self.L.smhip_map_download_anchors_of(self._h, ptr(ch), len(ch), ptr(anchors), len(anchors), ptr(offs))
        self._check(rc, "smhip_map_download_anchors_of")
        return anchors[:int(offs[-1])], offs

    def download_chains(self):
        """Every primary chain of every chunk (smhip_map_download_chains): (records, offsets)."""
        offs = np.zeros(self.n_chunks + 1, dtype=np.uint64)
        self._check(self.L.smhip_map_download_chains(self._h, None, 0, ptr(offs)), "smhip_map_download_chains")
        chains = np.zeros(max(int(offs[-1]), 1), dtype=_abi.CHAIN_DTYPE)
        self._check(self.L.smhip_map_download_chains(self._h, ptr(chains), len(chains), ptr(offs)), "smhip_map_download_chains")
        return chains[:int(offs[-1])], offs

    def map_chunks(self, samples, offsets, n_refs, **kw):
        self.upload_batch(samples, offsets, **kw)
        self.run(n_refs)
        self.sync()
        return self.download()

    def upload_reference(self, signals, offsets):
        """signals: concatenated z-scored expected signals (contig0 +, contig0 -, contig1 +, ...)."""
        signals = np.ascontiguousarray(signals, dtype=np.float32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._check(self.L.smhip_reference_upload(self._h, ptr(signals), ptr(offsets), (len(offsets) - 1) // 2),
                    "smhip_reference_upload")

    def share_reference(self, owner):
        """Borrow the reference signal another context of this GPU uploaded (smhip_reference_share)."""
        self._check(self.L.smhip_reference_share(self._h, owner._h), "smhip_reference_share")

    def mem_info(self):
        free, total = C.c_uint64(), C.c_uint64()
        self._check(self.L.smhip_mem_info(self._h, C.byref(free), C.byref(total)), "smhip_mem_info")
        return free.value, total.value

    def refine(self, band=64, pad=32):
        self._check(self.L.smhip_refine_run(self._h, band, pad), "smhip_refine_run")

    def refine_path(self, band=64, pad=32):
        self._check(self.L.smhip_refine_path_run(self._h, band, pad), "smhip_refine_path_run")

    def refine_path_download(self):
        """dist, end, start (0-based absolute target indices) and the list of path-code arrays."""
        n = self.n_chunks
        dist = np.zeros(n, np.float32)
        end = np.zeros(n, np.int64)
        start = np.zeros(n, np.int64)
        plen = np.zeros(n, np.uint32)
        poff = np.zeros(n + 1, np.uint64)
        # first call: sizes only (SMHIP_ECAP when there are codes to fetch); second: with the buffer
        rc = self.L.smhip_refine_path_download(self._h, ptr(dist), ptr(end), ptr(start), ptr(plen), ptr(poff), None, 0)
        self._check(rc, "smhip_refine_path_download")
        codes = np.zeros(max(int(poff[-1]), 1), np.uint8)
        self._check(self.L.smhip_refine_path_download(self._h, ptr(dist), ptr(end), ptr(start), ptr(plen), ptr(poff),
                                                      ptr(codes), len(codes)), "smhip_refine_path_download")
        return dist, end, start, [codes[int(poff[i]):int(poff[i]) + int(plen[i])].copy() for i in range(n)]

    def refine_download(self):
        dist = np.zeros(self.n_chunks, dtype=np.float32)
        end = np.zeros(self.n_chunks, dtype=np.int64)
        self._check(self.L.smhip_refine_download(self._h, ptr(dist), ptr(end)), "smhip_refine_download")
        return dist, end

    def cwt_features(self, samples, offsets, scale0=8.0, feat_stride=2736, want_cwt=False):
        """The reference's CWT feature path (smhip_cwt_features): per chunk the peak / valley values,
        their sample positions and count; optionally the CWT signal rows (n_chunks x 4096)."""
        samples = np.ascontiguousarray(samples, dtype=np.float32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        n = len(offsets) - 1
        feat = np.zeros((n, feat_stride), np.float32)
        pos = np.zeros((n, feat_stride), np.uint32)
        cnt = np.zeros(n, np.uint32)
        rows = np.zeros((n, 4096), np.float32) if want_cwt else None
        self._check(self.L.smhip_cwt_features(self._h, ptr(samples), ptr(offsets), n, scale0, feat_stride, ptr(feat),
                                              ptr(pos), ptr(cnt), ptr(rows) if want_cwt else None), "smhip_cwt_features")
        return feat, pos, cnt, rows

    def set_knob(self, name, value):
        """A development knob of this context (smhip_dev_knob; needs SMHIP_DEV=1 when the context was created)."""
        self._check(self.L.smhip_dev_knob(self._h, name.encode(), int(value)), "smhip_dev_knob")

    def resolved_params(self):
        """The parameters of the last run() with the automatic choices resolved (smhip_resolved_params)."""
        out = _abi.Params()
        self._check(self.L.smhip_resolved_params(self._h, C.byref(out)), "smhip_resolved_params")
        return out

    def index_census(self):
        """Ascending hit counts of the index census (smhip_index_census)."""
        n = C.c_uint32()
        out = np.zeros(4096, np.uint32)
        self._check(self.L.smhip_index_census(self._h, C.byref(self.params), ptr(out), len(out), C.byref(n)),
                    "smhip_index_census")
        return out[:n.value]

    def start_coef_sweep(self, lo_bits, hi_bits, radius):
        """Mismatches of the classifier's start-score factor against the reference's expression over the floats with
        bit patterns in [lo_bits, hi_bits) (smhip_debug_start_coef_sweep)."""
        out = C.c_uint64(0)
        self._check(self.L.smhip_debug_start_coef_sweep(self._h, lo_bits, hi_bits, C.c_float(radius), C.byref(out)),
                    "smhip_debug_start_coef_sweep")
        return out.value

    def div_window_sweep(self, lo_bits, hi_bits, w):
        """Mismatches of the t-statistics' x / w against the IEEE division over the floats with bit patterns in
        [lo_bits, hi_bits) (smhip_debug_div_window_sweep)."""
        out = C.c_uint64(0)
        self._check(self.L.smhip_debug_div_window_sweep(self._h, lo_bits, hi_bits, w, C.byref(out)),
                    "smhip_debug_div_window_sweep")
        return out.value

    def sort_stats(self):
        """[parts by 32-bit wave sort, by 64-bit wave sort, by workgroup sort, chunks on the exact partition route]."""
        out = np.zeros(4, dtype=np.uint32)
        self._check(self.L.smhip_debug_sort_stats(self._h, ptr(out)), "smhip_debug_sort_stats")
        return out

    def debug_counters(self):
        out = np.zeros(16, dtype=np.uint32)
        self._check(self.L.smhip_debug_counters(self._h, ptr(out)), "smhip_debug_counters")
        return out

    def timing(self, enable=True):
        self._check(self.L.smhip_timing_enable(self._h, 1 if enable else 0), "smhip_timing_enable")

    def timing_reset(self):
        self._check(self.L.smhip_timing_reset(self._h), "smhip_timing_reset")

    def timing_get(self, name):
        ms, n = C.c_double(), C.c_uint64()
        self._check(self.L.smhip_timing_get(self._h, name.encode(), C.byref(ms), C.byref(n)), "smhip_timing_get")
        return ms.value, n.value

    def events(self, samples, offsets, feat_stride=2736):
        samples = np.ascontiguousarray(samples, dtype=np.float32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        n = len(offsets) - 1
        feat = np.zeros((n, feat_stride), dtype=np.float32)
        stdv = np.zeros((n, feat_stride), dtype=np.float32)
        nf = np.zeros(n, dtype=np.uint32)
        ne = np.zeros(n, dtype=np.uint32)
        st = np.zeros(n, dtype=np.uint32)
        self._check(self.L.smhip_events(self._h, C.byref(self.params), ptr(samples), ptr(offsets), n,
                                        feat_stride, ptr(feat), ptr(stdv), ptr(nf), ptr(ne), ptr(st)),
                    "smhip_events")
        return feat, stdv, nf, ne, st

    def events_raw(self, n_chunks, feat_stride=2736, row_stride=4097):
        """Stage outputs of the last events() / run(): dict of n_peaks, peaks, start, length, mean, stdv
        (n_chunks x feat_stride) and ps, pss, t1, t2 rows (n_chunks x row_stride)."""
        o = {"n_peaks": np.zeros(n_chunks, np.uint32)}
        for k in ("peaks", "start", "length"):
            o[k] = np.zeros((n_chunks, feat_stride), np.uint32)
        for k in ("mean", "stdv"):
            o[k] = np.zeros((n_chunks, feat_stride), np.float32)
        for k in ("ps", "pss", "t1", "t2"):
            o[k] = np.zeros((n_chunks, row_stride), np.float32)
        self._check(self.L.smhip_events_raw(self._h, n_chunks, feat_stride, row_stride, ptr(o["n_peaks"]),
                                            ptr(o["peaks"]), ptr(o["start"]), ptr(o["length"]), ptr(o["mean"]),
                                            ptr(o["stdv"]), ptr(o["ps"]), ptr(o["pss"]), ptr(o["t1"]), ptr(o["t2"])),
                    "smhip_events_raw")
        return o

    def seed_lookup(self, queries, cap=512):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        n = q.shape[0]
        counts = np.zeros(n, dtype=np.uint32)
        idx = np.zeros((n, cap), dtype=np.uint64)
        dist = np.zeros((n, cap), dtype=np.float32)
        nodes = np.zeros(n, dtype=np.uint64)
        pts = np.zeros(n, dtype=np.uint64)
        self._check(self.L.smhip_seed_lookup(self._h, C.byref(self.params), ptr(q), n, cap, ptr(counts),
                                             ptr(idx), ptr(dist), ptr(nodes), ptr(pts)), "smhip_seed_lookup")
        return counts, idx, dist, nodes, pts

    def sdtw_banded(self, target, t_off, query, q_off, lo, hi):
        target = np.ascontiguousarray(target, dtype=np.float32)
        query = np.ascontiguousarray(query, dtype=np.float32)
        t_off = np.ascontiguousarray(t_off, dtype=np.uint64)
        q_off = np.ascontiguousarray(q_off, dtype=np.uint64)
        lo = np.ascontiguousarray(lo, dtype=np.uint32)
        hi = np.ascontiguousarray(hi, dtype=np.uint32)
        n = len(t_off) - 1
        dist = np.zeros(n, dtype=np.float32)
        end = np.zeros(n, dtype=np.int64)
        self._check(self.L.smhip_sdtw_banded(self._h, ptr(target), ptr(t_off), ptr(query), ptr(q_off),
                                             ptr(lo), ptr(hi), n, ptr(dist), ptr(end)), "smhip_sdtw_banded")
        return dist, end

    def dtw_path(self, target, t_off, query, q_off, lo, hi):
        """fast_dtw.cc::DTW over per-row windows (smhip_dtw_path): dist, end (0-based target), the list
        of path-code arrays (start -> end) and the 0-based target coordinate of each path's first cell."""
        target = np.ascontiguousarray(target, dtype=np.float32)
        query = np.ascontiguousarray(query, dtype=np.float32)
        t_off = np.ascontiguousarray(t_off, dtype=np.uint64)
        q_off = np.ascontiguousarray(q_off, dtype=np.uint64)
        lo = np.ascontiguousarray(lo, dtype=np.uint32)
        hi = np.ascontiguousarray(hi, dtype=np.uint32)
        n = len(t_off) - 1
        cap = (np.diff(t_off) + np.diff(q_off)).astype(np.uint64)
        p_off = np.concatenate([[0], np.cumsum(cap)]).astype(np.uint64)
        dist = np.zeros(n, np.float32)
        end = np.zeros(n, np.int64)
        codes = np.zeros(max(int(p_off[-1]), 1), np.uint8)
        plen = np.zeros(n, np.uint32)
        start = np.zeros(n, np.int64)
        self._check(self.L.smhip_dtw_path(self._h, ptr(target), ptr(t_off), ptr(query), ptr(q_off), ptr(lo), ptr(hi), n,
                                          ptr(dist), ptr(end), ptr(codes), ptr(p_off), ptr(plen), ptr(start)),
                    "smhip_dtw_path")
        paths = [codes[int(p_off[i]):int(p_off[i]) + int(plen[i])].copy() for i in range(n)]
        return dist, end, paths, start

    def close(self):
        if self._h:
            self.L.smhip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MapperPool:
    """`depth` DeviceMapper contexts on ONE GPU (one HIP stream each) that share one resident index:
    the first context owns the walker arrays, the others borrow them (smhip_index_export ->
    smhip_index_attach, no copy).  The reads of a run are independent (one taskloop over all reads in
    Sigmap::StreamingMap, sigmap.cc:618-666), so every batch is mapped on whichever context is
    free next, each context driven by its own host thread: the tail of one batch's tree walk and its sort /
    chain / sDTW kernels overlap the next batch's walk.  Results do not depend on the context."""

    def __init__(self, device=0, depth=2):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.devs = [DeviceMapper(device) for _ in range(depth)]

    @property
    def first(self):
        return self.devs[0]

    def share_index(self):
        """Call after the first context has its index (upload_index or attach_index)."""
        d = self.devs[0].export_index()
        for dv in self.devs[1:]:
            dv.attach_index(d)

    def for_each(self, fn):
        for dv in self.devs:
            fn(dv)

    def run_steps(self, n_steps, step_fn, stagger_s=0.0, static=False):
        """step_fn(dev, step_index) is called once for every step 0..n_steps-1: each context's thread
        takes the next step not yet started when it has finished its previous one (static=True: context k
        takes steps k, k + depth, ... -- a context then knows its next step, e.g. to prefetch its inputs).
        Returns the list of results in step order.  `stagger_s`: context k starts k * stagger_s late, so that the
        contexts do not all begin with the same (longest) kernel and then share the GPU for it."""
        import threading
        import time
        out = [None] * n_steps
        err = []
        lock = threading.Lock()
        nxt = [0]

        def work(k):
            try:
                if stagger_s > 0 and k > 0:
                    time.sleep(k * stagger_s)
                mine = k
                while not err:
                    if static:
                        i = mine
                        mine += len(self.devs)
                    else:
                        with lock:
                            i = nxt[0]
                            nxt[0] += 1
                    if i >= n_steps:
                        break
                    out[i] = step_fn(self.devs[k], i)
            except BaseException as e:  # re-raised on the caller's thread
                err.append(e)

        th = [threading.Thread(target=work, args=(k,)) for k in range(min(len(self.devs), n_steps))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if err:
            raise err[0]
        return out

    def close(self):
        for dv in reversed(self.devs):  # borrowers first
            dv.close()


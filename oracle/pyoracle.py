"""ctypes bindings for the CPU checker (oracle/libsigmap_oracle.so) and, when present, the
real-reference build (oracle/_ref/libsigmap_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by sigmap_amd/.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_PATH = os.environ.get("SM_ORACLE_LIB") or os.path.join(_HERE, "libsigmap_oracle.so")
REF_PATH = os.path.join(_HERE, "_ref", "libsigmap_ref.so")


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Event(C.Structure):
    _fields_ = [("start", C.c_uint64), ("length", C.c_uint64), ("mean", C.c_float), ("stdv", C.c_float)]


EVENT_DTYPE = np.dtype([("start", "<u8"), ("length", "<u8"), ("mean", "<f4"), ("stdv", "<f4")])


class SearchStats(C.Structure):
    _fields_ = [("nodes", C.c_uint64), ("leaves", C.c_uint64), ("points", C.c_uint64)]


class ChainStats(C.Structure):
    _fields_ = [("queries", C.c_uint64), ("hits", C.c_uint64), ("anchors", C.c_uint64),
                ("search", SearchStats)]


class Anchor(C.Structure):
    _fields_ = [("target_position", C.c_uint32), ("query_position", C.c_uint32), ("distance", C.c_float)]


ANCHOR_DTYPE = np.dtype([("target_position", "<u4"), ("query_position", "<u4"), ("distance", "<f4")])


class Chain(C.Structure):
    _fields_ = [("score", C.c_float), ("ref", C.c_uint32), ("start", C.c_uint32), ("end", C.c_uint32),
                ("n_anchors", C.c_uint32), ("mapq", C.c_uint8), ("direction", C.c_int),
                ("anchors", C.POINTER(Anchor))]


class Chains(C.Structure):
    _fields_ = [("v", C.POINTER(Chain)), ("n", C.c_size_t), ("cap", C.c_size_t)]


class MapParams(C.Structure):
    _fields_ = [("search_radius", C.c_float), ("step_size", C.c_int), ("max_num_chunks", C.c_int),
                ("stop_min_anchors", C.c_int), ("output_min_anchors", C.c_int),
                ("stop_ratio", C.c_float), ("output_ratio", C.c_float),
                ("stop_mean_ratio", C.c_float), ("output_mean_ratio", C.c_float)]


class Mapping(C.Structure):
    _fields_ = [("mapped", C.c_int), ("ref", C.c_uint32), ("read_length", C.c_uint32),
                ("read_start", C.c_uint32), ("read_end", C.c_uint32), ("frag_start", C.c_uint32),
                ("frag_length", C.c_uint32), ("mapq", C.c_uint8), ("direction", C.c_uint8),
                ("chunks_used", C.c_uint32), ("n_chains", C.c_uint32), ("cm", C.c_uint32),
                ("s1", C.c_float), ("s2", C.c_float), ("sm", C.c_float), ("ad", C.c_float),
                ("at", C.c_float), ("aq", C.c_float), ("stats", ChainStats)]


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(ORACLE_PATH)
        vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
        L.so_add_signal.restype = sz
        L.so_add_signal.argtypes = [vp, sz, C.c_float, C.c_float, C.c_float, vp]
        L.so_detect_events.restype = sz
        L.so_detect_events.argtypes = [vp, sz, vp, vp, C.POINTER(sz)]
        L.so_prefix_sums.argtypes = [vp, sz, vp, vp]
        L.so_tstat.argtypes = [vp, vp, sz, sz, vp]
        L.so_zscore.argtypes = [vp, sz, vp]
        L.so_generate_events.restype = sz
        L.so_generate_events.argtypes = [vp, sz, vp, vp]
        L.so_level_means.restype = sz
        L.so_level_means.argtypes = [vp, C.c_int, C.c_char_p, C.c_uint32, C.c_uint32, vp]
        L.so_reverse_complement.argtypes = [C.c_char_p, sz, C.c_char_p]
        L.so_masked_positions.argtypes = [C.c_int, C.c_float, C.c_uint32, C.POINTER(C.c_char_p),
                                          C.POINTER(C.c_char_p), vp, C.POINTER(vp), C.POINTER(vp)]
        L.so_point_cloud_one_direction.restype = sz
        L.so_point_cloud_one_direction.argtypes = [C.c_int, C.c_uint32, vp, vp, sz, C.c_int, vp, vp, sz]
        L.so_kdtree_build.restype = vp
        L.so_kdtree_build.argtypes = [vp, u64, C.c_int, C.c_int]
        L.so_kdtree_from_preorder.restype = vp
        L.so_kdtree_from_preorder.argtypes = [vp, u64, C.c_int, u64, vp, u64, vp, u64, vp]
        L.so_kdtree_free.argtypes = [vp]
        L.so_kdtree_save.argtypes = [vp, C.c_char_p]
        L.so_kdtree_load.restype = vp
        L.so_kdtree_load.argtypes = [C.c_char_p, vp, u64]
        L.so_points_save.argtypes = [C.c_char_p, C.c_int, C.c_int, vp, vp, u64]
        L.so_points_load.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]
        L.so_radius_search.restype = u64
        L.so_radius_search.argtypes = [vp, vp, C.c_float, vp, vp, u64, C.POINTER(SearchStats)]
        L.so_kdtree_depth.argtypes = [vp]
        L.so_chains_clear.argtypes = [C.POINTER(Chains)]
        L.so_generate_chains.argtypes = [vp, vp, vp, sz, C.c_uint32, C.c_int, C.c_float, sz,
                                         C.POINTER(Chains), C.POINTER(ChainStats)]
        L.so_generate_chains_ex.argtypes = [vp, vp, vp, sz, C.c_uint32, C.c_int, C.c_float, sz,
                                            C.POINTER(Chains), C.POINTER(ChainStats), u64]
        L.so_chain_sorted_anchors.argtypes = [vp, vp, sz, C.c_int, C.c_float, C.POINTER(Chains)]
        L.so_seed_positions.restype = sz
        L.so_seed_positions.argtypes = [sz, C.c_int, C.c_int, vp]
        L.so_default_params.argtypes = [C.POINTER(MapParams)]
        L.so_map_read.argtypes = [vp, vp, vp, sz, sz, vp, C.POINTER(MapParams), C.POINTER(Mapping)]
        L.so_map_reads.argtypes = [vp, vp, vp, vp, sz, sz, vp, C.POINTER(MapParams), vp, C.c_int]
        L.so_format_paf.argtypes = [C.POINTER(Mapping), C.c_char_p, C.c_char_p, C.c_uint32, C.c_char_p, sz]
        L.so_sdtw.restype = C.c_float
        L.so_sdtw.argtypes = [vp, sz, vp, sz, C.POINTER(C.c_int64)]
        L.so_sdtw_banded.restype = C.c_float
        L.so_sdtw_banded.argtypes = [vp, sz, vp, sz, vp, vp, C.POINTER(C.c_int64)]
        L.so_dtw_window.restype = C.c_float
        L.so_dtw_window.argtypes = [vp, sz, vp, sz, vp, vp, C.POINTER(C.c_int64), vp, C.POINTER(sz),
                                    C.POINTER(C.c_int64)]
        L.so_print_alignment.restype = sz
        L.so_print_alignment.argtypes = [vp, sz, C.c_char_p, sz]
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(REF_PATH)


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_PATH)
        vp, sz = C.c_void_p, C.c_size_t
        L.ref_detect_events.restype = C.c_long
        L.ref_detect_events.argtypes = [vp, sz, vp, vp, vp, vp, vp, C.POINTER(sz), vp, vp, vp, vp]
        L.ref_detect_events_entry.restype = C.c_long
        L.ref_detect_events_entry.argtypes = [vp, sz, vp, vp]
        L.ref_kdtree_build.restype = vp
        L.ref_kdtree_build.argtypes = [vp, vp, sz, C.c_int, C.c_int]
        L.ref_kdtree_save.argtypes = [vp, C.c_char_p]
        L.ref_kdtree_load.restype = vp
        L.ref_kdtree_load.argtypes = [C.c_char_p, vp, vp, sz, C.c_int, C.c_int]
        L.ref_kdtree_free.argtypes = [vp]
        L.ref_radius_search.restype = sz
        L.ref_radius_search.argtypes = [vp, vp, C.c_float, vp, vp, sz]
        L.ref_kdtree_dump.restype = sz
        L.ref_kdtree_dump.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
        L.ref_level_means.restype = C.c_long
        L.ref_level_means.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, vp, C.POINTER(C.c_int)]
        u32 = C.c_uint32
        L.ref_fasta_load.restype = vp
        L.ref_fasta_load.argtypes = [C.c_char_p]
        L.ref_fasta_count.restype = u32
        L.ref_fasta_count.argtypes = [vp]
        L.ref_fasta_num_bases.restype = C.c_uint64
        L.ref_fasta_num_bases.argtypes = [vp]
        L.ref_fasta_get.argtypes = [vp, u32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(u32),
                                    C.POINTER(C.c_char_p)]
        L.ref_fasta_free.argtypes = [vp]
        L.ref_fasta_free.restype = None
        L.ref_paf_mapped.restype = C.c_long
        L.ref_paf_mapped.argtypes = [vp, u32, u32, C.c_char_p] + [u32] * 7 + [C.c_char_p, C.c_char_p, C.c_char_p, sz]
        L.ref_paf_unmapped.restype = C.c_long
        L.ref_paf_unmapped.argtypes = [vp, u32, u32, C.c_char_p, u32, u32, C.c_char_p, C.c_char_p, C.c_char_p, sz]
        L.ref_to_string_float.argtypes = [C.c_float, C.c_char_p, sz]
        L.ref_to_string_double.argtypes = [C.c_double, C.c_char_p, sz]
        _ref = L
    return _ref


# ---------------------------------------------------------------- convenience wrappers

def add_signal(raw, digitisation, rng, offset):
    """SignalBatch::AddSignal (A1): int16 ADC counts -> filtered pA."""
    raw = np.ascontiguousarray(raw, dtype=np.int16)
    out = np.empty(len(raw), dtype=np.float32)
    n = lib().so_add_signal(_p(raw), len(raw), digitisation, rng, offset, _p(out))
    return out[:n].copy()


def detect_events(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    ev = np.zeros(len(x) + 1, dtype=EVENT_DTYPE)
    peaks = np.zeros(len(x) + 1, dtype=np.uint64)
    npk = C.c_size_t()
    n = lib().so_detect_events(_p(x), len(x), _p(ev), _p(peaks), C.byref(npk))
    return ev[:n], peaks[:npk.value]


def generate_events(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    feat = np.zeros(len(x) + 1, dtype=np.float32)
    stdv = np.zeros(len(x) + 1, dtype=np.float32)
    n = lib().so_generate_events(_p(x), len(x), _p(feat), _p(stdv))
    return feat[:n].copy(), stdv[:n].copy()


class OracleTree:
    def __init__(self, values, dim=6, leaf_max=20, handle=None):
        self.values = np.ascontiguousarray(values, dtype=np.float32)
        self.h = handle if handle is not None else lib().so_kdtree_build(_p(self.values), len(self.values), dim, leaf_max)
        if not self.h:
            raise RuntimeError("so_kdtree_build failed")
        self.dim = dim

    @classmethod
    def load(cls, path, values, dim=6):
        values = np.ascontiguousarray(values, dtype=np.float32)
        h = lib().so_kdtree_load(path.encode(), _p(values), len(values))
        return cls(values, dim=dim, handle=h)

    @classmethod
    def from_host_index(cls, ix):
        """The oracle's tree over the arrays of a host index (pipeline.HostIndex): the content of its
        .si file handed over in memory (so_kdtree_from_preorder)."""
        values = np.ascontiguousarray(ix.values, dtype=np.float32)
        vind = np.ascontiguousarray(ix.vind, dtype=np.uint32)
        nodes = np.ascontiguousarray(ix.nodes)
        bbox = np.ascontiguousarray(ix.bbox, dtype=np.float32)
        h = lib().so_kdtree_from_preorder(_p(values), len(values), ix.dim, ix.max_leaf, _p(vind), len(vind),
                                          _p(nodes), len(nodes), _p(bbox))
        t = cls(values, dim=ix.dim, handle=h)
        t._keep = (vind, nodes, bbox)
        return t

    def save(self, path):
        assert lib().so_kdtree_save(self.h, path.encode()) == 0

    def depth(self):
        return lib().so_kdtree_depth(self.h)

    def radius_search(self, q, radius, cap=1 << 16):
        q = np.ascontiguousarray(q, dtype=np.float32)
        idx = np.zeros(cap, dtype=np.uint64)
        dist = np.zeros(cap, dtype=np.float32)
        st = SearchStats()
        n = lib().so_radius_search(self.h, _p(q), radius, _p(idx), _p(dist), cap, C.byref(st))
        m = min(n, cap)
        return n, idx[:m].copy(), dist[:m].copy(), (st.nodes, st.leaves, st.points)

    def close(self):
        if self.h:
            lib().so_kdtree_free(self.h)
            self.h = None


def chains_to_list(ch):
    out = []
    for i in range(ch.n):
        c = ch.v[i]
        anchors = np.array([(c.anchors[k].target_position, c.anchors[k].query_position, c.anchors[k].distance)
                            for k in range(c.n_anchors)], dtype=ANCHOR_DTYPE)
        out.append(dict(score=c.score, ref=c.ref, start=c.start, end=c.end, n_anchors=c.n_anchors,
                        mapq=c.mapq, direction=c.direction, anchors=anchors))
    return out


def generate_chains(tree, positions, query, n_refs, query_start_offset=0, step=2, radius=0.08, prev=None,
                    num_nearest_points=5000):
    """prev: list of chain dicts (as returned) to carry over; returns (chains, stats)."""
    query = np.ascontiguousarray(query, dtype=np.float32)
    positions = np.ascontiguousarray(positions, dtype=np.uint64)
    ch = Chains()
    keep = []
    if prev:
        arr = (Chain * len(prev))()
        for i, c in enumerate(prev):
            a = (Anchor * len(c["anchors"]))()
            for k, an in enumerate(c["anchors"]):
                a[k] = Anchor(int(an["target_position"]), int(an["query_position"]), float(an["distance"]))
            keep.append(a)
            arr[i] = Chain(c["score"], c["ref"], c["start"], c["end"], c["n_anchors"], c["mapq"], c["direction"],
                           C.cast(a, C.POINTER(Anchor)))
        raise NotImplementedError("carry-over through python needs malloc'd anchors; use map_read")
    st = ChainStats()
    lib().so_generate_chains_ex(tree.h, _p(positions), _p(query), len(query), query_start_offset, step,
                                radius, n_refs, C.byref(ch), C.byref(st), num_nearest_points)
    out = chains_to_list(ch)
    lib().so_chains_clear(C.byref(ch))
    return out, st


def chain_anchors(ref_strand, target, query, distance, n_refs, radius=0.08, dim=6):
    """The chaining stage alone (so_chain_sorted_anchors, spatial_index.cc:411-576) on one chunk's anchors given as
    four parallel arrays (ref_strand = ref << 1 | strand, strand 0 = +): per-bucket sort by SignalAnchor::operator<
    (target, query, distance; spatial_index.h:22-25), then DP, traceback, primary chains, MAPQ."""
    ref_strand = np.asarray(ref_strand, np.uint32); target = np.asarray(target, np.uint32)
    query = np.asarray(query, np.uint32); distance = np.asarray(distance, np.float32)
    groups, ptrs, counts = [], (C.c_void_p * (2 * n_refs))(), (C.c_size_t * (2 * n_refs))()
    for strand in range(2):
        for ref in range(n_refs):
            sel = np.flatnonzero(ref_strand == ((ref << 1) | strand))
            order = sel[np.lexsort((distance[sel], query[sel], target[sel]))]
            a = np.zeros(len(order), dtype=ANCHOR_DTYPE)
            a["target_position"], a["query_position"], a["distance"] = target[order], query[order], distance[order]
            groups.append(a)
            g = strand * n_refs + ref
            ptrs[g] = a.ctypes.data if len(a) else None
            counts[g] = len(a)
    ch = Chains()
    lib().so_chain_sorted_anchors(ptrs, counts, n_refs, dim, radius, C.byref(ch))
    out = chains_to_list(ch)
    lib().so_chains_clear(C.byref(ch))
    return out


def map_read(tree, positions, signal_pa, n_refs, ref_lens, params=None):
    signal_pa = np.ascontiguousarray(signal_pa, dtype=np.float32)
    positions = np.ascontiguousarray(positions, dtype=np.uint64)
    ref_lens = np.ascontiguousarray(ref_lens, dtype=np.uint32)
    p = MapParams()
    lib().so_default_params(C.byref(p))
    if params:
        for k, v in params.items():
            setattr(p, k, v)
    m = Mapping()
    lib().so_map_read(tree.h, _p(positions), _p(signal_pa), len(signal_pa), n_refs, _p(ref_lens),
                      C.byref(p), C.byref(m))
    return m


def map_reads(tree, positions, pa_list, n_refs, ref_lens, n_threads):
    """so_map_reads: the reads of pa_list on n_threads OpenMP threads; returns the Mapping array."""
    flat = np.ascontiguousarray(np.concatenate(pa_list), dtype=np.float32)
    offs = np.concatenate([[0], np.cumsum([len(x) for x in pa_list])]).astype(np.uint64)
    positions = np.ascontiguousarray(positions, dtype=np.uint64)
    ref_lens = np.ascontiguousarray(ref_lens, dtype=np.uint32)
    p = MapParams()
    lib().so_default_params(C.byref(p))
    out = (Mapping * len(pa_list))()
    lib().so_map_reads(tree.h, _p(positions), _p(flat), _p(offs), len(pa_list), n_refs, _p(ref_lens), C.byref(p),
                       C.cast(out, C.c_void_p), n_threads)
    return out


def format_paf(m, read_name, ref_names, ref_lens):
    buf = C.create_string_buffer(2048)
    lib().so_format_paf(C.byref(m), read_name.encode(), ref_names[m.ref].encode(), int(ref_lens[m.ref]), buf, 2048)
    return buf.value.decode()


def sdtw(target, query):
    t = np.ascontiguousarray(target, dtype=np.float32)
    q = np.ascontiguousarray(query, dtype=np.float32)
    end = C.c_int64()
    d = lib().so_sdtw(_p(t), len(t), _p(q), len(q), C.byref(end))
    return d, end.value


def sdtw_banded(target, query, lo, hi):
    t = np.ascontiguousarray(target, dtype=np.float32)
    q = np.ascontiguousarray(query, dtype=np.float32)
    lo = np.ascontiguousarray(lo, dtype=np.uint32)
    hi = np.ascontiguousarray(hi, dtype=np.uint32)
    end = C.c_int64()
    d = lib().so_sdtw_banded(_p(t), len(t), _p(q), len(q), _p(lo), _p(hi), C.byref(end))
    return d, end.value


def dtw_window(target, query, lo, hi):
    """fast_dtw.cc::DTW over per-row windows + generate_path: (dist, end target 0-based, path codes
    start -> end, start target 0-based)."""
    t = np.ascontiguousarray(target, dtype=np.float32)
    q = np.ascontiguousarray(query, dtype=np.float32)
    lo = np.ascontiguousarray(lo, dtype=np.uint32)
    hi = np.ascontiguousarray(hi, dtype=np.uint32)
    end, st, n = C.c_int64(), C.c_int64(), C.c_size_t()
    codes = np.zeros(len(t) + len(q) + 1, np.uint8)
    d = lib().so_dtw_window(_p(t), len(t), _p(q), len(q), _p(lo), _p(hi), C.byref(end), _p(codes), C.byref(n),
                            C.byref(st))
    return d, end.value, codes[:n.value].copy(), st.value


def print_alignment(codes):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    n = lib().so_print_alignment(_p(codes), len(codes), None, 0)
    buf = C.create_string_buffer(int(n) + 1)
    lib().so_print_alignment(_p(codes), len(codes), buf, int(n) + 1)
    return buf.value.decode()


# ---- CWT feature path (D1-D3)
def mad_normalize(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros(len(x), np.float32)
    lib().so_mad_normalize.restype = C.c_float
    mad = lib().so_mad_normalize(_p(x), C.c_size_t(len(x)), _p(out))
    return out, mad


def cwt_dog(x, scale0):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros(len(x), np.float32)
    lib().so_cwt_dog(_p(x), C.c_size_t(len(x)), C.c_float(scale0), _p(out))
    return out


def ref_cwt_dog(x, scale0):
    """The REAL reference cwt.cc (oracle/_ref), when it was built."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros(len(x), np.float32)
    ref().ref_cwt_dog(_p(x), C.c_size_t(len(x)), C.c_float(scale0), _p(out))
    return out


def generate_peaks(x, selective):
    x = np.ascontiguousarray(x, dtype=np.float32)
    pk, pos = np.zeros(len(x), np.float32), np.zeros(len(x), np.uint32)
    lib().so_generate_peaks.restype = C.c_size_t
    n = lib().so_generate_peaks(_p(x), C.c_size_t(len(x)), C.c_float(selective), _p(pk), _p(pos))
    return pk[:n].copy(), pos[:n].copy()


def cwt_features(x, scale0):
    x = np.ascontiguousarray(x, dtype=np.float32)
    pk, pos, cw = np.zeros(len(x), np.float32), np.zeros(len(x), np.uint32), np.zeros(len(x), np.float32)
    lib().so_cwt_features.restype = C.c_size_t
    n = lib().so_cwt_features(_p(x), C.c_size_t(len(x)), C.c_float(scale0), _p(pk), _p(pos), _p(cw))
    return pk[:n].copy(), pos[:n].copy(), cw

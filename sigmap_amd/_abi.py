"""ctypes bindings for include/sigmap_hip.h and include/sigmap_host.h."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
# (SMHIP_LIB: an alternative build of the device library, for A/B experiments)
HIP_LIB_PATH = os.environ.get("SMHIP_LIB") or os.path.join(LIB_DIR, "libsigmap_hip.so")
# (SMHOST_LIB: an alternative build of the host library, e.g. the ASan/UBSan one of tools/sanitize_cpu.sh)
HOST_LIB_PATH = os.environ.get("SMHOST_LIB") or os.path.join(LIB_DIR, "libsigmap_host.so")

SMHIP_TOPK = 4
SEARCH_AUTO, SEARCH_EXACT, SEARCH_SET = 0, 1, 2   # smhip_params.search_mode

# every symbol include/sigmap_hip.h declares (checked by tests/test_abi.py)
HIP_SYMBOLS = [
    "smhip_default_params", "smhip_init", "smhip_destroy", "smhip_last_error", "smhip_stream",
    "smhip_index_upload", "smhip_index_export", "smhip_index_attach", "smhip_batch_upload",
    "smhip_map_run", "smhip_map_sync", "smhip_map_download", "smhip_map_chunks",
    "smhip_reference_upload", "smhip_reference_share", "smhip_mem_info", "smhip_refine_run", "smhip_refine_download", "smhip_debug_counters", "smhip_timing_enable", "smhip_timing_reset", "smhip_timing_get", "smhip_events", "smhip_events_raw",
    "smhip_seed_lookup", "smhip_sdtw_banded", "smhip_cwt_features",
    "smhip_reads_configure", "smhip_reads_upload_raw", "smhip_batch_select", "smhip_reads_download_pa",
    "smhip_host_alloc", "smhip_host_free", "smhip_batch_reserve", "smhip_batch_reserve_for", "smhip_reads_upload_pieces", "smhip_map_download_anchors_of", "smhip_index_export_copy", "smhip_index_bcast", "smhip_index_bcast_loopback", "smhip_device_count", "smhip_dtw_path", "smhip_refine_path_run", "smhip_refine_path_download",
    "smhip_index_census", "smhip_dev_knob", "smhip_resolved_params", "smhip_features_upload", "smhip_generate_chains",
    "smhip_chain_anchors", "smhip_map_download_chains", "smhip_debug_sort_stats",
    "smhip_debug_start_coef_sweep", "smhip_debug_div_window_sweep",
]
HOST_SYMBOLS = [
    "smhost_index_build", "smhost_index_from_points", "smhost_index_load", "smhost_index_save",
    "smhost_index_free", "smhost_index_get", "smhost_reference_signal", "smhost_level_means",
    "smhost_raw_to_pa", "smhost_default_map_params", "smhost_should_stop", "smhost_format_paf",
    "smhost_have_fast5", "smhost_signals_load", "smhost_signals_free", "smhost_signals_count",
    "smhost_signals_get", "smhost_write_fast5", "smhost_write_blow5", "smhost_have_zstd", "smhost_shard_range", "smhost_print_alignment",
    "smhost_format_paf_record", "smhost_sequences_load", "smhost_sequences_free", "smhost_sequences_count",
    "smhost_sequences_num_bases", "smhost_sequences_get", "smhost_index_bind_device", "smhost_generate_chains",
    "smhost_generate_events", "smhost_chunk_bounds", "smhost_chunk_budget", "smhost_contexts_for_reads",
]


class Node(C.Structure):
    _fields_ = [("feat", C.c_int32), ("a", C.c_uint32), ("b", C.c_uint32), ("child2", C.c_uint32)]


NODE_DTYPE = np.dtype([("feat", "<i4"), ("a", "<u4"), ("b", "<u4"), ("child2", "<u4")])


class Params(C.Structure):
    _fields_ = [
        ("window_length1", C.c_uint32), ("window_length2", C.c_uint32),
        ("threshold1", C.c_float), ("threshold2", C.c_float), ("peak_height", C.c_float),
        ("compress_delta", C.c_double), ("min_num_events", C.c_uint32),
        ("search_radius", C.c_float), ("step_size", C.c_int32),
        ("max_gap_length", C.c_int32), ("max_target_gap_length", C.c_int32),
        ("chaining_band_length", C.c_int32), ("max_num_skips", C.c_int32),
        ("min_num_anchors", C.c_int32), ("num_best_chains", C.c_int32),
        ("num_nearest_points", C.c_int32), ("min_chaining_score", C.c_float),
        ("hits_per_query_fast", C.c_uint32), ("walk_blocks_per_cu", C.c_uint32),
        ("reference_walk", C.c_uint32), ("search_mode", C.c_uint32), ("set_walk_max_over_cap", C.c_float),
    ]


class Chain(C.Structure):
    _fields_ = [("score", C.c_float), ("ref", C.c_uint32), ("start", C.c_uint32),
                ("end", C.c_uint32), ("n_anchors", C.c_uint32), ("mapq", C.c_uint32),
                ("direction", C.c_uint32), ("pad", C.c_uint32)]


class ChunkResult(C.Structure):
    _fields_ = [
        ("status", C.c_uint32), ("n_events", C.c_uint32), ("n_features", C.c_uint32),
        ("n_queries", C.c_uint32), ("n_anchors", C.c_uint32), ("n_chains", C.c_uint32),
        ("sum_score", C.c_float), ("pad0", C.c_uint32),
        ("nodes_visited", C.c_uint64), ("points_tested", C.c_uint64), ("hits_total", C.c_uint64),
        ("chains", Chain * SMHIP_TOPK),
        ("c0_sum_dist", C.c_float), ("c0_sum_tgap", C.c_float), ("c0_sum_qgap", C.c_float),
        ("c0_first_qpos", C.c_uint32), ("c0_last_qpos", C.c_uint32), ("pad1", C.c_uint32),
    ]


CHAIN_DTYPE = np.dtype([("score", "<f4"), ("ref", "<u4"), ("start", "<u4"), ("end", "<u4"),
                        ("n_anchors", "<u4"), ("mapq", "<u4"), ("direction", "<u4"), ("pad", "<u4")])
RESULT_DTYPE = np.dtype([
    ("status", "<u4"), ("n_events", "<u4"), ("n_features", "<u4"), ("n_queries", "<u4"),
    ("n_anchors", "<u4"), ("n_chains", "<u4"), ("sum_score", "<f4"), ("pad0", "<u4"),
    ("nodes_visited", "<u8"), ("points_tested", "<u8"), ("hits_total", "<u8"),
    ("chains", CHAIN_DTYPE, (SMHIP_TOPK,)),
    ("c0_sum_dist", "<f4"), ("c0_sum_tgap", "<f4"), ("c0_sum_qgap", "<f4"),
    ("c0_first_qpos", "<u4"), ("c0_last_qpos", "<u4"), ("pad1", "<u4"),
])
assert RESULT_DTYPE.itemsize == C.sizeof(ChunkResult)

CARRIED_DTYPE = np.dtype([("ref_strand", "<u4"), ("target_position", "<u4"),
                          ("query_position", "<u4"), ("distance", "<f4")])


class IndexDev(C.Structure):
    _fields_ = [("dim", C.c_int32), ("root_ref", C.c_uint32),
                ("n_nodes", C.c_uint64), ("n_leaves", C.c_uint64), ("m_size", C.c_uint64),
                ("nodes", C.c_void_p), ("nodes_bytes", C.c_uint64),
                ("leaf_tab", C.c_void_p), ("leaf_tab_bytes", C.c_uint64),
                ("leaf_pts", C.c_void_p), ("leaf_pts_bytes", C.c_uint64),
                ("leaf_vind", C.c_void_p), ("leaf_vind_bytes", C.c_uint64),
                ("bbox", C.c_void_p), ("bbox_bytes", C.c_uint64),
                ("quads", C.c_void_p), ("quads_bytes", C.c_uint64), ("n_quads", C.c_uint64)]


class IndexView(C.Structure):
    _fields_ = [("dim", C.c_int32), ("max_leaf", C.c_int32), ("depth", C.c_int32), ("pad", C.c_int32),
                ("n_points", C.c_uint64), ("m_size", C.c_uint64), ("n_nodes", C.c_uint64),
                ("values", C.POINTER(C.c_float)), ("positions", C.POINTER(C.c_uint64)),
                ("vind", C.POINTER(C.c_uint32)), ("nodes", C.POINTER(Node)),
                ("bbox", C.POINTER(C.c_float))]


class MapParams(C.Structure):
    _fields_ = [("search_radius", C.c_float), ("step_size", C.c_int32), ("max_num_chunks", C.c_int32),
                ("stop_min_anchors", C.c_int32), ("output_min_anchors", C.c_int32),
                ("stop_ratio", C.c_float), ("output_ratio", C.c_float),
                ("stop_mean_ratio", C.c_float), ("output_mean_ratio", C.c_float)]


_host = None
_hip = None


def host_lib():
    """libsigmap_host.so (pure CPU C++)."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(HOST_LIB_PATH)
        L.smhost_index_build.restype = C.c_void_p
        L.smhost_index_build.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.c_void_p,
                                         C.c_uint32, C.c_int, C.c_int, C.c_int]
        L.smhost_index_from_points.restype = C.c_void_p
        L.smhost_index_from_points.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int]
        L.smhost_index_load.restype = C.c_void_p
        L.smhost_index_load.argtypes = [C.c_char_p]
        L.smhost_index_save.argtypes = [C.c_void_p, C.c_char_p]
        L.smhost_index_free.argtypes = [C.c_void_p]
        L.smhost_index_get.argtypes = [C.c_void_p, C.POINTER(IndexView)]
        L.smhost_reference_signal.restype = C.c_uint64
        L.smhost_reference_signal.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_uint32, C.c_int, C.c_void_p]
        L.smhost_level_means.restype = C.c_uint64
        L.smhost_level_means.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_uint32, C.c_void_p]
        L.smhost_raw_to_pa.restype = C.c_uint32
        L.smhost_raw_to_pa.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.smhost_chunk_budget.restype = C.c_uint64
        L.smhost_chunk_budget.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_double, C.c_uint32]
        L.smhost_contexts_for_reads.restype = C.c_uint64
        L.smhost_contexts_for_reads.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.smhost_chunk_bounds.restype = C.c_uint32
        L.smhost_chunk_bounds.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32,
                                          C.c_void_p, C.c_uint32, C.c_void_p]
        L.smhost_signals_load.restype = C.c_void_p
        L.smhost_signals_load.argtypes = [C.c_char_p]
        L.smhost_signals_free.argtypes = [C.c_void_p]
        L.smhost_signals_free.restype = None
        L.smhost_signals_count.restype = C.c_uint64
        L.smhost_signals_count.argtypes = [C.c_void_p]
        L.smhost_signals_get.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_char_p), C.POINTER(C.POINTER(C.c_float)),
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                         C.POINTER(C.c_float)]
        L.smhost_write_fast5.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.POINTER(C.c_char_p),
                                         C.POINTER(C.c_void_p), C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.smhost_write_blow5.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_char_p),
                                         C.POINTER(C.c_void_p), C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.smhost_print_alignment.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_uint64]
        L.smhost_print_alignment.restype = C.c_int64
        L.smhost_shard_range.argtypes = [C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.smhost_shard_range.restype = None
        L.smhost_default_map_params.argtypes = [C.POINTER(MapParams)]
        L.smhost_should_stop.argtypes = [C.c_void_p, C.POINTER(MapParams)]
        L.smhost_format_paf.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.POINTER(C.c_char_p), C.c_void_p, C.c_uint32,
                                        C.POINTER(MapParams), C.c_char_p, C.c_size_t,
                                        C.POINTER(C.c_uint32)]
        L.smhost_format_paf_record.argtypes = [C.c_char_p] + [C.c_uint32] * 7 + [C.c_char_p, C.c_char_p, C.c_uint32,
                                                                                 C.c_char_p, C.c_size_t]
        L.smhost_sequences_load.restype = C.c_void_p
        L.smhost_sequences_load.argtypes = [C.c_char_p]
        L.smhost_sequences_free.argtypes = [C.c_void_p]
        L.smhost_sequences_free.restype = None
        L.smhost_sequences_count.argtypes = [C.c_void_p]
        L.smhost_sequences_count.restype = C.c_uint32
        L.smhost_sequences_num_bases.argtypes = [C.c_void_p]
        L.smhost_sequences_num_bases.restype = C.c_uint64
        L.smhost_sequences_get.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                           C.POINTER(C.c_uint32), C.POINTER(C.c_char_p)]
        L.smhost_index_bind_device.argtypes = [C.c_void_p, C.c_void_p]
        L.smhost_generate_chains.restype = C.c_int64
        L.smhost_generate_chains.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_float, C.c_uint32,
                                             C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                             C.c_uint64, C.POINTER(C.c_uint64)]
        L.smhost_generate_events.restype = C.c_int64
        L.smhost_generate_events.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p,
                                             C.c_void_p, C.c_uint32, C.c_uint32]
        _host = L
    return _host


def hip_lib():
    """libsigmap_hip.so (HIP kernels + C ABI).  Raises if it has not been built."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise RuntimeError(f"{HIP_LIB_PATH} missing: the HIP extension is required "
                               "(python -c 'import __graft_entry__ as g; g.build()')")
        L = C.CDLL(HIP_LIB_PATH)
        vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
        L.smhip_default_params.argtypes = [C.POINTER(Params)]
        L.smhip_default_params.restype = None
        L.smhip_init.argtypes = [C.c_int, C.POINTER(vp)]
        L.smhip_destroy.argtypes = [vp]
        L.smhip_destroy.restype = None
        L.smhip_last_error.argtypes = [vp]
        L.smhip_last_error.restype = C.c_char_p
        L.smhip_stream.argtypes = [vp]
        L.smhip_stream.restype = vp
        L.smhip_index_upload.argtypes = [vp, C.c_int, vp, vp, u64, vp, u64, vp, u64, vp]
        L.smhip_index_export.argtypes = [vp, C.POINTER(IndexDev)]
        L.smhip_index_attach.argtypes = [vp, C.POINTER(IndexDev)]
        L.smhip_batch_upload.argtypes = [vp, vp, vp, u32, vp, vp, vp]
        L.smhip_map_run.argtypes = [vp, C.POINTER(Params), u32]
        L.smhip_map_sync.argtypes = [vp]
        L.smhip_map_download.argtypes = [vp, vp, vp, u64, vp]
        L.smhip_map_chunks.argtypes = [vp, C.POINTER(Params), u32, vp, vp, u32, vp, vp, vp, vp]
        L.smhip_reference_upload.argtypes = [vp, vp, vp, u32]
        L.smhip_reference_share.argtypes = [vp, vp]
        L.smhip_mem_info.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
        L.smhip_refine_run.argtypes = [vp, u32, u32]
        L.smhip_refine_download.argtypes = [vp, vp, vp]
        L.smhip_debug_counters.argtypes = [vp, vp]
        L.smhip_cwt_features.argtypes = [vp, vp, vp, u32, C.c_float, u32, vp, vp, vp, vp]
        L.smhip_timing_enable.argtypes = [vp, C.c_int]
        L.smhip_timing_reset.argtypes = [vp]
        L.smhip_timing_get.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(u64)]
        L.smhip_events.argtypes = [vp, C.POINTER(Params), vp, vp, u32, u32, vp, vp, vp, vp, vp]
        L.smhip_events_raw.argtypes = [vp, u32, u32, u32] + [vp] * 10
        L.smhip_reads_configure.argtypes = [vp, u32, u32]
        L.smhip_reads_upload_raw.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32, vp]
        L.smhip_batch_select.argtypes = [vp, vp, vp, u32, u32, vp, vp, vp]
        L.smhip_reads_download_pa.argtypes = [vp, u32, vp, u32, C.POINTER(u32), C.POINTER(u32)]
        L.smhip_batch_reserve.argtypes = [vp, u32, u32]
        L.smhip_batch_reserve_for.argtypes = [vp, vp, u32, u32]
        L.smhip_reads_upload_pieces.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, u32]
        L.smhip_map_download_anchors_of.argtypes = [vp, vp, u32, vp, C.c_uint64, vp]
        L.smhip_host_alloc.argtypes = [C.POINTER(vp), C.c_size_t]
        L.smhip_host_free.argtypes = [vp]
        L.smhip_host_free.restype = None
        L.smhip_index_export_copy.argtypes = [vp, C.c_int, vp, u64]
        L.smhip_index_bcast.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
        L.smhip_index_bcast_loopback.argtypes = [vp, vp]
        L.smhip_device_count.argtypes = []
        L.smhip_index_census.argtypes = [vp, C.POINTER(Params), vp, u32, C.POINTER(u32)]
        L.smhip_dev_knob.argtypes = [vp, C.c_char_p, C.c_longlong]
        L.smhip_resolved_params.argtypes = [vp, C.POINTER(Params)]
        L.smhip_features_upload.argtypes = [vp, vp, vp, u32, vp, vp, vp]
        L.smhip_generate_chains.argtypes = [vp, C.POINTER(Params), u32, vp, vp, u32, vp, vp, vp, vp]
        L.smhip_chain_anchors.argtypes = [vp, C.POINTER(Params), u32, vp, vp, u32, vp]
        L.smhip_map_download_chains.argtypes = [vp, vp, u64, vp]
        L.smhip_debug_sort_stats.argtypes = [vp, vp]
        L.smhip_debug_start_coef_sweep.argtypes = [vp, u32, u32, C.c_float, vp]
        L.smhip_debug_div_window_sweep.argtypes = [vp, u32, u32, u32, vp]
        L.smhip_refine_path_run.argtypes = [vp, u32, u32]
        L.smhip_refine_path_download.argtypes = [vp, vp, vp, vp, vp, vp, vp, u64]
        L.smhip_dtw_path.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32, vp, vp, vp, vp, vp, vp]
        L.smhip_seed_lookup.argtypes = [vp, C.POINTER(Params), vp, u32, u32, vp, vp, vp, vp, vp]
        L.smhip_sdtw_banded.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32, vp, vp]
        _hip = L
    return _hip


def ptr(a):
    """Raw pointer of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)

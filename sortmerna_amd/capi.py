"""ctypes binding of include/smr_hip.h (the C ABI of libsmr_hip.so)."""
import ctypes as C
import os

from . import build as _build


class Params(C.Structure):
    _fields_ = [
        ("skiplengths", C.c_uint32 * 3), ("num_seeds", C.c_int32), ("min_lis", C.c_int32), ("edges", C.c_int32),
        ("is_as_percent", C.c_int32), ("match", C.c_int32), ("mismatch", C.c_int32), ("score_N", C.c_int32),
        ("gap_open", C.c_int32), ("gap_ext", C.c_int32), ("num_alignments", C.c_uint32), ("is_best", C.c_int32),
        ("is_full_search", C.c_int32), ("is_forward", C.c_int32), ("is_reverse", C.c_int32), ("minoccur", C.c_uint32),
        ("minimal_score", C.c_uint32), ("index_num", C.c_uint32), ("part", C.c_uint32),
        ("is_last_index_part", C.c_int32),
    ]


class IndexInfo(C.Structure):
    _fields_ = [
        ("lnwin", C.c_uint32), ("n_kmers", C.c_uint32), ("trie_words", C.c_uint64), ("n_ids", C.c_uint32),
        ("n_pos", C.c_uint64), ("n_refs", C.c_uint32), ("ref_bytes", C.c_uint64), ("n_nodes", C.c_uint64),
        ("n_buckets", C.c_uint64), ("n_entries", C.c_uint64), ("bg", C.c_double * 4), ("full_len", C.c_uint64),
        ("numseq", C.c_uint64), ("n_parts", C.c_uint32),
    ]


class ReportOpts(C.Structure):
    _fields_ = [("fastx", C.c_int), ("other", C.c_int), ("blast_tabular", C.c_int), ("blast_cols", C.c_char * 64), ("sam", C.c_int),
                ("blast_pairwise", C.c_int), ("sam_sq", C.c_int), ("paired_in", C.c_int), ("paired_out", C.c_int), ("out2", C.c_int), ("sout", C.c_int), ("zip_out", C.c_int)]


class SummaryDb(C.Structure):
    _fields_ = [("ref_file", C.c_char_p), ("skiplengths", C.c_uint32 * 3), ("lam", C.c_double), ("K", C.c_double),
                ("minimal_score", C.c_uint32), ("reads_matched", C.c_uint64)]


class Summary(C.Structure):
    _fields_ = [("cmdline", C.c_char_p), ("pid", C.c_char_p), ("timestamp", C.c_char_p), ("seed_len", C.c_uint32),
                ("num_seeds", C.c_int32), ("edges", C.c_int32), ("match", C.c_int32), ("mismatch", C.c_int32), ("gap_open", C.c_int32),
                ("gap_ext", C.c_int32), ("score_N", C.c_int32), ("sam_sq", C.c_int32), ("threads", C.c_int32),
                ("reads_files", C.POINTER(C.c_char_p)), ("n_reads_files", C.c_uint32),
                ("total_reads", C.c_uint64), ("num_aligned", C.c_uint64), ("all_reads_len", C.c_uint64),
                ("min_read_len", C.c_uint32), ("max_read_len", C.c_uint32), ("dbs", C.POINTER(SummaryDb)), ("n_dbs", C.c_uint32)]


class Prof(C.Structure):
    _fields_ = [
        ("seed_ms", C.c_double), ("seed_launches", C.c_uint64), ("chain_ms", C.c_double), ("chain_launches", C.c_uint64),
        ("trace_ms", C.c_double), ("trace_launches", C.c_uint64), ("n_windows", C.c_uint64), ("n_lookup", C.c_uint64),
        ("n_node", C.c_uint64), ("n_entry", C.c_uint64), ("n_hit", C.c_uint64), ("n_read_bytes", C.c_uint64),
        ("n_sw_fwd", C.c_uint64), ("n_sw_rev", C.c_uint64), ("n_sw_cells", C.c_uint64), ("n_sw_spec", C.c_uint64), ("n_sw_spec_used", C.c_uint64), ("n_seed_redo", C.c_uint64), ("hit_list_cap", C.c_uint64),
        ("n_seed_shared", C.c_uint64), ("n_seed_shared_builds", C.c_uint64),
    ]


class Kprof(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("ms", C.c_double), ("launches", C.c_uint64), ("bytes", C.c_uint64)]


EXPORTS = [
    "smr_params_default", "smr_params_refused", "smr_index_load_files", "smr_index_build", "smr_index_build_gpu", "smr_index_write_files", "smr_index_save", "smr_index_load_flat", "smr_index_selfcheck", "smr_index_free",
    "smr_index_get_info", "smr_minimal_score", "smr_minimal_score_split", "smr_refstats_corrected_split", "smr_reads_pack", "smr_reads_load_fastx", "smr_reads_load_fastx_mt", "smr_reads_load_fastx_text", "smr_reads_is_fastq", "smr_reads_record_text", "smr_reads_free", "smr_reads_slice",
    "smr_reads_digest", "smr_reads_count", "smr_reads_total_len", "smr_reads_min_len", "smr_reads_max_len", "smr_create", "smr_device_count", "smr_destroy",
    "smr_last_error", "smr_index_upload", "smr_index_check_device", "smr_index_pigeonhole", "smr_seed_tuples_fetch", "smr_index_unload", "smr_batch_select", "smr_set_seed_mode", "smr_reads_upload", "smr_reads_upload_batch", "smr_state_reset", "smr_align_part",
    "smr_traceback", "smr_counters", "smr_counters_device", "smr_results_fetch", "smr_result_record", "smr_result_record_batch", "smr_counters_accumulate",
    "smr_result_is_hit", "smr_seed_scan", "smr_seed_hits_fetch", "smr_sw_selfcheck", "smr_sw_mode", "smr_walk_rounds", "smr_ssw_batch", "smr_cigar_batch", "smr_prof_reset", "smr_prof_get", "smr_prof_kernels", "smr_refstats_corrected", "smr_report_open",
    "smr_report_set_db", "smr_report_set_part", "smr_report_add", "smr_report_add_pair", "smr_report_set_cmdline", "smr_report_close", "smr_report_last_error",
    "smr_summary_write", "smr_readstats_record", "smr_readstats_key",
]

_lib = None


def load(rebuild_if_stale=True):
    """Load libsmr_hip.so (building it with hipcc when missing/stale).  Raises if it cannot be built:
    there is no CPU fallback for the product path."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if rebuild_if_stale and _build.is_stale():
        path = _build.build_library()
    if not os.path.isfile(path):
        raise RuntimeError("libsmr_hip.so is missing; run sortmerna_amd/build.py (needs hipcc)")
    _lib = bind(C.CDLL(path))
    return _lib


def bind(L):
    """Attach the argument/result types of include/smr_hip.h to a loaded library."""
    vp, cp, u32, u64, i32 = C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint64, C.c_int
    L.smr_params_default.argtypes = [C.POINTER(Params)]
    L.smr_index_load_files.restype = i32
    L.smr_index_load_files.argtypes = [cp, u32, cp, C.POINTER(vp), cp, C.c_size_t]
    L.smr_index_build.restype = i32
    L.smr_index_build.argtypes = [cp, u32, C.c_double, u32, u32, C.POINTER(vp), u32, C.POINTER(u32), cp, C.c_size_t]
    L.smr_index_build_gpu.restype = i32
    L.smr_index_build_gpu.argtypes = [vp, cp, u32, C.c_double, u32, C.POINTER(vp), u32, C.POINTER(u32), cp, C.c_size_t]
    L.smr_index_write_files.restype = i32
    L.smr_index_write_files.argtypes = [C.POINTER(vp), u32, cp, cp, cp, C.c_size_t]
    L.smr_index_save.restype = i32
    L.smr_index_save.argtypes = [vp, cp, u64, cp, C.c_size_t]
    L.smr_index_load_flat.restype = i32
    L.smr_index_load_flat.argtypes = [cp, u64, C.POINTER(vp), cp, C.c_size_t]
    L.smr_index_selfcheck.restype = i32
    L.smr_index_selfcheck.argtypes = [vp, cp, C.c_size_t]
    L.smr_index_free.argtypes = [vp]
    L.smr_index_get_info.restype = i32
    L.smr_index_get_info.argtypes = [vp, C.POINTER(IndexInfo)]
    L.smr_minimal_score.restype = u32
    L.smr_minimal_score.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double), u64, u64, u64, u64, C.c_double]
    L.smr_minimal_score_split.restype = u32
    L.smr_minimal_score_split.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double), u64, u64, u64, u64, C.c_double, u32]
    L.smr_reads_pack.restype = i32
    L.smr_reads_pack.argtypes = [cp, vp, u32, C.POINTER(vp)]
    L.smr_reads_load_fastx.restype = i32
    L.smr_reads_load_fastx.argtypes = [cp, u64, u64, C.POINTER(vp), cp, C.c_size_t]
    L.smr_reads_load_fastx_mt.restype = i32
    L.smr_reads_load_fastx_mt.argtypes = [cp, u32, C.POINTER(vp), cp, C.c_size_t]
    L.smr_reads_load_fastx_text.restype = i32
    L.smr_reads_load_fastx_text.argtypes = [cp, u32, C.POINTER(vp), cp, C.c_size_t]
    L.smr_reads_is_fastq.restype = i32
    L.smr_reads_is_fastq.argtypes = [vp]
    L.smr_reads_record_text.restype = i32
    L.smr_reads_record_text.argtypes = [vp, u32, cp, C.c_size_t, cp, C.c_size_t, cp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.smr_reads_free.argtypes = [vp]
    for f in ("smr_reads_count", "smr_reads_min_len", "smr_reads_max_len"):
        getattr(L, f).restype = u32
        getattr(L, f).argtypes = [vp]
    L.smr_reads_digest.restype = u64
    L.smr_reads_digest.argtypes = [vp]
    L.smr_reads_total_len.restype = u64
    L.smr_reads_total_len.argtypes = [vp]
    L.smr_create.restype = i32
    L.smr_create.argtypes = [i32, C.POINTER(vp), cp, C.c_size_t]
    L.smr_destroy.argtypes = [vp]
    L.smr_last_error.restype = cp
    L.smr_last_error.argtypes = [vp]
    L.smr_index_upload.restype = i32
    L.smr_index_upload.argtypes = [vp, vp, i32]
    L.smr_index_check_device.restype = i32
    L.smr_index_check_device.argtypes = [vp, i32, vp]
    L.smr_index_pigeonhole.restype = i32
    L.smr_index_pigeonhole.argtypes = [vp, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(u64), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(u64), cp, C.c_size_t]
    L.smr_seed_tuples_fetch.restype = i32
    L.smr_seed_tuples_fetch.argtypes = [vp, vp, u64, vp, u32, C.POINTER(u32)]
    L.smr_index_unload.restype = i32
    L.smr_index_unload.argtypes = [vp, i32]
    L.smr_batch_select.restype = i32
    L.smr_batch_select.argtypes = [vp, i32]
    L.smr_set_seed_mode.restype = i32
    L.smr_set_seed_mode.argtypes = [vp, i32]
    L.smr_reads_upload.restype = i32
    L.smr_reads_upload.argtypes = [vp, vp, u32]
    L.smr_state_reset.restype = i32
    L.smr_state_reset.argtypes = [vp]
    L.smr_align_part.restype = i32
    L.smr_align_part.argtypes = [vp, i32, C.POINTER(Params)]
    L.smr_traceback.restype = i32
    L.smr_traceback.argtypes = [vp, i32, C.POINTER(Params)]
    L.smr_counters.restype = i32
    L.smr_counters.argtypes = [vp, C.POINTER(u64), u32]
    L.smr_counters_device.restype = i32
    L.smr_counters_device.argtypes = [vp, C.POINTER(vp), C.POINTER(u32)]
    L.smr_results_fetch.restype = i32
    L.smr_results_fetch.argtypes = [vp]
    L.smr_result_record.restype = C.c_size_t
    L.smr_result_record.argtypes = [vp, u32, vp, C.c_size_t]
    L.smr_result_record_batch.restype = C.c_size_t
    L.smr_result_record_batch.argtypes = [vp, i32, u32, vp, C.c_size_t]
    L.smr_counters_accumulate.restype = i32
    L.smr_counters_accumulate.argtypes = [vp, vp, u32]
    L.smr_result_is_hit.restype = i32
    L.smr_result_is_hit.argtypes = [vp, u32]
    L.smr_seed_scan.restype = i32
    L.smr_seed_scan.argtypes = [vp, i32, C.POINTER(Params), i32, i32, C.POINTER(u64)]
    L.smr_seed_hits_fetch.restype = i32
    L.smr_seed_hits_fetch.argtypes = [vp, vp, u64, C.POINTER(u64)]
    L.smr_sw_selfcheck.restype = i32
    L.smr_sw_selfcheck.argtypes = [vp, u32, u32, u32, C.POINTER(u64)]
    L.smr_ssw_batch.restype = i32
    L.smr_ssw_batch.argtypes = [vp, u32, vp, vp, vp, vp, i32, i32, i32, i32, i32, u32, i32, vp]
    L.smr_readstats_record.restype = C.c_size_t
    L.smr_readstats_record.argtypes = [u64, u64, u32, u32, u64, u64, vp, u32, vp, C.c_size_t]
    L.smr_readstats_key.restype = C.c_size_t
    L.smr_readstats_key.argtypes = [vp, u32, vp, C.c_size_t]
    L.smr_reads_upload_batch.restype = i32
    L.smr_reads_upload_batch.argtypes = [vp, i32, vp, u32]
    L.smr_reads_slice.restype = i32
    L.smr_reads_slice.argtypes = [vp, u64, u64, C.POINTER(vp)]
    L.smr_cigar_batch.restype = i32
    L.smr_cigar_batch.argtypes = [vp, u32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, u64, vp]
    L.smr_sw_mode.restype = i32
    L.smr_sw_mode.argtypes = [vp, i32]
    L.smr_walk_rounds.restype = i32
    L.smr_walk_rounds.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.smr_prof_reset.restype = i32
    L.smr_prof_reset.argtypes = [vp]
    L.smr_prof_get.restype = i32
    L.smr_prof_get.argtypes = [vp, C.POINTER(Prof)]
    L.smr_prof_kernels.restype = i32
    L.smr_prof_kernels.argtypes = [vp, C.POINTER(Kprof), u32, C.POINTER(u32)]
    L.smr_refstats_corrected.argtypes = [C.c_double, C.POINTER(C.c_double), u64, u64, u64, u64, C.POINTER(u64), C.POINTER(u64)]
    L.smr_refstats_corrected_split.argtypes = [C.c_double, C.POINTER(C.c_double), u64, u64, u64, u64, C.c_uint32, C.POINTER(u64), C.POINTER(u64)]
    L.smr_refstats_corrected_split.restype = None
    L.smr_params_refused.argtypes = [C.c_void_p]
    L.smr_params_refused.restype = C.c_char_p
    L.smr_report_open.restype = i32
    L.smr_report_open.argtypes = [cp, C.POINTER(ReportOpts), i32, C.POINTER(vp), cp, C.c_size_t]
    L.smr_report_set_db.restype = i32
    L.smr_report_set_db.argtypes = [vp, u32, C.c_double, C.c_double, u64, u64]
    L.smr_report_set_part.restype = i32
    L.smr_report_set_part.argtypes = [vp, u32, u32, vp]
    L.smr_report_add.restype = i32
    L.smr_report_add.argtypes = [vp, cp, cp, cp, cp, C.c_size_t]
    L.smr_report_add_pair.restype = i32
    L.smr_report_add_pair.argtypes = [vp, cp, cp, cp, cp, C.c_size_t, cp, cp, cp, cp, C.c_size_t]
    L.smr_report_set_cmdline.restype = i32
    L.smr_report_set_cmdline.argtypes = [vp, cp]
    L.smr_summary_write.restype = i32
    L.smr_summary_write.argtypes = [cp, C.POINTER(Summary)]
    L.smr_report_close.restype = i32
    L.smr_report_close.argtypes = [vp]
    L.smr_report_last_error.restype = cp
    L.smr_report_last_error.argtypes = [vp]
    return L

"""ctypes binding of the C-ABI in include/kmernator_amd.h.

The shared library is built in-tree (kmernator_amd/csrc/libkmernator_amd.so) by
__graft_entry__.build() / `make -C kmernator_amd/csrc`.  There is no fallback: if the
library is missing, load() raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libkmernator_amd.so")

KMR_VALUE_COUNT_DIR, KMR_VALUE_EXT = 0, 1
KMR_MAP_WEAK, KMR_MAP_SINGLETON, KMR_MAP_SOLID = 0, 1, 2

STATUS = {0: "KMR_OK", -1: "KMR_ERR_INVALID_ARG", -2: "KMR_ERR_NO_DEVICE", -3: "KMR_ERR_HIP", -4: "KMR_ERR_OOM",
          -5: "KMR_ERR_STATE", -6: "KMR_ERR_CAPACITY", -7: "KMR_ERR_UNSUPPORTED"}


class KmrConfig(C.Structure):
    """kmr_config"""
    _fields_ = [
        ("struct_size", C.c_uint32), ("k", C.c_uint32),
        ("num_buckets_weak", C.c_uint64), ("num_buckets_singleton", C.c_uint64),
        ("estimated_raw_kmers", C.c_uint64),
        ("value_kind", C.c_uint32), ("min_weight", C.c_float),
        ("min_quality_score", C.c_uint32), ("fastq_start_char", C.c_uint32),
        ("ext_min_quality", C.c_uint32), ("separate_singletons", C.c_uint32),
        ("kmer_subsample", C.c_uint32), ("device", C.c_int32),
        ("rank", C.c_uint32), ("world_size", C.c_uint32),
        ("estimated_depth", C.c_double), ("estimated_error_rate", C.c_double),
        ("kmers_per_bucket", C.c_uint32), ("num_parts", C.c_uint32),
        ("part_idx", C.c_uint32), ("build_mode", C.c_uint32),
        ("max_table_entries", C.c_uint64),
        ("hash_kind", C.c_uint32), ("size_tracker", C.c_uint32),
    ]


class KmrStats(C.Structure):
    """kmr_stats"""
    _fields_ = [(n, C.c_uint64) for n in (
        "raw_kmers", "raw_good_kmers", "unique_kmers", "singleton_kmers",
        "discarded", "weak_entries", "singleton_entries", "reads")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class KmrDigest(C.Structure):
    """kmr_digest"""
    _fields_ = [(n, C.c_uint64) for n in ("entries", "count_sum", "dir_sum", "hash_sum", "hash_xor")] + [("weighted_sum", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class KmrArtifactConfig(C.Structure):
    """kmr_artifact_config"""
    _fields_ = [(n, C.c_uint32) for n in ("match_length", "edit_distance", "build_edits", "simple_repeat_begin", "simple_repeat_end",
                                          "phix_idx", "reference_begin", "min_quality", "fastq_start_char")] + [("min_read_length", C.c_float)]


# every symbol include/kmernator_amd.h declares
EXPORTS = [
    "kmr_abi_version", "kmr_config_init", "kmr_create", "kmr_destroy", "kmr_last_error", "kmr_num_buckets",
    "kmr_add_reads", "kmr_add_reads_dev", "kmr_add_reads_twobit", "kmr_add_reads_twobit_dev", "kmr_sync", "kmr_finalize", "kmr_get_stats", "kmr_lookup",
    "kmr_lookup_reads", "kmr_image_size", "kmr_write_image", "kmr_load_image", "kmr_count_histogram",
    "kmr_dump_mercount", "kmr_dump_mergraph", "kmr_hash", "kmr_hash_of_kind", "kmr_bucket_idx", "kmr_local_thread_id",
    "kmr_distributed_thread_id", "kmr_compress_sequence", "kmr_least_complement", "kmr_extract_by_owner_dev",
    "kmr_insert_records_dev", "kmr_stream", "kmr_kernel_time", "kmr_kernel_time_reset", "kmr_reset", "kmr_release_table", "kmr_score_reads",
    "kmr_ingest_fastq", "kmr_ingest_fastq_dev", "kmr_reads_info", "kmr_reads_device_ptrs", "kmr_reads_copy",
    "kmr_add_read_batch", "kmr_reads_free", "kmr_histogram", "kmr_histogram_bins", "kmr_merge_image", "kmr_subtract_reference", "kmr_subtracted", "kmr_score_read_batch",
    "kmr_artifact_config_init", "kmr_artifact_filter_create", "kmr_artifact_filter_info", "kmr_artifact_filter_entries",
    "kmr_artifact_filter_free", "kmr_artifact_filter_apply",
    "kmr_tune", "kmr_set_stream_origin", "kmr_size_tracker", "kmr_exchange_unique_id", "kmr_exchange_init", "kmr_exchange_init_transport", "kmr_exchange_add_reads_dev", "kmr_exchange_add_read_batch", "kmr_exchange_stats", "kmr_copy_to_host", "kmr_copy_to_device", "kmr_sk_exchange_begin", "kmr_sk_exchange_counts", "kmr_sk_exchange_pack_dev", "kmr_sk_exchange_adopt_dev", "kmr_extract_by_owner_host", "kmr_insert_records", "kmr_reads_from_host", "kmr_reads_from_twobit", "kmr_reads_twobit", "kmr_lookup_requests_dev", "kmr_lookup_keys_dev", "kmr_scatter_counts_dev", "kmr_score_counts_dev",
    "kmr_map_digest", "kmr_synth_reads_dev", "kmr_build_info", "kmr_sk_exchange_uniform", "kmr_sk_exchange_peer_uniform", "kmr_sk_exchange_range", "kmr_count_lists_prefix",
]

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, u8p, u32p, u64p, f64p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_double)
    lib.kmr_abi_version.restype = C.c_uint32
    lib.kmr_config_init.argtypes = [C.POINTER(KmrConfig)]
    lib.kmr_create.argtypes = [C.POINTER(KmrConfig), C.POINTER(vp)]
    lib.kmr_destroy.argtypes = [vp]
    lib.kmr_destroy.restype = None
    lib.kmr_last_error.argtypes = [vp]
    lib.kmr_last_error.restype = C.c_char_p
    lib.kmr_num_buckets.argtypes = [vp, C.c_int, u64p]
    lib.kmr_add_reads.argtypes = [vp, vp, vp, u64p, C.c_uint64, C.c_uint64, u8p]
    lib.kmr_add_reads_dev.argtypes = [vp, vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, vp]
    lib.kmr_reads_from_twobit.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_uint64, C.POINTER(vp)]
    lib.kmr_add_reads_twobit.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_uint64, C.c_uint64, vp]
    lib.kmr_add_reads_twobit_dev.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, vp]
    lib.kmr_sync.argtypes = [vp]
    lib.kmr_finalize.argtypes = [vp, C.c_uint32]
    lib.kmr_get_stats.argtypes = [vp, C.POINTER(KmrStats)]
    lib.kmr_lookup.argtypes = [vp, u8p, C.c_uint64, u32p]
    lib.kmr_lookup_reads.argtypes = [vp, vp, u64p, C.c_uint64, u32p, u64p]
    lib.kmr_image_size.argtypes = [vp, C.c_int, u64p]
    lib.kmr_write_image.argtypes = [vp, C.c_int, vp, C.c_uint64]
    lib.kmr_load_image.argtypes = [vp, C.c_int, vp, C.c_uint64]
    lib.kmr_count_histogram.argtypes = [vp, u64p, f64p, C.c_uint32]
    lib.kmr_dump_mercount.argtypes = [vp, C.c_char_p, C.c_uint32]
    lib.kmr_dump_mergraph.argtypes = [vp, C.c_char_p, C.c_uint32]
    lib.kmr_hash.restype = C.c_uint64
    lib.kmr_hash.argtypes = [C.c_char_p, C.c_uint32]
    lib.kmr_bucket_idx.restype = C.c_uint64
    lib.kmr_bucket_idx.argtypes = [C.c_uint64, C.c_uint64]
    lib.kmr_local_thread_id.restype = C.c_uint32
    lib.kmr_local_thread_id.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
    lib.kmr_distributed_thread_id.restype = C.c_uint32
    lib.kmr_distributed_thread_id.argtypes = [C.c_uint64, C.c_uint32]
    lib.kmr_compress_sequence.restype = C.c_int64
    lib.kmr_compress_sequence.argtypes = [C.c_char_p, C.c_uint64, u8p, u32p, C.c_char_p, C.c_uint64]
    lib.kmr_least_complement.argtypes = [u8p, C.c_uint32, u8p]
    lib.kmr_extract_by_owner_dev.argtypes = [vp, vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, vp, vp, C.c_uint64, vp]
    lib.kmr_insert_records_dev.argtypes = [vp, vp, C.c_uint64]
    lib.kmr_stream.restype = vp
    lib.kmr_stream.argtypes = [vp]
    lib.kmr_kernel_time.argtypes = [vp, C.c_int, f64p, u64p]
    lib.kmr_kernel_time_reset.argtypes = [vp]
    lib.kmr_score_reads.argtypes = [vp, vp, u64p, C.c_uint64, C.c_double, C.c_int, u32p, u32p, C.POINTER(C.c_float), u8p]
    lib.kmr_reset.argtypes = [vp]
    lib.kmr_release_table.argtypes = [vp]
    lib.kmr_ingest_fastq.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(vp)]
    lib.kmr_ingest_fastq_dev.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(vp)]
    lib.kmr_reads_info.argtypes = [vp, u64p, u64p, u32p, u64p]
    lib.kmr_reads_device_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.kmr_reads_copy.argtypes = [vp, vp, vp, u64p, u64p, u32p]
    lib.kmr_add_read_batch.argtypes = [vp, vp, C.c_uint64]
    lib.kmr_reads_free.argtypes = [vp]
    lib.kmr_reads_free.restype = None
    lib.kmr_score_read_batch.argtypes = [vp, vp, C.c_double, C.c_int, u32p, u32p, C.POINTER(C.c_float), u8p]
    lib.kmr_reads_twobit.argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(C.c_uint64), u32p, vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.kmr_reads_from_host.argtypes = [vp, vp, vp, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(vp)]
    lib.kmr_lookup_requests_dev.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64, vp, vp, C.c_uint64, vp]
    lib.kmr_lookup_keys_dev.argtypes = [vp, vp, C.c_uint64, vp]
    lib.kmr_scatter_counts_dev.argtypes = [vp, vp, vp, C.c_uint64, vp]
    lib.kmr_score_counts_dev.argtypes = [vp, vp, vp, C.c_uint64, vp, C.c_double, C.c_int, u32p, u32p, C.POINTER(C.c_float), u8p]
    lib.kmr_artifact_config_init.argtypes = [C.POINTER(KmrArtifactConfig)]
    lib.kmr_artifact_config_init.restype = None
    lib.kmr_artifact_filter_create.argtypes = [vp, C.POINTER(KmrArtifactConfig), C.c_char_p, C.c_uint64, C.POINTER(vp)]
    lib.kmr_artifact_filter_info.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    lib.kmr_artifact_filter_entries.argtypes = [vp, C.POINTER(C.c_uint64), u32p, C.c_uint64]
    lib.kmr_artifact_filter_free.argtypes = [vp]
    lib.kmr_artifact_filter_free.restype = None
    lib.kmr_artifact_filter_apply.argtypes = [vp, vp, vp, C.POINTER(C.c_int64), u32p, u32p, u32p, u8p, u32p, u32p, C.POINTER(vp)]
    lib.kmr_extract_by_owner_host.argtypes = [vp, vp, C.c_uint64, u64p, vp, C.c_uint64]
    lib.kmr_insert_records.argtypes = [vp, vp, C.c_uint64]
    lib.kmr_sk_exchange_begin.argtypes = [vp]
    lib.kmr_size_tracker.argtypes = [vp, C.c_int, u64p, C.c_uint64, u64p]
    lib.kmr_exchange_unique_id.argtypes = [vp]
    lib.kmr_exchange_init.argtypes = [vp, vp]
    lib.kmr_exchange_init_transport.argtypes = [vp, vp]
    lib.kmr_exchange_add_reads_dev.argtypes = [vp, vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, vp]
    lib.kmr_exchange_add_read_batch.argtypes = [vp, vp, C.c_uint64]
    lib.kmr_exchange_stats.argtypes = [vp, u64p, C.POINTER(C.c_double)]
    lib.kmr_sk_exchange_counts.argtypes = [vp, u64p, u64p]
    lib.kmr_sk_exchange_pack_dev.argtypes = [vp, vp, vp, u64p, u64p]
    lib.kmr_sk_exchange_adopt_dev.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64]
    lib.kmr_set_stream_origin.argtypes = [vp, C.c_uint64]
    lib.kmr_tune.argtypes = [vp, C.c_char_p, C.c_double]
    lib.kmr_merge_image.argtypes = [vp, C.c_int, vp, C.c_uint64]
    lib.kmr_subtract_reference.argtypes = [vp, vp]
    lib.kmr_subtracted.argtypes = [vp, u64p]
    lib.kmr_histogram_bins.restype = C.c_uint32
    lib.kmr_histogram_bins.argtypes = [C.c_uint32]
    lib.kmr_histogram.argtypes = [vp, C.c_uint32, C.c_double, u64p, u64p, f64p, C.c_uint32]
    lib.kmr_map_digest.argtypes = [vp, C.c_int, C.POINTER(KmrDigest)]
    lib.kmr_build_info.argtypes = [vp, C.c_char_p, f64p]
    lib.kmr_sk_exchange_uniform.argtypes = [vp, u64p]
    lib.kmr_sk_exchange_peer_uniform.argtypes = [vp, C.c_uint64]
    lib.kmr_sk_exchange_range.argtypes = [vp, C.c_uint64, C.c_uint64]
    lib.kmr_count_lists_prefix.argtypes = [vp, C.c_uint32, C.c_uint64]
    lib.kmr_synth_reads_dev.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, vp, vp, vp]
    _lib = lib
    return lib


def default_config(k, **kw):
    c = KmrConfig()
    load().kmr_config_init(C.byref(c))
    c.k = k
    for name, v in kw.items():
        setattr(c, name, v)
    return c


def record_bytes(k, value_kind=KMR_VALUE_COUNT_DIR):
    """KMR_RECORD_BYTES(k, value_kind)"""
    return 8 * ((((k + 3) // 4) + 7) // 8) + (8 if value_kind == KMR_VALUE_EXT else 4)

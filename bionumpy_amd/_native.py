"""ctypes binding of libbnpk.so (the C-ABI declared in include/bnpk.h).

There is no CPU fallback: if the shared library is missing this module raises at import, and
creating a context without a visible gfx950 device raises ``BnpkError``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BNPK_LIB") or os.path.join(_HERE, "csrc", "libbnpk.so")   # (BNPK_LIB: experiment builds, scripts/exp/build_variant.sh)

NONE = (1 << 63) - 1          # BNPK_NONE


class BnpkError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        super().__init__("bnpk error %d: %s%s" % (status, _strerror(status), (" (%s)" % detail) if detail else ""))


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "bionumpy_amd: %s is missing. Build it with `python bionumpy_amd/csrc/build.py` "
        "(hipcc --offload-arch=gfx950); there is no CPU fallback." % LIB_PATH)

# torch bundles its own libamdhip64.so.7; it must be the first HIP runtime in the process (loading the
# system one first leaves torch.cuda unusable), so import torch before dlopen-ing libbnpk.so, whose
# libamdhip64.so.7 dependency then resolves to the runtime torch already loaded.
import torch  # noqa: E402,F401

lib = C.CDLL(LIB_PATH)

_p = C.c_void_p
_i64 = C.c_int64
_int = C.c_int
_u8 = C.c_uint8

# name -> (restype, argtypes); must list every symbol of include/bnpk.h (checked by tests/test_abi.py)
SIGNATURES = {
    "bnpk_version": (_int, []),
    "bnpk_strerror": (C.c_char_p, [_int]),
    "bnpk_device_count": (_int, []),
    "bnpk_ctx_create": (_int, [_int, C.POINTER(_p)]),
    "bnpk_ctx_destroy": (None, [_p]),
    "bnpk_last_hip_error": (C.c_char_p, [_p]),
    "bnpk_device_info": (_int, [_p, C.c_char_p, C.POINTER(_int), C.POINTER(_i64)]),
    "bnpk_prof_enable": (_int, [_p, _int]),
    "bnpk_copy_peak": (_int, [_p, _p, _p, _i64, _int, C.POINTER(C.c_double), _p]),
    "bnpk_copy_rates": (_int, [_p, _p, _p, _i64, _int, C.POINTER(C.c_double), _p]),
    "bnpk_set_option": (_int, [_p, C.c_char_p, _i64]),
    "bnpk_count_sparse_workspace": (_i64, [_i64, _int, _int, _i64, _int, _int]),
    "bnpk_count_sparse": (_int, [_p, _p, _i64, _int, _int, _i64, _p, _int, _p, _i64, _p, _p, C.POINTER(_i64), C.POINTER(_i64), _p]),
    "bnpk_index_build_workspace": (_i64, [_i64, _int, _i64]),
    "bnpk_index_build": (_int, [_p, _p, _p, _i64, _int, _i64, _p, _i64, _p, _p, _p, C.POINTER(_i64), _p]),
    "bnpk_comm_available": (_int, []),
    "bnpk_comm_unique_id": (_int, [_p]),
    "bnpk_comm_init": (_int, [_p, _p, _int, _int, C.POINTER(C.c_void_p)]),
    "bnpk_comm_destroy": (_int, [_p]),
    "bnpk_comm_shape": (_int, [_p, C.POINTER(_int), C.POINTER(_int)]),
    "bnpk_last_comm_error": (C.c_char_p, []),
    "bnpk_allreduce_hist": (_int, [_p, _p, _p, _i64, _p]),
    "bnpk_exchange_counts": (_int, [_p, _p, _p, _int, _p, _p]),
    "bnpk_exchange_by_key_range": (_int, [_p, _p, _p, _p, _p, _p, _p]),
    "bnpk_exchange_slices": (_int, [_p, _p, _p, _p, _p, _p, _p, _p]),
    "bnpk_prof_reset": (_int, [_p]),
    "bnpk_prof_count": (_int, [_p]),
    "bnpk_prof_get": (_int, [_p, _int, C.c_char_p, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "bnpk_host_alloc": (_int, [C.c_size_t, C.POINTER(_p)]),
    "bnpk_host_free": (_int, [_p]),
    "bnpk_copy_h2d_async": (_int, [_p, _p, C.c_size_t, _p]),
    "bnpk_copy_d2h_async": (_int, [_p, _p, C.c_size_t, _p]),
    "bnpk_pread_parallel": (_int, [_p, _int, _i64, _p, _i64, _int, _i64, _p, _p, C.POINTER(_i64)]),
    "bnpk_count_byte_file": (_int, [_int, _i64, _i64, _u8, _int, C.POINTER(_i64)]),
    "bnpk_stream_sync": (_int, [_p]),
    "bnpk_fetch_i64": (_int, [_p, _p, _i64, C.POINTER(_i64), _p]),
    "bnpk_scan_tiles": (_i64, [_i64]),
    "bnpk_byte_census": (_int, [_p, _p, _i64, _u8, _p, _p]),
    "bnpk_byte_positions": (_int, [_p, _p, _i64, _u8, _p, _i64, _p, _p]),
    "bnpk_line_positions": (_int, [_p, _p, _i64, _p, _i64, _int, _u8, _int, _p, _p, _p]),
    "bnpk_validate_entries": (_int, [_p, _p, _p, _i64, _int, _u8, _int, _p, _p]),
    "bnpk_field_table": (_int, [_p, _p, _p, _i64, _int, _int, _int, _int, _p, _p, _p]),
    "bnpk_window_cuts_words": (_i64, [_int]),
    "bnpk_window_cuts": (_int, [_p, _p, _p, _i64, _int, _i64, _i64, _int, _i64, _i64, _int, _p, _p]),
    "bnpk_rebase_lines": (_int, [_p, _p, _i64, _p, _int, _p, _p]),
    "bnpk_fastq_tiles": (_i64, [_i64]),
    "bnpk_fastq_table_words": (_i64, [_i64]),
    "bnpk_fastq_census": (_int, [_p, _p, _i64, _int, _int, _p, C.POINTER(_i64), _p]),
    "bnpk_fastq_encode": (_int, [_p, _p, _i64, _int, _int, _u8, _int, _p, _i64, _i64, _p, _p, _p, _p]),
    "bnpk_kmer_starts_from_ends": (_int, [_p, _p, _i64, _int, _p, _p, _p]),
    "bnpk_row_offsets": (_int, [_p, _p, _i64, _int, _p, _p]),
    "bnpk_packed_rows_slice": (_int, [_p, _p, _i64, _p, _i64, _i64, _i64, _i64, _p, _p, _p]),
    "bnpk_gather_encode_dna": (_int, [_p, _p, _i64, _p, _p, _i64, _i64, _p, _p, _p, _p, _p]),
    "bnpk_gather_rows": (_int, [_p, _p, _p, _p, _i64, _i64, _int, _p, _p]),
    "bnpk_take_bytes": (_int, [_p, _p, _p, _i64, _i64, _p, _p]),
    "bnpk_encode_dna_flat": (_int, [_p, _p, _i64, _p, _p, _p, _p]),
    "bnpk_pack_codes": (_int, [_p, _p, _i64, _p, _p]),
    "bnpk_unpack_codes": (_int, [_p, _p, _i64, _int, _p, _p]),
    "bnpk_kmers": (_int, [_p, _p, _p, _p, _i64, _i64, _int, _p, _p]),
    "bnpk_multiline_cut": (_int, [_p, _p, _p, _i64, _u8, C.POINTER(_i64), C.POINTER(_i64), _p]),
    "bnpk_multiline_table": (_int, [_p, _p, _i64, _p, _i64, _u8, _int, _p, _p, _p, _p, _p, _p, _p]),
    "bnpk_multiline_wrap": (_int, [_p, _p, _p, _p, _p, _i64, _int, _u8, _p, _p, _i64, C.POINTER(_i64), _p]),
    "bnpk_merge_add": (_int, [_p, _p, _p, _i64, _p, _p, _i64, _p, _p, C.POINTER(_i64), _p]),
    "bnpk_pair_compose": (_int, [_p, _p, _p, _i64, _i64, _p, _p]),
    "bnpk_pair_split": (_int, [_p, _p, _i64, _i64, _p, _p, _p, _p]),
    "bnpk_kmers_generic": (_int, [_p, _p, _p, _p, _i64, _i64, _int, _int, _p, _p]),
    "bnpk_minimizers_generic": (_int, [_p, _p, _p, _p, _i64, _i64, _int, _int, _int, _p, _p]),
    "bnpk_lut_bytes": (_int, [_p, _p, _i64, _p, _p, _p, _p]),
    "bnpk_kmer_start_mask": (_int, [_p, _p, _i64, _i64, _int, _p, _p]),
    "bnpk_row_end_mask": (_int, [_p, _p, _i64, _i64, _p, _p]),
    "bnpk_join_lines": (_int, [_p, _i64, _int, _p, _p, _p, _p, _p, _p, _p, _u8, _p, _i64, _p, _p]),
    "bnpk_col_sums_u8": (_int, [_p, _p, _p, _i64, _i64, _i64, _p, _p, _p]),
    "bnpk_row_reduce_u8": (_int, [_p, _p, _p, _i64, _p, _p, _p, _p]),
    "bnpk_row_reduce_u8_view": (_int, [_p, _p, _i64, _p, _p, _i64, _int, _p, _p, _p, _p]),
    "bnpk_row_reduce_wide": (_int, [_p, _p, _int, _p, _i64, _p, _p, _p, _p]),
    "bnpk_vec_ratio_rows": (_int, [_p, _p, _p, _i64, _p, _p]),
    "bnpk_vec_compare": (_int, [_p, _p, _i64, _int, _int, C.c_double, _i64, _p, _p]),
    "bnpk_mask_logic": (_int, [_p, _p, _p, _i64, _int, _p, _p]),
    "bnpk_mask_fill": (_int, [_p, _p, _i64, _i64, _i64, _i64, _int, _p]),
    "bnpk_take_i64": (_int, [_p, _p, _p, _i64, _p, _p]),
    "bnpk_entry_table": (_int, [_p, _p, _int, _p, _i64, _p, _p, _p]),
    "bnpk_join_line_lens": (_int, [_p, _i64, _int, _p, _p, _p, _p]),
    "bnpk_reverse_complement_packed": (_int, [_p, _p, _p, _i64, _i64, _p, _p]),
    "bnpk_reverse_complement_bytes": (_int, [_p, _p, _p, _i64, _i64, _p, _p]),
    "bnpk_reverse_complement_rows": (_int, [_p, _p, _i64, _p, _p, _i64, _i64, _p, _p]),
    "bnpk_canonical_kmers": (_int, [_p, _p, _i64, _int, _p]),
    "bnpk_windows_flat": (_int, [_p, _p, _p, _i64, _int, _int, _i64, _p, _p]),
    "bnpk_match_windows_packed": (_int, [_p, _p, _p, _i64, _int, C.c_uint64, _i64, _p, _p]),
    "bnpk_match_windows_bytes": (_int, [_p, _p, _p, _i64, _int, _p, _i64, _p, _p]),
    "bnpk_match_rows_packed": (_int, [_p, _p, _i64, _p, _i64, _int, C.c_uint64, _p, _p]),
    "bnpk_pwm_scores": (_int, [_p, _p, _p, _i64, _int, _p, _i64, _p, _p]),
    "bnpk_kmers_partition": (_int, [_p, _p, _p, _i64, _int, _int, _int, _int, _p, _p, _p]),
    "bnpk_minimizers": (_int, [_p, _p, _p, _p, _i64, _i64, _int, _int, _p, _p]),
    "bnpk_count_bytes": (_int, [_p, _p, _i64, _int, _p, _p]),
    "bnpk_count_packed2": (_int, [_p, _p, _i64, _p, _p]),
    "bnpk_count_bytes_rows": (_int, [_p, _p, _p, _i64, _i64, _int, _p, _p]),
    "bnpk_count_dense": (_int, [_p, _p, _i64, _i64, _p, _p]),
    "bnpk_count_dense_rows": (_int, [_p, _p, _p, _i64, _i64, _i64, _p, _p]),
    "bnpk_count_weighted": (_int, [_p, _p, _p, _int, _i64, _i64, _i64, _i64, _i64, _p, C.POINTER(_int), _p]),
    "bnpk_sort_keys": (_int, [_p, _p, _p, _i64, _int, _int, C.POINTER(_int), _p]),
    "bnpk_sort_pairs": (_int, [_p, _p, _p, _p, _p, _i64, _int, C.POINTER(_int), _p]),
    "bnpk_radix_max_bits": (_i64, []),
    "bnpk_finish_capacity": (_i64, []),
    "bnpk_radix_partition": (_int, [_p, _p, _i64, _p, _i64, _int, _int, _p, _p, _p]),
    "bnpk_claimed_stride": (_i64, []),
    "bnpk_claimed_cap_lo": (_i64, []),
    "bnpk_radix_partition_claimed": (_int, [_p, _p, _i64, _p, _i64, _int, _int, _p, _p, _p, _i64, _p, _p]),
    "bnpk_claimed_finalize": (_int, [_p, _p, _p, _i64, _p, _p]),
    "bnpk_radix_small_capacity": (_i64, []),
    "bnpk_radix_partition_small": (_int, [_p, _p, _i64, _p, _i64, _int, _int, _p, _p, _p]),
    "bnpk_bucket_census": (_int, [_p, _p, _i64, _i64, _int, _p, _p]),
    "bnpk_finish_state_words": (_i64, [_i64]),
    "bnpk_finish_sorted": (_int, [_p, _p, _i64, _p, _i64, _int, _p, _p, _p, _p, _int, _p, _p, C.POINTER(_i64),
                                  C.POINTER(_int), _p]),
    "bnpk_finish_sorted_strided": (_int, [_p, _p, _i64, _i64, _p, _i64, _int, _p, _p, _p, _p, _int, _p, _p, C.POINTER(_i64),
                                          C.POINTER(_int), _p]),
    "bnpk_run_tiles": (_i64, [_i64]),
    "bnpk_run_census": (_int, [_p, _p, _p, _i64, _p, C.POINTER(_i64), _p]),
    "bnpk_run_heads": (_int, [_p, _p, _p, _i64, _p, _i64, _p, _p, _p, _p]),
    "bnpk_run_sums": (_int, [_p, _p, _i64, _p, _p, _p]),
    "bnpk_exclusive_scan_i64": (_int, [_p, _p, _i64, _p, _p]),
    "bnpk_row_ids": (_int, [_p, _p, _i64, _i64, _p, _p]),
    "bnpk_search_sorted": (_int, [_p, _p, _i64, _p, _i64, _int, _p, _p]),
    "bnpk_fill_i64": (_int, [_p, _p, _i64, _i64, _p]),
    "bnpk_synth_record_bytes": (_i64, [_int]),
    "bnpk_synth_fastq": (_int, [_p, _p, _i64, _i64, _int, C.c_uint64, _int, _i64, _p]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _f = getattr(lib, _name)          # AttributeError here == symbol missing from the build
    _f.restype = _res
    _f.argtypes = _args


def _strerror(status):
    return lib.bnpk_strerror(int(status)).decode()


def check(status, ctx=None):
    if status != 0:
        detail = ""
        if ctx is not None and status == -3:
            detail = lib.bnpk_last_hip_error(ctx).decode()
        raise BnpkError(status, detail)

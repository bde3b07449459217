"""ctypes binding of the C ABI declared in include/cafehip.h.

There is no fallback of any kind: if libcafehip.so is missing it is built with hipcc, and if it
cannot be built or loaded an exception is raised.
"""
import ctypes as C
import os

from . import build as _build

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)

# name -> (restype, argtypes); must list every symbol of include/cafehip.h
SIGNATURES = {
    "cafehip_abi_version": (C.c_int, []),
    "cafehip_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "cafehip_destroy": (None, [C.c_void_p]),
    "cafehip_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "cafehip_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    "cafehip_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cafehip_get_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "cafehip_set_tree": (C.c_int, [C.c_void_p, C.c_int, _ip, _ip, _ip, _dp]),
    "cafehip_set_families": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _ip, _ip, C.c_int, C.c_int, C.c_int, C.c_int]),
    "cafehip_last_setup_ms": (C.c_int, [C.c_void_p, _dp]),
    "cafehip_set_error_model": (C.c_int, [C.c_void_p, C.c_int, _dp, _u8p]),
    "cafehip_eval_posterior": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _ip, _dp, _ip, _dp]),
    "cafehip_eval_posterior_sequence": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _ip, C.c_int]),
    "cafehip_eval_posterior_multi": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _ip]),
    "cafehip_eval_clustered_posterior": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp, _ip, _dp, _dp, _dp]),
    "cafehip_launch_info": (C.c_int, [C.c_void_p, _ip, _ip]),
    "cafehip_last_issued_flops": (C.c_int, [C.c_void_p, _dp, _dp]),
    "cafehip_last_tables_ms": (C.c_int, [C.c_void_p, _dp]),
    "cafehip_eval_posterior_async": (C.c_int, [C.c_void_p, _dp, _dp, _dp, C.c_void_p, C.c_void_p]),
    "cafehip_num_chunks": (C.c_int, [C.c_void_p]),
    "cafehip_get_matrix": (C.c_int, [C.c_void_p, C.c_int, _dp, C.POINTER(C.c_int)]),
    "cafehip_matrix_size": (C.c_int, [C.c_void_p]),
    "cafehip_prefetch_matrices": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, C.c_int]),
    "cafehip_prearm_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_long)]),
    "cafehip_matrix_cache_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_long)]),
    "cafehip_reset_birthdeath_cache": (C.c_int, [C.c_void_p, _dp, _dp]),
    "cafehip_set_exact_matrices": (C.c_int, [C.c_void_p, C.c_int]),
    "cafehip_eval_root_likelihoods": (C.c_int, [C.c_void_p, C.c_int, _ip, _ip, _ip, _ip, _dp]),
    "cafehip_viterbi": (C.c_int, [C.c_void_p, C.c_int, _ip, _ip, _ip, _ip, _ip]),
    "cafehip_comm_unique_id": (C.c_int, [C.c_void_p]),
    "cafehip_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "cafehip_comm_set_blocks": (C.c_int, [C.c_void_p, _ip, _ip]),
    "cafehip_eval_posterior_sharded": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _ip]),
    "cafehip_comm_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "cafehip_comm_host_selftest": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "cafehip_comm_info": (C.c_int, [C.c_void_p, _ip, _ip, _ip, _dp, _dp, C.POINTER(C.c_long)]),
    "cafehip_comm_resync": (C.c_int, [C.c_void_p]),
    "cafehip_comm_status": (C.c_int, [C.c_void_p, _ip, _dp]),
    "cafehip_comm_cleanup": (C.c_int, [C.c_void_p]),
    "cafehip_comm_mode_selftest": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "cafehip_exp_like_host_selftest": (C.c_int, [C.c_long, C.c_uint, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "cafehip_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "cafehip_fetch_small": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "cafehip_last_kernel_ms": (C.c_int, [C.c_void_p, _dp]),
    "cafehip_last_batch_ms": (C.c_int, [C.c_void_p, _dp]),
    "cafehip_describe": (C.c_char_p, [C.c_void_p]),
    "cafehip_last_error": (C.c_char_p, []),
}

# include/cafehost.h (host driver above the kernel boundary)
HOST_SIGNATURES = {
    "cafehost_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_char_p]),
    "cafehost_destroy": (None, [C.c_void_p]),
    "cafehost_dispatch": (C.c_int, [C.c_void_p, C.c_char_p]),
    "cafehost_run_script": (C.c_int, [C.c_void_p, C.c_char_p]),
    "cafehost_rng_selftest": (C.c_int, [C.c_uint, C.c_int, C.c_int, C.c_int]),
    "cafehost_poisson_fit_selftest": (C.c_int, [_ip, C.c_long, C.c_double, C.c_int, _dp, _dp, C.POINTER(C.c_int), C.POINTER(C.c_long)]),
    "cafehost_format_selftest": (C.c_long, [_dp, C.c_long, _dp]),
    "cafehost_pvalue_selftest": (C.c_double, [C.c_double, _dp, C.c_int]),
    "cafehost_fminsearch_selftest": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _dp, C.c_double, C.c_double, _dp, _dp, C.POINTER(C.c_int)]),
    "cafehost_lookahead_selftest": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _dp, C.c_double, C.c_double, _dp, _dp, C.POINTER(C.c_long)]),
    "cafehost_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "cafehost_set_shard": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "cafehost_shard_bounds": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "cafehost_set_exchange": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cafehost_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cafehost_comm_unique_id": (C.c_int, [C.c_void_p]),
    "cafehost_comm_cleanup": (C.c_int, [C.c_void_p]),
    "cafehost_init_comm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "cafehost_speculation_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "cafehost_lookahead_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_long)]),
    "cafehost_exchange_stats": (C.c_int, [C.c_void_p, _dp, C.POINTER(C.c_long)]),
    "cafehost_set_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cafehost_fetch_small": (C.c_int, [C.c_void_p, C.c_void_p, C.c_ulong, C.POINTER(C.c_void_p)]),
    "cafehost_upload": (C.c_int, [C.c_void_p]),
    "cafehost_num_params": (C.c_int, [C.c_void_p]),
    "cafehost_get_params": (C.c_int, [C.c_void_p, _dp, C.c_int]),
    "cafehost_last_score": (C.c_double, [C.c_void_p]),
    "cafehost_search_iterations": (C.c_int, [C.c_void_p]),
    "cafehost_num_evaluations": (C.c_int, [C.c_void_p]),
    "cafehost_search_seconds": (C.c_double, [C.c_void_p]),
    "cafehost_poisson_lambda": (C.c_double, [C.c_void_p]),
    "cafehost_get_trace": (C.c_int, [C.c_void_p, _dp, C.c_int]),
    "cafehost_last_error": (C.c_char_p, []),
}

CHUNK = 256  # CAFEHIP_CHUNK

_lib = None


def lib_path():
    return _build.LIB


def load():
    """Load (building first if needed) libcafehip.so and bind every ABI symbol."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.build()
    L = C.CDLL(path)
    for table in (SIGNATURES, HOST_SIGNATURES):
        for name, (res, args) in table.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
    _lib = L
    return L


class CafeHipError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise CafeHipError(load().cafehip_last_error().decode("utf-8", "replace"))

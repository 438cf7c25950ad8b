"""ctypes binding of include/fastplaid.h.  The product path has NO CPU fallback: if the HIP
library is missing or does not load, every entry point raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FP_LIB_PATH") or os.path.join(_HERE, "libfastplaid_hip.so")   # (FP_LIB_PATH: A/B runs of library variants, tools/)


class FpIndexDesc(C.Structure):
    _fields_ = [
        ("nbits", C.c_int32), ("dim", C.c_int32), ("n_centroids", C.c_int64),
        ("centroids", C.c_void_p), ("avg_residual", C.c_void_p), ("bucket_cutoffs", C.c_void_p),
        ("bucket_weights", C.c_void_p), ("ivf", C.c_void_p), ("ivf_lengths", C.c_void_p),
        ("n_ivf_lists", C.c_int64), ("doc_codes", C.c_void_p), ("doc_residuals", C.c_void_p),
        ("doc_lengths", C.c_void_p), ("n_docs", C.c_int64), ("pid_offset", C.c_int64),
    ]


class FpSearchParams(C.Structure):
    _fields_ = [("batch_size", C.c_int64), ("n_full_scores", C.c_int64), ("top_k", C.c_int64),
                ("n_ivf_probe", C.c_int64)]


class FpSynthDesc(C.Structure):
    _fields_ = [
        ("nbits", C.c_int32), ("dim", C.c_int32), ("n_centroids", C.c_int64), ("centroids", C.c_void_p),
        ("bucket_weights", C.c_void_p), ("n_docs_total", C.c_int64), ("doc_begin", C.c_int64),
        ("doc_end", C.c_int64), ("doc_len", C.c_int32), ("variable_len", C.c_int32), ("seed", C.c_uint64),
    ]


# name -> (restype, argtypes); must list every symbol include/fastplaid.h declares
_vp, _i64, _i32 = C.c_void_p, C.c_int64, C.c_int32
SYMBOLS = {
    "fp_last_error": (C.c_char_p, []),
    "fp_version": (C.c_char_p, []),
    "fp_device_count": (C.c_int, []),
    "fp_index_create": (C.c_int, [C.POINTER(FpIndexDesc), C.c_int, C.POINTER(_vp)]),
    "fp_index_destroy": (None, [_vp]),
    "fp_index_num_docs": (_i64, [_vp]),
    "fp_index_num_tokens": (_i64, [_vp]),
    "fp_index_num_centroids": (_i64, [_vp]),
    "fp_index_dim": (_i32, [_vp]),
    "fp_index_nbits": (_i32, [_vp]),
    "fp_index_device_bytes": (_i64, [_vp]),
    "fp_index_num_unique_codes": (_i64, [_vp]),
    "fp_index_num_code_lines": (_i64, [_vp]),
    "fp_index_num_hard_tokens": (_i64, [_vp]),
    "fp_index_tickets_ok": (C.c_int32, [_vp]),
    "fp_search": (C.c_int, [_vp, _vp, _i32, _i32, _i32, C.POINTER(FpSearchParams), _vp, _vp, _vp, _vp, _vp]),
    "fp_search_shared_subset": (C.c_int, [_vp, _vp, _i32, _i32, _i32, C.POINTER(FpSearchParams), _vp, _i64, _vp, _vp, _vp]),
    "fp_device_free_bytes": (_i64, [C.c_int]),
    "fp_device_total_bytes": (_i64, [C.c_int]),
    "fp_dev_alloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(_vp)]),
    "fp_dev_free": (C.c_int, [C.c_int, _vp]),
    "fp_dev_upload": (C.c_int, [C.c_int, _vp, _vp, C.c_size_t]),
    "fp_dev_download": (C.c_int, [C.c_int, _vp, _vp, C.c_size_t]),
    "fp_search_device": (C.c_int, [_vp, _vp, _i32, _i32, _i32, C.POINTER(FpSearchParams), _vp, _vp, _vp]),
    "fp_search_trace": (C.c_int, [_vp, _vp, _i32, _i32, C.POINTER(FpSearchParams), _vp, _i64, _i32] + [_vp] * 10),
    "fp_last_search_timings": (C.c_int, [_vp, _vp, C.c_int]),
    "fp_set_graph_replay": (C.c_int, [C.c_int]),
    "fp_graph_replay_count": (C.c_uint64, []),
    "fp_reconstruct_embeddings": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp]),
    "fp_compress": (C.c_int, [C.c_int, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp, _vp]),
    "fp_assign_l2": (C.c_int, [C.c_int, _vp, _vp, _i64, _i32, _vp, _i64, _vp]),
    "fp_maxsim_columns": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp]),
    "fp_token_scores": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _i64]),
    "fp_shard_begin": (C.c_int, [_vp, _vp, _i32, _i32, _i32, C.POINTER(FpSearchParams), C.POINTER(_vp)]),
    "fp_shard_R": (_i64, [_vp]),
    "fp_shard_stage1": (C.c_int, [_vp, _vp]),
    "fp_shard_stage2": (C.c_int, [_vp, _vp, _i32, _vp]),
    "fp_shard_stage3": (C.c_int, [_vp, _vp, _i32, _i32, _vp]),
    "fp_shard_stage4": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp]),
    "fp_shard_end": (None, [_vp]),
    "fp_comm_unique_id": (C.c_int, [_vp]),
    "fp_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, C.POINTER(_vp)]),
    "fp_comm_destroy": (None, [_vp]),
    "fp_comm_n_ranks": (C.c_int, [_vp]),
    "fp_comm_rank": (C.c_int, [_vp]),
    "fp_shard_search": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, C.POINTER(FpSearchParams), _vp, _vp, _vp]),
    "fp_index_create_synthetic": (C.c_int, [C.POINTER(FpSynthDesc), C.c_int, C.POINTER(_vp)]),
    "fp_index_read_doc": (_i64, [_vp, _i64, _vp, _vp, _i64]),
    "fp_index_read_ivf": (_i64, [_vp, _i64, _vp, _i64]),
    "fp_selftest_arith": (C.c_int, [C.c_int, _vp]),
    "fp_last_search_counts": (C.c_int, [_vp, C.c_int]),
    "fp_last_s1_counts": (C.c_int, [_vp, C.c_int]),
    "fp_index_ivf_total": (_i64, [_vp]),
    "fp_index_export": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None
TORCH_LOADED_FIRST = False


def lib():
    global _lib, TORCH_LOADED_FIRST
    if _lib is None:
        import sys
        # whichever libamdhip64.so.7 is loaded first serves the whole process (see sharded.py)
        TORCH_LOADED_FIRST = "torch" in sys.modules
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built (run __graft_entry__.build() or "
                f"`make -C fast-plaid_amd/csrc`). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    return lib().fp_last_error().decode("utf-8", "replace")


def check(rc: int):
    """C error codes -> ValueError (rust/utils/errors.rs:5-7 anyhow_to_pyerr -> PyValueError)."""
    if rc != 0:
        raise ValueError(last_error())

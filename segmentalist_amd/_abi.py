"""
ctypes binding of libsegk.so (include/segk.h).  There is NO fallback: if the shared
library is missing or no MI355X is present, importing/creating a context raises.

PyTorch-ROCm tensors are used only as device buffers (`tensor.data_ptr()`), and
`torch.cuda.current_stream()` as the stream handle.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SEGK_LIB_PATH") or os.path.join(_HERE, "libsegk.so")      # (SEGK_LIB_PATH: a development build kept beside the product build)

SEGK_F32, SEGK_F64 = 0, 1
SEGK_ERR_UNSUPPORTED = -4          # include/segk.h
ABI_VERSION = 8          # SEGK_ABI_VERSION of include/segk.h this binding was written against


class SegkError(RuntimeError):
    pass


class Corpus(C.Structure):
    _fields_ = [
        ("X", C.c_void_p), ("X32", C.c_void_p), ("x_dtype", C.c_int32), ("D", C.c_int32),
        ("n_emb", C.c_int64), ("ldx", C.c_int64), ("ld32", C.c_int64), ("xnorm", C.c_void_p),
        ("vec_ids", C.c_void_p), ("durations", C.c_void_p), ("lengths", C.c_void_p),
        ("n_utt", C.c_int32), ("N_max", C.c_int32), ("Xb3", C.c_void_p), ("sp_pieces", C.c_int32), ("band_W", C.c_int32),
        ("band_ids", C.c_void_p), ("band_dur", C.c_void_p),
    ]


class KMeansDev(C.Structure):
    _fields_ = [
        ("means", C.c_void_p), ("mean_numerators", C.c_void_p), ("counts", C.c_void_p),
        ("random_means", C.c_void_p), ("assignments", C.c_void_p), ("K", C.c_void_p),
        ("K_max", C.c_int32), ("tiles", C.c_void_p), ("mnorm_max", C.c_void_p), ("tiles_b3", C.c_void_p),
    ]


class FbatchDev(C.Structure):
    _fields_ = [
        ("n_slices", C.c_int32), ("n_blocks", C.c_int32), ("u_max", C.c_int32), ("fast_dp", C.c_int32),
        ("utt_range", C.c_void_p), ("row_range", C.c_void_p), ("partials", C.c_void_p), ("cnt", C.c_void_p),
        ("mean_t", C.c_void_p), ("q_t", C.c_void_p), ("lconst", C.c_void_p), ("zconst", C.c_void_p),
        ("half", C.c_void_p), ("scal", C.c_void_p), ("slot", C.c_void_p), ("lm_tok", C.c_void_p),
        ("seed", C.c_uint64), ("y", C.c_void_p), ("ldy", C.c_int64), ("tiles32", C.c_void_p),
        ("y16", C.c_void_p), ("tiles16", C.c_void_p), ("rows32", C.c_void_p), ("consts16", C.c_void_p),
        ("prior_rows", C.c_void_p),
    ]


class CandDev(C.Structure):
    _fields_ = [("k", C.c_void_p), ("f", C.c_void_p), ("s", C.c_void_p), ("queue", C.c_void_p),
                ("count", C.c_void_p)]


class FbgmmDev(C.Structure):
    _fields_ = [
        ("cov_type", C.c_int32), ("K_max", C.c_int32), ("alpha", C.c_double), ("lms", C.c_double),
        ("k_0", C.c_double), ("v_0", C.c_double), ("prior_a", C.c_void_p), ("prior_b", C.c_void_p),
        ("prior_c", C.c_void_p), ("stat_a", C.c_void_p), ("stat_b", C.c_void_p), ("log_prod", C.c_void_p),
        ("pred", C.c_void_p), ("counts", C.c_void_p), ("assignments", C.c_void_p), ("K", C.c_void_p),
        ("lm_unigram", C.c_void_p), ("lm_bigram", C.c_void_p), ("lm_lambda", C.c_double), ("lm_a", C.c_double),
        ("lm_b", C.c_double), ("kconst", C.c_void_p),
    ]


_P = C.c_void_p
_i32, _i64, _f64, _u64 = C.c_int32, C.c_int64, C.c_double, C.c_uint64
_CP, _KP, _FP, _DP = C.POINTER(Corpus), C.POINTER(KMeansDev), C.POINTER(FbgmmDev), C.POINTER(CandDev)
_BP = C.POINTER(FbatchDev)

# name -> (restype, argtypes); every symbol include/segk.h declares
SIGNATURES = {
    "segk_create": (_i32, [_i32, C.POINTER(_P)]),
    "segk_destroy": (_i32, [_P]),
    "segk_last_error": (C.c_char_p, []),
    "segk_abi_version": (_i32, []),
    "segk_corpus_prepare": (_i32, [_P, _CP, _P, _P, _P]),
    "segk_kmeans_tiles_floats": (_i64, [_i32, _i32]),
    "segk_kmeans_prepare": (_i32, [_P, _CP, _KP, _P]),
    "segk_kmeans_mark_duplicates": (_i32, [_P, _CP, _KP, _P, _P]),
    "segk_kmeans_init_stats": (_i32, [_P, _CP, _KP, _P]),
    "segk_kmeans_score": (_i32, [_P, _CP, _KP, _P, _i64, _i64, _DP, _P, _P]),
    "segk_kmeans_score_hinted": (_i32, [_P, _CP, _KP, _P, _i64, _i64, _DP, _P, _P, _P]),
    "segk_kmeans_clear_queue": (_i32, [_P, _DP, _P]),
    "segk_kmeans_filter": (_i32, [_P, _CP, _KP, _P, _i64, _i64, _DP, _P]),
    "segk_kmeans_resolve": (_i32, [_P, _CP, _KP, _P, _i64, _i64, _DP, _P, _P]),
    "segk_kmeans_stage_counts": (_i32, [_P, _DP, C.POINTER(_i32), _P]),
    "segk_kmeans_hint_feedback": (_i32, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(_i32)]),
    "segk_kmeans_exact_max": (_i32, [_P, _CP, _KP, _P, _i64, _DP, _P, _P, _P]),
    "segk_kmeans_neg_sqrd_norm": (_i32, [_P, _CP, _KP, _i64, _P, _P]),
    "segk_kmeans_segment": (_i32, [_P, _CP, _KP, _P, _i32, _i32, _i32, _i32, _f64, _DP, _P, _P, _P, _P, _P, _P,
                                   _P, _P, _P, _P]),
    "segk_dp_tri": (_i32, [_P, _i32, _P, _P, _P, _i32, _i32, _i32, _f64, _f64, _P, _i64, _P, _i64, _P, _P, _P,
                           _P, _i64, _P]),
    "segk_kmeans_update_utt": (_i32, [_P, _CP, _KP, _i32, _P, _P, _P, _P, _P, _P, _P]),
    "segk_kmeans_sequential_sweep": (_i32, [_P, _CP, _KP, _P, _i32, _i32, _i32, _f64, _DP, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                            _P, _P]),
    "segk_kmeans_add_item": (_i32, [_P, _CP, _KP, _i64, _i32, _P, _P]),
    "segk_kmeans_del_item": (_i32, [_P, _CP, _KP, _i64, _P, _P]),
    "segk_kmeans_clean_components": (_i32, [_P, _CP, _KP, _P, _P]),
    "segk_kmeans_del_component": (_i32, [_P, _CP, _KP, _i32, _P, _P]),
    "segk_kmeans_batch_record_words": (_i64, [_i32, _i32, _i32, _i32, _i32]),
    "segk_kmeans_batch_scratch_words": (_i32, [_i32, _i64, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "segk_kmeans_batch_partials": (_i32, [_P, _CP, _KP, _P, _i32, _P, _P, _P, _P, _P, _P, _P, _i32, _i32, _P, _P]),
    "segk_kmeans_batch_finalize": (_i32, [_P, _CP, _KP, _i32, _i32, _P, _i32, _i32, _i64, _i32, _i32, _i32, _P, _P, _P, _P,
                                          _P]),
    "segk_kmeans_batch_record": (_i32, [_P, _CP, _KP, _i32, _i32, _P, _P, _P, _P, _P]),
    "segk_kmeans_assignments_from_tokens": (_i32, [_P, _CP, _KP, _i32, _i32, _P, _P, _P, _P]),
    "segk_kmeans_sum_neg_sqrd_norm": (_i32, [_P, _CP, _KP, _P, _P]),
    "segk_fbgmm_record_metrics": (_i32, [_P, _CP, _FP, _i32, _f64, _P, _P]),
    "segk_fbgmm_init_stats": (_i32, [_P, _CP, _FP, _P]),
    "segk_fbgmm_update": (_i32, [_P, _CP, _FP, _i32, _i32, _i64, _i32, _P, _P]),
    "segk_fbgmm_score": (_i32, [_P, _CP, _FP, _P, _i64, _i64, _P, _P]),
    "segk_fbgmm_pred_vector": (_i32, [_P, _CP, _FP, _i64, _P, _P]),
    "segk_unigram_segment": (_i32, [_P, _CP, _i32, _i32, _i32, _i32, _f64, _f64, _f64, _f64, _P, _P, _P, _i64, _P,
                                    _P, _P, _P, _P, _P]),
    "segk_fbgmm_assign": (_i32, [_P, _CP, _FP, _i32, _i32, _i32, _f64, _P, _P, _P, _P, _i64, _P, _P]),
    "segk_fbgmm_gibbs_items": (_i32, [_P, _CP, _FP, _P, _i64, _i32, _f64, _P, _P, _i64, _P, _P]),
    "segk_fbgmm_sequential_sweep": (_i32, [_P, _CP, _FP, _P, _i32, _P, _i32, _i32, _i32, _i32, _f64, _f64, _f64, _f64, _f64,
                                           _P, _P, _P, _i64, _P, _P, _P, _P, _P, _P]),
    "segk_fbb_collect": (_i32, [_P, _CP, _P, _P, _P, _P]),
    "segk_fbb_partials": (_i32, [_P, _CP, _FP, _BP, _i32, _i32, _i32, _P, _P, _P]),
    "segk_fbb_prepare": (_i32, [_P, _CP, _FP, _BP, _i32, _P]),
    "segk_fbb_score": (_i32, [_P, _CP, _FP, _BP, _i32, _i32, _i32, _P, _P, _P]),
    "segk_fbb_score_diag32": (_i32, [_P, _CP, _FP, _BP, _i32, _i32, _i32, _P, _P, _P]),
    "segk_calibrate_vlog": (_i32, [_P, C.POINTER(_f64), _P]),
    "segk_fbb_make_y": (_i32, [_P, _CP, _BP, _P]),
    "segk_fbb_prior_rows": (_i32, [_P, _CP, _FP, _P, _P]),
    "segk_fbb_score_f32": (_i32, [_P, _CP, _FP, _BP, _P, _i64, _P, _P]),
    "segk_fbb_segment": (_i32, [_P, _CP, _FP, _BP, _i32, _i32, _i32, _P, _u64, _i32, _i32, _f64, _f64, _f64, _P, _P,
                                _P, _P, _P, _P, _P]),
    "segk_fbb_assign": (_i32, [_P, _CP, _FP, _BP, _i32, _i32, _i32, _P, _u64, _f64, _P, _P, _P, _i64, _P]),
    "segk_fbb_assign_diag32": (_i32, [_P, _CP, _FP, _BP, _i32, _i32, _i32, _P, _u64, _f64, _P, _P, _P]),
    "segk_fbb_step_diag32": (_i32, [_P, _CP, _FP, _BP, _i32, _i32, _i32, _P, _u64, _i32, _i32, _f64, _f64, _f64, _f64,
                                    _P, _P, _P, _P, _P, _P, _P]),
    "segk_fbb_token_scores": (_i32, [_P, _CP, _FP, _BP, _P, _i64, _P, _i64, _P]),
    "segk_fbb_set_probe": (_i32, [_P, _P, _P, _i64]),
    "segk_fbb_lm_apply": (_i32, [_P, _CP, _FP, _BP, _i32, _i32, _P]),
    "segk_fbb_lm_fill": (_i32, [_P, _CP, _FP, _BP, _i32, _i32, _i32, _P, _P, _P, _P]),
    "segk_fbb_canonical": (_i32, [_P, _CP, _FP, _BP, _P, _P]),
    "segk_corpus_prepare_b3": (_i32, [_P, _CP, _P, _i32, _P]),
    "segk_corpus_b3_bytes": (_i64, [_i64, _i32]),
    "segk_kmeans_tiles_b3_floats": (_i64, [_i32, _i32]),
    "segk_graph_begin": (_i32, [_P, _P]),
    "segk_graph_end": (_i32, [_P, _P, C.POINTER(_P)]),
    "segk_graph_launch": (_i32, [_P, _P, _P]),
    "segk_graph_destroy": (_i32, [_P, _P]),
    "segk_profile_enable": (_i32, [_P, _i32]),
    "segk_profile_read": (_i32, [_P, _P, _P, _i32]),
    "segk_profile_last_kind": (_i32, [_P]),
    "segk_profile_last_launches": (_i32, [_P]),
    "segk_logsumexp": (_f64, [_P, _i64]),
    "segk_draw": (_i32, [_P, _i64, _f64]),
    "segk_sum_doubles": (_f64, [_P, _i64]),
    "segk_sum_ints": (_i64, [_P, _i64]),
    "segk_sum_log": (_f64, [_P, _i64]),
    "segk_sum_square_a_times_b": (_f64, [_P, _P, _i64]),
}

_lib = None


def build(force=False):
    """Compile libsegk.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-s"])
    if not os.path.exists(LIB_PATH):
        raise SegkError("building libsegk.so failed")


def lib():
    """Load libsegk.so; raises SegkError if it is absent (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SegkError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C segmentalist_amd/csrc`.  segmentalist_amd has no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                raise SegkError("libsegk.so does not export %s (stale build?)" % name)
            fn.restype = res
            fn.argtypes = args
        got = int(L.segk_abi_version())
        if got != ABI_VERSION:
            raise SegkError("libsegk.so reports ABI version %d, this binding is written against %d (include/segk.h "
                            "SEGK_ABI_VERSION): rebuild the library (`make -C segmentalist_amd/csrc`)" % (got, ABI_VERSION))
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise SegkError("libsegk error %d: %s" % (rc, lib().segk_last_error().decode()))


_ctx = {}


def ctx(device_index=None):
    """One segk_ctx per device, created on first use."""
    import torch
    if not torch.cuda.is_available():
        raise SegkError("no ROCm device visible: segmentalist_amd runs its hot path on MI355X only "
                        "(there is no CPU fallback)")
    if device_index is None:
        device_index = torch.cuda.current_device()
    if device_index not in _ctx:
        h = C.c_void_p()
        check(lib().segk_create(int(device_index), C.byref(h)))
        _ctx[device_index] = h
    return _ctx[device_index]


def stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())

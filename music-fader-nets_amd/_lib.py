"""ctypes binding of libfadernets_hip.so (the C ABI in include/fadernets.h).

The product path has NO fallback: if the library is missing or a symbol is absent this module
raises, and every op in hipops.py raises on a non-zero return code.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfadernets_hip.so")

FN_MAX_SCANS = 8
ABI_VERSION = 6
GEMM_LEAN = 0x10000          # FN_GEMM_LEAN
GEMM_BF16X6 = 0x20000        # FN_GEMM_BF16X6 (exact bf16 triple split on the bf16 MFMA)
GEMM_X6_WIDE = 0x80000       # FN_GEMM_X6_WIDE (128 x 256 output tiles per workgroup)
GEMM_X6_PERTILE = 0x100000   # FN_GEMM_X6_PERTILE (tests / A-B: one workgroup per tile / (tile, K range) item instead of one per CU walking its items)
GEMM_X6_PERWAVE = 0x40000    # FN_GEMM_X6_PERWAVE (tests / A-B: the round-5 kernel, every wavefront splits its own operands)
FN_E_NULL, FN_E_SHAPE, FN_E_ALIGN, FN_E_WORKSPACE, FN_E_COUNT = -1, -2, -3, -4, -5
FN_E_UNSUPPORTED = -6
FN_E_COMM = -7
FN_COMM_ID_BYTES = 128
FN_COLSUM_MAX_JOBS = 64
_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int32)
vp = C.c_void_p


class FnGruFwd(C.Structure):
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("H", C.c_int32), ("reverse", C.c_int32),
                ("w_hh_frag", vp), ("b_hh", vp), ("b_ih", vp), ("h0", vp), ("gx_dense", vp), ("gx_table", vp),
                ("idx", vp), ("idx_ld", C.c_int32), ("idx_shift", C.c_int32), ("start_token", C.c_int32),
                ("gx_rowbias", vp), ("h_all", vp), ("gates", vp), ("frag_ws", vp), ("sync_ws", vp), ("cu_budget", C.c_int32), ("h0_frag", vp), ("h_last_frag", vp), ("variant", C.c_int32), ("err_ws", vp)]


class FnGruBwd(C.Structure):
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("H", C.c_int32),
                ("w_hh_t_frag", vp), ("h0", vp), ("h_all", vp), ("gates", vp), ("dh_last", vp), ("dh_ext", vp),
                ("dgx_all", vp), ("dghn_all", vp), ("dh0", vp), ("dgx_rowsum", vp), ("dghn_rowsum", vp), ("scratch", vp), ("frag_ws", vp),
                ("sync_ws", vp), ("cu_budget", C.c_int32), ("variant", C.c_int32), ("err_ws", vp)]


class FnEmbedGrad(C.Structure):
    _fields_ = [("dgx_all", vp), ("out", vp), ("out_ld", C.c_int32), ("transposed", C.c_int32), ("reverse", C.c_int32),
                ("idx_shift", C.c_int32), ("start_token", C.c_int32)]


class FnGemmSeg(C.Structure):
    _fields_ = [("A", vp), ("lda", C.c_int32), ("B", vp), ("ldb", C.c_int32), ("K", C.c_int32)]


class FnGemmJob(C.Structure):
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("seg", FnGemmSeg * 4), ("n_seg", C.c_int32), ("beta", C.c_float), ("C", vp),
                ("ldc", C.c_int32), ("bias", vp)]


class FnColsumJob(C.Structure):
    _fields_ = [("X", vp), ("M", C.c_int32), ("N", C.c_int32), ("ld", C.c_int32), ("beta", C.c_float), ("out", vp)]


class FnGruCell(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("x", vp), ("ldx", C.c_int32), ("K1", C.c_int32), ("w_ih", vp), ("ldw_ih", C.c_int32),
                ("gx_table", vp), ("idx", vp), ("idx_ld", C.c_int32), ("start_token", C.c_int32), ("gx_rowbias", vp), ("h_prev", vp),
                ("ldh", C.c_int32), ("w_hh", vp), ("ldw_hh", C.c_int32), ("b_ih", vp), ("b_hh", vp), ("h_out", vp), ("ldo", C.c_int32), ("variant", C.c_int32),
                ("idx_best", vp), ("best_v", C.c_int32)]


class FnWeightImage(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("rows", C.c_int32), ("cols", C.c_int32), ("ld", C.c_int32), ("kind", C.c_int32)]


class FnDecode(C.Structure):
    _fields_ = [("B", C.c_int32), ("steps", C.c_int32), ("H", C.c_int32), ("V", C.c_int32), ("start_token", C.c_int32),
                ("w_hh1_frag", vp), ("b_hh1", vp), ("b_ih1", vp), ("table1", vp), ("rowbias1", vp), ("h0", vp),
                ("w_ih2_frag", vp), ("b_ih2", vp), ("w_hh2_frag", vp), ("b_hh2", vp), ("w_out_frag", vp), ("b_out", vp),
                ("tokens", vp), ("tok_ld", C.c_int32), ("logp", vp), ("ws", vp), ("sync_ws", vp)]


# name -> (restype, argtypes); every symbol declared in include/fadernets.h
SIGNATURES = {
    "fn_version": (C.c_int, []),
    "fn_strerror": (C.c_char_p, [C.c_int]),
    "fn_frag3_pack": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "fn_gemm_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "fn_gemm_f32": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp, C.c_int, vp, C.c_int,
                              C.c_float, vp, C.c_int, vp, C.c_int, vp, C.c_size_t, vp]),
    "fn_gemm_multi": (C.c_int, [C.c_int, C.c_int, C.POINTER(FnGemmJob), C.c_int, vp]),
    "fn_transpose_f32": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    "fn_colsum_ws_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "fn_colsum_f32": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp, C.c_size_t, vp]),
    "fn_colsum_multi": (C.c_int, [C.POINTER(FnColsumJob), C.c_int, vp]),
    "fn_axpy_f32": (C.c_int, [C.c_int64, C.c_float, vp, vp, vp]),
    "fn_sum_f32": (C.c_int, [vp, C.c_int64, C.c_float, vp, vp]),
    "fn_gru_gates_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "fn_frag_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "fn_frag_pack": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "fn_weight_images": (C.c_int, [C.POINTER(FnWeightImage), C.c_int, vp]),
    "fn_gru_sync_ws_bytes": (C.c_size_t, []),
    "fn_gru_seq_fwd": (C.c_int, [C.POINTER(FnGruFwd), C.c_int, vp]),
    "fn_gru_fwd_x6_ok": (C.c_int, [C.POINTER(FnGruFwd), C.c_int]),
    "fn_gru_cell_f32": (C.c_int, [C.POINTER(FnGruCell), vp]),
    "fn_out_argmax_f32": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "fn_best_tokens": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    "fn_gru_seq_bwd": (C.c_int, [C.POINTER(FnGruBwd), C.c_int, vp]),
    "fn_gru_bwd_x6_ok": (C.c_int, [C.POINTER(FnGruBwd), C.c_int]),
    "fn_decode_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "fn_decode_sync_ws_bytes": (C.c_size_t, []),
    "fn_decode_greedy": (C.c_int, [C.POINTER(FnDecode), vp]),
    "fn_gru_dwhh_ws_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "fn_gru_dwhh_f32": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int, C.c_float, vp, C.c_int, vp, C.c_size_t, vp]),
    "fn_embed_grad_ws_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "fn_embed_grad_f32": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    vp, vp, C.c_size_t, vp]),
    "fn_token_sort_ints": (C.c_size_t, [C.c_int64, C.c_int]),
    "fn_token_sort_ws_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "fn_token_sort": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_size_t, vp]),
    "fn_embed_grad_sorted_ws_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]),
    "fn_embed_grad_sorted": (C.c_int, [C.POINTER(FnEmbedGrad), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_size_t, vp]),
    "fn_time_sum_f32": (C.c_int, [vp, C.c_int, C.c_int64, vp, vp]),
    "fn_vocab_logsoftmax": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_float, vp, vp]),
    "fn_vocab_logsoftmax_bwd": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "fn_out_head_f32": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_float, vp, vp, C.c_int, vp]),
    "fn_vocab_argmax": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int64, vp, C.c_int, vp]),
    "fn_time_logsoftmax": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_float, vp, vp]),
    "fn_time_logsoftmax_bwd": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "fn_latent_fwd": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]),
    "fn_latent_bwd": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "fn_masked_prob": (C.c_int, [vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    "fn_adv_head": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp, vp, C.c_int, vp]),
    "fn_pairwise_reg": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_float, vp, vp]),
    "fn_sumsq_ws_bytes": (C.c_size_t, [C.c_int64]),
    "fn_sumsq_f32": (C.c_int, [vp, C.c_int64, vp, vp, C.c_size_t, vp]),
    "fn_step_params": (C.c_int, [vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_int, vp, vp]),
    "fn_clip_adam": (C.c_int, [vp, vp, vp, vp, C.c_int64, vp, C.c_float, vp, C.c_float, C.c_float, C.c_float, vp]),
    "fn_onehot_to_index": (C.c_int, [vp, C.c_int64, C.c_int, vp, vp]),
    "fn_comm_unique_id": (C.c_int, [vp]),
    "fn_comm_init": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, vp]),
    "fn_comm_destroy": (C.c_int, [vp]),
    "fn_comm_all_reduce_f32": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "fn_comm_all_gather": (C.c_int, [vp, vp, vp, C.c_size_t, vp]),
    "fn_comm_count": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "fn_comm_rank": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "fn_occupy_cus": (C.c_int, [C.c_int, C.c_int, C.c_longlong, vp]),
}

_lib = None


def load():
    """Load the shared library and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libfadernets_hip.so not found at %s - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C music-fader-nets_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.fn_version() != ABI_VERSION:
        raise RuntimeError("libfadernets_hip.so ABI version %d != %d (rebuild: make -C music-fader-nets_amd/csrc)" % (lib.fn_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().fn_strerror(code)
        raise RuntimeError("%s failed: %s (code %d)" % (what, msg.decode() if msg else "?", code))

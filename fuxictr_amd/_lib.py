"""ctypes binding of libfxctr.so (the C-ABI declared in include/fxctr.h).

The product path has no fallback: if the shared library is missing or a call fails, this module
raises.  Nothing here imports the oracle.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfxctr.so")

FX_OK = 0
FX_F32, FX_F64, FX_I32, FX_I64, FX_BF16 = 0, 1, 2, 3, 4
FX_FLAG_BAD_ID = 1
FX_FLAG_A2A_OVERFLOW = 2
FX_MT_BLOCKS = 96
FX_MT_MAX = 64
FX_COLSUM_CHUNKS = 64
FX_NUMGRAD_CHUNKS = 16
FX_REG_BLOCKS = 1024
FX_REG_CROSS_BLOCKS = 256
FX_PACK_MAX_COLS = 64
FX_PACKM_MAX_COLS = 96
FX_MAX_TABLES = 4
FX_CLIP_MAX_PARTS = 16

# indices of the 4-byte words of struct fx_scalars (include/fxctr.h)
SC_STEP, SC_ERR, SC_LR, SC_BETA1, SC_BETA2, SC_EPS = 0, 1, 2, 3, 4, 5
SC_BC1, SC_BC2S, SC_STEP_SIZE, SC_CLIP, SC_TOTAL_NORM, SC_MAX_NORM, SC_LOSS = 6, 7, 8, 9, 10, 11, 12
SC_REG_L1, SC_REG_L2 = 13, 14
SC_SERIES_TCAP = 15     # int32: entries of the Adam series table that follows the block (0: none)
SC_WORDS = 16

vp = C.c_void_p
i32 = C.c_int32
i64 = C.c_int64


class RowState(C.Structure):
    """struct fx_row_state"""
    _fields_ = [("table", vp), ("m", vp), ("v", vp), ("last_step", vp), ("G", vp), ("D", i32),
                ("table_dtype", i32), ("table_ld", i64), ("m_ld", i64), ("v_ld", i64), ("last_ld", i64)]


class GemmEpilogue(C.Structure):
    """struct fx_gemm_epilogue"""
    _fields_ = [("bias", vp), ("zout", vp), ("ldz", i64), ("act", i32),
                ("mul", vp), ("ldmul", i64), ("mask", vp), ("ldmask", i64),
                ("add", vp), ("ldadd", i64), ("rowsum", vp)]


class GemmProblem(C.Structure):
    """struct fx_gemm_problem"""
    _fields_ = [("transa", i32), ("transb", i32), ("M", i64), ("N", i64), ("K", i64),
                ("A", vp), ("lda", i64), ("B", vp), ("ldb", i64), ("C", vp), ("ldc", i64),
                ("epilogue", C.POINTER(GemmEpilogue)), ("split_k", i32), ("workspace", vp)]


# name -> (restype, argtypes).  This table is also what tests/test_abi.py checks against the header.
SIGNATURES = {
    "fx_abi_version": (i32, []),
    "fx_last_error": (C.c_char_p, []),
    "fx_pack_columns": (i32, [C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), i32, i64, i32, vp,
                              i64, i64, vp]),
    "fx_emb_gather_fwd": (i32, [vp, i32, vp, i64, vp, vp, vp, i32, vp, i64, vp, vp, i32, vp, i64,
                                i64, vp, i64, vp]),
    "fx_emb_seq_pool_fwd": (i32, [vp, i32, vp, i64, vp, vp, vp, vp, vp, vp, i32, vp, i64, vp, i64,
                                  vp, i64, vp]),
    "fx_dedup_workspace_bytes": (C.c_size_t, [i64]),
    "fx_dedup": (i32, [vp, i64, i64, i32, vp, vp, vp, i64, vp, C.c_size_t, vp, vp, vp, vp, vp, vp,
                       i32, i32, vp, vp]),
    "fx_dedup_sorted_runs": (i32, [vp, i32, i64, i32, i32, vp, C.c_size_t, vp, vp, vp, vp, vp, vp,
                                   vp]),
    "fx_shard_plan_workspace_ints": (i64, [i64, i32]),
    "fx_shard_plan": (i32, [vp, vp, vp, vp, i64, i32, i64, i32, vp, vp, vp, vp, i32, vp, vp, vp]),
    "fx_fill_grad_block": (i32, [C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), i32, vp, i64, vp, i64, vp]),
    "fx_owner_grad_reduce_partials": (i64, [i64]),
    "fx_owner_grad_reduce": (i32, [vp, i64, vp, vp, vp, i64, C.POINTER(vp), C.POINTER(i32),
                                   C.POINTER(i32), i32, vp, vp]),
    "fx_owner_fetch_rows": (i32, [C.POINTER(RowState), C.POINTER(i32), i32, vp, vp, vp, vp, i64, vp, i64,
                                  i32, i32, vp, vp, i32, vp]),
    "fx_scatter_rows": (i32, [vp, vp, vp, i64, i32, vp, i64, vp]),
    "fx_split_rows": (i32, [vp, i64, i64, i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), i32, vp]),
    "fx_sum_parts": (i32, [C.POINTER(vp), C.POINTER(i64), i32, vp, vp]),
    "fx_emb_grad_reduce_partials": (i64, [i64, i32]),
    "fx_emb_grad_reduce_scratch_ints": (i64, [i64]),
    "fx_emb_grad_reduce": (i32, [vp, i64, vp, i32, i32, vp, vp, vp, i64, vp, vp, vp, vp]),
    "fx_emb_grad_reduce_scaled": (i32, [vp, i64, vp, vp, vp, i64, i32, i32, vp, vp, vp, i64, vp, vp,
                                        vp, vp]),
    "fx_emb_numeric_grad": (i32, [vp, i64, vp, vp, i64, i32, i32, i64, vp, vp, vp]),
    "fx_opt_begin_step": (i32, [vp, vp]),
    "fx_adam_series_words": (i64, [i32]),
    "fx_adam_series_build": (i32, [vp, i32, vp]),
    "fx_clip_coef": (i32, [C.POINTER(vp), C.POINTER(i64), i32, vp, vp]),
    "fx_sparse_adam": (i32, [vp, vp, vp, vp, i32, vp, vp, i64, vp, vp, vp]),
    "fx_adam_catchup": (i32, [vp, vp, vp, vp, i32, vp, vp, i64, i64, i32, vp, vp]),
    "fx_sparse_sgd": (i32, [vp, vp, i32, vp, vp, i64, vp, vp, vp]),
    "fx_reg_stats": (i32, [vp, i64, vp, vp, vp]),
    "fx_reg_cross": (i32, [vp, i32, vp, vp, i64, vp, vp, vp, vp]),
    "fx_reg_dense_update": (i32, [vp, vp, vp, vp, i64, i32, i32, vp, vp]),
    "fx_mt_sqnorm": (i32, [C.POINTER(vp), C.POINTER(i64), i32, vp, vp]),
    "fx_mt_adam": (i32, [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                         C.POINTER(i64), i32, vp, vp]),
    "fx_mt_sgd": (i32, [C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), i32, vp, vp]),
    "fx_fm_fwd": (i32, [vp, i64, i32, i32, vp, vp, i64, vp]),
    "fx_fm_bwd": (i32, [vp, i64, i32, i32, vp, vp, i64, i32, i64, vp]),
    "fx_dot_interact_fwd": (i32, [vp, i64, i32, i32, i64, vp, i64, i32, vp]),
    "fx_dot_interact_bwd": (i32, [vp, i64, vp, i64, i32, i32, i32, i64, vp, i64, vp]),
    "fx_lr_fwd": (i32, [vp, vp, i64, vp, vp, i32, vp, i64, vp, i32, vp, vp, i64, vp, i64, vp]),
    "fx_gemm_f32_batch": (i32, [C.POINTER(GemmProblem), i32, vp]),
    "fx_gemm_f32": (i32, [i32, i32, i64, i64, i64, vp, i64, vp, i64, vp, i64,
                          C.POINTER(GemmEpilogue), i32, vp, vp]),
    "fx_colsum": (i32, [vp, i64, i64, i64, vp, vp, vp]),
    "fx_mask_mul": (i32, [vp, i64, vp, i64, vp, i64, i64, vp]),
    "fx_cross_bwd_prep": (i32, [vp, i64, vp, vp, vp, vp, i64, i64, i32, i32, vp]),
    "fx_sigmoid_bce": (i32, [vp, vp, i64, vp, vp, vp, vp]),
    "fx_head_train_workspace": (i64, [i64, i64]),
    "fx_head_train": (i32, [vp, i64, vp, vp, vp, i64, vp, i64, i64, i32, C.c_float, vp, vp, vp, i64, vp, vp,
                            vp, vp, vp]),
    "fx_din_concat_fwd": (i32, [vp, i64, vp, i64, i64, i64, i32, i32, vp, vp]),
    "fx_din_concat_bwd": (i32, [vp, vp, i64, vp, i64, i64, i64, i32, i32, vp, vp, i64, i64, i32, vp]),
    "fx_din_pool_fwd": (i32, [vp, vp, i64, vp, i64, i64, i64, i32, i32, vp, vp]),
    "fx_din_pool_bwd": (i32, [vp, vp, i64, vp, i64, i64, vp, i64, i32, i32, vp, vp, i64, i64, vp]),
    "fx_cin_workgroups": (i64, []),
    "fx_cin_wimg_floats": (i64, [i32, i32, i32, i32]),
    "fx_cin_pack_w": (i32, [i32, vp, vp, vp, i32, vp, vp, vp]),
    "fx_cin_fwd": (i32, [vp, i64, i32, vp, i64, i32, i32, vp, vp, i32, vp, vp, i64, i64, vp, vp]),
    "fx_cin_bwd": (i32, [vp, i64, i32, vp, i64, i32, i32, vp, i32, vp, vp, i64, vp, i64, i32, vp,
                         i64, vp, i64, i64, vp, vp]),
    "fx_binary_metrics_workspace_bytes": (C.c_size_t, [i64]),
    "fx_binary_metrics": (i32, [vp, vp, i64, vp, C.c_size_t, vp, vp, vp]),
    "fx_dice_workspace_floats": (i64, [i32]),
    "fx_dice_fwd": (i32, [vp, i64, i32, vp, C.c_float, C.c_float, i32, vp, vp, vp, vp, vp, vp]),
    "fx_dice_bwd": (i32, [vp, vp, i64, i32, vp, C.c_float, i32, vp, vp, vp, vp, vp]),
    "fx_dice_local_sums": (i32, [vp, i64, i32, vp, vp, vp]),
    "fx_dice_fwd_from_sums": (i32, [vp, i64, i32, vp, C.c_float, C.c_float, vp, i64, vp, vp, vp, vp,
                                    vp]),
    "fx_dice_bwd_local_sums": (i32, [vp, vp, i64, i32, vp, C.c_float, vp, vp, vp, vp]),
    "fx_dice_bwd_from_sums": (i32, [vp, vp, i64, i32, vp, C.c_float, vp, vp, i64, vp, vp]),
    "fx_din_attn_workspace_floats": (i64, [i64, i32, i32, i32]),
    "fx_din_attn_stats": (i32, [vp, i64, vp, i64, i64, i64, i32, i32, vp, vp, i32, vp, vp, vp, C.c_float, vp,
                                vp, vp, vp]),
    "fx_dice_stats_from_sums": (i32, [vp, i32, i64, C.c_float, i32, vp, vp, vp, vp, vp]),
    "fx_din_attn_fwd": (i32, [vp, i64, vp, i64, i64, i64, i32, i32, vp, vp, i32, vp, C.c_float, vp,
                              vp, vp, vp, i64, vp, vp, i64, vp]),
    "fx_din_attn_bwd_sums": (i32, [vp, i64, vp, i64, i64, i64, i32, i32, vp, vp, i32, vp, C.c_float,
                                   vp, vp, vp, i64, vp, i64, vp, vp, vp, vp]),
    "fx_din_attn_bwd": (i32, [vp, i64, vp, i64, i64, i64, i32, i32, vp, vp, i32, vp, C.c_float, i32,
                              vp, vp, vp, i64, vp, vp, i64, vp, vp, i64, vp, i64, i32, vp, i64, i64, vp,
                              vp, vp]),
    "fx_dedup_catchup": (i32, [vp, i64, i64, i32, vp, vp, vp, vp, C.c_size_t, vp, vp, vp, vp, vp, vp,
                               vp, C.POINTER(RowState), i32, i32, vp, vp]),
    "fx_emb_fm_fwd": (i32, [vp, i32, i32, vp, i64, vp, vp, vp, i32, vp, i64, vp, vp, i32, vp, i64, i64,
                            vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i32, i64, i32, vp]),
    "fx_emb_fm_bwd_partials": (i64, [i64, i32]),
    "fx_emb_fm_bwd_workspace_floats": (i64, [i64, i32, i32]),
    "fx_emb_fm_bwd": (i32, [vp, i64, vp, i64, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, i64, vp, vp,
                            vp, vp, vp, i64, vp, i32, i64, vp, vp, vp, vp, vp]),
    "fx_sparse_adam_multi": (i32, [C.POINTER(RowState), i32, vp, vp, i64, vp, vp]),
    "fx_sparse_sgd_multi": (i32, [C.POINTER(RowState), i32, vp, vp, i64, vp, vp]),
    "fx_adam_catchup_all": (i32, [C.POINTER(RowState), i64, i32, vp, vp]),
    "fx_adam_catchup_rows": (i32, [C.POINTER(RowState), i32, vp, vp, i64, i32, vp, vp]),
    "fx_pack_columns_multi": (i32, [C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), C.POINTER(vp),
                                    C.POINTER(i32), C.POINTER(i64), i32, i64, vp]),
}

_lib = None


class FxError(RuntimeError):
    pass


def load():
    """Load libfxctr.so once; raise loudly when it is missing (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FxError(
            "libfxctr.so not found at %s — build it with `python -m fuxictr_amd.build` "
            "(hipcc --offload-arch=gfx950). The native path has no fallback." % LIB_PATH)
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so); it must be the one already
    # resident when libfxctr.so resolves libamdhip64.so.7, or two runtimes end up in the process
    # and launches fail with "no ROCm-capable device is detected".
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    ver = lib.fx_abi_version()
    if ver != 1:
        raise FxError("libfxctr.so ABI version %d, expected 1" % ver)
    _lib = lib
    return lib


def check(status, what):
    if status != FX_OK:
        msg = load().fx_last_error()
        raise FxError("%s failed (status %d): %s" % (what, status, msg.decode() if msg else ""))


def ptr(t):
    """Device (or host) address of a tensor, or NULL."""
    return vp(0) if t is None else vp(t.data_ptr())


def ptr_array(tensors):
    arr = (vp * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


def i64_array(vals):
    arr = (i64 * len(vals))()
    for i, v in enumerate(vals):
        arr[i] = int(v)
    return arr


def i32_array(vals):
    arr = (i32 * len(vals))()
    for i, v in enumerate(vals):
        arr[i] = int(v)
    return arr


def stream_ptr(device=None):
    """hipStream_t of torch's current stream (so launches order with torch work and are
    captured by torch.cuda.graph)."""
    import torch
    return vp(torch.cuda.current_stream(device).cuda_stream)


def row_lanes(D):
    """Lanes that serve one D-float row (fx_row_geom in csrc/fx_common.h)."""
    vec = 4 if D % 4 == 0 else (2 if D % 2 == 0 else 1)
    need, lanes = D // vec, 1
    while lanes < need:
        lanes <<= 1
    return lanes

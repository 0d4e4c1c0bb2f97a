"""ctypes binding of the C-ABI hot-path library (include/dr_hotpath.h).

There is NO fallback: if libdr_hotpath.so is missing or a tensor is not on a HIP device the call
raises.  PyTorch is used only for device memory and streams; every computation on the product path
happens inside the hand-written gfx950 kernels behind this boundary.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "lib", "libdr_hotpath.so")

_i64, _i32, _u64, _f32, _p = ctypes.c_int64, ctypes.c_int32, ctypes.c_uint64, ctypes.c_float, ctypes.c_void_p

# name -> argtypes (all return int unless listed in _RESTYPE)
SIGNATURES = {
    "dr_hash_bucket_i64": [_p, _i64, _i32, _p, _p, _p],
    "dr_hash_bucket_bytes": [_p, _p, _i64, _u64, _p, _p],
    "dr_vocab_lookup_i64": [_p, _i64, _p, _i32, _p, _p],
    "dr_vocab_lookup_bytes": [_p, _p, _i64, _p, _p, _i32, _p, _p],
    "dr_emb_pool_fwd": [_p, _i64, _i32, _i32, _p, _p, _p, _i32, _p, _p, _p, _i64, _p, _p, _p],
    "dr_emb_pool_bwd": [_p, _i64, _i32, _i32, _p, _p, _i32, _p, _i64, _p, _i64, _p, _p, _f32, _p, _p, _p, _p],
    "dr_fm2_fwd": [_p, _i64, _i32, _i32, _p, _p],
    "dr_fm2_bwd": [_p, _p, _i64, _i32, _i32, _p, _p],
    "dr_linear_fwd": [_p, _i64, _p, _i64, _p, _i64, _i32, _i32, _i32, _p, _i64, _p],
    "dr_linear_bwd_dx": [_p, _i64, _p, _i64, _i64, _i32, _i32, _p, _i64, _i32, _p, _i64, _p],
    "dr_linear_bwd_dw": [_p, _i64, _p, _i64, _i64, _i32, _i32, _f32, _p, _i64, _p, _p, _i64, _p],
    "dr_emb_pool_fwd_ex": [_p, _i64, _i32, _i32, _p, _p, _p, _i32, _p, _p, _p, _i64, _p, _p, _i32, _p],
    "dr_lin_fields_fwd": [_p, _i64, _i32, _i32, _p, _p, _p, _p, _i64, _p],
    "dr_lin_fields_bwd": [_p, _i64, _i32, _i32, _p, _p, _p, _i64, _f32, _p, _p],
    "dr_emb_pool_bwd_sorted_adam": [_p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i64, _p, _i64, _p, _i64, _p, _p, _p, _f32, _f32,
                                    _f32, _f32, _p, _p, _p, _p, _p, _p, _p, _p],
    "dr_emb_pool_bwd_sorted_adam_ex": [_p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i64, _p, _i64, _p, _i64, _p, _p, _p, _f32, _f32,
                                    _f32, _f32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "dr_adam_step": [_p, _p, _p, _p, _i64, _f32, _f32, _f32, _f32, _f32, _p],
    "dr_ftrl_step": [_p, _p, _p, _p, _i64, _f32, _f32, _f32, _f32, _f32, _p],
    "dr_linear_bwd_dw_workspace_bytes": [_i64, _i32, _i32],
    "dr_linear_bwd_narrow_workspace_bytes": [_i64, _i32, _i32],
    "dr_tower_head_workspace_bytes": [_i64],
    "dr_tower_head_fwd_bwd": [_p, _i64, _p, _i64, _p, _i64, _i64, _i32, _i32, _i32, _p, _i64, _p, _p, _p, _i32, _f32, _p, _i64, _p,
                              _p, _i64, _p, _p, _p, _i64, _p, _p, _i64, _p],
    "dr_linear_bwd_narrow": [_p, _i64, _p, _i64, _p, _i64, _i64, _i32, _i32, _i32, _f32, _p, _i64, _p, _p, _i64, _p, _i64, _p],
    "dr_linear_bwd_narrow_parts": [_p, _i64, _p, _i64, _p, _i64, _i64, _i32, _i32, _i32, _f32, _p, _i64, _p, _p, _i64, _p, _i64, _i32, _p],
    "dr_linear_bwd_narrow_amax": [_p, _i64, _p, _i64, _p, _i64, _i64, _i32, _i32, _i32, _f32, _p, _i64, _p, _p, _i64, _p, _i64, _i32, _p, _p],
    "dr_tower_head_fwd_bwd_parts": [_p, _i64, _p, _i64, _p, _i64, _i64, _i32, _i32, _i32, _p, _i64, _p, _p, _p, _i32, _f32, _p, _i64, _p,
                                    _p, _i64, _p, _p, _p, _i64, _p, _p, _i64, _i32, _p],
    "dr_cross_fwd": [_p, _p, _i64, _p, _i64, _p, _f32, _i64, _i32, _p, _p, _p],
    "dr_cross_combine_bwd": [_p, _p, _p, _i64, _i32, _i64, _f32, _p, _p, _p, _p],
    "dr_bce_fwd_bwd": [_p, _p, _i64, _p, _i64, _i32, _p, _p, _p, _p, _p],
    "dr_sigmoid_fwd": [_p, _i64, _p, _p],
    "dr_sigmoid_bwd": [_p, _p, _i64, _p, _p],
    "dr_bce_prob_fwd_bwd": [_p, _p, _i64, _i32, _p, _p, _p, _p],
    "dr_shard_bucket_workspace_bytes": [_i64, _i32],
    "dr_shard_bucket_ids": [_p, _i64, _i32, _i64, _i32, _p, _p, _p, _p, _p],
    "dr_shard_bucket_ids_dedup": [_p, _p, _i64, _i32, _i64, _i32, _p, _p, _p, _p, _p],
    "dr_shard_dedup_slots": [_p, _p, _p, _i64, _i64, _p, _p],
    "dr_rows_gather": [_p, _i64, _p, _i32, _p, _p, _p, _p],
    "dr_rows_scatter_add": [_p, _i64, _p, _i32, _p, _f32, _p, _p, _p],
    "dr_axpy": [_i64, _f32, _p, _p, _p],
    "dr_emb_pack_grads": [_p, _i64, _i32, _i32, _p, _i64, _p, _i64, _p, _p, _p, _p, _p, _p],
    "dr_emb_pack_grads_dedup": [_p, _p, _i64, _i32, _i32, _p, _i64, _p, _i64, _p, _p, _p, _p, _p, _p],
    "dr_emb_sort_workspace_bytes": [_i64],
    "dr_emb_sort_slots": [_p, _i64, _i32, _p, _i64, _p, _p, _p, _p, _p, _p, _i64, _p],
    "dr_hash_sort_slots": [_p, _i64, _i32, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _i64, _p],
    "dr_emb_plan_set_small_limit": [_i32],
    "dr_adam_catchup_rows": [_p, _i64, _i32, _p, _i32, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _f32, _f32, _f32, _f32, _p],
    "dr_emb_pool_bwd_sorted": [_p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i64, _p, _i64, _p, _i64, _p, _p, _p, _f32, _p, _p,
                               _p, _p, _p],
    "dr_emb_pool_bwd_sorted_parts": [_p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i64, _p, _i64, _p, _i64, _p, _p, _p, _f32, _p, _p,
                                     _p, _p, _i32, _p],
    "dr_emb_pool_bwd_sorted_ex": [_p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i64, _p, _i64, _p, _i64, _p, _p, _p, _f32, _p, _p,
                                  _p, _p, _p, _i32, _p, _p],
    "dr_emb_lin_update_unique": [_p, _p, _i64, _i32, _p, _p, _p, _f32, _p, _p],
    "dr_emb_snapshot_sorted_rows": [_p, _p, _p, _p, _i32, _i64, _p, _p],
    "dr_ids_transpose_i32": [_p, _i64, _i32, _p, _p],
    "dr_linear_bwd_dx_fm": [_p, _i64, _p, _i64, _i64, _i32, _i32, _p, _p, _p, _i64, _i32, _i32, _p, _i64, _p],
    "dr_inbatch_softmax_workspace_bytes": [_i64],
    "dr_inbatch_softmax_fwd": [_p, _p, _i64, _i32, _p, _p, _p, _f32, _p, _p, _p, _p, _i64, _p],
    "dr_inbatch_softmax_grad_scores": [_p, _p, _i64, _i32, _p, _p, _p, _f32, _p, _f32, _p, _i64, _p],
    "dr_inbatch_softmax_grad_scores_ws": [_p, _p, _i64, _i32, _p, _p, _p, _f32, _p, _f32, _p, _i64, _p, _i64, _p],
    "dr_scores_nt": [_p, _i64, _p, _i64, _i64, _i32, _i32, _p, _i64, _p],
    "dr_topk_select": [_p, _i64, _i64, _i64, _i32, _i64, _i32, _p, _p, _p],
    "dr_topk_workspace_bytes": [_i64, _i64, _i32],
    "dr_topk_mips": [_p, _i64, _p, _i64, _i32, _i32, _i64, _i32, _p, _p, _p, _i64, _p],
    "dr_topk_index_bytes": [_i64, _i32],
    "dr_topk_index_build": [_p, _i64, _i32, _p, _i64, _p],
    "dr_topk_mips_indexed": [_p, _i64, _p, _p, _i64, _i32, _i32, _i64, _i32, _p, _p, _p, _i64, _p],
    "dr_ivf_pack": [_p, _i64, _i32, _p, _p, _p, _i32, _i64, _p, _p, _p, _p],
    "dr_ivf_scan": [_p, _i64, _i32, _p, _i32, _p, _p, _p, _i32, _p, _p, _p],
    "dr_topk_merge": [_p, _p, _i32, _p, _p, _i32, _i64, _i32, _p, _p, _p],
    "dr_rowdot": [_p, _p, _i64, _i32, _p, _p],
    "dr_rows_scale": [_p, _p, _i32, _p, _i64, _i32, _p, _p],
    "dr_gather_i64": [_p, _i64, _p, _i64, _p, _p],
    "dr_take_along_rows_f32": [_p, _i64, _i64, _i32, _p, _i32, _p, _p],
    "dr_take_along_rows_i64": [_p, _i64, _i64, _i32, _p, _i32, _p, _p],
    "dr_topk_hits": [_p, _p, _i64, _i32, _p, _i32, _p, _p],
    "dr_exclude_adjust": [_p, _p, _i64, _i32, _p, _i32, _p, _p],
    "dr_logits_adjust": [_p, _p, _i64, _i32, _p, _p, _f32, _p, _p],
    "dr_softmax_ce_rows": [_p, _p, _i64, _i32, _f32, _p, _p, _p, _p],
    "dr_softmax_ce_rows_bwd": [_p, _p, _i64, _i32, _f32, _p, _f32, _p, _p, _i64, _p],
    "dr_bf3_split": [_p, _i64, _i64, _i32, _p, _i64, _i64, _i64, _i64, _i32, _p],
    "dr_bf3_join": [_p, _i64, _i64, _i64, _i32, _p, _i64, _p],
    "dr_bf3_gemm_nt": [_p, _i64, _i64, _p, _i64, _i64, _i64, _i32, _i32, _p, _i32, _p, _i64, _p, _i64, _p],
    "dr_bf3_linear_nt": [_p, _i64, _p, _i64, _i64, _i64, _i32, _i32, _p, _i32, _p, _i64, _i32, _p, _i64, _p],
    "dr_bf3_linear_nt_pack": [_p, _i64, _p, _i64, _i64, _i64, _i32, _i32, _p, _i32, _p, _p, _p, _i64, _p, _p, _p, _p],
    "dr_h2_linear_nt_pack": [_p, _i64, _p, _p, _i64, _i64, _p, _i64, _i32, _i32, _p, _i32, _p, _p, _p, _i64, _p, _p, _p, _p],
    "dr_bf3_cross_fwd": [_p, _p, _i64, _p, _i64, _i64, _p, _f32, _i64, _i32, _p, _p, _p],
    "dr_linear_fwd_splitk_workspace_bytes": [_i64, _i32, _i32],
    "dr_linear_fwd_splitk": [_p, _i64, _p, _i64, _i64, _i32, _i32, _p, _i64, _p, _i64, _p],
    "dr_bf3_emb_linear_fwd": [_p, _i64, _i32, _p, _i64, _p, _i32, _p, _p, _p, _p, _i64, _i32, _p, _i64, _i64, _i32, _p, _i32, _p, _p, _p, _i64, _p],
    "dr_bf3_emb_linear_fwd_lv": [_p, _i64, _i32, _p, _i64, _p, _i32, _p, _p, _p, _p, _i64, _i32, _p, _i64, _i64, _i32, _p, _i32, _p, _p, _p, _i64,
                                 _p, _p],
    "dr_bf3_wgrad_workspace_bytes": [_i64, _i32, _i32],
    "dr_bf3_wgrad": [_p, _i64, _p, _i64, _i64, _i32, _i32, _f32, _p, _i64, _p, _p, _i64, _p],
    "dr_bf3_wgrad_emb": [_p, _i64, _i32, _p, _p, _i32, _p, _p, _i64, _i32, _i32, _f32, _p, _i64, _p, _p, _i64, _p],
    "dr_bf3_wgrad_emb_parts": [_p, _i64, _i32, _p, _p, _i32, _p, _p, _i64, _i32, _i32, _f32, _p, _i64, _p, _p, _i64, _i32, _p],
    "dr_bf3_gemm_tn_workspace_bytes": [_i64, _i32, _i32],
    "dr_bf3_gemm_tn": [_p, _i64, _i64, _p, _i64, _i64, _i64, _i32, _i32, _f32, _p, _i64, _p, _p, _p, _i64, _p],
    "dr_h2_amax": [_p, _i64, _i64, _i32, _p, _i32, _p],
    "dr_h2_split": [_p, _i64, _i64, _i32, _p, _i64, _i64, _i64, _i64, _i32, _p, _p],
    "dr_h2_refresh_weight": [_p, _i64, _i64, _i32, _p, _i64, _i64, _p, _i64, _i64, _p, _p, _p],
    "dr_h2_linear_nt": [_p, _i64, _p, _p, _i64, _i64, _p, _i64, _i32, _i32, _p, _i32, _p, _i64, _i32, _p, _i64, _p, _p],
    "dr_h2_cross_fwd": [_p, _p, _i64, _p, _p, _i64, _i64, _p, _p, _f32, _i64, _i32, _p, _p, _p, _p],
    "dr_h2_dgrad_emb_sgd": [_p, _i64, _p, _p, _i64, _i64, _p, _i64, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _f32, _p, _i64, _p, _p],
    "dr_cross_combine_bwd_amax": [_p, _p, _p, _i64, _i32, _i64, _f32, _p, _p, _p, _p, _p],
    "dr_h2_emb_linear_fwd": [_p, _i64, _i32, _p, _i64, _p, _i32, _p, _p, _p, _p, _p, _p, _i64, _i32, _p, _i64, _i64, _p, _i32, _p, _i32,
                             _p, _p, _p, _i64, _p, _p],
    "dr_h2_wgrad": [_p, _i64, _p, _p, _i64, _p, _i64, _i32, _i32, _f32, _p, _i64, _p, _p, _i64, _p],
    "dr_h2_wgrad_emb": [_p, _i64, _i32, _p, _p, _i32, _p, _p, _p, _p, _i64, _p, _i32, _i32, _f32, _p, _i64, _p, _p, _i64, _i32, _p],
    "dr_cin_fwd": [_p, _p, _i64, _i32, _i32, _i32, _p, _i32, _p, _i32, _p, _p],
    "dr_cin_bwd": [_p, _p, _i64, _i32, _i32, _i32, _p, _i32, _i32, _p, _p, _p, _p, _p, _p, _p],
    "dr_din_concat_fwd": [_p, _p, _i64, _i32, _i32, _p, _i64, _p],
    "dr_din_concat_bwd": [_p, _p, _i64, _i32, _i32, _p, _i64, _p, _p, _p],
    "dr_act_fwd": [_p, _i64, _i32, _i64, _i32, _p],
    "dr_act_bwd": [_p, _i64, _p, _i64, _i64, _i32, _i32, _p],
    "dr_dropout_fwd": [_p, _i64, _i64, _i32, _f32, _u64, _p, _i64, _p, _p],
    "dr_dropout_bwd": [_p, _i64, _p, _i64, _i32, _f32, _p, _i64, _p],
    "dr_reduce_sum": [_p, _i64, _i32, _f32, _i32, _p, _p, _p],
    "dr_clock_stamp": [_p, _p],
    "dr_tower_tail_workspace_bytes": [_i64, _i32],
    "dr_tower_tail_fused": [_p, _i64, _p, _i64, _p, _i64, _i64, _i32, _i32, _p, _i64, _p, _p, _p, _i32, _f32, _p, _i64, _p, _p, _i64, _p,
                            _p, _p, _p, _i64, _p, _i64, _p, _p, _i64, _i32, _p, _p],
    "dr_copy_nt": [_p, _p, _i64, _p],
    "dr_ivf_build_workspace_bytes": [_i64, _i32],
    "dr_ivf_build_lists": [_p, _i64, _i32, _p, _p, _p, _i64, _p],
    "dr_version": [],
    "dr_set_gemm_mode": [_i32],
    "dr_get_gemm_mode": [],
    "dr_set_gemm_split": [_i32],
    "dr_get_gemm_split": [],
}
_RESTYPE = {"dr_version": ctypes.c_char_p, "dr_shard_bucket_workspace_bytes": ctypes.c_int64,
            "dr_emb_sort_workspace_bytes": ctypes.c_int64, "dr_tower_tail_workspace_bytes": ctypes.c_int64, "dr_ivf_build_workspace_bytes": ctypes.c_int64,
            "dr_linear_bwd_dw_workspace_bytes": ctypes.c_int64,
            "dr_linear_fwd_splitk_workspace_bytes": ctypes.c_int64,
            "dr_bf3_gemm_tn_workspace_bytes": ctypes.c_int64,
            "dr_bf3_wgrad_workspace_bytes": ctypes.c_int64,
            "dr_linear_bwd_narrow_workspace_bytes": ctypes.c_int64,
            "dr_tower_head_workspace_bytes": ctypes.c_int64,
            "dr_inbatch_softmax_workspace_bytes": ctypes.c_int64, "dr_topk_workspace_bytes": ctypes.c_int64,
            "dr_topk_index_bytes": ctypes.c_int64}

DR_OK, DR_EINVAL, DR_ELAUNCH, DR_ESHAPE = 0, -1, -2, -3
_ERR = {DR_EINVAL: "DR_EINVAL (bad argument)", DR_ELAUNCH: "DR_ELAUNCH (HIP launch error)",
        DR_ESHAPE: "DR_ESHAPE (shape contract violated)"}

_LIB = None


class HotPathLibraryMissing(RuntimeError):
    pass


def lib():
    """Returns the loaded library; raises (never falls back) when it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise HotPathLibraryMissing(
                "HIP hot-path library not built: %s is missing. Build it with "
                "`python -m deep_recommenders_amd.build` (or __graft_entry__.build()). "
                "There is no CPU/PyTorch fallback." % SO_PATH)
        L = ctypes.CDLL(SO_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError if the .so does not export a declared symbol
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, ctypes.c_int)
        _LIB = L
    return _LIB


def check(status, what):
    if status != DR_OK:
        raise RuntimeError("%s failed: %s" % (what, _ERR.get(status, status)))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("hot-path kernels need tensors in HBM (got a %s tensor); there is no CPU fallback"
                           % t.device)
    return t.data_ptr()


_raw_stream = None


def stream_ptr():
    """The current torch stream of the current device as a raw hipStream_t.  Through torch's C entry points, not through
    torch.cuda.current_stream() (a Python object per call, ~2.5 us): every op of a training step asks once, ~22 times per step, and the
    host's launch work per step is what a busy shared host can turn into the step's bound (0.33 ms against 1.14 ms of GPU work)."""
    global _raw_stream
    if _raw_stream is None:
        import torch
        if hasattr(torch._C, "_cuda_getCurrentRawStream") and hasattr(torch._C, "_cuda_getDevice"):
            _raw_stream = (torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice)
        else:
            _raw_stream = (lambda dev: torch.cuda.current_stream(dev).cuda_stream, torch.cuda.current_device)
    return _raw_stream[0](_raw_stream[1]())

"""ctypes binding of the C-ABI library arco_amd/lib/libarco_hip.so (include/arco_hip.h).

The product path has NO CPU fallback: every op below raises if the HIP extension
is missing or a launch fails.  PyTorch only owns memory and streams.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ARCO_LIB") or os.path.join(_HERE, "lib", "libarco_hip.so")     # ARCO_LIB: an A/B build (tools/build_variant.sh)
_lib = None

_P, _I, _L, _F, _U64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_uint64

# name -> argtypes  (all return int status unless noted)
_SIGS = {
    "arco_mask_codes": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _L, _F, _F, _I, _I, _P, _P, _P, _P, _P],
    "arco_compact_rows": [_P, _L, _I, _P, _P, _P],
    "arco_masked_proto": [_P, _L, _P, _L, _I, _I, _P, _P, _P, _P],
    "arco_lv_weights": [_P, _L, _I, _I, _P, _P],
    "arco_weighted_row_sum": [_P, _L, _P, _L, _L, _I, _I, _P, _P, _P, _L, _P],
    "arco_gather_rows": [_P, _L, _I, _P, _P, _P, _L, _L, _P, _L, _P],
    "arco_weighted_row_sum_h": [_P, _L, _P, _L, _L, _I, _I, _P, _P, _P, _L, _P],
    "arco_gather_rows_h": [_P, _L, _I, _P, _P, _P, _L, _L, _P, _L, _P],
    "arco_bank_append": [_P, _L, _P, _L, _L, _I, _P, _P],
    "arco_normalize_rows": [_P, _L, _L, _I, _F, _P, _L, _P, _L, _P, _P],
    "arco_neg_multiplicity": [_P, _I, _I, _L, _L, _P, _P],
    "arco_infonce_fwd": [_P, _L, _P, _L, _P, _P, _L, _I, _I, _F, _P, _P, _P, _P],
    "arco_infonce_anchor_grad": [_P, _P, _P, _L, _P, _P, _I, _I, _F, _F, _P, _P],
    "arco_normalize_rows_pad": [_P, _L, _L, _I, _I, _F, _P, _L, _P, _P],
    "arco_nce_normalize_banks": [_P, _P, _I, _I, _I, _L, _F, _P, _P, _P],
    "arco_gemm_batched": [_P, _L, _I, _P, _I, _P, _L, _L, _I, _L, _L, _L, _I, _P, _P],
    "arco_nce_fused": [_P, _L, _P, _P, _I, _P, _L, _L, _I, _I, _P, _P, _I, _F, _P, _P, _P, _P],
    "arco_nce_anchor_grad": [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _F, _F, _P, _L, _P],
    "arco_nce_prep": [_P, _L, _P, _L, _I, _I, _F, _P, _P, _P, _P, _I, _P, _L, _L, _I, _I, _L, _P, _P],
    "arco_nce_score": [_P, _I, _I, _P, _P, _P, _I, _L, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P],
    "arco_nce_finish": [_P, _L, _P, _L, _F, _F, _P, _P, _P, _P, _P],
    "arco_nce_anchor_grad_scaled": [_P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _F, _F, _P, _L, _P],
    "arco_anchor_pix": [_P, _L, _P, _I, _P, _L, _I, _P, _P],
    "arco_scatter_add_rows": [_P, _L, _I, _P, _P, _L, _P, _F, _P, _L, _P],
    "arco_sum_scale": [_P, _I, _F, _P, _I, _P],
    "arco_pack_conv_weight": [_P, _I, _I, _I, _I, _P, _P],
    "arco_pack_many": [_P, _I, _L, _P],
    "arco_overlap_counts": [_P, _P, _L, _I, _P, _P],
    "arco_mix_unsup": [_P, _I, _P, _P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P],
    "arco_label_presence": [_P, _I, _L, _P, _P],
    "arco_window_accumulate": [_P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "arco_score_finalize": [_P, _P, _I, _L, _P, _P],
    "arco_tps_grid": [_P, _P, _I, _L, _I, _P, _P],
    "arco_grid_sample_fwd": [_P, _L, _I, _I, _I, _I, _I, _P, _I, _I, _I, _P, _L, _P],
    "arco_jitter_blur": [_P, _I, _I, _I, _I, _P, _P, _P, _P],
    "arco_quantize8": [_P, _L, _P, _P],
    "arco_field_axpb": [_P, _F, _F, _P, _F, _I, _I, _I, _I, _P, _P],
    "arco_field_smooth": [_P, _I, _I, _I, _I, _I, _P, _P, _P],
    "arco_field_resize": [_P, _I, _I, _I, _I, _I, _I, _P, _P],
    "arco_eqv_loss_fwd": [_P, _L, _P, _L, _P, _I, _L, _I, _P, _P, _P],
    "arco_eqv_loss_bwd": [_P, _L, _P, _L, _P, _I, _L, _I, _P, _P, _P, _L, _P],
    "arco_gemm_splitk": [_P, _L, _I, _P, _I, _P, _L, _L, _I, _P, _P],
    "arco_conv_fwd": [_P, _L, _I, _P, _I, _P, _L, _P, _P, _L, _P, _P, _I, _I, _I, _I, _P],
    "arco_conv_wgrad": [_P, _L, _I, _P, _L, _I, _I, _I, _I, _I, _P, _P, _I, _P],
    "arco_colsum": [_P, _L, _L, _I, _P, _P, _I, _P],
    "arco_transpose2d": [_P, _L, _I, _I, _P, _L, _P],
    "arco_bn_finalize": [_P, _P, _I, _I, _L, _F, _F, _P, _P, _P, _P, _P, _I, _I, _P, _P],
    "arco_bn_apply_deferred": [_P, _I, _P],
    "arco_chan_stats": [_P, _L, _L, _I, _P, _P, _I, _P],
    "arco_bn_act_fwd": [_P, _L, _L, _I, _P, _P, _P, _P, _F, _I, _F, _U64, _L, _P, _L, _P, _I, _P],
    "arco_bn_act_bwd": [_P, _L, _P, _L, _L, _I, _P, _P, _P, _P, _F, _I, _F, _U64, _L, _P, _P, _P, _I, _P, _L, _P, _I, _P],
    "arco_chan_stats_h": [_P, _L, _L, _I, _P, _P, _I, _P],
    "arco_bn_act_fwd_h": [_P, _L, _L, _I, _P, _P, _P, _P, _F, _I, _F, _U64, _L, _P, _L, _P, _I, _P],
    "arco_bn_act_bwd_h": [_P, _L, _P, _L, _L, _I, _P, _P, _P, _P, _F, _I, _F, _U64, _L, _P, _P, _P, _I, _P, _L, _P, _I, _P],
    "arco_colsum_h": [_P, _L, _L, _I, _P, _P, _I, _P],
    "arco_bn_act_add_fwd": [_P, _L, _L, _I, _P, _P, _P, _P, _F, _P, _L, _P, _L, _I, _P],
    "arco_bn_act_add_fwd_h": [_P, _L, _L, _I, _P, _P, _P, _P, _F, _P, _L, _P, _L, _I, _P],
    "arco_bn_act_d2s_fwd": [_P, _L, _I, _P, _P, _P, _P, _F, _P, _L, _P, _L, _I, _I, _I, _I, _P],
    "arco_bn_act_d2s_fwd_h": [_P, _L, _I, _P, _P, _P, _P, _F, _P, _L, _P, _L, _I, _I, _I, _I, _P],
    "arco_bn_act_d2s_bwd": [_P, _L, _P, _L, _I, _P, _P, _P, _P, _F, _P, _P, _P, _I, _P, _I, _I, _I, _I, _P],
    "arco_bn_act_d2s_bwd_h": [_P, _L, _P, _L, _I, _P, _P, _P, _P, _F, _P, _P, _P, _I, _P, _I, _I, _I, _I, _P],
    "arco_d2s3_add": [_P, _L, _I, _I, _I, _I, _I, _P, _L, _P, _L, _P],
    "arco_d2s3_add_h": [_P, _L, _I, _I, _I, _I, _I, _P, _L, _P, _L, _P],
    "arco_cast_h2f": [_P, _L, _P, _P],
    "arco_cast_f2h": [_P, _L, _F, _P, _P],
    "arco_gn_finalize": [_P, _P, _I, _I, _I, _I, _L, _F, _P, _P, _P],
    "arco_gn_act_bwd": [_P, _L, _P, _L, _L, _I, _P, _P, _P, _P, _F, _P, _P, _P, _I, _P, _L, _I, _I, _P],
    "arco_maxpool2_fwd": [_P, _L, _I, _I, _I, _I, _P, _L, _P],
    "arco_bn_act_pool_fwd": [_P, _L, _I, _I, _I, _I, _P, _P, _P, _P, _F, _P, _L, _P, _L, _I, _P],
    "arco_maxpool2_bwd": [_P, _L, _I, _I, _I, _I, _P, _L, _P, _L, _P],
    "arco_maxpool2_bwd_add": [_P, _L, _I, _I, _I, _I, _P, _L, _P, _L, _P, _L, _P],
    "arco_bilinear_fwd": [_P, _L, _I, _I, _I, _I, _I, _I, _P, _L, _P],
    "arco_bilinear_bwd": [_P, _L, _I, _I, _I, _I, _I, _I, _P, _L, _I, _P],
    "arco_gather_upcat_rows": [_P, _L, _I, _I, _I, _P, _L, _I, _I, _I, _P, _L, _P, _L, _P],
    "arco_scatter_upcat_rows": [_P, _L, _P, _L, _P, _L, _I, _I, _I, _P, _L, _I, _I, _I, _P],
    "arco_up_neighbors": [_P, _L, _I, _I, _I, _I, _P, _P, _P],
    "arco_lerp4_cat_rows": [_P, _L, _I, _P, _P, _L, _I, _P, _L, _P, _L, _P],
    "arco_lerp4_cat_rows_bwd": [_P, _L, _I, _P, _P, _L, _P, _L, _P, _L, _I, _P],
    "arco_s2d3": [_P, _L, _I, _I, _I, _I, _I, _P, _L, _I, _P],
    "arco_trilinear_fwd": [_P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P, _L, _P],
    "arco_trilinear_bwd": [_P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P, _L, _P],
    "arco_conv1x1_upres_fwd": [_P, _L, _I, _P, _I, _P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _I],
    "arco_conv3d_fwd": [_P, _L, _I, _P, _I, _P, _L, _P, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "arco_conv3d_wgrad": [_P, _L, _I, _P, _L, _I, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P],
    "arco_conv3d_fwd_pro": [_P, _L, _I, _P, _I, _P, _L, _P, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "arco_conv3d_wgrad_pro": [_P, _L, _I, _P, _L, _I, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P],
    "arco_gather_upcat_rows3d": [_P, _L, _I, _I, _I, _I, _P, _L, _I, _I, _I, _I, _P, _L, _P, _L, _P],
    "arco_scatter_upcat_rows3d": [_P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _P, _L, _I, _I, _I, _I, _P],
    "arco_zero_rows": [_P, _L, _I, _P, _L, _P],
    "arco_gather_upcat_rows3d_h": [_P, _L, _I, _I, _I, _I, _P, _L, _I, _I, _I, _I, _P, _L, _P, _L, _P],
    "arco_lerp8_cat_rows3d": [_P, _L, _I, _I, _I, _I, _P, _L, _I, _I, _I, _I, _P, _L, _P, _L, _P],
    "arco_lerp8_cat_rows3d_h": [_P, _L, _I, _I, _I, _I, _P, _L, _I, _I, _I, _I, _P, _L, _P, _L, _P],
    "arco_lerp8_rows3d_bwd": [_P, _L, _I, _P, _L, _P, _L, _P],
    "arco_cast_rows_f2h": [_P, _L, _I, _P, _L, _F, _P, _L, _P],
    "arco_zero_rows_h": [_P, _L, _I, _P, _L, _P],
    "arco_det_absmax": [_P, _L, _I, _L, _P, _P],
    "arco_det_scatter_rows": [_P, _L, _I, _I, _P, _P, _P, _L, _P, _L, _P, _P],
    "arco_det_finish_rows": [_P, _P, _L, _P, _L, _I, _P, _F, _P, _L, _P],
    "arco_det_clear_rows": [_P, _P, _L, _P, _L, _I, _P],
    "arco_corner_rows3d": [_P, _L, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "arco_corner_rows2d": [_P, _L, _I, _I, _I, _I, _P, _P, _P],
    "arco_row_nonzero": [_P, _L, _I, _L, _P, _P],
    "arco_put_rows": [_P, _L, _I, _P, _L, _P, _L, _P],
    "arco_fold_residual": [_P, _I, _I, _P, _P, _P],
    "arco_unfold_residual": [_P, _P, _I, _I, _P, _P],
    "arco_combine_terms": [_P, _P, _I, _P, _P],
    "arco_combine_terms_bwd": [_P, _I, _P, _P, _P],
    "arco_copy_rows": [_P, _L, _L, _I, _P, _L, _I, _P],
    "arco_nchw_to_nhwc": [_P, _I, _I, _L, _P, _L, _P],
    "arco_nhwc_to_nchw": [_P, _L, _I, _I, _L, _P, _P],
    "arco_softmax_rows": [_P, _L, _L, _I, _L, _P, _P, _P, _P, _P],
    "arco_label_onehot": [_P, _L, _I, _L, _P, _P],
    "arco_entropy_masks": [_P, _P, _P, _L, _L, ctypes.c_double, ctypes.c_double, _P, _P, _P, _P],
    "arco_entropy_masks_phase": [_I, _I, _P, _P, _P, _L, _L, ctypes.c_double, ctypes.c_double, _P, _P, _P, _P],
    "arco_sup_loss_fwd": [_P, _L, _L, _I, _P, _P, _P, _P],
    "arco_sup_loss_bwd": [_P, _L, _L, _I, _P, _P, _P, _P, _P, _L, _P],
    "arco_dice_probs_fwd": [_P, _L, _L, _I, _P, _P, _P, _P, _P],
    "arco_dice_probs_bwd": [_P, _L, _L, _I, _P, _P, _P, _P, _P, _L, _P],
    "arco_unsup_loss_fwd": [_P, _L, _I, _L, _I, _P, _P, _F, _P, _P, _P],
    "arco_unsup_loss_bwd": [_P, _L, _I, _L, _I, _P, _P, _P, _P, _L, _P],
    "arco_sgd_nesterov": [_P, _P, _P, _L, _F, _F, _F, _I, _P],
    "arco_sgd_momentum": [_P, _P, _P, _L, _F, _F, _F, _I, _P],
    "arco_ema": [_P, _P, _L, _F, _P],
}
_QUERIES = {   # plain host helpers returning sizes
    "arco_proto_ws_floats": ([_L, _I, _I], _L),
    "arco_nce_max_len": ([], _L),
    "arco_nce_score_ltiles": ([_L], _L),
    "arco_jitter_desc_bytes": ([], _L),
    "arco_conv_mblocks": ([_I, _I, _I, _I, _I, _I, _L, _I], _I),
    "arco_conv_mblocks_mma": ([_I, _I, _I, _I, _I, _I, _L, _I, _I], _I),
    "arco_conv_mblocks_pro": ([_I, _I, _I, _I, _I, _I, _L, _I, _I, _I], _I),
    "arco_conv_config": ([_I, _I, _I, _I, _I, _I, _L, _P], _I),
    "arco_conv_config_mma": ([_I, _I, _I, _I, _I, _I, _L, _I], _I),
    "arco_conv_split_ok": ([_I, _I, _I, _I, _I, _I, _L], _I),
    "arco_conv_pro_ok": ([_I, _I, _I, _I, _I, _I, _I, _L, _I, _I], _I),
    "arco_conv_sp_set": ([_I], _I),
    "arco_conv3d_fl_set": ([_I], _I),
    "arco_gemm_sp_set": ([_I, _L], _I),
    "arco_wgrad_ws_floats": ([_I, _I, _I, _L], _L),
    "arco_chan_stats_blocks": ([_L], _I),
    "arco_sel_state_bytes": ([], _L),
    "arco_sel_state_offset": ([_I], _L),
    "arco_pack_desc_bytes": ([], _L),
    "arco_bn_defer_desc_bytes": ([], _L),
    "arco_loss_slabs": ([_I], _L),
    "arco_seg_ws_doubles": ([_L, _I, _I], _L),
    # host-side native sampler replay (no GPU work)
    "arco_grid_sample": ([_P, _L, _L, _L, _I, _I, _P], _L),
    "arco_randint": ([_P, _L, _L, _L, _P], _L),
    "arco_mt_skip": ([_P, _L, _U64], _L),
    "arco_mt_pregen": ([_P, _L, _L, _I], _L),
    "arco_grid_sample_many": ([_P, _L, _I, _P, _P, _I, _I, _P, _I], _L),
    "arco_grid_sample_many_async": ([_P, _L, _I, _P, _P, _I, _I, _P, _I], _L),
    "arco_grid_sample_many_finish": ([], None),
}
EXPORTS = sorted(list(_SIGS) + list(_QUERIES))


def load():
    """Load the shared library (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"arco_amd: HIP extension {LIB_PATH} is missing - build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (make -C arco_amd/csrc). "
                "There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = args, _I
        for name, (args, res) in _QUERIES.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = args, res
        _lib = lib
    return _lib


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class ActPro(ctypes.Structure):
    """include/arco_hip.h ArcoActPro: the consumer-side activation descriptor of arco_conv3d_fwd_pro / arco_conv3d_wgrad_pro."""
    _fields_ = [("mean", _P), ("istd", _P), ("gamma", _P), ("beta", _P), ("slope", _F), ("groups", _I),
                ("drop_mode", _I), ("p", _F), ("seed", _U64), ("seed_dev", _P)]


def act_pro(mean, istd, gamma, beta, slope, groups, drop_mode, p, seed, seed_dev):
    d = ActPro(mean.data_ptr(), istd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(slope), int(groups), int(drop_mode),
               float(p), int(seed), None if seed_dev is None else seed_dev.data_ptr())
    return ctypes.byref(d)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """The current torch HIP stream of the current device as a hipStream_t.  torch.cuda.current_stream() costs
    ~9 us of Python per call (device-index plumbing); the raw accessors cost ~0.3 us - at ~500 launches per step
    that is 4 ms of host time."""
    if _raw_stream is not None and _raw_device is not None:
        return ctypes.c_void_p(_raw_stream(_raw_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name, *args):
    """Invoke a kernel entry point on the current torch HIP stream; raise on error."""
    rc = getattr(load(), name)(*args, stream())
    if rc != 0:
        raise RuntimeError(f"arco_amd: {name} failed with code {rc}")


def query(name, *args):
    return getattr(load(), name)(*args)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("arco_amd: tensors must live on the GPU (no CPU fallback); got a CPU tensor")

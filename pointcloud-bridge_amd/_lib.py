"""ctypes binding of libpcb_hip.so (C ABI: include/pcb_hip.h)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PCB_LIB") or os.path.join(_HERE, "libpcb_hip.so")  # PCB_LIB: A/B builds

_p = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_l = ctypes.c_long

# name -> argtypes, exactly the declarations of include/pcb_hip.h
SIGNATURES = {
    "pcb_version": [],
    "pcb_status_string": [_i],
    "pcb_square_distance": [_p, _p, _i, _i, _i, _p, _p],
    "pcb_fps": [_p, _i, _i, _i, _p, _p, _p],
    "pcb_ball_query": [_p, _p, _i, _i, _i, _f, _i, _p, _p],
    "pcb_ball_query2": [_p, _p, _i, _i, _i, _f, _i, _p, _f, _i, _p, _p],
    "pcb_gather_rows": [_p, _p, _i, _i, _i, _i, _p, _p],
    "pcb_gather_rows_bwd": [_p, _p, _i, _i, _i, _i, _p, _p],
    "pcb_group_points": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p],
    "pcb_group_points_bwd": [_p, _p, _i, _i, _i, _i, _i, _p, _p],
    "pcb_three_nn": [_p, _p, _i, _i, _i, _i, _p, _p, _p],
    "pcb_interpolate": [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p],
    "pcb_interpolate_bwd": [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p],
    "pcb_knn": [_p, _i, _i, _i, _i, _p, _p, _p],
    "pcb_knn_screen_workspace": [_i, _i, _i, _i],
    "pcb_knn_screened": [_p, _i, _i, _i, _i, _p, _p, _p, _p],
    "pcb_knn_xyz_workspace": [_i, _i],
    "pcb_knn_xyz": [_p, _i, _i, _i, _p, _p, _p, _p],
    "pcb_structure_features": [_p, _p, _i, _i, _i, _p, _p, _p],
    "pcb_rows_linear_f32": [_p, _p, _p, _l, _i, _i, _p, _p],
    "pcb_rows_linear_dgrad_f32": [_p, _p, _l, _i, _i, _p, _p],
    "pcb_rows_linear_wgrad_partials": [_l],
    "pcb_rows_linear_wgrad_f32": [_p, _p, _l, _i, _i, _p, _p],
    "pcb_nbr_mlp_partials": [_l],
    "pcb_nbr_mlp_stats": [_p, _p, _l, _i, _i, _p, _p, _p],
    "pcb_nbr_mlp_forward": [_p, _p, _l, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p],
    "pcb_nbr_mlp_backward_reduce": [_p, _p, _l, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "pcb_nbr_mlp_backward_apply": [_p, _p, _l, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "pcb_edge_features": [_p, _p, _i, _i, _i, _i, _p, _p],
    "pcb_edge_features_bwd": [_p, _p, _i, _i, _i, _i, _p, _p],
    "pcb_colstats_bf16": [_p, _l, _i, _p, _p],
    "pcb_colstats_f32": [_p, _l, _i, _p, _p],
    "pcb_colstats_slabs_bf16": [_p, _l, _i, _p, _i, _p],
    "pcb_colstats_slabs_f32": [_p, _l, _i, _p, _i, _p],
    "pcb_bn_act_bwd_apply_bf16": [_p, _p, _p, _p, _p, _p, _p, _l, _i, _i, _i, _p, _p],
    "pcb_bn_act_bwd_apply_f32": [_p, _p, _p, _p, _p, _p, _p, _l, _i, _i, _i, _p, _p],
    "pcb_sum_slabs": [_p, _i, _i, _p, _p],
    "pcb_rows_bn_partials": [_l, _i],
    "pcb_rows_bn_stats_f32": [_p, _l, _i, _p, _i, _p],
    "pcb_rows_bn_act_f32": [_p, _p, _p, _l, _i, _i, _p, _p],
    "pcb_rows_bn_act_bwd_reduce_f32": [_p, _p, _p, _p, _p, _p, _l, _i, _i, _p, _i, _p],
    "pcb_rows_bn_act_bwd_apply_f32": [_p, _p, _p, _p, _p, _p, _p, _l, _i, _i, _i, _p, _p],
    "pcb_scene_sum_partials": [_i],
    "pcb_scene_sum_f32": [_p, _i, _i, _i, _p, _i, _p],
    "pcb_bn_finalize": [_p, _i, _l, _l, _i, _p, _p, _p, _p, _p, _f, _f, _i, _p, _p, _p, _p, _p, _p],
    "pcb_set_concurrency_hint": [_i],
    "pcb_gemm_nt_partials": [_i, _l, _i],
    "pcb_gemm_nt_bias_bf16": [_p, _p, _p, _l, _i, _i, _p, _p],
    "pcb_gemm_nt_bias_f32": [_p, _p, _p, _l, _i, _i, _p, _p],
    "pcb_gemm_nt_bias_add_bf16": [_p, _p, _p, _p, _l, _i, _i, _p, _p],
    "pcb_prep_linear_bias_bf16": [_p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p],
    "pcb_prep_linear_bias_f32": [_p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p],
    "pcb_gemm_nt_f32out_bf16": [_p, _p, _l, _i, _i, _p, _p],
    "pcb_bn_act_bf16": [_p, _p, _p, _l, _i, _i, _p, _p],
    "pcb_bn_act_f32": [_p, _p, _p, _l, _i, _i, _p, _p],
    "pcb_bn_act_max_bf16": [_p, _p, _p, _l, _i, _i, _i, _p, _p, _p],
    "pcb_bn_act_max_f32": [_p, _p, _p, _l, _i, _i, _i, _p, _p, _p],
    "pcb_bn_act_bwd_bf16": [_p, _p, _p, _p, _p, _p, _l, _i, _i, _i, _p, _p, _p],
    "pcb_bn_act_bwd_f32": [_p, _p, _p, _p, _p, _p, _l, _i, _i, _i, _p, _p, _p],
    "pcb_bn_act_max_bwd_bf16": [_p, _p, _p, _p, _p, _p, _p, _l, _i, _i, _i, _i, _p, _p, _p],
    "pcb_bn_act_max_bwd_f32": [_p, _p, _p, _p, _p, _p, _p, _l, _i, _i, _i, _i, _p, _p, _p],
    "pcb_group_rows_bf16": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "pcb_group_rows_f32": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "pcb_group_rows_bf16_bwd": [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "pcb_group_rows_f32_bwd": [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "pcb_interpolate_bf16": [_p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _i, _p, _p],
    "pcb_interpolate_skip_bf16": [_p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _i, _p, _p, _i, _i, _p],
    "pcb_interpolate_rows_skip_f32": [_p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _i, _p, _p, _i, _i, _p],
    "pcb_interpolate_rows_f32": [_p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _i, _p, _p],
    "pcb_interp_csr_count": [_p, _i, _i, _i, _i, _p, _p],
    "pcb_interp_csr_fill": [_p, _i, _i, _i, _i, _p, _p, _p, _p],
    "pcb_interpolate_bwd_csr_bf16": [_p, _i, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p],
    "pcb_interpolate_bwd_csr_f32": [_p, _i, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p],
    "pcb_gemm_nt_bf16": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _l, _i, _i, _p, _p, _i, _p],
    "pcb_gemm_nt_f32": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _l, _i, _i, _p, _p, _i, _p],
    "pcb_gemm_tn_bf16": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _i, _l, _i, _i, _p, _p, _i, _i, _p],
    "pcb_gemm_tn_f32": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _i, _l, _i, _i, _p, _p, _i, _i, _p],
    "pcb_gemm_tn_bias_bf16": [_p, _p, _l, _i, _i, _p, _p, _i, _i, _p, _p],
    "pcb_gemm_tn_workspace": [_l, _i, _i],
    "pcb_bn_bwd_finalize": [_p, _i, _l, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p],
    "pcb_prep_weights_bf16": [_i, _p, _p],
    "pcb_prep_weights_f32": [_i, _p, _p],
    "pcb_prep_weights_zero_bf16": [_i, _p, _p, _l, _p],
    "pcb_prep_weights_zero_f32": [_i, _p, _p, _l, _p],
    "pcb_mlp_stack_wbuf_elems": [_i, _p, _i, _i],
    "pcb_mlp_stack_forward": [_i, _i, _p, _p, _p, _l, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p],
    "pcb_mlp_stack_backward": [_i, _i, _p, _p, _p, _p, _l, _i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p],
    "pcb_gather_add_partials": [_l, _i],
    "pcb_gather_add_bf16": [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _i, _p, _p],
    "pcb_scatter_dy_bf16": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p],
    "pcb_scatter_dy_slabs": [_i, _i, _i, _i],
    "pcb_segment_sum_f32": [_p, _l, _i, _i, _p, _p, _l, _p, _l, _i, _p],
    "pcb_segment_sum_bf16": [_p, _l, _i, _i, _p, _p, _l, _p, _l, _i, _p],
    "pcb_scatter_dy_csr_bf16": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _l, _p, _p],
    "pcb_dropout_rows_bf16": [_p, _l, _p, _f, _p, _p],
    "pcb_dropout_rows_f32": [_p, _l, _p, _f, _p, _p],
    "pcb_gate_bf16": [_p, _p, _p, _l, _p],
    "pcb_gate_f32": [_p, _p, _p, _l, _p],
    "pcb_gate_bwd_bf16": [_p, _p, _p, _p, _p, _l, _p],
    "pcb_gate_bwd_f32": [_p, _p, _p, _p, _p, _l, _p],
    "pcb_scene_max_workspace": [_i, _i],
    "pcb_scene_max_bf16": [_p, _i, _i, _i, _p, _p, _p, _p],
    "pcb_scene_max_f32": [_p, _i, _i, _i, _p, _p, _p, _p],
    "pcb_scene_max_bwd_bf16": [_p, _p, _i, _i, _i, _p, _p],
    "pcb_scene_max_bwd_f32": [_p, _p, _i, _i, _i, _p, _p],
    "pcb_repeat_concat_bf16": [_i, _p, _p, _p, _l, _p, _p],
    "pcb_repeat_concat_f32": [_i, _p, _p, _p, _l, _p, _p],
    "pcb_repeat_concat_bwd_bf16": [_i, _p, _p, _p, _l, _p, _p],
    "pcb_repeat_concat_bwd_f32": [_i, _p, _p, _p, _l, _p, _p],
    "pcb_attention_fwd_bf16": [_p, _i, _i, _i, _i, _f, _p, _p],
    "pcb_add_layernorm_bf16": [_p, _p, _p, _p, _p, _f, _l, _i, _p, _p, _p],
    "pcb_geglu_bf16": [_p, _l, _i, _p, _p],
    "pcb_scene_colsum_workspace": [_i, _i, _i],
    "pcb_scene_concat_bf16": [_p, _p, _i, _i, _i, _i, _p, _p],
    "pcb_scene_concat_f32": [_p, _p, _i, _i, _i, _i, _p, _p],
    "pcb_scene_colsum_bf16": [_p, _i, _i, _i, _i, _i, _p, _p, _p],
    "pcb_scene_colsum_f32": [_p, _i, _i, _i, _i, _i, _p, _p, _p],
    "pcb_timer_start": [],
    "pcb_timer_enable": [_i],
    "pcb_timer_stop": [_p, _p, _p],
    "pcb_timer_read": [_i, _p, _p, _p],
    "pcb_bwd_fused_supported": [_i, _i],
    "pcb_bwd_fused_bf16": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _p, _p, _i, _p, _p,
                           _i, _i, _p],
    "pcb_gemm_nt_red_bf16": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _l, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p, _i, _p],
    "pcb_gemm_nt_red_f32": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _l, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p, _i, _p],
    "pcb_bn_act_bwd_reduce_bf16": [_p, _p, _p, _p, _p, _p, _l, _i, _i, _p, _i, _p],
    "pcb_bn_act_bwd_reduce_f32": [_p, _p, _p, _p, _p, _p, _l, _i, _i, _p, _i, _p],
    "pcb_bn_act_max_bwd_reduce_bf16": [_p, _p, _p, _p, _p, _p, _p, _l, _i, _i, _i, _p, _i, _p],
    "pcb_bn_act_max_bwd_reduce_f32": [_p, _p, _p, _p, _p, _p, _p, _l, _i, _i, _i, _p, _i, _p],
    "pcb_prep_weights_table_bf16": [_p, _i, _l, _p],
    "pcb_prep_weights_table_f32": [_p, _i, _l, _p],
    "pcb_mlp_stack_dzbuf_elems": [_i, _i, _p, _l, _i, _i, _i],
    "pcb_dy_rows_bf16": [_p, _p, _p, _p, _p, _p, _i, _l, _i, _p, _p],
    "pcb_gemm_nt_stats_add_bf16": [_p, _p, _l, _i, _i, _p, _p, _i, _p, _i, _p, _i, _p, _p],
    "pcb_gemm_nt_stats_bf16": [_i, _p, _p, _p, _i, _p, _l, _i, _i, _p, _p, _i, _p, _p],
    "pcb_bn_finalize_centred": [_p, _i, _l, _l, _i, _p, _p, _p, _p, _p, _f, _f, _i, _p, _p, _p, _p, _p, _p, _i, _p],
    "pcb_dy_repeat_sums_bf16": [_p, _p, _p, _p, _p, _p, _i, _l, _i, _i, _p, _i, _p, _p],
    "pcb_copy_table": [_p, _i, _l, _p],
    "pcb_copy_list": [_p, _p, _p, _i, _p],
    "pcb_pad_rows_bf16": [_p, _l, _l, _i, _i, _p, _p],
    "pcb_pad_rows_f32": [_p, _l, _l, _i, _i, _p, _p],
    "pcb_cross_entropy_partials": [_l],
    "pcb_cross_entropy_fwd": [_p, _l, _p, _l, _i, _l, _p, _p, _p],
    "pcb_cross_entropy_bwd": [_p, _l, _p, _l, _i, _l, _p, _p, _p, _p],
    "pcb_bridge_loss_weights": [_p, _l, _p, _p, _i, _i, _f, _f, _p, _p, _p, _p],
    "pcb_cross_entropy_w_fwd": [_p, _l, _p, _l, _i, _l, _p, _f, _p, _p, _p],
    "pcb_cross_entropy_w_bwd": [_p, _l, _p, _l, _i, _l, _p, _f, _p, _p, _p, _p],
}

_lib = None


class PcbError(RuntimeError):
    """A libpcb_hip.so entry point returned a negative pcb_status."""


def load():
    """Load libpcb_hip.so; raises if it has not been built (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP kernels are not built. "
                "Run `python __graft_entry__.py build` (hipcc --offload-arch=gfx950). "
                "This package has no CPU or eager fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the build lacks a declared symbol
            fn.argtypes = argtypes
            fn.restype = (ctypes.c_char_p if name == "pcb_status_string"
                          else ctypes.c_long if name in ("pcb_gemm_tn_workspace", "pcb_mlp_stack_wbuf_elems", "pcb_mlp_stack_dzbuf_elems", "pcb_knn_xyz_workspace", "pcb_knn_screen_workspace",
                                                     "pcb_scene_max_workspace", "pcb_scene_colsum_workspace", "pcb_scatter_dy_slabs")
                          else ctypes.c_int)
        _lib = lib
    return _lib


def check(status, what):
    if status != 0:
        msg = load().pcb_status_string(status).decode()
        raise PcbError(f"{what}: {msg} (status {status})")

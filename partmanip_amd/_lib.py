"""ctypes loader for libpartmanip_hip.so -- the only way compute enters this package.

There is NO fallback: if the shared library is missing or a symbol cannot be bound the
import raises, and every op refuses non-GPU tensors (see ops.py).  Build the library with
`python -m partmanip_amd.build` (or `__graft_entry__.build()`).
"""
import ctypes as C
import os

# torch FIRST: its wheel bundles its own HIP runtime (torch/lib/libamdhip64.so) while this library links /opt/rocm's.  Whichever is
# loaded first serves both; loaded in the other order (this library before torch -- e.g. __graft_entry__.build() followed by
# smoke() in one process) the process ends up with kernels registered in one runtime and torch's device context in the other, and
# the first launch fails with hipErrorNoDevice (100).
import torch  # noqa: F401,E402

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PARTMANIP_HIP_LIB") or os.path.join(HERE, "lib", "libpartmanip_hip.so")   # env: A/B kernel builds

P, I, L, F, D, Z = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes); mirrors include/partmanip_hip.h one to one
# (tests/test_capi_symbols.py parses the header and checks this table and the .so against it)
SIGNATURES = {
    "pm_version": (I, []),
    "pm_gae_scan_f32": (I, [P, P, P, P, P, P, P, I, I, F, F, I, F, P]),
    "pm_moments_workspace_bytes": (Z, [L]),
    "pm_moments_f64": (I, [P, L, P, P, Z, P]),
    "pm_normalize_apply_f32": (I, [P, L, P, D, F, P]),
    "pm_gather_rows_f32": (I, [P, P, P, L, L, L, L, P]),
    "pm_linear_fwd_f32": (I, [P, L, P, L, P, P, L, I, I, I, I, P]),
    "pm_linear_bwd_data_f32": (I, [P, L, P, L, P, L, P, L, I, I, I, I, P]),
    "pm_linear_bwd_weight_workspace_bytes": (Z, [I, I, I]),
    "pm_linear_bwd_weight_f32": (I, [P, L, P, L, P, L, P, I, I, I, P, Z, P]),
    "pm_linear_fwd_group_f32": (I, [I, P, P]),
    "pm_linear_bwd_data_group_f32": (I, [I, P, P]),
    "pm_linear_chain_workspace_bytes": (C.c_size_t, [I]),
    "pm_linear_fwd_chain_f32": (I, [I, P, P, C.c_size_t, P]),
    "pm_linear_bwd_data_chain_f32": (I, [I, P, P, C.c_size_t, P]),
    "pm_linear_bwd_weight_group_f32": (I, [I, P, I, P]),
    "pm_clip_adam_group_f32": (I, [I, P, P]),
    "pm_grad_slab_sum_f32": (I, [P, P, L, L, I, P]),
    "pm_pointnet_packed_elems": (Z, []),
    "pm_pointnet_pack_weights_f32": (I, [P, P, P, P]),
    "pm_pointnet_enc_fwd_f32": (I, [P, L, I, I, I, I, P, P, P, P, P, I, P, L, P, P, I, P]),
    "pm_pointnet_packed_bf3_bytes": (Z, []),
    "pm_pointnet_pack_weights_bf3": (I, [P, P, P, P]),
    "pm_pointnet_enc_fwd_bf3": (I, [P, L, I, I, I, I, P, P, P, P, P, I, P, L, P, P, P]),
    "pm_pointnet_packed_bf6_bytes": (Z, []),
    "pm_pointnet_pack_weights_bf6": (I, [P, P, P, P]),
    "pm_pointnet_enc_fwd_bf6": (I, [P, L, I, I, I, I, P, P, P, P, P, I, P, L, P, P, P]),
    "pm_pointnet_enc_bwd_workspace_bytes": (Z, [I, I, I]),
    "pm_pointnet_enc_bwd_f32": (I, [P, L, I, I, I, I, P, P, P, P, P, I, P, L, P, P, P, P, P, P, P, P, I, P, Z, P]),
    "pm_pointnet_packed_bwd_bf6_bytes": (Z, []),
    "pm_pointnet_pack_weights_bwd_bf6": (I, [P, P, P]),
    "pm_pointnet_enc_bwd_bf6": (I, [P, L, I, I, I, I, P, P, P, P, P, P, I, P, L, P, P, P, P, P, P, P, P, P, Z, P]),
    "pm_ppo_actor_loss_fwd_bwd_f32": (I, [P, L, P, P, L, P, P, P, L, P, L, I, I, F, I, F, F, P, D, P, P, L, P, P, Z, P]),
    "pm_ppo_actor_loss_workspace_bytes": (Z, [I]),
    "pm_ppo_actor_head_supported": (I, [P, L, P, L, I, I, P, L]),
    "pm_ppo_actor_head_f32": (I, [P, L, P, L, P, I, I, P, P, L, P, P, P, L, P, L, I, I, F, I, F, F, P, D, P, P, L, P, L, P, L, P, P, Z, P, P]),
    "pm_gaussian_logp_f32": (I, [P, L, P, P, L, I, I, F, I, P, P, P]),
    "pm_gaussian_logp_bwd_f32": (I, [P, L, P, P, L, I, I, F, I, P, P, P, L, P, P]),
    "pm_action_activation_bwd_f32": (I, [P, P, P, L, F, I, P]),
    "pm_value_loss_fwd_bwd_f32": (I, [P, P, P, I, I, F, P, F, P, P, L, P]),
    "pm_value_head_workspace_bytes": (Z, []),
    "pm_value_head_supported": (I, [P, L, P, I, P, L]),
    "pm_value_head_f32": (I, [P, L, P, P, I, I, P, P, I, I, F, P, F, P, P, P, L, P, L, P, Z, P, P]),
    "pm_mse_tanh_loss_fwd_bwd_f32": (I, [P, L, P, L, I, I, F, I, F, P, P, L, P]),
    "pm_action_activation_f32": (I, [P, P, L, F, I, P]),
    "pm_clip_adam_workspace_bytes": (Z, [L]),
    "pm_clip_adam_step_f32": (I, [P, P, P, P, L, L, F, D, D, D, D, P, P, P, P, Z, P]),
    "pm_ppo_accumulate_stats_f32": (I, [P, P, I, P]),
    "pm_fps_workspace_bytes": (Z, [I, I]),
    "pm_fps_f32": (I, [P, I, I, I, I, P, P, Z, P]),
    "pm_ball_query_f32": (I, [P, P, I, I, I, F, I, P, P]),
    "pm_group_points_f32": (I, [P, P, I, I, I, I, I, P, P]),
    "pm_group_points_bwd_f32": (I, [P, P, I, I, I, I, I, P, P]),
    "pm_group_concat_f32": (I, [P, P, P, P, I, I, I, I, I, I, P, P]),
    "pm_group_concat_bwd_f32": (I, [P, P, I, I, I, I, I, I, P, P]),
    "pm_maxpool_rows_f32": (I, [P, L, I, I, P, L, P, P]),
    "pm_maxpool_rows_bwd_f32": (I, [P, L, P, L, I, I, P, P, P]),
    "pm_depth_backproject_f32": (I, [P, I, I, I, I, P, F, F, F, F, P, P, P, P]),
    "pm_depth_compact_f32": (I, [P, I, I, P, P, P]),
    "pm_fps_varlen_f32": (I, [P, I, I, I, I, P, I, P, P, Z, P]),
    "pm_fps_varlen_workspace_bytes": (Z, [I, I]),
    "pm_tsdf_select_f32": (I, [P, I, I, F, F, P, P, P]),
    "pm_gaussian_sample_f32": (I, [P, L, P, P, I, I, F, I, P, P, P, P]),
    "pm_rms_update_workspace_bytes": (Z, [I]),
    "pm_rms_update_f32": (I, [P, L, I, I, I, P, P, P, P, Z, P]),
    "pm_rms_normalize_f32": (I, [P, L, I, I, P, P, P, L, P]),
    "pm_rms_moments_f64": (I, [P, L, I, I, P, P, Z, P]),
    "pm_rms_apply_moments_f32": (I, [P, L, I, I, P, P, P, P]),
    "pm_tsdf_sparse_gather_f32": (I, [P, P, P, I, I, I, P, P]),
    "pm_im2col3d_f32": (I, [P, I, I, I, I, I, I, I, I, L, L, L, L, L, P, I, P]),
    "pm_conv3d_c1_supported": (I, [I, I]),
    "pm_conv3d_c1_wgrad_workspace_bytes": (Z, [I]),
    "pm_conv3d_c1_fwd_f32": (I, [P, I, I, I, I, I, I, I, L, L, L, L, P, P, I, I, P, L, P, P]),
    "pm_conv3d_c1_wgrad_f32": (I, [P, L, P, I, I, I, I, I, I, I, L, L, L, L, I, P, L, P, P, P, Z, P]),
    "pm_col2im3d_f32": (I, [P, I, I, I, I, I, I, I, I, L, L, L, L, L, P, I, P, I, P]),
    "pm_tsdf_integrate_f32": (I, [P, P, P, I, I, L, L, F, F, P, P]),
    "pm_voxel_grid0_f32": (I, [P, L, I, I, I, I, P, P, P, P]),
    "pm_voxel_nbr27_i32": (I, [P, L, P, I, P, I, P]),
    "pm_voxel_mirror27_i32": (I, [P, L, P, P]),
    "pm_voxel_down_count_i32": (I, [P, L, I, I, P, P, P]),
    "pm_voxel_down_build_i32": (I, [P, L, P, I, I, I, P, P, L, P, P, P, P, P, P]),
    "pm_rows_gather_f32": (I, [P, L, P, L, I, I, P, L, P]),
    "pm_sparse_conv_fwd_f32": (I, [P, L, P, L, I, I, P, L, P, P, L, I, I, P, P]),
    "pm_sparse_conv_bwd_data_f32": (I, [P, L, P, L, I, I, P, L, P, L, P, L, I, I, P, P]),
    "pm_sparse_conv_bwd_data_scatter_f32": (I, [P, L, P, L, P, L, I, I, I, P, P, I, P]),
    "pm_sparse_conv_bwd_weight_workspace_bytes": (Z, [L, I, I, I]),
    "pm_sparse_conv_bwd_weight_f32": (I, [P, L, P, L, P, L, I, I, P, L, P, I, P, P, Z, P]),
    "pm_rows_gather_bwd_f32": (I, [P, L, P, P, I, I, I, L, I, I, P, L, I, P, L, P]),
    "pm_rows_gather_bwd_mapped_f32": (I, [P, L, P, P, I, I, I, L, I, I, P, L, I, P, L, P, P]),
    "pm_rows_gather_bwd_skip_f32": (I, [P, L, P, P, I, I, I, L, I, I, P, L, P, L, P, L, P, P]),
    "pm_col_blocks_f32": (I, [P, L, P, L, L, I, I, I, I, I, I, I, I, I, P]),
    "pm_rows_uniq_i32": (I, [P, L, I, I, L, P, I, I, P, P, P, P]),
    "pm_child_sum_f32": (I, [P, L, P, I, I, I, P, L, P]),
    "pm_rowmap_scatter_i32": (I, [P, L, P, L, I, P]),
    "pm_table_rows_i32": (I, [P, L, I, P, L, P, P]),
    "pm_voxel_vcat_table_i32": (I, [P, L, I, L, P, P]),
    "pm_exclusive_scan_i32": (I, [P, I, P, P, P]),
    "pm_sa_supported": (I, [I, I, I, I]),
    "pm_sa_packed_elems": (Z, [I, I, I]),
    "pm_sa_pack_weights_f32": (I, [P, P, I, I, I, P, P]),
    "pm_sa_fwd_f32": (I, [P, P, P, P, I, I, I, I, P, L, P, P, P, P, I, I, I, P, L, P, P, P]),
    "pm_sa_bwd_workspace_bytes": (Z, [I, I, I]),
    "pm_sa_bwd_f32": (I, [P, P, P, P, I, I, I, I, P, L, P, P, P, P, I, I, I, P, L, P, P, L, P, L, P, P, P, P, P, P,
                          P, P, Z, P]),
    "pm_fps_varlen_groups": (I, [I, I, I]),
    "pm_fps_varlen_groups_cfg": (I, [I, I, I, P]),
    "pm_fps_varlen_cfg_f32": (I, [P, I, I, I, I, P, I, P, P, P, Z, P]),
    "pm_adv_normalize_workspace_bytes": (Z, [L]),
    "pm_adv_normalize_f32": (I, [P, L, F, P, Z, P]),
    "pm_mlp_fwd_f32": (I, [P, L, I, I, P, P, P, I, P, P]),
    "pm_mlp_bwd_workspace_bytes": (Z, [I, I, P]),
    "pm_mlp_bwd_f32": (I, [P, L, I, I, P, P, P, I, P, P, P, P, P, Z, P]),
    "pm_sa_packed_tile": (I, [I, I, I, P, P]),
    "pm_sa_plan_workspace_bytes": (Z, [I, I]),
    "pm_sa_plan_i32": (I, [P, P, P, I, I, I, I, I, I, P, P, P, P, P, P, Z, P]),
    "pm_sa_fwd_packed_f32": (I, [P, I, I, I, P, P, P, P, P, P, L, P, P, P, P, I, I, I, P, L, P, P, P, I, P]),
    "pm_gather_copy_f32": (I, [P, P, P, L, P]),
    "pm_sa_bwd_packed_f32": (I, [P, I, I, I, P, P, P, P, P, P, L, P, P, P, P, I, I, I, P, L, P, P, L, P, L, P, P, P,
                                 P, P, P, P, I, P, P, Z, P]),
    "pm_sa_plan_inverse_i32": (I, [P, P, I, I, I, P, P, P]),
    "pm_sa_dy_segsum_f32": (I, [P, P, P, L, I, P, L, P]),
    "pm_sa_dy_consume_supported": (I, [I, I]),
    "pm_sa_dy_consume_packed_elems": (Z, [I, I]),
    "pm_sa_dy_consume_workspace_bytes": (Z, [I, I]),
    "pm_sa_dy_consume_pack_f32": (I, [P, L, I, I, P, P]),
    "pm_sa_dy_consume_f32": (I, [P, P, P, L, I, I, P, L, P, P, L, P, L, I, P, L, P, Z, P]),
    "pm_sa_groupall_supported": (I, [I, I, I]),
    "pm_sa_groupall_packed_elems": (Z, [I, I]),
    "pm_sa_groupall_pack_f32": (I, [P, I, I, P, P]),
    "pm_sa_groupall_fwd_f32": (I, [P, I, I, I, I, P, P, P, L, P, P]),
    "pm_sa_groupall_bwd_workspace_bytes": (Z, [I, I, I]),
    "pm_sa_groupall_bwd_f32": (I, [P, L, P, L, P, P, P, I, I, I, I, P, P, P, P, Z, P]),
}



# descriptor records of the grouped entry points (field for field the typedefs of include/partmanip_hip.h)
class LinearFwdDesc(C.Structure):
    _fields_ = [("X", P), ("ldx", L), ("W", P), ("ldw", L), ("b", P), ("Y", P), ("ldy", L), ("M", I), ("N", I), ("K", I),
                ("act", I)]


class LinearBwdDataDesc(C.Structure):
    _fields_ = [("dY", P), ("lddy", L), ("W", P), ("ldw", L), ("H", P), ("ldh", L), ("dX", P), ("lddx", L), ("M", I),
                ("N", I), ("K", I), ("act", I)]


class LinearBwdWeightDesc(C.Structure):
    _fields_ = [("dY", P), ("lddy", L), ("X", P), ("ldx", L), ("dW", P), ("lddw", L), ("db", P), ("slab_stride", L),
                ("M", I), ("N", I), ("K", I), ("dy_cols", I), ("x_cols", I), ("pad_", I)]


class ClipAdamDesc(C.Structure):
    _fields_ = [("params", P), ("grads", P), ("exp_avg", P), ("exp_avg_sq", P), ("n", L), ("n_clip", L), ("extra", P),
                ("extra_stride", L), ("n_sum", L), ("n_extra", I), ("max_norm", F), ("lr", D), ("b1", D), ("b2", D),
                ("eps", D), ("state", P), ("skip_flag", P), ("gnorm_out", P), ("workspace", P), ("stats_acc", P), ("stats_scal", P),
                ("stats_which", I), ("grad_scale", F), ("dp_kl_desired", F), ("dp_scal", P)]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: the HIP extension has not been built. Run `python -m partmanip_amd.build` "
        "(needs hipcc; cross-compiles gfx950 without a GPU). partmanip_amd has no CPU fallback.")

lib = C.CDLL(LIB_PATH)
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here = symbol missing from the .so: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args

ABI_VERSION = 152                      # == PM_ABI_VERSION in include/partmanip_hip.h (checked by tests/test_capi_symbols.py)
if lib.pm_version() != ABI_VERSION:
    raise ImportError(f"{LIB_PATH} is stale: it reports ABI {lib.pm_version()}, this package needs {ABI_VERSION}. "
                      "Rebuild it with `python -m partmanip_amd.build`.")

ERRORS = {-1: "PM_EINVAL (bad argument)", -2: "PM_EWORKSPACE (workspace too small)", -3: "PM_EALIGN (misaligned pointer)",
          -4: "PM_EUNSUPPORTED (no fused instantiation for this shape)"}


def check(rc, what):
    if rc != 0:
        msg = ERRORS.get(rc, f"HIP error {-rc - 1000}" if rc <= -1000 else f"code {rc}")
        raise RuntimeError(f"{what} failed: {msg}")

"""ctypes binding of ``libpapc_hip.so`` (the C ABI declared in ``include/papc_hip.h``).

There is NO fallback: if the shared library is missing or a call fails, this module raises.  Build it with
``python -m papc_amd.build`` (or ``__graft_entry__.build()``).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PAPC_LIB") or os.path.join(_HERE, "libpapc_hip.so")   # PAPC_LIB: A/B a second build of the library

ABI_VERSION = 8          # include/papc_hip.h: PAPC_ABI_VERSION

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_l = ctypes.c_int64
c_f = ctypes.c_float


class PapcError(RuntimeError):
    pass


class GroupSrc(ctypes.Structure):
    """papc_group_src"""
    _fields_ = [("xyz", c_p), ("sb", c_l), ("sn", c_l), ("sc", c_l), ("new_xyz", c_p), ("feats", c_p),
                ("idx", c_p), ("N", c_i), ("S", c_i), ("K", c_i), ("D", c_i), ("xyz_first", c_i),
                ("cidx", c_p), ("seg_grp", c_p), ("rows_dev", c_p), ("wstat", c_p), ("plists", c_p)]


class PointLists(ctypes.Structure):
    """papc_point_lists"""
    _fields_ = [("prange", c_p), ("prow", c_p), ("pmeta", c_p), ("compact", c_i), ("pmom", c_p)]


class BwdDy(ctypes.Structure):
    """papc_bwd_dy"""
    _fields_ = [("dz_mode", c_i), ("dz", c_p), ("gout", c_p), ("argmax", c_p), ("K", c_i), ("y", c_p),
                ("mean", c_p), ("invstd", c_p), ("scale", c_p), ("shift", c_p), ("c1", c_p), ("c2", c_p),
                ("wrow", c_p), ("seg_grp", c_p), ("rows_dev", c_p), ("psel", c_p)]


class GroupMax(ctypes.Structure):
    """papc_group_max"""
    _fields_ = [("gmax", c_p), ("gmin", c_p), ("amax", c_p), ("amin", c_p), ("K", c_i), ("sign_src", c_p)]


class BwdRed(ctypes.Structure):
    """papc_bwd_red"""
    _fields_ = [("y", c_p), ("mean", c_p), ("invstd", c_p), ("scale", c_p), ("shift", c_p), ("red_partial", c_p), ("store_masked", c_i)]


class CopyJob(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("B", ctypes.c_int32), ("R", ctypes.c_int32), ("C", ctypes.c_int32),
                ("sb", ctypes.c_int64), ("sr", ctypes.c_int64), ("sc", ctypes.c_int64), ("db", ctypes.c_int64), ("dr", ctypes.c_int64),
                ("dc", ctypes.c_int64)]


class ReduceJob(ctypes.Structure):
    """papc_reduce_job"""
    _fields_ = [("partial", c_p), ("n_chunks", ctypes.c_int32), ("accumulate", ctypes.c_int32), ("ld", c_l), ("n1", c_l), ("n2", c_l),
                ("out1", c_p), ("out2", c_p)]


class ScatterDst(ctypes.Structure):
    """papc_scatter_dst"""
    _fields_ = [("grad_feats", c_p), ("idx", c_p), ("N", c_i), ("S", c_i), ("K", c_i), ("D", c_i), ("col0", c_i)]


# every exported symbol of include/papc_hip.h: name -> (restype, argtypes)
SIGNATURES = {
    "papc_version": (c_i, []),
    "papc_abi_version": (c_i, []),
    "papc_abi_sizeof": (c_l, [ctypes.c_char_p]),
    "papc_last_error_string": (ctypes.c_char_p, []),
    "papc_fps_f32": (c_i, [c_p, c_l, c_l, c_l, c_i, c_i, c_i, c_p, c_f, c_p, c_p, c_p]),
    "papc_ball_query_f32": (c_i, [c_p, c_l, c_l, c_l, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_p]),
    "papc_square_distance_f32": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p]),
    "papc_index_points_f32": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "papc_index_points_bwd_f32": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "papc_group_points_f32": (c_i, [c_p, c_l, c_l, c_l, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "papc_group_points_bwd_f32": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "papc_three_nn_f32": (c_i, [c_p, c_l, c_l, c_l, c_p, c_l, c_l, c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "papc_three_interpolate_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "papc_three_interpolate_bwd_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "papc_three_interpolate_bwd_first3_f32": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "papc_mlp_gemm_parts": (c_i, [c_l]),
    "papc_mlp_gemm_gmax_ok": (c_i, [c_l, c_i, c_i]),
    "papc_mlp_gemm_f32": (c_i, [c_i, c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p, c_p, c_p, c_p]),
    "papc_mlp_gemm_rows_f32": (c_i, [c_i, c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "papc_mlp_gemm_rows_w_f32": (c_i, [c_i, c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "papc_sa_mlp_plan": (c_i, [c_p, c_p, c_p]),
    "papc_sa_mlp_fwd": (c_i, [c_p, c_p, c_p]),
    "papc_sa_mlp_bwd": (c_i, [c_p, c_p, c_p, c_p]),
    "papc_pfn_workspace": (c_i, [c_p, c_p, c_p]),
    "papc_pfn_fwd": (c_i, [c_p, c_p, c_p]),
    "papc_pfn_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p]),
    "papc_compact_plan_f32": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "papc_compact_corr_parts": (c_i, []),
    "papc_mlp_compact_ok": (c_i, [c_l, c_i, c_i, c_p]),
    "papc_bn_stats_corr_f32": (c_i, [c_p, c_i, c_p, c_p, c_i, c_p, c_p]),
    "papc_bn_relu_max_seg_f32": (c_i, [c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p]),
    "papc_bn_select_max_f32": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_p, c_p, c_p]),
    "papc_bn_finalize_f32": (c_i, [c_p, c_i, c_l, c_i, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "papc_bn_relu_max_f32": (c_i, [c_p, c_p, c_p, c_l, c_i, c_i, c_p, c_p, c_p]),
    "papc_bn_relu_f32": (c_i, [c_p, c_p, c_p, c_l, c_i, c_p, c_p]),
    "papc_bn_bwd_reduce_max_f32": (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p, c_p, c_p]),
    "papc_bn_bwd_reduce_f32": (c_i, [c_i, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p, c_p]),
    "papc_bn_bwd_finalize_f32": (c_i, [c_p, c_i, c_l, c_i, c_p, c_p, c_p, c_p, c_i, c_p]),
    "papc_bn_eval_consts_f32": (c_i, [c_p, c_p, c_p, c_p, c_f, c_i, c_p, c_p, c_p, c_p, c_p]),
    "papc_group_max_bwd_f32": (c_i, [c_p, c_p, c_l, c_i, c_i, c_p, c_p]),
    "papc_mlp_bwd_dx_f32": (c_i, [c_p, c_p, c_l, c_i, c_i, c_p, c_p, c_p, c_p]),
    "papc_mlp_bwd_dw_f32": (c_i, [c_p, c_i, c_p, c_l, c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_p, c_p, c_l, c_p]),
    "papc_reduce_partials2_f32": (c_i, [c_p, c_i, c_l, c_l, c_p, c_l, c_p, c_i, c_p]),
    "papc_reduce_partials_batch_f32": (c_i, [c_p, c_i, c_p]),
    "papc_fold_jobs_f32": (c_i, [c_p, c_i, c_p]),
    "papc_mlp_bwd_dw_chunk_hint": (c_i, [c_l, c_i, c_i, c_i, c_i, c_i]),
    "papc_reduce_partials_f32": (c_i, [c_p, c_i, c_l, c_p, c_i, c_p]),
    "papc_pfn_decorate_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_f, c_f, c_f, c_f, c_i, c_p, c_p]),
    "papc_pfn_decorate_nf_f32": (c_i, [c_p, c_i, c_p, c_p, c_i, c_i, c_f, c_f, c_f, c_f, c_i, c_p, c_p]),
    "papc_pfn_stats_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_f, c_f, c_f, c_f, c_p, c_i, c_p, c_p, c_p]),
    "papc_pfn_apply_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_f, c_f, c_f, c_f, c_p, c_i, c_p, c_p, c_p, c_p, c_p]),
    "papc_pfn_num_blocks": (c_i, [c_i]),
    "papc_pfn_bwd_reduce_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_f, c_f, c_f, c_f, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "papc_pfn_bwd_dw_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_f, c_f, c_f, c_f, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "papc_pfn_gram_blocks": (c_i, [c_i]),
    "papc_pfn_gram_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_f, c_f, c_f, c_f, c_p, c_p]),
    "papc_pfn_gram_finalize_f32": (c_i, [c_p, c_i, c_l, c_p, c_i, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "papc_pfn_bwd_sparse_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_f, c_f, c_f, c_f, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "papc_pfn_bwd_finalize_f32": (c_i, [c_p, c_l, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p]),
    "papc_points_to_voxel_workspace": (ctypes.c_size_t, [c_i]),
    "papc_points_to_voxel_f32": (c_i, [c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, ctypes.c_size_t, c_p]),
    "papc_pillar_scatter_f32": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p]),
    "papc_pillar_scatter_bwd_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "papc_head_fc_f32": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_f, c_p, c_p, c_p, c_f, c_p, c_i, c_p, c_p, c_p, c_p,
                               c_p, c_p, c_p]),
    "papc_head_bwd_f32": (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_f, c_i, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_p]),
    "papc_softmax_xent_f32": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p, c_p]),
    "papc_head_chain_fwd_f32": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "papc_head_chain_bwd_f32": (c_i, [c_p, c_i, c_i, c_p, c_p]),
    "papc_nms_workspace": (ctypes.c_size_t, [c_i]),
    "papc_nms_f32": (c_i, [c_p, c_i, c_f, c_p, c_p, c_p, ctypes.c_size_t, c_p]),
    "papc_lingather_parts": (c_i, [c_l]),
    "papc_lingather_fwd_f32": (c_i, [c_p, c_p, c_i, c_p, c_i, c_i, c_p, c_i, c_p, c_p, c_p]),
    "papc_lingather_bwd_f32": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p, c_p]),
    "papc_lingather_bwd_pp_f32": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p]),
    "papc_lingather_list_parts": (c_i, [c_l]),
    "papc_lingather_bwd_parts": (c_i, [c_p, c_i, c_i]),
    "papc_lingather_bwd_lists_ok": (c_i, [c_p, c_i]),
    "papc_point_lists_f32": (c_i, [c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "papc_lingather_bwd_pp_ok": (c_i, [c_p, c_i, c_i]),
    "papc_bn_max_prep_f32": (c_i, [c_p] * 10 + [c_l, c_i, c_i] + [c_p] * 6),
    "papc_mlp_max_nostore_ok": (c_i, [c_l, c_i, c_i, c_i]),
    "papc_mlp_bwd_dw_max_ws_floats": (c_l, [c_l, c_i, c_i]),
    "papc_mlp_bwd_dw_max_f32": (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p, c_p, c_i, c_p]),
    "papc_mlp_bwd_dx_xyz_ok": (c_i, [c_l, c_i, c_i]),
    "papc_mlp_bwd_dx_xyz_f32": (c_i, [c_p, c_p, c_l, c_i, c_i, c_p, c_p, c_p, c_p]),
    "papc_mlp_bwd_dx_max_f32": (c_i, [c_p, c_p, c_i, c_p, c_l, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p, c_p, c_p]),
    "papc_mlp_xyz_ok": (c_i, [c_l, c_i, c_i]),
    "papc_xyz_parts": (c_i, [c_l]),
    "papc_xyz_bwd_parts": (c_i, [c_l]),
    "papc_xyz_group_f32": (c_i, [c_p, c_i, c_p, c_p, c_p]),
    "papc_xyz_gram_fold_f32": (c_i, [c_p, c_i, c_p, c_p]),
    "papc_xyz_l1_finalize_f32": (c_i, [c_p, c_i, c_l, c_p, c_i, c_i, c_p, c_p, c_p, c_f, c_f, c_i] + [c_p] * 9),
    "papc_xyz_l1_bwd_f32": (c_i, [c_p, c_p, c_p, c_l, c_i, c_p, c_p]),
    "papc_xyz_l1_bwd_finalize_f32": (c_i, [c_p, c_i, c_l, c_i, c_p, c_p, c_i, c_i] + [c_p] * 7 + [c_i, c_p]),
    "papc_pg_planes_bytes": (ctypes.c_size_t, [c_l, c_l]),
    "papc_pg_prep_weights_f32": (c_i, [c_p, c_i, c_p]),
    "papc_pg_prep_rows_f32": (c_i, [c_p, c_p]),
    "papc_pg_gemm_f32": (c_i, [c_p, c_p]),
    "papc_pg_final_f32": (c_i, [c_p, c_i, c_l, c_i, c_p, c_p, c_f, c_f] + [c_p] * 10 + [c_l, c_p, c_p, c_p]),
    "papc_pg_final_groups_f32": (c_i, [c_p, c_i, c_l, c_i, c_p, c_p, c_f, c_f] + [c_p] * 10 + [c_l, c_i, c_p, c_p, c_p, c_p]),
    "papc_pg_fold_f32": (c_i, [c_p, c_i, c_p]),
    "papc_transpose_batch_f32": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p]),
    "papc_transpose_batch_ld_f32": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p]),
    "papc_rotate_nms_f32": (c_i, [c_p, c_i, c_f, c_p, c_p, c_p, ctypes.c_size_t, c_p]),
    "papc_rotate_iou_f32": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p]),
    "papc_rbbox_iou_f32": (c_i, [c_p, c_p, c_p, c_f, c_i, c_i, c_p, c_p]),
    "papc_riou_f32": (c_i, [c_p, c_p, c_f, c_i, c_i, c_p, c_p]),
    "papc_fill_f32": (c_i, [c_p, c_l, c_f, c_p]),
    "papc_copy_strided_batch_f32": (c_i, [c_p, c_i, c_p]),
    "papc_copy2d_f32": (c_i, [c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_p]),
    "papc_reduce_partials_strided_f32": (c_i, [c_p, c_i, c_l, c_i, c_i, c_p, c_l, c_i, c_p]),
    "papc_scale_by_f32": (c_i, [c_p, c_p, c_l, c_p, c_p]),
    "papc_adam_step_f32": (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, ctypes.c_double, ctypes.c_double, c_f, c_f, c_i, c_f, c_p]),
    "papc_adam_step_zero_f32": (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, ctypes.c_double, ctypes.c_double, c_f, c_f, c_i, c_f, c_p]),
    "papc_adam_tick": (c_i, [c_p, c_p]),
    "papc_flag_set": (c_i, [c_p, ctypes.c_uint32, c_p, c_p]),
    "papc_flag_wait": (c_i, [c_p, c_l, c_p]),
    "papc_flag_wait_slot": (c_i, [c_p, c_i, c_l, c_p]),
    "papc_adam_step_dev_f32": (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, ctypes.c_double, ctypes.c_double, c_f, c_f, c_p, c_f, c_i, c_p]),
    "papc_knob_set": (c_i, [ctypes.c_char_p, c_i]),
    "papc_knob_get": (c_i, [ctypes.c_char_p, ctypes.POINTER(c_i)]),
    "papc_prof_enable": (c_i, [ctypes.c_uint]),
    "papc_prof_reset": (c_i, []),
    "papc_prof_read": (c_i, [c_i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_l)]),
}

_lib = None


def load():
    """Return the loaded library with typed signatures; raises PapcError when it is not built."""
    global _lib
    if _lib is None:
        # torch must own the process's HIP runtime: it bundles its own libamdhip64, and loading ours first would pull
        # in /opt/rocm's copy as a second, device-less runtime ("no ROCm-capable device is detected").
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise PapcError("libpapc_hip.so not found at %s -- build it with `python -m papc_amd.build` "
                            "(there is no CPU fallback)" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        missing = [n for n in SIGNATURES if not hasattr(lib, n)]
        if missing:  # the header and the library drifted apart: refuse to run on half a library
            raise PapcError("libpapc_hip.so lacks symbols declared in include/papc_hip.h: %s" % ", ".join(missing))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.papc_abi_version() != ABI_VERSION:   # the structs this package mirrors by hand would not match the library's
            raise PapcError("libpapc_hip.so speaks ABI %d, this package ABI %d (include/papc_hip.h: PAPC_ABI_VERSION): rebuild with "
                            "`python -m papc_amd.build`" % (lib.papc_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


def check(status, what):
    if status != 0:
        msg = load().papc_last_error_string().decode("utf-8", "replace")
        raise PapcError("%s failed (%d): %s" % (what, status, msg))


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()


def zeros(shape, device):
    """float32 zeros written by this library's fill kernel (torch.zeros would launch a library kernel)"""
    import torch
    t = torch.empty(shape, device=device, dtype=torch.float32)
    if t.numel():
        check(load().papc_fill_f32(t.data_ptr(), t.numel(), 0.0, stream_ptr()), "papc_fill_f32")
    return t


_CONSTS = {}


def _capturing():
    import torch
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def const_zeros(shape, device):
    """cached read-only float32 zeros of a small fixed shape (e.g. the [B,1,3] centroid of sample_and_group_all)"""
    import torch
    key = ("z", tuple(shape), str(device))
    t = _CONSTS.get(key)
    if t is None:
        if _capturing():     # a tensor born inside a hipGraph capture lives in that graph's pool and is only filled on replay: never cache it
            return torch.zeros(tuple(shape), device=device, dtype=torch.float32)
        t = _CONSTS[key] = torch.zeros(tuple(shape), device=device, dtype=torch.float32)
    return t


def const_idx3(B, N, device):
    """read-only int32 [B, N, 3] = (0, 1, 2) per row: the neighbour indices the reference's sort-then-argsort yields
    (pointnet2_basic_layers.py:316-317, see layers.PointNetFeaturePropagation).  ONE buffer per device, grown to the largest B*N seen
    and handed out as a narrowed contiguous view (segmentation over varying batch / point counts does not accumulate one tensor per shape).
    A buffer that has been handed out is NEVER freed: a captured hipGraph may have baked its address into a three_interpolate launch, so a
    superseded buffer moves to a keep-alive list (growth is geometric: at most ~log2 of them, together below 2x the largest)."""
    import torch
    key = ("idx3", str(device))
    rows = int(B) * int(N)
    t = _CONSTS.get(key)
    if t is None or t.shape[0] < rows:
        mk = lambda n: torch.arange(3, device=device, dtype=torch.int32).expand(n, 3).contiguous()
        if _capturing():
            return mk(rows).view(B, N, 3)
        if t is not None:
            _CONSTS.setdefault(("idx3-retired", str(device)), []).append(t)
            rows_alloc = max(rows, 2 * int(t.shape[0]))
        else:
            rows_alloc = rows
        t = _CONSTS[key] = mk(rows_alloc)
    return t[:rows].view(B, N, 3)


def const_vec(value, n, device):
    """cached read-only float32 vector of n copies of value (BN 'identity' constants and the like): filled once per device"""
    import torch
    key = (float(value), int(n), str(device))
    t = _CONSTS.get(key)
    if t is None:
        if _capturing():
            return torch.full((n,), float(value), device=device, dtype=torch.float32)
        t = _CONSTS[key] = torch.full((n,), float(value), device=device, dtype=torch.float32)
    return t

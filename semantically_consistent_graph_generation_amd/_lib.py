"""ctypes binding of libscgg.so (C ABI: include/scgg.h).  Fails loudly when the library is missing."""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SCGG_LIB_PATH") or os.path.join(_HERE, "libscgg.so")     # the override: timing builds (scripts/probe)

c_f32p = C.c_void_p
_i32, _i64, _vp, _sz = C.c_int32, C.c_int64, C.c_void_p, C.c_size_t


class DenoiserConfig(C.Structure):
    _fields_ = [(k, C.c_int32) for k in (
        "n_layers", "in_X", "in_E", "in_y", "out_X", "out_E", "out_y", "hid_X", "hid_E", "hid_y",
        "dx", "de", "dy", "n_head", "dim_ffX", "dim_ffE", "dim_ffy")]


class Denoiser3DConfig(C.Structure):
    _fields_ = [(k, C.c_int32) for k in (
        "n_layers", "in_X", "in_E", "in_y", "out_X", "out_E", "hid_X", "hid_E", "hid_y",
        "dx", "de", "dy", "n_head", "dim_ffX", "dim_ffE", "dim_ffy", "equivariant")]


class SitConfig(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("in_channels", "hidden", "n_blocks", "n_heads", "mlp_hidden", "noise_steps")] + \
               [("beta_start", C.c_float), ("beta_end", C.c_float)]


class Rng(C.Structure):
    _fields_ = [("u0", C.c_void_p), ("uk", C.c_void_p), ("seed", C.c_uint64), ("step", C.c_uint32),
                ("graph_id_base", C.c_uint32)]


class TrainEdgeArgs(C.Structure):       # scgg_train_edge_args (include/scgg.h)
    _PTRS = ("E", "Q", "K", "V", "yea", "yem1", "mask", "F_mul", "F_add", "F_out", "F_ff1", "F_ff2", "F_mulT", "F_addT", "F_outT",
             "F_ff1T", "F_ff2T", "b_mul", "b_add", "b_out", "g1", "b1", "b_ff1", "b_ff2", "g2", "b2", "E_out", "part", "wV_out",
             "lse_out", "dOut", "dwV", "wV", "lse", "dE", "G_e", "G_da", "G_z", "G_dnewE", "G_h1", "G_df", "G_f", "G_do", "G_dkv",
             "upart", "st_mean", "st_cm", "st_cv", "st_gmi", "st_gma", "st_imn", "st_imx")
    _fields_ = [(k, C.c_void_p) for k in _PTRS] + [(k, C.c_int32) for k in ("B", "n", "nch", "dx", "de", "ffe", "b0")] + \
               [("inv_sqrt_df", C.c_float), ("p", C.c_float)] + [(k, C.c_uint32) for k in ("k0", "k1", "site1", "site2", "site3")] + \
               [("key_dev", C.c_void_p)]


class TrainYArgs(C.Structure):          # scgg_train_y_args (include/scgg.h)
    _PTRS = ("hy", "xstats", "estats", "W_yy", "b_yy", "W_xy", "b_xy", "W_ey", "b_ey", "W_o0", "b_o0", "W_o2", "b_o2", "g1", "be1",
             "W_1", "b_1", "W_2", "b_2", "g2", "be2", "out", "dout", "dhy", "dxstats", "destats", "save")
    _fields_ = [(k, C.c_void_p) for k in _PTRS] + [(k, C.c_int32) for k in ("B", "dy", "dx4", "de4", "ffy")] + [("p", C.c_float)] + \
               [(k, C.c_uint32) for k in ("k0", "k1", "site1", "site2", "site3")] + [("key_dev", C.c_void_p), ("fsave", C.c_void_p)]


class TrainPackList(C.Structure):       # scgg_train_pack_list (include/scgg.h)
    _fields_ = [("W", C.c_void_p * 10), ("out", C.c_void_p * 10), ("rows", C.c_int32 * 10), ("cols", C.c_int32 * 10),
                ("transpose", C.c_int32 * 10), ("count", C.c_int32)]


SIGNATURES = {
    "scgg_train_edge_pack": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    "scgg_train_edge_forward": (C.c_int, [C.POINTER(TrainEdgeArgs), _vp]),
    "scgg_train_edge_backward": (C.c_int, [C.POINTER(TrainEdgeArgs), _vp]),
    "scgg_train_edge_stats": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "scgg_train_wgrad_slabs": (_i64, [_i64, _i32]),
    "scgg_train_wgrad": (C.c_int, [_vp, _i32, _vp, _i32, _i64, _i32, _vp, _i64, _vp]),
    "scgg_train_edge_pack_multi": (C.c_int, [C.POINTER(TrainPackList), _vp]),
    "scgg_train_edge_stats_finish": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "scgg_version": (C.c_int, []),
    "scgg_last_error": (C.c_char_p, []),
    "scgg_denoiser_create": (_vp, [C.POINTER(DenoiserConfig)]),
    "scgg_denoiser_destroy": (None, [_vp]),
    "scgg_denoiser_set_param": (C.c_int, [_vp, C.c_char_p, _vp, _i64]),
    "scgg_denoiser_finalize": (C.c_int, [_vp, _vp]),
    "scgg_denoiser_refresh": (C.c_int, [_vp, _vp]),
    "scgg_denoiser_workspace_bytes": (_sz, [_vp, _i32, _i32]),
    "scgg_denoiser_forward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "scgg_denoiser_forward_static": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "scgg_denoiser_static_scratch_bytes": (_sz, []),
    "scgg_denoiser_forward_hidden": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "scgg_denoiser_side_stream": (C.c_void_p, [_vp]),
    "scgg_denoiser_profile": (C.c_int, [_vp, _i32, _i32]),
    "scgg_denoiser_profile_read": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "scgg_denoiser3d_create": (_vp, [C.POINTER(Denoiser3DConfig)]),
    "scgg_denoiser3d_destroy": (None, [_vp]),
    "scgg_denoiser3d_set_param": (C.c_int, [_vp, C.c_char_p, _vp, _i64]),
    "scgg_denoiser3d_finalize": (C.c_int, [_vp, _vp]),
    "scgg_denoiser3d_workspace_bytes": (_sz, [_vp, _i32, _i32]),
    "scgg_denoiser3d_forward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "scgg_denoiser3d_forward_hidden": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "scgg_sit_create": (_vp, [C.POINTER(SitConfig)]),
    "scgg_sit_destroy": (None, [_vp]),
    "scgg_sit_set_param": (C.c_int, [_vp, C.c_char_p, _vp, _i64]),
    "scgg_sit_finalize": (C.c_int, [_vp, _vp]),
    "scgg_sit_workspace_bytes": (_sz, [_vp, _i32, _i32]),
    "scgg_sit_forward": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _sz, _vp]),
    "scgg_philox_normal_rows": (C.c_int, [_vp, _i32, _i32, C.c_uint64, C.c_uint32, C.c_uint32, _vp]),
    "scgg_sit_sample": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, C.c_uint64, C.c_uint32, _vp, _vp, _sz, _vp]),
    "scgg_reverse_step_edges": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32,
                                          C.POINTER(Rng), _vp, _vp, _vp, _vp]),
    "scgg_sample_edges_from_probs": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, C.POINTER(Rng), _vp, _vp, _vp]),
    "scgg_labels_to_onehot": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "scgg_onehot_to_labels": (C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "scgg_adjacency_features": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp,
                                          _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp]),
    "scgg_adjacency_features_tiled": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32,
                                                _vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp, C.c_size_t, _vp]),
    "scgg_adjacency_features_workspace_bytes": (C.c_size_t, [_i32, _i32]),
    "scgg_new_edges": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _vp]),
    "scgg_apply_edge_edits": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "scgg_projector_create": (C.c_int, [_i32, _i32, _vp, C.POINTER(_vp)]),
    "scgg_projector_destroy": (None, [_vp]),
    "scgg_projector_begin": (C.c_int, [_vp, _i64, _vp, _i32, _vp, _vp, C.POINTER(_i32)]),
    "scgg_projector_finish": (C.c_int, [_vp, _vp, _i64, _vp, _i64, C.POINTER(_i64)]),
    "scgg_projector_export": (C.c_int, [_vp, _i32, _vp, _vp]),
    "scgg_projector_counter": (_i64, [_vp]),
    "scgg_edge_block_bf16_image_bytes": (C.c_size_t, [_i32, _i32]),
    "scgg_edge_block_bf16_pack": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "scgg_edge_block_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32,
                                       _i32, _vp]),
    "scgg_attn_merge": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "scgg_linear_rows_image_bytes": (C.c_size_t, [_i32, _i32]),
    "scgg_linear_rows_pack": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp]),
    "scgg_linear_rows": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _vp]),
    "scgg_dropout": (C.c_int, [_vp, _vp, _i64, C.c_double, C.c_uint64, C.c_uint32, _vp]),
    "scgg_train_y_save_width": (_i64, [_i32, _i32]),
    "scgg_train_y_keep_width": (_i64, [_i32, _i32]),
    "scgg_train_y_forward": (C.c_int, [C.POINTER(TrainYArgs), _vp]),
    "scgg_train_y_backward": (C.c_int, [C.POINTER(TrainYArgs), _vp]),
    "scgg_train_y_wgrad": (C.c_int, [C.POINTER(TrainYArgs), C.POINTER(C.c_void_p), _vp]),
    "scgg_train_node_stats": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "scgg_train_node_stats_backward": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "scgg_sym_mask": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "scgg_dropout_devkey": (C.c_int, [_vp, _vp, _i64, C.c_double, _vp, C.c_uint32, _vp]),
    "scgg_masked_cross_entropy_blocks": (_i64, [_i64]),
    "scgg_masked_cross_entropy": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "scgg_train_sum_rows_splits": (_i64, [_i64, _i64, _i32]),
    "scgg_train_sum_rows": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _vp, _i64, _vp]),
    "scgg_laplacian": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp]),
    "scgg_eigen_features": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp]),
    "scgg_eigen_features_tridiag": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp]),
}

_lib = None


class ScggError(RuntimeError):
    pass


def lib():
    """Load libscgg.so once.  No fallback: a missing library is an error."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ScggError(f"{LIB_PATH} is missing: build it first (python -c 'import __graft_entry__ as g; g.build()' "
                            f"or make -C {_HERE}/csrc).  There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc: int, what: str = "scgg"):
    if rc != 0:
        raise ScggError(f"{what}: {lib().scgg_last_error().decode()}")


def ptr(t):
    """device pointer of a contiguous CUDA(HIP) tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise ScggError("scgg kernels need tensors on the GPU (there is no CPU path)")
    if not t.is_contiguous():
        raise ScggError("scgg kernels need contiguous tensors")
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)

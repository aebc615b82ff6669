"""ctypes binding of libhmvae_hip.so (C ABI in include/hmvae.h).

There is NO fallback: if the library is missing or a call fails, an exception is raised.  The library
is built in-tree by `hm_vae_amd.build` (hipcc, gfx950) and travels to the GPU box with the repo.
"""
import ctypes as C
import os

import torch

from . import build as _build

_c_float_p = C.c_void_p
_i32 = C.c_int32
_i64 = C.c_int64
_f32 = C.c_float
_ptr = C.c_void_p


class HmvaeError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [("src", _ptr), ("dst", _ptr), ("wp", _ptr), ("bias", _ptr), ("bias_idx", _ptr), ("g_nc4", _ptr), ("g_woff", _ptr),
                ("g_choff", _ptr), ("chan_rows", _ptr), ("dst_rows", _ptr), ("p_off", _ptr), ("p_len", _ptr),
                ("pairs", _ptr), ("n_groups", _i32), ("mt", _i32), ("kt", _i32), ("n_pos", _i32), ("bp", _i32),
                ("src_ts", _i32), ("dst_ts", _i32), ("flags", _i32), ("slope", _f32)]


class Gemm2Desc(C.Structure):
    _fields_ = [("src", _ptr), ("dst", _ptr), ("wp", _ptr), ("bias", _ptr), ("bias_idx", _ptr), ("g_nc4", _ptr),
                ("g_woff", _ptr), ("g_choff", _ptr), ("chan_rows", _ptr), ("dst_rows", _ptr), ("pb_off", _ptr),
                ("pb_len", _ptr), ("items", _ptr), ("n_groups", _i32), ("mt", _i32), ("kt", _i32), ("n_pos", _i32),
                ("bp", _i32), ("np", _i32), ("n_pblk", _i32), ("n_units", _i32), ("ksplit", _i32), ("xcd_mode", _i32), ("wpos", _i32), ("max_items", _i32), ("max_nc4", _i32), ("src_ts", _i32),
                ("dst_ts", _i32), ("flags", _i32), ("slope", _f32), ("debug", _ptr), ("bias_idx2", _ptr), ("act", _ptr),
                ("act_rows", _ptr)]


class Gemm4Desc(C.Structure):
    _fields_ = [("src", _ptr), ("dst", _ptr), ("wp4", _ptr), ("bias", _ptr), ("bias_idx", _ptr), ("bias_idx2", _ptr),
                ("g_nc32", _ptr), ("g_woff4", _ptr), ("g_choff", _ptr), ("g_ntiles", _ptr), ("chan_rows", _ptr),
                ("dst_rows", _ptr), ("pb_off", _ptr), ("pb_len", _ptr), ("items", _ptr), ("scratch", _ptr),
                ("counters", _ptr), ("n_groups", _i32), ("mtw", _i32), ("nc", _i32), ("kt", _i32), ("n_pos", _i32),
                ("bp", _i32), ("n_split", _i32), ("n_units", _i32), ("max_items", _i32), ("max_nc32", _i32),
                ("src_ts", _i32), ("dst_ts", _i32), ("flags", _i32), ("slope", _f32), ("debug", _ptr), ("act", _ptr),
                ("act_rows", _ptr), ("blk_pos", _ptr), ("pp", _i32), ("n_img", _i32)]


class Conv3Desc(C.Structure):
    _fields_ = [("src", _ptr), ("dst", _ptr), ("wp", _ptr), ("bias", _ptr), ("bias_idx", _ptr),
                ("chan_rows", _ptr), ("dst_rows", _ptr),
                ("wave_tab", _ptr), ("win_rows", _ptr), ("items", _ptr), ("segs", _ptr), ("worker_seg", _ptr),
                ("fixups", _ptr),
                ("scratch", _ptr), ("n_groups", _i32), ("kt", _i32), ("bp", _i32), ("n_chunks", _i32), ("ts", _i32),
                ("win", _i32), ("max_nc4", _i32), ("max_items", _i32), ("un", _i32), ("n_workers", _i32),
                ("n_fixups", _i32), ("lds_floor", _i32), ("src_ts", _i32), ("dst_ts", _i32), ("flags", _i32),
                ("slope", _f32), ("bias_idx2", _ptr)]


class WgradDesc(C.Structure):
    _fields_ = [("dz", _ptr), ("x", _ptr), ("slab", _ptr), ("tile_g", _ptr), ("tile_col0", _ptr), ("g_ncols", _ptr),
                ("g_choff", _ptr), ("chan_rows", _ptr), ("dz_rows", _ptr), ("n_tiles", _i32), ("n_groups", _i32),
                ("mt", _i32), ("kt", _i32), ("n_split", _i32), ("t_dst", _i32), ("t_src", _i32), ("stride", _i32),
                ("pad", _i32), ("pad_mode", _i32), ("bp", _i32), ("dz_ts", _i32), ("x_ts", _i32), ("transposed", _i32),
                ("items", _ptr), ("worker_items", _ptr), ("n_workers", _i32), ("mode", _i32),
                ("dz_planes", _ptr), ("x_planes", _ptr), ("dz_plane_stride", _i64), ("x_plane_stride", _i64),
                ("fused_reduce", _ptr), ("pair_tab", _ptr), ("n_pairs", _i32), ("pad2_", _i32)]


class WgradPqDesc(C.Structure):
    """hmvae_wgrad_pq_desc (include/hmvae.h)."""
    _fields_ = [("dz", _ptr), ("x", _ptr), ("dense", _ptr), ("dz_base", _ptr), ("x_base", _ptr), ("tile_r0", _ptr),
                ("tile_c0", _ptr), ("ok", _ptr), ("child_tab", _ptr), ("child_scale", _ptr), ("m", _ptr),
                ("n_tiles", _i32), ("n_rows", _i32), ("n_chan", _i32), ("kt", _i32), ("log_t", _i32), ("log_s", _i32),
                ("bp", _i32), ("dz_ts", _i32), ("x_ts", _i32), ("row_stride", _i32), ("chan_stride", _i32),
                ("child_side", _i32), ("accumulate", _i32), ("pad_", _i32)]


class WreduceDesc(C.Structure):
    _fields_ = [("slab", _ptr), ("dense", _ptr), ("tile_g", _ptr), ("tile_col0", _ptr), ("g_ncols", _ptr),
                ("g_choff", _ptr), ("chan_id", _ptr), ("row_id", _ptr), ("g_nc4", _ptr), ("g_okoff", _ptr), ("pair_ok", _ptr),
                ("n_tiles", _i32), ("mt", _i32), ("kt", _i32), ("n_split", _i32), ("row_stride", _i32), ("chan_stride", _i32), ("accumulate", _i32), ("child_tab", _ptr), ("child_scale", _ptr), ("child_side", _i32), ("pad_", _i32), ("tile_part0", _ptr), ("tile_nparts", _ptr)]


class BridgeFwdDesc(C.Structure):
    _fields_ = [("x", _ptr), ("w1", _ptr), ("b1", _ptr), ("eps", _ptr), ("w2", _ptr), ("b2", _ptr), ("dist", _ptr), ("z", _ptr),
                ("y", _ptr), ("kl_partial", _ptr), ("k_edges", _i32), ("n_in", _i32), ("d", _i32), ("n_out", _i32),
                ("x_es", _i32), ("y_es", _i32), ("y_off", _i32), ("B", _i32), ("bp", _i32), ("sample", _i32)]


class BridgeBwdDesc(C.Structure):
    _fields_ = [("dy", _ptr), ("w2", _ptr), ("dist", _ptr), ("eps", _ptr), ("w1", _ptr), ("act", _ptr), ("ddist", _ptr),
                ("dx", _ptr), ("k_edges", _i32), ("n_in", _i32), ("d", _i32), ("n_out", _i32), ("x_es", _i32), ("y_es", _i32),
                ("y_off", _i32), ("B", _i32), ("bp", _i32), ("flags", _i32), ("kl_scale", _f32), ("slope", _f32)]


class PackOp(C.Structure):
    _fields_ = [("dense", _ptr), ("packed", _ptr), ("g_nc4", _ptr), ("g_woff", _ptr), ("g_choff", _ptr), ("chan_id", _ptr),
                ("row_id", _ptr), ("g_okoff", _ptr), ("pair_ok", _ptr), ("n_groups", _i32), ("max_nc4", _i32), ("mt", _i32),
                ("kt", _i32), ("row_stride", _i32), ("chan_stride", _i32), ("tap_stride", _i32), ("block_start", _i32),
                ("cb", _i32), ("fold_taps", _ptr), ("fold_coef", _ptr), ("child_tab", _ptr), ("fold_nf", _i32),
                ("fold_kt", _i32), ("child_side", _i32), ("fold_pad", _i32), ("child_scale", _ptr)]


class RowsumOp(C.Structure):
    _fields_ = [("src", _ptr), ("dst", _ptr), ("rows", _ptr), ("n_out", _i32), ("rows_per_out", _i32), ("row_step", _i32),
                ("n_cols", _i32), ("accumulate", _i32), ("block_start", _i32)]


class Clip(C.Structure):
    """hmvae_clip (include/hmvae.h): one clip of a batch assembled by hmvae_motion_batch."""
    _fields_ = [("row0", _i64), ("stride", _i32), ("reserved", _i32), ("rot", _f32 * 9), ("pad_", _f32 * 3)]


class AdamTensor(C.Structure):
    _fields_ = [("p", _ptr), ("g", _ptr), ("m", _ptr), ("v", _ptr), ("n", _i64)]


FLAG_LEAKY = 1
FLAG_ACCUM = 2
FLAG_LEAKY_GRAD = 4
PAD_REFLECT = 0
PAD_ZEROS = 1

_SIGS = {
    "hmvae_version": ([], _i32),
    "hmvae_error_string": ([_i32], C.c_char_p),
    "hmvae_gather_gemm": ([C.POINTER(GemmDesc), _ptr], _i32),
    "hmvae_gather_gemm_v2": ([C.POINTER(Gemm2Desc), _ptr], _i32),
    "hmvae_gather_gemm_v3": ([C.POINTER(Conv3Desc), _ptr], _i32),
    "hmvae_gather_gemm_v4": ([C.POINTER(Gemm4Desc), _ptr], _i32),
    "hmvae_wgrad_gemm": ([C.POINTER(WgradDesc), _ptr], _i32),
    "hmvae_wgrad_reduce": ([C.POINTER(WreduceDesc), _ptr], _i32),
    "hmvae_wgrad_pq": ([C.POINTER(WgradPqDesc), _ptr], _i32),
    "hmvae_pack_weights": ([_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _ptr], _i32),
    "hmvae_wgrad_reduce_multi": ([_ptr, _ptr, _i32, _i32, _ptr], _i32),
    "hmvae_row_sums_scaled": ([_ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _ptr], _i32),
    "hmvae_row_sums_multi": ([_ptr, _i32, _i32, _ptr], _i32),
    "hmvae_pack_weights_multi": ([_ptr, _i32, _i32, _i32, _ptr], _i32),
    "hmvae_pack_weights_v4": ([_ptr, _i32, _i32, _i32, _ptr], _i32),
    "hmvae_pool_leaky_fwd": ([_ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _f32, _ptr], _i32),
    "hmvae_pool_leaky_bwd": ([_ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _f32, _ptr], _i32),
    "hmvae_upsample_unpool_fwd": ([_ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _ptr], _i32),
    "hmvae_upsample_unpool_bwd": ([_ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _ptr], _i32),
    "hmvae_leaky_bwd": ([_ptr, _ptr, _ptr, _i64, _i64, _i64, _f32, _ptr], _i32),
    "hmvae_row_sums": ([_ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _ptr], _i32),
    "hmvae_aos_to_soa": ([_ptr, _ptr, _i32, _i32, _i32, _i32, _ptr], _i32),
    "hmvae_soa_to_aos": ([_ptr, _ptr, _i32, _i32, _i32, _i32, _ptr], _i32),
    "hmvae_reparam_kl_fwd": ([_ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _ptr], _i32),
    "hmvae_randn": ([_ptr, _i64, C.c_uint64, _ptr, _ptr], _i32),
    "hmvae_latent_bridge_fwd": ([_ptr, _ptr], _i32),
    "hmvae_latent_bridge_bwd": ([_ptr, _ptr], _i32),
    "hmvae_reparam_kl_bwd": ([_ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _i32, _i32, _f32, _ptr], _i32),
    "hmvae_fk_fwd": ([_ptr, _i32, _ptr, _i64, C.POINTER(_i32), _i32, _i64, _ptr, _ptr, _ptr], _i32),
    "hmvae_fk_bwd": ([_ptr, _i32, _ptr, _i64, C.POINTER(_i32), _i32, _i64, _ptr, _ptr, _ptr, _ptr], _i32),
    "hmvae_decode_tail_loss": ([_ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _f32, _f32, _f32, _ptr, _ptr,
                                C.POINTER(_i32), _ptr], _i32),
    "hmvae_decode_tail_loss_masked": ([_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _f32, _f32, _f32, _ptr, _ptr,
                                       C.POINTER(_i32), _ptr], _i32),
    "hmvae_l2_anchor": ([_ptr, _ptr, C.c_int64, _f32, _f32, _ptr, _ptr, C.POINTER(_i32), _ptr], _i32),
    "hmvae_decode_tail_fwd": ([_ptr, _ptr, _i32, _i32, _i32, _ptr, _ptr, _ptr, _ptr], _i32),
    "hmvae_vae_loss_finalize": ([_ptr, _i32, _ptr, _i32, _ptr, _i32] + [_f32] * 10 + [_ptr, _ptr], _i32),
    "hmvae_finalize_sums": ([_ptr, _i32, _i32, _i32, _f32, _ptr, _ptr], _i32),
    "hmvae_traj_loss": ([_ptr, _ptr, _ptr, _i32, _i32, _i32, _f32, _f32, _ptr, _ptr, _ptr, _ptr], _i32),
    "hmvae_split_bf16x3": ([_ptr, _ptr, _i64, _i64, _ptr], _i32),
    "hmvae_motion_batch": ([_ptr, _i64, _ptr, _ptr, _ptr, _i32, _i32, _i32] + [_ptr] * 7 + [_ptr], _i32),
    "hmvae_traj_loss_finalize": ([_ptr, _i32, _f32, _f32, _f32, _ptr, _ptr], _i32),
    "hmvae_keyframe_root_loss": ([_ptr, _ptr, _ptr, _ptr, _ptr, _i32, _i32, _i32, _f32, _ptr, _ptr, C.POINTER(_i32), _ptr], _i32),
    "hmvae_adam_step": ([_ptr, _i32, _i64, _ptr, _ptr, _f32, _f32, _f32, _f32, _ptr], _i32),
}

EXPORTS = sorted(_SIGS)
_lib = None


def lib_path():
    return _build.LIB_PATH


def load():
    """Load (building first if the sources are newer) and type the library.  Raises on any problem."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("HMVAE_LIB") or _build.LIB_PATH      # HMVAE_LIB: a variant build (kernel A/B experiments)
    if not os.path.exists(path):
        if os.path.exists("/opt/rocm/bin/hipcc") or os.environ.get("HIPCC"):
            _build.build(verbose=False)
        else:
            raise HmvaeError("libhmvae_hip.so not found at %s and no hipcc to build it; there is no CPU fallback" % path)
    lib = C.CDLL(path)
    for name, (args, res) in _SIGS.items():
        fn = getattr(lib, name)      # AttributeError if an export is missing
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib


def check(code, what=""):
    if code != 0:
        msg = load().hmvae_error_string(code).decode()
        raise HmvaeError("%s failed: %s (%d)" % (what or "hmvae call", msg, code))


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must be contiguous fp32/int32 on a HIP device."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HmvaeError("hm_vae_amd kernels need device tensors (got a CPU tensor); there is no CPU fallback")
    if not t.is_contiguous():
        raise HmvaeError("non-contiguous tensor passed to a HIP kernel")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_gpu():
    if not torch.cuda.is_available():
        raise HmvaeError("no HIP device visible: hm_vae_amd has no CPU path (use oracle/ only for checking)")

"""ctypes binding of libuav_hip.so (C ABI declared in include/uav_hip.h).

The product path has no CPU / PyTorch fallback: if the HIP library cannot be loaded, every op
raises.  `load()` builds the library in-tree on first use when hipcc is available.
"""
import ctypes as C
import os

from . import build as _build

_LIB = None

c_p = C.c_void_p
i32 = C.c_int32
i64 = C.c_int64
f32 = C.c_float
u32 = C.c_uint32


class ConvParams(C.Structure):
    """Mirror of `uav_conv_params` (include/uav_hip.h)."""
    _fields_ = [
        ("a1", c_p), ("a2", c_p), ("c1", i32), ("c2", i32),
        ("w", c_p), ("bias", c_p), ("rowbias", c_p),
        ("rows_per_batch", i32), ("rowbias_stride", i32),
        ("residual", c_p), ("res_stride", i32),
        ("out", c_p), ("out_stride", i32),
        ("n_img", i32), ("t_len", i32), ("hi", i32), ("wi", i32), ("ho", i32), ("wo", i32),
        ("kt", i32), ("kh", i32), ("kw", i32), ("stride", i32),
        ("pad_t", i32), ("pad_h", i32), ("pad_w", i32), ("upsample", i32),
        ("n", i32), ("n_pad", i32), ("k_pad", i32),
        ("out_scale", f32), ("flags", u32), ("zero_page", c_p),
        ("gn_partials", c_p), ("gn_groups", i32),
        ("out_map_w", i32), ("out_map_sy", i32), ("out_map_sx", i32), ("out_map_off", i32),
        ("a2_images", i32), ("a2_center_tap", i32),
        ("ln_raw_out", c_p), ("ln_stat_out", c_p), ("ln_stat_in", c_p), ("ln_colsum", c_p),
        ("ln_chunks", i32), ("ln_n", i32), ("ln_eps", f32),
        ("gn_chunk_cpi", i32), ("gn_chunk_stride", i32), ("gn_chunk_off", i32), ("gn_chunks_total", i64),
    ]


class XattnParams(C.Structure):
    """Mirror of `uav_xattn_params` (include/uav_hip.h): one fused cross-attention sub-layer."""
    _fields_ = [("ln_gamma", c_p), ("ln_beta", c_p), ("ln_eps", f32), ("wq_packed", c_p), ("kv_packed", c_p), ("wo_packed", c_p), ("out_bias", c_p)]


class TattnParams(C.Structure):
    """Mirror of `uav_tattn_params` (include/uav_hip.h): the fused temporal attention sub-layer."""
    _fields_ = [("ln_gamma", c_p), ("ln_beta", c_p), ("ln_eps", f32), ("wq_packed", c_p), ("wk_packed", c_p), ("wv_packed", c_p),
                ("wo_packed", c_p), ("out_bias", c_p), ("rel_bias", c_p), ("rope_cos", c_p), ("rope_sin", c_p), ("rot_dim", i32),
                ("next_ln_out", c_p), ("next_ln_gamma", c_p), ("next_ln_beta", c_p), ("next_ln_eps", f32)]


class ProjInParams(C.Structure):
    """Mirror of `uav_projin_params` (include/uav_hip.h): GroupNorm apply + proj_in in front of the whole-block launch."""
    _fields_ = [("gn_scale", c_p), ("gn_shift", c_p), ("w_packed", c_p), ("bias", c_p)]


class FfParams(C.Structure):
    """Mirror of `uav_ff_params` (include/uav_hip.h): the fused feed-forward sub-layer."""
    _fields_ = [("ln_gamma", c_p), ("ln_beta", c_p), ("ln_eps", f32), ("w_packed", c_p), ("up_bias", c_p), ("down_bias", c_p)]


CONV_GEGLU = 1
CONV_OUT_F32 = 2
CONV_PERSISTENT = 64
CONV_RES_F32 = 128
CONV_GELU, CONV_QUICK_GELU = 256, 512
CONV_NO_SHORTK = 1024
CONV_NO_W4 = 2048
CONV_OUT_HILO = 4096
CONV_RELU, CONV_SIGMOID, CONV_TANH = 4, 8, 16

# name -> (restype, argtypes); the complete export list of include/uav_hip.h
SIGNATURES = {
    "uav_version": (C.c_int, []),
    "uav_has_dev_kernels": (C.c_int, []),
    "uav_device_check": (C.c_int, [C.c_int, C.c_char_p]),
    "uav_conv_gemm_f16": (C.c_int, [C.POINTER(ConvParams), c_p]),
    "uav_conv_gemm_gn_chunk_rows": (C.c_int, [C.POINTER(ConvParams)]),
    "uav_conv_gemm_hilo_ok": (C.c_int, [C.POINTER(ConvParams)]),
    "uav_groupnorm_finalize_partials": (C.c_int, [c_p, i64, i32, i32, i32, i64, i32, f32, c_p, c_p, c_p, c_p, c_p]),
    "uav_groupnorm_workspace_bytes": (i64, [i32, i32]),
    "uav_groupnorm_scale_shift": (C.c_int, [c_p, c_p, i32, i32, i32, i64, i32, i32, i64, i32, f32, c_p, c_p, c_p, c_p, c_p, i64, c_p]),
    "uav_conv_gemm_ln_ok": (C.c_int, [C.POINTER(ConvParams)]),
    "uav_frames_to_clip_f32": (C.c_int, [c_p, i32, c_p, i32, i32, i64, c_p]),
    "uav_clip_to_frames_u8": (C.c_int, [c_p, c_p, i32, i32, i64, c_p]),
    "uav_groupnorm_finalize_partials2": (C.c_int, [c_p, i64, i32, i32, i32, c_p, i64, i32, i32, i32, i32, i32, i64, i32, f32, c_p, c_p, c_p, c_p, c_p]),
    "uav_groupnorm_apply": (C.c_int, [c_p, c_p, i32, i32, i32, i64, i32, i64, c_p, c_p, i32, c_p, c_p, i32, c_p]),
    "uav_layernorm_f16": (C.c_int, [c_p, c_p, c_p, c_p, i64, i32, f32, c_p]),
    "uav_layernorm_f32in": (C.c_int, [c_p, c_p, c_p, c_p, i64, i32, f32, c_p]),
    "uav_attention_f16": (C.c_int, [c_p, i64, c_p, i64, c_p, i64, c_p, i64, i32, i32, i32, i32, i32, i32, f32, i32, c_p, c_p]),
    "uav_xattn_sublayers_f32": (C.c_int, [c_p, c_p, c_p, i32, i64, i32, i32, i32, i32, f32, c_p]),
    "uav_block_attn_sublayers_f32": (C.c_int, [c_p, c_p, c_p, i32, f32, c_p, i32, i32, i64, i32, i32, f32, c_p]),
    "uav_ff_sublayer_f32": (C.c_int, [c_p, c_p, c_p, c_p, i64, i32, i32, c_p]),
    "uav_attention512_pack_bytes": (C.c_int64, [i32, i32]),
    "uav_attention512_pack_kv": (C.c_int, [c_p, i64, c_p, i64, i32, i32, c_p, c_p, c_p]),
    "uav_attention512_packed_f16": (C.c_int, [c_p, i64, c_p, c_p, c_p, i64, i32, i32, i32, f32, c_p]),
    "uav_block_sublayers_f32": (C.c_int, [c_p, c_p, c_p, c_p, c_p, i32, f32, c_p, c_p, i32, i32, i64, i32, i32, i32, f32, c_p]),
    "uav_tattn_sublayer_f32": (C.c_int, [c_p, c_p, c_p, i32, i32, i64, i32, i32, f32, c_p]),
    "uav_xattn_pack_kv": (C.c_int, [c_p, i64, c_p, i64, i32, i32, i32, i32, c_p, c_p]),
    "uav_temporal_attention_f16": (C.c_int, [c_p, c_p, i32, i32, i64, i32, i32, f32, c_p, c_p, i32, c_p, c_p]),
    "uav_linear_small": (C.c_int, [c_p, c_p, c_p, c_p, i32, i32, i32, i32, i32, c_p]),
    "uav_timestep_embedding": (C.c_int, [c_p, i32, i32, i32, f32, c_p, c_p]),
    "uav_pack_nhwc": (C.c_int, [c_p, i32, c_p, i32, i32, c_p, i32, i32, i32, i64, f32, c_p]),
    "uav_unpack_ncthw": (C.c_int, [c_p, i32, i32, c_p, i32, i32, i32, i32, i64, f32, f32, c_p]),
    "uav_cfg_ddim_v0": (C.c_int, [c_p, c_p, c_p, c_p, c_p, i64, f32, f32, f32, i32, f32, c_p]),
    "uav_cfg_ddim_v0_f32": (C.c_int, [c_p, c_p, c_p, c_p, c_p, i64, f32, f32, f32, i32, f32, c_p]),
    "uav_ddim_vt": (C.c_int, [c_p, c_p, c_p, c_p, i64, f32, f32, f32, f32, f32, i32, f32, c_p]),
    "uav_ddim_vt_f32": (C.c_int, [c_p, c_p, c_p, c_p, i64, f32, f32, f32, f32, f32, i32, f32, c_p]),
    "uav_axpby_f16": (C.c_int, [c_p, c_p, c_p, i64, f32, f32, c_p]),
    "uav_cast_f32_f16": (C.c_int, [c_p, c_p, i64, c_p]),
    "uav_cast_f32_hilo": (C.c_int, [c_p, c_p, i64, i32, c_p]),
    "uav_sft_fuse": (C.c_int, [c_p, c_p, c_p, c_p, i64, f32, i32, i32, c_p]),
    "uav_plane_stats_workspace_bytes": (i64, [i32]),
    "uav_plane_stats_f32": (C.c_int, [c_p, i32, i64, c_p, c_p, c_p, i64, c_p]),
    "uav_adain_apply_f32": (C.c_int, [c_p, c_p, i32, i64, c_p, c_p, c_p, c_p, f32, c_p]),
    "uav_atrous_blur_f32": (C.c_int, [c_p, c_p, c_p, i32, i32, i32, i32, c_p]),
    "uav_resize_bicubic_f32": (C.c_int, [c_p, c_p, i32, i32, i32, i32, i32, f32, f32, c_p]),
    "uav_resize_area_f32": (C.c_int, [c_p, c_p, i32, i32, i32, i32, i32, f32, c_p]),
    "uav_conv_gemm_f32": (C.c_int, [C.POINTER(ConvParams), c_p]),
    "uav_instnorm_f32": (C.c_int, [c_p, c_p, i32, i32, i32, f32, i32, c_p]),
    "uav_add_relu_f32": (C.c_int, [c_p, c_p, c_p, i64, i32, c_p]),
    "uav_axpby_f32": (C.c_int, [c_p, c_p, c_p, i64, f32, f32, c_p]),
    "uav_copy_cols_f32": (C.c_int, [c_p, i32, i32, c_p, i32, i32, i32, i64, i32, c_p]),
    "uav_gru_gates_f32": (C.c_int, [c_p, c_p, c_p, c_p, i64, i32, i32, c_p]),
    "uav_avgpool2_f32": (C.c_int, [c_p, i64, i32, i32, c_p, i64, c_p]),
    "uav_resize_bilinear_f32": (C.c_int, [c_p, c_p, i64, i32, i32, i32, i32, C.c_float, C.c_float, c_p]),
    "uav_corr_lookup_f32": (C.c_int, [C.POINTER(c_p), C.POINTER(i64), C.POINTER(i32), C.POINTER(i32), c_p, i32, c_p, i32, i64, i32, c_p]),
    "uav_convex_upsample_f32": (C.c_int, [c_p, i32, c_p, c_p, i32, i32, i32, c_p]),
    "uav_propagate_step_f16": (C.c_int, [c_p, c_p, c_p, c_p, c_p, i32, i32, i32, i64, i64, i32, i32, f32, f32, f32, c_p]),
    "uav_propagate_step_f32": (C.c_int, [c_p, c_p, c_p, c_p, c_p, i32, i32, i32, i64, i64, i32, f32, f32, f32, c_p]),
}

EXPECTED_ABI = 6          # include/uav_hip.h UAV_ABI_VERSION this binding was written against


class UavError(RuntimeError):
    pass


def lib_path():
    return _build.LIB


def load(build_if_missing=True):
    """Load (building if necessary) libuav_hip.so and attach prototypes.  Raises on failure."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _build.LIB
    override = os.environ.get("UAV_HIP_LIB")        # development A/B only: load another build of the same ABI
    if override:
        if not os.path.exists(override):
            raise UavError(f"UAV_HIP_LIB={override} does not exist")
        path, build_if_missing = override, False
    if build_if_missing and not _build.is_fresh():
        if os.path.exists(_build.HIPCC):
            _build.build()
        elif not os.path.exists(path):
            raise UavError("libuav_hip.so is missing and hipcc is unavailable — the HIP extension is required "
                           "(there is no CPU fallback)")
    if not os.path.exists(path):
        raise UavError(f"{path} not found — run `python __graft_entry__.py` / uav.build.build() first")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if an exported symbol is missing
        fn.restype = res
        fn.argtypes = args
    lib.uav_version.restype = C.c_int
    if lib.uav_version() != EXPECTED_ABI:
        # a stale build or a UAV_HIP_LIB override of another ABI would read past the parameter structs it knows
        raise UavError(f"{path}: ABI version {lib.uav_version()} != {EXPECTED_ABI} expected by this binding — rebuild "
                       "(python __graft_entry__.py)")
    _LIB = lib
    return lib


def check(rc, what):
    if rc != 0:
        kind = {-1: "UAV_EINVAL", -2: "UAV_EALIGN", -3: "UAV_ESHAPE"}.get(rc, f"hipError {rc}")
        raise UavError(f"{what} failed: {kind}")

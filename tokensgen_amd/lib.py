"""ctypes binding of libtokensgen_hip.so (include/tokensgen_hip.h).  Fails loudly if the library is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TG_LIB_PATH") or os.path.join(_HERE, "libtokensgen_hip.so")   # override: A/B builds in tools/
TG_MAX_GROUPS = 16

EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_SILU, EPI_BIAS_GATE_RES = 0, 1, 2, 3
EPI_BIAS_KEEP_GELU, EPI_BIAS_MUL_GELU_GRAD = 4, 5      # training step; 4-wave GEMM shapes only (kernels.gemm_act_supported)


class GroupTable(C.Structure):
    """struct tg_group_table"""
    _fields_ = [("mod", C.c_void_p), ("mod_ld", C.c_long), ("mod_batch_stride", C.c_long),
                ("tok_group", C.c_void_p), ("row", C.c_int32 * TG_MAX_GROUPS),
                ("shift_col", C.c_int32 * TG_MAX_GROUPS), ("scale_col", C.c_int32 * TG_MAX_GROUPS),
                ("gate_col", C.c_int32 * TG_MAX_GROUPS)]


_vp, _l, _i, _f = C.c_void_p, C.c_long, C.c_int, C.c_float
# name -> argtypes, exactly the prototypes of include/tokensgen_hip.h
PROTOTYPES = {
    "tg_gemm_bf16": [_vp, _l, _l, _vp, _l, _vp, _vp, _l, _l, _i, _i, _i, _i, _i, _vp, _l, _l, C.POINTER(GroupTable), _vp],
    "tg_gemm_bf16_pair": [_vp, _l, _vp, _vp, _vp, _l, _i, _vp, _l, _vp, _vp, _vp, _l, _i, _l, _l, _l, _i, _i, _i, _i, _vp],
    "tg_gemm_bf16_qkv": [_vp, _l, _vp, _vp, _vp, _l, _i, _vp, _l, _vp, _l, _vp, _vp, _vp, _l, _i, _vp, _l, _l, _l, _l, _i, _i, _i, _i, _vp],
    "tg_adaln_modulate": [_vp, _l, _l, _vp, _l, _l, _vp, _vp, _f, _i, _i, _i, _i, C.POINTER(GroupTable), _vp],
    "tg_qk_layernorm_rope": [_vp, _l, _l, _i, _i, _i, _vp, _vp, _f, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _f, _vp],
    "tg_qk_layernorm_rope_pair": [_vp, _vp, _l, _l, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _f, _f, _vp],
    "tg_qk_layernorm_rope_pair_kmax": [_vp, _vp, _l, _l, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _f, _f, _vp, _vp, _vp],
    "tg_qk_layernorm_rope_pair_out": [_vp, _vp, _l, _l, _vp, _vp, _l, _l, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _f, _f, _vp, _vp, _vp],
    "tg_transpose_v": [_vp, _l, _l, _i, _i, _i, _i, _vp, _l, _vp],
    "tg_attention_fwd": [_vp, _l, _l, _vp, _l, _l, _vp, _l, _i, _vp, _l, _l, _vp, _l, _l, _vp, _l, _i, _f, _vp, _l, _l,
                         _i, _i, _i, _f, _i, _vp],
    "tg_attention_fwd_multi": [_vp, _i, _i, _i, _f, _i, _vp, _vp],
    "tg_attention_bwd": [_vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l,
                         _i, _i, _i, _i, _f, _i, _vp, _vp, _vp],
    "tg_attention_bwd_ex": [_vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _l, _l,
                            _i, _i, _i, _i, _f, _i, _vp, _vp, _i, _vp, _vp],
    "tg_attention_bwd_multi": [_vp, _i, _i, _i, _i, _vp, _vp],
    "tg_attention_bwd_probe": [_vp, _l, _vp],
    "tg_attention_bwd_probe_verdict": [_vp, _l],
    "tg_attention_fwd_lse": [_vp, _l, _l, _vp, _l, _l, _vp, _l, _i, _vp, _l, _l, _i, _i, _i, _f, _vp, _vp],
    "tg_attention_fwd_lse_ex": [_vp, _l, _l, _vp, _l, _l, _vp, _l, _i, _vp, _l, _l, _i, _i, _i, _f, _i, _vp, _vp, _l, _vp, _vp],
    "tg_qk_layernorm_rope_bwd": [_vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _i, _i, _i, _vp, _f, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _f, _vp, _vp],
    "tg_transpose_2d": [_vp, _l, _i, _i, _vp, _l, _i, _vp],
    "tg_colsum": [_vp, _l, _i, _i, _vp, _vp],
    "tg_adaln_modulate_bwd": [_vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _vp, _vp, _f, _i, _i, _i, _i, C.POINTER(GroupTable), _vp, _vp, _vp, _vp, _l, _l, _vp],
    "tg_gate_residual_bwd": [_vp, _l, _l, _vp, _l, _l, _vp, _l, _l, _i, _i, _i, C.POINTER(GroupTable), _vp, _i, _vp],
    "tg_act": [_vp, _vp, _vp, _l, _i, _vp],
    "tg_colsum_f32": [_vp, _l, _i, _i, _vp, _vp],
    "tg_colsum_multi": [_vp, _i, _i, _vp, _vp],
    "tg_grad_accumulate_multi": [_vp, _i, _f, _vp],
    "tg_grad_accumulate": [_vp, _i, _vp, _l, _f, _i, _vp],
    "tg_grad_clip_coef": [_vp, _l, _f, _vp, _vp, _vp],
    "tg_adamw_step": [_vp, _vp, _vp, _vp, _l, _i, _f, _f, _f, _f, _f, _vp, _i, _vp],
    "tg_vpred_loss_grad": [_vp, _vp, _vp, _vp, _i, _l, _f, _vp, _vp, _vp],
    "tg_timestep_sinusoid": [_vp, _i, _i, _vp, _vp],
    "tg_rope_table_3d": [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp],
    "tg_patchify": [_vp, _vp, _l, _i, _i, _i, _i, _i, _vp],
    "tg_unpatchify": [_vp, _l, _vp, _i, _i, _i, _i, _i, _vp],
    "tg_cfg_dpm_step": [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _l, _vp],
    "tg_cfg_dpm_step_f32": [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _l, _vp],
    "tg_cfg_dpm_step_ex": [_vp, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _i, _i, _i, _vp, _vp, _i, _l, _vp],
    "tg_pca_inverse": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "tg_pca_lowrank_filter": [_vp, _l, _vp, _vp, _vp, _l, _i, _i, _i, _vp],
    "tg_conv3d_cl": [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _l, _i, _i, _i, _vp, _vp, _vp, _vp],
    "tg_conv3d_up2_subpixel": [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _l, _i, _vp, _vp, _vp],
    "tg_groupnorm_finalize": [_vp, _l, _i, _f, _vp, _vp],
    "tg_groupnorm_stats": [_vp, _l, _i, _f, _vp, _vp, _vp],
    "tg_groupnorm_silu": [_vp, _l, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "tg_spatialnorm_silu": [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _vp, _i, _vp],
    "tg_groupnorm_reduce": [_vp, _l, _vp, _vp],
    "tg_groupnorm_silu_ex": [_vp, _l, _i, _vp, _vp, _l, _i, _f, _vp, _vp, _vp, _i, _vp],
    "tg_spatialnorm_silu_ex": [_vp, _i, _i, _i, _i, _vp, _vp, _l, _i, _f, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _vp, _i, _vp],
    "tg_avgpool_time": [_vp, _i, _l, _i, _vp, _vp],
    "tg_ncdhw_to_cl": [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _i, _vp],
    "tg_cl_to_ncdhw": [_vp, _l, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "tg_tile_blend": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
}

# name -> argtypes of the `long tg_*_floats(...)` workspace-size queries
QUERIES = {
    "tg_groupnorm_partial_floats": [C.c_long, C.c_int],
    "tg_conv3d_gn_partial_floats": [C.c_int, C.c_int, C.c_int],
    "tg_conv3d_up2_subpixel_gn_floats": [C.c_int, C.c_int, C.c_int],
    "tg_conv3d_up2_subpixel_ok": [C.c_int] * 5,
    "tg_groupnorm_reduce_rows": [C.c_long],
    "tg_conv3d_splitk_floats": [C.c_int] * 9,
    "tg_attention_bwd_ws_floats": [C.c_int, C.c_int, C.c_int, C.c_int],
    "tg_attention_bwd_probe_bytes": [],
    "tg_attention_retry_ints": [C.c_int, C.c_int, C.c_int, C.c_int],
    "tg_attention_split_floats": [C.c_int, C.c_int, C.c_int, C.c_int],
    "tg_qk_kmax_ws_floats": [C.c_int, C.c_int, C.c_int],
    "tg_qk_layernorm_rope_bwd_partial_floats": [C.c_int, C.c_int, C.c_int],
    "tg_colsum_partial_floats": [C.c_int, C.c_int],
    "tg_vpred_loss_partial_floats": [C.c_int, C.c_long],
    "tg_grad_norm_ws_floats": [],
}

TG_BWD_ONE_KERNEL = 1
# declared in the header besides PROTOTYPES and QUERIES (bound by hand in load())
OTHER_EXPORTS = ("tg_version", "tg_last_error_string", "tg_debug_set", "tg_debug_get", "tg_debug_knob_name")

_lib = None


def load():
    """Load the shared library once; raise (never fall back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(make -C tokensgen_amd/csrc). tokensgen_amd has no CPU/PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.argtypes = argtypes
        fn.restype = C.c_int
    for name, argtypes in QUERIES.items():   # workspace-size queries: pure host functions returning a count of floats
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_long
    lib.tg_version.restype = C.c_char_p
    lib.tg_last_error_string.restype = C.c_char_p
    lib.tg_debug_set.argtypes, lib.tg_debug_set.restype = [C.c_char_p, C.c_long], C.c_int
    lib.tg_debug_get.argtypes, lib.tg_debug_get.restype = [C.c_char_p, C.POINTER(C.c_long)], C.c_int
    lib.tg_debug_knob_name.argtypes, lib.tg_debug_knob_name.restype = [C.c_int], C.c_char_p
    _lib = lib
    # The library itself reads no environment variable.  The cross-check tests select between two product kernels of one operator per child process:
    # same-named environment variables are forwarded to tg_debug_set here, once, at load time (header: "Dispatch overrides for the cross-check tests").
    i = 0
    while True:
        name = lib.tg_debug_knob_name(i)
        if name is None:
            break
        if os.environ.get(name.decode()) not in (None, ""):
            check(lib.tg_debug_set(name, int(os.environ[name.decode()])), f"tg_debug_set({name.decode()})")
        i += 1
    return lib


def debug_set(knob, value):
    """tg_debug_set: dispatch override for a cross-check test (see the header); returns the previous value."""
    lib = load()
    old = C.c_long(0)
    check(lib.tg_debug_get(knob.encode(), C.byref(old)), f"tg_debug_get({knob})")
    check(lib.tg_debug_set(knob.encode(), int(value)), f"tg_debug_set({knob})")
    return old.value


def debug_get(knob):
    """tg_debug_get: the value a dispatch knob has NOW (default, environment at load time, or a later debug_set) — what the library itself will dispatch on."""
    lib = load()
    v = C.c_long(0)
    check(lib.tg_debug_get(knob.encode(), C.byref(v)), f"tg_debug_get({knob})")
    return v.value


def check(code, what):
    if code != 0:
        raise RuntimeError(f"{what} failed ({code}): {load().tg_last_error_string().decode()}")


class AttnBwdProblem(C.Structure):
    """tg_attn_bwd_problem (include/tokensgen_hip.h)"""
    _fields_ = ([(n, t) for base in ("q", "k", "v", "o") for n, t in ((base, C.c_void_p), (base + "_ld", C.c_long), (base + "_sb", C.c_long))] +
                [("dout", C.c_void_p), ("do_ld", C.c_long), ("do_sb", C.c_long)] +
                [(n, t) for base in ("dq", "dk", "dv") for n, t in ((base, C.c_void_p), (base + "_ld", C.c_long), (base + "_sb", C.c_long))] +
                [("nq", C.c_int), ("nk", C.c_int), ("scale", C.c_float), ("accumulate", C.c_int), ("lse", C.c_void_p), ("ws", C.c_void_p),
                 ("dv_bf16", C.c_void_p), ("dv_bf16_ld", C.c_long), ("dv_bf16_sb", C.c_long)])


class AccumItem(C.Structure):
    """tg_accum_item (include/tokensgen_hip.h)"""
    _fields_ = [("grad", C.c_void_p), ("acc", C.c_void_p), ("n", C.c_long), ("grad_is_bf16", C.c_int)]


class ColsumItem(C.Structure):
    """tg_colsum_item (include/tokensgen_hip.h)"""
    _fields_ = [("src", C.c_void_p), ("ld", C.c_long), ("rows", C.c_int), ("cols", C.c_int), ("src_is_f32", C.c_int)]


TG_ACCUM_MAX, TG_COLSUM_MAX = 48, 16


class AttnSegment(C.Structure):
    """tg_attn_segment (include/tokensgen_hip.h)"""
    _fields_ = [("q", C.c_void_p), ("q_ld", C.c_long), ("q_strideB", C.c_long),
                ("k", C.c_void_p), ("k_ld", C.c_long), ("k_strideB", C.c_long),
                ("vt", C.c_void_p), ("vt_ld", C.c_long), ("nk", C.c_int), ("k_norm2_max", C.c_void_p)]


class AttnProblem(C.Structure):
    """tg_attn_problem (include/tokensgen_hip.h)"""
    _fields_ = [("seg", AttnSegment * 2), ("nseg", C.c_int), ("seg2_scale", C.c_float),
                ("out", C.c_void_p), ("out_ld", C.c_long), ("out_strideB", C.c_long), ("nq", C.c_int),
                ("seg2_scale_batch", C.POINTER(C.c_float))]


class AttnWorkspace(C.Structure):
    """tg_attn_workspace (include/tokensgen_hip.h)"""
    _fields_ = [("retry", C.c_void_p), ("retry_ints", C.c_long), ("split", C.c_void_p), ("split_floats", C.c_long)]

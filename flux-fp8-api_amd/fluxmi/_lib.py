"""ctypes binding of libfluxmi.so (the C ABI declared in include/fluxmi.h).

The product path has no fallback: if the shared library is missing this module raises at import, and
every op raises `RuntimeError(fluxmi_last_error())` on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLUXMI_LIB", os.path.join(_HERE, "libfluxmi.so"))

E4M3, E5M2 = 0, 1
EPI_BF16, EPI_GELU_QUANT, EPI_GATE_RESID, EPI_SPLIT, EPI_QUANT, EPI_SILU_QUANT = range(6)
TILE_AUTO, TILE_GENERIC = -1, 100

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"fluxmi: {LIB_PATH} not found -- build it with `make -C flux-fp8-api_amd/csrc` "
        "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU/PyTorch fallback."
    )
lib = C.CDLL(LIB_PATH)

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_longlong, C.c_float


class GemmGroup(C.Structure):
    _fields_ = [
        ("A", vp), ("W", vp), ("bias", vp), ("sa_recip", vp), ("sb_recip", vp), ("C", vp), ("C2", vp),
        ("gate", vp), ("resid", vp), ("q_scale", vp),
        ("lda", i64), ("ldc", i64), ("ldc2", i64), ("ldr", i64),
        ("M", i32), ("m_tile_start", i32), ("split_n", i32), ("c2_col0", i32),
        ("vt_out", vp), ("k_out", vp), ("pe", vp), ("k_norm", vp), ("vt_ld", i64),
        ("k_rows", i32), ("tok0", i32), ("vt_rows", i32), ("kv_col0", i32), ("heads", i32), ("k_f16", i32),
        ("q_lut", vp), ("W_pairs", vp), ("a_pairs", i32), ("c8_pairs", i32),
    ]


class Tuning(C.Structure):
    """fluxmi_tuning_t (include/fluxmi.h): the kernel-selection knobs, resolved once from the FLUXMI_* environment by the library."""
    _fields_ = [
        ("struct_size", i32), ("gemm_cfg", i32), ("gemm_splitk", i32), ("gemm_hybrid", i32), ("gemm_esel", i32), ("gemm_persist", i32),
        ("attn_var", i32), ("attn_abl", i32), ("attn_defer_log2", f32), ("attn_f16k", i32), ("fuse_kv", i32), ("qlut", i32),
        ("ln_variant", i32), ("roctx", i32), ("prefetch", i32), ("w_pairs", i32), ("log", i32), ("gemm_tile192", i32), ("attn_split", i32), ("a_pairs", i32),
    ]


class Linear(C.Structure):
    _fields_ = [
        ("weight", vp), ("bias", vp), ("w_scale_recip", vp), ("in_scale", vp), ("in_scale_recip", vp),
        ("amax_trials", vp), ("kind", i32), ("N", i32), ("K", i32), ("in_fmt", i32),
    ]


class ModelDesc(C.Structure):
    _fields_ = [
        ("hidden", i32), ("heads", i32), ("mlp_hidden", i32), ("depth", i32), ("depth_single", i32),
        ("in_channels", i32), ("vec_in", i32), ("ctx_in", i32), ("guidance_embed", i32),
        ("axes_dim", i32 * 3), ("theta", i32), ("num_trials", i32),
    ]


_SIGS = {
    "fluxmi_abi_version": ([], i32),
    "fluxmi_get_tuning": ([C.POINTER(Tuning)], i32),
    "fluxmi_set_tuning": ([C.POINTER(Tuning)], i32),
    "fluxmi_gemm_debug_buffer": ([vp], i32),
    "fluxmi_clock_sample": ([vp, vp], i32),
    "fluxmi_gemm_grouped": ([C.POINTER(GemmGroup), i32, i32, i32, i32, i32, i32, i32, vp], i32),
    "fluxmi_f8_gemm": ([vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, i32, vp], i32),
    "fluxmi_gemv": ([vp, i64, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, i32, vp], i32),
    "fluxmi_quantize_act": ([vp, vp, vp, i32, i32, i64, i64, i32, vp], i32),
    "fluxmi_amax": ([vp, vp, i32, i32, i64, vp], i32),
    "fluxmi_calib_update": ([vp, vp, vp, vp, i32, i32, f32, vp], i32),
    "fluxmi_quantize_weight": ([vp, vp, vp, vp, vp, i32, i32, i32, vp], i32),
    "fluxmi_lora_fuse_f8": ([vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp], i32),
    "fluxmi_dequant": ([vp, vp, vp, i64, i32, vp], i32),
    "fluxmi_ln_modulate": ([vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    "fluxmi_act": ([vp, vp, i32, i32, i64, i64, i32, vp], i32),
    "fluxmi_gate_residual": ([vp, vp, vp, vp, i32, i32, i32, i64, i64, i64, i64, vp], i32),
    "fluxmi_add": ([vp, vp, vp, i64, vp], i32),
    "fluxmi_rope_table": ([vp, vp, vp, vp, i64, i32, i32, vp], i32),
    "fluxmi_qkv_rope": ([vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    "fluxmi_build_quant_lut": ([vp, i32, i32, vp, vp], i32),
    "fluxmi_pair_rows": ([vp, vp, i32, i64, vp], i32),
    "fluxmi_unpair_rows": ([vp, vp, i32, i64, vp], i32),
    "fluxmi_im2col3x3": ([vp, vp, i32, i32, i32, i32, i32, vp], i32),
    "fluxmi_conv3x3": ([vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    "fluxmi_groupnorm": ([vp, vp, vp, vp, vp, i32, i32, i32, i32, C.c_float, vp], i32),
    "fluxmi_softmax_rows": ([vp, vp, i32, i32, i64, C.c_float, vp], i32),
    "fluxmi_row_norm": ([vp, vp, vp, vp, i32, i32, i64, i64, C.c_float, i32, vp], i32),
    "fluxmi_act_mul": ([vp, vp, i32, i32, i64, i64, i32, vp], i32),
    "fluxmi_text_attention": ([vp, vp, i64, vp, i64, vp, i64, vp, i32, vp, C.c_float, i32, i32, i32, i32, vp], i32),
    "fluxmi_attention": ([vp, vp, vp, vp, i64, i32, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    "fluxmi_attention_debug_buffer": ([vp], i32),
    "fluxmi_attention_plan": ([i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_ulonglong)], i32),
    "fluxmi_attention_rawq": ([vp, i64, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    "fluxmi_timestep_embedding": ([vp, vp, vp, i32, i32, f32, vp], i32),
    "fluxmi_euler": ([vp, vp, vp, vp, i64, vp], i32),
    "fluxmi_engine_num_linears": ([C.POINTER(ModelDesc)], i32),
    "fluxmi_engine_create": ([C.POINTER(ModelDesc), C.POINTER(Linear), i32, C.POINTER(vp), i32, C.POINTER(vp)], i32),
    "fluxmi_engine_destroy": ([vp], i32),
    "fluxmi_engine_rebind": ([vp, C.POINTER(Linear), i32], i32),
    "fluxmi_engine_set_tables": ([vp, C.POINTER(f32), C.POINTER(f32), C.POINTER(i32)], i32),
    "fluxmi_engine_prepare": ([vp, i32, i32, i32, vp, vp, vp], i32),
    "fluxmi_engine_forward": ([vp, vp, vp, vp, vp, vp, vp, i32, i32, vp], i32),
    "fluxmi_engine_denoise": ([vp, vp, vp, vp, f32, C.POINTER(C.c_double), i32, C.POINTER(i32), i32, vp], i32),
    "fluxmi_engine_workspace_bytes": ([vp, C.POINTER(i64)], i32),
    "fluxmi_engine_get_buffer": ([vp, C.c_char_p, C.POINTER(vp), C.POINTER(i64)], i32),
    "fluxmi_engine_last_timing": ([vp, C.POINTER(f32), C.POINTER(i32)], i32),
    "fluxmi_engine_set_amax_exchange": ([vp, vp, i32, vp, vp], i32),
    "fluxmi_engine_run_block": ([vp, i32, i32, i32, i32, i32, vp], i32),
    "fluxmi_engine_copy_buffer": ([vp, C.c_char_p, i64, vp, i64, i32, vp], i32),
}
AMAX_HOOK = C.CFUNCTYPE(i32, vp, i32, i32, vp)
EXPORTS = sorted(list(_SIGS) + ["fluxmi_last_error"])

lib.fluxmi_last_error.restype = C.c_char_p
lib.fluxmi_last_error.argtypes = []
for _name, (_args, _res) in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here == header/library mismatch
    _fn.argtypes = _args
    _fn.restype = _res

ABI_VERSION = 5
if lib.fluxmi_abi_version() != ABI_VERSION:
    raise ImportError(f"fluxmi: ABI version mismatch ({lib.fluxmi_abi_version()} != {ABI_VERSION})")


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError("fluxmi: " + lib.fluxmi_last_error().decode("utf-8", "replace"))


def call(name: str, *args) -> None:
    check(getattr(lib, name)(*args))


def get_tuning() -> Tuning:
    t = Tuning()
    check(lib.fluxmi_get_tuning(C.byref(t)))
    return t


def set_tuning(**knobs) -> Tuning:
    """Replace fields of the library's tuning struct (validated by the library); returns the struct that was in force before."""
    old = get_tuning()
    new = get_tuning()
    for k, v in knobs.items():
        if k not in {f[0] for f in Tuning._fields_} or k == "struct_size":
            raise KeyError(f"fluxmi tuning has no knob {k!r}")
        setattr(new, k, v)
    check(lib.fluxmi_set_tuning(C.byref(new)))
    return old


class tuning:
    """`with _lib.tuning(attn_var=2): ...` -- run a block under other kernel-selection knobs, then restore the previous struct."""

    def __init__(self, **knobs):
        self.knobs = knobs

    def __enter__(self):
        self.old = set_tuning(**self.knobs)
        return self

    def __exit__(self, *exc):
        check(lib.fluxmi_set_tuning(C.byref(self.old)))
        return False

"""ctypes binding of libvllm_amd.so (C ABI declared in include/vllm_amd.h).

There is NO fallback: if the HIP library has not been built the import fails
loudly (build it with ``python -c 'import __graft_entry__ as g; g.build()'`` or
``make -C vllm_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VLLM_AMD_LIB: another build of the SAME library (tools/: the -DVA_PROBES measurement build)
LIB_PATH = os.environ.get("VLLM_AMD_LIB") or os.path.join(_HERE, "csrc", "libvllm_amd.so")

F32, F16, BF16 = 0, 1, 2
KV_AUTO, KV_FP8_E4M3 = 0, 1

_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> argtypes; kept in the order of include/vllm_amd.h
SIGNATURES = {
    "vllm_amd_paged_attention_v1": [
        _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp,
        _i64, _i64, _i64, _i, _i, _i, _i, _i, _vp],
    "vllm_amd_paged_attention_v2": [
        _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _i, _i, _i, _i, _vp,
        _i, _i, _vp, _vp, _i64, _i64, _i64, _i, _i, _i, _i, _i, _vp],
    "vllm_amd_paged_attention_rocm": [
        _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _i, _i, _i, _i,
        _vp, _i, _i, _vp, _vp, _vp, _i64, _i64, _i64, _vp],
    "vllm_amd_paged_attention_2d": [
        _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _i, _i, _i, _i,
        _vp, _i, _i, _vp, _vp, _vp, _i64, _i64, _i64, _i, _vp],
    "vllm_amd_reshape_and_cache": [
        _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i64, _i64, _i, _i, _vp, _vp, _vp],
    "vllm_amd_reshape_and_cache_flash": [
        _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i64, _i64, _i64, _i, _i, _vp,
        _vp, _vp],
    "vllm_amd_convert_fp8": [_vp, _vp, _i64, _f, _i, _i, _vp],
    "vllm_amd_copy_blocks": [_vp, _vp, _vp, _i, _i64, _vp],
    "vllm_amd_static_scaled_fp8_quant": [_vp, _vp, _vp, _i64, _i, _i, _vp],
    "vllm_amd_dynamic_scaled_fp8_quant": [_vp, _vp, _vp, _i64, _i, _i, _vp],
    "vllm_amd_dynamic_per_token_scaled_fp8_quant": [_vp, _vp, _vp, _vp, _i64, _i, _i, _vp],
    "vllm_amd_scaled_mm": [
        _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _i, _i, _i, _vp],
    "vllm_amd_set_workspace": [_vp, _i64, _vp],
    "vllm_amd_scaled_mm_static": [
        _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _i, _i, _i, _i, _vp],
    "vllm_amd_mfma_ceiling": [_vp, _i, _vp],
    "vllm_amd_wvSplitKQ": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "vllm_amd_rms_norm_static_fp8_quant": [_vp, _vp, _vp, _vp, _f, _i64, _i, _i64, _i, _vp],
    "vllm_amd_fused_add_rms_norm_static_fp8_quant": [
        _vp, _vp, _vp, _vp, _vp, _f, _i64, _i, _i64, _i, _vp],
    "vllm_amd_rms_norm_dynamic_per_token_quant_int8": [_vp, _vp, _vp, _vp, _f, _vp, _i64, _i, _i, _vp],
    "vllm_amd_rms_norm_dynamic_per_token_quant": [
        _vp, _vp, _vp, _vp, _f, _vp, _vp, _i64, _i, _i, _vp],
    "vllm_amd_silu_and_mul_quant": [_vp, _vp, _vp, _i64, _i, _i, _vp],
    "vllm_amd_compute_slot_mapping": [_vp, _vp, _vp, _vp, _i64, _i, _i, _vp],
    "vllm_amd_swap_blocks": [_vp, _vp, _i64, _vp, _i64, _vp],
    "vllm_amd_context_attention_fwd": [
        _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i,
        _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f, _i, _i,
        _i, _i, _i, _vp],
    "vllm_amd_unified_attention": [
        _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i,
        _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _f, _i, _f, _i, _i, _vp,
        _vp, _vp, _i, _i, _i, _vp],
    "vllm_amd_set_tuning": [C.c_char_p, _i],
    "vllm_amd_get_tuning": [C.c_char_p, C.POINTER(_i)],
    "vllm_amd_ar_init": [C.POINTER(_vp), C.POINTER(_vp), _i64, _vp, _i64, _i, _i, _i],
    "vllm_amd_ar_set_timeout_ms": [_vp, _i64],
    "vllm_amd_ar_get_graph_buffer_ipc_meta": [_vp, _vp, C.POINTER(_i64), _i, C.POINTER(_i)],
    "vllm_amd_ar_register_graph_buffers": [_vp, _vp, C.POINTER(_i64), _i],
    "vllm_amd_ar_register_buffer": [_vp, C.POINTER(_vp)],
    "vllm_amd_ar_all_reduce": [_vp, _vp, _vp, _i64, _i, _vp, _i64, _vp],
    "vllm_amd_ar_error": [_vp, C.POINTER(_i)],
    "vllm_amd_ar_dispose": [_vp],
    "vllm_amd_ar_alloc_shared": [_i64, C.POINTER(_vp), _vp],
    "vllm_amd_ar_open_handle": [_vp, C.POINTER(_vp)],
    "vllm_amd_ar_close_handle": [_vp],
    "vllm_amd_ar_free_shared": [_vp],
}


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP library is not built. Run "
            "`make -C vllm_amd/csrc` (hipcc, gfx950). vllm_amd has no CPU or "
            "eager fallback.")
    # RTLD_GLOBAL: the torch shim (libvllm_amd_torch.so) binds the vllm_amd_* symbols of THIS
    # library, also when VLLM_AMD_LIB names another build of it
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.vllm_amd_last_error.restype = C.c_char_p
    lib.vllm_amd_last_error.argtypes = []
    lib.vllm_amd_version.restype = C.c_int
    lib.vllm_amd_ar_meta_size.restype = C.c_int64
    lib.vllm_amd_ar_meta_size.argtypes = []
    return lib


lib = _load()


def check(rc: int) -> None:
    """Turn a negative status into the RuntimeError the reference's TORCH_CHECK raises."""
    if rc != 0:
        raise RuntimeError(lib.vllm_amd_last_error().decode())


def set_tuning(key: str, value: int) -> None:
    check(lib.vllm_amd_set_tuning(key.encode(), int(value)))

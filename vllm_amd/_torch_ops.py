"""PyTorch-ROCm custom-op layer over the C ABI.

The operators of
  torch.ops._C            (csrc/torch_bindings.cpp:45-69, 117-120, 188-211, 422-427, 575-593)
  torch.ops._C_cache_ops  (csrc/torch_bindings.cpp:676-708)
  torch.ops._rocm_C       (csrc/rocm/torch_bindings.cpp:34-56)
are implemented and registered IN C++ by ``csrc/torch_shim.cpp`` (libvllm_amd_torch.so), with the
reference's exact schema strings, for the CUDA(=HIP) dispatch key: each one unpacks tensors into
raw device pointers / shapes / strides and calls libvllm_amd.so on torch's current stream -- no
allocation, no synchronisation, HIP-graph capturable.  This module only loads that library,
calls its registration entry point, and keeps Python names for the same operators (they forward
to ``torch.ops``) plus the small helpers the ctypes callers of the non-operator entry points
(context_attention_fwd, unified_attention) use.  There is no Python implementation to fall back
to: without libvllm_amd_torch.so the import fails.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib
from ._lib import check, lib

SHIM_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libvllm_amd_torch.so")


def _load_shim() -> C.CDLL:
    if not os.path.exists(SHIM_PATH):
        raise ImportError(
            f"{SHIM_PATH} not found: the torch operator shim is not built. Run "
            "`make -C vllm_amd/csrc` (g++ against this image's torch headers). vllm_amd has no "
            "Python, CPU or eager fallback for its operators.")
    shim = C.CDLL(SHIM_PATH)
    shim.vllm_amd_torch_register.restype = C.c_int
    shim.vllm_amd_torch_register_error.restype = C.c_char_p
    if shim.vllm_amd_torch_register() != 0:
        raise ImportError("vllm_amd torch operator registration failed: "
                          + shim.vllm_amd_torch_register_error().decode())
    return shim


shim = _load_shim()


def _kv(kv_cache_dtype: str) -> int:
    # csrc/quantization/fp8/amd/quant_utils.cuh:641-667 (the shim checks the same strings)
    if kv_cache_dtype == "auto":
        return _lib.KV_AUTO
    if kv_cache_dtype in ("fp8", "fp8_e4m3"):
        return _lib.KV_FP8_E4M3
    raise RuntimeError(f"Unsupported data type of kv cache: {kv_cache_dtype}")


_DT = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise RuntimeError(f"Unsupported data type: {t.dtype}") from None


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


class _dev:
    """at::cuda::OptionalCUDAGuard(device_of(t)) equivalent for the ctypes callers."""

    def __init__(self, t: torch.Tensor):
        self.idx = t.device.index
        self.prev = None

    def __enter__(self):
        cur = torch.cuda.current_device()
        if self.idx is not None and self.idx != cur:
            self.prev = cur
            torch.cuda.set_device(self.idx)

    def __exit__(self, *a):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)


# Python names of the operators (same argument order as the schemas)
paged_attention_v1 = torch.ops._C.paged_attention_v1
paged_attention_v2 = torch.ops._C.paged_attention_v2
paged_attention_rocm = torch.ops._rocm_C.paged_attention
reshape_and_cache = torch.ops._C_cache_ops.reshape_and_cache
reshape_and_cache_flash = torch.ops._C_cache_ops.reshape_and_cache_flash
convert_fp8 = torch.ops._C_cache_ops.convert_fp8
copy_blocks = torch.ops._C_cache_ops.copy_blocks
swap_blocks = torch.ops._C_cache_ops.swap_blocks
static_scaled_fp8_quant = torch.ops._C.static_scaled_fp8_quant
dynamic_scaled_fp8_quant = torch.ops._C.dynamic_scaled_fp8_quant
dynamic_per_token_scaled_fp8_quant = torch.ops._C.dynamic_per_token_scaled_fp8_quant
cutlass_scaled_mm = torch.ops._C.cutlass_scaled_mm
wvSplitKQ = torch.ops._rocm_C.wvSplitKQ
rms_norm_static_fp8_quant = torch.ops._C.rms_norm_static_fp8_quant
fused_add_rms_norm_static_fp8_quant = torch.ops._C.fused_add_rms_norm_static_fp8_quant
rms_norm_dynamic_per_token_quant = torch.ops._C.rms_norm_dynamic_per_token_quant
silu_and_mul_quant = torch.ops._C.silu_and_mul_quant
#: device-side BlockTable.compute_slot_mapping (vllm/v1/worker/block_table.py:84-99); not a
#: reference op -- the reference does this with numpy on the host
compute_slot_mapping = torch.ops._vllm_amd_C.compute_slot_mapping

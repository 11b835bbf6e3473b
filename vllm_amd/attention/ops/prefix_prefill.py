"""Prefill / chunked-prefill attention with a paged prefix: mirror of
vllm/attention/ops/prefix_prefill.py:726-913 (`context_attention_fwd`, same arguments)
over `vllm_amd_context_attention_fwd` (HIP, prefill_attention.hip) instead of the Triton
kernels."""
from __future__ import annotations

from typing import Optional

import torch

from ... import _lib
from ..._lib import check, lib
from ..._torch_ops import _dev, _ptr, _stream

_DT = {torch.float16: _lib.F16, torch.bfloat16: _lib.BF16, torch.float32: _lib.F32}
_FP8 = (torch.uint8, torch.float8_e4m3fn)


def _i32(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == torch.int32 else t.to(torch.int32)


@torch.inference_mode()
def context_attention_fwd(q, k, v, o, kv_cache_dtype: str, k_cache, v_cache, b_loc, b_start_loc,
                          b_seq_len, max_seq_len, max_input_len, k_scale: torch.Tensor,
                          v_scale: torch.Tensor, alibi_slopes=None, sliding_window=None,
                          sm_scale=None, skip_decode=False,
                          fp8_out_scale: Optional[torch.Tensor] = None) -> None:
    if "fp8" in kv_cache_dtype:
        assert k_cache.dtype in _FP8 and v_cache.dtype in _FP8
        if kv_cache_dtype not in ("fp8", "fp8_e4m3"):
            raise ValueError("Unsupported FP8 dtype:", kv_cache_dtype)
        kvd = _lib.KV_FP8_E4M3
    else:
        if k_cache.dtype == torch.uint8 or v_cache.dtype == torch.uint8:
            raise ValueError("kv_cache_dtype='auto' unsupported for FP8 KV Cache prefill kernel")
        kvd = _lib.KV_AUTO
        assert k_cache.dtype == q.dtype and v_cache.dtype == q.dtype
    if q.dtype not in _DT:
        raise RuntimeError(f"context_attention_fwd: unsupported query dtype {q.dtype}")
    Lq, Lk, Lv = q.shape[-1], k.shape[-1], v.shape[-1]
    assert Lq == Lk and Lk == Lv
    if sm_scale is None:
        sm_scale = 1.0 / (Lq ** 0.5)
    batch, head = b_seq_len.shape[0], q.shape[1]
    assert batch + 1 == len(b_start_loc)
    if sliding_window is None or sliding_window <= 0:
        sliding_window = 0
    for t in (q, k, v, o):
        assert t.stride(-1) == 1
    # integer tensors: the kernel reads int32 (vLLM passes int32; the reference's own test
    # passes int64, which Triton accepts)
    b_loc, b_start_loc, b_seq_len = _i32(b_loc), _i32(b_start_loc), _i32(b_seq_len)
    assert b_loc.stride(1) == 1 and b_start_loc.is_contiguous() and b_seq_len.is_contiguous()
    if fp8_out_scale is not None:
        assert o.dtype in _FP8
    else:
        assert o.dtype == q.dtype
    # caches: [num_blocks, kv_heads, D/x, block_size, x] and [num_blocks, kv_heads, D, block_size]
    # x is the engine's 16-byte K vector (PagedAttention.split_kv_cache: x = 16 // element_size); the
    # Triton kernel takes any x (its test builds x = 8 for every dtype), this kernel does not
    if k_cache.shape[4] * k_cache.element_size() != 16:
        raise RuntimeError(f"context_attention_fwd: K cache vector width x={k_cache.shape[4]} with "
                           f"{k_cache.element_size()}-byte elements; x * element_size must be 16")
    block_size = v_cache.shape[3]
    assert k_cache.stride(4) == 1 and k_cache.stride(3) == k_cache.shape[4] \
        and k_cache.stride(2) == k_cache.shape[3] * k_cache.shape[4]
    assert v_cache.stride(3) == 1 and v_cache.stride(2) == block_size
    with _dev(q):
        check(lib.vllm_amd_context_attention_fwd(
            _ptr(o), _ptr(q), _ptr(k), _ptr(v), _ptr(k_cache), _ptr(v_cache), _ptr(b_loc),
            _ptr(b_start_loc), _ptr(b_seq_len), _ptr(k_scale), _ptr(v_scale),
            _ptr(alibi_slopes), _ptr(fp8_out_scale), batch, head, k.shape[1], Lq, block_size,
            int(max_input_len), b_loc.stride(0), q.stride(0), q.stride(1), k.stride(0),
            k.stride(1), v.stride(0), v.stride(1), o.stride(0), o.stride(1), k_cache.stride(0),
            k_cache.stride(1), v_cache.stride(0), v_cache.stride(1), float(sm_scale),
            int(sliding_window), int(bool(skip_decode)), q.shape[0], _DT[q.dtype], kvd, _stream(q)))

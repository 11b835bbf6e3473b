"""Python op surface of the hot path: same names, arguments and error behaviour
as the reference's ``vllm/_custom_ops.py`` (file:line cited per function), so a
caller written against the reference runs unchanged:

    from vllm_amd import _custom_ops as ops

Every function dispatches through ``torch.ops.<ns>.<op>`` (registered in
``vllm_amd/_torch_ops.py``) into the hand-written HIP kernels; there is no
eager / Triton / CPU fallback.
"""
from __future__ import annotations

from typing import Optional, Union

import torch

from . import _torch_ops  # noqa: F401  (registers torch.ops._C / _C_cache_ops / _rocm_C)

# vllm/platforms/rocm.py:387-397 -- gfx950 uses OCP e4m3fn (gfx94x used fnuz)
FP8_DTYPE = torch.float8_e4m3fn


# vllm/_custom_ops.py:39-65
def paged_attention_v1(
    out: torch.Tensor,
    query: torch.Tensor,
    key_cache: torch.Tensor,
    value_cache: torch.Tensor,
    num_kv_heads: int,
    scale: float,
    block_tables: torch.Tensor,
    seq_lens: torch.Tensor,
    block_size: int,
    max_seq_len: int,
    alibi_slopes: Optional[torch.Tensor],
    kv_cache_dtype: str,
    k_scale: torch.Tensor,
    v_scale: torch.Tensor,
    tp_rank: int = 0,
    blocksparse_local_blocks: int = 0,
    blocksparse_vert_stride: int = 0,
    blocksparse_block_size: int = 64,
    blocksparse_head_sliding_step: int = 0,
) -> None:
    torch.ops._C.paged_attention_v1(
        out, query, key_cache, value_cache, num_kv_heads, scale, block_tables, seq_lens,
        block_size, max_seq_len, alibi_slopes, kv_cache_dtype, k_scale, v_scale, tp_rank,
        blocksparse_local_blocks, blocksparse_vert_stride, blocksparse_block_size,
        blocksparse_head_sliding_step)


# vllm/_custom_ops.py:68-97
def paged_attention_v2(
    out: torch.Tensor,
    exp_sum: torch.Tensor,
    max_logits: torch.Tensor,
    tmp_out: torch.Tensor,
    query: torch.Tensor,
    key_cache: torch.Tensor,
    value_cache: torch.Tensor,
    num_kv_heads: int,
    scale: float,
    block_tables: torch.Tensor,
    seq_lens: torch.Tensor,
    block_size: int,
    max_seq_len: int,
    alibi_slopes: Optional[torch.Tensor],
    kv_cache_dtype: str,
    k_scale: torch.Tensor,
    v_scale: torch.Tensor,
    tp_rank: int = 0,
    blocksparse_local_blocks: int = 0,
    blocksparse_vert_stride: int = 0,
    blocksparse_block_size: int = 64,
    blocksparse_head_sliding_step: int = 0,
) -> None:
    torch.ops._C.paged_attention_v2(
        out, exp_sum, max_logits, tmp_out, query, key_cache, value_cache, num_kv_heads, scale,
        block_tables, seq_lens, block_size, max_seq_len, alibi_slopes, kv_cache_dtype, k_scale,
        v_scale, tp_rank, blocksparse_local_blocks, blocksparse_vert_stride,
        blocksparse_block_size, blocksparse_head_sliding_step)


# vllm/_custom_ops.py:100-126
def paged_attention_rocm(
    out: torch.Tensor,
    exp_sum: torch.Tensor,
    max_logits: torch.Tensor,
    tmp_out: torch.Tensor,
    query: torch.Tensor,
    key_cache: torch.Tensor,
    value_cache: torch.Tensor,
    num_kv_heads: int,
    scale: float,
    block_tables: torch.Tensor,
    seq_lens: torch.Tensor,
    query_start_loc: Optional[torch.Tensor],
    block_size: int,
    max_seq_len: int,
    alibi_slopes: Optional[torch.Tensor],
    kv_cache_dtype: str,
    k_scale: torch.Tensor,
    v_scale: torch.Tensor,
    fp8_out_scale: Optional[torch.Tensor] = None,
) -> None:
    torch.ops._rocm_C.paged_attention(out, exp_sum, max_logits, tmp_out, query, key_cache,
                                      value_cache, num_kv_heads, scale, block_tables, seq_lens,
                                      query_start_loc, block_size, max_seq_len, alibi_slopes,
                                      kv_cache_dtype, k_scale, v_scale, fp8_out_scale)


# vllm/_custom_ops.py:1617-1629
def reshape_and_cache(key, value, key_cache, value_cache, slot_mapping, kv_cache_dtype: str,
                      k_scale: torch.Tensor, v_scale: torch.Tensor) -> None:
    torch.ops._C_cache_ops.reshape_and_cache(key, value, key_cache, value_cache, slot_mapping,
                                             kv_cache_dtype, k_scale, v_scale)


# vllm/_custom_ops.py:1632-1645
def reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping,
                            kv_cache_dtype: str, k_scale: torch.Tensor,
                            v_scale: torch.Tensor) -> None:
    torch.ops._C_cache_ops.reshape_and_cache_flash(key, value, key_cache, value_cache,
                                                   slot_mapping, kv_cache_dtype, k_scale, v_scale)


# vllm/_custom_ops.py:1661-1664
def copy_blocks(key_caches: list[torch.Tensor], value_caches: list[torch.Tensor],
                block_mapping: torch.Tensor) -> None:
    torch.ops._C_cache_ops.copy_blocks(key_caches, value_caches, block_mapping)


# vllm/_custom_ops.py:1672-1674
def swap_blocks(src: torch.Tensor, dst: torch.Tensor, block_mapping: torch.Tensor) -> None:
    torch.ops._C_cache_ops.swap_blocks(src, dst, block_mapping)


# vllm/_custom_ops.py:1677-1681
def convert_fp8(output: torch.Tensor, input: torch.Tensor, scale: float = 1.0,
                kv_dtype: str = "fp8") -> None:
    torch.ops._C_cache_ops.convert_fp8(output, input, scale, kv_dtype)


# vllm/_custom_ops.py:1244-1304
def scaled_fp8_quant(
    input: torch.Tensor,
    scale: Optional[torch.Tensor] = None,
    num_token_padding: Optional[int] = None,
    scale_ub: Optional[torch.Tensor] = None,
    use_per_token_if_dynamic: bool = False,
    output: Optional[torch.Tensor] = None,
) -> tuple[torch.Tensor, torch.Tensor]:
    """Quantize ``input`` [num_tokens, hidden] to fp8 e4m3fn; returns (output, scale).

    scale given -> static per-tensor; else dynamic per-tensor, or per-token when
    ``use_per_token_if_dynamic``.  ``num_token_padding`` pads the first output
    dimension (rows beyond num_tokens are left uninitialised, as in the reference).
    """
    assert input.ndim == 2
    shape: Union[tuple[int, int], torch.Size] = input.shape
    out_dtype = FP8_DTYPE
    if num_token_padding:
        shape = (max(num_token_padding, input.shape[0]), shape[1])
    if output is None:
        output = torch.empty(shape, device=input.device, dtype=out_dtype)
    else:
        assert num_token_padding is None, "padding not supported if output passed in"
        assert output.dtype == out_dtype

    if scale is None:
        if use_per_token_if_dynamic:
            scale = torch.empty((shape[0], 1), device=input.device, dtype=torch.float32)
            torch.ops._C.dynamic_per_token_scaled_fp8_quant(output, input.contiguous(), scale,
                                                            scale_ub)
        else:
            scale = torch.zeros(1, device=input.device, dtype=torch.float32)
            torch.ops._C.dynamic_scaled_fp8_quant(output, input.contiguous(), scale)
    else:
        assert scale.numel() == 1, f"{scale.shape}"
        torch.ops._C.static_scaled_fp8_quant(output, input.contiguous(), scale)
    return output, scale


# vllm/_custom_ops.py:697-742
def cutlass_scaled_mm(a: torch.Tensor, b: torch.Tensor, scale_a: torch.Tensor,
                      scale_b: torch.Tensor, out_dtype: torch.dtype,
                      bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """output = torch.mm((scale_a * a), (scale_b * b)).to(out_dtype) (+ bias), with
    scale_a per-tensor [1] or per-token [M,1], scale_b per-tensor or per-channel [N,1].
    a: fp8 [M,K] row-major; b: fp8 [K,N] column-major (``weight.t()``)."""
    assert out_dtype is torch.bfloat16 or out_dtype is torch.float16
    assert bias is None or (bias.shape[0] == b.shape[1] and bias.dtype == out_dtype)
    m, n = a.shape[0], b.shape[1]
    out = torch.empty((m, n), dtype=out_dtype, device=a.device)
    torch.ops._C.cutlass_scaled_mm(out, a, b, scale_a, scale_b, bias)
    return out


#: the name BASELINE.json / the north star use for the same operator
scaled_mm = cutlass_scaled_mm


# vllm/_custom_ops.py:1501-1508
def wvSplitKQ(a: torch.Tensor, b: torch.Tensor, out_dtype: torch.dtype, scale_a: torch.Tensor,
              scale_b: torch.Tensor, cu_count: int) -> torch.Tensor:
    out = torch.empty((b.shape[0], a.shape[0]), dtype=out_dtype, device=b.device)
    torch.ops._rocm_C.wvSplitKQ(a, b, out, scale_a, scale_b, cu_count)
    return out


# ---- fused neighbours of the GEMM (vllm/_custom_ops.py:283-293, 368-384) -----------------
def scaled_rms_norm(out: torch.Tensor, input: torch.Tensor, weight: torch.Tensor,
                    scale: torch.Tensor, epsilon: float) -> None:
    torch.ops._C.rms_norm_static_fp8_quant(out, input, weight, scale, epsilon)


def scaled_fused_add_rms_norm(out: torch.Tensor, input: torch.Tensor, residual: torch.Tensor,
                              weight: torch.Tensor, scale: torch.Tensor, epsilon: float) -> None:
    torch.ops._C.fused_add_rms_norm_static_fp8_quant(out, input, residual, weight, scale, epsilon)


def rms_norm_dynamic_per_token_quant(
        input: torch.Tensor, weight: torch.Tensor, epsilon: float, quant_dtype: torch.dtype,
        scale_ub: Optional[torch.Tensor] = None,
        residual: Optional[torch.Tensor] = None) -> tuple[torch.Tensor, torch.Tensor]:
    output = torch.empty_like(input, dtype=quant_dtype)
    scales = torch.empty((input.numel() // input.shape[-1], 1), device=input.device,
                         dtype=torch.float32)
    torch.ops._C.rms_norm_dynamic_per_token_quant(output, input, weight, scales, epsilon,
                                                  scale_ub, residual)
    return output, scales


def silu_and_mul_quant(out: torch.Tensor, input: torch.Tensor, scale: torch.Tensor) -> None:
    torch.ops._C.silu_and_mul_quant(out, input, scale)

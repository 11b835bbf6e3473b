"""Mirror of vllm/attention/ops/chunked_prefill_paged_decode.py:215-380
(`chunked_prefill_paged_decode`, same arguments): the prefill chunks of a step go through
`context_attention_fwd(skip_decode=True)`, the one-token sequences through the ROCm paged
attention op over the same x-layout cache -- both HIP kernels here (prefill_attention.hip,
paged_attention_mfma.hip).  The reference takes `ops.paged_attention_rocm` only where
`use_rocm_custom_paged_attention` allows (platforms/rocm.py:121-158) and a Triton kernel
(`kernel_paged_attention_2d`, :27-212) elsewhere; this op covers every head size / block
size / GQA ratio of paged_attention_v1/v2, so it always takes the former.  With a sliding
window (the Triton kernel's `SLIDING_WINDOW`, :169-171; the reference's predicate sends such
calls to Triton) the decode tokens take `vllm_amd_paged_attention_2d` = the same kernels with
the keys in front of `seq_len - sliding_window` masked and their partitions skipped."""
from __future__ import annotations

import torch

from ... import _custom_ops as ops
from ... import _lib
from ..._lib import check, lib
from ..._torch_ops import _dev, _dt, _kv, _ptr, _stream
from .prefix_prefill import context_attention_fwd

_PARTITION_SIZE_ROCM = 256   # chunked_prefill_paged_decode.py:304


def chunked_prefill_paged_decode(query, key, value, output, kv_cache_dtype, key_cache, value_cache,
                                 block_table, query_start_loc, seq_lens, max_seq_len, max_query_len,
                                 k_scale, v_scale, alibi_slopes=None, sliding_window=None,
                                 sm_scale=None, output_scale=None) -> None:
    if sm_scale is None:
        sm_scale = 1.0 / (query.shape[1] ** 0.5)   # (sic) :238
    if sliding_window is None or sliding_window <= 0:
        sliding_window = 0

    if max_query_len > 1:
        context_attention_fwd(q=query, k=key, v=value, o=output, kv_cache_dtype=kv_cache_dtype,
                              k_cache=key_cache, v_cache=value_cache, b_loc=block_table,
                              b_start_loc=query_start_loc, b_seq_len=seq_lens,
                              max_seq_len=max_seq_len, max_input_len=max_query_len,
                              k_scale=k_scale, v_scale=v_scale, alibi_slopes=alibi_slopes,
                              sliding_window=sliding_window, sm_scale=sm_scale, skip_decode=True,
                              fp8_out_scale=output_scale)

    if "fp8" in kv_cache_dtype and kv_cache_dtype not in ("fp8", "fp8_e4m3"):
        raise ValueError("Unsupported FP8 dtype:", kv_cache_dtype)
    block_size = value_cache.shape[3]
    num_query_heads, head_size = query.shape[1], query.shape[2]
    max_num_partitions = (max_seq_len + _PARTITION_SIZE_ROCM - 1) // _PARTITION_SIZE_ROCM
    assert _PARTITION_SIZE_ROCM % block_size == 0
    total_num_seq = block_table.shape[0]
    tmp_output = torch.empty((total_num_seq, num_query_heads, max_num_partitions, head_size),
                             dtype=query.dtype, device=output.device)
    exp_sums = torch.empty((total_num_seq, num_query_heads, max_num_partitions),
                           dtype=torch.float32, device=output.device)
    max_logits = torch.empty_like(exp_sums)
    if sliding_window > 0:
        assert block_table.dtype == torch.int32 and seq_lens.dtype == torch.int32
        assert query_start_loc.dtype == torch.int32 and output.is_contiguous()
        with _dev(query):
            check(lib.vllm_amd_paged_attention_2d(
                _ptr(output), _ptr(exp_sums), _ptr(max_logits), _ptr(tmp_output), _ptr(query),
                _ptr(key_cache), _ptr(value_cache), total_num_seq, num_query_heads, head_size,
                key.shape[1], float(sm_scale), _ptr(block_table), _ptr(seq_lens),
                _ptr(query_start_loc), block_table.shape[1], block_size, int(max_seq_len),
                max_num_partitions, _ptr(alibi_slopes), _dt(query), _kv(kv_cache_dtype),
                _ptr(k_scale), _ptr(v_scale), _ptr(output_scale), query.stride(0),
                key_cache.stride(0), key_cache.stride(1), int(sliding_window), _stream(query)))
        return
    ops.paged_attention_rocm(output, exp_sums, max_logits, tmp_output, query, key_cache, value_cache,
                             key.shape[1], scale=sm_scale, block_tables=block_table,
                             seq_lens=seq_lens, query_start_loc=query_start_loc,
                             block_size=block_size, max_seq_len=max_seq_len,
                             alibi_slopes=alibi_slopes, kv_cache_dtype=kv_cache_dtype,
                             k_scale=k_scale, v_scale=v_scale, fp8_out_scale=output_scale)

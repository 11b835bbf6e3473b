"""Host-side mirror of the reference's ``PagedAttention`` helper
(vllm/attention/ops/paged_attn.py:36-256): KV-cache shape / split, cache write
the v1/v2 decode heuristic with the reference's buffer shapes, and
`forward_prefix` (a Triton kernel in the reference, a HIP kernel here: SURVEY.md 8f-4)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch

from .. import _custom_ops as ops

# Same as PARTITION_SIZE of the v2 launcher (paged_attn.py:16)
_PARTITION_SIZE = 512
# partition of the ROCm custom op (csrc/rocm/attention.cu:3126, rocm_flash_attn.py:900)
_PARTITION_SIZE_ROCM = 256


@dataclass
class PagedAttentionMetadata:
    """paged_attn.py:19-33"""
    seq_lens_tensor: Optional[torch.Tensor]
    max_decode_seq_len: int
    block_tables: Optional[torch.Tensor]


class PagedAttention:

    @staticmethod
    def get_supported_head_sizes() -> List[int]:
        return [32, 64, 80, 96, 112, 120, 128, 192, 256]

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int,
                           head_size: int) -> Tuple[int, ...]:
        return (2, num_blocks, block_size * num_kv_heads * head_size)

    @staticmethod
    def split_kv_cache(kv_cache: torch.Tensor, num_kv_heads: int,
                       head_size: int) -> Tuple[torch.Tensor, torch.Tensor]:
        x = 16 // kv_cache.element_size()
        num_blocks = kv_cache.shape[1]
        key_cache = kv_cache[0].view(num_blocks, num_kv_heads, head_size // x, -1, x)
        value_cache = kv_cache[1].view(num_blocks, num_kv_heads, head_size, -1)
        return key_cache, value_cache

    @staticmethod
    def write_to_paged_cache(key, value, key_cache, value_cache, slot_mapping,
                             kv_cache_dtype: str, k_scale, v_scale) -> None:
        ops.reshape_and_cache(key, value, key_cache, value_cache, slot_mapping.flatten(),
                              kv_cache_dtype, k_scale, v_scale)

    @staticmethod
    def forward_decode(query, key_cache, value_cache, block_tables, seq_lens, max_seq_len: int,
                       kv_cache_dtype: str, num_kv_heads: int, scale: float,
                       alibi_slopes: Optional[torch.Tensor], k_scale, v_scale, tp_rank: int = 0,
                       blocksparse_local_blocks: int = 0, blocksparse_vert_stride: int = 0,
                       blocksparse_block_size: int = 64,
                       blocksparse_head_sliding_step: int = 0) -> torch.Tensor:
        if blocksparse_vert_stride is not None and blocksparse_vert_stride > 1:
            block_size = value_cache.size(-1)
            assert blocksparse_block_size > 0 and blocksparse_block_size % block_size == 0, (
                f"{blocksparse_block_size=} needs to be a multiple of {block_size=} used in "
                "block_tables.")
        output = torch.empty_like(query)
        block_size = value_cache.shape[3]
        num_seqs, num_heads, head_size = query.shape
        max_num_partitions = (max_seq_len + _PARTITION_SIZE - 1) // _PARTITION_SIZE
        # reference heuristic, paged_attn.py:129-130 ...
        use_v1 = (max_seq_len <= 8192
                  and (max_num_partitions == 1 or num_seqs * num_heads > 512))
        # ... refined for this implementation, whose decode kernel takes one workgroup per (sequence, kv head, 16 q heads of
        # the group) where the reference's takes one per q head: below one workgroup per CU an unpartitioned launch leaves
        # CUs idle (128 sequences x 4096 tokens, 8q / 1kv heads per GPU -- a TP-8 shard: 31.8 us as v1, 27 us partitioned)
        if use_v1 and max_num_partitions > 1:
            groups = num_kv_heads * ((num_heads // max(num_kv_heads, 1) + 15) // 16)
            use_v1 = num_seqs * groups >= 256
        if use_v1:
            ops.paged_attention_v1(output, query, key_cache, value_cache, num_kv_heads, scale,
                                   block_tables, seq_lens, block_size, max_seq_len, alibi_slopes,
                                   kv_cache_dtype, k_scale, v_scale, tp_rank,
                                   blocksparse_local_blocks, blocksparse_vert_stride,
                                   blocksparse_block_size, blocksparse_head_sliding_step)
        else:
            assert _PARTITION_SIZE % block_size == 0
            tmp_output = torch.empty((num_seqs, num_heads, max_num_partitions, head_size),
                                     dtype=output.dtype, device=output.device)
            exp_sums = torch.empty((num_seqs, num_heads, max_num_partitions),
                                   dtype=torch.float32, device=output.device)
            max_logits = torch.empty_like(exp_sums)
            ops.paged_attention_v2(output, exp_sums, max_logits, tmp_output, query, key_cache,
                                   value_cache, num_kv_heads, scale, block_tables, seq_lens,
                                   block_size, max_seq_len, alibi_slopes, kv_cache_dtype, k_scale,
                                   v_scale, tp_rank, blocksparse_local_blocks,
                                   blocksparse_vert_stride, blocksparse_block_size,
                                   blocksparse_head_sliding_step)
        return output

    @staticmethod
    def forward_decode_rocm(query, key_cache, value_cache, block_tables, seq_lens,
                            max_seq_len: int, kv_cache_dtype: str, num_kv_heads: int, scale: float,
                            alibi_slopes, k_scale, v_scale, query_start_loc=None,
                            fp8_out_scale=None, output=None) -> torch.Tensor:
        """Decode branch of ROCmFlashAttentionImpl.forward
        (vllm/attention/backends/rocm_flash_attn.py:883-941): temporaries with
        partition 256, then ops.paged_attention_rocm."""
        num_seqs, num_heads, head_size = query.shape
        block_size = value_cache.shape[3]
        if output is None:
            output = torch.empty_like(
                query, dtype=(torch.float8_e4m3fn if fp8_out_scale is not None else query.dtype))
        nparts = (max_seq_len + _PARTITION_SIZE_ROCM - 1) // _PARTITION_SIZE_ROCM
        nseq_bt = block_tables.shape[0]
        tmp_output = torch.empty((nseq_bt, num_heads, nparts, head_size), dtype=query.dtype,
                                 device=query.device)
        exp_sums = torch.empty((nseq_bt, num_heads, nparts), dtype=torch.float32,
                               device=query.device)
        max_logits = torch.empty_like(exp_sums)
        ops.paged_attention_rocm(output, exp_sums, max_logits, tmp_output, query, key_cache,
                                 value_cache, num_kv_heads, scale, block_tables, seq_lens,
                                 query_start_loc, block_size, max_seq_len, alibi_slopes,
                                 kv_cache_dtype, k_scale, v_scale, fp8_out_scale)
        return output

    @staticmethod
    def forward_prefix(query, key, value, kv_cache_dtype: str, key_cache, value_cache,
                       block_tables, query_start_loc, seq_lens_tensor, max_query_len: int,
                       alibi_slopes: Optional[torch.Tensor], sliding_window: Optional[int],
                       k_scale: torch.Tensor, v_scale: torch.Tensor) -> torch.Tensor:
        """paged_attn.py:196-239: chunked prefill over the cached prefix (HIP kernel instead of
        the reference's Triton context_attention_fwd)."""
        from .ops.prefix_prefill import context_attention_fwd
        output = torch.empty_like(query)
        context_attention_fwd(query, key, value, output, kv_cache_dtype, key_cache, value_cache,
                              block_tables, query_start_loc, seq_lens_tensor, None, max_query_len,
                              k_scale, v_scale, alibi_slopes, sliding_window)
        return output

    @staticmethod
    def swap_blocks(src_kv_cache, dst_kv_cache, src_to_dst) -> None:
        ops.swap_blocks(src_kv_cache[0], dst_kv_cache[0], src_to_dst)
        ops.swap_blocks(src_kv_cache[1], dst_kv_cache[1], src_to_dst)

    @staticmethod
    def copy_blocks(kv_caches: List[torch.Tensor], src_to_dists: torch.Tensor) -> None:
        key_caches = [kv_cache[0] for kv_cache in kv_caches]
        value_caches = [kv_cache[1] for kv_cache in kv_caches]
        ops.copy_blocks(key_caches, value_caches, src_to_dists)

"""Host-side mirror of the V0 engine's ROCm attention backend
(vllm/attention/backends/rocm_flash_attn.py:60-110 backend class, :112-263 metadata,
:462-980 impl) for DECODER self-attention -- the only attention type on the hot path
(SURVEY.md 8a-3/4, 8f-4).  Same class / method names, argument order and error texts, so a
caller written against the reference's backend runs against this one; the vllm package is not
importable in this image, so the classes are duck-typed stand-ins that take the reference's
bases in tree.

What differs from the reference, on purpose:
  * prompt runs without a cached prefix go to the same HIP prefill kernel as prompt runs with
    one (`context_attention_fwd` with every context length 0) instead of Triton flash
    attention / CK flash attention / SDPA (`:779-860`): one kernel, window + ALiBi + fp8 output
    in all cases;
  * the decode predicate (`use_rocm_custom_paged_attention`, platforms/rocm.py:121-158) is the
    reference's for gfx9 with the head-size clause widened to what `paged_attention_rocm`
    covers here (head sizes 32 .. 128 instead of 64 / 128); everything else takes
    `PagedAttention.forward_decode` as in the reference (`:942-977`);
  * encoder / encoder-decoder / encoder-only attention types raise NotImplementedError.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple, Type

import torch

from ... import _custom_ops as ops
from ..paged_attn import _PARTITION_SIZE_ROCM, PagedAttention, PagedAttentionMetadata


class AttentionType:
    """vllm/attention/backends/abstract.py: the string constants the impl is built with."""
    DECODER = "decoder"
    ENCODER = "encoder"
    ENCODER_ONLY = "encoder_only"
    ENCODER_DECODER = "encoder_decoder"


def use_rocm_custom_paged_attention(qtype: torch.dtype, head_size: int, block_size: int,
                                    gqa_ratio: int, max_seq_len: int, sliding_window,
                                    kv_cache_dtype: str, alibi_slopes=None) -> bool:
    """platforms/rocm.py:121-158, gfx9 branch, V0 engine (no sliding-window clause).  Head sizes:
    the reference's op takes 64 / 128; this library's takes every multiple of 16 up to 128."""
    return (qtype in (torch.half, torch.bfloat16) and head_size in (32, 64, 80, 96, 112, 128)
            and block_size in (16, 32) and 1 <= gqa_ratio <= 16 and max_seq_len <= 128 * 1024)


class ROCmFlashAttentionBackend:
    accept_output_buffer: bool = True

    @staticmethod
    def get_name() -> str:
        return "ROCM_FLASH"

    @staticmethod
    def get_impl_cls() -> Type["ROCmFlashAttentionImpl"]:
        return ROCmFlashAttentionImpl

    @staticmethod
    def get_metadata_cls() -> Type["ROCmFlashAttentionMetadata"]:
        return ROCmFlashAttentionMetadata

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int,
                           head_size: int) -> Tuple[int, ...]:
        return PagedAttention.get_kv_cache_shape(num_blocks, block_size, num_kv_heads, head_size)

    @staticmethod
    def swap_blocks(src_kv_cache: torch.Tensor, dst_kv_cache: torch.Tensor,
                    src_to_dst: torch.Tensor) -> None:
        PagedAttention.swap_blocks(src_kv_cache, dst_kv_cache, src_to_dst)

    @staticmethod
    def copy_blocks(kv_caches: List[torch.Tensor], src_to_dists: torch.Tensor) -> None:
        PagedAttention.copy_blocks(kv_caches, src_to_dists)


@dataclass
class ROCmFlashAttentionMetadata(PagedAttentionMetadata):
    """The decoder fields of rocm_flash_attn.py:112-181 (+ the AttentionMetadata base fields
    num_prefills / num_prefill_tokens / num_decode_tokens / slot_mapping).  A batch is
    |prefill tokens|decode tokens|; `prefill_metadata` / `decode_metadata` cut it (:183-263)."""
    num_prefills: int = 0
    num_prefill_tokens: int = 0
    num_decode_tokens: int = 0
    slot_mapping: Optional[torch.Tensor] = None
    seq_lens: Optional[List[int]] = None
    max_prefill_seq_len: int = 0
    use_cuda_graph: bool = False
    max_query_len: Optional[int] = None
    query_start_loc: Optional[torch.Tensor] = None
    seq_start_loc: Optional[torch.Tensor] = None
    context_lens_tensor: Optional[torch.Tensor] = None
    max_decode_query_len: Optional[int] = None
    _cached_prefill_metadata: Optional["ROCmFlashAttentionMetadata"] = None
    _cached_decode_metadata: Optional["ROCmFlashAttentionMetadata"] = None

    @property
    def prefill_metadata(self) -> Optional["ROCmFlashAttentionMetadata"]:
        if self.num_prefills == 0:
            return None
        if self._cached_prefill_metadata is None:
            assert self.seq_lens is not None and self.seq_lens_tensor is not None
            assert self.block_tables is not None
            n = self.num_prefills
            cut = lambda t, m: None if t is None else t[:m]   # noqa: E731
            self._cached_prefill_metadata = ROCmFlashAttentionMetadata(
                seq_lens_tensor=self.seq_lens_tensor[:n], max_decode_seq_len=0,
                block_tables=self.block_tables[:n], num_prefills=n,
                num_prefill_tokens=self.num_prefill_tokens, num_decode_tokens=0,
                slot_mapping=cut(self.slot_mapping, self.num_prefill_tokens),
                seq_lens=self.seq_lens[:n], max_prefill_seq_len=self.max_prefill_seq_len,
                use_cuda_graph=False, max_query_len=self.max_query_len,
                query_start_loc=cut(self.query_start_loc, n + 1),
                seq_start_loc=cut(self.seq_start_loc, n + 1),
                context_lens_tensor=cut(self.context_lens_tensor, n))
        return self._cached_prefill_metadata

    @property
    def decode_metadata(self) -> Optional["ROCmFlashAttentionMetadata"]:
        if self.num_decode_tokens == 0:
            return None
        if self._cached_decode_metadata is None:
            assert self.block_tables is not None and self.seq_lens_tensor is not None
            n = self.num_prefills
            self._cached_decode_metadata = ROCmFlashAttentionMetadata(
                seq_lens_tensor=self.seq_lens_tensor[n:],
                max_decode_seq_len=self.max_decode_seq_len, block_tables=self.block_tables[n:],
                num_prefills=0, num_prefill_tokens=0, num_decode_tokens=self.num_decode_tokens,
                slot_mapping=None if self.slot_mapping is None
                else self.slot_mapping[self.num_prefill_tokens:],
                seq_lens=None, max_prefill_seq_len=0, use_cuda_graph=self.use_cuda_graph)
        return self._cached_decode_metadata


class ROCmFlashAttentionImpl:
    """rocm_flash_attn.py:462-980.  Token layout of a step:
    |<- num_prefill_tokens: prompt_0 .. prompt_N-1 ->|<- num_decode_tokens: one per sequence ->|"""

    def __init__(self, num_heads: int, head_size: int, scale: float, num_kv_heads: int,
                 alibi_slopes: Optional[List[float]], sliding_window: Optional[int],
                 kv_cache_dtype: str, logits_soft_cap: Optional[float] = None,
                 attn_type: str = AttentionType.DECODER,
                 kv_sharing_target_layer_name: Optional[str] = None,
                 use_irope: bool = False) -> None:
        if kv_sharing_target_layer_name is not None:
            raise NotImplementedError("KV sharing is not supported in V0 ROCM_FLASH backend.")
        if attn_type != AttentionType.DECODER:
            raise NotImplementedError("vllm_amd mirrors the ROCM_FLASH backend for decoder "
                                      f"self-attention only (got attn_type={attn_type!r})")
        if logits_soft_cap:
            # :545-551: the reference's default (Triton) path refuses soft capping as well
            raise ValueError("ROCm Triton FlashAttention does not support attention logits soft "
                             "capping.")
        self.logits_soft_cap = 0.0
        self.attn_type = attn_type
        self.num_heads = num_heads
        self.head_size = head_size
        self.scale = float(scale)
        self.num_kv_heads = num_kv_heads
        self.alibi_slopes = (None if alibi_slopes is None
                             else torch.as_tensor(alibi_slopes, dtype=torch.float32))
        self.sliding_window = ((sliding_window, sliding_window)
                               if sliding_window is not None else (-1, -1))
        self.kv_cache_dtype = kv_cache_dtype
        self.num_queries_per_kv = self.num_heads // self.num_kv_heads
        self.paged_attn_module = PagedAttention
        supported_head_sizes = PagedAttention.get_supported_head_sizes()
        if head_size not in supported_head_sizes:
            raise ValueError(f"Head size {head_size} is not supported by PagedAttention. "
                             f"Supported head sizes are: {supported_head_sizes}.")
        self._no_cache = None   # stand-in cache / block table of prompt runs without a prefix

    def fused_output_quant_supported(self, dtype: torch.dtype, static: bool, group_shape) -> bool:
        # :599-606 (true for the Triton path only); per-tensor static fp8 is what both the
        # prefill kernel and paged_attention_rocm write here
        return dtype == ops.FP8_DTYPE and static and str(group_shape).upper().endswith("PER_TENSOR")

    def _slopes(self, device):
        if self.alibi_slopes is not None and self.alibi_slopes.device != device:
            self.alibi_slopes = self.alibi_slopes.to(device)
        return self.alibi_slopes

    def _prompt_without_prefix(self, query, key, value, out, prefill_meta, layer, output_scale):
        """:779-860.  q and k are the prompts, same lengths: the prefill kernel with an empty
        context (it never reads the stand-in cache)."""
        dev = query.device
        fp8 = "fp8" in self.kv_cache_dtype
        tag = (dev, query.dtype, fp8)
        if self._no_cache is None or self._no_cache[0] != tag:
            cdt = torch.uint8 if fp8 else query.dtype
            x = 16 // torch.empty((), dtype=cdt).element_size()
            kc = torch.zeros((1, self.num_kv_heads, self.head_size // x, 16, x), dtype=cdt, device=dev)
            vc = torch.zeros((1, self.num_kv_heads, self.head_size, 16), dtype=cdt, device=dev)
            self._no_cache = (tag, kc, vc)
        _, kc, vc = self._no_cache
        n = prefill_meta.num_prefills
        b_loc = torch.zeros((n, 1), dtype=torch.int32, device=dev)
        start = prefill_meta.seq_start_loc if prefill_meta.query_start_loc is None \
            else prefill_meta.query_start_loc
        lens = (start[1:] - start[:-1]).to(torch.int32)
        from ..ops.prefix_prefill import context_attention_fwd
        context_attention_fwd(query, key, value, out, self.kv_cache_dtype, kc, vc, b_loc, start,
                              lens, None, prefill_meta.max_prefill_seq_len, layer._k_scale,
                              layer._v_scale, self._slopes(dev), self.sliding_window[0],
                              self.scale, fp8_out_scale=output_scale)

    def forward(self, layer, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                kv_cache: torch.Tensor, attn_metadata: ROCmFlashAttentionMetadata,
                output: Optional[torch.Tensor] = None,
                output_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
        """query [num_tokens, num_heads * head_size], key / value [num_tokens, num_kv_heads *
        head_size], kv_cache [2, num_blocks, block_size * num_kv_heads * head_size] (empty in
        the profiling run); returns output viewed [num_tokens, num_heads * head_size]."""
        assert output is not None, "Output tensor must be provided."
        query = query.view(-1, self.num_heads, self.head_size)
        out3 = output.view(-1, self.num_heads, self.head_size)
        if key is not None:
            assert value is not None
            key = key.view(-1, self.num_kv_heads, self.head_size)
            value = value.view(-1, self.num_kv_heads, self.head_size)
        else:
            assert value is None
        key_cache = value_cache = None
        if kv_cache.numel() > 0:
            key_cache, value_cache = PagedAttention.split_kv_cache(kv_cache, self.num_kv_heads,
                                                                   self.head_size)
            if key is not None and value is not None:
                PagedAttention.write_to_paged_cache(key, value, key_cache, value_cache,
                                                    attn_metadata.slot_mapping,
                                                    self.kv_cache_dtype, layer._k_scale,
                                                    layer._v_scale)
        num_prefill_tokens = attn_metadata.num_prefill_tokens
        decode_query = query[num_prefill_tokens:]
        query = query[:num_prefill_tokens]
        if key is not None and value is not None:
            key = key[:num_prefill_tokens]
            value = value[:num_prefill_tokens]

        if prefill_meta := attn_metadata.prefill_metadata:
            if (kv_cache.numel() == 0 or prefill_meta.block_tables is None
                    or prefill_meta.block_tables.numel() == 0):
                self._prompt_without_prefix(query, key, value, out3[:num_prefill_tokens],
                                            prefill_meta, layer, output_scale)
            else:
                # prefix-enabled attention, :862-881 (forward_prefix allocates its result)
                from ..ops.prefix_prefill import context_attention_fwd
                context_attention_fwd(query, key, value, out3[:num_prefill_tokens],
                                      self.kv_cache_dtype, key_cache, value_cache,
                                      prefill_meta.block_tables, prefill_meta.query_start_loc,
                                      prefill_meta.seq_lens_tensor, None,
                                      prefill_meta.max_query_len, layer._k_scale, layer._v_scale,
                                      self._slopes(query.device), self.sliding_window[0],
                                      self.scale, fp8_out_scale=output_scale)

        if decode_meta := attn_metadata.decode_metadata:
            num_seqs, num_heads, head_size = decode_query.shape
            block_size = value_cache.shape[3]
            use_custom = use_rocm_custom_paged_attention(
                decode_query.dtype, head_size, block_size, num_heads // self.num_kv_heads,
                decode_meta.max_decode_seq_len, self.sliding_window, self.kv_cache_dtype,
                self.alibi_slopes)
            slopes = self._slopes(decode_query.device)
            if use_custom:
                max_seq_len = decode_meta.max_decode_seq_len
                max_num_partitions = (max_seq_len + _PARTITION_SIZE_ROCM - 1) // _PARTITION_SIZE_ROCM
                assert _PARTITION_SIZE_ROCM % block_size == 0
                tmp_output = torch.empty((num_seqs, num_heads, max_num_partitions, head_size),
                                         dtype=decode_query.dtype, device=output.device)
                exp_sums = torch.empty((num_seqs, num_heads, max_num_partitions),
                                       dtype=torch.float32, device=output.device)
                max_logits = torch.empty_like(exp_sums)
                ops.paged_attention_rocm(out3[num_prefill_tokens:], exp_sums, max_logits,
                                         tmp_output, decode_query, key_cache, value_cache,
                                         self.num_kv_heads, self.scale, decode_meta.block_tables,
                                         decode_meta.seq_lens_tensor, None, block_size, max_seq_len,
                                         slopes, self.kv_cache_dtype, layer._k_scale,
                                         layer._v_scale, output_scale)
            else:
                # PagedAttention has no fused output quantisation: quantise afterwards (:943-977)
                out_pa = PagedAttention.forward_decode(
                    decode_query, key_cache, value_cache, decode_meta.block_tables,
                    decode_meta.seq_lens_tensor, decode_meta.max_decode_seq_len,
                    self.kv_cache_dtype, self.num_kv_heads, self.scale, slopes, layer._k_scale,
                    layer._v_scale)
                if output_scale is None:
                    out3[num_prefill_tokens:] = out_pa
                else:
                    out_q = output.view(-1, self.num_heads * self.head_size)
                    ops.scaled_fp8_quant(out_pa.view(-1, self.num_heads * self.head_size),
                                         output_scale, output=out_q[num_prefill_tokens:])
        return output.view(-1, self.num_heads * self.head_size)

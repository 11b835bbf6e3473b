"""Host-side mirror of the V1 engine's attention backend on ROCm
(vllm/v1/attention/backends/triton_attn.py:28-384; selected by
`RocmPlatform.get_attn_backend_cls`, vllm/platforms/rocm.py:177-242): same class and method
names, same arguments, same checks -- with the cache write and the attention itself on the HIP
kernels of this library instead of Triton:

    TritonAttentionImpl.forward  ->  torch.ops._C_cache_ops.reshape_and_cache_flash
                                     + unified_attention                 (default), or
                                 ->  PagedAttention.write_to_paged_cache
                                     + chunked_prefill_paged_decode      (VLLM_V1_USE_PREFILL_DECODE_ATTENTION)

The reference's base classes (`AttentionBackend`, `AttentionImpl`,
`AttentionMetadataBuilder`, vllm/attention/backends/abstract.py:37-305,
vllm/v1/attention/backends/utils.py:32-110) live in the vllm package, which is not importable
here; the classes below are duck-typed stand-ins with the members the engine calls, and a
maintainer re-parents them.  `CommonAttentionMetadata` is taken as any object
with the reference's field names."""
from __future__ import annotations

import os
from dataclasses import field, make_dataclass
from typing import ClassVar, Optional

import torch

from .... import _torch_ops  # noqa: F401  (registers torch.ops._C_cache_ops.* / _rocm_C.*)
from ....attention.ops.chunked_prefill_paged_decode import chunked_prefill_paged_decode
from ....attention.ops.triton_unified_attention import unified_attention
from ....attention.paged_attn import PagedAttention

# vllm/attention/backends/abstract.py:19-34 (AttentionType is a str-constant class there too)
DECODER = "decoder"


# triton_attn.py:28-55.  In tree the reference's own dataclass (and its builder) are used
# unchanged -- nothing in them touches a kernel; here only the field names the impl and the
# engine read are kept (cascade attention is never selected: use_cascade_attention() is False).
TritonAttentionMetadata = make_dataclass("TritonAttentionMetadata", [
    ("num_actual_tokens", int), ("max_query_len", int), ("query_start_loc", torch.Tensor),
    ("max_seq_len", int), ("seq_lens", torch.Tensor), ("block_table", torch.Tensor),
    ("slot_mapping", torch.Tensor), ("use_cascade", bool, field(default=False)),
    ("common_prefix_len", int, field(default=0)),
    ("cu_prefix_query_lens", Optional[torch.Tensor], field(default=None)),
    ("prefix_kv_lens", Optional[torch.Tensor], field(default=None)),
    ("suffix_kv_lens", Optional[torch.Tensor], field(default=None)),
    ("scheduler_metadata", Optional[torch.Tensor], field(default=None)),
    ("prefix_scheduler_metadata", Optional[torch.Tensor], field(default=None))])


class TritonAttentionMetadataBuilder:
    """triton_attn.py:58-137, without the cascade tensors.  `kv_cache_spec` needs `.block_size`;
    `vllm_config` is accepted and not used."""
    full_cudagraph_supported: ClassVar[bool] = True

    def __init__(self, kv_cache_spec, vllm_config, device: torch.device):
        self.device = device
        self.block_size = kv_cache_spec.block_size
        self.kv_cache_spec = kv_cache_spec

    def build_for_cudagraph_capture(self, common_attn_metadata):
        md = self.build(0, common_attn_metadata)
        md.seq_lens.fill_(1)   # capture with seq_lens = max_model_len would crawl (:80-84)
        return md

    def build(self, common_prefix_len: int, common_attn_metadata, fast_build: bool = False):
        m = common_attn_metadata
        return TritonAttentionMetadata(
            num_actual_tokens=m.num_actual_tokens, max_query_len=m.max_query_len,
            query_start_loc=m.query_start_loc, max_seq_len=int(m.seq_lens_cpu.max()),
            seq_lens=m.seq_lens, block_table=m.block_table_tensor, slot_mapping=m.slot_mapping,
            use_cascade=common_prefix_len > 0, common_prefix_len=common_prefix_len)

    def can_run_in_cudagraph(self, common_attn_metadata) -> bool:
        return True   # the ops neither synchronise nor allocate


class TritonAttentionBackend:
    """triton_attn.py:138-197"""
    accept_output_buffer: bool = True

    @classmethod
    def get_supported_dtypes(cls) -> list[torch.dtype]:
        return [torch.float16, torch.bfloat16]

    @classmethod
    def get_supported_head_sizes(cls) -> list[int]:
        # :147-148.  Every size runs on the matrix-core prefill kernels and the bandwidth decode kernels
        # (the prefill + paged-decode path: the paged-attention op's sizes only, i.e. not 160 / 224)
        return [32, 64, 96, 128, 160, 192, 224, 256]

    @classmethod
    def validate_head_size(cls, head_size: int) -> None:
        supported_head_sizes = cls.get_supported_head_sizes()
        if head_size not in supported_head_sizes:
            raise ValueError(f"Head size {head_size} is not supported by "
                             f"{cls.__name__.removesuffix('Backend')}. "
                             f"Supported head sizes are: {supported_head_sizes}.")

    # the selector's accessors (:161-175, :195-197)
    get_name = staticmethod(lambda: "TRITON_ATTN_VLLM_V1")
    get_impl_cls = staticmethod(lambda: TritonAttentionImpl)
    get_metadata_cls = staticmethod(lambda: TritonAttentionMetadata)
    get_builder_cls = staticmethod(lambda: TritonAttentionMetadataBuilder)
    use_cascade_attention = staticmethod(lambda *args, **kwargs: False)

    @staticmethod
    def get_kv_cache_stride_order() -> tuple[int, ...]:
        """Memory order of the cache dims (the hook of `AttentionBackend`, abstract.py:83-85, that
        `GPUModelRunner._reshape_kv_cache_tensors` honours, v1/worker/gpu_model_runner.py:2699-2726;
        the reference's Triton backend does not define it and gets NHD).  Head-major (HND) by
        default here: the logical shape stays `get_kv_cache_shape`, but a (block, kv head) page is
        2 KiB of contiguous memory instead of 128-byte rows 1 KiB apart, which the decode kernel
        streams ~10 % faster.  `VLLM_KV_CACHE_LAYOUT=NHD` restores the
        reference's order (same variable as `get_kv_cache_layout`, v1/attention/backends/utils.py)."""
        layout = os.environ.get("VLLM_KV_CACHE_LAYOUT", "HND")
        if layout == "NHD":
            return (0, 1, 2, 3, 4)
        if layout == "HND":
            return (0, 1, 3, 2, 4)
        raise ValueError(f"Unknown cache layout format {layout}.")

    @staticmethod
    def get_kv_cache_shape(num_blocks: int, block_size: int, num_kv_heads: int,
                           head_size: int) -> tuple[int, ...]:
        if block_size % 16 != 0:
            raise ValueError("Block size must be a multiple of 16.")
        return (2, num_blocks, block_size, num_kv_heads, head_size)


class TritonAttentionImpl:
    """triton_attn.py:200-384"""

    def fused_output_quant_supported(self, dtype: torch.dtype, static: bool,
                                     group_shape: tuple[int, int]):
        return True

    def __init__(self, num_heads: int, head_size: int, scale: float, num_kv_heads: int,
                 alibi_slopes: Optional[list[float]], sliding_window: Optional[int],
                 kv_cache_dtype: str, logits_soft_cap: Optional[float] = None,
                 attn_type: str = DECODER,
                 kv_sharing_target_layer_name: Optional[int] = None) -> None:
        self.num_heads = num_heads
        self.head_size = head_size
        self.scale = float(scale)
        self.num_kv_heads = num_kv_heads
        if alibi_slopes is not None:
            alibi_slopes = torch.tensor(alibi_slopes, dtype=torch.float32)
        self.alibi_slopes = alibi_slopes
        if sliding_window is None:
            self.sliding_window = (-1, -1)
        else:
            self.sliding_window = (sliding_window - 1, 0)
        self.kv_cache_dtype = kv_cache_dtype
        if logits_soft_cap is None:
            logits_soft_cap = 0   # as in flash-attn, 0 means no soft cap
        self.logits_soft_cap = logits_soft_cap
        self.kv_sharing_target_layer_name = kv_sharing_target_layer_name
        self.num_queries_per_kv = self.num_heads // self.num_kv_heads
        TritonAttentionBackend.validate_head_size(head_size)
        if attn_type != DECODER:
            raise NotImplementedError("Encoder self-attention and "
                                      "encoder/decoder cross-attention "
                                      "are not implemented for "
                                      "TritonAttentionImpl")
        self.fp8_dtype = torch.float8_e4m3fn   # current_platform.fp8_dtype() on gfx950
        # envs.VLLM_V1_USE_PREFILL_DECODE_ATTENTION (vllm/envs.py)
        self.force_prefill_decode_attn = os.environ.get(
            "VLLM_V1_USE_PREFILL_DECODE_ATTENTION", "0").lower() in ("1", "true")
        # chunked_prefill_paged_decode = context_attention_fwd + the paged-attention op, which takes the reference's
        # list of head sizes (csrc/attention/paged_attention_v1.cu:158-187: no 160, no 224)
        pd_sizes = [h for h in TritonAttentionBackend.get_supported_head_sizes() if h not in (160, 224)]
        if self.force_prefill_decode_attn and head_size not in pd_sizes:
            raise ValueError(
                f"Head size {head_size} is not supported by TritonAttentionBackend with "
                f"VLLM_V1_USE_PREFILL_DECODE_ATTENTION=1. Supported head sizes there are: {pd_sizes}.")

    def forward(self, layer: torch.nn.Module, query: torch.Tensor, key: torch.Tensor,
                value: torch.Tensor, kv_cache: torch.Tensor, attn_metadata,
                output: Optional[torch.Tensor] = None,
                output_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
        """query [num_tokens, num_heads, head_size], key / value [num_tokens, num_kv_heads,
        head_size], kv_cache [2, num_blocks, block_size, num_kv_heads, head_size] -> output
        [num_tokens, num_heads, head_size] (written in place up to num_actual_tokens)."""
        assert output is not None, "Output tensor must be provided."
        if attn_metadata is None:
            return output   # profiling run
        assert attn_metadata.use_cascade is False

        use_prefill_decode_attn = self.force_prefill_decode_attn
        num_actual_tokens = attn_metadata.num_actual_tokens
        if use_prefill_decode_attn:
            key_cache, value_cache = PagedAttention.split_kv_cache(kv_cache, self.num_kv_heads,
                                                                   self.head_size)
        else:
            key_cache, value_cache = kv_cache.unbind(0)

        if self.kv_sharing_target_layer_name is None:
            # store the new keys / values (skipped when the cache is shared with an earlier layer)
            if use_prefill_decode_attn:
                PagedAttention.write_to_paged_cache(key, value, key_cache, value_cache,
                                                    attn_metadata.slot_mapping, self.kv_cache_dtype,
                                                    layer._k_scale, layer._v_scale)
            else:
                torch.ops._C_cache_ops.reshape_and_cache_flash(
                    key, value, key_cache, value_cache, attn_metadata.slot_mapping,
                    self.kv_cache_dtype, layer._k_scale, layer._v_scale)

        if self.kv_cache_dtype.startswith("fp8"):
            key_cache = key_cache.view(self.fp8_dtype)
            value_cache = value_cache.view(self.fp8_dtype)
            assert layer._q_scale == 1.0, "A non 1.0 q_scale is not currently supported."
            # no Q quantisation on ROCm (:317-325)

        if self.alibi_slopes is not None and self.alibi_slopes.device != query.device:
            # (the reference builds the slopes under the worker's default device)
            self.alibi_slopes = self.alibi_slopes.to(query.device)
        cu_seqlens_q = attn_metadata.query_start_loc
        seqused_k = attn_metadata.seq_lens
        max_seqlen_q = attn_metadata.max_query_len
        max_seqlen_k = attn_metadata.max_seq_len
        block_table = attn_metadata.block_table

        if use_prefill_decode_attn:
            chunked_prefill_paged_decode(query=query[:num_actual_tokens],
                                         key=key[:num_actual_tokens],
                                         value=value[:num_actual_tokens],
                                         output=output[:num_actual_tokens],
                                         kv_cache_dtype=self.kv_cache_dtype,
                                         key_cache=key_cache, value_cache=value_cache,
                                         block_table=block_table, query_start_loc=cu_seqlens_q,
                                         seq_lens=seqused_k, max_seq_len=max_seqlen_k,
                                         max_query_len=max_seqlen_q, k_scale=layer._k_scale,
                                         v_scale=layer._v_scale, alibi_slopes=self.alibi_slopes,
                                         sliding_window=self.sliding_window[0],
                                         sm_scale=self.scale, output_scale=output_scale)
        else:
            descale_shape = (cu_seqlens_q.shape[0] - 1, key.shape[1])
            unified_attention(q=query[:num_actual_tokens], k=key_cache, v=value_cache,
                              out=output[:num_actual_tokens], cu_seqlens_q=cu_seqlens_q,
                              max_seqlen_q=max_seqlen_q, seqused_k=seqused_k,
                              max_seqlen_k=max_seqlen_k, softmax_scale=self.scale, causal=True,
                              alibi_slopes=self.alibi_slopes, window_size=self.sliding_window,
                              block_table=block_table, softcap=self.logits_soft_cap,
                              q_descale=None,  # Not supported
                              k_descale=layer._k_scale.expand(descale_shape),
                              v_descale=layer._v_scale.expand(descale_shape),
                              output_scale=output_scale)
        return output

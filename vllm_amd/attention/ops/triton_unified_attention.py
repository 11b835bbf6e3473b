"""The V1 engine's attention over the flash-layout KV cache: mirror of
vllm/attention/ops/triton_unified_attention.py:588-700 (`unified_attention`, same arguments)
over `vllm_amd_unified_attention` (HIP, prefill_attention.hip) instead of the Triton kernels."""
from __future__ import annotations

import torch

from ... import _lib
from ..._lib import check, lib
from ..._torch_ops import _dev, _ptr, _stream

_DT = {torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}
_FP8 = (torch.uint8, torch.float8_e4m3fn)


def _i32(t: torch.Tensor) -> torch.Tensor:
    return t if t.dtype == torch.int32 else t.to(torch.int32)


@torch.inference_mode()
def unified_attention(q, k, v, out, cu_seqlens_q, max_seqlen_q, seqused_k, max_seqlen_k,
                      softmax_scale, causal, window_size, block_table, softcap, q_descale,
                      k_descale, v_descale, alibi_slopes=None, output_scale=None) -> None:
    assert causal, "Only causal attention is supported"
    assert q_descale is None, "Q scales not supported"
    if q.dtype not in _DT:
        raise RuntimeError(f"unified_attention: unsupported query dtype {q.dtype}")
    fp8 = k.dtype in _FP8
    if fp8:
        assert v.dtype in _FP8 and k_descale is not None and v_descale is not None
    else:
        assert k.dtype == q.dtype and v.dtype == q.dtype
    block_size = v.shape[1]
    num_seqs = len(seqused_k)
    assert len(cu_seqlens_q) == num_seqs + 1
    for t in (q, out, k, v):
        assert t.stride(-1) == 1
    cu_seqlens_q, seqused_k, block_table = _i32(cu_seqlens_q), _i32(seqused_k), _i32(block_table)
    assert block_table.stride(1) == 1 and cu_seqlens_q.is_contiguous() and seqused_k.is_contiguous()
    if output_scale is not None:
        assert out.dtype in _FP8
    else:
        assert out.dtype == q.dtype
    # decode batches of less than two rounds of workgroups: scratch for the split over the
    # keys, allocated here as the reference does for its 3-D kernel (:659-683, NUM_SEGMENTS = 16)
    segm_out = segm_max = segm_sum = None
    nseg = 0
    num_kv_heads = k.shape[2]
    if max_seqlen_q == 1 and num_seqs * num_kv_heads * ((q.shape[1] // num_kv_heads + 15) // 16) < 1024:
        nseg = 16
        segm_out = torch.empty((q.shape[0], q.shape[1], nseg, q.shape[2]), dtype=torch.float32,
                               device=q.device)
        segm_max = torch.empty((q.shape[0], q.shape[1], nseg), dtype=torch.float32, device=q.device)
        segm_sum = torch.empty_like(segm_max)
    with _dev(q):
        check(lib.vllm_amd_unified_attention(
            _ptr(out), _ptr(q), _ptr(k), _ptr(v), _ptr(cu_seqlens_q), _ptr(seqused_k),
            _ptr(block_table), _ptr(k_descale), _ptr(v_descale), _ptr(alibi_slopes),
            _ptr(output_scale), num_seqs, q.shape[1], k.shape[2], q.shape[2], block_size,
            int(max_seqlen_q), block_table.stride(0), q.stride(0), q.stride(1), out.stride(0),
            out.stride(1), k.stride(0), k.stride(1), k.stride(2), v.stride(0), v.stride(1),
            v.stride(2), float(softmax_scale), 1 + int(window_size[0]), float(softcap or 0.0),
            q.shape[0], int(max_seqlen_k), _ptr(segm_out), _ptr(segm_max), _ptr(segm_sum), nseg,
            _DT[q.dtype], _lib.KV_FP8_E4M3 if fp8 else _lib.KV_AUTO, _stream(q)))

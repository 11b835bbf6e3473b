"""Thin PyTorch-ROCm custom-op layer over the C ABI.

Registers, with the reference's exact schema strings, the operators of
  torch.ops._C            (csrc/torch_bindings.cpp:45-69, 422-427, 575-593)
  torch.ops._C_cache_ops  (csrc/torch_bindings.cpp:676-708)
  torch.ops._rocm_C       (csrc/rocm/torch_bindings.cpp:34-56)
for the CUDA(=HIP) dispatch key.  Each implementation only unpacks tensors into
raw device pointers / shapes / strides and calls libvllm_amd.so on torch's
current stream: no allocation, no synchronisation, HIP-graph capturable.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import check, lib

_DT = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}
_FP8_DTYPES = (torch.uint8, torch.float8_e4m3fn)


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise RuntimeError(f"Unsupported data type: {t.dtype}") from None


def _kv(kv_cache_dtype: str) -> int:
    # csrc/quantization/fp8/amd/quant_utils.cuh:641-667
    if kv_cache_dtype == "auto":
        return _lib.KV_AUTO
    if kv_cache_dtype in ("fp8", "fp8_e4m3"):
        return _lib.KV_FP8_E4M3
    raise RuntimeError(f"Unsupported data type of kv cache: {kv_cache_dtype}")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _cuda(*ts: Optional[torch.Tensor]) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("vllm_amd ops need device tensors (cuda/HIP); got " + str(t.device))


class _dev:
    """at::cuda::OptionalCUDAGuard(device_of(t)) equivalent."""

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


# ---------------------------------------------------------------------------
def paged_attention_v1(out, query, key_cache, value_cache, num_kv_heads, scale, block_tables,
                       seq_lens, block_size, max_seq_len, alibi_slopes, kv_cache_dtype, k_scale,
                       v_scale, tp_rank, blocksparse_local_blocks, blocksparse_vert_stride,
                       blocksparse_block_size, blocksparse_head_sliding_step) -> None:
    _cuda(out, query, key_cache, value_cache, block_tables, seq_lens, alibi_slopes, k_scale, v_scale)
    num_seqs, num_heads, head_size = query.shape
    assert out.is_contiguous() and block_tables.dtype == torch.int32 and seq_lens.dtype == torch.int32
    _ensure_workspace(query)
    with _dev(query):
        check(lib.vllm_amd_paged_attention_v1(
            _ptr(out), _ptr(query), _ptr(key_cache), _ptr(value_cache), num_seqs, num_heads,
            head_size, num_kv_heads, scale, _ptr(block_tables), _ptr(seq_lens),
            block_tables.shape[1], block_size, max_seq_len, _ptr(alibi_slopes), _dt(query),
            _kv(kv_cache_dtype), _ptr(k_scale), _ptr(v_scale), query.stride(0),
            key_cache.stride(0), key_cache.stride(1), tp_rank, blocksparse_local_blocks,
            blocksparse_vert_stride, blocksparse_block_size, blocksparse_head_sliding_step,
            _stream(query)))


def paged_attention_v2(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache,
                       num_kv_heads, scale, block_tables, seq_lens, block_size, max_seq_len,
                       alibi_slopes, kv_cache_dtype, k_scale, v_scale, tp_rank,
                       blocksparse_local_blocks, blocksparse_vert_stride, blocksparse_block_size,
                       blocksparse_head_sliding_step) -> None:
    _cuda(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, block_tables,
          seq_lens, alibi_slopes, k_scale, v_scale)
    num_seqs, num_heads, head_size = query.shape
    assert out.is_contiguous() and tmp_out.is_contiguous() and exp_sums.is_contiguous()
    assert max_logits.is_contiguous() and exp_sums.dtype == torch.float32
    assert block_tables.dtype == torch.int32 and seq_lens.dtype == torch.int32
    with _dev(query):
        check(lib.vllm_amd_paged_attention_v2(
            _ptr(out), _ptr(exp_sums), _ptr(max_logits), _ptr(tmp_out), _ptr(query),
            _ptr(key_cache), _ptr(value_cache), num_seqs, num_heads, head_size, num_kv_heads,
            scale, _ptr(block_tables), _ptr(seq_lens), block_tables.shape[1], block_size,
            max_seq_len, exp_sums.shape[-1], _ptr(alibi_slopes), _dt(query), _kv(kv_cache_dtype),
            _ptr(k_scale), _ptr(v_scale), query.stride(0), key_cache.stride(0),
            key_cache.stride(1), tp_rank, blocksparse_local_blocks, blocksparse_vert_stride,
            blocksparse_block_size, blocksparse_head_sliding_step, _stream(query)))


def paged_attention_rocm(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache,
                         num_kv_heads, scale, block_tables, context_lens, query_start_loc,
                         block_size, max_context_len, alibi_slopes, kv_cache_dtype, k_scale,
                         v_scale, fp8_out_scale) -> None:
    _cuda(out, exp_sums, max_logits, tmp_out, query, key_cache, value_cache, block_tables,
          context_lens, query_start_loc, alibi_slopes, k_scale, v_scale, fp8_out_scale)
    num_seqs = block_tables.shape[0]
    _, num_heads, head_size = query.shape
    assert out.is_contiguous() and tmp_out.is_contiguous() and exp_sums.is_contiguous()
    assert block_tables.dtype == torch.int32 and context_lens.dtype == torch.int32
    if query_start_loc is not None:
        assert query_start_loc.dtype == torch.int32
    with _dev(query):
        check(lib.vllm_amd_paged_attention_rocm(
            _ptr(out), _ptr(exp_sums), _ptr(max_logits), _ptr(tmp_out), _ptr(query),
            _ptr(key_cache), _ptr(value_cache), num_seqs, num_heads, head_size, num_kv_heads,
            scale, _ptr(block_tables), _ptr(context_lens), _ptr(query_start_loc),
            block_tables.shape[1], block_size, max_context_len, exp_sums.shape[-1],
            _ptr(alibi_slopes), _dt(query), _kv(kv_cache_dtype), _ptr(k_scale), _ptr(v_scale),
            _ptr(fp8_out_scale), query.stride(0), key_cache.stride(0), key_cache.stride(1),
            _stream(query)))


def reshape_and_cache(key, value, key_cache, value_cache, slot_mapping, kv_cache_dtype, k_scale,
                      v_scale) -> None:
    _cuda(key, value, key_cache, value_cache, slot_mapping, k_scale, v_scale)
    num_tokens = slot_mapping.shape[0]
    num_heads, head_size = key.shape[1], key.shape[2]
    block_size, x = key_cache.shape[3], key_cache.shape[4]
    assert slot_mapping.dtype == torch.int64 and key.stride(2) == 1 and key.stride(1) == head_size
    with _dev(key):
        check(lib.vllm_amd_reshape_and_cache(
            _ptr(key), _ptr(value), _ptr(key_cache), _ptr(value_cache), _ptr(slot_mapping),
            num_tokens, num_heads, head_size, block_size, x, key.stride(0), value.stride(0),
            _dt(key), _kv(kv_cache_dtype), _ptr(k_scale), _ptr(v_scale), _stream(key)))


def reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping, kv_cache_dtype,
                            k_scale, v_scale) -> None:
    _cuda(key, value, key_cache, value_cache, slot_mapping, k_scale, v_scale)
    num_tokens = slot_mapping.shape[0]
    num_heads, head_size = key.shape[1], key.shape[2]
    block_size = key_cache.shape[1]
    assert slot_mapping.dtype == torch.int64 and key.stride(2) == 1 and key.stride(1) == head_size
    if key_cache.stride(0) != value_cache.stride(0):  # csrc/cache_kernels.cu:441
        raise RuntimeError("key_cache.stride(0) == value_cache.stride(0)")
    with _dev(key):
        check(lib.vllm_amd_reshape_and_cache_flash(
            _ptr(key), _ptr(value), _ptr(key_cache), _ptr(value_cache), _ptr(slot_mapping),
            num_tokens, num_heads, head_size, block_size, key.stride(0), value.stride(0),
            key_cache.stride(0), key_cache.stride(1), key_cache.stride(2), _dt(key),
            _kv(kv_cache_dtype), _ptr(k_scale), _ptr(v_scale), _stream(key)))


def convert_fp8(dst_cache, src_cache, scale, kv_cache_dtype) -> None:
    _cuda(dst_cache, src_cache)
    _kv(kv_cache_dtype)  # validates the string; "auto" is treated as e4m3 (gfx950 has one fp8 KV type)
    if src_cache.dtype in _DT and dst_cache.dtype in _FP8_DTYPES:
        to_fp8, dt = 1, _dt(src_cache)
    elif src_cache.dtype in _FP8_DTYPES and dst_cache.dtype in _DT:
        to_fp8, dt = 0, _dt(dst_cache)
    else:
        raise RuntimeError(f"Unsupported data type: {src_cache.dtype} -> {dst_cache.dtype}")
    assert src_cache.is_contiguous() and dst_cache.is_contiguous()
    with _dev(src_cache):
        check(lib.vllm_amd_convert_fp8(_ptr(dst_cache), _ptr(src_cache), src_cache.numel(),
                                       scale, dt, to_fp8, _stream(src_cache)))


def copy_blocks(key_caches, value_caches, block_mapping) -> None:
    # csrc/cache_kernels.cu:123-170; one launch per layer (no pointer table upload, no sync)
    if len(key_caches) != len(value_caches):
        raise RuntimeError("num_layers == value_caches.size()")
    if not key_caches:
        return
    _cuda(block_mapping, *key_caches, *value_caches)
    assert block_mapping.dtype == torch.int64 and block_mapping.is_contiguous()
    num_pairs = block_mapping.shape[0]
    for kc, vc in zip(key_caches, value_caches):
        block_bytes = kc.stride(0) * kc.element_size()
        with _dev(kc):
            check(lib.vllm_amd_copy_blocks(_ptr(kc), _ptr(vc), _ptr(block_mapping), num_pairs,
                                           block_bytes, _stream(kc)))


def swap_blocks(src, dst, block_mapping) -> None:
    # csrc/cache_kernels.cu:25-67: per-pair async copies on the current stream of the
    # device side; block_mapping must live on the CPU.
    if block_mapping.device.type != "cpu":
        raise RuntimeError("block_mapping must be on CPU")
    if src.is_cuda and dst.is_cuda and src.device != dst.device:
        raise RuntimeError("src and dst must be on the same GPU")
    if not (src.is_cuda or dst.is_cuda):
        raise RuntimeError("Invalid device combination")
    bm = block_mapping.to(torch.int64).contiguous()
    gpu = src if src.is_cuda else dst
    block_bytes = src.element_size() * src.stride(0)
    with _dev(gpu):
        check(lib.vllm_amd_swap_blocks(_ptr(src), _ptr(dst), block_bytes, _ptr(bm), bm.shape[0],
                                       _stream(gpu)))


# ---- neighbours of the path --------------------------------------------
def _fp8_out(result) -> None:
    if result.dtype not in _FP8_DTYPES:
        raise RuntimeError(f"Unsupported fp8 type: {result.dtype}")
    if not result.is_contiguous():
        raise RuntimeError("out.is_contiguous()")


def rms_norm_static_fp8_quant(result, input, weight, scale, epsilon) -> None:
    # csrc/layernorm_quant_kernels.cu:172-198
    _cuda(result, input, weight, scale)
    _fp8_out(result)
    hidden = input.shape[-1]
    nt = input.numel() // hidden if hidden else 0
    stride = input.stride(-2) if input.dim() >= 2 else hidden
    with _dev(input):
        check(lib.vllm_amd_rms_norm_static_fp8_quant(
            _ptr(result), _ptr(input), _ptr(weight), _ptr(scale), float(epsilon), nt, hidden,
            stride, _dt(input), _stream(input)))


def fused_add_rms_norm_static_fp8_quant(result, input, residual, weight, scale, epsilon) -> None:
    # csrc/layernorm_quant_kernels.cu:214-253
    _cuda(result, input, residual, weight, scale)
    _fp8_out(result)
    if not residual.is_contiguous():
        raise RuntimeError("residual.is_contiguous()")
    hidden = input.shape[-1]
    nt = input.numel() // hidden if hidden else 0
    stride = input.stride(-2) if input.dim() >= 2 else hidden
    with _dev(input):
        check(lib.vllm_amd_fused_add_rms_norm_static_fp8_quant(
            _ptr(result), _ptr(input), _ptr(residual), _ptr(weight), _ptr(scale), float(epsilon),
            nt, hidden, stride, _dt(input), _stream(input)))


def rms_norm_dynamic_per_token_quant(result, input, weight, scale, epsilon, scale_ub,
                                     residual) -> None:
    # csrc/quantization/fused_kernels/fused_layernorm_dynamic_per_token_quant.cu:132-155
    _cuda(result, input, weight, scale, scale_ub, residual)
    if result.dtype == torch.int8:
        raise RuntimeError("rms_norm_dynamic_per_token_quant: int8 output is outside the fp8 "
                           "path this library implements")
    _fp8_out(result)
    if not input.is_contiguous():
        raise RuntimeError("input.is_contiguous()")
    if scale.dtype != torch.float32:
        raise RuntimeError("scales.dtype() == torch::kFloat32")
    if residual is not None and not residual.is_contiguous():
        raise RuntimeError("residual.is_contiguous()")
    hidden = input.shape[-1]
    nt = input.numel() // hidden if hidden else 0
    with _dev(input):
        check(lib.vllm_amd_rms_norm_dynamic_per_token_quant(
            _ptr(result), _ptr(input), _ptr(weight), _ptr(scale), float(epsilon), _ptr(scale_ub),
            _ptr(residual), nt, hidden, _dt(input), _stream(input)))


def silu_and_mul_quant(result, input, scale) -> None:
    # csrc/quantization/activation_kernels.cu:112-121
    _cuda(result, input, scale)
    _fp8_out(result)
    if input.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("input.dtype() == torch::kFloat16 || input.dtype() == torch::kBFloat16")
    if input.shape[-1] % 2 != 0:
        raise RuntimeError("input.size(-1) % 2 == 0")
    if not input.is_contiguous():
        raise RuntimeError("input.is_contiguous()")
    d = input.shape[-1] // 2
    nt = input.numel() // input.shape[-1] if d else 0
    with _dev(input):
        check(lib.vllm_amd_silu_and_mul_quant(_ptr(result), _ptr(input), _ptr(scale), nt, d,
                                              _dt(input), _stream(input)))


def compute_slot_mapping(slot_mapping, block_table, req_indices, positions, block_size) -> None:
    """Device-side BlockTable.compute_slot_mapping (vllm/v1/worker/block_table.py:84-99).
    Not a reference op: the reference does this with numpy on the host."""
    _cuda(slot_mapping, block_table, req_indices, positions)
    assert slot_mapping.dtype == torch.int64 and positions.dtype == torch.int64
    assert block_table.dtype == torch.int32 and req_indices.dtype == torch.int32
    assert block_table.is_contiguous() and req_indices.is_contiguous() and positions.is_contiguous()
    n = req_indices.shape[0]
    assert slot_mapping.is_contiguous() and slot_mapping.shape[0] >= n and positions.shape[0] >= n
    with _dev(block_table):
        check(lib.vllm_amd_compute_slot_mapping(
            _ptr(slot_mapping), _ptr(block_table), _ptr(req_indices), _ptr(positions), n,
            block_table.shape[1], block_size, _stream(block_table)))


def _quant_common(result, input):
    _cuda(result, input)
    if not input.is_contiguous():
        raise RuntimeError("input.is_contiguous()")
    if not result.is_contiguous():
        raise RuntimeError("out.is_contiguous()")
    if result.dtype not in _FP8_DTYPES:
        raise RuntimeError(f"Unsupported fp8 type: {result.dtype}")
    hidden = input.shape[-1]
    return (input.numel() // hidden if hidden else 0), hidden


def static_scaled_fp8_quant(result, input, scale) -> None:
    nt, hidden = _quant_common(result, input)
    with _dev(input):
        check(lib.vllm_amd_static_scaled_fp8_quant(_ptr(result), _ptr(input), _ptr(scale), nt,
                                                   hidden, _dt(input), _stream(input)))


def dynamic_scaled_fp8_quant(result, input, scale) -> None:
    nt, hidden = _quant_common(result, input)
    with _dev(input):
        check(lib.vllm_amd_dynamic_scaled_fp8_quant(_ptr(result), _ptr(input), _ptr(scale), nt,
                                                    hidden, _dt(input), _stream(input)))


def dynamic_per_token_scaled_fp8_quant(result, input, scale, scale_ub) -> None:
    nt, hidden = _quant_common(result, input)
    with _dev(input):
        check(lib.vllm_amd_dynamic_per_token_scaled_fp8_quant(
            _ptr(result), _ptr(input), _ptr(scale), _ptr(scale_ub), nt, hidden, _dt(input),
            _stream(input)))


# The library workspace (vllm_amd_set_workspace): the C ABI allocates nothing, so the op layer
# owns one buffer per device (4 stream slots x 16 MiB + ticket counters) and registers it once,
# outside stream capture, before that device's first scaled_mm / paged_attention_v1.  Without it
# the GEMM never splits over K and paged_attention_v1 runs one workgroup per (sequence, kv head)
# (same results up to fp32 summation order).
_WS_BYTES = (64 << 20) + (512 << 10)
_ws = {}


def _ensure_workspace(t: torch.Tensor) -> None:
    idx = t.device.index if t.device.index is not None else torch.cuda.current_device()
    if idx in _ws or torch.cuda.is_current_stream_capturing():
        return
    with _dev(t):
        buf = torch.empty(_WS_BYTES, dtype=torch.uint8, device=t.device)
        check(lib.vllm_amd_set_workspace(_ptr(buf), _WS_BYTES, _stream(t)))
        # every stream's first launch must see the zeroed counters
        torch.cuda.current_stream(t.device).synchronize()
    _ws[idx] = buf


def cutlass_scaled_mm(out, a, b, a_scales, b_scales, bias) -> None:
    _cuda(out, a, b, a_scales, b_scales, bias)
    _ensure_workspace(a)
    # csrc/quantization/cutlass_w8a8/scaled_mm_entry.cu checks, restated
    if not (a.dim() == 2 and b.dim() == 2 and out.dim() == 2):
        raise RuntimeError("a.dim() == 2 && b.dim() == 2 && c.dim() == 2")
    M, K = a.shape
    N = b.shape[1]
    if not (out.shape[0] == M and b.shape[0] == K and out.shape[1] == N):
        raise RuntimeError("c.size(0) == a.size(0) && a.size(1) == b.size(0) && b.size(1) == c.size(1)")
    if not (a.stride(1) == 1 and out.stride(1) == 1):
        raise RuntimeError("a.stride(1) == 1 && c.stride(1) == 1")  # row major
    if b.stride(0) != 1:
        raise RuntimeError("b.stride(0) == 1")  # column major
    if a.dtype not in _FP8_DTYPES or b.dtype not in _FP8_DTYPES:
        raise RuntimeError("scaled_mm: only fp8 e4m3 x fp8 e4m3 is implemented on gfx950")
    if a_scales.dtype != torch.float32 or b_scales.dtype != torch.float32:
        raise RuntimeError("scales must be float32")
    if not (a_scales.is_contiguous() and b_scales.is_contiguous()):
        raise RuntimeError("a_scales.is_contiguous() && b_scales.is_contiguous()")
    if bias is not None and not (bias.numel() == N and bias.is_contiguous() and bias.dim() == 1
                                 and bias.dtype == out.dtype):
        raise RuntimeError("bias.numel() == b.size(1) && bias.is_contiguous() && bias.dim() == 1")
    with _dev(a):
        check(lib.vllm_amd_scaled_mm(
            _ptr(out), _ptr(a), _ptr(b), _ptr(a_scales), _ptr(b_scales), _ptr(bias), M, N, K,
            a.stride(0), b.stride(1), out.stride(0), a_scales.numel(), b_scales.numel(),
            _dt(out), _stream(a)))


def wvSplitKQ(in_a, in_b, out_c, scale_a, scale_b, CuCount) -> None:
    _cuda(in_a, in_b, out_c, scale_a, scale_b)
    M_in, K_in = in_a.shape
    N_in = in_b.shape[0]
    if not (in_a.is_contiguous() and in_b.is_contiguous() and out_c.is_contiguous()):
        raise RuntimeError("wvSplitKQ: contiguous tensors required")
    if in_b.shape[1] != K_in:
        raise RuntimeError("in_a.size(1) == in_b.size(1)")
    with _dev(in_a):
        check(lib.vllm_amd_wvSplitKQ(_ptr(in_a), _ptr(in_b), _ptr(out_c), _ptr(scale_a),
                                     _ptr(scale_b), M_in, K_in, N_in, _dt(out_c), CuCount,
                                     _stream(in_a)))


# ---------------------------------------------------------------------------
# registration (schemas verbatim from the reference)
# ---------------------------------------------------------------------------
_PA_TAIL = (" int num_kv_heads, float scale, Tensor block_tables, Tensor seq_lens, int block_size,"
            " int max_seq_len, Tensor? alibi_slopes, str kv_cache_dtype, Tensor k_scale,"
            " Tensor v_scale, int tp_rank, int blocksparse_local_blocks,"
            " int blocksparse_vert_stride, int blocksparse_block_size,"
            " int blocksparse_head_sliding_step) -> ()")

_SCHEMAS = {
    "_C": [
        ("paged_attention_v1",
         "paged_attention_v1(Tensor! out, Tensor query, Tensor key_cache, Tensor value_cache,"
         + _PA_TAIL, paged_attention_v1),
        ("paged_attention_v2",
         "paged_attention_v2(Tensor! out, Tensor! exp_sums, Tensor! max_logits, Tensor! tmp_out,"
         " Tensor query, Tensor key_cache, Tensor value_cache," + _PA_TAIL, paged_attention_v2),
        ("static_scaled_fp8_quant",
         "static_scaled_fp8_quant(Tensor! result, Tensor input, Tensor scale) -> ()",
         static_scaled_fp8_quant),
        ("dynamic_scaled_fp8_quant",
         "dynamic_scaled_fp8_quant(Tensor! result, Tensor input, Tensor! scale) -> ()",
         dynamic_scaled_fp8_quant),
        ("dynamic_per_token_scaled_fp8_quant",
         "dynamic_per_token_scaled_fp8_quant(Tensor! result, Tensor input, Tensor! scale,"
         " Tensor? scale_ub) -> ()", dynamic_per_token_scaled_fp8_quant),
        ("cutlass_scaled_mm",
         "cutlass_scaled_mm(Tensor! out, Tensor a, Tensor b, Tensor a_scales, Tensor b_scales,"
         " Tensor? bias) -> ()", cutlass_scaled_mm),
        # csrc/torch_bindings.cpp:117-120, 188-211
        ("silu_and_mul_quant",
         "silu_and_mul_quant(Tensor! result, Tensor input, Tensor scale) -> ()",
         silu_and_mul_quant),
        ("rms_norm_static_fp8_quant",
         "rms_norm_static_fp8_quant(Tensor! result, Tensor input, Tensor weight, Tensor scale,"
         " float epsilon) -> ()", rms_norm_static_fp8_quant),
        ("fused_add_rms_norm_static_fp8_quant",
         "fused_add_rms_norm_static_fp8_quant(Tensor! result, Tensor input, Tensor! residual,"
         " Tensor weight, Tensor scale, float epsilon) -> ()",
         fused_add_rms_norm_static_fp8_quant),
        ("rms_norm_dynamic_per_token_quant",
         "rms_norm_dynamic_per_token_quant(Tensor! result, Tensor input, Tensor weight,"
         " Tensor! scale, float epsilon, Tensor? scale_ub, Tensor!? residual) -> ()",
         rms_norm_dynamic_per_token_quant),
    ],
    "_C_cache_ops": [
        ("reshape_and_cache",
         "reshape_and_cache(Tensor key, Tensor value, Tensor! key_cache, Tensor! value_cache,"
         " Tensor slot_mapping, str kv_cache_dtype, Tensor k_scale, Tensor v_scale) -> ()",
         reshape_and_cache),
        ("reshape_and_cache_flash",
         "reshape_and_cache_flash(Tensor key, Tensor value, Tensor! key_cache,"
         " Tensor! value_cache, Tensor slot_mapping, str kv_cache_dtype, Tensor k_scale,"
         " Tensor v_scale) -> ()", reshape_and_cache_flash),
        ("convert_fp8",
         "convert_fp8(Tensor! dst_cache, Tensor src_cache, float scale, str kv_cache_dtype) -> ()",
         convert_fp8),
        ("copy_blocks",
         "copy_blocks(Tensor(a!)[] key_caches, Tensor[](b!) value_caches,"
         " Tensor block_mapping) -> ()", copy_blocks),
        ("swap_blocks",
         "swap_blocks(Tensor src, Tensor! dst, Tensor block_mapping) -> ()", swap_blocks),
    ],
    "_rocm_C": [
        ("paged_attention",
         "paged_attention(Tensor! out, Tensor exp_sums, Tensor max_logits, Tensor tmp_out,"
         " Tensor query, Tensor key_cache, Tensor value_cache, int num_kv_heads, float scale,"
         " Tensor block_tables, Tensor context_lens, Tensor? query_start_loc, int block_size,"
         " int max_context_len, Tensor? alibi_slopes, str kv_cache_dtype, Tensor k_scale,"
         " Tensor v_scale, Tensor? fp8_out_scale) -> ()", paged_attention_rocm),
        ("wvSplitKQ",
         "wvSplitKQ(Tensor in_a, Tensor in_b, Tensor! out_c, Tensor scale_a, Tensor scale_b,"
         " int CuCount) -> ()", wvSplitKQ),
    ],
}

_libs = []  # keep the Library objects alive


def _register() -> None:
    for ns, ops in _SCHEMAS.items():
        frag = torch.library.Library(ns, "FRAGMENT")
        _libs.append(frag)
        for name, schema, fn in ops:
            try:
                frag.define(schema)
            except RuntimeError:
                pass  # already defined (e.g. real vLLM imported first): override the impl only
            key = "CompositeExplicitAutograd" if name == "swap_blocks" else "CUDA"
            try:
                frag.impl(name, fn, key)
            except RuntimeError:
                over = torch.library.Library(ns, "IMPL", key)
                _libs.append(over)
                over.impl(name, fn)


_register()

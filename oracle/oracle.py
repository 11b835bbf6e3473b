"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of oracle/liboracle.so.

Mirrors the argument lists of the reference's `vllm/_custom_ops.py` wrappers
(file:line cited per function) but runs the plain-C restatement in
`oracle/oracle.c` on CPU tensors.  Only tests/, `__graft_entry__.smoke()` and
`bench.py`'s cpu_baseline leg import this module; nothing under `vllm_amd/`
does.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "librefcpu.so")

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"{_LIB_PATH} missing: run `make -C oracle` (or __graft_entry__.build())")
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_fp8_e4m3_to_f32.restype = C.c_float
        _lib.oracle_fp8_e4m3_to_f32.argtypes = [C.c_uint8]
        _lib.oracle_f32_to_fp8_e4m3.restype = C.c_uint8
        _lib.oracle_f32_to_fp8_e4m3.argtypes = [C.c_float]
    return _lib


_ref_loaded = False


def load_ref() -> bool:
    """Load oracle/_ref/librefcpu.so (the reference's own CPU kernels compiled
    by oracle/Makefile) into torch.ops.refcpu.  Returns False if absent."""
    global _ref_loaded
    if _ref_loaded:
        return True
    if not os.path.exists(_REF_PATH):
        return False
    torch.ops.load_library(_REF_PATH)
    _ref_loaded = True
    return True


DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def _kv(kv_cache_dtype: str) -> int:
    if kv_cache_dtype == "auto":
        return 0
    if kv_cache_dtype in ("fp8", "fp8_e4m3"):
        return 1
    raise RuntimeError(f"Unsupported data type of kv cache: {kv_cache_dtype}")


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    assert t.device.type == "cpu"
    return C.c_void_p(t.data_ptr())


def _f(x) -> C.c_float:
    return C.c_float(float(x))


def _i64(x) -> C.c_int64:
    return C.c_int64(int(x))


# ----------------------------------------------------------------------------
def fp8_decode_table() -> torch.Tensor:
    l = lib()
    return torch.tensor([l.oracle_fp8_e4m3_to_f32(i) for i in range(256)],
                        dtype=torch.float32)


def fp8_encode(x: torch.Tensor) -> torch.Tensor:
    """float32 tensor -> uint8 e4m3fn codes (RNE, saturating)."""
    x = x.float().contiguous()
    out = torch.empty(x.shape, dtype=torch.uint8)
    l = lib()
    l.oracle_convert_fp8(_p(out), _p(x), _i64(x.numel()), _f(1.0), 0, 1)
    return out


# vllm/_custom_ops.py:1617-1629
def reshape_and_cache(key, value, key_cache, value_cache, slot_mapping,
                      kv_cache_dtype, k_scale, v_scale) -> None:
    num_tokens = slot_mapping.shape[0]
    _, num_heads, head_size = key.shape
    block_size, x = key_cache.shape[3], key_cache.shape[4]
    lib().oracle_reshape_and_cache(
        _p(key), _p(value), _p(key_cache), _p(value_cache), _p(slot_mapping),
        num_tokens, num_heads, head_size, block_size, x, _i64(key.stride(0)),
        _i64(value.stride(0)), DT[key.dtype], _kv(kv_cache_dtype),
        _f(k_scale), _f(v_scale))


# vllm/_custom_ops.py:1632-1645
def reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping,
                            kv_cache_dtype, k_scale, v_scale) -> None:
    num_tokens = slot_mapping.shape[0]
    _, num_heads, head_size = key.shape
    block_size = key_cache.shape[1]
    lib().oracle_reshape_and_cache_flash(
        _p(key), _p(value), _p(key_cache), _p(value_cache), _p(slot_mapping),
        num_tokens, num_heads, head_size, block_size, _i64(key.stride(0)),
        _i64(value.stride(0)), _i64(key_cache.stride(0)),
        _i64(key_cache.stride(1)), _i64(key_cache.stride(2)), DT[key.dtype],
        _kv(kv_cache_dtype), _f(k_scale), _f(v_scale))


# vllm/_custom_ops.py:1677-1681
def convert_fp8(output, input, scale: float = 1.0, kv_dtype: str = "fp8") -> None:
    assert input.is_contiguous() and output.is_contiguous()
    if output.dtype == torch.uint8:
        lib().oracle_convert_fp8(_p(output), _p(input), _i64(input.numel()),
                                 _f(scale), DT[input.dtype], 1)
    else:
        lib().oracle_convert_fp8(_p(output), _p(input), _i64(input.numel()),
                                 _f(scale), DT[output.dtype], 0)


# vllm/_custom_ops.py:1244-1304
def scaled_fp8_quant(input, scale=None, num_token_padding=None, scale_ub=None,
                     use_per_token_if_dynamic=False):
    assert input.ndim == 2
    input = input.contiguous()
    shape = input.shape
    if num_token_padding:
        shape = (max(num_token_padding, shape[0]), shape[1])
    out = torch.zeros(shape, dtype=torch.uint8)
    n = input.numel()
    if scale is None:
        if use_per_token_if_dynamic:
            scale = torch.zeros((shape[0], 1), dtype=torch.float32)
            lib().oracle_dynamic_per_token_scaled_fp8_quant(
                _p(out), _p(input), _p(scale), _p(scale_ub),
                _i64(input.shape[0]), input.shape[1], DT[input.dtype])
        else:
            scale = torch.zeros(1, dtype=torch.float32)
            lib().oracle_dynamic_scaled_fp8_quant(_p(out), _p(input), _p(scale),
                                                  _i64(n), DT[input.dtype])
    else:
        assert scale.numel() == 1
        lib().oracle_static_scaled_fp8_quant(_p(out), _p(input),
                                             _f(scale.item()), _i64(n),
                                             DT[input.dtype])
    return out, scale


# vllm/_custom_ops.py:697-742  (a: [M,K] uint8 codes; b: [K,N] with stride(0)==1)
def scaled_mm(a, b, scale_a, scale_b, out_dtype, bias=None):
    M, K = a.shape
    K2, N = b.shape
    assert K == K2 and b.stride(0) == 1 and a.stride(1) == 1
    out = torch.empty((M, N), dtype=out_dtype)
    scale_a = scale_a.float().contiguous()
    scale_b = scale_b.float().contiguous()
    lib().oracle_scaled_mm(_p(out), _p(a), _p(b), _p(scale_a), _p(scale_b),
                           _p(bias), M, N, K, _i64(a.stride(0)),
                           _i64(b.stride(1)), scale_a.numel(), scale_b.numel(),
                           DT[out_dtype])
    return out


def _attn_common(query, key_cache, value_cache, num_kv_heads, scale,
                 block_tables, seq_lens, block_size, alibi_slopes,
                 kv_cache_dtype, k_scale, v_scale):
    num_seqs, num_heads, head_size = query.shape
    return [
        _p(query), _p(key_cache), _p(value_cache), num_seqs, num_heads,
        head_size, num_kv_heads, _f(scale), _p(block_tables), _p(seq_lens),
        block_tables.shape[1], block_size, _p(alibi_slopes), DT[query.dtype],
        _kv(kv_cache_dtype), _f(k_scale), _f(v_scale), _i64(query.stride(0)),
        _i64(key_cache.stride(0)), _i64(key_cache.stride(1))
    ]


# vllm/_custom_ops.py:39-65
def paged_attention_v1(out, query, key_cache, value_cache, num_kv_heads, scale,
                       block_tables, seq_lens, block_size, max_seq_len,
                       alibi_slopes, kv_cache_dtype, k_scale, v_scale,
                       tp_rank=0, blocksparse_local_blocks=0,
                       blocksparse_vert_stride=0, blocksparse_block_size=64,
                       blocksparse_head_sliding_step=0) -> None:
    args = _attn_common(query, key_cache, value_cache, num_kv_heads, scale,
                        block_tables, seq_lens, block_size, alibi_slopes,
                        kv_cache_dtype, k_scale, v_scale)
    lib().oracle_paged_attention_v1(
        _p(out), *args, tp_rank, blocksparse_local_blocks,
        blocksparse_vert_stride, blocksparse_block_size,
        blocksparse_head_sliding_step)


# vllm/_custom_ops.py:68-97
def paged_attention_v2(out, exp_sum, max_logits, tmp_out, query, key_cache,
                       value_cache, num_kv_heads, scale, block_tables, seq_lens,
                       block_size, max_seq_len, alibi_slopes, kv_cache_dtype,
                       k_scale, v_scale, tp_rank=0, blocksparse_local_blocks=0,
                       blocksparse_vert_stride=0, blocksparse_block_size=64,
                       blocksparse_head_sliding_step=0,
                       partition_size=512) -> None:
    args = _attn_common(query, key_cache, value_cache, num_kv_heads, scale,
                        block_tables, seq_lens, block_size, alibi_slopes,
                        kv_cache_dtype, k_scale, v_scale)
    lib().oracle_paged_attention_v2(
        _p(out), _p(exp_sum), _p(max_logits), _p(tmp_out), *args, tp_rank,
        blocksparse_local_blocks, blocksparse_vert_stride,
        blocksparse_block_size, blocksparse_head_sliding_step, partition_size,
        exp_sum.shape[-1])


# vllm/_custom_ops.py:100-126
def paged_attention_rocm(out, exp_sum, max_logits, tmp_out, query, key_cache,
                         value_cache, num_kv_heads, scale, block_tables,
                         seq_lens, query_start_loc, block_size, max_seq_len,
                         alibi_slopes, kv_cache_dtype, k_scale, v_scale,
                         fp8_out_scale=None) -> None:
    num_seqs = block_tables.shape[0]
    _, num_heads, head_size = query.shape
    lib().oracle_paged_attention_rocm(
        _p(out), _p(exp_sum), _p(max_logits), _p(tmp_out), _p(query),
        _p(key_cache), _p(value_cache), num_seqs, num_heads, head_size,
        num_kv_heads, _f(scale), _p(block_tables), _p(seq_lens),
        _p(query_start_loc), block_tables.shape[1], block_size,
        _p(alibi_slopes), DT[query.dtype], _kv(kv_cache_dtype), _f(k_scale),
        _f(v_scale), _i64(query.stride(0)), _i64(key_cache.stride(0)),
        _i64(key_cache.stride(1)), exp_sum.shape[-1], _p(fp8_out_scale))


def paged_attention_f64(query, key_cache, value_cache, num_kv_heads, scale,
                        block_tables, seq_lens, block_size, alibi_slopes,
                        kv_cache_dtype, k_scale, v_scale) -> torch.Tensor:
    """fp64 ground truth (no intermediate rounding); bounds both sides' error."""
    out = torch.empty(query.shape, dtype=torch.float64)
    args = _attn_common(query, key_cache, value_cache, num_kv_heads, scale,
                        block_tables, seq_lens, block_size, alibi_slopes,
                        kv_cache_dtype, k_scale, v_scale)
    lib().oracle_paged_attention_f64(_p(out), *args)
    return out


# ---- neighbours of the path ---------------------------------------------
# vllm/_custom_ops.py:283-286 (torch.ops._C.rms_norm_static_fp8_quant)
def rms_norm_static_fp8_quant(input, weight, scale: float, epsilon: float) -> torch.Tensor:
    hidden = input.shape[-1]
    nt = input.numel() // hidden
    out = torch.empty(input.shape, dtype=torch.uint8)
    stride = input.stride(-2) if input.dim() >= 2 else hidden
    lib().oracle_rms_norm_static_fp8_quant(_p(out), _p(input), _i64(stride), _p(weight),
                                           _f(scale), _f(epsilon), _i64(nt), hidden,
                                           DT[input.dtype])
    return out


# vllm/_custom_ops.py:289-293; residual is updated in place
def fused_add_rms_norm_static_fp8_quant(input, residual, weight, scale: float,
                                        epsilon: float) -> torch.Tensor:
    hidden = input.shape[-1]
    nt = input.numel() // hidden
    assert residual.is_contiguous()
    out = torch.empty(input.shape, dtype=torch.uint8)
    stride = input.stride(-2) if input.dim() >= 2 else hidden
    lib().oracle_fused_add_rms_norm_static_fp8_quant(
        _p(out), _p(input), _i64(stride), _p(residual), _p(weight), _f(scale), _f(epsilon),
        _i64(nt), hidden, DT[input.dtype])
    return out


# vllm/_custom_ops.py:368-384 (fp8 output); residual is updated in place
def rms_norm_dynamic_per_token_quant(input, weight, epsilon: float, scale_ub=None,
                                     residual=None):
    assert input.is_contiguous()
    hidden = input.shape[-1]
    nt = input.numel() // hidden
    out = torch.empty(input.shape, dtype=torch.uint8)
    scales = torch.empty((nt, 1), dtype=torch.float32)
    ub = None if scale_ub is None else scale_ub.float().reshape(1).contiguous()
    lib().oracle_rms_norm_dynamic_per_token_quant(
        _p(out), _p(input), _p(weight), _p(scales), _f(epsilon), _p(ub), _p(residual), _i64(nt),
        hidden, DT[input.dtype])
    return out, scales


# torch.ops._C.silu_and_mul_quant (csrc/quantization/activation_kernels.cu:112-121)
def silu_and_mul_quant(input, scale: float) -> torch.Tensor:
    assert input.is_contiguous()
    d = input.shape[-1] // 2
    nt = input.numel() // input.shape[-1]
    out = torch.empty(input.shape[:-1] + (d,), dtype=torch.uint8)
    lib().oracle_silu_and_mul_quant(_p(out), _p(input), _f(scale), _i64(nt), d, DT[input.dtype])
    return out


# vllm/v1/worker/block_table.py:84-99
def compute_slot_mapping(block_table, req_indices, positions, block_size: int) -> torch.Tensor:
    n = req_indices.shape[0]
    out = torch.empty(n, dtype=torch.int64)
    lib().oracle_compute_slot_mapping(_p(out), _p(block_table.contiguous()),
                                      _p(req_indices.contiguous()), _p(positions.contiguous()),
                                      _i64(n), block_table.shape[1], block_size)
    return out


# vllm/attention/ops/prefix_prefill.py:726-913
def context_attention_fwd(q, k, v, o, kv_cache_dtype, k_cache, v_cache, b_loc, b_start_loc,
                          b_seq_len, max_seq_len, max_input_len, k_scale, v_scale,
                          alibi_slopes=None, sliding_window=None, sm_scale=None,
                          skip_decode=False, fp8_out_scale=None) -> None:
    if sm_scale is None:
        sm_scale = 1.0 / (q.shape[-1] ** 0.5)
    if sliding_window is None or sliding_window <= 0:
        sliding_window = 0
    b_loc = b_loc.to(torch.int32).contiguous()
    b_start_loc = b_start_loc.to(torch.int32).contiguous()
    b_seq_len = b_seq_len.to(torch.int32).contiguous()
    slopes = None if alibi_slopes is None else alibi_slopes.float().contiguous()
    fo = None if fp8_out_scale is None else fp8_out_scale.float().reshape(1).contiguous()
    lib().oracle_context_attention_fwd(
        _p(o), _p(q), _p(k), _p(v), _p(k_cache), _p(v_cache), _p(b_loc), _p(b_start_loc),
        _p(b_seq_len), _f(k_scale), _f(v_scale), _p(slopes), _p(fo), b_seq_len.shape[0],
        q.shape[1], k.shape[1], q.shape[2], v_cache.shape[3], k_cache.shape[4],
        _i64(b_loc.stride(0)), _i64(q.stride(0)), _i64(q.stride(1)), _i64(k.stride(0)),
        _i64(k.stride(1)), _i64(v.stride(0)), _i64(v.stride(1)), _i64(o.stride(0)),
        _i64(o.stride(1)), _i64(k_cache.stride(0)), _i64(k_cache.stride(1)),
        _i64(v_cache.stride(0)), _i64(v_cache.stride(1)), _f(sm_scale), int(sliding_window),
        int(bool(skip_decode)), DT[q.dtype], _kv(kv_cache_dtype))


# vllm/attention/ops/triton_unified_attention.py:52-272 (kernel_unified_attention_2d), 588-700.
# Plain torch fp32 on the CPU, one (sequence, head) at a time, in the kernel's order of
# operations: K / V dequantised and rounded to the query dtype (:172-192), S = scale * q.k
# (:205), softcap (:207-208, 25-30), causal / padding mask (:198, 210-211), sliding window
# (:213-215), ALiBi bias slope * (key position - context_len) (:217-218), softmax, P rounded
# to V's dtype (:246), acc / L (:250), fp8 output (:253-256).  Same maths as the reference
# test's ref_paged_attn (tests/kernels/attention/test_triton_unified_attention.py:25-77).
# PARITY UNPINNED against the Triton kernel itself (Triton is not runnable here): pinned
# instead against the context_attention_fwd restatement above, which computes the same
# function when the new tokens' K / V are both passed densely and present in the cache
# (tests/test_oracle.py::test_unified_oracle_vs_prefix_oracle).
def unified_attention(q, k, v, out, cu_seqlens_q, max_seqlen_q, seqused_k, max_seqlen_k,
                      softmax_scale, causal, window_size, block_table, softcap, q_descale,
                      k_descale, v_descale, alibi_slopes=None, output_scale=None) -> None:
    assert causal and q_descale is None
    qd = q.dtype
    block_size = v.shape[1]
    nkv = k.shape[2]
    H = q.shape[1]
    g = H // nkv
    win = 1 + int(window_size[0])
    fp8 = k.dtype in (torch.uint8, torch.float8_e4m3fn)
    for i in range(len(seqused_k)):
        q0, q1 = int(cu_seqlens_q[i]), int(cu_seqlens_q[i + 1])
        qlen, kvlen = q1 - q0, int(seqused_k[i])
        if qlen == 0:
            continue
        ctx = kvlen - qlen
        nblk = (kvlen + block_size - 1) // block_size
        idx = block_table[i, :nblk].long()
        kk = k[idx].reshape(-1, nkv, k.shape[3])[:kvlen]
        vv = v[idx].reshape(-1, nkv, v.shape[3])[:kvlen]
        if fp8:
            kk = (kk.view(torch.float8_e4m3fn).float() * float(k_descale.flatten()[0])).to(qd)
            vv = (vv.view(torch.float8_e4m3fn).float() * float(v_descale.flatten()[0])).to(qd)
        kk, vv = kk.float(), vv.float()
        qpos = ctx + torch.arange(qlen)[:, None]
        kpos = torch.arange(kvlen)[None, :]
        keep = kpos <= qpos
        if win > 0:
            keep &= (qpos - kpos) < win
        for h in range(H):
            S = softmax_scale * (q[q0:q1, h].float() @ kk[:, h // g].t())
            if softcap and softcap > 0:
                S = softcap * torch.tanh(S / softcap)
            S = torch.where(keep, S, torch.tensor(float("-inf")))
            if alibi_slopes is not None:
                S = S + float(alibi_slopes[h]) * (kpos - ctx)
            m = S.max(dim=1, keepdim=True).values
            m = torch.where(m > float("-inf"), m, torch.zeros_like(m))
            P = torch.exp(S - m)
            L = P.sum(dim=1, keepdim=True)
            L = torch.where(L > 0, L, torch.ones_like(L))
            acc = (P.to(qd).float() @ vv[:, h // g]) / L
            if output_scale is not None:
                acc = (acc / float(output_scale.flatten()[0])).clamp(-448.0, 448.0)
                out[q0:q1, h] = acc.to(torch.float8_e4m3fn).view(out.dtype)
            else:
                out[q0:q1, h] = acc.to(out.dtype)

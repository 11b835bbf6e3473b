"""CPU tests (no GPU): the C-ABI library loads, exports every symbol declared in
include/vllm_amd.h, the torch ops carry the reference's schema strings, and argument
validation returns the reference's error text BEFORE anything is launched."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vllm_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vllm_amd_[A-Za-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from vllm_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 22
    for s in syms:
        assert hasattr(_lib.lib, s), f"{s} declared in include/vllm_amd.h but not exported"
    # and the ctypes signatures cover every entry point that takes arguments
    for s in syms:
        if s not in ("vllm_amd_last_error", "vllm_amd_version", "vllm_amd_ar_meta_size"):
            assert s in _lib.SIGNATURES, s
    assert _lib.lib.vllm_amd_version() >= 100


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import importlib.util
    src = os.path.join(ROOT, "vllm_amd", "_lib.py")
    spec = importlib.util.spec_from_file_location("lib_probe", src)
    mod = importlib.util.module_from_spec(spec)
    monkeypatch.setattr(os.path, "exists", lambda p: False if p.endswith("libvllm_amd.so") else True)
    with pytest.raises(ImportError, match="no CPU or\\s+eager fallback|not built"):
        spec.loader.exec_module(mod)


SCHEMAS = {
    # SURVEY.md 8b, verbatim from csrc/torch_bindings.cpp / csrc/rocm/torch_bindings.cpp
    "_C::paged_attention_v1": "(Tensor! out, Tensor query, Tensor key_cache, Tensor value_cache, int num_kv_heads, float scale, Tensor block_tables, Tensor seq_lens, int block_size, int max_seq_len, Tensor? alibi_slopes, str kv_cache_dtype, Tensor k_scale, Tensor v_scale, int tp_rank, int blocksparse_local_blocks, int blocksparse_vert_stride, int blocksparse_block_size, int blocksparse_head_sliding_step) -> ()",
    "_C::static_scaled_fp8_quant": "(Tensor! result, Tensor input, Tensor scale) -> ()",
    "_C::dynamic_per_token_scaled_fp8_quant": "(Tensor! result, Tensor input, Tensor! scale, Tensor? scale_ub) -> ()",
    "_C::cutlass_scaled_mm": "(Tensor! out, Tensor a, Tensor b, Tensor a_scales, Tensor b_scales, Tensor? bias) -> ()",
    "_C_cache_ops::reshape_and_cache": "(Tensor key, Tensor value, Tensor! key_cache, Tensor! value_cache, Tensor slot_mapping, str kv_cache_dtype, Tensor k_scale, Tensor v_scale) -> ()",
    "_C_cache_ops::convert_fp8": "(Tensor! dst_cache, Tensor src_cache, float scale, str kv_cache_dtype) -> ()",
    "_rocm_C::wvSplitKQ": "(Tensor in_a, Tensor in_b, Tensor! out_c, Tensor scale_a, Tensor scale_b, int CuCount) -> ()",
    # csrc/torch_bindings.cpp:117-120, 188-211
    "_C::silu_and_mul_quant": "(Tensor! result, Tensor input, Tensor scale) -> ()",
    "_C::rms_norm_static_fp8_quant": "(Tensor! result, Tensor input, Tensor weight, Tensor scale, float epsilon) -> ()",
    "_C::fused_add_rms_norm_static_fp8_quant": "(Tensor! result, Tensor input, Tensor! residual, Tensor weight, Tensor scale, float epsilon) -> ()",
    "_C::rms_norm_dynamic_per_token_quant": "(Tensor! result, Tensor input, Tensor weight, Tensor! scale, float epsilon, Tensor? scale_ub, Tensor!? residual) -> ()",
}


def test_torch_ops_registered_with_reference_schemas():
    import vllm_amd._custom_ops  # noqa: F401
    for name, want in SCHEMAS.items():
        ns, op = name.split("::")
        schema = getattr(getattr(torch.ops, ns), op).default._schema
        assert str(schema) == str(torch._C.parse_schema(name + want)), str(schema)
    for ns, op in (("_C", "paged_attention_v2"), ("_rocm_C", "paged_attention"),
                   ("_C_cache_ops", "reshape_and_cache_flash"), ("_C_cache_ops", "copy_blocks"),
                   ("_C_cache_ops", "swap_blocks"), ("_C", "dynamic_scaled_fp8_quant")):
        assert hasattr(getattr(torch.ops, ns), op)


def _err(rc):
    from vllm_amd import _lib
    assert rc != 0
    return _lib.lib.vllm_amd_last_error().decode()


def test_validation_errors_without_launch():
    """Errors are decided on the host from shapes / enums only (dummy non-null pointers are
    never dereferenced, nothing is launched): same messages as the reference's TORCH_CHECKs."""
    from vllm_amd import _lib
    L = _lib.lib
    p = 0x1000  # aligned, never touched
    def v1(head_size=128, block_size=16, dtype=_lib.BF16, kv=_lib.KV_AUTO, num_seqs=1, heads=8, kvh=8):
        return L.vllm_amd_paged_attention_v1(p, p, p, p, num_seqs, heads, head_size, kvh, 1.0, p, p, 4,
                                             block_size, 64, None, dtype, kv, p, p, heads * head_size,
                                             kvh * head_size * block_size, head_size * block_size,
                                             0, 0, 0, 64, 0, None)
    assert "Unsupported head size: 100" in _err(v1(head_size=100))
    assert "Unsupported block size: 24" in _err(v1(block_size=24))
    assert "Unsupported data type of kv cache" in _err(v1(kv=7))
    assert "Unsupported data type" in _err(v1(dtype=9))
    assert "not divisible" in _err(v1(heads=8, kvh=3))
    assert v1(num_seqs=0) == 0          # empty batch: nothing to do, no launch
    # v2: scratch too small for max_seq_len
    rc = L.vllm_amd_paged_attention_v2(p, p, p, p, p, p, p, 1, 8, 128, 8, 1.0, p, p, 4, 16, 4096, 2,
                                       None, _lib.BF16, _lib.KV_AUTO, p, p, 1024, 16384, 2048,
                                       0, 0, 0, 64, 0, None)
    assert "max_num_partitions 2 too small" in _err(rc)
    # rocm op: fp32 query is not accepted (csrc/rocm/attention.cu:3494-3539)
    rc = L.vllm_amd_paged_attention_rocm(p, p, p, p, p, p, p, 1, 8, 128, 8, 1.0, p, p, None, 4, 16,
                                         64, 1, None, _lib.F32, _lib.KV_AUTO, p, p, None, 1024,
                                         16384, 2048, None)
    assert "Unsupported data type" in _err(rc)
    # cache write
    rc = L.vllm_amd_reshape_and_cache(p, p, p, p, p, 4, 8, 120, 16, 16, 960, 960, _lib.BF16,
                                      _lib.KV_FP8_E4M3, p, p, None)
    assert "not divisible by x" in _err(rc)
    assert L.vllm_amd_reshape_and_cache(p, p, p, p, p, 0, 8, 128, 16, 8, 1024, 1024, _lib.BF16,
                                        _lib.KV_AUTO, None, None, None) == 0
    # quant / gemm
    assert L.vllm_amd_static_scaled_fp8_quant(p, p, p, 0, 128, _lib.BF16, None) == 0
    assert "unsupported dtype" in _err(L.vllm_amd_static_scaled_fp8_quant(p, p, p, 4, 128, 9, None))
    rc = L.vllm_amd_scaled_mm(p, p, p, p, p, None, 4, 8, 100, 100, 100, 8, 1, 1, _lib.BF16, None)
    assert "multiples of 16" in _err(rc)
    rc = L.vllm_amd_scaled_mm(p, p, p, p, p, None, 4, 8, 128, 128, 128, 8, 1, 1, _lib.F32, None)
    assert "bfloat16 or float16" in _err(rc)
    rc = L.vllm_amd_scaled_mm(p, p, p, p, p, None, 4, 8, 128, 128, 128, 8, 3, 1, _lib.BF16, None)
    assert "per-tensor or per-row" in _err(rc)
    assert "batch 5 > 4" in _err(L.vllm_amd_wvSplitKQ(p, p, p, p, p, 64, 128, 5, _lib.BF16, 256, None))
    assert "unknown key" in _err(L.vllm_amd_set_tuning(b"nope", 1))
    for key, default in ((b"pa_variant", 1), (b"pa_sp", 0), (b"pa_group", 0), (b"mm_kernel", 0), (b"mm_splitk", 0),
                         (b"pf_rows", 0), (b"uni_decode", 1), (b"cache_runs", 1)):
        assert L.vllm_amd_set_tuning(key, default) == 0, key   # every documented knob exists


def test_torch_layer_rejects_cpu_tensors_and_bad_strings():
    from vllm_amd import _custom_ops as ops, _torch_ops
    x = torch.zeros((2, 16), dtype=torch.bfloat16)
    # CPU tensors: the operators exist for the CUDA(=HIP) key only, the dispatcher refuses them
    # (NotImplementedError, a RuntimeError) exactly as it does for the reference's C++ ops
    with pytest.raises(RuntimeError, match="'CPU' backend"):
        _torch_ops.static_scaled_fp8_quant(torch.zeros((2, 16), dtype=torch.float8_e4m3fn), x,
                                           torch.ones(1))
    # the implementations live in the C++ shim, not in Python
    assert "torch_shim.cpp" in torch._C._dispatch_dump("_C::cutlass_scaled_mm")
    assert "torch_shim.cpp" in torch._C._dispatch_dump("_C_cache_ops::swap_blocks")
    with pytest.raises(RuntimeError, match="Unsupported data type of kv cache: fp8_e5m2"):
        _torch_ops._kv("fp8_e5m2")
    assert ops.FP8_DTYPE == torch.float8_e4m3fn


def test_validation_errors_of_the_neighbour_ops():
    """Same host-side validation for the SURVEY 8f entry points: nothing is launched."""
    from vllm_amd import _lib
    L = _lib.lib
    p = 0x1000
    # fused norm / activation quant
    assert L.vllm_amd_rms_norm_static_fp8_quant(p, p, p, p, 1e-6, 0, 4096, 4096, _lib.BF16, None) == 0
    assert "unsupported input dtype" in _err(
        L.vllm_amd_rms_norm_static_fp8_quant(p, p, p, p, 1e-6, 4, 4096, 4096, 9, None))
    assert "input stride" in _err(
        L.vllm_amd_rms_norm_static_fp8_quant(p, p, p, p, 1e-6, 4, 4096, 1024, _lib.BF16, None))
    assert "null scale or residual" in _err(
        L.vllm_amd_fused_add_rms_norm_static_fp8_quant(p, p, None, p, p, 1e-6, 4, 64, 64, _lib.BF16, None))
    assert "null scales" in _err(
        L.vllm_amd_rms_norm_dynamic_per_token_quant(p, p, p, None, 1e-6, None, None, 4, 64, _lib.BF16, None))
    assert "float16 or bfloat16" in _err(L.vllm_amd_silu_and_mul_quant(p, p, p, 4, 64, _lib.F32, None))
    assert L.vllm_amd_silu_and_mul_quant(p, p, p, 0, 64, _lib.BF16, None) == 0
    # slot mapping / swap
    assert "must be positive" in _err(L.vllm_amd_compute_slot_mapping(p, p, p, p, 4, 8, 0, None))
    assert L.vllm_amd_compute_slot_mapping(p, p, p, p, 0, 8, 16, None) == 0
    assert L.vllm_amd_swap_blocks(p, p, 4096, p, 0, None) == 0
    assert "null pointer" in _err(L.vllm_amd_swap_blocks(p, p, 4096, None, 2, None))

    # prefill attention
    def pf(head_size=128, block_size=16, dtype=_lib.BF16, kv=_lib.KV_AUTO, heads=8, kvh=2, batch=1,
           qs=1024):
        return L.vllm_amd_context_attention_fwd(
            p, p, p, p, p, p, p, p, p, p, p, None, None, batch, heads, kvh, head_size, block_size,
            128, 16, qs, head_size, kvh * head_size, head_size, kvh * head_size, head_size,
            heads * head_size, head_size, kvh * head_size * block_size, head_size * block_size,
            kvh * head_size * block_size, head_size * block_size, 0.1, 0, 0, 0, dtype, kv, None)
    assert pf(batch=0) == 0
    assert "unsupported head size: 264" in _err(pf(head_size=264))
    assert "unsupported head size: 24" in _err(pf(head_size=24, kv=_lib.KV_FP8_E4M3, qs=192))
    assert "unsupported block size: 8" in _err(pf(block_size=8))
    assert "float16, bfloat16 or float32" in _err(pf(dtype=7))
    assert "float32 queries need num_tokens" in _err(pf(dtype=_lib.F32))
    assert "unsupported head size: 260" in _err(pf(dtype=_lib.F32, head_size=260))
    assert "Unsupported FP8 dtype" in _err(pf(kv=5))
    assert "multiple of num_kv_heads" in _err(pf(heads=8, kvh=3))
    assert "16-byte aligned" in _err(pf(qs=1021))

    # unified attention (flash-layout cache)
    def ua(head_size=128, block_size=16, dtype=_lib.BF16, kv=_lib.KV_AUTO, heads=8, kvh=2, seqs=1,
           qs=1024, window=0, softcap=0.0, kd=p, nseg=0):
        return L.vllm_amd_unified_attention(
            p, p, p, p, p, p, p, kd, kd, None, None, seqs, heads, kvh, head_size, block_size, 64,
            16, qs, head_size, heads * head_size, head_size, block_size * kvh * head_size,
            kvh * head_size, head_size, block_size * kvh * head_size, kvh * head_size, head_size,
            0.1, window, softcap, 64, 4096, None, None, None, nseg, dtype, kv, None)
    assert ua(seqs=0) == 0
    assert "segment buffers" in _err(ua(nseg=16))
    assert "unsupported head size: 264" in _err(ua(head_size=264))
    assert "unsupported head size: 40" in _err(ua(head_size=40, kv=_lib.KV_FP8_E4M3, qs=320))
    assert "unsupported block size: 48" in _err(ua(block_size=48))
    assert "float16 or bfloat16" in _err(ua(dtype=_lib.F32))
    assert "Unsupported FP8 dtype" in _err(ua(kv=7))
    assert "k_descale and v_descale" in _err(ua(kv=_lib.KV_FP8_E4M3, kd=None))
    assert "must not be negative" in _err(ua(softcap=-1.0))
    assert "16-byte" in _err(ua(qs=1020))


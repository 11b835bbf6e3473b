"""GPU parity of the V1 attention backend mirror (vllm_amd/v1/attention/backends/triton_attn.py,
reference vllm/v1/attention/backends/triton_attn.py:200-384): TritonAttentionImpl.forward =
cache write + attention on a mixed batch, against the oracle's reshape_and_cache_flash +
unified_attention, through both of the reference's paths (unified_attention over the flash
layout; VLLM_V1_USE_PREFILL_DECODE_ATTENTION: context_attention_fwd + paged_attention_rocm over
the x layout)."""
import types

import pytest
import torch

from helpers import make_unified_case

pytestmark = pytest.mark.gpu

SEQS = [(1, 500), (70, 70), (1, 64), (33, 200), (1, 1), (9, 9), (1, 129), (150, 1100), (1, 31),
        (1, 1000), (2, 2), (1, 333), (1, 65), (17, 80), (1, 256), (1, 255), (1, 77), (5, 600)]


def _to_x_layout(kc_f, vc_f):
    """flash [NB, BS, KVH, D] -> x layout K [NB, KVH, D/x, BS, x], V [NB, KVH, D, BS]"""
    NB, BS, KVH, D = kc_f.shape
    x = 16 // kc_f.element_size()
    kx = kc_f.permute(0, 2, 3, 1).reshape(NB, KVH, D // x, x, BS).permute(0, 1, 2, 4, 3).contiguous()
    vx = vc_f.permute(0, 2, 3, 1).contiguous()
    return kx, vx


@pytest.mark.parametrize("path", ["unified", "unified_nhd", "prefill_decode"])
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("heads", [(8, 2), (32, 8)])
@pytest.mark.parametrize("mode", ["plain", "alibi", "window", "softcap"])
@pytest.mark.parametrize("D", [128, 256])
def test_triton_attention_impl_forward(oracle, monkeypatch, path, fp8, heads, mode, D):
    if path == "prefill_decode" and mode == "softcap":
        pytest.skip("the prefill + paged-decode path has no soft cap (nor has the reference's)")
    if D == 256 and heads == (32, 8):
        pytest.skip("head 256 runs on the smaller head grid only (time)")
    from vllm_amd.v1.attention.backends import (TritonAttentionBackend, TritonAttentionImpl,
                                                TritonAttentionMetadataBuilder)
    monkeypatch.setenv("VLLM_V1_USE_PREFILL_DECODE_ATTENTION", "1" if path == "prefill_decode" else "0")
    monkeypatch.setenv("VLLM_KV_CACHE_LAYOUT", "NHD" if path == "unified_nhd" else "HND")
    nhd = path == "unified_nhd"
    if nhd:
        path = "unified"
    H, KVH, BS, dtype = heads[0], heads[1], 16, torch.bfloat16
    c = make_unified_case(SEQS, H, KVH, D, BS, dtype, fp8=fp8, seed=H + fp8, kv_scale=0.5)
    T = c["q"].shape[0]
    g = torch.Generator().manual_seed(7)
    key = torch.randn((T + 3, KVH, D), generator=g).to(dtype)     # 3 padding tokens at the end
    value = torch.randn((T + 3, KVH, D), generator=g).to(dtype)
    query = torch.cat([c["q"], torch.zeros((3, H, D), dtype=dtype)])
    slots = []
    for b, (ql, kl) in enumerate(SEQS):
        for pos in range(kl - ql, kl):
            slots.append(int(c["block_table"][b, pos // BS]) * BS + pos % BS)
    slots = torch.tensor(slots + [-1, -1, -1], dtype=torch.int64)
    kvd = "fp8" if fp8 else "auto"
    sc = float(c["scale"]) if fp8 else 1.0
    window = 40 if mode == "window" else None
    softcap = 20.0 if mode == "softcap" else None
    slopes = [2.0 ** -(0.25 * i + 1) for i in range(H)] if mode == "alibi" else None

    # expected: the oracle writes the new tokens, then attends
    kc_o, vc_o = c["k"].clone(), c["v"].clone()
    oracle.reshape_and_cache_flash(key[:T], value[:T], kc_o, vc_o, slots[:T], kvd, sc, sc)
    ref = torch.zeros((T, H, D), dtype=dtype)
    # the reference hands `self.sliding_window[0]` = window - 1 to chunked_prefill_paged_decode
    # (triton_attn.py:349), whose kernels keep the keys with q_pos - k_pos < that value: one key
    # fewer than unified_attention's window_size = (window - 1, 0).  Mirrored as is.
    weff = None if window is None else (window - 1 if path == "prefill_decode" else window)
    ws = (weff - 1, 0) if window is not None else (-1, -1)
    kd = c["scale"] if fp8 else None
    oracle.unified_attention(c["q"], kc_o, vc_o, ref, c["cu_seqlens_q"], c["max_seqlen_q"],
                             c["seqused_k"], c["max_seqlen_k"], D ** -0.5, True, ws, c["block_table"],
                             softcap or 0.0, None, kd, kd,
                             None if slopes is None else torch.tensor(slopes), None)

    if path == "prefill_decode":
        # this path reads the NEW tokens of a prefill chunk from the dense key / value, not from
        # the (quantised) cache: expected rows of the chunks = the oracle's context_attention_fwd
        kx_o, vx_o = _to_x_layout(kc_o, vc_o)
        ref_pf = torch.zeros_like(ref)
        oracle.context_attention_fwd(c["q"], key[:T], value[:T], ref_pf, kvd, kx_o, vx_o, c["block_table"],
                                     c["cu_seqlens_q"], c["seqused_k"], c["max_seqlen_k"], c["max_seqlen_q"],
                                     sc, sc, None if slopes is None else torch.tensor(slopes), weff,
                                     D ** -0.5)
        for b, (ql, _) in enumerate(SEQS):
            if ql > 1:
                lo = int(c["cu_seqlens_q"][b])
                ref[lo:lo + ql] = ref_pf[lo:lo + ql]

    # the backend
    dev = torch.device("cuda", 0)
    impl = TritonAttentionImpl(H, D, D ** -0.5, KVH, slopes, window, kvd, softcap)
    if path == "unified":
        # allocated as GPUModelRunner._reshape_kv_cache_tensors does (:2699-2726): memory in the
        # backend's stride order, the logical shape of get_kv_cache_shape as a permuted view
        shape = TritonAttentionBackend.get_kv_cache_shape(c["k"].shape[0], BS, KVH, D)
        order = TritonAttentionBackend.get_kv_cache_stride_order()
        assert order == ((0, 1, 2, 3, 4) if nhd else (0, 1, 3, 2, 4))
        inv = [order.index(i) for i in range(len(order))]
        kv_cache = torch.empty([shape[i] for i in order], dtype=c["k"].dtype, device=dev).permute(*inv)
        assert tuple(kv_cache.shape) == shape and kv_cache.is_contiguous() == nhd
        kv_cache.copy_(torch.stack([c["k"], c["v"]]))
    else:
        kx, vx = _to_x_layout(c["k"], c["v"])
        kv_cache = torch.stack([kx.reshape(kx.shape[0], -1), vx.reshape(vx.shape[0], -1)]).to(dev)
    layer = types.SimpleNamespace(_k_scale=torch.tensor(sc, device=dev), _v_scale=torch.tensor(sc, device=dev),
                                  _q_scale=1.0)
    common = types.SimpleNamespace(
        num_actual_tokens=T, max_query_len=c["max_seqlen_q"], query_start_loc=c["cu_seqlens_q"].to(dev),
        seq_lens=c["seqused_k"].to(dev), seq_lens_cpu=c["seqused_k"], block_table_tensor=c["block_table"].to(dev),
        slot_mapping=slots.to(dev))
    builder = TritonAttentionMetadataBuilder(types.SimpleNamespace(block_size=BS), None, dev)
    md = builder.build(0, common)
    assert md.max_seq_len == c["max_seqlen_k"] and md.use_cascade is False
    output = torch.full((T + 3, H, D), 7.0, dtype=dtype, device=dev)
    ret = impl.forward(layer, query.to(dev), key.to(dev), value.to(dev), kv_cache, md, output=output)
    assert ret is output
    torch.cuda.synchronize()

    # cache contents: bit-exact with the oracle's write (compared as integers: unused slots of
    # a 16-bit cache hold NaN on purpose)
    bits = lambda t: t.contiguous().view(torch.int16 if t.element_size() == 2 else torch.uint8)
    if path == "unified":
        assert torch.equal(bits(kv_cache[0].cpu()), bits(kc_o)) and torch.equal(bits(kv_cache[1].cpu()), bits(vc_o))
    else:
        kx_o, vx_o = _to_x_layout(kc_o, vc_o)
        assert torch.equal(bits(kv_cache[0].cpu().view(kx_o.shape)), bits(kx_o))
        assert torch.equal(bits(kv_cache[1].cpu().view(vx_o.shape)), bits(vx_o))
    torch.testing.assert_close(output[:T].cpu().float(), ref.float(), atol=1.5e-2, rtol=1e-2)
    assert torch.all(output[T:] == 7.0), "padding rows past num_actual_tokens must stay untouched"


def test_profiling_run_and_unsupported():
    from vllm_amd.v1.attention.backends import TritonAttentionImpl
    impl = TritonAttentionImpl(8, 128, 0.1, 2, None, None, "auto")
    out = torch.zeros((2, 8, 128), dtype=torch.bfloat16, device="cuda")
    assert impl.forward(None, out, None, None, None, None, output=out) is out   # attn_metadata None
    with pytest.raises(ValueError, match="Head size 80 is not supported by TritonAttention"):
        TritonAttentionImpl(8, 80, 0.1, 2, None, None, "auto")
    with pytest.raises(NotImplementedError, match="Encoder self-attention"):
        TritonAttentionImpl(8, 128, 0.1, 2, None, None, "auto", attn_type="encoder")


# ---- V0 engine: ROCmFlashAttentionBackend / Impl (vllm/attention/backends/rocm_flash_attn.py) ----
V0_SEQS = [(70, 70), (33, 200), (9, 9), (150, 1100), (2, 2), (17, 80), (5, 600),        # prefills first
           (1, 500), (1, 64), (1, 1), (1, 129), (1, 31), (1, 1000), (1, 333), (1, 65), (1, 256)]


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("heads", [(8, 2), (32, 8)])
@pytest.mark.parametrize("mode", ["plain", "alibi", "fp8_out"])
def test_rocm_flash_attention_impl_forward(oracle, fp8, heads, mode):
    """ROCmFlashAttentionImpl.forward on |prefill chunks|decode tokens|: cache write (x layout)
    bit-exact with the oracle's reshape_and_cache, prefill rows against the oracle's
    context_attention_fwd, decode rows against its attention over the written cache."""
    from vllm_amd.attention.backends.rocm_flash_attn import (ROCmFlashAttentionBackend,
                                                             ROCmFlashAttentionImpl,
                                                             ROCmFlashAttentionMetadata)
    assert ROCmFlashAttentionBackend.get_name() == "ROCM_FLASH"
    assert ROCmFlashAttentionBackend.get_impl_cls() is ROCmFlashAttentionImpl
    H, KVH, D, BS, dtype = heads[0], heads[1], 128, 16, torch.bfloat16
    c = make_unified_case(V0_SEQS, H, KVH, D, BS, dtype, fp8=fp8, seed=3 * H + fp8, kv_scale=0.5)
    T = c["q"].shape[0]
    npf = sum(1 for ql, _ in V0_SEQS if ql > 1)
    npf_tok = sum(ql for ql, _ in V0_SEQS if ql > 1)
    g = torch.Generator().manual_seed(11)
    key = torch.randn((T, KVH, D), generator=g).to(dtype)
    value = torch.randn((T, KVH, D), generator=g).to(dtype)
    slots = []
    for b, (ql, kl) in enumerate(V0_SEQS):
        for pos in range(kl - ql, kl):
            slots.append(int(c["block_table"][b, pos // BS]) * BS + pos % BS)
    slots = torch.tensor(slots, dtype=torch.int64)
    kvd = "fp8" if fp8 else "auto"
    sc = float(c["scale"]) if fp8 else 1.0
    slopes = [2.0 ** -(0.25 * i + 1) for i in range(H)] if mode == "alibi" else None
    slopes_t = None if slopes is None else torch.tensor(slopes)

    kc_o, vc_o = c["k"].clone(), c["v"].clone()
    oracle.reshape_and_cache_flash(key, value, kc_o, vc_o, slots, kvd, sc, sc)
    ref = torch.zeros((T, H, D), dtype=dtype)
    kd = c["scale"] if fp8 else None
    oracle.unified_attention(c["q"], kc_o, vc_o, ref, c["cu_seqlens_q"], c["max_seqlen_q"], c["seqused_k"],
                             c["max_seqlen_k"], D ** -0.5, True, (-1, -1), c["block_table"], 0.0, None, kd, kd,
                             slopes_t, None)
    kx_o, vx_o = _to_x_layout(kc_o, vc_o)
    ref_pf = torch.zeros_like(ref)
    oracle.context_attention_fwd(c["q"], key, value, ref_pf, kvd, kx_o, vx_o, c["block_table"],
                                 c["cu_seqlens_q"], c["seqused_k"], c["max_seqlen_k"], c["max_seqlen_q"],
                                 sc, sc, slopes_t, None, D ** -0.5)
    ref[:npf_tok] = ref_pf[:npf_tok]

    dev = torch.device("cuda", 0)
    shape = ROCmFlashAttentionBackend.get_kv_cache_shape(c["k"].shape[0], BS, KVH, D)
    kx, vx = _to_x_layout(c["k"], c["v"])
    kv_cache = torch.stack([kx.reshape(kx.shape[0], -1), vx.reshape(vx.shape[0], -1)]).to(dev)
    assert tuple(kv_cache.shape) == shape
    layer = types.SimpleNamespace(_k_scale=torch.tensor(sc, device=dev), _v_scale=torch.tensor(sc, device=dev))
    kvl = c["seqused_k"]
    md = ROCmFlashAttentionMetadata(
        seq_lens_tensor=kvl.to(dev), max_decode_seq_len=int(kvl[npf:].max()),
        block_tables=c["block_table"].to(dev), num_prefills=npf, num_prefill_tokens=npf_tok,
        num_decode_tokens=T - npf_tok, slot_mapping=slots.to(dev), seq_lens=kvl.tolist(),
        max_prefill_seq_len=int(kvl[:npf].max()), max_query_len=c["max_seqlen_q"],
        query_start_loc=c["cu_seqlens_q"].to(dev),
        seq_start_loc=torch.cat([torch.zeros(1, dtype=torch.int32), kvl.cumsum(0).to(torch.int32)]).to(dev),
        context_lens_tensor=(kvl - torch.tensor([q for q, _ in V0_SEQS], dtype=torch.int32)).to(dev))
    assert md.decode_metadata.block_tables.shape[0] == len(V0_SEQS) - npf
    assert md.prefill_metadata.query_start_loc.shape[0] == npf + 1
    impl = ROCmFlashAttentionImpl(H, D, D ** -0.5, KVH, slopes, None, kvd)
    out_scale = torch.tensor([0.05], device=dev) if mode == "fp8_out" else None
    output = torch.zeros((T, H * D), dtype=torch.float8_e4m3fn if out_scale is not None else dtype, device=dev)
    ret = impl.forward(layer, c["q"].reshape(T, H * D).to(dev), key.reshape(T, KVH * D).to(dev),
                       value.reshape(T, KVH * D).to(dev), kv_cache, md, output=output, output_scale=out_scale)
    torch.cuda.synchronize()
    assert ret.shape == (T, H * D) and ret.data_ptr() == output.data_ptr()
    bits = lambda t: t.contiguous().view(torch.int16 if t.element_size() == 2 else torch.uint8)
    assert torch.equal(bits(kv_cache[0].cpu().view(kx_o.shape)), bits(kx_o))
    assert torch.equal(bits(kv_cache[1].cpu().view(vx_o.shape)), bits(vx_o))
    got = output.cpu().float().view(T, H, D)
    if out_scale is None:
        torch.testing.assert_close(got, ref.float(), atol=1.5e-2, rtol=1e-2)
    else:
        # fp8 output = quantise(attention / scale): compare dequantised, one e4m3 step of slack
        torch.testing.assert_close(got * 0.05, ref.float(), atol=1.5e-2 + 0.05 * 0.07, rtol=7e-2)


@pytest.mark.parametrize("window", [None, 48])
def test_rocm_flash_attention_prompt_run_without_block_tables(oracle, window):
    """Prompt run before any block table exists (rocm_flash_attn.py:779-860; the profiling run
    has no kv cache at all): q and k are the prompts themselves."""
    from vllm_amd.attention.backends.rocm_flash_attn import ROCmFlashAttentionImpl, ROCmFlashAttentionMetadata
    H, KVH, D, dtype = 8, 2, 128, torch.float16
    lens = [37, 128, 5, 300]
    T = sum(lens)
    g = torch.Generator().manual_seed(5)
    q = torch.randn((T, H, D), generator=g).to(dtype)
    k = torch.randn((T, KVH, D), generator=g).to(dtype)
    v = torch.randn((T, KVH, D), generator=g).to(dtype)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    # expected: fp32 causal (+ local) attention per prompt
    ref = torch.zeros((T, H, D))
    for b, n in enumerate(lens):
        lo = int(cu[b])
        qq = q[lo:lo + n].float().transpose(0, 1)
        kk = k[lo:lo + n].float().repeat_interleave(H // KVH, 1).transpose(0, 1)
        vv = v[lo:lo + n].float().repeat_interleave(H // KVH, 1).transpose(0, 1)
        s = qq @ kk.transpose(1, 2) * D ** -0.5
        i, j = torch.arange(n)[:, None], torch.arange(n)[None, :]
        mask = j > i
        if window is not None:
            mask |= j <= i - window
        s.masked_fill_(mask, float("-inf"))
        ref[lo:lo + n] = (torch.softmax(s, -1) @ vv).transpose(0, 1)
    dev = torch.device("cuda", 0)
    md = ROCmFlashAttentionMetadata(
        seq_lens_tensor=torch.tensor(lens, dtype=torch.int32, device=dev), max_decode_seq_len=0,
        block_tables=torch.empty((len(lens), 0), dtype=torch.int32, device=dev), num_prefills=len(lens),
        num_prefill_tokens=T, num_decode_tokens=0, slot_mapping=torch.zeros(T, dtype=torch.int64, device=dev),
        seq_lens=lens, max_prefill_seq_len=max(lens), max_query_len=max(lens), query_start_loc=cu.to(dev),
        seq_start_loc=cu.to(dev), context_lens_tensor=torch.zeros(len(lens), dtype=torch.int32, device=dev))
    layer = types.SimpleNamespace(_k_scale=torch.tensor(1.0, device=dev), _v_scale=torch.tensor(1.0, device=dev))
    impl = ROCmFlashAttentionImpl(H, D, D ** -0.5, KVH, None, window, "auto")
    output = torch.zeros((T, H * D), dtype=dtype, device=dev)
    impl.forward(layer, q.view(T, -1).to(dev), k.view(T, -1).to(dev), v.view(T, -1).to(dev),
                 torch.empty(0, device=dev), md, output=output)
    torch.cuda.synchronize()
    torch.testing.assert_close(output.cpu().float().view(T, H, D), ref, atol=2e-3, rtol=2e-3)


def test_rocm_flash_attention_impl_rejects():
    from vllm_amd.attention.backends.rocm_flash_attn import ROCmFlashAttentionImpl
    with pytest.raises(ValueError, match="Head size 100 is not supported by PagedAttention"):
        ROCmFlashAttentionImpl(8, 100, 0.1, 2, None, None, "auto")
    with pytest.raises(NotImplementedError, match="KV sharing is not supported"):
        ROCmFlashAttentionImpl(8, 128, 0.1, 2, None, None, "auto", kv_sharing_target_layer_name="x")
    with pytest.raises(NotImplementedError, match="decoder self-attention only"):
        ROCmFlashAttentionImpl(8, 128, 0.1, 2, None, None, "auto", attn_type="encoder")


# ---- decode side of chunked_prefill_paged_decode with a sliding window ----------------------
@pytest.mark.parametrize("window", [1, 40, 256, 700, 5000])
@pytest.mark.parametrize("shape", [(32, 8, 128, 16, True), (8, 2, 128, 16, False), (4, 4, 64, 32, True),
                                   (8, 8, 256, 16, False), (16, 1, 96, 16, False)])
@pytest.mark.parametrize("mode", ["plain", "alibi", "fp8_out"])
def test_paged_decode_sliding_window(oracle, window, shape, mode):
    """vllm_amd_paged_attention_2d (kernel_paged_attention_2d, chunked_prefill_paged_decode.py:27-212):
    decode tokens see the last `window` keys only; windows inside one partition, across partitions,
    longer than the sequence; MFMA path (head <= 128) and generic path (head 256); the mixed batch's
    prefill chunks are skipped through query_start_loc."""
    from vllm_amd.attention.ops.chunked_prefill_paged_decode import chunked_prefill_paged_decode
    H, KVH, D, BS, fp8 = shape
    dtype = torch.bfloat16
    seqs = [(1, 1), (1, 39), (1, 40), (1, 41), (1, 255), (1, 256), (1, 257), (1, 700), (1, 1300), (1, 2049),
            (1, 3000), (1, 513)]
    c = make_unified_case(seqs, H, KVH, D, BS, dtype, fp8=fp8, seed=window + H, kv_scale=0.5)
    T = len(seqs)
    slopes = torch.tensor([2.0 ** -(0.25 * i + 1) for i in range(H)]) if mode == "alibi" else None
    kd = c["scale"] if fp8 else None
    ref = torch.zeros((T, H, D), dtype=dtype)
    oracle.unified_attention(c["q"], c["k"], c["v"], ref, c["cu_seqlens_q"], 1, c["seqused_k"],
                             c["max_seqlen_k"], D ** -0.5, True, (window - 1, 0), c["block_table"], 0.0,
                             None, kd, kd, slopes, None)
    dev = torch.device("cuda", 0)
    kx, vx = _to_x_layout(c["k"], c["v"])
    out_scale = torch.tensor([0.04], device=dev) if mode == "fp8_out" else None
    out = torch.zeros((T, H, D), dtype=torch.float8_e4m3fn if out_scale is not None else dtype, device=dev)
    sc = torch.tensor(float(c["scale"]) if fp8 else 1.0, device=dev)
    key = torch.zeros((T, KVH, D), dtype=dtype, device=dev)   # shapes only: max_query_len == 1
    chunked_prefill_paged_decode(c["q"].to(dev), key, key, out, "fp8" if fp8 else "auto", kx.to(dev), vx.to(dev),
                                 c["block_table"].to(dev), c["cu_seqlens_q"].to(dev), c["seqused_k"].to(dev),
                                 c["max_seqlen_k"], 1, sc, sc, None if slopes is None else slopes.to(dev),
                                 window, D ** -0.5, out_scale)
    torch.cuda.synchronize()
    got = out.cpu().float()
    if out_scale is None:
        torch.testing.assert_close(got, ref.float(), atol=1.5e-2, rtol=1e-2)
    else:
        torch.testing.assert_close(got * 0.04, ref.float(), atol=1.5e-2 + 0.04 * 0.07, rtol=7e-2)

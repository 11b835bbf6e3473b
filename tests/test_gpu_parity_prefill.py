"""GPU parity: prefill / chunked-prefill attention with a paged prefix
(`context_attention_fwd`, C ABI `vllm_amd_context_attention_fwd`) against the CPU oracle and
against an fp32 torch attention (the check of the reference's own test,
tests/kernels/attention/test_prefix_prefill.py:195-232).

Stated tolerance: the reference test allows atol 1e-4 (auto) / 1e-3 (fp8) on inputs of
amplitude 1e-3; here inputs have amplitude 1 (outputs O(1)), and the bound is the rounding of
the probabilities to the value dtype: atol 1.5e-3 (fp16) / 6e-3 (bf16), rtol 0, against both
the oracle and the fp32 torch reference.
"""
import pytest
import torch

from helpers import make_prefill_case, prefill_reference

pytestmark = pytest.mark.gpu


def run(c, kv, dev_extra=None, **kw):
    from vllm_amd.attention.ops import context_attention_fwd
    o = torch.zeros_like(c["q"]).cuda() if kw.get("fp8_out_scale") is None else \
        torch.zeros(c["q"].shape, dtype=torch.uint8).cuda()
    sc = c["scale"].cuda()
    context_attention_fwd(c["q"].cuda(), c["k"].cuda(), c["v"].cuda(), o, kv, c["k_cache"].cuda(),
                          c["v_cache"].cuda(), c["b_loc"].cuda(), c["b_start_loc"].cuda(),
                          c["b_seq_len"].cuda(), c["max_seq_len"], c["max_input_len"], sc, sc, **kw)
    return o.cpu()


def run_oracle(oracle, c, kv, **kw):
    o = torch.zeros_like(c["q"]) if kw.get("fp8_out_scale") is None else \
        torch.zeros(c["q"].shape, dtype=torch.uint8)
    oracle.context_attention_fwd(c["q"], c["k"], c["v"], o, kv, c["k_cache"], c["v_cache"],
                                 c["b_loc"], c["b_start_loc"], c["b_seq_len"], c["max_seq_len"],
                                 c["max_input_len"], float(c["scale"]), float(c["scale"]), **kw)
    return o


def tol(dtype):
    return 6e-3 if dtype == torch.bfloat16 else 1.5e-3


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("kv", ["auto", "fp8"])
@pytest.mark.parametrize("head_size", [128, 64, 96, 32])
@pytest.mark.parametrize("block_size", [16, 32])
@pytest.mark.parametrize("gqa", [1, 4])
def test_prefill_shapes(oracle, dtype, kv, head_size, block_size, gqa):
    """Ragged batch: a decode-like sequence (1 query token), a pure prefill (no context), a
    chunked prefill whose lengths straddle tile boundaries, and a block-multiple one."""
    H = 4
    c = make_prefill_case([1, 150, 77, 128], [200, 0, 131, 64], H, H // gqa, head_size, block_size,
                          dtype, kv, seed=head_size + block_size + gqa, kv_scale=0.02)
    out = run(c, kv)
    assert not out.isnan().any()
    torch.testing.assert_close(out.float(), run_oracle(oracle, c, kv).float(), atol=tol(dtype), rtol=0)
    torch.testing.assert_close(out.float(), prefill_reference(c), atol=tol(dtype), rtol=0)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("kv", ["auto", "fp8"])
@pytest.mark.parametrize("window", [16, 64, 128, 512])   # test_prefix_prefill.py:28
def test_prefill_sliding_window(oracle, dtype, kv, window):
    c = make_prefill_case([1, 200, 90], [150, 40, 300], 4, 2, 128, 16, dtype, kv, seed=window,
                          kv_scale=0.02)
    out = run(c, kv, sliding_window=window)
    torch.testing.assert_close(out.float(), run_oracle(oracle, c, kv, sliding_window=window).float(),
                               atol=tol(dtype), rtol=0)
    torch.testing.assert_close(out.float(), prefill_reference(c, window), atol=tol(dtype), rtol=0)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("kv", ["auto", "fp8"])
def test_prefill_alibi(oracle, dtype, kv):
    H = 8
    slopes = torch.tensor([2.0 ** -(i + 1) for i in range(H)])
    c = make_prefill_case([1, 140, 60], [100, 33, 260], H, 2, 128, 16, dtype, kv, seed=11,
                          kv_scale=0.02)
    out = run(c, kv, alibi_slopes=slopes.cuda(), sliding_window=64)   # window ignored with alibi
    ref = run_oracle(oracle, c, kv, alibi_slopes=slopes, sliding_window=64)
    torch.testing.assert_close(out.float(), ref.float(), atol=tol(dtype), rtol=0)
    torch.testing.assert_close(out.float(), prefill_reference(c, 0, slopes), atol=tol(dtype), rtol=0)


@pytest.mark.parametrize("head_size", [24, 40, 80, 112, 120])   # test_prefix_prefill.py:23 has 24
@pytest.mark.parametrize("kv", ["auto", "fp8"])
def test_prefill_padded_head_sizes(oracle, head_size, kv):
    """Head sizes that are not a multiple of the MFMA K step are zero-padded inside the kernel."""
    if kv == "fp8" and head_size % 16:
        pytest.skip("fp8 cache layout needs head_size % 16 == 0 (x = 16)")
    c = make_prefill_case([1, 90, 130], [70, 0, 45], 4, 2, head_size, 16, torch.float16, kv,
                          seed=head_size, kv_scale=0.02)
    out = run(c, kv)
    assert not out.isnan().any()
    torch.testing.assert_close(out.float(), run_oracle(oracle, c, kv).float(), atol=1.5e-3, rtol=0)
    torch.testing.assert_close(out.float(), prefill_reference(c), atol=1.5e-3, rtol=0)


@pytest.mark.parametrize("kv", ["auto", "fp8"])
@pytest.mark.parametrize("head_size", [128, 96, 24])   # test_prefix_prefill.py:23
@pytest.mark.parametrize("gqa", [1, 4])
def test_prefill_float32(oracle, kv, head_size, gqa):
    """float32 queries (test_prefix_prefill.py:513-560 runs its grid with torch.float32): fp32
    arithmetic end to end; atol 1e-4 is the reference test's own bound for the 'auto' cache (:231)."""
    if kv == "fp8" and head_size % 16:
        pytest.skip("fp8 cache layout needs head_size % 16 == 0 (x = 16)")
    c = make_prefill_case([1, 150, 77, 128], [200, 0, 131, 300], 4, 4 // gqa, head_size, 16,
                          torch.float32, kv, seed=head_size + gqa, kv_scale=0.02)
    out = run(c, kv)
    assert not out.isnan().any()
    torch.testing.assert_close(out, run_oracle(oracle, c, kv), atol=1e-4, rtol=0)
    torch.testing.assert_close(out, prefill_reference(c), atol=1e-4, rtol=0)


@pytest.mark.parametrize("kv", ["auto", "fp8"])
def test_prefill_float32_window_alibi_skip(oracle, kv):
    c = make_prefill_case([1, 200, 90], [150, 40, 300], 4, 2, 128, 32, torch.float32, kv, seed=3,
                          kv_scale=0.02)
    for window in (16, 128):
        out = run(c, kv, sliding_window=window)
        torch.testing.assert_close(out, run_oracle(oracle, c, kv, sliding_window=window), atol=1e-4, rtol=0)
        torch.testing.assert_close(out, prefill_reference(c, window), atol=1e-4, rtol=0)
    slopes = torch.tensor([2.0 ** -(i + 1) for i in range(4)])
    out = run(c, kv, alibi_slopes=slopes.cuda(), sliding_window=64)
    torch.testing.assert_close(out, run_oracle(oracle, c, kv, alibi_slopes=slopes, sliding_window=64),
                               atol=1e-4, rtol=0)
    torch.testing.assert_close(out, prefill_reference(c, 0, slopes), atol=1e-4, rtol=0)
    out = run(c, kv, skip_decode=True)
    assert bool((out[0] == 0).all())   # the single-token sequence is left untouched
    torch.testing.assert_close(out[1:], prefill_reference(c)[1:], atol=1e-4, rtol=0)


def test_prefill_skip_decode_and_strides(oracle):
    """skip_decode leaves single-token sequences untouched (:104-105); q / k / v / o may be
    strided views (qkv slices of one projection output)."""
    H, KVH, D = 4, 2, 128
    c = make_prefill_case([1, 70, 1, 33], [50, 20, 9, 0], H, KVH, D, 16, torch.bfloat16, "auto", seed=5)
    T = c["q"].shape[0]
    qkv = torch.zeros((T, (H + 2 * KVH) * D), dtype=torch.bfloat16)
    qkv[:, : H * D] = c["q"].reshape(T, -1)
    qkv[:, H * D:(H + KVH) * D] = c["k"].reshape(T, -1)
    qkv[:, (H + KVH) * D:] = c["v"].reshape(T, -1)
    qkv = qkv.cuda()
    from vllm_amd.attention.ops import context_attention_fwd
    o = torch.full((T, H, D), 7.0, dtype=torch.bfloat16, device="cuda")
    one = torch.ones(1, device="cuda")
    context_attention_fwd(qkv[:, : H * D].view(T, H, D), qkv[:, H * D:(H + KVH) * D].view(T, KVH, D),
                          qkv[:, (H + KVH) * D:].view(T, KVH, D), o, "auto", c["k_cache"].cuda(),
                          c["v_cache"].cuda(), c["b_loc"].cuda(), c["b_start_loc"].cuda(),
                          c["b_seq_len"].cuda(), c["max_seq_len"], c["max_input_len"], one, one,
                          skip_decode=True)
    ref = prefill_reference(c)
    o = o.cpu().float()
    for b, ql in enumerate(c["query_lens"]):
        s = int(c["b_start_loc"][b])
        if ql == 1:
            assert bool((o[s] == 7.0).all())
        else:
            torch.testing.assert_close(o[s:s + ql], ref[s:s + ql], atol=6e-3, rtol=0)


def test_prefill_fp8_output(oracle):
    """fp8_out_scale: e4m3 output = clamp(acc / l / scale) (:282-287).  Compared after
    dequantisation: the attention tolerance plus one e4m3 step (2^-3 relative) where the
    6e-3 difference straddles a rounding boundary; and exactly saturated where it must be."""
    c = make_prefill_case([100, 40], [64, 10], 4, 4, 128, 16, torch.bfloat16, "fp8", seed=2, kv_scale=0.02)
    fs = torch.tensor([0.01])
    out = run(c, "fp8", fp8_out_scale=fs.cuda())
    ref = run_oracle(oracle, c, "fp8", fp8_out_scale=fs)
    dec = lambda t: t.view(torch.float8_e4m3fn).float() * 0.01
    assert not dec(out).isnan().any()
    torch.testing.assert_close(dec(out), dec(ref), atol=8e-3, rtol=0.13)
    full = prefill_reference(c)
    torch.testing.assert_close(dec(out), full.clamp(-4.48, 4.48), atol=8e-3, rtol=0.13)


def test_prefill_large_matches_torch():
    """configs-sized chunk (2048 new tokens over a 2048-token fp8 prefix, 8 q / 2 kv heads) vs the
    fp32 torch reference on the GPU (no oracle: size-independent check of the tiling)."""
    c = make_prefill_case([2048, 333], [2048, 1000], 8, 2, 128, 16, torch.bfloat16, "fp8", seed=9,
                          kv_scale=0.02)
    out = run(c, "fp8")
    ref = prefill_reference({k: ([t.cuda() for t in v] if isinstance(v, list) and torch.is_tensor(v[0])
                                 else (v.cuda() if torch.is_tensor(v) else v)) for k, v in c.items()})
    torch.testing.assert_close(out.float(), ref.cpu(), atol=6e-3, rtol=0)


def test_forward_prefix_wrapper(oracle):
    """PagedAttention.forward_prefix (paged_attn.py:196-239) allocates the output and forwards."""
    from vllm_amd.attention import PagedAttention
    c = make_prefill_case([40, 9], [33, 70], 4, 2, 64, 16, torch.float16, "auto", seed=4)
    one = torch.ones(1, device="cuda")
    out = PagedAttention.forward_prefix(c["q"].cuda(), c["k"].cuda(), c["v"].cuda(), "auto",
                                        c["k_cache"].cuda(), c["v_cache"].cuda(), c["b_loc"].cuda(),
                                        c["b_start_loc"].cuda(), c["b_seq_len"].cuda(),
                                        c["max_input_len"], None, None, one, one)
    torch.testing.assert_close(out.cpu().float(), prefill_reference(c), atol=1.5e-3, rtol=0)


@pytest.fixture
def force_fa():
    """The 256-row kernel (prefill_fa.hip, PREFIX form) for every launch that it supports ("pf_fa" = 2)."""
    from vllm_amd import _lib
    _lib.set_tuning("pf_fa", 2)
    yield
    _lib.set_tuning("pf_fa", 0)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("kv", ["auto", "fp8"])
@pytest.mark.parametrize("block_size", [16, 32])
@pytest.mark.parametrize("heads", [(4, 4), (4, 1), (8, 1), (6, 2), (24, 1)])
def test_prefill_fa_kernel(oracle, force_fa, dtype, kv, block_size, heads):
    """The 256-row kernel on the prefix layout: a one-token sequence, pure prefill, contexts that end inside a tile and
    on a tile boundary, several row blocks per sequence; GQA groups that divide 256 rows, one that does not (3) and
    unpacked rows (24)."""
    c = make_prefill_case([1, 150, 77, 128, 300, 64], [200, 0, 131, 64, 500, 1], heads[0], heads[1], 128, block_size,
                          dtype, kv, seed=block_size + heads[0], kv_scale=0.02)
    out = run(c, kv)
    assert not out.isnan().any()
    torch.testing.assert_close(out.float(), run_oracle(oracle, c, kv).float(), atol=tol(dtype), rtol=0)
    torch.testing.assert_close(out.float(), prefill_reference(c), atol=tol(dtype), rtol=0)


@pytest.mark.parametrize("kv", ["auto", "fp8"])
def test_prefill_fa_kernel_features(oracle, force_fa, kv):
    """skip_decode, fp8 output, and a launch without num_tokens through the C ABI (grid = blocks per sequence x batch)."""
    import ctypes
    from vllm_amd import _lib
    c = make_prefill_case([1, 90, 1, 260], [64, 70, 5, 129], 8, 2, 128, 16, torch.bfloat16, kv, seed=21, kv_scale=0.02)
    ref = run_oracle(oracle, c, kv)
    out = run(c, kv, skip_decode=True)
    cu = c["b_start_loc"].tolist()
    for i, ql in enumerate([1, 90, 1, 260]):
        if ql > 1:
            torch.testing.assert_close(out[cu[i]:cu[i + 1]].float(), ref[cu[i]:cu[i + 1]].float(), atol=6e-3, rtol=0)
        else:
            assert float(out[cu[i]:cu[i + 1]].abs().max()) == 0.0
    sc8 = torch.tensor([0.02])
    o8 = run(c, kv, fp8_out_scale=sc8.cuda())
    r8 = run_oracle(oracle, c, kv, fp8_out_scale=sc8)
    a = o8.view(torch.float8_e4m3fn).float() * 0.02
    b = r8.view(torch.float8_e4m3fn).float() * 0.02
    torch.testing.assert_close(a, b, atol=8e-3 + 6e-3, rtol=0.13)
    # num_tokens = 0: the kernel cannot enumerate the row blocks of all sequences
    d = {k: v.cuda() for k, v in c.items() if torch.is_tensor(v)}
    o = torch.zeros_like(d["q"])
    q, k, v, kc, vc = d["q"], d["k"], d["v"], d["k_cache"], d["v_cache"]
    bl, bs, bsl = d["b_loc"].int(), d["b_start_loc"].int(), d["b_seq_len"].int()
    sc = d["scale"]
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = _lib.lib.vllm_amd_context_attention_fwd(
        P(o), P(q), P(k), P(v), P(kc), P(vc), P(bl), P(bs), P(bsl), P(sc), P(sc), None, None, 4, 8, 2, 128, 16,
        c["max_input_len"], bl.stride(0), q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
        o.stride(0), o.stride(1), kc.stride(0), kc.stride(1), vc.stride(0), vc.stride(1), float(128 ** -0.5), 0, 0, 0,
        _lib.BF16, _lib.KV_FP8_E4M3 if kv == "fp8" else _lib.KV_AUTO, None)
    assert rc == 0
    torch.cuda.synchronize()
    torch.testing.assert_close(o.cpu().float(), ref.float(), atol=6e-3, rtol=0)


@pytest.mark.parametrize("head_size", [160, 192, 224, 256])
@pytest.mark.parametrize("kv", ["auto", "fp8"])
@pytest.mark.parametrize("gqa", [1, 4])
def test_prefill_wide_heads(oracle, head_size, kv, gqa):
    """Head sizes above 128 (the reference's kernel pads any head size to a power of two, prefix_prefill.py:760-790; its
    backend list goes to 256): the tile kernel's 256-wide instantiation."""
    c = make_prefill_case([1, 150, 77, 128], [200, 0, 131, 64], 4, 4 // gqa, head_size, 16, torch.bfloat16, kv,
                          seed=head_size + gqa, kv_scale=0.02)
    out = run(c, kv)
    assert not out.isnan().any()
    torch.testing.assert_close(out.float(), run_oracle(oracle, c, kv).float(), atol=tol(torch.bfloat16), rtol=0)
    torch.testing.assert_close(out.float(), prefill_reference(c), atol=tol(torch.bfloat16), rtol=0)


@pytest.mark.parametrize("head_size", [72, 80, 96, 112, 120, 136, 160, 192, 224, 256])
@pytest.mark.parametrize("kv", ["auto", "fp8"])
def test_prefill_fa_kernel_head_sizes(oracle, force_fa, head_size, kv):
    """Head sizes 72 .. 120 through the 256-row kernel (padded to 128) on the prefix layout, 136 .. 256 through its
    four-wave form (padded to 256)."""
    if kv == "fp8" and head_size % 16:
        pytest.skip("an fp8 cache needs head sizes that are multiples of 16")
    for bs in (16, 32) if head_size in (96, 256) else (16,):
        c = make_prefill_case([1, 150, 77, 300], [200, 0, 131, 64], 4, 2, head_size, bs, torch.bfloat16, kv,
                              seed=head_size, kv_scale=0.02)
        out = run(c, kv)
        assert not out.isnan().any()
        torch.testing.assert_close(out.float(), run_oracle(oracle, c, kv).float(), atol=tol(torch.bfloat16), rtol=0)

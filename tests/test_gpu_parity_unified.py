"""GPU parity: unified_attention (flash-layout cache, causal, window / softcap / ALiBi / fp8)
vs the CPU oracle's restatement of the Triton kernel.  Shapes and tolerance follow the
reference's tests/kernels/attention/test_triton_unified_attention.py:80-200 (atol 1.5e-2,
rtol 1e-2 on N(0,1) data), head sizes 128 (tile / decode kernels) and 256 (wide-head kernel)."""
import pytest
import torch

from helpers import make_unified_case

pytestmark = pytest.mark.gpu

ATOL, RTOL = 1.5e-2, 1e-2


def run(orc, c, dtype, window=None, softcap=0.0, alibi=None, out_scale=None, fp8=False, view=None):
    from vllm_amd.attention.ops import unified_attention
    ws = (window - 1, 0) if window is not None else (-1, -1)
    D = c["q"].shape[2]
    sm = D ** -0.5
    odt = torch.uint8 if out_scale is not None else dtype
    ref = torch.zeros(c["q"].shape, dtype=odt)
    kd = c["scale"] if fp8 else None
    orc.unified_attention(c["q"], c["k"], c["v"], ref, c["cu_seqlens_q"], c["max_seqlen_q"],
                          c["seqused_k"], c["max_seqlen_k"], sm, True, ws, c["block_table"],
                          softcap, None, kd, kd, alibi, out_scale)
    kc, vc = c["k"].cuda(), c["v"].cuda()
    if view is not None:
        kc, vc = view(kc), view(vc)
    out = torch.zeros(c["q"].shape, dtype=odt, device="cuda")
    unified_attention(c["q"].cuda(), kc, vc, out, c["cu_seqlens_q"].cuda(), c["max_seqlen_q"],
                      c["seqused_k"].cuda(), c["max_seqlen_k"], sm, True, ws,
                      c["block_table"].cuda(), softcap, None,
                      None if kd is None else kd.cuda(), None if kd is None else kd.cuda(),
                      None if alibi is None else alibi.cuda(),
                      None if out_scale is None else out_scale.cuda())
    if out_scale is not None:
        a = out.cpu().view(torch.float8_e4m3fn).float() * float(out_scale)
        b = ref.view(torch.float8_e4m3fn).float() * float(out_scale)
        torch.testing.assert_close(a, b, atol=8e-3 + ATOL, rtol=0.13)
    else:
        torch.testing.assert_close(out.cpu().float(), ref.float(), atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("seq_lens", [[(1, 1328), (5, 18), (129, 463)], [(1, 523), (1, 37), (1, 2011)]])
@pytest.mark.parametrize("num_heads", [(4, 4), (8, 2), (16, 2)])
@pytest.mark.parametrize("block_size", [16, 32])
@pytest.mark.parametrize("sliding_window", [None, 256])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("soft_cap", [None, 10.0, 50.0])
@pytest.mark.parametrize("head_size", [128, 256])
def test_unified_reference_grid(oracle, seq_lens, num_heads, block_size, sliding_window, dtype, soft_cap, head_size):
    """The reference test's grid (HEAD_SIZES = [128, 256])."""
    c = make_unified_case(seq_lens, num_heads[0], num_heads[1], head_size, block_size, dtype,
                          seed=len(seq_lens) + num_heads[0] + block_size)
    run(oracle, c, dtype, window=sliding_window, softcap=soft_cap or 0.0)


@pytest.mark.parametrize("D", [160, 192, 224, 256])
@pytest.mark.parametrize("mode", ["plain", "fp8", "alibi", "fp8_out"])
@pytest.mark.parametrize("kernel", ["tile", "valu"])
def test_unified_wide_heads(oracle, D, mode, kernel):
    """Head sizes 160 .. 256 of TritonAttentionBackend.get_supported_head_sizes (triton_attn.py:147-148):
    the tile kernel's 256-wide instantiation (default) and the plain VALU kernel ("pf_rows" = 2), mixed batch,
    fp8 cache / ALiBi / fp8 output."""
    from vllm_amd import _lib
    _lib.set_tuning("pf_rows", 2 if kernel == "valu" else 0)
    try:
        _wide_heads(oracle, D, mode)
    finally:
        _lib.set_tuning("pf_rows", 0)


def _wide_heads(oracle, D, mode):
    fp8 = mode in ("fp8", "fp8_out")
    c = make_unified_case([(90, 400), (1, 700), (17, 17), (1, 1), (3, 260)], 8, 2, D, 16, torch.bfloat16,
                          fp8=fp8, seed=D, kv_scale=0.5)
    slopes = 2.0 ** -(torch.arange(8).float() * 0.5 + 1) if mode == "alibi" else None
    run(oracle, c, torch.bfloat16, fp8=fp8, alibi=slopes, window=300 if mode == "alibi" else None,
        out_scale=torch.tensor([0.03]) if mode == "fp8_out" else None)


@pytest.mark.parametrize("seq_lens", [[(700, 700), (300, 1500), (64, 64), (1, 1)], [(257, 1000), (130, 131)]])
@pytest.mark.parametrize("D", [128, 64])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("window", [None, 100])
def test_unified_long_prefill(oracle, seq_lens, D, dtype, window):
    """Several row blocks per sequence: interior, diagonal and skipped tiles, pure prefill
    (ctx 0) and chunked prefill, window narrower than a row block."""
    c = make_unified_case(seq_lens, 8, 2, D, 16, dtype, seed=D + len(seq_lens))
    run(oracle, c, dtype, window=window)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("block_size", [16, 32])
@pytest.mark.parametrize("D", [128, 64])
@pytest.mark.parametrize("soft_cap", [None, 30.0])
def test_unified_fp8_cache(oracle, dtype, block_size, D, soft_cap):
    c = make_unified_case([(200, 900), (1, 77), (33, 33)], 8, 2, D, block_size, dtype, fp8=True,
                          seed=block_size + D, kv_scale=0.75)
    run(oracle, c, dtype, softcap=soft_cap or 0.0, fp8=True)


@pytest.mark.parametrize("block_size", [64, 128])
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("seq_lens", [[(1, 1328), (5, 18), (129, 463), (64, 64)], [(1, 523), (1, 37), (1, 2011), (1, 128)],
                                      [(1, 70)] * 10 + [(40, 300), (1, 129), (1, 1)] * 3])
def test_unified_large_blocks(oracle, block_size, fp8, seq_lens):
    """Pages of 64 / 128 tokens (the block size is a run-time power of two in both kernels): mixed,
    decode-only (split) and >= 16-sequence batches (decode kernel + tile kernel)."""
    c = make_unified_case(seq_lens, 8, 2, 128, block_size, torch.bfloat16, fp8=fp8, seed=block_size + len(seq_lens),
                          kv_scale=0.5)
    run(oracle, c, torch.bfloat16, fp8=fp8)
    run(oracle, c, torch.bfloat16, fp8=fp8, window=100, softcap=25.0)


@pytest.mark.parametrize("D", [32, 80, 96, 120, 128])
def test_unified_head_sizes(oracle, D):
    c = make_unified_case([(90, 400), (17, 17)], 4, 2, D, 16, torch.bfloat16, seed=D)
    run(oracle, c, torch.bfloat16)


@pytest.mark.parametrize("window", [None, 64])
def test_unified_alibi(oracle, window):
    c = make_unified_case([(150, 600), (1, 300), (40, 41)], 8, 4, 128, 16, torch.float16, seed=5)
    slopes = 2.0 ** -(torch.arange(8).float() * 0.5 + 1)
    run(oracle, c, torch.float16, alibi=slopes, window=window)


@pytest.mark.parametrize("heads", [(8, 2), (32, 8), (16, 1), (6, 2)])
@pytest.mark.parametrize("D", [128, 64])
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("mode", ["plain", "window", "alibi", "softcap"])
def test_unified_decode_batches(oracle, heads, D, fp8, mode):
    """Decode-only batches (every query length 1): the decode kernel and the GQA-packed tile
    launch (tile rows = the q heads of a kv head's group).  Lengths around the 64-key tile and
    the page size."""
    c = make_unified_case([(1, 1), (1, 64), (1, 65), (1, 300), (1, 17), (1, 1000)], heads[0], heads[1],
                          D, 16, torch.bfloat16, fp8=fp8, seed=heads[0] + D, kv_scale=0.5)
    slopes = 2.0 ** -(torch.arange(heads[0]).float() * 0.25 + 1) if mode == "alibi" else None
    from vllm_amd import _lib
    try:
        for knob in (1, 0):   # the decode kernel, then the packed tile kernel
            _lib.set_tuning("uni_decode", knob)
            run(oracle, c, torch.bfloat16, fp8=fp8, alibi=slopes, window=48 if mode == "window" else None,
                softcap=20.0 if mode == "softcap" else 0.0)
    finally:
        _lib.set_tuning("uni_decode", 1)


@pytest.mark.parametrize("heads", [(8, 2), (32, 8), (6, 2), (14, 2), (16, 1)])
@pytest.mark.parametrize("D", [128, 64])
@pytest.mark.parametrize("mode", ["plain", "window", "alibi"])
def test_unified_mixed_batch_packed(oracle, heads, D, mode):
    """Prefill chunks and decode tokens in one call, GQA groups that do and do not divide the
    tile's 64 / 128 rows (6, 7): rows are (token, head) pairs, blocks of 64 // group tokens."""
    c = make_unified_case([(1, 500), (70, 70), (1, 64), (33, 200), (9, 9), (150, 1100), (1, 1)],
                          heads[0], heads[1], D, 16, torch.float16, seed=heads[0] * 3 + D)
    slopes = 2.0 ** -(torch.arange(heads[0]).float() * 0.25 + 1) if mode == "alibi" else None
    run(oracle, c, torch.float16, alibi=slopes, window=40 if mode == "window" else None)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("block_size", [16, 32])
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("nseq", [3, 140, 260])
def test_unified_decode_kernel(oracle, dtype, block_size, fp8, nseq):
    """The flash-layout decode kernel (unified_decode.hip): split over key segments (3 and 140
    sequences: 12 and 560 workgroups) and unsplit with the XCD-grouped order (260: 1040), both
    block sizes and dtypes, lengths around the step (32 / 64 tokens), the page and the
    64-token segment tile."""
    lens = [1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 257, 700, 1, 16, 17]
    seqs = [(1, lens[i % len(lens)] + (i // len(lens))) for i in range(nseq)]
    if nseq == 3:
        seqs = [(1, 2500), (1, 64), (1, 1023)]
    c = make_unified_case(seqs, 8, 4, 128, block_size, dtype, fp8=fp8, seed=nseq + block_size, kv_scale=0.5)
    run(oracle, c, dtype, fp8=fp8)


@pytest.mark.parametrize("D", [32, 48, 80, 96, 112, 120, 136, 160, 192, 224, 256])
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("nseq", [5, 260])
def test_unified_decode_head_sizes(oracle, D, fp8, nseq):
    """Head sizes below the kernel's padded 64 / 128 / 256 (whole 16-byte K chunks): split and unsplit
    launches, and a mixed batch of >= 16 sequences."""
    if fp8 and D % 16:
        pytest.skip("an fp8 cache needs head sizes that are multiples of 16")
    lens = [1, 31, 64, 65, 129, 300, 700]
    seqs = [(1, lens[i % len(lens)] + i // 7) for i in range(nseq)]
    c = make_unified_case(seqs, 8, 4, D, 16, torch.bfloat16, fp8=fp8, seed=D + nseq, kv_scale=0.5)
    run(oracle, c, torch.bfloat16, fp8=fp8)
    if nseq == 5:
        c = make_unified_case(seqs * 4 + [(37, 120)], 8, 4, D, 16, torch.bfloat16, fp8=fp8, seed=D, kv_scale=0.5)
        run(oracle, c, torch.bfloat16, fp8=fp8, window=90)


@pytest.mark.parametrize("D_heads", [(128, (8, 4)), (64, (16, 8)), (64, (8, 4)), (32, (8, 8)), (128, (16, 2)), (64, (32, 4))])
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("hp", [1, 2, 4])
@pytest.mark.parametrize("mode", ["plain", "alibi", "window", "fp8_out"])
@pytest.mark.parametrize("nseq", [5, 150])
def test_unified_decode_packed_kv_heads(oracle, D_heads, fp8, hp, mode, nseq):
    """Token-major caches: "uni_hp" adjacent kv heads are taken as one row of the decode kernel (their q heads are the
    columns of one Q^T operand, zero outside the own head's d range; 1 = never, 2 / 4 = that many where the shape allows,
    else fewer) -- split (5 sequences) and unsplit launches, every per-column feature, against the oracle."""
    from vllm_amd import _lib
    D, (H, KVH) = D_heads
    lens = [3, 64, 65, 300, 1000]
    seqs = [(1, lens[i % 5] + i // 5) for i in range(nseq)]
    c = make_unified_case(seqs, H, KVH, D, 16, torch.bfloat16, fp8=fp8, seed=D + H + nseq, kv_scale=0.5)
    slopes = 2.0 ** -(torch.arange(H).float() * 0.25 + 1) if mode == "alibi" else None
    _lib.set_tuning("uni_hp", hp)
    try:
        run(oracle, c, torch.bfloat16, fp8=fp8, alibi=slopes, window=90 if mode == "window" else None,
            out_scale=torch.tensor([0.02]) if mode == "fp8_out" else None)
    finally:
        _lib.set_tuning("uni_hp", 0)


@pytest.mark.parametrize("heads", [(32, 1), (48, 2), (20, 1), (17, 1)])
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("nseq", [4, 600])
def test_unified_decode_wide_groups(oracle, heads, fp8, nseq):
    """GQA groups of more than 16 q heads (MQA): 16 heads per workgroup, the last chunk partial;
    split and unsplit launches, ALiBi per head."""
    lens = [1, 33, 64, 130, 500]
    seqs = [(1, lens[i % len(lens)] + i // 5) for i in range(nseq)]
    c = make_unified_case(seqs, heads[0], heads[1], 64, 16, torch.float16, fp8=fp8, seed=heads[0] + nseq, kv_scale=0.5)
    slopes = 2.0 ** -(torch.arange(heads[0]).float() * 0.125 + 1)
    run(oracle, c, torch.float16, fp8=fp8, alibi=slopes)


@pytest.mark.parametrize("mode", ["window", "alibi", "softcap", "fp8_out", "hnd"])
@pytest.mark.parametrize("fp8", [False, True])
def test_unified_decode_kernel_features(oracle, mode, fp8):
    """Window edges inside / across steps and segments, ALiBi, softcap, fp8 output and a
    head-major (HND) cache allocation; unsplit launch (1040 workgroups; the split launches of
    these features are in test_unified_decode_batches)."""
    lens = [5, 100, 129, 300, 640, 1000]
    seqs = [(1, lens[i % len(lens)] + i // 6) for i in range(260)]
    c = make_unified_case(seqs, 8, 4, 64 if mode == "alibi" else 128, 16, torch.bfloat16, fp8=fp8, seed=77,
                          kv_scale=0.5)
    slopes = 2.0 ** -(torch.arange(8).float() * 0.25 + 1) if mode == "alibi" else None
    view = (lambda t: t.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)) if mode == "hnd" else None
    run(oracle, c, torch.bfloat16, fp8=fp8, alibi=slopes, window=100 if mode == "window" else None,
        softcap=15.0 if mode == "softcap" else 0.0,
        out_scale=torch.tensor([0.02]) if mode == "fp8_out" else None, view=view)


@pytest.mark.parametrize("mode", ["window", "alibi", "softcap", "fp8_out", "hnd", "f16", "mqa"])
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("nseq", [6, 300])
def test_unified_decode_kernel_wide_heads(oracle, mode, fp8, nseq):
    """The decode kernel's 256-wide instantiation (head sizes 136 .. 256: one workgroup per CU, the 16 q heads
    written in two passes): split (6 sequences) and unsplit launches, every feature, groups of 16 and 20 q heads."""
    lens = [5, 100, 129, 300, 640, 1000]
    seqs = [(1, lens[i % len(lens)] + i // 6) for i in range(nseq)]
    dtype = torch.float16 if mode == "f16" else torch.bfloat16
    H, KVH = (20, 1) if mode == "mqa" else (16, 2) if mode == "hnd" else (8, 4)
    c = make_unified_case(seqs, H, KVH, 192 if mode == "alibi" else 256, 16, dtype, fp8=fp8, seed=78 + nseq, kv_scale=0.5)
    slopes = 2.0 ** -(torch.arange(H).float() * 0.25 + 1) if mode == "alibi" else None
    view = (lambda t: t.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)) if mode == "hnd" else None
    run(oracle, c, dtype, fp8=fp8, alibi=slopes, window=100 if mode == "window" else None,
        softcap=15.0 if mode == "softcap" else 0.0,
        out_scale=torch.tensor([0.02]) if mode == "fp8_out" else None, view=view)


@pytest.mark.parametrize("heads", [(8, 2), (32, 8), (4, 4), (24, 1)])
@pytest.mark.parametrize("D", [128, 64])
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("mode", ["plain", "window", "alibi", "softcap"])
def test_unified_mixed_batch_two_kernels(oracle, heads, D, fp8, mode):
    """Mixed batches of >= 16 sequences: the decode kernel takes the sequences of query length
    1 (also a one-token prefill chunk), the tile kernel the rest."""
    seqs = [(1, 500), (70, 70), (1, 64), (33, 200), (1, 1), (9, 9), (1, 129), (150, 1100), (1, 31),
            (1, 1000), (2, 2), (1, 333), (1, 65), (17, 80), (1, 256), (1, 255), (1, 77), (5, 600)]
    c = make_unified_case(seqs, heads[0], heads[1], D, 16, torch.bfloat16, fp8=fp8, seed=heads[0] + D,
                          kv_scale=0.5)
    slopes = 2.0 ** -(torch.arange(heads[0]).float() * 0.25 + 1) if mode == "alibi" else None
    run(oracle, c, torch.bfloat16, fp8=fp8, alibi=slopes, window=40 if mode == "window" else None,
        softcap=20.0 if mode == "softcap" else 0.0)


def test_unified_decode_kernel_equals_packed(oracle):
    """Same decode batch through the decode kernel and the packed tile kernel ("uni_decode" = 0)."""
    from vllm_amd import _lib
    from vllm_amd.attention.ops import unified_attention
    c = make_unified_case([(1, 700), (1, 33), (1, 4000)], 16, 4, 128, 16, torch.bfloat16, fp8=True, seed=4,
                          kv_scale=0.5)
    outs = []
    sc = c["scale"].cuda()
    for knob in (1, 0):
        _lib.set_tuning("uni_decode", knob)
        out = torch.zeros(c["q"].shape, dtype=torch.bfloat16, device="cuda")
        unified_attention(c["q"].cuda(), c["k"].cuda(), c["v"].cuda(), out, c["cu_seqlens_q"].cuda(), 1,
                          c["seqused_k"].cuda(), 4000, 128 ** -0.5, True, (-1, -1),
                          c["block_table"].cuda(), 0.0, None, sc, sc)
        outs.append(out.cpu())
    _lib.set_tuning("uni_decode", 1)
    torch.testing.assert_close(outs[0].float(), outs[1].float(), atol=4e-3, rtol=4e-3)


def test_unified_decode_packed_equals_unpacked(oracle):
    """Same decode batch through the packed and the per-head launch ("pf_rows" = 1)."""
    from vllm_amd import _lib
    from vllm_amd.attention.ops import unified_attention
    c = make_unified_case([(1, 700), (1, 33)], 16, 4, 128, 32, torch.float16, seed=4)
    outs = []
    for knob in (0, 1):
        _lib.set_tuning("pf_rows", knob)
        out = torch.zeros(c["q"].shape, dtype=torch.float16, device="cuda")
        unified_attention(c["q"].cuda(), c["k"].cuda(), c["v"].cuda(), out, c["cu_seqlens_q"].cuda(), 1,
                          c["seqused_k"].cuda(), 700, 128 ** -0.5, True, (-1, -1),
                          c["block_table"].cuda(), 0.0, None, None, None)
        outs.append(out.cpu())
    _lib.set_tuning("pf_rows", 0)
    torch.testing.assert_close(outs[0].float(), outs[1].float(), atol=2e-3, rtol=2e-3)


def test_unified_fp8_output(oracle):
    c = make_unified_case([(130, 500), (1, 90)], 8, 2, 128, 16, torch.bfloat16, fp8=True, seed=9,
                          kv_scale=0.5)
    run(oracle, c, torch.bfloat16, fp8=True, out_scale=torch.tensor([0.02]))


def test_unified_strided_cache(oracle):
    """Caches as permuted views of an HND allocation (strides, not shapes, address the cache)."""
    c = make_unified_case([(70, 300), (3, 50)], 4, 2, 128, 16, torch.bfloat16, seed=11)
    view = lambda t: t.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
    run(oracle, c, torch.bfloat16, view=view)


def test_unified_rejects_what_it_does_not_cover():
    from vllm_amd.attention.ops import unified_attention
    c = make_unified_case([(4, 40)], 4, 4, 264, 16, torch.bfloat16)
    out = torch.zeros(c["q"].shape, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError, match="unsupported head size"):
        unified_attention(c["q"].cuda(), c["k"].cuda(), c["v"].cuda(), out, c["cu_seqlens_q"].cuda(),
                          4, c["seqused_k"].cuda(), 40, 0.1, True, (-1, -1), c["block_table"].cuda(),
                          0, None, None, None)


@pytest.mark.parametrize("fp8", [False, True])
def test_unified_mixed_batch_with_caller_segments_c_abi(oracle, fp8):
    """C-ABI level (include/vllm_amd.h): a mixed batch of >= 16 sequences with query lengths
    1..4 whose caller passes num_segments = 16 scratch.  The decode tokens are written by the
    decode kernel; the segment merge of the tile path must leave them alone (their segm_* rows
    are never written: poisoned with NaN here)."""
    from vllm_amd import _lib
    from vllm_amd._lib import check, lib
    from vllm_amd._torch_ops import _ptr, _stream
    dtype, H, KVH, D, BS = torch.bfloat16, 8, 2, 128, 16
    seq_lens = [(1 + (i % 4) if i % 3 else 1, 40 + 97 * i) for i in range(20)]
    c = make_unified_case(seq_lens, H, KVH, D, BS, dtype, fp8=fp8, seed=5, kv_scale=0.05 if fp8 else 1.0)
    sm = D ** -0.5
    kd = c["scale"] if fp8 else None
    ref = torch.zeros(c["q"].shape, dtype=dtype)
    oracle.unified_attention(c["q"], c["k"], c["v"], ref, c["cu_seqlens_q"], c["max_seqlen_q"],
                             c["seqused_k"], c["max_seqlen_k"], sm, True, (-1, -1), c["block_table"],
                             0.0, None, kd, kd, None, None)
    q, k, v = c["q"].cuda(), c["k"].cuda(), c["v"].cuda()
    cu, sk, bt = c["cu_seqlens_q"].cuda(), c["seqused_k"].cuda(), c["block_table"].cuda()
    kdd = None if kd is None else kd.cuda()
    out = torch.zeros_like(q)
    nseg = 16
    T = q.shape[0]
    segm_out = torch.full((T, H, nseg, D), float("nan"), dtype=torch.float32, device="cuda")
    segm_max = torch.full((T, H, nseg), float("nan"), dtype=torch.float32, device="cuda")
    segm_sum = torch.full((T, H, nseg), float("nan"), dtype=torch.float32, device="cuda")
    check(lib.vllm_amd_unified_attention(
        _ptr(out), _ptr(q), _ptr(k), _ptr(v), _ptr(cu), _ptr(sk), _ptr(bt), _ptr(kdd), _ptr(kdd),
        None, None, len(seq_lens), H, KVH, D, BS, c["max_seqlen_q"], bt.stride(0), q.stride(0),
        q.stride(1), out.stride(0), out.stride(1), k.stride(0), k.stride(1), k.stride(2),
        v.stride(0), v.stride(1), v.stride(2), float(sm), 0, 0.0, T, c["max_seqlen_k"],
        _ptr(segm_out), _ptr(segm_max), _ptr(segm_sum), nseg, _lib.BF16,
        _lib.KV_FP8_E4M3 if fp8 else _lib.KV_AUTO, _stream(q)))
    torch.cuda.synchronize()
    assert not out.isnan().any()
    torch.testing.assert_close(out.cpu().float(), ref.float(), atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("shape", [(64, 16, 8), (200, 4, 2), (256, 8, 8)])   # the last: XCD-grouped
def test_unified_decode_longest_first_order(oracle, fp8, shape):
    """Unsplit decode launches of 257..1024 workgroups hand the sequences out longest first
    (pa_lpt_wg, common.cuh): the output equals the oracle's and is bit-identical to the hardware
    order ("pa_order" = 0), with ties among the lengths and a mixed batch (non-decode sequences
    leave their workgroups at once)."""
    from vllm_amd import _lib
    from vllm_amd.attention.ops import unified_attention
    S, H, KVH = shape
    g = torch.Generator().manual_seed(S)
    kvl = torch.randint(1, 400, (S,), generator=g).tolist()
    kvl[2] = kvl[11] = kvl[S - 1] = 123
    for mixed in (False, True):
        sl = [(1, n) for n in kvl]
        if mixed:
            sl[4] = (3, kvl[4] + 3)
            sl[20] = (40, 40)
        c = make_unified_case(sl, H, KVH, 128, 16, torch.bfloat16, fp8=fp8, seed=S + mixed, kv_scale=0.5)
        run(oracle, c, torch.bfloat16, fp8=fp8)
        if not mixed:
            run(oracle, c, torch.bfloat16, fp8=fp8, window=100)   # keys clamp at the window
        kd = c["scale"].cuda() if fp8 else None
        outs = []
        for order in (1, 0):
            _lib.set_tuning("pa_order", order)
            out = torch.zeros(c["q"].shape, dtype=torch.bfloat16, device="cuda")
            unified_attention(c["q"].cuda(), c["k"].cuda(), c["v"].cuda(), out, c["cu_seqlens_q"].cuda(),
                              c["max_seqlen_q"], c["seqused_k"].cuda(), c["max_seqlen_k"], 128 ** -0.5,
                              True, (-1, -1), c["block_table"].cuda(), 0.0, None, kd, kd)
            outs.append(out)
        _lib.set_tuning("pa_order", 1)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1])


@pytest.fixture
def force_fa():
    """The 256-row kernel (prefill_fa.hip) for every launch that it supports ("pf_fa" = 2), not only long chunks."""
    from vllm_amd import _lib
    _lib.set_tuning("pf_fa", 2)
    yield
    _lib.set_tuning("pf_fa", 0)


@pytest.mark.parametrize("heads", [(8, 2), (32, 8), (4, 4), (6, 2), (14, 2), (16, 1), (24, 1)])
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("seq_lens", [[(700, 700), (300, 1500), (64, 64), (1, 1)],
                                      [(257, 1000), (130, 131), (1, 900), (5, 18), (63, 64), (65, 65)]])
def test_unified_fa_kernel(oracle, force_fa, heads, fp8, dtype, seq_lens):
    """The 256-row kernel on every GQA packing (groups that divide 256 rows, 3 and 7 that do not, 24 = unpacked
    rows): pure and chunked prefill, row blocks with one valid row, interior / diagonal / skipped tiles, stale NaN
    slots behind the sequences, both cache types."""
    c = make_unified_case(seq_lens, heads[0], heads[1], 128, 16, dtype, fp8=fp8, seed=heads[0] + len(seq_lens),
                          kv_scale=0.5)
    run(oracle, c, dtype, fp8=fp8)


@pytest.mark.parametrize("mode", ["fp8_out", "hnd", "bs64", "mixed16"])
@pytest.mark.parametrize("fp8", [False, True])
def test_unified_fa_kernel_features(oracle, force_fa, mode, fp8):
    """fp8 output, a head-major cache allocation, 64-token pages, and a batch of >= 16 sequences whose decode tokens
    go to the decode kernel while the 256-row kernel skips them."""
    seqs = [(200, 900), (1, 77), (33, 33), (400, 400)]
    if mode == "mixed16":
        seqs = seqs + [(1, 100 + 37 * i) for i in range(14)]
    c = make_unified_case(seqs, 8, 2, 128, 64 if mode == "bs64" else 16, torch.bfloat16, fp8=fp8, seed=11, kv_scale=0.75)
    view = (lambda t: t.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)) if mode == "hnd" else None
    run(oracle, c, torch.bfloat16, fp8=fp8, out_scale=torch.tensor([0.03]) if mode == "fp8_out" else None, view=view)


def test_unified_fa_kernel_sharp_rows(oracle, force_fa):
    """Rows whose maximum keeps growing along the keys (the lazy rescale must fire on some tiles and not on others):
    keys scaled up with their position."""
    c = make_unified_case([(300, 1200)], 8, 2, 128, 16, torch.bfloat16, seed=3)
    bt = c["block_table"][0]
    for t in range(1200):
        c["k"][int(bt[t // 16]), t % 16] *= (0.2 + 2.5 * t / 1200)
    run(oracle, c, torch.bfloat16)


@pytest.mark.parametrize("window", [1, 16, 64, 100, 257, 1000])
@pytest.mark.parametrize("heads", [(8, 2), (4, 4), (6, 2)])
@pytest.mark.parametrize("fp8", [False, True])
def test_unified_fa_kernel_window(oracle, force_fa, window, heads, fp8):
    """Sliding windows narrower than a key tile, than a row block's span, and wider: tiles left of the window skipped
    (not even staged for the workgroup's first rows), left-edge tiles masked, interior, diagonal."""
    c = make_unified_case([(700, 700), (300, 1500), (64, 64), (1, 900), (130, 131)], heads[0], heads[1], 128, 16,
                          torch.bfloat16, fp8=fp8, seed=window + heads[0], kv_scale=0.5)
    run(oracle, c, torch.bfloat16, fp8=fp8, window=window)


@pytest.mark.parametrize("D", [72, 80, 96, 112, 120, 136, 160, 192, 224, 256])
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_unified_fa_kernel_head_sizes(oracle, force_fa, D, fp8, dtype):
    """Head sizes 72 .. 120 run the 256-row kernel padded to 128 (zero columns from clamped addresses, the last
    sequence ending at the end of the cache allocation), 136 .. 256 its four-wave form padded to 256."""
    if fp8 and D % 16:
        pytest.skip("an fp8 cache needs head sizes that are multiples of 16")
    c = make_unified_case([(300, 700), (1, 90), (70, 70)], 8, 2, D, 16, dtype, fp8=fp8, seed=D, kv_scale=0.5)
    run(oracle, c, dtype, fp8=fp8)
    run(oracle, c, dtype, fp8=fp8, window=50)
    if D == 256:   # unpacked rows, a group that does not divide the 128 rows, fp8 output
        c = make_unified_case([(200, 333), (64, 64)], 6, 2, D, 32, dtype, fp8=fp8, seed=1, kv_scale=0.5)
        run(oracle, c, dtype, fp8=fp8)
        c = make_unified_case([(200, 333), (64, 64)], 4, 4, D, 16, dtype, fp8=fp8, seed=2, kv_scale=0.5)
        run(oracle, c, dtype, fp8=fp8, out_scale=torch.tensor([0.03]))


@pytest.mark.parametrize("soft_cap", [10.0, 50.0])
@pytest.mark.parametrize("D", [128, 256, 96])
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("window", [None, 100])
def test_unified_fa_kernel_softcap(oracle, force_fa, soft_cap, D, fp8, window):
    """Soft-capped scores (Gemma-2: 50 with head 256 and alternating sliding windows) in the 256-row kernel: interior and
    masked tiles (a masked score must stay -inf, not -softcap), both head widths, fp8 cache."""
    for dtype in (torch.bfloat16, torch.float16):
        c = make_unified_case([(300, 700), (129, 463), (1, 90), (70, 70)], 8, 2, D, 16, dtype, fp8=fp8,
                              seed=int(soft_cap) + D, kv_scale=0.5)
        run(oracle, c, dtype, fp8=fp8, softcap=soft_cap, window=window)

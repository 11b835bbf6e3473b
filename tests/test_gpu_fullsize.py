"""GPU parity at BASELINE.json's FULL sizes (configs[1], the per-GPU shapes of configs[3] and
configs[4]): the oracle cannot run the whole batch in seconds, so the checks are
  * the oracle on a SAMPLE of the batch's sequences (their pages gathered into a compact
    cache on the CPU -- same bytes, same order), to the tolerance of
    test_gpu_parity_attention.py (reference tests/kernels/attention/test_attention.py:345-348);
  * size-independent properties over the WHOLE batch: the three entry points (v1 / v2 /
    rocm -- different partitionings and merges of the same sum) agree; linearity in V
    (v_scale x 2 => out x 2, exact in binary floating point); permuting the sequences of the
    batch permutes the output rows bit for bit; and the cache write -> attention round trip
    (reshape_and_cache of a new token changes exactly the rows the slot mapping names).
"""
import pytest
import torch

from vllm_amd.attention import PagedAttention

pytestmark = pytest.mark.gpu

ATOL = 2e-3   # fp8 KV, as in test_gpu_parity_attention.py

# (name, num_seqs, q heads, kv heads, ctx, sampled sequences)
CASES = [
    ("configs1", 64, 32, 8, 4096, [0, 37, 63]),
    ("configs3_per_gpu", 256, 8, 1, 32768, [0, 255]),
    ("configs4", 8, 64, 8, 131072, [5]),
    ("head256", 64, 16, 8, 4096, [0, 41], 256),   # the reference's largest head size at the configs[1] batch
]
D, BS = 128, 16   # (D: the head size of the running case, set by test_fullsize_decode)


def build(S, H, KVH, ctx, seed, ragged=False):
    """fp8 e4m3 paged caches built on the device in 16 Mi-element chunks (no full-size fp32 copy)."""
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(seed)
    nb_seq = ctx // BS
    NB = S * nb_seq + 7
    kc = torch.empty((NB, KVH, D // 16, BS, 16), dtype=torch.uint8, device=dev)
    vc = torch.empty((NB, KVH, D, BS), dtype=torch.uint8, device=dev)
    kv_scale = 0.5
    step = max(1, (1 << 24) // (KVH * D * BS))        # blocks per chunk
    for b0 in range(0, NB, step):
        n = min(step, NB - b0)
        for cache in (kc, vc):
            x = (torch.rand((n,) + tuple(cache.shape[1:]), generator=g, device=dev) * 2 - 1) * D ** -0.5
            cache[b0:b0 + n] = (x / kv_scale).to(torch.float8_e4m3fn).view(torch.uint8)
    q = ((torch.rand((S, H, D), generator=g, device=dev) * 2 - 1) * D ** -0.5).to(torch.bfloat16)
    perm = torch.randperm(NB, generator=g, device=dev)[: S * nb_seq]
    bt = perm.view(S, nb_seq).to(torch.int32).contiguous()
    if ragged:
        sl = torch.randint(1, ctx + 1, (S,), generator=g, device=dev).to(torch.int32)
        sl[0] = ctx
    else:
        sl = torch.full((S,), ctx, dtype=torch.int32, device=dev)
    sc = torch.tensor([kv_scale], device=dev)
    return dict(q=q, kc=kc, vc=vc, bt=bt, sl=sl, sc=sc, kv_scale=kv_scale, ctx=ctx, KVH=KVH)


def decode(c, v_scale=None, q=None, bt=None, sl=None):
    return PagedAttention.forward_decode(
        c["q"] if q is None else q, c["kc"], c["vc"], c["bt"] if bt is None else bt,
        c["sl"] if sl is None else sl, c["ctx"], "fp8", c["KVH"], D ** -0.5, None, c["sc"],
        c["sc"] if v_scale is None else v_scale)


def decode_rocm(c):
    return PagedAttention.forward_decode_rocm(c["q"], c["kc"], c["vc"], c["bt"], c["sl"], c["ctx"],
                                              "fp8", c["KVH"], D ** -0.5, None, c["sc"], c["sc"])


def oracle_rows(oracle, c, seqs):
    """The oracle on the sequences `seqs`: their pages gathered (block-table order) into a
    compact CPU cache."""
    idx = c["bt"][seqs].long()                           # [n, nb]
    n, nb = idx.shape
    kc = c["kc"][idx.flatten()].cpu()
    vc = c["vc"][idx.flatten()].cpu()
    bt = torch.arange(n * nb, dtype=torch.int32).view(n, nb)
    q = c["q"][seqs].cpu()
    sl = c["sl"][seqs].cpu()
    out = torch.zeros_like(q)
    oracle.paged_attention_v1(out, q, kc, vc, c["KVH"], D ** -0.5, bt, sl, BS, c["ctx"], None,
                              "fp8", c["kv_scale"], c["kv_scale"])
    return out


@pytest.mark.parametrize("ragged", [False, True])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_fullsize_decode(ops, oracle, case, ragged):
    global D
    D = case[6] if len(case) > 6 else 128
    try:
        _fullsize_decode(ops, oracle, case[:6], ragged)
    finally:
        D = 128


def _fullsize_decode(ops, oracle, case, ragged):
    name, S, H, KVH, ctx, sample = case
    c = build(S, H, KVH, ctx, seed=len(name) + ragged, ragged=ragged)
    out = decode(c)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()

    # (1) oracle on a sample of sequences
    want = oracle_rows(oracle, c, sample)
    torch.testing.assert_close(out[sample].cpu().float(), want.float(), atol=ATOL, rtol=1e-5)

    # (2) the three entry points agree over the whole batch
    S_, P = S, (ctx + 511) // 512
    es = torch.empty((S_, H, P), dtype=torch.float32, device="cuda")
    ml = torch.empty_like(es)
    tmp = torch.empty((S_, H, P, D), dtype=torch.bfloat16, device="cuda")
    out_v2 = torch.empty_like(out)
    ops.paged_attention_v2(out_v2, es, ml, tmp, c["q"], c["kc"], c["vc"], KVH, D ** -0.5, c["bt"],
                           c["sl"], BS, ctx, None, "fp8", c["sc"], c["sc"])
    out_rocm = decode_rocm(c)
    out_v1 = torch.empty_like(out)
    ops.paged_attention_v1(out_v1, c["q"], c["kc"], c["vc"], KVH, D ** -0.5, c["bt"], c["sl"], BS,
                           ctx, None, "fp8", c["sc"], c["sc"])
    for other in (out_v1, out_v2, out_rocm):
        torch.testing.assert_close(other.float(), out.float(), atol=ATOL, rtol=1e-5)

    # (3) linearity in V: doubling v_scale doubles the output exactly
    out2 = decode(c, v_scale=c["sc"] * 2)
    assert torch.equal(out2.float(), out.float() * 2)

    # (4) permuting the batch permutes the rows bit for bit (every sequence is summed by one
    # workgroup in a fixed order)
    p = torch.randperm(S, device="cuda")
    outp = decode(c, q=c["q"][p].contiguous(), bt=c["bt"][p].contiguous(), sl=c["sl"][p].contiguous())
    assert torch.equal(outp, out[p])

    # (5) the balanced v1 form ("pa_balance" = 1) cuts a ragged batch where the order of the
    # sequences puts the cuts: same rows within the tolerance, in either order
    from vllm_amd import _lib
    _lib.set_tuning("pa_balance", 1)
    try:
        bal = decode(c)
        balp = decode(c, q=c["q"][p].contiguous(), bt=c["bt"][p].contiguous(), sl=c["sl"][p].contiguous())
    finally:
        _lib.set_tuning("pa_balance", 0)
    torch.testing.assert_close(bal.float(), out.float(), atol=ATOL, rtol=1e-5)
    torch.testing.assert_close(balp.float(), out[p].float(), atol=ATOL, rtol=1e-5)


def test_fullsize_cache_write_roundtrip(ops, oracle):
    """configs[1]: one decode step's cache write (64 new tokens, fp8) changes exactly the 64
    slots named, to the codes the oracle computes, and attention over the grown sequences
    matches the oracle on a sample."""
    name, S, H, KVH, ctx, sample = CASES[0]
    c = build(S, H, KVH, ctx, seed=99)
    c["sl"] = torch.full((S,), ctx - 5, dtype=torch.int32, device="cuda")
    g = torch.Generator().manual_seed(5)
    new_k = (torch.randn((S, KVH, D), generator=g) * 0.1).to(torch.bfloat16)
    new_v = (torch.randn((S, KVH, D), generator=g) * 0.1).to(torch.bfloat16)
    pos = ctx - 5                                       # position of the new token
    slots = (c["bt"][:, pos // BS].long() * BS + pos % BS).contiguous()
    k_before, v_before = c["kc"].clone(), c["vc"].clone()
    PagedAttention.write_to_paged_cache(new_k.cuda(), new_v.cuda(), c["kc"], c["vc"], slots, "fp8",
                                        c["sc"], c["sc"])
    c["sl"] += 1
    torch.cuda.synchronize()
    # exactly S * KVH * D bytes per cache may differ, all inside the named slots
    blk, off = slots // BS, slots % BS
    k_before[blk, :, :, off, :] = c["kc"][blk, :, :, off, :]
    v_before[blk, :, :, off] = c["vc"][blk, :, :, off]
    assert torch.equal(k_before, c["kc"]) and torch.equal(v_before, c["vc"])
    # the written codes are the oracle's (bit-exact)
    kc_o = torch.zeros((S, KVH, D // 16, BS, 16), dtype=torch.uint8)
    vc_o = torch.zeros((S, KVH, D, BS), dtype=torch.uint8)
    slots_o = torch.arange(S, dtype=torch.int64) * BS + pos % BS
    oracle.reshape_and_cache(new_k, new_v, kc_o, vc_o, slots_o, "fp8", c["kv_scale"], c["kv_scale"])
    assert torch.equal(c["kc"][blk, :, :, off, :].cpu(), kc_o[:, :, :, pos % BS, :])
    assert torch.equal(c["vc"][blk, :, :, off].cpu(), vc_o[:, :, :, pos % BS])
    out = decode(c)
    want = oracle_rows(oracle, c, sample)
    torch.testing.assert_close(out[sample].cpu().float(), want.float(), atol=ATOL, rtol=1e-5)

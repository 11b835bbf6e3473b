"""Shared input builders for the parity tests (mirrors the reference's
create_kv_caches_with_random, vllm/utils/__init__.py:1076-1127, and the shapes
of tests/kernels/attention/test_attention.py:124-348)."""
from __future__ import annotations

import torch


def x_of(dtype, kv_cache_dtype):
    return 16 if kv_cache_dtype != "auto" else 16 // torch.tensor([], dtype=dtype).element_size()


def make_kv_cache(num_blocks, block_size, num_kv_heads, head_size, dtype, kv_cache_dtype, seed,
                  kv_scale=1.0, poison_fp8_nan=False):
    """Returns CPU tensors (key_cache, value_cache) in the paged layout.
    fp8 caches are uint8 e4m3fn codes of U(-d^-.5, d^-.5) / kv_scale."""
    g = torch.Generator().manual_seed(seed)
    sc = head_size ** -0.5
    x = x_of(dtype, kv_cache_dtype)
    kshape = (num_blocks, num_kv_heads, head_size // x, block_size, x)
    vshape = (num_blocks, num_kv_heads, head_size, block_size)
    k = (torch.rand(kshape, generator=g) * 2 - 1) * sc
    v = (torch.rand(vshape, generator=g) * 2 - 1) * sc
    if kv_cache_dtype == "auto":
        return k.to(dtype), v.to(dtype)
    k8 = (k / kv_scale).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    v8 = (v / kv_scale).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    return k8.contiguous(), v8.contiguous()


def make_attn_case(num_seqs, num_heads, num_kv_heads, head_size, block_size, dtype, kv_cache_dtype,
                   seq_lens, seed=0, num_blocks=None, kv_scale=1.0, alibi=False,
                   distinct_blocks=True):
    g = torch.Generator().manual_seed(seed + 1000)
    max_len = max(seq_lens)
    mb = (max_len + block_size - 1) // block_size
    need = sum((l + block_size - 1) // block_size for l in seq_lens)
    if num_blocks is None:
        num_blocks = max(need + 3, 8)
    kc, vc = make_kv_cache(num_blocks, block_size, num_kv_heads, head_size, dtype, kv_cache_dtype,
                           seed, kv_scale)
    q = ((torch.rand((num_seqs, num_heads, head_size), generator=g) * 2 - 1) * head_size ** -0.5).to(dtype)
    if distinct_blocks:
        perm = torch.randperm(num_blocks, generator=g)
        bt = torch.zeros((num_seqs, mb), dtype=torch.int32)
        pos = 0
        for s, l in enumerate(seq_lens):
            nb = (l + block_size - 1) // block_size
            bt[s, :nb] = perm[pos:pos + nb].to(torch.int32)
            pos += nb
    else:
        bt = torch.randint(0, num_blocks, (num_seqs, mb), generator=g, dtype=torch.int32)
    sl = torch.tensor(seq_lens, dtype=torch.int32)
    slopes = torch.randn(num_heads, generator=g) if alibi else None
    return dict(q=q, kc=kc, vc=vc, bt=bt, sl=sl, slopes=slopes, max_len=max_len,
                num_kv_heads=num_kv_heads, block_size=block_size, scale=head_size ** -0.5,
                kv_cache_dtype=kv_cache_dtype, kv_scale=kv_scale, dtype=dtype)


def to_dev(t, dev="cuda"):
    return None if t is None else t.to(dev)


def codes_close(a, b, max_frac):
    """fp8 codes equal except for a small fraction that may sit one code apart (a value
    within fp32 rounding of a tie: the variance is summed in a different order)."""
    a, b = a.view(torch.uint8).flatten().int(), b.view(torch.uint8).flatten().int()
    diff = a != b
    # neighbouring codes of the same sign differ by 1; +-0 by 0x80
    adj = ((a - b).abs() == 1) | ((a & 0x7f) + (b & 0x7f) <= 1)
    assert bool((diff & ~adj).sum() == 0), "codes differ by more than one step"
    assert diff.float().mean().item() <= max_frac, f"{int(diff.sum())} of {diff.numel()} codes differ"


def make_prefill_case(query_lens, ctx_lens, H, KVH, D, BS, dtype, kv_cache_dtype="auto", seed=0,
                      kv_scale=1.0, amp=1.0):
    """Inputs of context_attention_fwd built like the reference test
    (tests/kernels/attention/test_prefix_prefill.py:66-153): per sequence ctx cached tokens in a
    paged cache (random distinct blocks) + query_len new tokens in dense k / v.  Returns CPU
    tensors plus `key_full`/`value_full` (per-sequence dense K/V as the kernel sees them, i.e.
    context dequantised) for torch reference checks."""
    g = torch.Generator().manual_seed(seed)
    B = len(query_lens)
    seq_lens = [a + b for a, b in zip(query_lens, ctx_lens)]
    T = sum(query_lens)
    q = ((torch.rand((T, H, D), generator=g) * 2 - 1) * amp).to(dtype)
    k = ((torch.rand((T, KVH, D), generator=g) * 2 - 1) * amp).to(dtype)
    v = ((torch.rand((T, KVH, D), generator=g) * 2 - 1) * amp).to(dtype)
    max_blocks = max((c + BS - 1) // BS for c in ctx_lens) + 2
    NB = B * max_blocks + 3
    perm = torch.randperm(NB, generator=g)
    b_loc = perm[: B * max_blocks].view(B, max_blocks).to(torch.int32)
    fp8 = kv_cache_dtype != "auto"
    x = 16 if fp8 else 16 // torch.empty((), dtype=dtype).element_size()   # the engine's 16-byte K vector
    cdt = torch.uint8 if fp8 else dtype
    # stale contents everywhere (incl. NaN patterns) so that masking of tokens >= ctx is tested
    if fp8:
        k_cache = torch.randint(0, 256, (NB, KVH, D // x, BS, x), generator=g, dtype=torch.uint8)
        v_cache = torch.randint(0, 256, (NB, KVH, D, BS), generator=g, dtype=torch.uint8)
    else:
        k_cache = torch.full((NB, KVH, D // x, BS, x), float("nan"), dtype=dtype)
        v_cache = torch.full((NB, KVH, D, BS), float("nan"), dtype=dtype)
    ctx_k, ctx_v = [], []
    for b in range(B):
        c = ctx_lens[b]
        ck = ((torch.rand((c, KVH, D), generator=g) * 2 - 1) * amp).to(dtype)
        cv = ((torch.rand((c, KVH, D), generator=g) * 2 - 1) * amp).to(dtype)
        if fp8:
            ck8 = (ck.float() / kv_scale).clamp(-448, 448).to(torch.float8_e4m3fn)
            cv8 = (cv.float() / kv_scale).clamp(-448, 448).to(torch.float8_e4m3fn)
            ck = (ck8.float() * kv_scale).to(dtype)
            cv = (cv8.float() * kv_scale).to(dtype)
            ckw, cvw = ck8.view(torch.uint8), cv8.view(torch.uint8)
        else:
            ckw, cvw = ck, cv
        for n in range(c):
            blk, off = int(b_loc[b, n // BS]), n % BS
            k_cache[blk, :, :, off, :] = ckw[n].view(KVH, D // x, x)
            v_cache[blk, :, :, off] = cvw[n]
        ctx_k.append(ck)
        ctx_v.append(cv)
    b_start_loc = torch.tensor([0] + list(torch.tensor(query_lens).cumsum(0)), dtype=torch.int32)
    b_seq_len = torch.tensor(seq_lens, dtype=torch.int32)
    key_full, value_full = [], []
    for b in range(B):
        s, e = int(b_start_loc[b]), int(b_start_loc[b + 1])
        key_full.append(torch.cat([ctx_k[b], k[s:e]], 0))
        value_full.append(torch.cat([ctx_v[b], v[s:e]], 0))
    return dict(q=q, k=k, v=v, k_cache=k_cache, v_cache=v_cache, b_loc=b_loc,
                b_start_loc=b_start_loc, b_seq_len=b_seq_len, max_input_len=max(query_lens),
                max_seq_len=max(seq_lens), key_full=key_full, value_full=value_full,
                query_lens=query_lens, ctx_lens=ctx_lens, scale=torch.tensor([kv_scale]))


def prefill_reference(c, sliding_window=0, alibi_slopes=None, sm_scale=None):
    """fp32 torch attention with the bottom-right causal (+ local) mask the reference test
    builds with xformers (test_prefix_prefill.py:195-212)."""
    q = c["q"].float()
    T, H, D = q.shape
    KVH = c["k"].shape[1]
    sm_scale = sm_scale or D ** -0.5
    out = torch.empty_like(q)
    for b, (ql, cl) in enumerate(zip(c["query_lens"], c["ctx_lens"])):
        s = int(c["b_start_loc"][b])
        kf = c["key_full"][b].float().repeat_interleave(H // KVH, dim=1)   # [L, H, D]
        vf = c["value_full"][b].float().repeat_interleave(H // KVH, dim=1)
        L = cl + ql
        sc = torch.einsum("mhd,nhd->hmn", q[s:s + ql], kf) * sm_scale
        qpos = torch.arange(cl, L, device=q.device).view(1, ql, 1)
        kpos = torch.arange(L, device=q.device).view(1, 1, L)
        mask = kpos <= qpos
        if alibi_slopes is not None:
            sc = sc + (kpos - qpos).float() * alibi_slopes.to(q.device).view(H, 1, 1)
        elif sliding_window > 0:
            mask = mask & (qpos - kpos < sliding_window)
        sc = sc.masked_fill(~mask, float("-inf"))
        out[s:s + ql] = torch.einsum("hmn,nhd->mhd", sc.softmax(-1), vf)
    return out


def make_unified_case(seq_lens, H, KVH, D, BS, dtype, fp8=False, seed=0, num_blocks=None,
                      kv_scale=1.0):
    """Inputs of unified_attention built like the reference test
    (tests/kernels/attention/test_triton_unified_attention.py:108-160): seq_lens = [(query_len,
    kv_len)], N(0,1) data, flash-layout caches [NB, BS, KVH, D].  Unlike the reference (random
    cache everywhere, block table sampled with replacement) the block table holds distinct
    blocks and every slot past a sequence's kv_len holds NaN / random bytes, so that reading
    past the sequence shows up."""
    g = torch.Generator().manual_seed(seed)
    B = len(seq_lens)
    qls, kvls = [a for a, _ in seq_lens], [b for _, b in seq_lens]
    q = torch.randn((sum(qls), H, D), generator=g).to(dtype)
    mb = max((n + BS - 1) // BS for n in kvls) + 1
    NB = num_blocks or (B * mb + 5)
    bt = torch.randperm(NB, generator=g)[: B * mb].view(B, mb).to(torch.int32)
    if fp8:
        kc = torch.randint(0, 256, (NB, BS, KVH, D), generator=g, dtype=torch.uint8)
        vc = torch.randint(0, 256, (NB, BS, KVH, D), generator=g, dtype=torch.uint8)
    else:
        kc = torch.full((NB, BS, KVH, D), float("nan"), dtype=dtype)
        vc = torch.full((NB, BS, KVH, D), float("nan"), dtype=dtype)
    for b in range(B):
        n = kvls[b]
        kk = torch.randn((n, KVH, D), generator=g)
        vv = torch.randn((n, KVH, D), generator=g)
        if fp8:
            kk = (kk / kv_scale).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
            vv = (vv / kv_scale).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
        else:
            kk, vv = kk.to(dtype), vv.to(dtype)
        for t in range(n):
            kc[int(bt[b, t // BS]), t % BS] = kk[t]
            vc[int(bt[b, t // BS]), t % BS] = vv[t]
    cu = torch.tensor([0] + list(torch.tensor(qls).cumsum(0)), dtype=torch.int32)
    return dict(q=q, k=kc, v=vc, block_table=bt, cu_seqlens_q=cu,
                seqused_k=torch.tensor(kvls, dtype=torch.int32), max_seqlen_q=max(qls),
                max_seqlen_k=max(kvls), scale=torch.tensor([kv_scale]))

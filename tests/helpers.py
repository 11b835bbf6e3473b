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

"""CPU tests (no GPU) of the host-side mirrors: PagedAttention helper (shapes, v1/v2
heuristic, scratch shapes), Fp8LinearOp plumbing, TP sharding, and a world_size-2 gloo
run of the N>1 path (KV-head-sharded attention + row-parallel GEMM + all-reduce) with the
oracle standing in for the kernels."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from helpers import make_attn_case


def test_kv_cache_shape_and_split():
    from vllm_amd.attention import PagedAttention
    assert PagedAttention.get_kv_cache_shape(10, 16, 8, 128) == (2, 10, 16 * 8 * 128)
    kv = torch.zeros(PagedAttention.get_kv_cache_shape(10, 16, 8, 128), dtype=torch.bfloat16)
    k, v = PagedAttention.split_kv_cache(kv, 8, 128)
    assert k.shape == (10, 8, 16, 16, 8) and v.shape == (10, 8, 128, 16)
    kv8 = torch.zeros(PagedAttention.get_kv_cache_shape(10, 16, 8, 128), dtype=torch.uint8)
    k, v = PagedAttention.split_kv_cache(kv8, 8, 128)
    assert k.shape == (10, 8, 8, 16, 16) and v.shape == (10, 8, 128, 16)
    assert 120 in PagedAttention.get_supported_head_sizes()


@pytest.mark.parametrize("S,H,max_len,want", [
    (64, 32, 4096, "v1"),      # configs[1]: S*H = 2048 > 512
    (8, 64, 131072, "v2"),     # configs[4]: > 8192
    (4, 32, 300, "v1"),        # one partition
    (4, 32, 2048, "v2"),       # S*H = 128 <= 512 and 4 partitions
    (256, 8, 32768, "v2"),     # configs[3]
])
def test_forward_decode_heuristic(monkeypatch, S, H, max_len, want):
    """vllm/attention/ops/paged_attn.py:117-193: same choice and scratch shapes."""
    from vllm_amd.attention import paged_attn
    calls = {}
    monkeypatch.setattr(paged_attn.ops, "paged_attention_v1",
                        lambda *a, **k: calls.setdefault("v1", a))
    monkeypatch.setattr(paged_attn.ops, "paged_attention_v2",
                        lambda *a, **k: calls.setdefault("v2", a))
    q = torch.zeros((S, H, 128), dtype=torch.bfloat16)
    kc = torch.zeros((4, 8, 16, 16, 8), dtype=torch.bfloat16)
    vc = torch.zeros((4, 8, 128, 16), dtype=torch.bfloat16)
    bt = torch.zeros((S, (max_len + 15) // 16), dtype=torch.int32)
    sl = torch.ones(S, dtype=torch.int32)
    one = torch.ones(1)
    out = paged_attn.PagedAttention.forward_decode(q, kc, vc, bt, sl, max_len, "auto", 8, 0.1,
                                                   None, one, one)
    assert out.shape == q.shape and list(calls) == [want]
    if want == "v2":
        P = (max_len + 511) // 512
        _, es, ml, tmp = calls["v2"][:4]
        assert es.shape == (S, H, P) and ml.shape == (S, H, P) and tmp.shape == (S, H, P, 128)
        assert es.dtype == torch.float32 and tmp.dtype == q.dtype


def test_shard_helpers():
    from vllm_amd.distributed import shard_kv_heads, shard_range
    assert shard_range(8192, 8, 3) == (3072, 4096)
    assert shard_kv_heads(64, 8, 8, 5) == ((40, 48), (5, 6))
    assert shard_kv_heads(32, 8, 2, 1) == ((16, 32), (4, 8))
    with pytest.raises(ValueError):
        shard_kv_heads(32, 8, 3, 0)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _tp_worker(rank, world, port, q):
    import torch.distributed as dist
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    from oracle import oracle as O
    from vllm_amd.distributed import shard_kv_heads, shard_range, tensor_model_parallel_all_reduce
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # ---- attention: every rank owns KVH/world kv heads, no collective inside ----
        H, KVH, D, BS = 8, 2, 64, 16
        c = make_attn_case(3, H, KVH, D, BS, torch.bfloat16, "fp8", [5, 200, 530], seed=1, kv_scale=0.5)
        (h0, h1), (k0, k1) = shard_kv_heads(H, KVH, world, rank)
        qs = c["q"][:, h0:h1].contiguous()
        kcs, vcs = c["kc"][:, k0:k1].contiguous(), c["vc"][:, k0:k1].contiguous()
        out_s = torch.empty_like(qs)
        O.paged_attention_v1(out_s, qs, kcs, vcs, k1 - k0, c["scale"], c["bt"], c["sl"], BS,
                             c["max_len"], None, "fp8", 0.5, 0.5)
        gathered = [torch.empty_like(out_s) for _ in range(world)]
        dist.all_gather(gathered, out_s)
        full = torch.empty_like(c["q"])
        O.paged_attention_v1(full, c["q"], c["kc"], c["vc"], KVH, c["scale"], c["bt"], c["sl"], BS,
                             c["max_len"], None, "fp8", 0.5, 0.5)
        ok_attn = torch.equal(torch.cat(gathered, dim=1), full)   # sharding is exact
        # ---- row-parallel fp8 linear: quantise the K-shard per token on this rank, GEMM,
        #      then ONE sum all-reduce (linear.py:1385-1389) ----
        g = torch.Generator().manual_seed(7)
        M, K, N = 5, 256, 48
        x = torch.randn((M, K), generator=g).to(torch.bfloat16)
        w = torch.randn((N, K), generator=g).to(torch.bfloat16)
        ka, kb = shard_range(K, world, rank)
        xq, xs = O.scaled_fp8_quant(x[:, ka:kb].contiguous(), use_per_token_if_dynamic=True)
        wq, ws = O.scaled_fp8_quant(w[:, ka:kb].contiguous(), use_per_token_if_dynamic=True)
        part = O.scaled_mm(xq, wq.t(), xs, ws, torch.bfloat16).float()
        total = tensor_model_parallel_all_reduce(part, dist)
        ref = x.float() @ w.float().t()
        err = (total - ref).abs().max().item() / ref.abs().max().item()
        if rank == 0:
            q.put((ok_attn, err, part.data_ptr() != total.data_ptr()))
    finally:
        dist.destroy_process_group()


def test_tp2_gloo_sharded_attention_and_row_parallel():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok_attn, err, out_of_place = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok_attn, "KV-head-sharded attention must equal the unsharded result exactly"
    assert err < 0.06, err            # two fp8 quantisations (~2^-4 relative each, averaged over K)
    assert out_of_place

#!/usr/bin/env python3
"""Decode paged attention over a batch x context grid through PagedAttention.forward_decode
(the reference's v1 / v2 choice), HIP-graph timed: us, GB/s of algorithmic bytes, % of 8 TB/s.
usage: python tools/bench_decode_grid.py [--kv fp8] [--heads 32,8] [--ragged]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd.attention import PagedAttention
from vllm_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--kv", default="fp8")
ap.add_argument("--heads", default="32,8")
ap.add_argument("--batches", default="1,4,16,64,256")
ap.add_argument("--ctxs", default="1024,4096,16384,65536")
ap.add_argument("--bs", type=int, default=16)
ap.add_argument("--sp", default="0", help="comma list of pa_sp tuning values (0 = heuristic)")
ap.add_argument("--op", default="auto", help="auto (forward_decode) | rocm (paged_attention_rocm)")
ap.add_argument("--ragged", action="store_true", help="sequence lengths uniform in [ctx/8, ctx]")
ap.add_argument("--order", default="1", help="comma list of pa_order tuning values (1 = default)")
a = ap.parse_args()
H, KVH = (int(x) for x in a.heads.split(","))
D, BS = 128, a.bs
dev = torch.device("cuda", 0)
fp8 = a.kv == "fp8"
for S in (int(x) for x in a.batches.split(",")):
    for ctx in (int(x) for x in a.ctxs.split(",")):
        if S * ctx * KVH * D * 2 * (1 if fp8 else 2) > 24e9:
            continue
        q, caches, bt, sl, one = bench.make_attention_inputs(S, ctx, H, KVH, D, BS, fp8, dev, n_caches=1)
        if a.ragged:
            g = torch.Generator(device=dev).manual_seed(1)
            sl = torch.randint(ctx // 8, ctx + 1, (S,), generator=g, device=dev, dtype=torch.int32)
        kc, vc = caches[0]
        if a.op == "rocm":
            fn = lambda: PagedAttention.forward_decode_rocm(q, kc, vc, bt, sl, ctx, a.kv, KVH, D ** -0.5, None, one, one)
        else:
            fn = lambda: PagedAttention.forward_decode(q, kc, vc, bt, sl, ctx, a.kv, KVH, D ** -0.5, None, one, one)
        es = 1 if fp8 else 2
        toks = int(sl.sum().item())
        nbytes = 2 * toks * KVH * D * es + 2 * 2 * S * H * D
        for sp, order in ((int(x), int(y)) for x in a.sp.split(",") for y in a.order.split(",")):
            _lib.set_tuning("pa_sp", sp)
            _lib.set_tuning("pa_order", order)
            t = bench.time_gpu_graph(fn, 10)
            print(f"S={S:4d} ctx={ctx:6d} {a.kv} H={H}/{KVH} bs={BS} {a.op} sp={sp} order={order}{' ragged' if a.ragged else ''}: "
                  f"{t * 1e6:9.1f} us {nbytes / t / 1e9:8.1f} GB/s ({nbytes / t / 8e12 * 100:5.1f}% of HBM peak)",
                  flush=True)
        _lib.set_tuning("pa_sp", 0)
        _lib.set_tuning("pa_order", 1)
        del q, caches, kc, vc

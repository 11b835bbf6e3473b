#!/usr/bin/env python3
"""A mixed V0 step through context_attention_fwd(skip_decode=True): one prefill chunk among
one-token sequences (which the kernel skips): cost of the launch grid (x-layout fp8 cache,
32q/8kv, head 128)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd.attention.ops import context_attention_fwd
dev = torch.device("cuda", 0)
H, KVH, D, BS = 32, 8, 128, 16
for chunk, cctx, ndec in ((2048, 2048, 0), (2048, 2048, 63), (2048, 2048, 255), (512, 0, 255)):
    qls = [chunk] + [1] * ndec
    ctxs = [cctx] + [1024] * ndec
    B, T = len(qls), sum(qls)
    q = (torch.rand((T, H, D), device=dev) - 0.5).to(torch.bfloat16)
    k = (torch.rand((T, KVH, D), device=dev) - 0.5).to(torch.bfloat16)
    v = (torch.rand((T, KVH, D), device=dev) - 0.5).to(torch.bfloat16)
    o = torch.empty_like(q)
    nb = max(ctxs) // BS + 1
    kc = torch.randint(0, 120, (B * nb, KVH, D // 16, BS, 16), device=dev, dtype=torch.uint8)
    vc = torch.randint(0, 120, (B * nb, KVH, D, BS), device=dev, dtype=torch.uint8)
    b_loc = torch.randperm(B * nb, device=dev).to(torch.int32).view(B, nb)
    start = torch.tensor([0] + list(torch.tensor(qls).cumsum(0)), device=dev, dtype=torch.int32)
    seql = torch.tensor([a + b for a, b in zip(qls, ctxs)], dtype=torch.int32, device=dev)
    one = torch.full((1,), 0.02, device=dev)
    msl = max(a + b for a, b in zip(qls, ctxs))
    fn = lambda: context_attention_fwd(q, k, v, o, "fp8", kc, vc, b_loc, start, seql, msl, max(qls), one, one,
                                       skip_decode=True)
    t = bench.time_gpu_graph(fn, 5)
    print(f"chunk {chunk} (+{cctx} cached) + {ndec} skipped one-token sequences: {t * 1e6:9.1f} us", flush=True)

#!/usr/bin/env python3
"""Eager decode-only unified_attention launches for rocprofv3 (flash-layout fp8 cache, 32q/8kv,
head 128): usage prof_unified_decode.py [S ctx]."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_amd.attention.ops import unified_attention
dev = torch.device("cuda", 0)
H, KVH, D, BS = 32, 8, 128, 16
cases = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(64, 4096), (256, 4096), (1, 4096)]
for S, ctx in cases:
    nb = (ctx + BS - 1) // BS
    q = (torch.rand((S, H, D), device=dev) - 0.5).to(torch.bfloat16)
    o = torch.empty_like(q)
    kf = torch.randint(0, 120, (S * nb, BS, KVH, D), device=dev, dtype=torch.uint8)
    vf = torch.randint(0, 120, (S * nb, BS, KVH, D), device=dev, dtype=torch.uint8)
    bt = torch.randperm(S * nb, device=dev).to(torch.int32).view(S, nb)
    cu = torch.arange(S + 1, device=dev, dtype=torch.int32)
    sl = torch.full((S,), ctx, dtype=torch.int32, device=dev)
    sc = torch.full((1,), 0.02, device=dev)
    for _ in range(10):
        unified_attention(q, kf, vf, o, cu, 1, sl, ctx, D ** -0.5, True, (-1, -1), bt, 0.0, None, sc, sc)
    torch.cuda.synchronize()
    del kf, vf

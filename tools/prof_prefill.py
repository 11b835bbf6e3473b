#!/usr/bin/env python3
"""Eager context_attention_fwd launches for rocprofv3 (B=4, q_len 2048, ctx 2048, fp8 cache)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_amd.attention.ops import context_attention_fwd
dev = torch.device("cuda", 0)
H, KVH, D, BS, B, ql, cl = 32, 8, 128, 16, 4, 2048, 2048
T = B * ql
q = (torch.rand((T, H, D), device=dev) - 0.5).to(torch.bfloat16)
k = (torch.rand((T, KVH, D), device=dev) - 0.5).to(torch.bfloat16)
v = (torch.rand((T, KVH, D), device=dev) - 0.5).to(torch.bfloat16)
o = torch.empty_like(q)
nb = cl // BS + 1
kc = torch.randint(0, 120, (B * nb, KVH, D // 16, BS, 16), device=dev, dtype=torch.uint8)
vc = torch.randint(0, 120, (B * nb, KVH, D, BS), device=dev, dtype=torch.uint8)
b_loc = torch.randperm(B * nb, device=dev).to(torch.int32).view(B, nb)
start = (torch.arange(B + 1, device=dev) * ql).to(torch.int32)
seql = torch.full((B,), ql + cl, dtype=torch.int32, device=dev)
one = torch.full((1,), 0.02, device=dev)
for _ in range(5):
    context_attention_fwd(q, k, v, o, "fp8", kc, vc, b_loc, start, seql, ql + cl, ql, one, one)
torch.cuda.synchronize()

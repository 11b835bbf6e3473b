#!/usr/bin/env python3
"""Eager unified_attention launches on one prefill shape, for rocprofv3 passes:
   SHAPE=1x8192x0 KV=fp8 SET=pf_fa_stagger=0 rocprofv3 --pmc ... -d OUT -- python3 tools/prof_prefill.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_amd import _lib
from vllm_amd.attention.ops import unified_attention

dev = torch.device("cuda", 0)
for kv_ in filter(None, os.environ.get("SET", "").split(",")):
    _lib.set_tuning(kv_.split("=")[0], int(kv_.split("=")[1]))
B, ql, cl = (int(x) for x in os.environ.get("SHAPE", "1x8192x0").split("x"))
fp8 = os.environ.get("KV", "fp8") == "fp8"
H, KVH, D, BS = 32, 8, 128, 16
T = B * ql
g = torch.Generator(device=dev).manual_seed(0)
q = (torch.rand((T, H, D), device=dev, generator=g) - 0.5).to(torch.bfloat16)
o = torch.empty_like(q)
nb = (ql + cl + BS - 1) // BS
if fp8:
    kf = torch.randint(0, 120, (B * nb, KVH, BS, D), device=dev, dtype=torch.uint8).permute(0, 2, 1, 3)
    vf = torch.randint(0, 120, (B * nb, KVH, BS, D), device=dev, dtype=torch.uint8).permute(0, 2, 1, 3)
else:
    kf = (torch.rand((B * nb, KVH, BS, D), device=dev) - 0.5).to(torch.bfloat16).permute(0, 2, 1, 3)
    vf = (torch.rand((B * nb, KVH, BS, D), device=dev) - 0.5).to(torch.bfloat16).permute(0, 2, 1, 3)
bt = torch.randperm(B * nb, device=dev).to(torch.int32).view(B, nb)
start = (torch.arange(B + 1, device=dev) * ql).to(torch.int32)
seql = torch.full((B,), ql + cl, dtype=torch.int32, device=dev)
one = torch.full((1,), 0.02, device=dev)
for _ in range(int(os.environ.get("LAUNCHES", "20"))):
    unified_attention(q, kf, vf, o, start, ql, seql, ql + cl, D ** -0.5, True, (-1, -1), bt, 0.0, None,
                      one if fp8 else None, one if fp8 else None)
torch.cuda.synchronize()
print(B, ql, cl, "done", flush=True)

#!/usr/bin/env python3
"""Decode-only batches through unified_attention (flash-layout fp8 cache): the flash-layout
decode kernel (unified_decode.hip) vs the GQA-packed tile kernel ("uni_decode" = 0) vs one
workgroup per q head ("pf_rows" = 1)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd import _lib
from vllm_amd.attention.ops import unified_attention
dev = torch.device("cuda", 0)
H, KVH, D, BS = 32, 8, 128, 16
for S, ctx in ((64, 4096), (256, 4096), (16, 16384), (4, 8192), (1, 4096)):
    nb = (ctx + BS - 1) // BS
    q = (torch.rand((S, H, D), device=dev) - 0.5).to(torch.bfloat16)
    o = torch.empty_like(q)
    kf = torch.randint(0, 120, (S * nb, BS, KVH, D), device=dev, dtype=torch.uint8)
    vf = torch.randint(0, 120, (S * nb, BS, KVH, D), device=dev, dtype=torch.uint8)
    bt = torch.randperm(S * nb, device=dev).to(torch.int32).view(S, nb)
    cu = torch.arange(S + 1, device=dev, dtype=torch.int32)
    sl = torch.full((S,), ctx, dtype=torch.int32, device=dev)
    sc = torch.full((1,), 0.02, device=dev)
    fn = lambda: unified_attention(q, kf, vf, o, cu, 1, sl, ctx, D ** -0.5, True, (-1, -1), bt, 0.0, None, sc, sc)
    nbytes = 2 * S * ctx * KVH * D
    for ud, knob, name in ((1, 0, "decode"), (2, 0, "exp"), (0, 0, "packed")):
        _lib.set_tuning("uni_decode", ud)
        _lib.set_tuning("pf_rows", knob)
        t = bench.time_gpu_graph(fn, 5)
        print(f"unified decode S={S:4d} ctx={ctx:6d} {name:9s}: {t * 1e6:9.1f} us {nbytes / t / 1e9:8.0f} GB/s", flush=True)
    _lib.set_tuning("pf_rows", 0)
    _lib.set_tuning("uni_decode", 1)

#!/usr/bin/env python3
"""Decode-only unified_attention at a given head size: python tools/bench_uni_decode_d.py [D] [KVH] [nhd] (bytes = K and V
once; nhd: token-major cache memory -- a (token, head) row is D bytes, rows of one head KVH * D bytes apart)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd.attention.ops import unified_attention
dev = torch.device("cuda", 0)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 256
KVH = int(sys.argv[2]) if len(sys.argv) > 2 else 8
NHD = len(sys.argv) > 3 and sys.argv[3] == "nhd"
if len(sys.argv) > 4:   # tuning "uni_hp": kv heads packed per workgroup row (1 = never)
    from vllm_amd import _lib
    _lib.set_tuning("uni_hp", int(sys.argv[4]))
H, BS = 2 * KVH, 16
for S, ctx in ((16, 4096), (64, 4096), (128, 4096), (256, 4096)):
    nb = (ctx + BS - 1) // BS
    q = (torch.rand((S, H, D), device=dev) - 0.5).to(torch.bfloat16)
    o = torch.empty_like(q)
    if NHD:
        kf = torch.randint(0, 120, (S * nb, BS, KVH, D), device=dev, dtype=torch.uint8)
        vf = torch.randint(0, 120, (S * nb, BS, KVH, D), device=dev, dtype=torch.uint8)
    else:
        kf = torch.randint(0, 120, (S * nb, KVH, BS, D), device=dev, dtype=torch.uint8).permute(0, 2, 1, 3)
        vf = torch.randint(0, 120, (S * nb, KVH, BS, D), device=dev, dtype=torch.uint8).permute(0, 2, 1, 3)
    bt = torch.randperm(S * nb, device=dev).to(torch.int32).view(S, nb)
    cu = torch.arange(S + 1, device=dev, dtype=torch.int32)
    sl = torch.full((S,), ctx, dtype=torch.int32, device=dev)
    sc = torch.full((1,), 0.02, device=dev)
    fn = lambda: unified_attention(q, kf, vf, o, cu, 1, sl, ctx, D ** -0.5, True, (-1, -1), bt, 0.0, None, sc, sc)
    t = bench.time_gpu_graph(fn, 5)
    byt = 2.0 * S * ctx * KVH * D
    print(f"D={D} KVH={KVH} {'nhd' if NHD else 'hnd'} S={S} ctx={ctx}: {t*1e6:9.1f} us  {byt/t/1e9:8.1f} GB/s ({100*byt/t/8e12:.1f} % of 8 TB/s)", flush=True)

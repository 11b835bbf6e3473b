#!/usr/bin/env python3
"""A mixed V1 step through unified_attention (flash-layout fp8 cache, 32q/8kv, head 128): one
prefill chunk among decode tokens; decode kernel + tile kernel vs the tile kernel alone
("uni_decode" = 0)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd import _lib
from vllm_amd.attention.ops import unified_attention
dev = torch.device("cuda", 0)
H, KVH, D, BS = 32, 8, 128, 16
for chunk, cctx, ndec, dctx in ((2048, 2048, 63, 4096), (512, 0, 127, 4096), (8192, 0, 32, 8192), (256, 4096, 255, 2048)):
    qls = [chunk] + [1] * ndec
    kvl = [chunk + cctx] + [dctx] * ndec
    S = len(qls)
    nb = (max(kvl) + BS - 1) // BS
    T = sum(qls)
    q = (torch.rand((T, H, D), device=dev) - 0.5).to(torch.bfloat16)
    o = torch.empty_like(q)
    kf = torch.randint(0, 120, (S * nb, BS, KVH, D), device=dev, dtype=torch.uint8)
    vf = torch.randint(0, 120, (S * nb, BS, KVH, D), device=dev, dtype=torch.uint8)
    bt = torch.randperm(S * nb, device=dev).to(torch.int32).view(S, nb)
    cu = torch.tensor([0] + list(torch.tensor(qls).cumsum(0)), device=dev, dtype=torch.int32)
    sl = torch.tensor(kvl, dtype=torch.int32, device=dev)
    sc = torch.full((1,), 0.02, device=dev)
    fn = lambda: unified_attention(q, kf, vf, o, cu, max(qls), sl, max(kvl), D ** -0.5, True, (-1, -1), bt, 0.0,
                                   None, sc, sc)
    for ud, name in ((1, "decode + tile"), (0, "tile only")):
        _lib.set_tuning("uni_decode", ud)
        t = bench.time_gpu_graph(fn, 5)
        print(f"chunk {chunk} (+{cctx} cached) + {ndec} decodes @ {dctx}: {name:14s} {t * 1e6:9.1f} us", flush=True)
    _lib.set_tuning("uni_decode", 1)
    del kf, vf

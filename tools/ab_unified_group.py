import os, sys
import torch
sys.path.insert(0, "/root/repo")
import bench
from vllm_amd import _lib
from vllm_amd.attention.ops import unified_attention
dev = torch.device("cuda", 0)
H, KVH, D, BS = 32, 8, 128, 16
for S, ctx in ((64, 4096), (128, 4096)):
    nb = (ctx + BS - 1) // BS
    q = (torch.rand((S, H, D), device=dev) - 0.5).to(torch.bfloat16)
    o = torch.empty_like(q)
    kf = torch.randint(0, 120, (S * nb, BS, KVH, D), device=dev, dtype=torch.uint8)
    vf = torch.randint(0, 120, (S * nb, BS, KVH, D), device=dev, dtype=torch.uint8)
    bt = torch.randperm(S * nb, device=dev).to(torch.int32).view(S, nb)
    cu = torch.arange(S + 1, device=dev, dtype=torch.int32)
    sl = torch.full((S,), ctx, dtype=torch.int32, device=dev)
    sc = torch.full((1,), 0.02, device=dev)
    nbytes = 2 * S * ctx * KVH * D
    for layout in ("nhd", "hnd"):
        k_, v_ = kf, vf
        if layout == "hnd":
            k_ = kf.view(S * nb, KVH, BS, D).permute(0, 2, 1, 3)
            v_ = vf.view(S * nb, KVH, BS, D).permute(0, 2, 1, 3)
        fn = lambda: unified_attention(q, k_, v_, o, cu, 1, sl, ctx, D ** -0.5, True, (-1, -1), bt, 0.0, None, sc, sc)
        for ud in (1, 2, 3, 1, 2, 3):
            _lib.set_tuning("uni_decode", ud)
            t = bench.time_gpu_graph(fn, 10)
            print(f"S={S} {layout} uni_decode={ud} ({'heuristic' if ud == 1 else 'grouped' if ud == 2 else 'not grouped'}): {t * 1e6:7.1f} us {nbytes / t / 1e12:5.2f} TB/s", flush=True)
_lib.set_tuning("uni_decode", 1)

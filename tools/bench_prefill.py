#!/usr/bin/env python3
"""context_attention_fwd / unified_attention timing:
python tools/bench_prefill.py [--kv fp8|auto] [--unified]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd.attention.ops import context_attention_fwd, unified_attention

ap = argparse.ArgumentParser()
ap.add_argument("--kv", default="fp8")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--unified", action="store_true", help="unified_attention over a flash-layout cache")
ap.add_argument("--rows", type=int, default=0, help="query rows per workgroup (tuning pf_rows)")
ap.add_argument("--hnd", action="store_true", help="unified: head-major cache memory (permuted view)")
ap.add_argument("--shapes", default="4x2048x0,4x2048x2048,1x8192x0,16x512x3584,64x128x4096", help="BxQxCTX,...")
ap.add_argument("--softcap", type=float, default=0.0, help="unified: soft cap of the scores")
ap.add_argument("--window", type=int, default=0, help="unified: sliding window (keys)")
ap.add_argument("--D", type=int, default=128, help="head size")
ap.add_argument("--set", default="", help="tuning keys: k=v[,k=v]")
ap.add_argument("--fa", type=int, default=0, help="tuning pf_fa: 1 = never the 256-row kernel, 2 = always")
a = ap.parse_args()
if a.fa:
    from vllm_amd import _lib
    _lib.set_tuning("pf_fa", a.fa)
for kv_ in filter(None, a.set.split(",")):
    from vllm_amd import _lib
    _lib.set_tuning(kv_.split("=")[0], int(kv_.split("=")[1]))
dev = torch.device("cuda", 0)
if a.rows:
    from vllm_amd import _lib
    _lib.set_tuning("pf_rows", a.rows)
H, KVH, D, BS = 32, 8, a.D, 16
for (B, ql, cl) in [tuple(int(x) for x in sh.split('x')) for sh in a.shapes.split(',')]:
    T = B * ql
    g = torch.Generator(device=dev).manual_seed(0)
    q = (torch.rand((T, H, D), device=dev, generator=g) - 0.5).to(torch.bfloat16)
    k = (torch.rand((T, KVH, D), device=dev, generator=g) - 0.5).to(torch.bfloat16)
    v = (torch.rand((T, KVH, D), device=dev, generator=g) - 0.5).to(torch.bfloat16)
    o = torch.empty_like(q)
    nb_seq = (cl + BS - 1) // BS + 1
    NB = B * nb_seq
    fp8 = a.kv == "fp8"
    x = 16 if fp8 else 8
    if fp8:
        kc = torch.randint(0, 120, (NB, KVH, D // x, BS, x), device=dev, dtype=torch.uint8)
        vc = torch.randint(0, 120, (NB, KVH, D, BS), device=dev, dtype=torch.uint8)
    else:
        kc = (torch.rand((NB, KVH, D // x, BS, x), device=dev) - 0.5).to(torch.bfloat16)
        vc = (torch.rand((NB, KVH, D, BS), device=dev) - 0.5).to(torch.bfloat16)
    b_loc = torch.randperm(NB, device=dev).to(torch.int32).view(B, nb_seq)
    start = (torch.arange(B + 1, device=dev) * ql).to(torch.int32)
    seql = torch.full((B,), ql + cl, dtype=torch.int32, device=dev)
    one = torch.full((1,), 0.02, device=dev)
    fn = lambda: context_attention_fwd(q, k, v, o, a.kv, kc, vc, b_loc, start, seql, ql + cl, ql, one, one)
    if a.unified:
        nb_u = (ql + cl + BS - 1) // BS
        if fp8:
            kf = torch.randint(0, 120, (B * nb_u, BS, KVH, D), device=dev, dtype=torch.uint8)
            vf = torch.randint(0, 120, (B * nb_u, BS, KVH, D), device=dev, dtype=torch.uint8)
        else:
            kf = (torch.rand((B * nb_u, BS, KVH, D), device=dev) - 0.5).to(torch.bfloat16)
            vf = (torch.rand((B * nb_u, BS, KVH, D), device=dev) - 0.5).to(torch.bfloat16)
        if a.hnd:   # head-major memory behind the NHD shape
            kf = kf.view(B * nb_u, KVH, BS, D).permute(0, 2, 1, 3)
            vf = vf.view(B * nb_u, KVH, BS, D).permute(0, 2, 1, 3)
        bt = torch.randperm(B * nb_u, device=dev).to(torch.int32).view(B, nb_u)
        fn = lambda: unified_attention(q, kf, vf, o, start, ql, seql, ql + cl, D ** -0.5, True, ((a.window - 1, 0) if a.window else (-1, -1)),
                                       bt, a.softcap, None, one if fp8 else None, one if fp8 else None)
    t = bench.time_gpu_graph(fn, a.iters)
    flops = 4.0 * D * H * B * ql * (cl + (ql + 1) / 2)
    print(f"{'unified ' if a.unified else ''}B={B:3d} q_len={ql:5d} ctx={cl:5d} kv={a.kv}: {t*1e6:9.1f} us  {flops/t/1e12:7.1f} TFLOP/s "
          f"({100*flops/t/2.5e15:.1f}% of 2.5 PF bf16)", flush=True)

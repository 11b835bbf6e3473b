#!/usr/bin/env python3
"""paged_attention (V0 decode, x-layout cache) at other head sizes: python tools/bench_pa_d.py [D ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
for D in [int(x) for x in sys.argv[1:]] or [256]:
    for (S, ctx, H, KVH) in ((64, 4096, 16, 8), (256, 4096, 16, 8), (16, 16384, 16, 8), (64, 4096, 8, 1)):
        r = bench.run_attention_shape(f"D{D}", S, ctx, H, KVH, dev, D=D)
        print(f"D={D} S={S} ctx={ctx} {H}q/{KVH}kv: {r['us']:8.1f} us  {r['frac_of_hbm_peak'] * 100:5.1f} % of 8 TB/s", flush=True)

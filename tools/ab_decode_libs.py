#!/usr/bin/env python3
"""Decode shapes with whatever library VLLM_AMD_LIB names (A/B of two builds = two runs of this
script on one box): median of 7 x 20 launches, HIP events, two alternating caches."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd.attention import PagedAttention
dev = torch.device("cuda", 0)
for name, S, ctx, fp8 in (("C2", 64, 4096, True), ("C2h", 64, 2048, True), ("C2d", 64, 8192, True),
                          ("C2bf16", 64, 4096, False), ("C2x4", 256, 4096, True), ("S32", 32, 4096, True)):
    q, caches, bt, sl, one = bench.make_attention_inputs(S, ctx, 32, 8, 128, 16, fp8, dev,
                                                         n_caches=2 if S * ctx <= 2 ** 19 else 1)
    st = {"i": 0}
    def step():
        kc, vc = caches[st["i"] % len(caches)]; st["i"] += 1
        return PagedAttention.forward_decode(q, kc, vc, bt, sl, ctx, "fp8" if fp8 else "auto", 8, 128 ** -0.5, None, one, one)
    ts = []
    for _ in range(7):
        _, ev = bench.time_gpu(step, 20, 3)
        ts.append(ev / 20 * 1e6)
    print(f"{name:7s} {statistics.median(ts):8.2f} us  (min {min(ts):.2f})", flush=True)
    del caches
    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""x-layout reshape_and_cache (fp8) at prefill sizes by heads per workgroup of the runs kernel ("cache_hpw")."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd import _lib, _custom_ops as ops
dev = torch.device("cuda", 0)
KVH, D, BS = 8, 128, 16
for T in (2048, 8192, 32768, 131072):
    NB = T // BS * 2 + 8
    k = torch.randn((T, KVH, D), device=dev, dtype=torch.bfloat16)
    v = torch.randn_like(k)
    sc = torch.tensor([0.05], device=dev)
    perm = torch.randperm(NB, device=dev)
    t = torch.arange(T, device=dev)
    kc = torch.zeros((NB, KVH, D // 16, BS, 16), dtype=torch.uint8, device=dev)
    vc = torch.zeros((NB, KVH, D, BS), dtype=torch.uint8, device=dev)
    for name, slots in (("aligned", perm[t // BS] * BS + t % BS), ("start5", perm[(t + 5) // BS] * BS + (t + 5) % BS)):
        slots = slots.to(torch.int64)
        out = []
        for hpw in (0, 1, 2, 4):
            _lib.set_tuning("cache_hpw", hpw)
            us = bench.time_gpu_graph(lambda: ops.reshape_and_cache(k, v, kc, vc, slots, "fp8", sc, sc), 20) * 1e6
            nbytes = T * KVH * D * 6 + 8 * T
            out.append(f"hpw {hpw}: {us:7.1f} us {nbytes / us / 1e3:6.0f} GB/s")
        print(f"T={T:6d} {name:8s} " + " | ".join(out), flush=True)
_lib.set_tuning("cache_hpw", 0)

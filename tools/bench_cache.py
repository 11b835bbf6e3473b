#!/usr/bin/env python3
"""reshape_and_cache (fp8 / bf16 cache) at prefill sizes: consecutive-slot runs (aligned and
unaligned sequence starts) and fully random slots, run-aware kernel vs per-token kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd import _lib, _custom_ops as ops

dev = torch.device("cuda", 0)
KVH, D, BS = 8, 128, 16
for T in (1024, 8192, 32768):
    NB = T // BS * 2 + 8
    k = torch.randn((T, KVH, D), device=dev, dtype=torch.bfloat16)
    v = torch.randn_like(k)
    sc = torch.tensor([0.05], device=dev)
    perm = torch.randperm(NB, device=dev)
    t = torch.arange(T, device=dev)
    for kv in ("fp8", "auto"):
        x = 16 if kv == "fp8" else 8
        cdt = torch.uint8 if kv == "fp8" else torch.bfloat16
        kc = torch.zeros((NB, KVH, D // x, BS, x), dtype=cdt, device=dev)
        vc = torch.zeros((NB, KVH, D, BS), dtype=cdt, device=dev)
        cases = {
            "runs_aligned": perm[t // BS] * BS + t % BS,
            "runs_start5": perm[(t + 5) // BS] * BS + (t + 5) % BS,
            "random": torch.randperm(NB * BS, device=dev)[:T],
        }
        for name, slots in cases.items():
            slots = slots.to(torch.int64)
            res = []
            for runs in (1, 0):
                _lib.set_tuning("cache_runs", runs)
                fn = lambda: ops.reshape_and_cache(k, v, kc, vc, slots, kv, sc, sc)
                us = bench.time_gpu_graph(fn, 20) * 1e6
                nbytes = T * KVH * D * (2 * 2 + 2 * (1 if kv == "fp8" else 2)) + 8 * T
                res.append(f"{us:8.1f} us {nbytes / us / 1e3:7.1f} GB/s")
            print(f"T={T:6d} {kv:4s} {name:13s} runs-kernel {res[0]} | per-token {res[1]}", flush=True)
_lib.set_tuning("cache_runs", 1)

# flash layout [NB, BS, KVH, D] (reshape_and_cache_flash): random slots and consecutive runs
for T in (64, 8192, 32768):
    NB = max(T // BS * 2, 64)
    k = torch.randn((T, KVH, D), device=dev, dtype=torch.bfloat16)
    v = torch.randn_like(k)
    sc = torch.tensor([0.05], device=dev)
    t = torch.arange(T, device=dev)
    perm = torch.randperm(NB, device=dev)
    for kv in ("fp8", "auto"):
        cdt = torch.uint8 if kv == "fp8" else torch.bfloat16
        kc = torch.zeros((NB, BS, KVH, D), dtype=cdt, device=dev)
        vc = torch.zeros_like(kc)
        for name, slots in (("random", torch.randperm(NB * BS, device=dev)[:T]),
                            ("runs_start5", perm[(t + 5) // BS % NB] * BS + (t + 5) % BS)):
            slots = slots.to(torch.int64)
            fn = lambda: ops.reshape_and_cache_flash(k, v, kc, vc, slots, kv, sc, sc)
            us = bench.time_gpu_graph(fn, 20) * 1e6
            nbytes = T * KVH * D * (2 * 2 + 2 * (1 if kv == "fp8" else 2)) + 8 * T
            print(f"flash T={T:6d} {kv:4s} {name:12s} {us:8.1f} us {nbytes / us / 1e3:7.1f} GB/s", flush=True)

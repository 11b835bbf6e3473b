#!/usr/bin/env python3
"""A/B of (kernel, tuning) configurations of scaled_mm, interleaved in one process.
python tools/ab_gemm2.py --configs "2;21;21,mm_patch=1" --shapes MxKxN,...   (config = kernel[,key=value...])"""
import argparse, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="2;21")
ap.add_argument("--shapes", default="4096x4096x4096,4096x14336x4096,4096x4096x28672,4096x4096x6144")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--defaults", default="mm_patch=8,mm_store=0,mm_prio=0")
ap.add_argument("--no-rotate", action="store_true", help="one weight buffer (Infinity-Cache resident) instead of the rotation")
a = ap.parse_args()
dev = torch.device("cuda", 0)
defaults = dict(kv.split("=") for kv in a.defaults.split(",") if kv)
cfgs = []
for c in a.configs.split(";"):
    parts = c.split(",")
    cfgs.append((c, int(parts[0]), dict(kv.split("=") for kv in parts[1:])))
for sh in a.shapes.split(","):
    M, K, N = (int(x) for x in sh.split("x"))
    ts = {c[0]: [] for c in cfgs}
    for _ in range(a.rounds):
        for name, k, kv in cfgs:
            for key, v in {**defaults, **kv}.items():
                _lib.set_tuning(key, int(v))
            _lib.set_tuning("mm_kernel", k)
            r = bench.run_scaled_mm(dev, [(K, N)], [M], a.iters, rotate=not a.no_rotate)[0]
            ts[name].append(r["us"])
    _lib.set_tuning("mm_kernel", 0)
    print(f"M={M} K={K} N={N}: " + "  ".join(
        f"[{n}] {statistics.median(v):8.2f} us {2 * M * N * K / statistics.median(v) / 1e6:7.1f} TF" for n, v in ts.items()), flush=True)

#!/usr/bin/env python3
"""A/B of forced scaled_mm kernels on large shapes, interleaved in one process:
python tools/ab_gemm.py --kernels 2,14 [--shapes MxKxN,...]"""
import argparse, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--kernels", default="2,14")
ap.add_argument("--shapes", default="4096x4096x4096,4096x14336x4096,4096x4096x28672,4096x4096x6144,2048x4096x4096,8192x8192x8192")
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda", 0)
ks = [int(k) for k in a.kernels.split(",")]
for sh in a.shapes.split(","):
    M, K, N = (int(x) for x in sh.split("x"))
    ts = {k: [] for k in ks}
    for _ in range(a.rounds):
        for k in ks:
            _lib.set_tuning("mm_kernel", k)
            r = bench.run_scaled_mm(dev, [(K, N)], [M], 20)[0]
            ts[k].append(r["us"])
    _lib.set_tuning("mm_kernel", 0)
    print(f"M={M} K={K} N={N}: " + "  ".join(
        f"k{k} {statistics.median(v):8.2f} us {2 * M * N * K / statistics.median(v) / 1e6:7.1f} TF" for k, v in ts.items()), flush=True)

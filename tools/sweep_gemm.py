#!/usr/bin/env python3
"""Time every scaled_mm kernel choice ("mm_kernel" tuning value) on the configs[2] grid:
python tools/sweep_gemm.py [--ms ...] [--shapes KxN,...] [--kernels 1,2,...]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="33,64,128,256,512,1024,2048,4096")
ap.add_argument("--shapes", default="4096x6144,4096x4096,4096x28672,14336x4096,1024x8192,3584x8192")
ap.add_argument("--kernels", default="0,1,2,3,4,5,6,7,8,9,10")
ap.add_argument("--splits", default="0", help="mm_splitk values; columns are kernel/splits pairs")
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
shapes = [tuple(int(x) for x in s.split("x")) for s in a.shapes.split(",")]
ms = [int(m) for m in a.ms.split(",")]
kernels = [int(k) for k in a.kernels.split(",")]
splits = [int(k) for k in a.splits.split(",")]
cols = [(k, s) for k in kernels for s in splits if not (k in (0, 1, 2, 11) and s > 1)]
name = lambda c: f"k{c[0]}" + (f"s{c[1]}" if c[1] > 1 else "")
dev = torch.device("cuda", 0)
print("K,N,M," + ",".join(name(c) for c in cols) + ",best")
for (K, N) in shapes:
    for M in ms:
        row = []
        for (k, sp) in cols:
            _lib.set_tuning("mm_kernel", k)
            _lib.set_tuning("mm_splitk", sp)
            r = bench.run_scaled_mm(dev, [(K, N)], [M], a.iters)[0]
            row.append(r["us"])
        best = cols[min(range(len(row)), key=lambda i: row[i] if cols[i][0] else 1e9)]
        print(f"{K},{N},{M}," + ",".join(f"{u:.1f}" for u in row) + f",{name(best)}", flush=True)
_lib.set_tuning("mm_kernel", 0)
_lib.set_tuning("mm_splitk", 0)

#!/usr/bin/env python3
"""fp8 scaled_mm sweep (configs[2]): python tools/bench_gemm.py [--ms 1,16,...] [--shapes 4096x6144,...]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="1,16,64,128,256,512,1024,4096")
ap.add_argument("--shapes", default="4096x6144,4096x4096,4096x28672,14336x4096")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--mm-kernel", type=int, default=0, help="0 auto, 1 = 128x128, 2 = 128x256 glds, 3 = 256x256 glds")
a = ap.parse_args()
if a.mm_kernel:
    from vllm_amd import _lib
    _lib.set_tuning("mm_kernel", a.mm_kernel)
shapes = [tuple(int(x) for x in s.split("x")) for s in a.shapes.split(",")]
res = bench.run_scaled_mm(torch.device("cuda", 0), shapes, [int(m) for m in a.ms.split(",")], a.iters)
for r in res:
    print(f"K={r['K']:6d} N={r['N']:6d} M={r['M']:5d}  {r['us']:9.2f} us  {r['tflops']:8.1f} TF  {r['gbps']:7.1f} GB/s  frac={r['frac']:.3f}")

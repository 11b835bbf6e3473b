#!/usr/bin/env python3
"""scaled_mm kernel x split-K sweep timed EAGERLY (events around 200 launches, weights rotating through > 256 MiB):
split-K needs the registered workspace, which a stream under graph capture does not get, so the graph-timed
tools/sweep_gemm.py never splits.  python tools/sweep_gemm_eager.py [--ms ...] [--shapes ...] [--kernels ...] [--splits ...]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_amd import _lib, _custom_ops as ops

ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="256,512,1024")
ap.add_argument("--shapes", default="4096x6144,4096x4096,4096x28672,14336x4096")
ap.add_argument("--kernels", default="0,3,4,5,6,7")
ap.add_argument("--splits", default="1,2,4")
ap.add_argument("--iters", type=int, default=200)
a = ap.parse_args()
dev = torch.device("cuda", 0)
cols = [(int(k), int(s)) for k in a.kernels.split(",") for s in a.splits.split(",") if not (int(k) == 0 and int(s) > 1)]
print("K,N,M," + ",".join(f"k{k}" + (f"s{s}" if s > 1 else "") for k, s in cols) + ",best", flush=True)
for sh in a.shapes.split(","):
    K, N = (int(x) for x in sh.split("x"))
    copies = max(3, (256 << 20) // (K * N) + 2)
    ws = []
    for c in range(copies):
        w = torch.randn((N, K), device=dev, dtype=torch.bfloat16)
        wq, wsc = ops.scaled_fp8_quant(w, use_per_token_if_dynamic=True)
        ws.append((wq.t(), wsc))
        del w
    for M in (int(m) for m in a.ms.split(",")):
        x = torch.randn((M, K), device=dev, dtype=torch.bfloat16)
        aq, asc = ops.scaled_fp8_quant(x, use_per_token_if_dynamic=True)
        out = torch.empty((M, N), device=dev, dtype=torch.bfloat16)
        row = []
        for (k, s) in cols:
            _lib.set_tuning("mm_kernel", k)
            _lib.set_tuning("mm_splitk", 0 if k == 0 else s)
            for i in range(20):
                torch.ops._C.cutlass_scaled_mm(out, aq, ws[i % copies][0], asc, ws[i % copies][1], None)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.iters):
                torch.ops._C.cutlass_scaled_mm(out, aq, ws[i % copies][0], asc, ws[i % copies][1], None)
            e1.record()
            torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) * 1e3 / a.iters)
        best = min(range(len(row)), key=lambda i: row[i] if cols[i][0] else 1e9)
        print(f"{K},{N},{M}," + ",".join(f"{u:.1f}" for u in row) + f",k{cols[best][0]}s{cols[best][1]}", flush=True)
    del ws
    torch.cuda.empty_cache()
_lib.set_tuning("mm_kernel", 0)
_lib.set_tuning("mm_splitk", 0)

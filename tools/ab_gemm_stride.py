#!/usr/bin/env python3
"""Does the power-of-two row stride of A / B cost the large GEMM anything (L2 / HBM channel
conflicts)?  Same 4096^3 problem with rows padded by `pad` bytes (views of wider tensors)."""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd import _custom_ops as ops

dev = torch.device("cuda", 0)
res = {}
for (M, N, K) in [(4096, 4096, 4096), (4096, 28672, 4096), (4096, 4096, 14336)]:
    for pad in (0, 128, 256, 1024):
        af = (torch.randn((M, K + pad), device=dev) * 0.5).to(torch.float8_e4m3fn)
        wf = (torch.randn((N, K + pad), device=dev) * 0.5).to(torch.float8_e4m3fn)
        a, w = af[:, :K], wf[:, :K]
        sa = torch.rand((M, 1), device=dev) + 0.5
        sb = torch.rand((N, 1), device=dev) + 0.5
        fn = lambda: ops.cutlass_scaled_mm(a, w.t(), sa, sb, torch.bfloat16)
        ts = []
        for r in range(5):
            _, ev = bench.time_gpu(fn, 20, 3)
            ts.append(ev / 20 * 1e6)
        us = statistics.median(ts)
        print(json.dumps(dict(M=M, N=N, K=K, pad=pad, med_us=round(us, 2),
                              tflops=round(2 * M * N * K / us / 1e6, 1))), flush=True)

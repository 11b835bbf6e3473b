#!/usr/bin/env python3
"""Eager (non-graph) fp8 scaled_mm launches for rocprofv3:
   rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/prof_gemm.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_amd import _custom_ops as ops

dev = torch.device("cuda", 0)
for (M, K, N) in ((4096, 4096, 4096), (4096, 14336, 4096), (4096, 4096, 28672), (256, 4096, 6144), (64, 4096, 6144), (1, 4096, 4096)):
    w = torch.randn((N, K), device=dev, dtype=torch.bfloat16)
    wq, ws = ops.scaled_fp8_quant(w, use_per_token_if_dynamic=True)
    a = torch.randn((M, K), device=dev, dtype=torch.bfloat16)
    aq, as_ = ops.scaled_fp8_quant(a, use_per_token_if_dynamic=True)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    for _ in range(20):
        torch.ops._C.cutlass_scaled_mm(out, aq, wq.t(), as_, ws, None)
    torch.cuda.synchronize()
    print(M, K, N, "done", flush=True)

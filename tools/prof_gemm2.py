#!/usr/bin/env python3
"""Eager fp8 scaled_mm launches of ONE forced kernel on the large shapes, for rocprofv3 passes:
   MM_KERNEL=21 SHAPES=4096x14336x4096 rocprofv3 --pmc ... -d OUT -- python3 tools/prof_gemm2.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_amd import _lib, _custom_ops as ops

dev = torch.device("cuda", 0)
_lib.set_tuning("mm_kernel", int(os.environ.get("MM_KERNEL", "0")))
for sh in os.environ.get("SHAPES", "4096x4096x4096,4096x14336x4096,4096x4096x28672").split(","):
    M, K, N = (int(x) for x in sh.split("x"))
    w = torch.randn((N, K), device=dev, dtype=torch.bfloat16)
    wq, ws = ops.scaled_fp8_quant(w, use_per_token_if_dynamic=True)
    a = torch.randn((M, K), device=dev, dtype=torch.bfloat16)
    aq, as_ = ops.scaled_fp8_quant(a, use_per_token_if_dynamic=True)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    for _ in range(int(os.environ.get("LAUNCHES", "30"))):
        torch.ops._C.cutlass_scaled_mm(out, aq, wq.t(), as_, ws, None)
    torch.cuda.synchronize()
    print(M, K, N, "done", flush=True)

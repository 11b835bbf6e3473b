#!/usr/bin/env python3
"""Same question as decode_drift.py for the large-M fp8 GEMM (4096^3, 4096 x 6144 x K 4096): event-timed groups of
20 back-to-back launches from the start of a fresh process."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_amd import _custom_ops as ops
dev = torch.device("cuda", 0)
for (M, K, N) in ((4096, 4096, 4096), (4096, 4096, 6144)):
    w = torch.randn((N, K), device=dev, dtype=torch.bfloat16)
    wq, ws = ops.scaled_fp8_quant(w, use_per_token_if_dynamic=True)
    a = torch.randn((M, K), device=dev, dtype=torch.bfloat16)
    aq, as_ = ops.scaled_fp8_quant(a, use_per_token_if_dynamic=True)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    wt = wq.t()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    evs[0].record()
    for g in range(40):
        for _ in range(20):
            torch.ops._C.cutlass_scaled_mm(out, aq, wt, as_, ws, None)
        evs[g + 1].record()
    torch.cuda.synchronize()
    print(f"M {M} K {K} N {N} us per launch, groups of 20:", " ".join(f"{evs[g].elapsed_time(evs[g + 1]) * 1000 / 20:.1f}" for g in range(40)))

#!/usr/bin/env python3
"""Row quantisation kernels (dynamic per-token fp8 quant, RMSNorm + quant): workgroups per launch
timing over a few shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd import _lib
from vllm_amd import _custom_ops as ops  # registers torch.ops._C
dev = torch.device("cuda", 0)
for (M, H) in ((4096, 4096), (4096, 8192), (4096, 14336), (8192, 4096), (1024, 4096), (64, 4096)):
    x = torch.randn((M, H), device=dev, dtype=torch.bfloat16)
    res = torch.randn((M, H), device=dev, dtype=torch.bfloat16)
    w = torch.randn((H,), device=dev, dtype=torch.bfloat16)
    out = torch.empty((M, H), device=dev, dtype=torch.float8_e4m3fn)
    sc = torch.empty((M, 1), device=dev, dtype=torch.float32)
    one = torch.tensor([0.05], device=dev)
    cases = {
        "per_token_quant": (lambda: torch.ops._C.dynamic_per_token_scaled_fp8_quant(out, x, sc, None), M * H * 3),
        "rms_dyn": (lambda: torch.ops._C.rms_norm_dynamic_per_token_quant(out, x, w, sc, 1e-6, None, None), M * H * 3),
        "rms_dyn_res": (lambda: torch.ops._C.rms_norm_dynamic_per_token_quant(out, x, w, sc, 1e-6, None, res), M * H * 7),
        "add_rms_static": (lambda: torch.ops._C.fused_add_rms_norm_static_fp8_quant(out, x, res, w, one, 1e-6), M * H * 7),
    }
    if H == 4096 and M >= 4096:
        x2 = torch.randn((M, 2 * 14336), device=dev, dtype=torch.bfloat16)
        o2 = torch.empty((M, 14336), device=dev, dtype=torch.float8_e4m3fn)
        cases["silu_mul_quant_d14336"] = (lambda: torch.ops._C.silu_and_mul_quant(o2, x2, one), M * 14336 * 5)
    for name, (fn, nbytes) in cases.items():
        t = bench.time_gpu_graph(fn, 20)
        print(f"M={M} H={H} {name:16s} {t * 1e6:6.2f} us ({nbytes / t / 1e12:4.2f} TB/s)", flush=True)

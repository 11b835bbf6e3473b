#!/usr/bin/env python3
"""Forced scaled_mm kernels against the shipped 256 x 256 kernel on random operands (bit-identical expected:
same K order, same MFMA): python tools/check_gemm_variant.py --kernels 17 [--ref 2]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_amd import _lib, _custom_ops as ops

ap = argparse.ArgumentParser()
ap.add_argument("--kernels", default="17")
ap.add_argument("--ref", type=int, default=2)
ap.add_argument("--shapes", default="4096x4096x4096,1000x1024x1288,4096x14336x4096,520x4096x6144")
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
bad = 0
for sh in a.shapes.split(","):
    M, K, N = (int(x) for x in sh.split("x"))
    w = torch.randn((N, K), device=dev, dtype=torch.bfloat16)
    wq, ws = ops.scaled_fp8_quant(w, use_per_token_if_dynamic=True)
    x = torch.randn((M, K), device=dev, dtype=torch.bfloat16)
    aq, as_ = ops.scaled_fp8_quant(x, use_per_token_if_dynamic=True)
    bias = torch.randn((N,), device=dev, dtype=torch.bfloat16)
    outs = {}
    for k in [a.ref] + [int(v) for v in a.kernels.split(",")]:
        _lib.set_tuning("mm_kernel", k)
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
        torch.ops._C.cutlass_scaled_mm(out, aq, wq.t(), as_, ws, bias)
        torch.cuda.synchronize()
        outs[k] = out
    _lib.set_tuning("mm_kernel", 0)
    for k, o in outs.items():
        if k == a.ref:
            continue
        same = torch.equal(o.view(torch.int16), outs[a.ref].view(torch.int16))
        d = (o.float() - outs[a.ref].float()).abs().max().item()
        print(f"{sh}: kernel {k} vs {a.ref}: {'identical' if same else 'DIFFERENT'} max|d| {d:.3g}", flush=True)
        bad += 0 if same else 1
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Timeline of one launch of the staggered / persistent GEMM kernel from in-kernel 100 MHz stamps (probe
build: VLLM_AMD_LIB=vllm_amd/csrc/libvllm_amd_probe.so MM_KERNEL=21 python tools/pp_stamps.py): kernel entry,
first barrier passed (first K block landed), K loop done, epilogue stores issued / done, per wave; and the
shader clock over the loop.  The last of 20 back-to-back launches is what the stamps hold."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_amd import _lib, _custom_ops as ops
assert "probe" in _lib.LIB_PATH
dev = torch.device("cuda", 0)
kern = int(os.environ.get("MM_KERNEL", "21"))
_lib.set_tuning("mm_kernel", kern)
for kv in os.environ.get("TUNING", "").split(","):
    if kv:
        _lib.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
shapes = os.environ.get("SHAPES", "4096x256x4096,4096x4096x4096,4096x14336x4096,4096x4096x28672")
for sh in shapes.split(","):
    M, K, N = (int(x) for x in sh.split("x"))
    w = torch.randn((N, K), device=dev, dtype=torch.bfloat16)
    wq, ws = ops.scaled_fp8_quant(w, use_per_token_if_dynamic=True)
    a = torch.randn((M, K), device=dev, dtype=torch.bfloat16)
    aq, as_ = ops.scaled_fp8_quant(a, use_per_token_if_dynamic=True)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    fn = lambda: torch.ops._C.cutlass_scaled_mm(out, aq, wq.t(), as_, ws, None)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    n = int(0.3 / (20 * 2.0 * M * N * K / 2.5e15)) + 3
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    nwg = min(256, tiles)
    buf = (ctypes.c_ulonglong * (nwg * 64))()
    assert _lib.lib.vllm_amd_probe_pp_stamps(buf, nwg) == 0
    t = torch.tensor(list(buf), dtype=torch.int64).view(nwg, 8, 8).double()
    t0 = t[:, :, 0].min()
    rel = lambda i: (t[:, :, i] - t0) / 100.0
    clk = ((t[:, :, 6] - t[:, :, 5]) / ((t[:, :, 2] - t[:, :, 1]) / 100.0) / 1e3).median().item()
    def st(x, sel=slice(None)):
        x = x[:, sel]
        return f"med {x.median():6.2f} max {x.max():6.2f}"
    print(f"M {M} K {K} N {N} kernel {kern}: {us:.2f} us/launch = {2*M*N*K/us/1e6:.0f} TF; clock over the loop {clk:.2f} GHz")
    print(f"   us after the first wave's entry: entry {st(rel(0))} | first barrier passed {st(rel(1))} | loop done: waves 0-3 {st(rel(2), slice(0, 4))}, "
          f"waves 4-7 {st(rel(2), slice(4, 8))} | stores issued {st(rel(3))} | stores done {st(rel(4))}", flush=True)

#!/usr/bin/env python3
"""In-kernel clock and cycles per K block of the large-tile GEMM (probe build:
VLLM_AMD_LIB=vllm_amd/csrc/libvllm_amd_probe.so).  s_memtime counts shader cycles, s_memrealtime
100 MHz ticks: clock = d(memtime) / d(memrealtime) x 100 MHz, measured over the K loop of every
workgroup after ~1 s of back-to-back launches on random operands."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_amd import _lib, _custom_ops as ops
assert "probe" in _lib.LIB_PATH
dev = torch.device("cuda", 0)
if os.environ.get("MM_KERNEL"):   # force a tile variant (14, 15, 16 ...)
    _lib.set_tuning("mm_kernel", int(os.environ["MM_KERNEL"]))
for (M, K, N) in ((4096, 4096, 4096), (4096, 14336, 4096), (4096, 4096, 28672)):
    w = torch.randn((N, K), device=dev, dtype=torch.bfloat16)
    wq, ws = ops.scaled_fp8_quant(w, use_per_token_if_dynamic=True)
    a = torch.randn((M, K), device=dev, dtype=torch.bfloat16)
    aq, as_ = ops.scaled_fp8_quant(a, use_per_token_if_dynamic=True)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    n = int(1.0 / (2.0 * M * N * K / 2.5e15)) + 10
    for _ in range(n):
        torch.ops._C.cutlass_scaled_mm(out, aq, wq.t(), as_, ws, None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        torch.ops._C.cutlass_scaled_mm(out, aq, wq.t(), as_, ws, None)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    nwg = min(4096, ((M + 255) // 256) * ((N + 255) // 256))
    buf = (ctypes.c_ulonglong * (nwg * 4))()
    assert _lib.lib.vllm_amd_probe_mm_stamps(buf, nwg) == 0
    t = torch.tensor(list(buf), dtype=torch.int64).view(nwg, 4)
    cyc = (t[:, 2] - t[:, 0]).double(); real = (t[:, 3] - t[:, 1]).double() / 100.0   # us
    clk = (cyc / real / 1e3).median().item()      # GHz
    buf2 = (ctypes.c_ulonglong * nwg)()
    assert _lib.lib.vllm_amd_probe_mm_stamps_end(buf2, nwg) == 0
    te = torch.tensor(list(buf2), dtype=torch.int64)
    t_first = t[:, 1].min()
    epi = (te - t[:, 3]).double() / 100.0
    print(f"   loop start (us after first WG): med {((t[:,1]-t_first).double()/100).median():.1f}; loop end: med {((t[:,3]-t_first).double()/100).median():.1f} "
          f"max {((t[:,3]-t_first).double()/100).max():.1f}; epilogue per WG: med {epi.median():.1f} max {epi.max():.1f} us; last store done at {((te-t_first).double()/100).max():.1f} us")
    nb = K // 128
    nph = min(nwg, 512)
    buf3 = (ctypes.c_ulonglong * (nph * 24))()
    assert _lib.lib.vllm_amd_probe_mm_phase(buf3, nph) == 0
    ph = torch.tensor(list(buf3), dtype=torch.int64).view(nph, 8, 3).double() / nb
    print("   cycles per K block and wave (median over workgroups): wait for DMA + barrier "
          f"{ph[:, :, 0].median():.0f}, barrier -> first fragments {ph[:, :, 1].median():.0f}, MFMA section {ph[:, :, 2].median():.0f}"
          f"   per wave wait: {[int(ph[:, w, 0].median()) for w in range(8)]}  mfma: {[int(ph[:, w, 2].median()) for w in range(8)]}")
    print(f"M {M} K {K} N {N}: {us:.1f} us/launch; K loop per WG: median {real.median():.1f} us, {cyc.median():.0f} cycles "
          f"= {cyc.median() / nb:.0f} cycles / K block at {clk:.2f} GHz; start spread {(t[:,1]-t[:,1].min()).double().max()/100:.1f} us")

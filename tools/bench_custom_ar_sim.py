#!/usr/bin/env python3
"""Protocol floor of the peer-to-peer all-reduce: simulated ranks (one process, one stream and one
hardware queue per rank, all buffers in the ONE GPU's memory), so the numbers hold the launch, the
flag barriers and the local reads / writes -- no xGMI link.  Run with GPU_MAX_HW_QUEUES=16.
Per message size and world: microseconds per all-reduce on rank 0's stream (all ranks launched
back to back from the host, 50 rounds, HIP events)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_amd import _lib
L = _lib.lib
arr = lambda ptrs: (C.c_void_p * len(ptrs))(*ptrs)
dev = torch.device("cuda", 0)
ms, max_size = L.vllm_amd_ar_meta_size(), 8 << 20
for world in (2, 4, 8):
    metas = [torch.zeros(ms + max_size, dtype=torch.uint8, device=dev) for _ in range(world)]
    regs = [torch.empty(max_size, dtype=torch.uint8, device=dev) for _ in range(world)]
    hs = []
    for r in range(world):
        h = C.c_void_p()
        assert L.vllm_amd_ar_init(C.byref(h), arr([m.data_ptr() for m in metas]), world, r, 1) == 0
        assert L.vllm_amd_ar_register_buffer(h, arr([b.data_ptr() for b in regs])) == 0
        hs.append(h)
    streams = [torch.cuda.Stream(device=dev) for _ in range(world)]
    for nbytes in (32 * 1024, 256 * 1024, 1 << 20, 4 << 20):
        numel = nbytes // 2
        xs = [torch.randn(numel, device=dev).to(torch.bfloat16) for _ in range(world)]
        outs = [torch.empty_like(x) for x in xs]

        def rounds(n):
            for _ in range(n):
                for r in range(world):
                    # registered = the inputs sit in the registered buffers already (no staging copy)
                    L.vllm_amd_ar_all_reduce(hs[r], regs[r].data_ptr(), outs[r].data_ptr(), numel, 2, None, 0,
                                             streams[r].cuda_stream)
        rounds(5)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(streams[0])
        rounds(50)
        e1.record(streams[0])
        torch.cuda.synchronize()
        t = C.c_int(0)
        L.vllm_amd_ar_error(hs[0], C.byref(t))
        print(f"world {world} {nbytes >> 10:5d} KiB: {e0.elapsed_time(e1) * 1e3 / 50:7.1f} us per all-reduce"
              f"{'  (TIMED OUT)' if t.value else ''}", flush=True)
    for h in hs:
        L.vllm_amd_ar_dispose(h)

#!/usr/bin/env python3
"""Simulated ranks (one process, a stream per rank, one GPU): back-to-back peer-to-peer all-reduces without host
synchronisation in between.  GPU_MAX_HW_QUEUES=16 python tools/ar_b2b.py [world] [bytes] [launches]"""
import ctypes as C, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_amd import _lib
L = _lib.lib
arr = lambda ptrs: (C.c_void_p * len(ptrs))(*ptrs)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nbytes = int(sys.argv[2]) if len(sys.argv) > 2 else 4 << 20
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda", 0)
ms = L.vllm_amd_ar_meta_size()
max_size = 16 << 20
metas = [torch.zeros(ms + max_size, dtype=torch.uint8, device=dev) for _ in range(world)]
regs = [torch.empty(max_size, dtype=torch.uint8, device=dev) for _ in range(world)]
tables = [torch.zeros(1 << 16, dtype=torch.uint8, device=dev) for _ in range(world)]
hs = []
for r in range(world):
    h = C.c_void_p()
    assert L.vllm_amd_ar_init(C.byref(h), arr([m.data_ptr() for m in metas]), max_size, tables[r].data_ptr(), tables[r].numel(), world, r, 1) == 0
    assert L.vllm_amd_ar_set_timeout_ms(h, 2000) == 0
    assert L.vllm_amd_ar_register_buffer(h, arr([b.data_ptr() for b in regs])) == 0
    hs.append(h)
streams = [torch.cuda.Stream(device=dev) for _ in range(world)]
numel = nbytes // 2
ins = [torch.randn(numel, device=dev).to(torch.bfloat16) for _ in range(world)]
outs = [torch.zeros(numel, dtype=torch.bfloat16, device=dev) for _ in range(world)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(launches):
    for r in range(world):
        rc = L.vllm_amd_ar_all_reduce(hs[r], ins[r].data_ptr(), outs[r].data_ptr(), numel, 2, regs[r].data_ptr(), max_size, streams[r].cuda_stream)
        if rc != 0:
            print("launch", i, "rank", r, L.vllm_amd_last_error().decode()); sys.exit(1)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
to = []
for r in range(world):
    t = C.c_int(0); L.vllm_amd_ar_error(hs[r], C.byref(t)); to.append(t.value)
want = sum(x.float() for x in ins).to(torch.bfloat16)
print(f"world {world} {nbytes} B x {launches} launches: {dt*1e3:.1f} ms, timed_out {to}, ok {[bool(torch.equal(o, want)) for o in outs]}")

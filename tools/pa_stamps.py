#!/usr/bin/env python3
"""Decode-kernel time stamps in the library's own environment (torch memory, bench inputs).
Run with VLLM_AMD_LIB=vllm_amd/csrc/libvllm_amd_probe.so (make -C vllm_amd/csrc libvllm_amd_probe.so).
Prints, per config: event-timed us for variant 1 (full) and 5 (loads only), and the distribution
over workgroups of entry / first K consumed / loop end / exit (us after the earliest entry)."""
import ctypes, os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd import _lib
from vllm_amd.attention import PagedAttention

assert "probe" in _lib.LIB_PATH, "set VLLM_AMD_LIB to the probe build"
dev = torch.device("cuda", 0)
for name, S, ctx in (("C2", 64, 4096), ("S32d", 32, 8192)):
    H, KVH, D, BS = 32, 8, 128, 16
    q, caches, bt, sl, one = bench.make_attention_inputs(S, ctx, H, KVH, D, BS, True, dev, n_caches=2)
    nbytes = bench.attention_bytes(S, ctx, H, KVH, D, BS, 1)
    st = {"i": 0}
    def step():
        kc, vc = caches[st["i"] % 2]; st["i"] += 1
        return PagedAttention.forward_decode(q, kc, vc, bt, sl, ctx, "fp8", KVH, D ** -0.5, None, one, one)
    for variant in (1, 5, 1, 5):
        _lib.set_tuning("pa_variant", variant)
        _, ev = bench.time_gpu(step, 20, 5)
        us = ev / 20 * 1e6
        print(f"{name} variant {variant}: {us:.2f} us  {nbytes / us / 1e3:.0f} GB/s")
        nwg = S * KVH
        buf = (ctypes.c_ulonglong * (nwg * 4))()
        assert _lib.lib.vllm_amd_probe_stamps(buf, nwg) == 0
        t = torch.tensor(list(buf), dtype=torch.int64).view(nwg, 4)
        t = (t - t[:, 0].min()).double() / 100.0
        for k, lab in enumerate(("entry", "first K consumed", "loop end", "exit")):
            v = t[:, k].sort().values
            print(f"    {lab:18s} min {v[0]:6.2f}  p10 {v[nwg // 10]:6.2f}  med {v[nwg // 2]:6.2f}  p90 {v[nwg * 9 // 10]:6.2f}  max {v[-1]:6.2f} us")
        if variant == 1:
            # who is slow?  exit time by hardware workgroup id % 8 (= XCD under round-robin
            # placement) and by kv head (= id % 8 here as well: KVH = 8) / by sequence
            ex = t[:, 3]
            print("    exit by id%8 :", " ".join(f"{ex[i::8].median():6.1f}" for i in range(8)))
            print("    exit by id//8 (seq), first 16:", " ".join(f"{ex[8*i:8*i+8].median():5.1f}" for i in range(16)))
            dur = (t[:, 3] - t[:, 0])
            print(f"    duration per WG: min {dur.min():.1f} med {dur.median():.1f} max {dur.max():.1f}")
    _lib.set_tuning("pa_variant", 1)
    del q, caches

#!/usr/bin/env python3
"""pa_sp on ragged v2 batches with fewer than 512 (sequence, kv head) pairs: the other side of the trade
that tools/sweep_pa_sp.py shows on uniform batches."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd import _lib
from vllm_amd.attention.paged_attn import PagedAttention
dev = torch.device("cuda", 0)
H, KVH, D, BS = 32, 8, 128, 16
for S, lo, hi in ((16, 8192, 65536), (32, 4096, 32768), (8, 16384, 131072), (48, 2048, 16384)):
    q, caches, bt, sl, one = bench.make_attention_inputs(S, hi, H, KVH, D, BS, True, dev, n_caches=1)
    kc, vc = caches[0]
    g = torch.Generator().manual_seed(S)
    lens = torch.randint(lo, hi + 1, (S,), generator=g).to(torch.int32)
    sl = lens.to(dev)
    max_len = int(lens.max())
    nbytes = int(lens.sum()) * KVH * D * 2
    for sp in (0, 4, 8, 16, 32):
        _lib.set_tuning("pa_sp", sp)
        fn = lambda: PagedAttention.forward_decode(q, kc, vc, bt, sl, max_len, "fp8", KVH, D ** -0.5, None, one, one)
        bench.precondition(fn, 30.0)
        _, ev = bench.time_gpu(fn, 30, 3)
        per = ev / 30
        print(f"{S} x U[{lo},{hi}]: pa_sp {sp:3d}: {per*1e6:8.1f} us  {nbytes/per/8e12*100:5.1f} % of peak", flush=True)
    del kc, vc, caches
    torch.cuda.empty_cache()
_lib.set_tuning("pa_sp", 0)

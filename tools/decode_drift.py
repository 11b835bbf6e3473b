#!/usr/bin/env python3
"""Does the headline decode launch speed up over the first few hundred launches of a process?
Event-timed groups of 20 launches, back to back, configs[1] (alternating two cache copies)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd.attention.paged_attn import PagedAttention
dev = torch.device("cuda", 0)
S, ctx, H, KVH, D, BS = 64, 4096, 32, 8, 128, 16
q, caches, bt, sl, one = bench.make_attention_inputs(S, ctx, H, KVH, D, BS, True, dev, n_caches=2)
torch.cuda.synchronize()
st = {"i": 0}
def fn():
    kc, vc = caches[st["i"] % 2]; st["i"] += 1
    return PagedAttention.forward_decode(q, kc, vc, bt, sl, ctx, "fp8", KVH, D ** -0.5, None, one, one)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
evs[0].record()
for g in range(40):
    for _ in range(20):
        fn()
    evs[g + 1].record()
torch.cuda.synchronize()
print("us per launch, groups of 20:", " ".join(f"{evs[g].elapsed_time(evs[g + 1]) * 1000 / 20:.1f}" for g in range(40)))

#!/usr/bin/env python3
"""paged_attention_v1 (through PagedAttention.forward_decode): one workgroup per (sequence, kv head)
("plain", the default) against the balanced form ("pa_balance" = 1) on uniform, ragged and skewed
batches.  fp8 KV, 32q / 8kv heads, head 128, block 16; HIP-event timed, median of 5 x 20 launches,
both forms interleaved in one process; GB/s = bytes the sequences actually hold / time."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd import _lib
from vllm_amd.attention import PagedAttention

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)


def run(name, S, ctx, lens=None, H=32, KVH=8):
    q, caches, bt, sl, one = bench.make_attention_inputs(S, ctx, H, KVH, 128, 16, True, dev,
                                                         n_caches=2 if S * ctx * KVH < 2 ** 25 else 1)
    if lens is not None:
        sl = lens.to(torch.int32).to(dev)
    ml = int(sl.max())
    st = {"i": 0}

    def step():
        kc, vc = caches[st["i"] % len(caches)]
        st["i"] += 1
        return PagedAttention.forward_decode(q, kc, vc, bt, sl, ml, "fp8", KVH, 128 ** -0.5, None, one, one)
    ts = {0: [], 1: []}
    for _ in range(5):
        for form in (0, 1):
            _lib.set_tuning("pa_balance", form)
            _, ev = bench.time_gpu(step, 20, 3)
            ts[form].append(ev / 20 * 1e6)
    _lib.set_tuning("pa_balance", 0)
    nbytes = int(sl.sum().item()) * KVH * 256 + S * H * 128 * 4
    a, b = statistics.median(ts[0]), statistics.median(ts[1])
    print(f"{name:34s} plain {a:8.2f} us {nbytes / a / 1e3:6.0f} GB/s   balanced {b:8.2f} us {nbytes / b / 1e3:6.0f} GB/s",
          flush=True)


run("uniform 64 x 4096 (configs[1])", 64, 4096)
run("uniform 32 x 8192", 32, 8192)
run("uniform 256 x 4096", 256, 4096)
run("ragged 64 x U[512,4096]", 64, 4096, torch.randint(512, 4097, (64,), generator=g))
run("ragged 256 x U[64,4096]", 256, 4096, torch.randint(64, 4097, (256,), generator=g))
run("skewed 1 x 8192 + 63 x 256", 64, 8192, torch.tensor([8192] + [256] * 63))
run("skewed 4 x 8192 + 124 x 512", 128, 8192, torch.tensor([8192] * 4 + [512] * 124))
run("skewed 2 x 4096 + 62 x 128", 64, 4096, torch.tensor([4096] * 2 + [128] * 62))

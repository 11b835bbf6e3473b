#!/usr/bin/env python3
"""Super-partition width ("pa_sp") of the decode kernel at shapes with fewer (sequence, kv head) pairs than workgroup
slots and SHORT contexts (TP-sharded serving: 8q / 1kv heads per GPU)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd import _lib
dev = torch.device("cuda", 0)
for name, S, ctx, H, KVH in (("64 x 4096, 8q/1kv", 64, 4096, 8, 1), ("64 x 8192, 8q/1kv", 64, 8192, 8, 1),
                             ("128 x 4096, 8q/1kv", 128, 4096, 8, 1), ("32 x 4096, 16q/2kv", 32, 4096, 16, 2),
                             ("16 x 8192, 32q/8kv", 16, 8192, 32, 8), ("256 x 2048, 8q/1kv", 256, 2048, 8, 1)):
    out = []
    for sp in (0, 1, 2, 4, 8):
        _lib.set_tuning("pa_sp", sp)
        r = bench.run_attention_shape(name, S, ctx, H, KVH, dev, iters=30)
        out.append(f"sp {sp}: {r['us']:6.1f} us {r['frac_of_hbm_peak']*100:4.1f} %")
    print(f"{name}: " + " | ".join(out), flush=True)
_lib.set_tuning("pa_sp", 0)

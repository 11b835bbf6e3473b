#!/usr/bin/env python3
"""Sweep the super-partition width ("pa_sp") of the decode kernel at the single-GPU shapes of configs[3] / [4]
with bench.run_attention_shape's own timing."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd import _lib
dev = torch.device("cuda", 0)
for name, S, ctx, H, KVH in (("configs[3] per-GPU", 256, 32768, 8, 1), ("configs[4] N=1", 8, 131072, 64, 8),
                             ("configs[4] N=8", 8, 131072, 8, 1)):
    for sp in (0, 2, 4, 8, 16, 32, 64):
        _lib.set_tuning("pa_sp", sp)
        r = bench.run_attention_shape(name, S, ctx, H, KVH, dev, iters=int(os.environ.get("ITERS", "30")))
        print(f"{name}: pa_sp {sp:3d}: {r['us']:8.1f} us  {r['frac_of_hbm_peak']*100:5.1f} % of peak", flush=True)
_lib.set_tuning("pa_sp", 0)

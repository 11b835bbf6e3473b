#!/usr/bin/env python3
"""A/B of the tail-round split of the 256 x 256 GEMM tile (scaled_mm.hip, "mm_tail" tuning
flag: 0 = auto, 1 = off), interleaved rounds in one process."""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd import _lib

dev = torch.device("cuda", 0)
shapes = [((4096, 6144), 4096), ((4096, 6144), 3584), ((4096, 4096), 4096), ((8192, 6144), 4096)]
res = {}
for r in range(5):
    for off in (0, 1):
        _lib.set_tuning("mm_tail", off)
        for (kn, m) in shapes:
            e = bench.run_scaled_mm(dev, [kn], [m])[0]
            res.setdefault((kn, m, off), []).append(e["us"])
for (kn, m, off), v in sorted(res.items()):
    us = statistics.median(v)
    print(json.dumps(dict(K=kn[0], N=kn[1], M=m, mm_tail_off=off, med_us=round(us, 2),
                          tflops=round(2 * m * kn[0] * kn[1] / us / 1e6, 1))), flush=True)

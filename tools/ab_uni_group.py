#!/usr/bin/env python3
"""unified_attention decode, NHD cache: workgroup-order variants ("uni_decode": 1 default, 2 XCD-grouped always, 3 never)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd import _lib
dev = torch.device("cuda", 0)
for v in (1, 2, 3):
    _lib.set_tuning("uni_decode", v)
    r = bench.run_unified_decode(dev, iters=10)
    print(v, {k: (x["us"], x["frac_of_hbm_peak"]) for k, x in r.items()}, flush=True)
_lib.set_tuning("uni_decode", 1)

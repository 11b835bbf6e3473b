#!/usr/bin/env python3
"""Where the waves of the 256-row prefill kernel spend their time: per-phase shader-clock sums from the probe
build (make -C vllm_amd/csrc libvllm_amd_probe.so; VLLM_AMD_LIB=vllm_amd/csrc/libvllm_amd_probe.so
[SET=pf_fa_stagger=0] [SHAPE=1x8192x0] python tools/fa_stamps.py).  Cycles per computed tile, workgroups with
at least 32 tiles, waves 0-3 and 4-7."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vllm_amd import _lib
from vllm_amd.attention.ops import unified_attention
assert "probe" in _lib.LIB_PATH
dev = torch.device("cuda", 0)
for kv_ in filter(None, os.environ.get("SET", "").split(",")):
    _lib.set_tuning(kv_.split("=")[0], int(kv_.split("=")[1]))
B, ql, cl = (int(x) for x in os.environ.get("SHAPE", "1x8192x0").split("x"))
fp8 = os.environ.get("KV", "fp8") == "fp8"
H, KVH, D, BS = 32, 8, 128, 16
T = B * ql
g = torch.Generator(device=dev).manual_seed(0)
q = (torch.rand((T, H, D), device=dev, generator=g) - 0.5).to(torch.bfloat16)
o = torch.empty_like(q)
nb = (ql + cl + BS - 1) // BS
mk = (lambda: torch.randint(0, 120, (B * nb, KVH, BS, D), device=dev, dtype=torch.uint8)) if fp8 else \
     (lambda: (torch.rand((B * nb, KVH, BS, D), device=dev) - 0.5).to(torch.bfloat16))
kf, vf = mk().permute(0, 2, 1, 3), mk().permute(0, 2, 1, 3)
bt = torch.randperm(B * nb, device=dev).to(torch.int32).view(B, nb)
start = (torch.arange(B + 1, device=dev) * ql).to(torch.int32)
seql = torch.full((B,), ql + cl, dtype=torch.int32, device=dev)
one = torch.full((1,), 0.02, device=dev)
for _ in range(5):
    unified_attention(q, kf, vf, o, start, ql, seql, ql + cl, D ** -0.5, True, (-1, -1), bt, 0.0, None,
                      one if fp8 else None, one if fp8 else None)
torch.cuda.synchronize()
nwg = min(2048, T // 64 + B)
buf = (ctypes.c_ulonglong * (nwg * 64))()
assert _lib.lib.vllm_amd_probe_fa_stamps(buf, nwg) == 0
t = torch.tensor(list(buf), dtype=torch.int64).view(nwg, 8, 8).double()
names = ["staging (bottom)", "staging (top)", "K fill + S^T", "-", "softmax + PV", "barrier"]
sel = t[:, 0, 6] >= 32
print(f"{B} x {ql} (+{cl} cached), {os.environ.get('SET', '')}: {int(sel.sum())} workgroups with >= 32 tiles; cycles per computed tile")
for lo, hi in ((0, 4), (4, 8)):
    x = t[sel][:, lo:hi]
    tiles = x[:, :, 6].clamp(min=1)
    line = " | ".join(f"{names[i]} {float((x[:, :, i] / tiles).mean()):7.0f}" for i in range(6))
    tot = float((x[:, :, :6].sum(dim=2) / tiles).mean())
    print(f"  waves {lo}-{hi-1}: {line} | sum {tot:7.0f} | kernel / tile {float((x[:, :, 7] / tiles).mean()):7.0f}")

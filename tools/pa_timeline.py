#!/usr/bin/env python3
"""Timeline of a partitioned (v2) decode launch from the probe build's per-workgroup stamps:
how many real workgroups are resident over time, and when they stream.
VLLM_AMD_LIB=vllm_amd/csrc/libvllm_amd_probe.so python tools/pa_timeline.py [--uniform]"""
import argparse, ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd import _lib
from vllm_amd.attention import PagedAttention

ap = argparse.ArgumentParser()
ap.add_argument("--uniform", action="store_true")
ap.add_argument("--S", type=int, default=256)
ap.add_argument("--ctx", type=int, default=16384)
a = ap.parse_args()
assert "probe" in _lib.LIB_PATH
dev = torch.device("cuda", 0)
S, ctx, H, KVH, D, BS = a.S, a.ctx, 32, 8, 128, 16
q, caches, bt, sl, one = bench.make_attention_inputs(S, ctx, H, KVH, D, BS, True, dev, n_caches=1)
if not a.uniform:
    g = torch.Generator(device=dev).manual_seed(1)
    sl = torch.randint(ctx // 8, ctx + 1, (S,), generator=g, device=dev, dtype=torch.int32)
kc, vc = caches[0]
fn = lambda: PagedAttention.forward_decode(q, kc, vc, bt, sl, ctx, "fp8", KVH, D ** -0.5, None, one, one)
for _ in range(3):
    fn()
torch.cuda.synchronize()
_, ev = bench.time_gpu(fn, 10, 2)
print(f"{'uniform' if a.uniform else 'ragged'} S={S} ctx={ctx}: {ev / 10 * 1e6:.1f} us per call (main + reduce)")
n = 16384
buf = (ctypes.c_ulonglong * (n * 4))()
assert _lib.lib.vllm_amd_probe_stamps(buf, n) == 0
t = torch.tensor(list(buf), dtype=torch.int64).view(n, 4)
real = t[:, 2] > 0            # workgroups that reached the loop end stamp
t0 = t[t[:, 0] > 0, 0].min()
tt = (t - t0).double() / 100.0
r = tt[real]
print(f"workgroups with stamps {int((t[:, 0] > 0).sum())}, real {int(real.sum())}; span {r[:, 3].max():.1f} us")
dur = r[:, 3] - r[:, 0]
print(f"real workgroup duration: min {dur.min():.1f} med {dur.median():.1f} max {dur.max():.1f} us; "
      f"entry->first K {(r[:, 1] - r[:, 0]).median():.2f} us, loop end->exit {(r[:, 3] - r[:, 2]).median():.2f} us")
span = r[:, 3].max().item()
edges = torch.linspace(0, span, 21)
for i in range(20):
    mid = 0.5 * (edges[i] + edges[i + 1])
    resident = int(((r[:, 0] <= mid) & (r[:, 3] > mid)).sum())
    streaming = int(((r[:, 1] <= mid) & (r[:, 2] > mid)).sum())
    print(f"  t={mid:7.1f} us: resident {resident:4d}  streaming {streaming:4d}")

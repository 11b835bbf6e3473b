#!/usr/bin/env python3
"""Where the decode kernel's workgroups run: probe build with -DVA_PROBE_HWID (stamp 1 = HW_ID |
XCC_ID << 32).  Prints, for a 512-workgroup v1 launch, the (XCC, SE, CU) of every hardware
workgroup id and whether ids i and i + 256 share a CU; then exit times per CU for a ragged batch."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd import _lib
from vllm_amd.attention import PagedAttention

assert "probe" in _lib.LIB_PATH, "set VLLM_AMD_LIB to the probe build"
dev = torch.device("cuda", 0)
S, ctx, H, KVH, D, BS = 64, 4096, 32, 8, 128, 16
q, caches, bt, sl, one = bench.make_attention_inputs(S, ctx, H, KVH, D, BS, True, dev, n_caches=2)
g = torch.Generator().manual_seed(0)
lens = torch.randint(512, ctx + 1, (S,), generator=g).to(torch.int32)
for name, sl_ in (("uniform", sl), ("ragged", lens.to(dev))):
    mx = int(sl_.max())
    for it in range(3):
        PagedAttention.forward_decode(q, caches[0][0], caches[0][1], bt, sl_, mx, "fp8", KVH, D ** -0.5, None, one, one)
    torch.cuda.synchronize()
    nwg = S * KVH
    buf = (ctypes.c_ulonglong * (nwg * 4))()
    assert _lib.lib.vllm_amd_probe_stamps(buf, nwg) == 0
    t = torch.tensor(list(buf), dtype=torch.int64).view(nwg, 4)
    hw = t[:, 1]
    xcc = (hw >> 32) & 0xf
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print(f"== {name}: distinct CUs {key.unique().numel()}")
    print("  xcc of ids 0..15:", xcc[:16].tolist())
    print("  (se,sh,cu) of ids 0,8,16,..,248 (XCD 0):", [(int(se[i]), int(sh[i]), int(cu[i])) for i in range(0, 256, 8)])
    same = (key[:256] == key[256:512]).float().mean().item()
    print(f"  ids i and i+256 on the same CU: {100 * same:.1f} %")
    per_cu = {}
    for i in range(nwg):
        per_cu.setdefault(int(key[i]), []).append(i)
    sizes = [len(v) for v in per_cu.values()]
    print("  workgroups per CU: min", min(sizes), "max", max(sizes))
    tt = (t[:, [0, 3]] - t[:, 0].min()).double() / 100.0
    if name == "ragged":
        rows = []
        for k, ids in per_cu.items():
            rows.append((max(tt[i, 1].item() for i in ids), [int(sl_[i // KVH]) for i in ids]))
        rows.sort(reverse=True)
        print("  slowest CUs (exit us, lens of their workgroups):", rows[:6])
        print("  fastest CUs:", rows[-4:])
        print(f"  kernel span {tt[:, 1].max():.1f} us")

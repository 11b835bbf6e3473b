#!/usr/bin/env python3
"""Debug aid: unified_attention through the 256-row kernel ("pf_fa" = 2) against the packed 64-row kernel
("pf_fa" = 1) on one case; prints where they differ."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import make_unified_case
from vllm_amd import _lib
from vllm_amd.attention.ops import unified_attention

seqs = eval(sys.argv[1]) if len(sys.argv) > 1 else [(700, 700), (300, 1500), (64, 64), (1, 1)]
H, KVH = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8, 2)
fp8 = len(sys.argv) > 4 and sys.argv[4] == "fp8"
c = make_unified_case(seqs, H, KVH, 128, 16, torch.bfloat16, fp8=fp8, seed=7, kv_scale=0.5)
outs = []
sc = c["scale"].cuda() if fp8 else None
for knob in (1, 2):
    _lib.set_tuning("pf_fa", knob)
    out = torch.zeros(c["q"].shape, dtype=torch.bfloat16, device="cuda")
    unified_attention(c["q"].cuda(), c["k"].cuda(), c["v"].cuda(), out, c["cu_seqlens_q"].cuda(), c["max_seqlen_q"],
                      c["seqused_k"].cuda(), c["max_seqlen_k"], 128 ** -0.5, True, (-1, -1), c["block_table"].cuda(),
                      0.0, None, sc, sc)
    torch.cuda.synchronize()
    outs.append(out.float().cpu())
d = (outs[0] - outs[1]).abs()
bad = d > 2e-2
print("bad elements", int(bad.sum()), "of", bad.numel(), "max", float(d.max()), "nan", int(torch.isnan(outs[1]).sum()))
cu = c["cu_seqlens_q"].tolist()
for i in range(len(seqs)):
    b = bad[cu[i]:cu[i + 1]]
    if b.numel() == 0:
        continue
    toks = b.any(dim=2).any(dim=1).nonzero().flatten().tolist()
    print(f"seq {i} {seqs[i]}: bad tokens {len(toks)} of {cu[i+1]-cu[i]}: first {toks[:12]} last {toks[-5:]}")
    print("   per head:", b.sum(dim=(0, 2)).tolist())
    print("   per d (first 40):", b.sum(dim=(0, 1)).tolist()[:40])
if len(seqs) == 1 and seqs[0][0] <= 4:
    bt = c["block_table"][0]
    torch.set_printoptions(linewidth=200, precision=3)
    for tk in range(seqs[0][0]):
        for h in range(min(H, 2)):
            r = (outs[1][tk, h] / outs[0][tk, h])
            print("tok", tk, "head", h, "ratio new/old: min %.3f max %.3f" % (float(r.min()), float(r.max())), "old[:4]", outs[0][tk, h, :4].tolist(), "new[:4]", outs[1][tk, h, :4].tolist())
    print("v0 ", c["v"][int(bt[0]), 0, 0, :32].float() * (0.5 if fp8 else 1))
    print("v1 ", c["v"][int(bt[0]), 1, 0, :32].float() * (0.5 if fp8 else 1))

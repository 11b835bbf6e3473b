#!/usr/bin/env python3
"""A/B timing of paged-attention kernel variants in ONE process, interleaved
rounds (cdna_hip_programming.md rule 24).  Prints median / min us and GB/s per
(config, variant).  usage: python tools/ab_attention.py [--rounds 7] [--cfg C2,C4,C5]"""
import argparse, json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vllm_amd import _lib
from vllm_amd.attention import PagedAttention

CFGS = {
    "C2": dict(S=64, ctx=4096, H=32, KVH=8, D=128, BS=16, kv="fp8"),
    "C2h": dict(S=64, ctx=2048, H=32, KVH=8, D=128, BS=16, kv="fp8"),
    "C2d": dict(S=64, ctx=8192, H=32, KVH=8, D=128, BS=16, kv="fp8"),
    "C2x4": dict(S=256, ctx=4096, H=32, KVH=8, D=128, BS=16, kv="fp8"),
    "C4v1": dict(S=1024, ctx=8192, H=8, KVH=1, D=128, BS=16, kv="fp8"),
    "C2bf16": dict(S=64, ctx=4096, H=32, KVH=8, D=128, BS=16, kv="auto"),
    "C4": dict(S=256, ctx=32768, H=8, KVH=1, D=128, BS=16, kv="fp8"),
    "C5": dict(S=8, ctx=131072, H=64, KVH=8, D=128, BS=16, kv="fp8"),
    "C2d96": dict(S=64, ctx=4096, H=32, KVH=8, D=96, BS=16, kv="fp8"),   # padded head size
    "C2d80bf16": dict(S=64, ctx=4096, H=32, KVH=8, D=80, BS=16, kv="auto"),
    "S20": dict(S=20, ctx=4096, H=32, KVH=8, D=128, BS=16, kv="fp8"),   # v1 by the reference's rule, < 512 workgroups
    "S32": dict(S=32, ctx=4096, H=32, KVH=8, D=128, BS=16, kv="fp8"),
    "S48": dict(S=48, ctx=4096, H=32, KVH=8, D=128, BS=16, kv="fp8"),
    "S32d": dict(S=32, ctx=8192, H=32, KVH=8, D=128, BS=16, kv="fp8"),
    "S32h": dict(S=32, ctx=1024, H=32, KVH=8, D=128, BS=16, kv="fp8"),
    "S48bf16": dict(S=48, ctx=4096, H=32, KVH=8, D=128, BS=16, kv="auto"),
    "C5tp8": dict(S=8, ctx=131072, H=8, KVH=1, D=128, BS=16, kv="fp8"),
}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cfg", default="C2")
    ap.add_argument("--variants", default="0,1,2,3")
    ap.add_argument("--ppw", default="0", help="comma list of partitions-per-wave overrides (0 = heuristic)")
    ap.add_argument("--v2", action="store_true", help="force paged_attention_v2")
    ap.add_argument("--rocm", action="store_true", help="use paged_attention_rocm (partition 256)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    variants = [int(v) for v in a.variants.split(",")]
    for name in a.cfg.split(","):
        c = CFGS[name]
        fp8 = c["kv"] == "fp8"
        q, caches, bt, sl, one = bench.make_attention_inputs(c["S"], c["ctx"], c["H"], c["KVH"], c["D"], c["BS"], fp8, dev,
                                                             n_caches=2 if c["S"] * c["ctx"] < 2**22 else 1)
        nbytes = bench.attention_bytes(c["S"], c["ctx"], c["H"], c["KVH"], c["D"], c["BS"], 1 if fp8 else 2)
        st = {"i": 0}
        def step():
            kc, vc = caches[st["i"] % len(caches)]; st["i"] += 1
            if a.v2:
                P = (c["ctx"] + 511) // 512
                out = torch.empty_like(q)
                tmp = torch.empty((c["S"], c["H"], P, c["D"]), dtype=q.dtype, device=dev)
                es = torch.empty((c["S"], c["H"], P), dtype=torch.float32, device=dev); ml = torch.empty_like(es)
                from vllm_amd import _custom_ops as ops
                ops.paged_attention_v2(out, es, ml, tmp, q, kc, vc, c["KVH"], c["D"] ** -0.5, bt, sl, c["BS"], c["ctx"], None, c["kv"], one, one)
                return out
            if a.rocm:
                return PagedAttention.forward_decode_rocm(q, kc, vc, bt, sl, c["ctx"], c["kv"], c["KVH"], c["D"] ** -0.5, None, one, one)
            return PagedAttention.forward_decode(q, kc, vc, bt, sl, c["ctx"], c["kv"], c["KVH"], c["D"] ** -0.5, None, one, one)
        ppws = [int(x) for x in a.ppw.split(",")]
        variants_ = [(v, w) for v in variants for w in ppws]
        times = {v: [] for v in variants_}
        for r in range(a.rounds):
            for v in variants_:
                _lib.set_tuning("pa_variant", v[0] & 3)
                _lib.set_tuning("pa_group", v[0] >> 3)   # + 8: XCD-grouped order always, + 16: never
                _lib.set_tuning("pa_sp", v[1])
                _, ev = bench.time_gpu(step, a.iters, 3)
                times[v].append(ev / a.iters * 1e6)
        for v in variants_:
            med, mn = statistics.median(times[v]), min(times[v])
            print(json.dumps(dict(cfg=name, rocm=a.rocm, variant=v[0], ppw=v[1], med_us=round(med, 2), min_us=round(mn, 2),
                                  med_gbps=round(nbytes / med / 1e3, 1), max_gbps=round(nbytes / mn / 1e3, 1))), flush=True)
        del q, caches, bt, sl

if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- headline measurement of the MI355X quantized decode hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-extra]

One "step" = one pass of the hot path over one batch of synthetic input: a
paged-attention decode call (through torch.ops -> C ABI -> HIP) on
BASELINE.json configs[1]: Llama-3-8B shape, fp8 e4m3 KV with per-tensor scales,
batch 64, ctx 4096, 32 q / 8 kv heads, head 128, block 16, bf16 query/output.
Inputs are resident in HBM before the timed region.  `value` = algorithmic bytes
(K and V once + q/out + block tables + seq_lens) / time.
Steady state: a fresh process runs its first ~180 decode launches (~15 ms) 3-4 % slower than every
launch after them (tools/decode_drift.py, profiles/r02_decode_drift.txt), so ~100 ms of untimed
launches precede the W warmup steps (`roofline.steady_state`); the graph-timed legs of `extra`
replay ~25 ms before their timed replays for the same reason.

N > 1: one rank per GPU, either launched by torch.distributed.run (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the env) or, when WORLD_SIZE is not set, by this script itself:
`python bench.py --gpus N` spawns its N ranks before it touches a GPU and exits non-zero if
any rank fails or the RCCL world is not N.  Attention shards by KV head with no collective
(SURVEY.md 8e): the headline keeps the per-GPU configs[1] shape on every rank ("weak" scaling);
value = bytes of all ranks / max time.  `extra` then carries, timed over all ranks (max):
  configs4_strong   configs[4] (8 x 128k, 64q/8kv, fp8) with heads / N per rank: strong scaling
  configs3_per_gpu  configs[3] per-GPU attention shape (256 x 32k, 8q/1kv) on every rank
  custom_all_reduce peer-to-peer all-reduce vs RCCL, 256 KiB / 4 MiB (result-checked first; VLLM_AMD_BENCH_CUSTOM_AR=0 skips)
  row_parallel_tp   configs[3] row-parallel fp8 GEMMs (K 1024 / 3584 -> 8192, M 256) + one 4 MiB
                    RCCL all-reduce (us, bus GB/s)

The JSON line also carries
  roofline      dominant kernel vs the 8 TB/s HBM3E peak, HIP-event timed
  cpu_baseline  the reference's own CPU paged_attention_v1 (oracle/_ref, built
                from /root/reference/csrc/cpu by oracle/Makefile) on configs[0],
                timed on the host cores of this box; falls back to our C port
  extra         fp8 scaled_mm TFLOP/s at Llama-3-8B shapes (configs[2]), cache
                write / quant / fused-neighbour GB/s, and (N>1) row-parallel GEMM +
                RCCL all-reduce
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP8_DENSE_PEAK_TFLOPS = 5000.0  # MI355X_MICROARCH.md: fp8 MFMA ~5 PF dense (block-scaled form)


def make_attention_inputs(S, ctx, H, KVH, D, BS, kv_fp8, device, seed=0, n_caches=2):
    """Synthetic decode batch: q ~ U(-d^-.5, d^-.5) bf16, KV of the
    same distribution quantised to e4m3 with scale 1.0, all seq_lens = ctx,
    distinct random physical blocks."""
    from vllm_amd import _custom_ops as ops
    g = torch.Generator(device=device).manual_seed(seed)
    sc = D ** -0.5
    nb_seq = (ctx + BS - 1) // BS
    NB = S * nb_seq
    x = 16 if kv_fp8 else 8
    caches = []
    for _ in range(n_caches):
        kc16 = (torch.rand((NB, KVH, D // x, BS, x), generator=g, device=device) * 2 - 1) * sc
        vc16 = (torch.rand((NB, KVH, D, BS), generator=g, device=device) * 2 - 1) * sc
        if kv_fp8:
            kc = torch.empty(kc16.shape, dtype=torch.uint8, device=device)
            vc = torch.empty(vc16.shape, dtype=torch.uint8, device=device)
            ops.convert_fp8(kc, kc16.to(torch.bfloat16), 1.0, "fp8")
            ops.convert_fp8(vc, vc16.to(torch.bfloat16), 1.0, "fp8")
        else:
            kc, vc = kc16.to(torch.bfloat16), vc16.to(torch.bfloat16)
        del kc16, vc16
        caches.append((kc, vc))
    q = ((torch.rand((S, H, D), generator=g, device=device) * 2 - 1) * sc).to(torch.bfloat16)
    perm = torch.randperm(NB, generator=g, device=device).to(torch.int32)
    bt = perm.view(S, nb_seq).contiguous()
    sl = torch.full((S,), ctx, dtype=torch.int32, device=device)
    one = torch.ones(1, dtype=torch.float32, device=device)
    return q, caches, bt, sl, one


def attention_bytes(S, ctx, H, KVH, D, BS, kv_bytes):
    kv = S * KVH * ctx * D * 2 * kv_bytes
    qo = S * H * D * (2 + 2)
    tables = S * ((ctx + BS - 1) // BS) * 4 + 4 * S
    return kv + qo + tables


def time_gpu(fn, steps, warmup, dist=None):
    """W untimed + K timed calls bracketed by barrier + synchronize; returns
    (wall seconds, HIP-event seconds) for the K steps."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    return t1 - t0, e0.elapsed_time(e1) * 1e-3


PRECONDITION_MS = 100.0
GRAPH_WARM_MS = 25.0


def precondition(fn, ms=PRECONDITION_MS):
    """Untimed launches for ~`ms` of device time BEFORE the W warmup steps.  A freshly started
    process runs the first ~200 launches (~16 ms) of the headline kernel 3-4 % slower than every
    launch after them (tools/decode_drift.py: 84.6 .. 87.4 us per launch falling to 82.0 +- 0.3 us
    and staying there); a decode loop runs for seconds, so the timed K steps are taken from
    the steady state.  Returns the number of launches made."""
    total, n = 0.0, 0
    while total < ms and n < 20000:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        total += e0.elapsed_time(e1)
        n += 50
    return n


def time_gpu_graph(fn, iters, replays=5):
    """Per-call device time of a short op: `iters` calls captured in one HIP graph and
    replayed, so the host launch path (~15 us per eager call) is not what is measured."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    # steady state: replay for ~GRAPH_WARM_MS first (the large GEMM runs 66-83 us for its first few
    # hundred launches and 59 us after ~25 ms of continuous work: tools/gemm_drift.py)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    warmed = 0.0
    for _ in range(200):
        if warmed >= GRAPH_WARM_MS:
            break
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        warmed += max(e0.elapsed_time(e1), 0.05)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / (iters * replays)


def run_attention(cfg, steps, warmup, device, dist=None, version="auto"):
    from vllm_amd import _custom_ops as ops
    from vllm_amd.attention.paged_attn import PagedAttention
    S, ctx, H, KVH, D, BS = cfg["S"], cfg["ctx"], cfg["H"], cfg["KVH"], cfg["D"], cfg["BS"]
    fp8 = cfg["kv"] == "fp8"
    q, caches, bt, sl, one = make_attention_inputs(S, ctx, H, KVH, D, BS, fp8, device)
    kvd = "fp8" if fp8 else "auto"
    state = {"i": 0}

    def step():
        kc, vc = caches[state["i"] % len(caches)]
        state["i"] += 1
        return PagedAttention.forward_decode(q, kc, vc, bt, sl, ctx, kvd, KVH, D ** -0.5, None,
                                             one, one)

    # cold: exactly the driver's W warmups + K timed steps as the first launches of the process ...
    cold_wall, cold_ev = time_gpu(step, steps, warmup, dist)
    # ... and the steady state: the same W + K after ~100 ms of untimed launches
    pre = precondition(step)
    wall, ev = time_gpu(step, steps, warmup, dist)
    nbytes = attention_bytes(S, ctx, H, KVH, D, BS, 1 if fp8 else 2)
    return dict(wall=wall, event=ev, bytes=nbytes, precondition_launches=pre, cold_wall=cold_wall,
                cold_event=cold_ev)


def run_attention_shape(name, S, ctx, H, KVH, device, dist=None, world=1, iters=20, D=128, BS=16):
    """One fp8-KV decode shape through PagedAttention.forward_decode, HIP-event timed (max over
    ranks when distributed); bytes = K and V once + q / out + tables (per rank)."""
    from vllm_amd.attention.paged_attn import PagedAttention
    q, caches, bt, sl, one = make_attention_inputs(S, ctx, H, KVH, D, BS, True, device, n_caches=1)
    kc, vc = caches[0]
    fn = lambda: PagedAttention.forward_decode(q, kc, vc, bt, sl, ctx, "fp8", KVH, D ** -0.5, None,
                                               one, one)
    precondition(fn, 60.0)
    _, ev = time_gpu(fn, iters, 3, dist)
    if dist is not None:
        t = torch.tensor([ev], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ev = t.item()
    per = ev / iters
    nbytes = attention_bytes(S, ctx, H, KVH, D, BS, 1)
    del kc, vc, caches
    torch.cuda.empty_cache()
    return dict(shape=f"{name}: {S} seqs x ctx {ctx}, {H}q/{KVH}kv heads per GPU, fp8 KV",
                us=round(per * 1e6, 1), gbps_per_gpu=round(nbytes / per / 1e9, 1),
                gbps_total=round(world * nbytes / per / 1e9, 1),
                frac_of_hbm_peak=round(nbytes / per / 1e9 / HBM_PEAK_GBPS, 4))


def run_ragged_decode(device, iters=20):
    """configs[1] head shape with ragged lengths U[512, 4096] (seeded): bytes = what the
    sequences actually hold."""
    from vllm_amd.attention.paged_attn import PagedAttention
    S, ctx, H, KVH, D, BS = 64, 4096, 32, 8, 128, 16
    q, caches, bt, sl, one = make_attention_inputs(S, ctx, H, KVH, D, BS, True, device, n_caches=2)
    g = torch.Generator().manual_seed(0)
    lens = torch.randint(512, ctx + 1, (S,), generator=g).to(torch.int32)
    sl = lens.to(device)
    max_len = int(lens.max())
    state = {"i": 0}

    def fn():
        kc, vc = caches[state["i"] % 2]
        state["i"] += 1
        return PagedAttention.forward_decode(q, kc, vc, bt, sl, max_len, "fp8", KVH, D ** -0.5, None,
                                             one, one)
    precondition(fn, 60.0)
    _, ev = time_gpu(fn, iters, 3)
    per = ev / iters
    nbytes = int(lens.sum()) * KVH * D * 2 + S * H * D * 4 + int(((lens + BS - 1) // BS).sum()) * 4 + 4 * S
    return dict(shape="64 seqs x U[512,4096] tokens, 32q/8kv, fp8 KV", us=round(per * 1e6, 1),
                gbps=round(nbytes / per / 1e9, 1),
                frac_of_hbm_peak=round(nbytes / per / 1e9 / HBM_PEAK_GBPS, 4))


MALL_BYTES = 256 << 20          # Infinity Cache (MI355X_MICROARCH.md)


def run_scaled_mm(device, shapes, Ms, iters=20, rotate=True):
    """configs[2] grid.  The weight operand ROTATES through enough copies that together they exceed the
    256 MiB Infinity Cache (every (K, N) here is <= 117 MB: a single replayed copy would be served from
    that cache and the small-M "GB/s" would not be HBM numbers), so each launch streams its
    weights from HBM.  Activations, scales and the output are one buffer each (L2 / cache resident, as
    they are behind the producing kernel in a model)."""
    from vllm_amd import _custom_ops as ops
    res = []
    for (K, N) in shapes:
        copies = max(3, -(-int(1.25 * MALL_BYTES) // (K * N))) if rotate else 1
        copies = int(os.environ.get("VLLM_AMD_BENCH_WEIGHT_COPIES", copies))   # (experiments)
        wqs = []
        for c in range(copies):
            w = torch.randn((N, K), device=device, dtype=torch.bfloat16)
            wq, ws = ops.scaled_fp8_quant(w, use_per_token_if_dynamic=True)
            wqs.append((wq.t(), ws))
            del w
        n_calls = -(-iters // copies) * copies      # whole turns of the rotation per graph
        for M in Ms:
            a = torch.randn((M, K), device=device, dtype=torch.bfloat16)
            aq, as_ = ops.scaled_fp8_quant(a, use_per_token_if_dynamic=True)
            out = torch.empty((M, N), dtype=torch.bfloat16, device=device)
            state = {"i": 0}

            def fn():
                wt, ws = wqs[state["i"] % copies]
                state["i"] += 1
                torch.ops._C.cutlass_scaled_mm(out, aq, wt, as_, ws, None)
            state["i"] = 0
            t = time_gpu_graph(fn, n_calls)
            flops = 2.0 * M * N * K
            nbytes = M * K + K * N + 2 * M * N + 4 * (M + N)
            bound = min(FP8_DENSE_PEAK_TFLOPS, HBM_PEAK_GBPS * 1e9 * (flops / nbytes) / 1e12)
            row = dict(M=M, K=K, N=N, us=round(t * 1e6, 2), tflops=round(flops / t / 1e12, 2),
                       gbps=round(nbytes / t / 1e9, 1), roof_tflops=round(bound, 1),
                       frac=round(flops / t / 1e12 / bound, 4), weight_copies=copies)
            if rotate and M >= 4096:
                # the same launch with ONE weight buffer (Infinity-Cache resident, as rounds 1-2 timed it):
                # what the kernel does when the weight stream costs nothing
                wt0, ws0 = wqs[0]
                t1 = time_gpu_graph(lambda: torch.ops._C.cutlass_scaled_mm(out, aq, wt0, as_, ws0, None), iters)
                row["resident_weights_us"] = round(t1 * 1e6, 2)
                row["resident_weights_tflops"] = round(flops / t1 / 1e12, 2)
            res.append(row)
        del wqs
        torch.cuda.empty_cache()
    return res


def run_mfma_ceiling(device):
    """What the matrix pipes deliver with nothing else to do: a loop of the GEMM's own fp8 MFMA on random
    operands held in registers, 8 waves per CU (vllm_amd_mfma_ceiling; tools/mfmabench2.hip).  The 5 PF paper
    peak assumes 2.4 GHz; under load the chip holds 1.7-2.1 GHz."""
    from vllm_amd import _lib
    sink = torch.empty(256 * 512, dtype=torch.float32, device=device)
    iters = 4000
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _lib.check(_lib.lib.vllm_amd_mfma_ceiling(sink.data_ptr(), iters, st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        _lib.check(_lib.lib.vllm_amd_mfma_ceiling(sink.data_ptr(), iters, st))
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / 5
    flops = 256.0 * 8 * iters * 32 * 2 * 16 * 16 * 128
    return round(flops / t / 1e12, 1)


def run_cache_quant(device, iters=20):
    from vllm_amd import _custom_ops as ops
    out = {}
    # decode-step cache write at configs[1]: T=64 tokens ... plus a prefill-sized one
    # (prefill: 4 sequences of T/4 consecutive slots over scattered blocks, starting 5 slots
    # into a block as after a chunk boundary; "scattered": every token in a different block)
    # ("prefill_aligned": sequences that start on a block boundary -- a fresh prefill; "prefill": 5 slots into
    # a block, as after a chunk boundary)
    for name, T in (("decode_T64", 64), ("prefill_T8192", 8192), ("prefill_T32768", 32768),
                    ("prefill_aligned_T8192", 8192), ("prefill_aligned_T32768", 32768),
                    ("scattered_T8192", 8192)):
        KVH, D, BS = 8, 128, 16
        NB = max(T // BS * 2, 64)
        k = torch.randn((T, KVH, D), device=device, dtype=torch.bfloat16)
        v = torch.randn_like(k)
        kc = torch.zeros((NB, KVH, D // 16, BS, 16), dtype=torch.uint8, device=device)
        vc = torch.zeros((NB, KVH, D, BS), dtype=torch.uint8, device=device)
        if name.startswith("prefill"):
            t = torch.arange(T, device=device)
            seq, pos = t // (T // 4), t % (T // 4) + (0 if "aligned" in name else 5)
            per_seq = (T // 4 + 5 + BS - 1) // BS
            perm = torch.randperm(NB, device=device)
            slots = (perm[seq * per_seq + pos // BS] * BS + pos % BS).to(torch.int64)
        else:
            slots = torch.randperm(NB * BS, device=device)[:T].to(torch.int64)
        sc = torch.tensor([0.05], device=device)
        fn = lambda: ops.reshape_and_cache(k, v, kc, vc, slots, "fp8", sc, sc)
        t = time_gpu_graph(fn, iters)
        nbytes = T * KVH * D * (2 * 2 + 2 * 1) + 8 * T
        out["reshape_and_cache_fp8_" + name] = dict(us=round(t * 1e6, 2),
                                                    gbps=round(nbytes / t / 1e9, 1))
    # the V1 engine's cache write: flash layout [NB, BS, KVH, D]
    for name, T in (("decode_T64", 64), ("prefill_T8192", 8192), ("prefill_T32768", 32768)):
        KVH, D, BS = 8, 128, 16
        NB = max(T // BS * 2, 64)
        k = torch.randn((T, KVH, D), device=device, dtype=torch.bfloat16)
        v = torch.randn_like(k)
        kc = torch.zeros((NB, BS, KVH, D), dtype=torch.uint8, device=device)
        vc = torch.zeros_like(kc)
        slots = torch.randperm(NB * BS, device=device)[:T].to(torch.int64)
        sc = torch.tensor([0.05], device=device)
        fn = lambda: ops.reshape_and_cache_flash(k, v, kc, vc, slots, "fp8", sc, sc)
        t = time_gpu_graph(fn, iters)
        nbytes = T * KVH * D * (2 * 2 + 2 * 1) + 8 * T
        out["reshape_and_cache_flash_fp8_" + name] = dict(us=round(t * 1e6, 2),
                                                          gbps=round(nbytes / t / 1e9, 1))
    for M, K in ((64, 4096), (4096, 4096), (4096, 14336)):
        x = torch.randn((M, K), device=device, dtype=torch.bfloat16)
        q = torch.empty((M, K), dtype=torch.float8_e4m3fn, device=device)
        sc_ = torch.empty((M, 1), dtype=torch.float32, device=device)
        fn = lambda: torch.ops._C.dynamic_per_token_scaled_fp8_quant(q, x, sc_, None)
        t = time_gpu_graph(fn, iters)
        nbytes = M * K * 3 + 4 * M
        out[f"per_token_quant_M{M}_K{K}"] = dict(us=round(t * 1e6, 2),
                                                 gbps=round(nbytes / t / 1e9, 1))
    return out


def run_fused_neighbours(device, iters=20):
    """fused RMSNorm + quant, SiLU-mul + quant, slot mapping.  Algorithmic
    bytes = every input and output element once (the reference's kernels read the row 2-3x)."""
    from vllm_amd import _custom_ops as ops, _torch_ops
    out = {}
    for M, H in ((64, 4096), (4096, 4096), (4096, 8192)):
        x = torch.randn((M, H), device=device, dtype=torch.bfloat16)
        res = torch.randn_like(x)
        w = torch.ones(H, device=device, dtype=torch.bfloat16)
        q = torch.empty((M, H), dtype=torch.float8_e4m3fn, device=device)
        sc = torch.empty((M, 1), dtype=torch.float32, device=device)
        one = torch.tensor([0.05], device=device)
        legs = {
            "rms_norm_dynamic_per_token_quant":
                (lambda: torch.ops._C.rms_norm_dynamic_per_token_quant(q, x, w, sc, 1e-6, None, None),
                 M * H * 3 + H * 2 + 4 * M),
            "rms_norm_dynamic_per_token_quant_residual":
                (lambda: torch.ops._C.rms_norm_dynamic_per_token_quant(q, x, w, sc, 1e-6, None, res),
                 M * H * 7 + H * 2 + 4 * M),
            "fused_add_rms_norm_static_fp8_quant":
                (lambda: torch.ops._C.fused_add_rms_norm_static_fp8_quant(q, x, res, w, one, 1e-6),
                 M * H * 7 + H * 2),
        }
        for name, (fn, nbytes) in legs.items():
            t = time_gpu_graph(fn, iters)
            out[f"{name}_M{M}_H{H}"] = dict(us=round(t * 1e6, 2), gbps=round(nbytes / t / 1e9, 1))
    for M, d in ((64, 14336), (4096, 14336)):
        x = torch.randn((M, 2 * d), device=device, dtype=torch.bfloat16)
        q = torch.empty((M, d), dtype=torch.float8_e4m3fn, device=device)
        one = torch.tensor([0.05], device=device)
        t = time_gpu_graph(lambda: torch.ops._C.silu_and_mul_quant(q, x, one), iters)
        out[f"silu_and_mul_quant_M{M}_d{d}"] = dict(us=round(t * 1e6, 2),
                                                    gbps=round(M * d * 5 / t / 1e9, 1))
    R, MB, BS, T = 256, 2048, 16, 8192
    bt = torch.randint(0, R * MB, (R, MB), dtype=torch.int32, device=device)
    req = torch.randint(0, R, (T,), dtype=torch.int32, device=device)
    pos = torch.randint(0, MB * BS, (T,), dtype=torch.int64, device=device)
    slots = torch.empty(T, dtype=torch.int64, device=device)
    t = time_gpu_graph(lambda: _torch_ops.compute_slot_mapping(slots, bt, req, pos, BS), iters)
    out[f"compute_slot_mapping_T{T}"] = dict(us=round(t * 1e6, 2))
    return out


def run_prefill(device, iters=5):
    """context_attention_fwd (chunked prefill over a paged fp8 prefix) at the
    Llama-3-8B head shape; flops = 4 D H sum_seq q_len (ctx + (q_len + 1) / 2) against the bf16
    dense MFMA peak.  Both ops take the 256-row kernel (prefill_fa.hip) at these sizes."""
    from vllm_amd.attention.ops import context_attention_fwd
    H, KVH, D, BS = 32, 8, 128, 16
    out = {}
    for (B, ql, cl) in ((4, 2048, 2048), (1, 8192, 0), (64, 128, 4096)):
        T = B * ql
        g = torch.Generator(device=device).manual_seed(0)
        q = (torch.rand((T, H, D), device=device, generator=g) - 0.5).to(torch.bfloat16)
        k = (torch.rand((T, KVH, D), device=device, generator=g) - 0.5).to(torch.bfloat16)
        v = (torch.rand((T, KVH, D), device=device, generator=g) - 0.5).to(torch.bfloat16)
        o = torch.empty_like(q)
        nb = (cl + BS - 1) // BS + 1
        kc = torch.randint(0, 120, (B * nb, KVH, D // 16, BS, 16), device=device, dtype=torch.uint8)
        vc = torch.randint(0, 120, (B * nb, KVH, D, BS), device=device, dtype=torch.uint8)
        b_loc = torch.randperm(B * nb, device=device).to(torch.int32).view(B, nb)
        start = (torch.arange(B + 1, device=device) * ql).to(torch.int32)
        seql = torch.full((B,), ql + cl, dtype=torch.int32, device=device)
        sc = torch.full((1,), 0.02, device=device)
        fn = lambda: context_attention_fwd(q, k, v, o, "fp8", kc, vc, b_loc, start, seql, ql + cl,
                                           ql, sc, sc)
        t = time_gpu_graph(fn, iters)
        flops = 4.0 * D * H * B * ql * (cl + (ql + 1) / 2)
        out[f"B{B}_q{ql}_ctx{cl}_fp8kv"] = dict(us=round(t * 1e6, 1),
                                                tflops=round(flops / t / 1e12, 1),
                                                frac_of_bf16_peak=round(flops / t / 2.5e15, 4))
        # the same problem through unified_attention (everything in a flash-layout fp8 cache)
        from vllm_amd.attention.ops import unified_attention
        nbu = (ql + cl + BS - 1) // BS
        kf = torch.randint(0, 120, (B * nbu, BS, KVH, D), device=device, dtype=torch.uint8)
        vf = torch.randint(0, 120, (B * nbu, BS, KVH, D), device=device, dtype=torch.uint8)
        bt = torch.randperm(B * nbu, device=device).to(torch.int32).view(B, nbu)
        fn = lambda: unified_attention(q, kf, vf, o, start, ql, seql, ql + cl, D ** -0.5, True,
                                       (-1, -1), bt, 0.0, None, sc, sc)
        t = time_gpu_graph(fn, iters)
        out[f"unified_B{B}_q{ql}_ctx{cl}_fp8kv"] = dict(us=round(t * 1e6, 1),
                                                        tflops=round(flops / t / 1e12, 1),
                                                        frac_of_bf16_peak=round(flops / t / 2.5e15, 4))
        del kf, vf
    return out


def run_unified_decode(device, iters=5):
    """Decode-only batches through unified_attention (the V1 engine's op, flash-layout fp8
    cache, unified_decode.hip) at the configs[1] head shape; bytes = K and V read once."""
    from vllm_amd.attention.ops import unified_attention
    H, KVH, D, BS = 32, 8, 128, 16
    out = {}
    # hnd: the cache memory is head-major (what TritonAttentionBackend.get_kv_cache_stride_order
    # asks the engine for by default), passed as a permuted view of the reference's NHD shape
    for S, ctx, hnd in ((64, 4096, False), (64, 4096, True), (256, 4096, False), (256, 4096, True),
                        (16, 16384, True), (1, 4096, True)):
        nb = (ctx + BS - 1) // BS
        q = (torch.rand((S, H, D), device=device) - 0.5).to(torch.bfloat16)
        o = torch.empty_like(q)
        kf = torch.randint(0, 120, (S * nb, BS, KVH, D), device=device, dtype=torch.uint8)
        vf = torch.randint(0, 120, (S * nb, BS, KVH, D), device=device, dtype=torch.uint8)
        if hnd:
            kf = kf.view(S * nb, KVH, BS, D).permute(0, 2, 1, 3)
            vf = vf.view(S * nb, KVH, BS, D).permute(0, 2, 1, 3)
        bt = torch.randperm(S * nb, device=device).to(torch.int32).view(S, nb)
        cu = torch.arange(S + 1, device=device, dtype=torch.int32)
        sl = torch.full((S,), ctx, dtype=torch.int32, device=device)
        sc = torch.full((1,), 0.02, device=device)
        fn = lambda: unified_attention(q, kf, vf, o, cu, 1, sl, ctx, D ** -0.5, True, (-1, -1), bt,
                                       0.0, None, sc, sc)
        t = time_gpu_graph(fn, iters)
        nbytes = 2 * S * ctx * KVH * D + 2 * S * H * D * 2 + S * nb * 4
        out[f"S{S}_ctx{ctx}_fp8kv_{'hnd' if hnd else 'nhd'}"] = dict(us=round(t * 1e6, 1), gbps=round(nbytes / t / 1e9, 1),
                                           frac_of_hbm_peak=round(nbytes / t / 1e9 / HBM_PEAK_GBPS, 4))
        del kf, vf
    return out


def run_wide_heads(device, iters=5):
    """Head size 256 (the reference's largest: Gemma-class models) through the three attention ops, fp8 cache,
    16 q / 8 kv heads: V0 decode (x layout), V1 decode (flash layout, head-major memory) and a V1 prefill chunk."""
    from vllm_amd.attention.ops import unified_attention
    H, KVH, D, BS = 16, 8, 256, 16
    out = {}
    r = run_attention_shape("head 256", 64, 4096, H, KVH, device, D=D)
    out["paged_attention_S64_ctx4096_fp8kv"] = {k: r[k] for k in ("us", "gbps_total", "frac_of_hbm_peak")}
    sc = torch.full((1,), 0.02, device=device)
    S, ctx = 64, 4096
    nb = ctx // BS
    q = (torch.rand((S, H, D), device=device) - 0.5).to(torch.bfloat16)
    o = torch.empty_like(q)
    kf = torch.randint(0, 120, (S * nb, KVH, BS, D), device=device, dtype=torch.uint8).permute(0, 2, 1, 3)
    vf = torch.randint(0, 120, (S * nb, KVH, BS, D), device=device, dtype=torch.uint8).permute(0, 2, 1, 3)
    bt = torch.randperm(S * nb, device=device).to(torch.int32).view(S, nb)
    cu = torch.arange(S + 1, device=device, dtype=torch.int32)
    sl = torch.full((S,), ctx, dtype=torch.int32, device=device)
    t = time_gpu_graph(lambda: unified_attention(q, kf, vf, o, cu, 1, sl, ctx, D ** -0.5, True, (-1, -1), bt,
                                                 0.0, None, sc, sc), iters)
    nbytes = 2 * S * ctx * KVH * D + 2 * S * H * D * 2 + S * nb * 4
    out["unified_decode_S64_ctx4096_fp8kv_hnd"] = dict(us=round(t * 1e6, 1), gbps=round(nbytes / t / 1e9, 1),
                                                       frac_of_hbm_peak=round(nbytes / t / 1e9 / HBM_PEAK_GBPS, 4))
    del kf, vf
    B, ql = 1, 8192
    nbu = ql // BS
    q = (torch.rand((ql, H, D), device=device) - 0.5).to(torch.bfloat16)
    o = torch.empty_like(q)
    kf = torch.randint(0, 120, (nbu, BS, KVH, D), device=device, dtype=torch.uint8)
    vf = torch.randint(0, 120, (nbu, BS, KVH, D), device=device, dtype=torch.uint8)
    bt = torch.randperm(nbu, device=device).to(torch.int32).view(1, nbu)
    start = torch.tensor([0, ql], device=device, dtype=torch.int32)
    seql = torch.tensor([ql], device=device, dtype=torch.int32)
    t = time_gpu_graph(lambda: unified_attention(q, kf, vf, o, start, ql, seql, ql, D ** -0.5, True, (-1, -1), bt,
                                                 0.0, None, sc, sc), iters)
    flops = 4.0 * D * H * ql * (ql + 1) / 2
    out["unified_prefill_B1_q8192_fp8kv"] = dict(us=round(t * 1e6, 1), tflops=round(flops / t / 1e12, 1),
                                                 frac_of_bf16_peak=round(flops / t / 2.5e15, 4))
    return out


def cpu_baseline():
    """Reference CPU paged_attention_v1 on BASELINE.json configs[0] (32 seqs, ctx 1024,
    32q/8kv, d 128, block 16, bf16 KV), all host cores available to this process."""
    from oracle import oracle as O
    S, ctx, H, KVH, D, BS = 32, 1024, 32, 8, 128, 16
    g = torch.Generator().manual_seed(0)
    sc = D ** -0.5
    nb_seq = ctx // BS
    NB = S * nb_seq
    kc = ((torch.rand((NB, KVH, D // 8, BS, 8), generator=g) * 2 - 1) * sc).to(torch.bfloat16)
    vc = ((torch.rand((NB, KVH, D, BS), generator=g) * 2 - 1) * sc).to(torch.bfloat16)
    q = ((torch.rand((S, H, D), generator=g) * 2 - 1) * sc).to(torch.bfloat16)
    bt = torch.randperm(NB, generator=g).to(torch.int32).view(S, nb_seq).contiguous()
    sl = torch.full((S,), ctx, dtype=torch.int32)
    one = torch.ones(1)
    out = torch.empty_like(q)
    threads = torch.get_num_threads()
    if O.load_ref():
        kind = "reference"
        fn = lambda: torch.ops.refcpu.paged_attention_v1(out, q, kc, vc, KVH, sc, bt, sl, BS, ctx,
                                                         None, "auto", one, one, 0, 0, 0, 64, 0)
    else:
        kind = "port"
        fn = lambda: O.paged_attention_v1(out, q, kc, vc, KVH, sc, bt, sl, BS, ctx, None, "auto",
                                          1.0, 1.0)
    fn()
    t0 = time.perf_counter()
    n = 0
    while True:
        fn()
        n += 1
        el = time.perf_counter() - t0
        if el > 10.0 or n >= 100000:
            break
    per = el / n
    nbytes = attention_bytes(S, ctx, H, KVH, D, BS, 2)
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count()
    omp = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    used = omp if omp > 0 else min(threads, cores)
    return dict(value=round(nbytes / per / 1e9, 2), unit="GB/s", cores=used, kind=kind,
                ms_per_iter=round(per * 1e3, 3),
                sample=f"configs[0]: paged_attention_v1 CPU, 32 seqs x ctx 1024, 32q/8kv, d128, "
                       f"block 16, bf16 KV; {n} iterations in {el:.1f} s on {used} OpenMP threads")


def cpu_baseline_port(budget_s=8.0):
    """The oracle's C restatement (fp8-capable, unlike the reference's CPU kernel) on a bounded
    sample of the HEADLINE workload: 8 of the 64 sequences of configs[1] (fp8 KV, ctx 4096)."""
    from oracle import oracle as O
    S, ctx, H, KVH, D, BS = 8, 4096, 32, 8, 128, 16
    g = torch.Generator().manual_seed(0)
    nb = ctx // BS
    NB = S * nb
    kc = torch.randint(0, 120, (NB, KVH, D // 16, BS, 16), generator=g, dtype=torch.uint8)
    vc = torch.randint(0, 120, (NB, KVH, D, BS), generator=g, dtype=torch.uint8)
    q = ((torch.rand((S, H, D), generator=g) * 2 - 1) * D ** -0.5).to(torch.bfloat16)
    bt = torch.randperm(NB, generator=g).to(torch.int32).view(S, nb).contiguous()
    sl = torch.full((S,), ctx, dtype=torch.int32)
    out = torch.empty_like(q)
    fn = lambda: O.paged_attention_v1(out, q, kc, vc, KVH, D ** -0.5, bt, sl, BS, ctx, None, "fp8",
                                      1.0, 1.0)
    fn()
    t0 = time.perf_counter()
    n = 0
    while True:
        fn()
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    per = el / n
    nbytes = attention_bytes(S, ctx, H, KVH, D, BS, 1)
    omp = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count()
    used = omp if omp > 0 else cores
    return dict(value=round(nbytes / per / 1e9, 3), unit="GB/s", cores=used, kind="port",
                ms_per_iter=round(per * 1e3, 1),
                sample=f"configs[1] shape, 8 of its 64 sequences (fp8 KV, ctx 4096, 32q/8kv, d128, block "
                       f"16): oracle/oracle.c paged_attention_v1, {n} iterations in {el:.1f} s")


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: spawn the N ranks (one per GPU) from this
    parent, which has not touched a GPU (device_count() does not initialise one), rendezvous on
    127.0.0.1, and return non-zero if any rank fails."""
    import socket
    import subprocess
    if "--dry-run-backend" not in " ".join(sys.argv):
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        if p.wait() != 0:
            rc = 1
    return rc


def run_custom_ar_children(rank, local_rank, world, timeout_s=150):
    """One child per rank (started by the rank itself); rank 0 relays its child's JSON."""
    import subprocess
    port = int(os.environ.get("MASTER_PORT", "29500")) + 1
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world),
               MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_USE_AGENT_STORE", "GROUP_RANK", "ROLE_RANK"):
        env.pop(k, None)
    try:
        p = subprocess.Popen([sys.executable, "-m", "vllm_amd.distributed.ar_bench"], env=env, cwd=ROOT,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        try:
            out, err = p.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            p.kill()
            out, err = p.communicate()
            return {"error": f"child timed out after {timeout_s} s"}
        if rank != 0:
            return None
        for ln in reversed(out.strip().splitlines()):
            if ln.startswith("{"):
                res = json.loads(ln)
                res["child_rc"] = p.returncode
                return res
        return {"error": f"child rc {p.returncode}: {err.strip()[-300:]}"}
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def dry_run(args):
    """CPU rehearsal of the N-rank launch path (tests): rendezvous, one all-reduce, rank 0 prints."""
    import torch.distributed as dist
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dist.init_process_group(args.dry_run_backend, rank=rank, world_size=world)
    t = torch.ones(1)
    dist.all_reduce(t)
    ok = int(t.item()) == world and dist.get_world_size() == args.gpus
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": dist.get_world_size(), "sum": t.item()}), flush=True)
    dist.destroy_process_group()
    return 0 if ok else 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs (default: WORLD_SIZE, else 1)")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-extra", action="store_true", help="skip scaled_mm / cache / cpu legs")
    ap.add_argument("--dry-run-backend", default=None,
                    help="(tests) 'gloo': ranks only rendezvous on the CPU and rank 0 prints a stub line")
    args = ap.parse_args()

    if args.gpus is None:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if args.dry_run_backend:
        sys.exit(dry_run(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # VLLM_AMD_BENCH_DIST=1 exercises the distributed path with a single rank (rehearsal of the
    # N > 1 launch on a one-GPU box)
    use_dist = world > 1 or os.environ.get("VLLM_AMD_BENCH_DIST") == "1"
    if use_dist:
        import torch.distributed as dist  # noqa
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        # RCCL prints a version banner on stdout at its first communicator: keep stdout for the ONE JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
        if dist.get_world_size() != world:
            print(f"bench.py: RCCL world size {dist.get_world_size()} != {world}", file=sys.stderr)
            sys.exit(2)
    device = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(device)

    # configs[1]; per-GPU shape is the same at every N (head-sharded, weak scaling)
    cfg = dict(S=64, ctx=4096, H=32, KVH=8, D=128, BS=16, kv="fp8")
    r = run_attention(cfg, args.steps, args.warmup, device, dist)
    wall = r["wall"]
    if use_dist:
        t = torch.tensor([wall, r["event"], r["cold_wall"], r["cold_event"]], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, ev, cold_wall, cold_ev = t.tolist()
    else:
        ev, cold_wall, cold_ev = r["event"], r["cold_wall"], r["cold_event"]
    total_bytes = r["bytes"] * world * args.steps
    value = total_bytes / wall / 1e9
    per_launch = ev / args.steps
    achieved = r["bytes"] / per_launch / 1e9

    # HBM bytes per launch: from the COMMITTED rocprofv3 PMC passes (FETCH_SIZE x 2 + WRITE_SIZE,
    # profiles/), not re-measured in this run
    traffic, traffic_src = None, None
    for name in ("r03_attention_pmc.json", "r02_attention_pmc.json", "r01_attention_pmc.json"):
        pmc = os.path.join(ROOT, "profiles", name)
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            traffic_src = "profiles/" + name + " (committed rocprofv3 --pmc passes, not this run)"
            break

    # the same kernel's average duration in the COMMITTED rocprofv3 --kernel-trace --stats run (the tracer costs
    # ~1.7 us per launch: profiles/r02_decode_drift.txt)
    trace_us, trace_cold_us, trace_src = None, None, None
    for name in ("r03_attention_pmc.json", "r02_attention_pmc.json"):
        pmc = os.path.join(ROOT, "profiles", name)
        if os.path.exists(pmc):
            pj = json.load(open(pmc))
            trace_us, trace_cold_us = pj.get("kernel_trace_avg_us"), pj.get("kernel_trace_cold_avg_us")
            if trace_us:
                trace_src = "profiles/" + name + " (committed rocprofv3 --kernel-trace run, not this run)"
                break
    cold_per = cold_ev / args.steps

    line = {
        "metric": "paged-attn decode HBM GB/s (% peak) + fp8 scaled_mm TFLOPS at 1/2/4/8 GPUs",
        "value": round(value, 1),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "fp8_e4m3 KV / bf16 q,out / f32 accumulate",
        "data": "synthetic",
        "config": {"workload": "configs[1]: Llama-3-8B-shape decode paged attention, fp8 e4m3 KV "
                               "(per-tensor scales), batch 64, ctx 4096, 32q/8kv heads, head 128, "
                               "block 16, per GPU",
                   "op": "PagedAttention.forward_decode -> paged_attention_v1 (pa_mfma_kernel)",
                   "parallelism": f"kv-head sharded x{world}, no collective"},
        "pct_of_hbm_peak": round(100.0 * value / world / HBM_PEAK_GBPS, 2),
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                     "traffic_source": traffic_src,
                     "kernel": "pa_mfma_kernel<bf16, fp8, 128, 16>",
                     "us_per_launch": round(per_launch * 1e6, 2),
                     "frac_kernel_trace": (round(r["bytes"] / (trace_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)
                                           if trace_us else None),
                     "frac_kernel_trace_cold": (round(r["bytes"] / (trace_cold_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)
                                                if trace_cold_us else None),
                     "kernel_trace_avg_us": trace_us, "kernel_trace_cold_avg_us": trace_cold_us,
                     "kernel_trace_source": trace_src,
                     "cold": {"what": f"the process's first {args.warmup} + {args.steps} launches (no "
                                      "preconditioning): W warmups, K timed",
                              "us_per_launch": round(cold_per * 1e6, 2),
                              "frac": round(r["bytes"] / cold_per / 1e9 / HBM_PEAK_GBPS, 4),
                              "value_gbps": round(total_bytes / cold_wall / 1e9, 1)},
                     "algorithmic_bytes_per_launch": r["bytes"],
                     "steady_state": f"{r['precondition_launches']} untimed launches (~{PRECONDITION_MS:.0f} ms) before "
                                     f"the {args.warmup} warmup steps: the first ~200 launches of a process run "
                                     "3-4 % slower (tools/decode_drift.py)"},
    }

    if rank == 0 and not args.no_extra:
        extra = {}
        if world == 1:
            shapes = [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)]
            extra["scaled_mm"] = run_scaled_mm(device, shapes, [1, 16, 64, 256, 1024, 4096])
            best = max(extra["scaled_mm"], key=lambda e: e["tflops"])
            extra["scaled_mm_best_tflops"] = best["tflops"]
            extra["scaled_mm_pct_of_fp8_peak"] = round(100 * best["tflops"] / FP8_DENSE_PEAK_TFLOPS, 2)
            # the matrix pipes' own random-data rate under the chip's power management, measured in this run
            ceil = run_mfma_ceiling(device)
            extra["mfma_random_data_ceiling_tflops"] = ceil
            extra["scaled_mm_best_frac_of_mfma_ceiling"] = round(best["tflops"] / ceil, 4)
            extra["scaled_mm_note"] = ("weights rotate through `weight_copies` buffers (> 256 MiB in all): every "
                                       "launch streams them from HBM; frac = TFLOP/s over min(5 PF, 8 TB/s x AI)")
            extra["cache_quant"] = run_cache_quant(device)
            extra["fused_neighbours"] = run_fused_neighbours(device)
            extra["prefill_attention"] = run_prefill(device)
            extra["unified_decode"] = run_unified_decode(device)
            extra["head_256"] = run_wide_heads(device)
            # the single-GPU shapes of configs[3] / configs[4] and a ragged configs[1] batch
            extra["configs3_per_gpu"] = run_attention_shape("configs[3] per-GPU", 256, 32768, 8, 1, device)
            extra["configs4_n1"] = run_attention_shape("configs[4] at N=1", 8, 131072, 64, 8, device)
            extra["ragged_decode"] = run_ragged_decode(device)
            line["cpu_baseline"] = cpu_baseline()
            # same workload as `value` (fp8 KV), which the reference's CPU kernel cannot run
            line["cpu_baseline_port"] = cpu_baseline_port()
        line["extra"] = extra
    if use_dist:
        from vllm_amd.distributed import bench_row_parallel
        ex = {}
        try:
            if 8 % world == 0:
                ex["configs4_strong"] = run_attention_shape(
                    f"configs[4] strong-scaled over {world} GPUs", 8, 131072, 64 // world, 8 // world,
                    device, dist, world)
            ex["configs3_per_gpu"] = run_attention_shape("configs[3] per-GPU", 256, 32768, 8, 1,
                                                         device, dist, world)
        except Exception as e:
            ex["attention_error"] = f"{type(e).__name__}: {e}"
        if rank == 0:
            line.setdefault("extra", {}).update(ex)
        try:
            rp = bench_row_parallel(device, dist)
        except Exception as e:   # the headline line must still be printed
            rp = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            line.setdefault("extra", {})["row_parallel_tp"] = rp
        if world > 1 and os.environ.get("VLLM_AMD_BENCH_CUSTOM_AR", "1") != "0":
            # the peer-to-peer all-reduce beside RCCL (result-checked first, 2 s barrier limit).  It has only run
            # with simulated ranks on one GPU so far, so every rank runs it in a CHILD process with process groups
            # of its own (vllm_amd/distributed/ar_bench.py): whatever happens there, this process prints its line.
            ca = run_custom_ar_children(rank, local_rank, world)
            if rank == 0:
                line.setdefault("extra", {})["custom_all_reduce"] = ca
        dist.barrier()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Tensor-parallel pieces of the hot path: KV-head sharding of
attention (no collective), column/row-parallel fp8 linear, and the single SUM
all-reduce at the row-parallel boundary (vllm/model_executor/layers/linear.py
:1385-1389 -> RCCL ncclAllReduce on the compute stream, pynccl.py:110-133).

One process per GPU; ``torch.distributed`` backend "nccl" is RCCL on ROCm, "gloo"
is used by the CPU tests of the sharding logic."""
from __future__ import annotations

import torch


def shard_range(total: int, world: int, rank: int) -> tuple[int, int]:
    """[start, end) of a dimension split evenly over ranks (linear.py divide())."""
    if total % world != 0:
        raise ValueError(f"{total} is not divisible by {world}")
    per = total // world
    return rank * per, (rank + 1) * per


def shard_kv_heads(num_heads: int, num_kv_heads: int, world: int, rank: int):
    """Heads owned by `rank` (QKVParallelLinear, linear.py:886): each rank gets
    num_kv_heads/world KV heads and their GQA group of q heads.  KV-head
    replication (world > num_kv_heads) is the reference's behaviour for that case
    and out of scope here."""
    if num_kv_heads % world != 0:
        raise ValueError("num_kv_heads must be divisible by the TP world size")
    kv0, kv1 = shard_range(num_kv_heads, world, rank)
    g = num_heads // num_kv_heads
    return (kv0 * g, kv1 * g), (kv0, kv1)


def tensor_model_parallel_all_reduce(x: torch.Tensor, dist=None, group=None, custom_ar=None) -> torch.Tensor:
    """communication_op.py:12-14: out-of-place SUM over the TP group.  `custom_ar`: a
    `CustomAllreduce` (custom_all_reduce.py) tried first, as CudaCommunicator.all_reduce does
    (cuda_communicator.py:93-118); RCCL otherwise."""
    if dist is None:
        import torch.distributed as dist  # type: ignore
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return x
    if custom_ar is not None:
        out = custom_ar.custom_all_reduce(x)
        if out is not None:
            return out
    out = x.clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    return out


def row_parallel_fp8_linear(x_shard: torch.Tensor, wq_shard: torch.Tensor,
                            w_scale: torch.Tensor, dist=None, group=None,
                            linear_op=None) -> torch.Tensor:
    """RowParallelLinear.forward with a PTPC fp8 weight (linear.py:1369-1397):
    the activation shard [M, K/world] is quantised per token ON THIS RANK's shard
    (reference behaviour: quant runs inside apply() on input_parallel), multiplied
    with the [N, K/world] fp8 weight shard, then summed over ranks."""
    from ..quantization import Fp8LinearOp
    op = linear_op or Fp8LinearOp()
    partial = op.apply(x_shard, wq_shard.t(), w_scale, out_dtype=x_shard.dtype)
    return tensor_model_parallel_all_reduce(partial, dist, group)


def bench_row_parallel(device, dist, M: int = 256, hidden: int = 8192, iters: int = 20):
    """configs[3] GEMM + all-reduce legs: row-parallel fp8 GEMMs (K per GPU 1024 and
    3584 at TP8) and the RCCL all-reduce of the [256, 8192] bf16 result."""
    import time
    from .. import _custom_ops as ops
    world = dist.get_world_size()
    res = {}
    for name, K_full in (("o_proj", 8192), ("down_proj", 28672)):
        Ks = K_full // 8  # per-GPU K of the TP8 Llama-3-70B shard (weight_shapes.py:58-63 / 8)
        w = torch.randn((hidden, Ks), device=device, dtype=torch.bfloat16)
        wq, ws = ops.scaled_fp8_quant(w, use_per_token_if_dynamic=True)
        x = torch.randn((M, Ks), device=device, dtype=torch.bfloat16)

        def gemm_only():
            xq, xs = ops.scaled_fp8_quant(x, use_per_token_if_dynamic=True)
            return ops.cutlass_scaled_mm(xq, wq.t(), xs, ws, torch.bfloat16)

        def full():
            y = gemm_only()
            dist.all_reduce(y)
            return y

        for tag, fn in (("quant_gemm_us", gemm_only), ("quant_gemm_allreduce_us", full)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize(); dist.barrier()
            t = torch.tensor([(time.perf_counter() - t0) / iters], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            res[f"{name}_K{Ks}_{tag}"] = round(t.item() * 1e6, 2)
        # the two legs above include the host's launch cost (Python + dispatcher, ~15 us per op);
        # the GPU time alone, from a HIP graph of 20 quant + GEMM pairs
        if world > 1:   # capture next to a live RCCL watchdog is not worth the risk: one rank only
            continue
        try:
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                gemm_only()
            torch.cuda.current_stream(device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                for _ in range(iters):
                    gemm_only()
            graph.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                graph.replay()
            e1.record()
            torch.cuda.synchronize()
            res[f"{name}_K{Ks}_quant_gemm_graph_us"] = round(e0.elapsed_time(e1) * 1e3 / (5 * iters), 2)
            del graph
        except Exception as e:   # capture is an extra, never fatal
            res[f"{name}_K{Ks}_quant_gemm_graph_us"] = f"n/a ({type(e).__name__})"
    y = torch.randn((M, hidden), device=device, dtype=torch.bfloat16)
    for _ in range(3):
        dist.all_reduce(y)
    torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        dist.all_reduce(y)
    torch.cuda.synchronize()
    t = torch.tensor([(time.perf_counter() - t0) / iters], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    nbytes = y.numel() * 2
    res["allreduce_4MiB_us"] = round(t.item() * 1e6, 2)
    res["allreduce_bus_gbps"] = round(2 * (world - 1) / world * nbytes / t.item() / 1e9, 1)
    return res


def bench_custom_all_reduce(device, dist, M: int = 256, hidden: int = 8192, iters: int = 20):
    """The peer-to-peer all-reduce on the configs[3] message ([256, 8192] bf16 = 4 MiB) and on a
    decode-sized one ([16, 8192] = 256 KiB) next to RCCL, across the ranks of `dist` (bench.py --gpus N;
    VLLM_AMD_BENCH_CUSTOM_AR=0 skips it).  This path has never run across GPUs, so every step is guarded:
    a 2 s barrier time limit, the result is checked against RCCL's BEFORE anything is timed, and the leg
    stops at the first time-out or mismatch (all ranks agree on that through an RCCL reduction).  The IPC
    handles travel through a gloo group created here."""
    import os
    import sys
    import time
    from .custom_all_reduce import CustomAllreduce
    world = dist.get_world_size()
    verbose = os.environ.get("VLLM_AMD_AR_BENCH_VERBOSE") == "1"

    def say(msg):
        if verbose:
            print(f"[ar_bench rank {dist.get_rank()}] {msg}", file=sys.stderr, flush=True)

    say("new gloo group")
    cpu_group = dist.new_group(backend="gloo")
    say("CustomAllreduce")
    ar = CustomAllreduce(cpu_group, device)
    res = {"disabled": ar.disabled}
    if ar.disabled:
        return res

    def any_rank(flag: bool) -> bool:
        t = torch.tensor([1.0 if flag else 0.0], device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t.item() > 0)

    try:
        ar.set_timeout_ms(2000)
        for rows in (16, M):
            x = torch.randn((rows, hidden), device=device, dtype=torch.bfloat16)
            want = x.clone()
            say(f"rows {rows}: reference all_reduce")
            dist.all_reduce(want)
            torch.cuda.synchronize()
            say("custom all_reduce")
            failed, got = False, None
            try:
                got = ar.custom_all_reduce(x)
                torch.cuda.synchronize()
            except RuntimeError as e:
                failed, res["error"] = True, str(e)[:200]
            if any_rank(failed or ar.timed_out()):
                res["timed_out"] = True     # a barrier gave up somewhere: stop here, nothing is timed
                return res
            diff = float((got.float() - want.float()).abs().max())
            res[f"rows{rows}_max_abs_diff_vs_rccl"] = diff
            # both sum `world` bf16 values in fp32 and round once; the orders differ: a few bf16 ulp of the sum
            if any_rank(not (diff <= 0.05 * world)):
                res["mismatch"] = True
                return res
            nbytes = x.numel() * 2
            for tag, fn in (("custom", lambda: ar.custom_all_reduce(x)), ("rccl", lambda: dist.all_reduce(x))):
                say(f"timing {tag}")
                for _ in range(3):
                    fn()
                torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(iters):
                    fn()
                torch.cuda.synchronize()
                t = torch.tensor([(time.perf_counter() - t0) / iters], device=device, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                res[f"rows{rows}_{tag}_us"] = round(t.item() * 1e6, 2)
                res[f"rows{rows}_{tag}_bus_gbps"] = round(2 * (world - 1) / world * nbytes / t.item() / 1e9, 1)
                if any_rank(ar.timed_out()):
                    res["timed_out"] = True
                    return res
        res["timed_out"] = False
    finally:
        ar.close()
    return res

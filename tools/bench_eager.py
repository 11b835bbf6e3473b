"""Host launch cost of the op layer: eager quant + GEMM (ops.scaled_fp8_quant per token ->
ops.cutlass_scaled_mm) against the same pair replayed from a HIP graph, at the configs[3]
row-parallel shapes (M 256, K 1024 / 3584 -> N 8192) and a decode shape (M 16, 4096 x 4096).
Also single-op dispatch cost (static_scaled_fp8_quant on 1 x 128, a ~2 us kernel)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vllm_amd import _custom_ops as ops

dev = torch.device("cuda", 0)


def timed(fn, iters=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


def graphed(fn, iters=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (10 * iters)


for M, K, N in ((256, 1024, 8192), (256, 3584, 8192), (16, 4096, 4096)):
    w = torch.randn((N, K), device=dev, dtype=torch.bfloat16)
    wq, ws = ops.scaled_fp8_quant(w, use_per_token_if_dynamic=True)
    x = torch.randn((M, K), device=dev, dtype=torch.bfloat16)

    def pair():
        xq, xs = ops.scaled_fp8_quant(x, use_per_token_if_dynamic=True)
        return ops.cutlass_scaled_mm(xq, wq.t(), xs, ws, torch.bfloat16)

    e, g = timed(pair), graphed(pair)
    print(f"quant+gemm M{M} K{K} N{N}: eager {e:.1f} us  graph {g:.1f} us  ratio {e / g:.2f}", flush=True)

xs1 = torch.randn((1, 128), device=dev, dtype=torch.bfloat16)
o1 = torch.empty((1, 128), device=dev, dtype=torch.float8_e4m3fn)
sc = torch.ones(1, device=dev)
f = lambda: torch.ops._C.static_scaled_fp8_quant(o1, xs1, sc)
print(f"single op (static quant 1x128): eager {timed(f, 2000):.2f} us  graph {graphed(f):.2f} us", flush=True)

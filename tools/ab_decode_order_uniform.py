import sys, os, torch
sys.path.insert(0, "/root/repo")
import bench
from vllm_amd import _lib
from vllm_amd.attention.paged_attn import PagedAttention
dev = torch.device("cuda", 0)
S, ctx, H, KVH, D, BS = 64, 4096, 32, 8, 128, 16
q, caches, bt, sl, one = bench.make_attention_inputs(S, ctx, H, KVH, D, BS, True, dev, n_caches=2)
st = {"i": 0}
def fn():
    kc, vc = caches[st["i"] % 2]; st["i"] += 1
    return PagedAttention.forward_decode(q, kc, vc, bt, sl, ctx, "fp8", KVH, D ** -0.5, None, one, one)
tot = {0: 0.0, 1: 0.0}
for rep in range(12):
    for order in (0, 1):
        _lib.set_tuning("pa_order", order)
        _, ev = bench.time_gpu(fn, 100, 5)
        tot[order] += ev / 100 * 1e6
print("uniform 64x4096: order=0 %.2f us  order=1 %.2f us (12 alternations of 100 launches)" % (tot[0] / 12, tot[1] / 12))

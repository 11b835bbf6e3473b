import sys, torch
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
bench.precondition(fn, 150.0)
tot = {0: [], 1: []}
for rep in range(8):
    for v in (0, 1):
        _lib.set_tuning("pa_variant", v)
        _, ev = bench.time_gpu(fn, 200, 10)
        tot[v].append(ev / 200 * 1e6)
_lib.set_tuning("pa_variant", 1)
for v in (0, 1):
    print(f"steady state, pa_variant={v} ({'plain' if v == 0 else 'non-temporal'} K/V loads):", " ".join(f"{x:.2f}" for x in tot[v]), f" mean {sum(tot[v]) / len(tot[v]):.2f} us")

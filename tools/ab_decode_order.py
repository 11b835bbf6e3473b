import sys, os, torch
sys.path.insert(0, "/root/repo")
import bench
from vllm_amd import _lib
from vllm_amd.attention.paged_attn import PagedAttention
dev = torch.device("cuda", 0)
S, ctx, H, KVH, D, BS = 64, 4096, 32, 8, 128, 16
q, caches, bt, sl, one = bench.make_attention_inputs(S, ctx, H, KVH, D, BS, True, dev, n_caches=2)
g = torch.Generator().manual_seed(0)
lens = torch.randint(512, ctx + 1, (S,), generator=g).to(torch.int32)
for name, sl_ in (("uniform", sl), ("ragged", lens.to(dev))):
    mx = int(sl_.max())
    st = {"i": 0}
    def fn():
        kc, vc = caches[st["i"] % 2]; st["i"] += 1
        return PagedAttention.forward_decode(q, kc, vc, bt, sl_, mx, "fp8", KVH, D ** -0.5, None, one, one)
    for rep in range(3):
        for order in (0, 1):
            _lib.set_tuning("pa_order", order)
            _, ev = bench.time_gpu(fn, 50, 5)
            print(f"{name} order={order}: {ev / 50 * 1e6:.2f} us", flush=True)
_lib.set_tuning("pa_order", 1)

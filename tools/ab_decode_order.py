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

# the flash-layout decode kernel of unified_attention (V1 engine), same batch
from vllm_amd.attention.ops import unified_attention
NBK = S * (ctx // BS)
kf = torch.randint(0, 120, (NBK, BS, KVH, D), device=dev, dtype=torch.uint8)
vf = torch.randint(0, 120, (NBK, BS, KVH, D), device=dev, dtype=torch.uint8)
btu = torch.randperm(NBK, device=dev).to(torch.int32).view(S, ctx // BS)
cu = torch.arange(S + 1, device=dev, dtype=torch.int32)
qb = q.to(torch.bfloat16) if q.dtype != torch.bfloat16 else q
o = torch.empty_like(qb)
for name, sl_ in (("unified uniform", sl), ("unified ragged", lens.to(dev))):
    mx = int(sl_.max())
    fn = lambda: unified_attention(qb, kf, vf, o, cu, 1, sl_, mx, D ** -0.5, True, (-1, -1), btu, 0.0, None, one, one)
    for rep in range(2):
        for order in (0, 1):
            _lib.set_tuning("pa_order", order)
            _, ev = bench.time_gpu(fn, 50, 5)
            print(f"{name} order={order}: {ev / 50 * 1e6:.2f} us", flush=True)
_lib.set_tuning("pa_order", 1)

# 256 sequences (2048 workgroups, XCD-grouped launch, several rounds)
del kf, vf, caches
S2 = 256
q2, caches2, bt2, sl2, one2 = bench.make_attention_inputs(S2, ctx, H, KVH, D, BS, True, dev, n_caches=1)
lens2 = torch.randint(512, ctx + 1, (S2,), generator=g).to(torch.int32)
for name, sl_ in (("S256 uniform", sl2), ("S256 ragged", lens2.to(dev))):
    mx = int(sl_.max())
    kc, vc = caches2[0]
    fn = lambda: PagedAttention.forward_decode(q2, kc, vc, bt2, sl_, mx, "fp8", KVH, D ** -0.5, None, one2, one2)
    nbytes = int(sl_.sum()) * KVH * D * 2
    for rep in range(2):
        for order in (0, 1):
            _lib.set_tuning("pa_order", order)
            _, ev = bench.time_gpu(fn, 20, 3)
            print(f"{name} order={order}: {ev / 20 * 1e6:.2f} us  {nbytes / (ev / 20) / 1e12:.2f} TB/s", flush=True)
_lib.set_tuning("pa_order", 1)

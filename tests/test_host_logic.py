"""CPU tests (no GPU) of the host-side mirrors: PagedAttention helper (shapes, v1/v2
heuristic, scratch shapes), Fp8LinearOp plumbing, TP sharding, and a world_size-2 gloo
run of the N>1 path (KV-head-sharded attention + row-parallel GEMM + all-reduce) with the
oracle standing in for the kernels."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from helpers import make_attn_case


def test_kv_cache_shape_and_split():
    from vllm_amd.attention import PagedAttention
    assert PagedAttention.get_kv_cache_shape(10, 16, 8, 128) == (2, 10, 16 * 8 * 128)
    kv = torch.zeros(PagedAttention.get_kv_cache_shape(10, 16, 8, 128), dtype=torch.bfloat16)
    k, v = PagedAttention.split_kv_cache(kv, 8, 128)
    assert k.shape == (10, 8, 16, 16, 8) and v.shape == (10, 8, 128, 16)
    kv8 = torch.zeros(PagedAttention.get_kv_cache_shape(10, 16, 8, 128), dtype=torch.uint8)
    k, v = PagedAttention.split_kv_cache(kv8, 8, 128)
    assert k.shape == (10, 8, 8, 16, 16) and v.shape == (10, 8, 128, 16)
    assert 120 in PagedAttention.get_supported_head_sizes()


@pytest.mark.parametrize("S,H,max_len,want,KVH", [
    (64, 32, 4096, "v1", 8),      # configs[1]: S*H = 2048 > 512
    (8, 64, 131072, "v2", 8),     # configs[4]: > 8192
    (4, 32, 300, "v1", 8),        # one partition
    (4, 32, 2048, "v2", 8),       # S*H = 128 <= 512 and 4 partitions
    (256, 8, 32768, "v2", 8),     # configs[3]
    (128, 8, 4096, "v2", 1),      # S*H = 1024 > 512, but 128 (sequence, kv head) workgroups: partitioned here
    (256, 8, 2048, "v1", 1),      # 256 workgroups: one per CU
])
def test_forward_decode_heuristic(monkeypatch, S, H, max_len, want, KVH):
    """vllm/attention/ops/paged_attn.py:117-193: same choice (refined where this implementation's workgroup count
    differs from the reference's, see forward_decode) and scratch shapes."""
    from vllm_amd.attention import paged_attn
    calls = {}
    monkeypatch.setattr(paged_attn.ops, "paged_attention_v1",
                        lambda *a, **k: calls.setdefault("v1", a))
    monkeypatch.setattr(paged_attn.ops, "paged_attention_v2",
                        lambda *a, **k: calls.setdefault("v2", a))
    q = torch.zeros((S, H, 128), dtype=torch.bfloat16)
    kc = torch.zeros((4, 8, 16, 16, 8), dtype=torch.bfloat16)
    vc = torch.zeros((4, 8, 128, 16), dtype=torch.bfloat16)
    bt = torch.zeros((S, (max_len + 15) // 16), dtype=torch.int32)
    sl = torch.ones(S, dtype=torch.int32)
    one = torch.ones(1)
    out = paged_attn.PagedAttention.forward_decode(q, kc, vc, bt, sl, max_len, "auto", KVH, 0.1,
                                                   None, one, one)
    assert out.shape == q.shape and list(calls) == [want]
    if want == "v2":
        P = (max_len + 511) // 512
        _, es, ml, tmp = calls["v2"][:4]
        assert es.shape == (S, H, P) and ml.shape == (S, H, P) and tmp.shape == (S, H, P, 128)
        assert es.dtype == torch.float32 and tmp.dtype == q.dtype


def test_shard_helpers():
    from vllm_amd.distributed import shard_kv_heads, shard_range
    assert shard_range(8192, 8, 3) == (3072, 4096)
    assert shard_kv_heads(64, 8, 8, 5) == ((40, 48), (5, 6))
    assert shard_kv_heads(32, 8, 2, 1) == ((16, 32), (4, 8))
    with pytest.raises(ValueError):
        shard_kv_heads(32, 8, 3, 0)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _tp_worker(rank, world, port, q):
    import torch.distributed as dist
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    from oracle import oracle as O
    from vllm_amd.distributed import shard_kv_heads, shard_range, tensor_model_parallel_all_reduce
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # ---- attention: every rank owns KVH/world kv heads, no collective inside ----
        H, KVH, D, BS = 8, 2, 64, 16
        c = make_attn_case(3, H, KVH, D, BS, torch.bfloat16, "fp8", [5, 200, 530], seed=1, kv_scale=0.5)
        (h0, h1), (k0, k1) = shard_kv_heads(H, KVH, world, rank)
        qs = c["q"][:, h0:h1].contiguous()
        kcs, vcs = c["kc"][:, k0:k1].contiguous(), c["vc"][:, k0:k1].contiguous()
        out_s = torch.empty_like(qs)
        O.paged_attention_v1(out_s, qs, kcs, vcs, k1 - k0, c["scale"], c["bt"], c["sl"], BS,
                             c["max_len"], None, "fp8", 0.5, 0.5)
        gathered = [torch.empty_like(out_s) for _ in range(world)]
        dist.all_gather(gathered, out_s)
        full = torch.empty_like(c["q"])
        O.paged_attention_v1(full, c["q"], c["kc"], c["vc"], KVH, c["scale"], c["bt"], c["sl"], BS,
                             c["max_len"], None, "fp8", 0.5, 0.5)
        ok_attn = torch.equal(torch.cat(gathered, dim=1), full)   # sharding is exact
        # ---- row-parallel fp8 linear: quantise the K-shard per token on this rank, GEMM,
        #      then ONE sum all-reduce (linear.py:1385-1389) ----
        g = torch.Generator().manual_seed(7)
        M, K, N = 5, 256, 48
        x = torch.randn((M, K), generator=g).to(torch.bfloat16)
        w = torch.randn((N, K), generator=g).to(torch.bfloat16)
        ka, kb = shard_range(K, world, rank)
        xq, xs = O.scaled_fp8_quant(x[:, ka:kb].contiguous(), use_per_token_if_dynamic=True)
        wq, ws = O.scaled_fp8_quant(w[:, ka:kb].contiguous(), use_per_token_if_dynamic=True)
        part = O.scaled_mm(xq, wq.t(), xs, ws, torch.bfloat16).float()
        total = tensor_model_parallel_all_reduce(part, dist)
        ref = x.float() @ w.float().t()
        err = (total - ref).abs().max().item() / ref.abs().max().item()
        if rank == 0:
            q.put((ok_attn, err, part.data_ptr() != total.data_ptr()))
    finally:
        dist.destroy_process_group()


def test_tp2_gloo_sharded_attention_and_row_parallel():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok_attn, err, out_of_place = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok_attn, "KV-head-sharded attention must equal the unsharded result exactly"
    assert err < 0.06, err            # two fp8 quantisations (~2^-4 relative each, averaged over K)
    assert out_of_place


def test_v1_backend_host_logic():
    """TritonAttentionBackend / MetadataBuilder / Impl constructor: the reference's contract
    (vllm/v1/attention/backends/triton_attn.py:58-265) without a GPU."""
    import types
    from vllm_amd.v1.attention.backends import (TritonAttentionBackend, TritonAttentionImpl,
                                                TritonAttentionMetadata,
                                                TritonAttentionMetadataBuilder)
    B = TritonAttentionBackend
    assert B.get_name() == "TRITON_ATTN_VLLM_V1" and B.accept_output_buffer
    assert B.get_impl_cls() is TritonAttentionImpl and B.get_metadata_cls() is TritonAttentionMetadata
    assert B.get_builder_cls() is TritonAttentionMetadataBuilder
    assert B.get_kv_cache_shape(10, 16, 8, 128) == (2, 10, 16, 8, 128)
    with pytest.raises(ValueError, match="multiple of 16"):
        B.get_kv_cache_shape(10, 8, 8, 128)
    assert B.use_cascade_attention() is False
    assert B.get_kv_cache_stride_order() == (0, 1, 3, 2, 4)   # head-major memory by default
    assert torch.bfloat16 in B.get_supported_dtypes() and 128 in B.get_supported_head_sizes()
    assert B.get_supported_head_sizes() == [32, 64, 96, 128, 160, 192, 224, 256]   # triton_attn.py:147-148
    with pytest.raises(ValueError, match="Head size 80 is not supported by TritonAttention"):
        B.validate_head_size(80)

    common = types.SimpleNamespace(
        num_actual_tokens=7, max_query_len=5, query_start_loc=torch.tensor([0, 5, 6, 7], dtype=torch.int32),
        seq_lens=torch.tensor([9, 30, 2], dtype=torch.int32), seq_lens_cpu=torch.tensor([9, 30, 2], dtype=torch.int32),
        block_table_tensor=torch.zeros((3, 2), dtype=torch.int32), slot_mapping=torch.arange(7))
    cfg = types.SimpleNamespace(
        model_config=types.SimpleNamespace(get_num_attention_heads=lambda pc: 32 // pc.tp,
                                           get_num_kv_heads=lambda pc: 8 // pc.tp, get_head_size=lambda: 128),
        parallel_config=types.SimpleNamespace(tp=2))
    b = TritonAttentionMetadataBuilder(types.SimpleNamespace(block_size=16), cfg, torch.device("cpu"))
    assert b.block_size == 16
    md = b.build(0, common)
    assert (md.num_actual_tokens, md.max_query_len, md.max_seq_len) == (7, 5, 30)
    assert md.use_cascade is False and md.cu_prefix_query_lens is None and md.suffix_kv_lens is None
    assert md.block_table is common.block_table_tensor and md.slot_mapping is common.slot_mapping
    md = b.build(4, common)   # a common prefix is flagged (the impl rejects cascade); no cascade tensors
    assert md.use_cascade and md.common_prefix_len == 4 and md.prefix_kv_lens is None
    md = b.build_for_cudagraph_capture(common)
    assert md.seq_lens.tolist() == [1, 1, 1] and b.can_run_in_cudagraph(common)

    impl = TritonAttentionImpl(32, 128, 0.25, 8, [0.5] * 32, 1024, "fp8", 30.0)
    assert impl.sliding_window == (1023, 0) and impl.logits_soft_cap == 30.0 and impl.num_queries_per_kv == 4
    assert impl.alibi_slopes.dtype == torch.float32 and impl.fp8_dtype == torch.float8_e4m3fn
    impl = TritonAttentionImpl(32, 64, 0.25, 8, None, None, "auto")
    assert impl.sliding_window == (-1, -1) and impl.logits_soft_cap == 0 and impl.alibi_slopes is None
    assert impl.fused_output_quant_supported(torch.float8_e4m3fn, True, (-1, -1))


def test_bench_self_launch_two_ranks_gloo():
    """`python bench.py --gpus 2` with no launcher spawns its own ranks (127.0.0.1 rendezvous) and
    fails loudly when the world does not match: rehearsed on the CPU with gloo."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2",
                        "--dry-run-backend", "gloo"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["sum"] == 2.0
    env2 = dict(env, WORLD_SIZE="2", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4",
                        "--dry-run-backend", "gloo"], env=env2, capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr


def test_plugin_entry_point_metadata(tmp_path):
    """setup.py declares the `vllm.general_plugins` entry point the reference's plugin
    loader scans (vllm/plugins/__init__.py:19-81; packaging pattern of
    tests/plugins/vllm_add_dummy_platform/setup.py): build the metadata, discover the entry
    point through importlib.metadata as vLLM does, load and call it."""
    import importlib.metadata as md
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "setup.py", "-q", "egg_info", "--egg-base", str(tmp_path)],
                       cwd=root, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    dists = list(md.distributions(path=[str(tmp_path)]))
    assert len(dists) == 1 and dists[0].metadata["Name"] in ("vllm_amd", "vllm-amd")
    eps = [e for e in dists[0].entry_points if e.group == "vllm.general_plugins"]
    assert [e.name for e in eps] == ["vllm_amd"] and eps[0].value == "vllm_amd.plugin:register"
    register = eps[0].load()
    register()     # imports the op layer: the schemas are registered under the reference's names
    import torch
    for ns, op in (("_C", "paged_attention_v1"), ("_C", "cutlass_scaled_mm"),
                   ("_C_cache_ops", "reshape_and_cache"), ("_rocm_C", "paged_attention")):
        assert hasattr(getattr(torch.ops, ns), op)


def test_block_table_host_inputs_match_reference_arithmetic():
    """BlockTable.compute_slot_mapping with the numpy inputs the reference's model runner passes
    (block_table.py:84-99): host path, slot_mapping_np / slot_mapping_cpu filled, no device op."""
    import numpy as np
    from vllm_amd.worker import BlockTable
    bt = BlockTable(16, 4, 8, 64, False, torch.device("cpu"))
    bt.add_row([5, 9, 2], 0)
    bt.add_row([7, 1], 2)
    req = np.array([0, 0, 0, 2, 2], dtype=np.int32)
    pos = np.array([0, 17, 40, 3, 31], dtype=np.int64)
    bt.compute_slot_mapping(req, pos)
    want = np.array([5 * 16, 9 * 16 + 1, 2 * 16 + 8, 7 * 16 + 3, 1 * 16 + 15])
    assert (bt.slot_mapping_np[:5] == want).all() and (bt.slot_mapping_cpu[:5].numpy() == want).all()
    bt.commit_slot_mapping(5)
    assert (bt.slot_mapping[:5].numpy() == want).all()


def test_fp8_linear_static_zero_dim_scale_is_not_sliced(monkeypatch):
    """Static activation quantisation hands a 0-dim input_scale through (w8a8_utils.py:370-377);
    slicing it would raise IndexError."""
    from vllm_amd import quantization as Q
    seen = {}

    def fake_mm(a, b, sa, sb, out_dtype, bias=None):
        seen["sa"] = sa
        return torch.zeros((a.shape[0], b.shape[1]), dtype=out_dtype)

    monkeypatch.setattr(Q.ops, "cutlass_scaled_mm", fake_mm)
    op = Q.Fp8LinearOp(act_quant_static=True)
    monkeypatch.setattr(op, "quant_fp8", lambda x, s, ub: (x.to(torch.float8_e4m3fn), s))
    scale = torch.tensor(0.5)          # 0-dim, as Parameter(input_scale.max())
    out = op.apply(torch.zeros((3, 32), dtype=torch.bfloat16), torch.zeros((32, 16), dtype=torch.float8_e4m3fn),
                   torch.ones(16, 1), input_scale=scale)
    assert out.shape == (3, 16) and seen["sa"] is scale

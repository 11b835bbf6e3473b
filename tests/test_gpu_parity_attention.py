"""GPU parity: HIP paged attention (through torch.ops -> C ABI) vs the CPU oracle.

Tolerances are the reference's own (tests/kernels/attention/test_attention.py:345-348):
atol 1e-3 / rtol 1e-5 for 16/32-bit KV ("auto"), atol 1e-2 for fp8 KV -- we use
a tighter 2e-3 for fp8 because both sides use the exact fp8 values.
"""
import pytest
import torch

from helpers import make_attn_case, to_dev

pytestmark = pytest.mark.gpu

TOL = {"auto": dict(atol=1e-3, rtol=1e-5), "fp8": dict(atol=2e-3, rtol=1e-5)}


def run_case(ops, oracle, c, version, partition=None, query_start_loc=None, fp8_out_scale=None):
    q, kc, vc, bt, sl = c["q"], c["kc"], c["vc"], c["bt"], c["sl"]
    S, H, D = q.shape
    ks = torch.tensor([c["kv_scale"]], dtype=torch.float32)
    vs = torch.tensor([c["kv_scale"]], dtype=torch.float32)
    dq, dkc, dvc, dbt, dsl = (to_dev(t) for t in (q, kc, vc, bt, sl))
    dsl_, dks, dvs = to_dev(c["slopes"]), to_dev(ks), to_dev(vs)
    out_o = torch.zeros_like(q)
    if version == "v1":
        out = torch.full_like(dq, float("nan"))
        ops.paged_attention_v1(out, dq, dkc, dvc, c["num_kv_heads"], c["scale"], dbt, dsl,
                               c["block_size"], c["max_len"], dsl_, c["kv_cache_dtype"], dks, dvs)
        oracle.paged_attention_v1(out_o, q, kc, vc, c["num_kv_heads"], c["scale"], bt, sl,
                                  c["block_size"], c["max_len"], c["slopes"], c["kv_cache_dtype"],
                                  c["kv_scale"], c["kv_scale"])
    else:
        PS = 512 if version == "v2" else 256
        P = (c["max_len"] + PS - 1) // PS
        es = torch.zeros((S, H, P), dtype=torch.float32)
        ml = torch.zeros_like(es)
        tmp = torch.zeros((S, H, P, D), dtype=q.dtype)
        des, dml, dtmp = to_dev(es), to_dev(ml), to_dev(tmp)
        out = torch.full_like(dq, float("nan"))
        if version == "v2":
            ops.paged_attention_v2(out, des, dml, dtmp, dq, dkc, dvc, c["num_kv_heads"], c["scale"],
                                   dbt, dsl, c["block_size"], c["max_len"], dsl_,
                                   c["kv_cache_dtype"], dks, dvs)
            oracle.paged_attention_v2(out_o, es, ml, tmp, q, kc, vc, c["num_kv_heads"], c["scale"],
                                      bt, sl, c["block_size"], c["max_len"], c["slopes"],
                                      c["kv_cache_dtype"], c["kv_scale"], c["kv_scale"])
        else:
            ops.paged_attention_rocm(out, des, dml, dtmp, dq, dkc, dvc, c["num_kv_heads"],
                                     c["scale"], dbt, dsl, None, c["block_size"], c["max_len"],
                                     dsl_, c["kv_cache_dtype"], dks, dvs)
            oracle.paged_attention_rocm(out_o, es, ml, tmp, q, kc, vc, c["num_kv_heads"],
                                        c["scale"], bt, sl, None, c["block_size"], c["max_len"],
                                        c["slopes"], c["kv_cache_dtype"], c["kv_scale"],
                                        c["kv_scale"])
    torch.cuda.synchronize()
    got = out.cpu().float()
    kind = "auto" if c["kv_cache_dtype"] == "auto" else "fp8"
    torch.testing.assert_close(got, out_o.float(), **TOL[kind])
    # and both must sit near the unrounded fp64 answer
    f64 = oracle.paged_attention_f64(q, kc, vc, c["num_kv_heads"], c["scale"], bt, sl,
                                     c["block_size"], c["slopes"], c["kv_cache_dtype"],
                                     c["kv_scale"], c["kv_scale"])
    assert (got.double() - f64).abs().max().item() < TOL[kind]["atol"]


SEQ_LENS = [1, 15, 16, 17, 63, 64, 65, 300, 513, 1029]


@pytest.mark.parametrize("version", ["v1", "v2", "rocm"])
@pytest.mark.parametrize("kv", ["auto", "fp8"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("heads", [(32, 8), (8, 1), (5, 5), (16, 1)])
def test_mfma_path(ops, oracle, version, kv, dtype, heads, v1_form):
    if version != "v1" and v1_form == "plain":
        pytest.skip("the launch form only concerns v1")
    H, KVH = heads
    c = make_attn_case(len(SEQ_LENS), H, KVH, 128, 16, dtype, kv, SEQ_LENS, seed=H,
                       kv_scale=1.0 if kv == "auto" else 0.5)
    run_case(ops, oracle, c, version)


@pytest.mark.parametrize("version", ["v1", "v2", "rocm"])
@pytest.mark.parametrize("kv", ["auto", "fp8"])
@pytest.mark.parametrize("hs_bs", [(64, 16), (128, 32), (64, 32)])
def test_mfma_shapes(ops, oracle, version, kv, hs_bs):
    hs, bs = hs_bs
    c = make_attn_case(6, 8, 2, hs, bs, torch.bfloat16, kv, [1, 31, 32, 33, 600, 1100], seed=7,
                       kv_scale=1.0 if kv == "auto" else 0.25, alibi=(hs == 64))
    run_case(ops, oracle, c, version)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("kv", ["auto", "fp8"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hs_bs", [(80, 16), (256, 8), (120, 32), (32, 16), (192, 16), (96, 8), (112, 16)])
def test_generic_path(ops, oracle, version, kv, dtype, hs_bs):
    hs, bs = hs_bs
    if kv != "auto" and hs % 16:
        pytest.skip("fp8 cache needs head_size % 16 == 0 (x = 16), as in the reference tests")
    c = make_attn_case(5, 8, 4, hs, bs, dtype, kv, [1, 7, 130, 515, 700], seed=3,
                       kv_scale=1.0 if kv == "auto" else 0.5, alibi=(bs == 16))
    run_case(ops, oracle, c, version)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("kv", ["auto", "fp8"])
@pytest.mark.parametrize("hs_bs", [(32, 32), (80, 32), (96, 16), (112, 32), (120, 16)])
def test_mfma_padded_head_sizes(ops, oracle, version, kv, hs_bs, v1_form):
    if version != "v1" and v1_form == "plain":
        pytest.skip("the launch form only concerns v1")
    """head sizes below 64 / 128 run the MFMA kernel padded (absent chunks never loaded);
    GQA 4:1 and contexts past one partition (reference HEAD_SIZES, test_attention.py:33)."""
    hs, bs = hs_bs
    if kv != "auto" and hs % 16:
        pytest.skip("fp8 cache needs head_size % 16 == 0 (x = 16), as in the reference tests")
    c = make_attn_case(6, 16, 4, hs, bs, torch.bfloat16, kv, [1, 17, 64, 513, 1100, 2050], seed=hs,
                       kv_scale=1.0 if kv == "auto" else 0.5, alibi=(hs == 96))
    run_case(ops, oracle, c, version)


@pytest.mark.parametrize("version", ["v1", "v2", "rocm"])
@pytest.mark.parametrize("kv", ["auto", "fp8"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hs_bs", [(256, 16), (192, 16), (256, 32), (192, 32)])
@pytest.mark.parametrize("heads", [(16, 4), (20, 1)])
def test_mfma_wide_head_sizes(ops, oracle, version, kv, dtype, hs_bs, heads, v1_form):
    """The reference's head sizes 192 and 256 (csrc/attention/paged_attention_v1.cu:158-187; others above 128 are
    rejected as there) on the MFMA kernel's 256-wide instantiation (one workgroup per CU, 16 q heads written in two
    passes; 192 padded):
    GQA 4:1 and a 20-head MQA group (two chunks), contexts past one partition, ALiBi."""
    if version != "v1" and v1_form == "plain":
        pytest.skip("the launch form only concerns v1")
    hs, bs = hs_bs
    if kv != "auto" and hs % 16:
        pytest.skip("fp8 cache needs head_size % 16 == 0 (x = 16), as in the reference tests")
    if version == "rocm" and heads == (20, 1):
        pytest.skip("the rocm op's GQA ratio stops at 16 (csrc/rocm/attention.cu)")
    c = make_attn_case(6, heads[0], heads[1], hs, bs, dtype, kv, [1, 17, 64, 513, 1100, 2050], seed=hs + bs,
                       kv_scale=1.0 if kv == "auto" else 0.5, alibi=(hs == 192))
    run_case(ops, oracle, c, version)


@pytest.mark.parametrize("version", ["v1", "v2", "rocm"])
def test_alibi_gqa(ops, oracle, version):
    c = make_attn_case(4, 64, 8, 128, 16, torch.bfloat16, "fp8", [5, 100, 777, 2049], seed=11,
                       kv_scale=0.5, alibi=True)
    run_case(ops, oracle, c, version)


def test_blocksparse_v1_v2(ops, oracle):
    """block-sparse arguments of v1/v2 (attention_kernels.cuh:204-251, 380-388)."""
    c = make_attn_case(3, 8, 2, 128, 16, torch.bfloat16, "auto", [200, 640, 1500], seed=5)
    q, kc, vc, bt, sl = c["q"], c["kc"], c["vc"], c["bt"], c["sl"]
    one = torch.ones(1)
    args = dict(tp_rank=1, blocksparse_local_blocks=2, blocksparse_vert_stride=3,
                blocksparse_block_size=64, blocksparse_head_sliding_step=1)
    out_o = torch.zeros_like(q)
    oracle.paged_attention_v1(out_o, q, kc, vc, 2, c["scale"], bt, sl, 16, c["max_len"], None,
                              "auto", 1.0, 1.0, **args)
    out = torch.empty_like(q, device="cuda")
    ops.paged_attention_v1(out, q.cuda(), kc.cuda(), vc.cuda(), 2, c["scale"], bt.cuda(), sl.cuda(),
                           16, c["max_len"], None, "auto", one.cuda(), one.cuda(), **args)
    torch.testing.assert_close(out.cpu().float(), out_o.float(), atol=1e-3, rtol=1e-5)
    P = (c["max_len"] + 511) // 512
    S, H, D = q.shape
    es = torch.zeros((S, H, P)); ml = torch.zeros_like(es); tmp = torch.zeros((S, H, P, D), dtype=q.dtype)
    out2 = torch.empty_like(q, device="cuda")
    ops.paged_attention_v2(out2, es.cuda(), ml.cuda(), tmp.cuda(), q.cuda(), kc.cuda(), vc.cuda(), 2,
                           c["scale"], bt.cuda(), sl.cuda(), 16, c["max_len"], None, "auto",
                           one.cuda(), one.cuda(), **args)
    out_o2 = torch.zeros_like(q)
    oracle.paged_attention_v2(out_o2, es, ml, tmp, q, kc, vc, 2, c["scale"], bt, sl, 16,
                              c["max_len"], None, "auto", 1.0, 1.0, **args)
    torch.testing.assert_close(out2.cpu().float(), out_o2.float(), atol=1e-3, rtol=1e-5)


def test_rocm_query_start_loc_and_fp8_out(ops, oracle):
    """query_start_loc skips non-decode sequences and addresses q/out by token
    (csrc/rocm/attention.cu:296-299, 385-388, 1464-1467); fp8_out_scale -> e4m3 output
    (:1460-1474)."""
    lens = [40, 300, 700, 90]
    c = make_attn_case(4, 16, 2, 128, 16, torch.bfloat16, "fp8", lens, seed=9, kv_scale=0.5)
    # token rows: seq0 -> row 0, seq1 = prefill of 3 tokens (rows 1..3, skipped), seq2 -> 4, seq3 -> 5
    qsl = torch.tensor([0, 1, 4, 5, 6], dtype=torch.int32)
    g = torch.Generator().manual_seed(1)
    q_tok = ((torch.rand((6, 16, 128), generator=g) * 2 - 1) * 128 ** -0.5).to(torch.bfloat16)
    S, H, D, P = 4, 16, 128, (700 + 255) // 256
    ks = torch.tensor([0.5])
    for fp8_out in (None, torch.tensor([0.02])):
        es = torch.zeros((S, H, P)); ml = torch.zeros_like(es); tmp = torch.zeros((S, H, P, D), dtype=torch.bfloat16)
        if fp8_out is None:
            out_o = torch.full((6, H, D), 7.0, dtype=torch.bfloat16)
            out = out_o.clone().cuda()
        else:
            out_o = torch.full((6, H, D), 9, dtype=torch.uint8)
            out = out_o.clone().cuda()
        oracle.paged_attention_rocm(out_o, es, ml, tmp, q_tok, c["kc"], c["vc"], 2, c["scale"],
                                    c["bt"], c["sl"], qsl, 16, 700, None, "fp8", 0.5, 0.5, fp8_out)
        ops.paged_attention_rocm(out, es.cuda(), ml.cuda(), tmp.cuda(), q_tok.cuda(), c["kc"].cuda(),
                                 c["vc"].cuda(), 2, c["scale"], c["bt"].cuda(), c["sl"].cuda(),
                                 qsl.cuda(), 16, 700, None, "fp8", ks.cuda(), ks.cuda(),
                                 None if fp8_out is None else fp8_out.cuda())
        got = out.cpu()
        if fp8_out is None:
            torch.testing.assert_close(got.float(), out_o.float(), atol=2e-3, rtol=1e-5)
            assert (got[1:4] == 7.0).all()  # skipped rows untouched
        else:
            a = got.view(torch.float8_e4m3fn).float(); b = out_o.view(torch.float8_e4m3fn).float()
            assert (got[1:4] == 9).all()
            # one e4m3 ulp at most (the accumulators differ by ~1e-4 before rounding)
            assert ((a - b).abs() <= 0.125 * b.abs().clamp(min=2 ** -6) + 1e-6).all()
            assert (a != b).float().mean().item() < 0.05


def test_unsupported_raises(ops):
    q = torch.zeros((1, 8, 100), dtype=torch.bfloat16, device="cuda")
    kc = torch.zeros((4, 8, 100 // 4, 16, 4), dtype=torch.bfloat16, device="cuda")
    vc = torch.zeros((4, 8, 100, 16), dtype=torch.bfloat16, device="cuda")
    bt = torch.zeros((1, 1), dtype=torch.int32, device="cuda")
    sl = torch.ones(1, dtype=torch.int32, device="cuda")
    one = torch.ones(1, device="cuda")
    with pytest.raises(RuntimeError, match="Unsupported head size"):
        ops.paged_attention_v1(torch.empty_like(q), q, kc, vc, 8, 1.0, bt, sl, 16, 1, None, "auto", one, one)
    with pytest.raises(RuntimeError, match="Unsupported data type of kv cache"):
        ops.paged_attention_v1(torch.empty_like(q), q, kc, vc, 8, 1.0, bt, sl, 16, 1, None, "fp8_e5m2", one, one)


import glob as _glob
import os as _os

_GOLD = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("path", sorted(_glob.glob(_os.path.join(_GOLD, "attnfp8_ref_*.pt"))),
                         ids=lambda p: _os.path.basename(p))
@pytest.mark.parametrize("version", ["v1", "v2", "rocm"])
def test_fp8kv_reference_executed_golden(ops, path, version, v1_form):
    if version != "v1" and v1_form == "plain":
        pytest.skip("the launch form only concerns v1")
    """HIP fp8-KV decode (k_scale != v_scale) against outputs the reference's own CPU kernels
    produced on the per-element dequantised cache (tests/golden/make_golden.py attention_fp8kv;
    the route of the reference's test, test_attention.py:306-322).  Tolerance: the reference's for
    fp8 KV (:345-348) atol 1e-2 / rtol 1e-5, and our tighter 2e-3 bound."""
    d = torch.load(path, weights_only=True)
    q, kc, vc, bt, sl = (d[k].cuda() for k in ("q", "kc", "vc", "bt", "sl"))
    KVH, scale = int(d["num_kv_heads"]), float(d["scale"])
    slopes = d["slopes"].cuda() if "slopes" in d else None
    ks = d["k_scale"].reshape(1).float().cuda()
    vs = d["v_scale"].reshape(1).float().cuda()
    max_len = int(d["sl"].max())
    S, H, D = q.shape
    out = torch.full_like(q, float("nan"))
    if version == "v1":
        ops.paged_attention_v1(out, q, kc, vc, KVH, scale, bt, sl, 16, max_len, slopes, "fp8", ks, vs)
        want = d["out_v1"]
    else:
        PS = 512 if version == "v2" else 256
        P = (max_len + PS - 1) // PS
        es = torch.zeros((S, H, P), device="cuda"); ml = torch.zeros_like(es)
        tmp = torch.zeros((S, H, P, D), dtype=q.dtype, device="cuda")
        if version == "v2":
            ops.paged_attention_v2(out, es, ml, tmp, q, kc, vc, KVH, scale, bt, sl, 16, max_len,
                                   slopes, "fp8", ks, vs)
        else:
            ops.paged_attention_rocm(out, es, ml, tmp, q, kc, vc, KVH, scale, bt, sl, None, 16,
                                     max_len, slopes, "fp8", ks, vs)
        want = d["out_v2"]
    torch.cuda.synchronize()
    torch.testing.assert_close(out.cpu().float(), want.float(), atol=1e-2, rtol=1e-5)
    assert (out.cpu().float() - want.float()).abs().max().item() < 2e-3


@pytest.mark.parametrize("path", sorted(_glob.glob(_os.path.join(_GOLD, "attn_ref_*.pt"))),
                         ids=lambda p: _os.path.basename(p))
@pytest.mark.parametrize("version", ["v1", "v2", "rocm"])
def test_16_32bit_kv_reference_executed_golden(ops, path, version):
    """HIP decode over bf16 / fp16 / fp32 KV straight against the outputs of the reference's own CPU kernels
    (csrc/cpu/attention.cpp compiled as oracle/_ref, tests/golden/make_golden.py attention) -- no oracle in
    between -- at the reference's tolerance (test_attention.py:345-348): atol 1e-3, rtol 1e-5."""
    d = torch.load(path, weights_only=True)
    q, kc, vc, bt, sl = (d[k].cuda() for k in ("q", "kc", "vc", "bt", "sl"))
    if version == "rocm" and q.dtype == torch.float32:
        pytest.skip("the rocm op takes 16-bit queries only (platforms/rocm.py:138-148)")
    KVH, scale = int(d["num_kv_heads"]), float(d["scale"])
    slopes = d["slopes"].cuda() if "slopes" in d else None
    one = torch.ones(1, device="cuda")
    max_len = int(d["sl"].max())
    S, H, D = q.shape
    out = torch.full_like(q, float("nan"))
    if version == "v1":
        ops.paged_attention_v1(out, q, kc, vc, KVH, scale, bt, sl, 16, max_len, slopes, "auto", one, one)
        want = d["out_v1"]
    else:
        PS = 512 if version == "v2" else 256
        P = (max_len + PS - 1) // PS
        es = torch.zeros((S, H, P), device="cuda"); ml = torch.zeros_like(es)
        tmp = torch.zeros((S, H, P, D), dtype=q.dtype, device="cuda")
        if version == "v2":
            ops.paged_attention_v2(out, es, ml, tmp, q, kc, vc, KVH, scale, bt, sl, 16, max_len,
                                   slopes, "auto", one, one)
        else:
            ops.paged_attention_rocm(out, es, ml, tmp, q, kc, vc, KVH, scale, bt, sl, None, 16,
                                     max_len, slopes, "auto", one, one)
        want = d["out_v2"]
    torch.cuda.synchronize()
    torch.testing.assert_close(out.cpu().float(), want.float(), atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize("kv", ["auto", "fp8"])
@pytest.mark.parametrize("lens", [
    [4096, 1, 64, 65, 700, 3000, 17],             # one long sequence shared by many workgroups
    [100] * 40 + [5000],                          # many short sequences per workgroup + one long
    [64] * 3,                                     # fewer steps than workgroups
    [8192],                                       # a single sequence cut into every workgroup
    [1029, 2050, 513, 4000, 300, 16, 15, 2048, 1024, 1, 640, 3333],
])
def test_v1_balanced_ragged(ops, oracle, kv, lens):
    """Balanced v1 launches (paged_attention_mfma.hip): sequences shared between workgroups are
    merged through the workspace by the last finisher; two launches in a row reuse the
    self-resetting tickets; results equal the plain form within the stated tolerance."""
    from vllm_amd import _lib
    c = make_attn_case(len(lens), 32, 8, 128, 16, torch.bfloat16, kv, lens,
                       seed=len(lens), kv_scale=1.0 if kv == "auto" else 0.5)
    _lib.set_tuning("pa_balance", 1)
    try:
        for _ in range(2):
            run_case(ops, oracle, c, "v1")
    finally:
        _lib.set_tuning("pa_balance", 0)
    q, kc, vc, bt, sl = (to_dev(c[k]) for k in ("q", "kc", "vc", "bt", "sl"))
    sc = torch.tensor([c["kv_scale"]], device="cuda")
    outs = []
    for form in (1, 0):
        _lib.set_tuning("pa_balance", form)
        o = torch.full_like(q, float("nan"))
        ops.paged_attention_v1(o, q, kc, vc, 8, c["scale"], bt, sl, 16, c["max_len"], None, kv, sc, sc)
        outs.append(o)
    _lib.set_tuning("pa_balance", 0)
    torch.cuda.synchronize()
    assert not outs[0].isnan().any()
    torch.testing.assert_close(outs[0].float(), outs[1].float(), atol=2e-3, rtol=1e-5)


def test_v1_zero_length_sequences(ops, v1_form):
    """seq_len 0: no key, exp_sum 0 -> the reference writes zeros (attention_kernels.cuh:337)."""
    lens = [0, 5, 0, 130]
    c = make_attn_case(4, 8, 2, 128, 16, torch.bfloat16, "fp8", [1, 5, 1, 130], seed=2, kv_scale=0.5)
    q, kc, vc, bt = (to_dev(c[k]) for k in ("q", "kc", "vc", "bt"))
    sl = torch.tensor(lens, dtype=torch.int32, device="cuda")
    sc = torch.tensor([0.5], device="cuda")
    o = torch.full_like(q, float("nan"))
    ops.paged_attention_v1(o, q, kc, vc, 2, c["scale"], bt, sl, 16, 130, None, "fp8", sc, sc)
    torch.cuda.synchronize()
    assert not o.isnan().any()
    assert (o[0] == 0).all() and (o[2] == 0).all() and (o[1] != 0).any() and (o[3] != 0).any()


@pytest.mark.parametrize("shape", [(64, 32, 8), (200, 8, 2), (256, 3, 3), (40, 16, 8),
                                   (256, 8, 8), (128, 16, 8), (136, 8, 8), (250, 8, 8),   # XCD-grouped launches
                                   (512, 4, 2), (1000, 2, 2), (1024, 1, 1), (1032, 2, 2)])   # 8 / 16 registers of lengths; above the limit
@pytest.mark.parametrize("kv", ["auto", "fp8"])
def test_v1_longest_first_order(ops, oracle, shape, kv):
    """Unpartitioned launches of 257..1024 workgroups hand the sequences to the hardware workgroup
    ids longest first (pa_lpt_wg): a permutation of who computes what, so the output must be
    bit-identical to the hardware order ("pa_order" = 0) -- with ties, zero lengths, lengths above
    the max_seq_len argument (keys are clamped), and a batch of equal lengths (identity)."""
    from vllm_amd import _lib
    S, H, KVH = shape
    g = torch.Generator().manual_seed(S + H)
    lens = torch.randint(1, 400, (S,), generator=g).tolist()
    lens[3] = lens[7] = lens[S - 1] = 333          # ties
    lens[5] = 1
    c = make_attn_case(S, H, KVH, 128, 16, torch.bfloat16, kv, lens, seed=S, kv_scale=1.0 if kv == "auto" else 0.5)
    run_case(ops, oracle, c, "v1")                 # against the oracle, default order
    if H // KVH <= 16 and S <= 136:                # v2 / rocm with one super-partition per sequence: same order
        run_case(ops, oracle, c, "v2")
        run_case(ops, oracle, c, "rocm")
        # rocm op with query_start_loc (a mixed batch: rows 2 and 6 are not decode tokens and must
        # stay untouched), order on / off
        from helpers import to_dev as _d
        qsl = torch.arange(S + 1, dtype=torch.int32)
        qsl[3:] += 4
        qsl[7:] += 9
        Tq = int(qsl[-1])
        qbig = torch.zeros((Tq, H, 128), dtype=torch.bfloat16)
        rows = qsl[:-1].long()
        qbig[rows] = c["q"]
        P = (c["max_len"] + 255) // 256
        res = []
        for order in (1, 0):
            _lib.set_tuning("pa_order", order)
            o = torch.full((Tq, H, 128), 7.0, dtype=torch.bfloat16, device="cuda")
            es = torch.zeros((S, H, P), dtype=torch.float32, device="cuda")
            tmp = torch.zeros((S, H, P, 128), dtype=torch.bfloat16, device="cuda")
            ks = torch.tensor([c["kv_scale"]], device="cuda")
            ops.paged_attention_rocm(o, es, torch.zeros_like(es), tmp, _d(qbig), _d(c["kc"]), _d(c["vc"]), KVH,
                                     c["scale"], _d(c["bt"]), _d(c["sl"]), _d(qsl), 16, c["max_len"], None, kv, ks, ks)
            res.append(o)
        _lib.set_tuning("pa_order", 1)
        torch.cuda.synchronize()
        assert torch.equal(res[0], res[1])
        keep = torch.ones(Tq, dtype=torch.bool)
        keep[rows] = False
        keep[rows[2]] = keep[rows[6]] = True     # sequences 2 and 6 have 5 and 10 query tokens: skipped
        assert bool((res[0].cpu()[keep] == 7.0).all())
        dec = [i for i in range(S) if i not in (2, 6)]
        ref = torch.zeros_like(c["q"])
        oracle.paged_attention_v1(ref, c["q"], c["kc"], c["vc"], KVH, c["scale"], c["bt"], c["sl"], 16,
                                  c["max_len"], None, kv, c["kv_scale"], c["kv_scale"])
        torch.testing.assert_close(res[0].cpu()[rows[dec]].float(), ref[dec].float(), **TOL["auto" if kv == "auto" else "fp8"])
    q, kc, vc, bt = (to_dev(c[k]) for k in ("q", "kc", "vc", "bt"))
    sc = torch.tensor([c["kv_scale"]], device="cuda")
    zero = list(lens)
    zero[0] = zero[9] = 0
    for name, ll, max_len in (("ragged", lens, c["max_len"]), ("clamped keys", lens, 64),
                              ("zero lengths", zero, c["max_len"]), ("equal", [257] * S, 257)):
        sl = torch.tensor(ll, dtype=torch.int32, device="cuda")
        outs = []
        for order in (1, 0):
            _lib.set_tuning("pa_order", order)
            o = torch.full_like(q, float("nan"))
            ops.paged_attention_v1(o, q, kc, vc, KVH, c["scale"], bt, sl, 16, max_len, None, kv, sc, sc)
            outs.append(o)
        _lib.set_tuning("pa_order", 1)
        torch.cuda.synchronize()
        assert not outs[0].isnan().any(), name
        assert torch.equal(outs[0], outs[1]), name

"""GPU edge cases through the op surface: empty batches, empty sequences inside a batch,
padding slots, zero-row GEMMs.  Nothing may fault or touch memory it was not given."""
import pytest
import torch

from helpers import make_attn_case
from test_gpu_parity_attention import run_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("version", ["v1", "v2", "rocm"])
def test_attention_empty_batch(ops, version):
    c = make_attn_case(1, 8, 2, 128, 16, torch.bfloat16, "fp8", [40], seed=1, kv_scale=0.5)
    q = torch.empty((0, 8, 128), dtype=torch.bfloat16, device="cuda")
    out = torch.empty_like(q)
    bt = torch.empty((0, 4), dtype=torch.int32, device="cuda")
    sl = torch.empty((0,), dtype=torch.int32, device="cuda")
    one = torch.ones(1, device="cuda")
    es = torch.empty((0, 8, 1), device="cuda"); ml = torch.empty_like(es)
    tmp = torch.empty((0, 8, 1, 128), dtype=torch.bfloat16, device="cuda")
    kc, vc = c["kc"].cuda(), c["vc"].cuda()
    if version == "v1":
        ops.paged_attention_v1(out, q, kc, vc, 2, 1.0, bt, sl, 16, 64, None, "fp8", one, one)
    elif version == "v2":
        ops.paged_attention_v2(out, es, ml, tmp, q, kc, vc, 2, 1.0, bt, sl, 16, 64, None, "fp8", one, one)
    else:
        ops.paged_attention_rocm(out, es, ml, tmp, q, kc, vc, 2, 1.0, bt, sl, None, 16, 64, None,
                                 "fp8", one, one)
    torch.cuda.synchronize()


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("kv", ["auto", "fp8"])
@pytest.mark.parametrize("hs", [128, 256])   # MFMA and generic kernels
def test_attention_empty_sequence_in_batch(ops, oracle, version, kv, hs):
    """seq_len 0: no logits, exp_sum 0, every accumulator 0 => the row is 0
    (attention_kernels.cuh:160-166, 330-340, 437-470)."""
    lens = [0, 5, 0, 700, 0]
    c = make_attn_case(len(lens), 8, 2, hs, 16, torch.bfloat16, kv, [max(l, 1) for l in lens],
                       seed=2, kv_scale=1.0 if kv == "auto" else 0.5)
    c["sl"] = torch.tensor(lens, dtype=torch.int32)
    run_case(ops, oracle, c, version)


def test_cache_write_empty_and_padding(ops):
    NB, KVH, D, BS = 6, 2, 128, 16
    for kv, cdt in (("auto", torch.bfloat16), ("fp8", torch.uint8)):
        x = 8 if kv == "auto" else 16
        kc = torch.full((NB, KVH, D // x, BS, x), 3, dtype=cdt, device="cuda")
        vc = torch.full((NB, KVH, D, BS), 3, dtype=cdt, device="cuda")
        fk = torch.full((NB, BS, KVH, D), 3, dtype=cdt, device="cuda")
        fv = fk.clone()
        one = torch.ones(1, device="cuda")
        for T, slots in ((0, []), (5, [-1] * 5)):
            k = torch.randn((T, KVH, D), device="cuda").to(torch.bfloat16)
            sm = torch.tensor(slots, dtype=torch.int64, device="cuda")
            ops.reshape_and_cache(k, k, kc, vc, sm, kv, one, one)
            ops.reshape_and_cache_flash(k, k, fk, fv, sm, kv, one, one)
            torch.cuda.synchronize()
            for t in (kc, vc, fk, fv):
                assert (t == 3).all()


def test_quant_and_gemm_zero_rows(ops):
    x = torch.empty((0, 4096), dtype=torch.bfloat16, device="cuda")
    for kw in (dict(), dict(use_per_token_if_dynamic=True), dict(scale=torch.ones(1, device="cuda"))):
        q, s = ops.scaled_fp8_quant(x, **kw)
        assert q.shape == (0, 4096)
    a = torch.empty((0, 4096), dtype=torch.float8_e4m3fn, device="cuda")
    b = torch.zeros((4096, 4096), dtype=torch.uint8, device="cuda").view(torch.float8_e4m3fn).t()
    sa = torch.empty((0, 1), device="cuda")
    sb = torch.ones((4096, 1), device="cuda")
    out = ops.cutlass_scaled_mm(a, b, sa, sb, torch.bfloat16)
    assert out.shape == (0, 4096)
    torch.cuda.synchronize()

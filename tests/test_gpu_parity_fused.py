"""GPU parity of the ops next to the hot path: fused RMSNorm + fp8 quant,
SiLU-mul + fp8 quant, the device-side slot mapping and swap_blocks, through torch.ops ->
C ABI -> HIP, against the CPU oracle.

Stated tolerance (the reference's own tests use allclose on the dequantised fp8 values,
test_fused_quant_layernorm.py:156-166, test_fused_quant_activation.py:67-70):
  * integer / copy ops (slot mapping, swap_blocks, residual write-back): bit-exact;
  * per-token scales: rtol 1e-5 (the variance is summed in a different order), except that
    <= 1 % of the tokens may differ by one ulp of the INPUT dtype: the scale is the max of
    values rounded to that dtype, and 1/rms one fp32 ulp apart can flip that rounding;
  * fp8 codes: identical except <= 0.3 % that may sit ONE code apart (a value within fp32
    rounding of an fp8 / bf16 tie: summation order of the variance, expf of the device
    library vs glibc).
"""
import os

import pytest
import torch

from helpers import codes_close

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float16, torch.float32]
# test_fused_quant_layernorm.py:16-23 (minus the 4096-token rows: same code path, and the
# oracle stays within seconds), plus hidden > 8192 (the kernel's uncached path)
SHAPES = [*[(1, h) for h in (1, 64, 1024, 1025, 1026, 1027, 1028, 1029, 5120, 5137)],
          *[(83, h) for h in (1, 1033, 2048, 5120)],
          *[(2048, h) for h in (1, 64, 1024, 1027)], (257, 8192), (33, 16384), (5, 8200)]
EPS = 1e-6


def mk(T, H, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn((T, H), generator=g) * 0.5).to(dtype)
    res = (torch.randn((T, H), generator=g) * 0.5).to(dtype)
    w = (1.0 + 0.1 * torch.randn(H, generator=g)).to(dtype)
    return x, res, w


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dtype", DTYPES)
def test_rms_norm_static_fp8_quant(ops, oracle, shape, dtype):
    T, H = shape
    x, res, w = mk(T, H, dtype, T + H)
    scale = torch.tensor([0.013])
    out = torch.empty((T, H), dtype=torch.float8_e4m3fn, device="cuda")
    ops.scaled_rms_norm(out, x.cuda(), w.cuda(), scale.cuda(), EPS)
    codes_close(out.cpu(), oracle.rms_norm_static_fp8_quant(x, w, 0.013, EPS), 3e-3)

    r_gpu = res.cuda()
    ops.scaled_fused_add_rms_norm(out, x.cuda(), r_gpu, w.cuda(), scale.cuda(), EPS)
    r_ref = res.clone()
    ref = oracle.fused_add_rms_norm_static_fp8_quant(x, r_ref, w, 0.013, EPS)
    assert torch.equal(r_gpu.cpu(), r_ref)
    codes_close(out.cpu(), ref, 3e-3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_rms_norm_static_strided_input(ops, oracle, dtype):
    """input.stride(-2) > hidden (layernorm_quant_kernels.cu:179: input_stride)."""
    T, H = 19, 1024
    x, res, w = mk(T, 2 * H, dtype, 5)
    xs = x[:, :H]
    out = torch.empty((T, H), dtype=torch.float8_e4m3fn, device="cuda")
    scale = torch.tensor([0.02])
    ops.scaled_rms_norm(out, x.cuda()[:, :H], w[:H].cuda(), scale.cuda(), EPS)
    codes_close(out.cpu(), oracle.rms_norm_static_fp8_quant(xs, w[:H].contiguous(), 0.02, EPS), 3e-3)
    r_gpu, r_ref = res[:, :H].contiguous().cuda(), res[:, :H].contiguous()
    ops.scaled_fused_add_rms_norm(out, x.cuda()[:, :H], r_gpu, w[:H].cuda(), scale.cuda(), EPS)
    ref = oracle.fused_add_rms_norm_static_fp8_quant(xs, r_ref, w[:H].contiguous(), 0.02, EPS)
    assert torch.equal(r_gpu.cpu(), r_ref)
    codes_close(out.cpu(), ref, 3e-3)


def check_dynamic(q, s, q_ref, s_ref, dtype, H):
    ulp = {torch.float32: 1e-5, torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}[dtype]
    rel = ((s - s_ref).abs() / s_ref).flatten()
    # two nested roundings (x * rms, then * weight): up to ~2 ulp of the dtype
    assert rel.max().item() <= ulp * 2.5, f"scale off by {rel.max().item():.3g}"
    same = rel <= 1e-5
    assert (~same).float().mean().item() <= 0.01 or rel.numel() < 100 and (~same).sum() <= 1
    q, q_ref = q.view(torch.uint8).reshape(-1, H), q_ref.view(torch.uint8).reshape(-1, H)
    if same.any():
        codes_close(q[same], q_ref[same], 3e-3 if H > 8 else 1.0)
    if (~same).any():   # rescaled by one dtype ulp: a few more codes sit one step apart
        codes_close(q[~same], q_ref[~same], 0.1 if H > 8 else 1.0)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("add_residual", [False, True])
@pytest.mark.parametrize("use_ub", [False, True])
def test_rms_norm_dynamic_per_token_quant(ops, oracle, shape, dtype, add_residual, use_ub):
    T, H = shape
    x, res, w = mk(T, H, dtype, 2 * T + H)
    r_ref = res.clone() if add_residual else None
    ub = None
    if use_ub:
        _, s0 = oracle.rms_norm_dynamic_per_token_quant(x, w, EPS, None,
                                                        res.clone() if add_residual else None)
        ub = (s0.mean() * 448.0).reshape(1)   # an upper bound that clips about half the tokens
    q_ref, s_ref = oracle.rms_norm_dynamic_per_token_quant(x, w, EPS, ub, r_ref)
    r_gpu = res.cuda() if add_residual else None
    q, s = ops.rms_norm_dynamic_per_token_quant(x.cuda(), w.cuda(), EPS, torch.float8_e4m3fn,
                                                None if ub is None else ub.cuda(), r_gpu)
    assert q.dtype == torch.float8_e4m3fn and s.shape == (T, 1)
    if add_residual:
        assert torch.equal(r_gpu.cpu(), r_ref)
    check_dynamic(q.cpu(), s.cpu(), q_ref, s_ref, dtype, H)


@pytest.mark.parametrize("shape", [(1, 64), (7, 1027), (83, 2048), (33, 5120), (5, 8200)])
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("add_residual", [False, True])
def test_rms_norm_dynamic_per_token_quant_int8(ops, shape, dtype, add_residual):
    """int8 output (fused_layernorm_dynamic_per_token_quant.cu:33-36): against the reference test's own
    route (test_fused_quant_layernorm.py:40-74, 156-166): RMSNorm.forward_native, then the per-token int8
    quant of tests/kernels/quant_utils.py:22-62 (scale = amax / 127, round(x * (1 / scale)), clamp), restated
    in torch on the CPU.  Tolerance: the reference's atol = 1 on the codes; scales rtol 1e-5 up to one ulp of
    the input dtype in the row maximum; the residual write-back is exact.  Parity unpinned beyond that (no
    reference execution possible here)."""
    T, H = shape
    x, res, w = mk(T, H, dtype, T + H + 7)
    xf = x.float() + (res.float() if add_residual else 0.0)
    new_res = xf.to(dtype)
    var = xf.pow(2).mean(dim=-1, keepdim=True)
    y = ((xf * torch.rsqrt(var + EPS)).to(dtype) * w).float()
    scales = (y.abs().amax(dim=-1, keepdim=True) / 127.0).clamp(min=torch.finfo(torch.float32).eps)
    want = (y * (1.0 / scales)).round().clamp(-128, 127).to(torch.int8)
    r = res.cuda().clone() if add_residual else None
    q, s = ops.rms_norm_dynamic_per_token_quant(x.cuda(), w.cuda(), EPS, torch.int8, None, r)
    assert q.dtype == torch.int8 and q.shape == x.shape
    tol = {torch.float32: 1e-5, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
    torch.testing.assert_close(s.cpu(), scales, rtol=tol, atol=0)
    assert (q.cpu().int() - want.int()).abs().max().item() <= 1
    assert (q.cpu() != want).float().mean().item() <= 0.02
    if add_residual:
        assert torch.equal(r.cpu(), new_res)
    with pytest.raises(RuntimeError, match="scale_ub"):
        ops.rms_norm_dynamic_per_token_quant(x.cuda(), w.cuda(), EPS, torch.int8,
                                             torch.tensor([1.0], device="cuda"), None)


@pytest.mark.parametrize("num_tokens", [1, 17, 86, 1234])
@pytest.mark.parametrize("hidden", [16, 48, 128, 1562, 4096])   # test_fused_quant_activation.py:13-14
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_silu_and_mul_quant(ops, oracle, num_tokens, hidden, dtype):
    g = torch.Generator().manual_seed(num_tokens + hidden)
    x = torch.randn((num_tokens, hidden), generator=g).to(dtype)
    for scale in (0.37, -1.2):   # the reference draws the scale from randn: it may be negative
        out = torch.empty((num_tokens, hidden // 2), dtype=torch.float8_e4m3fn, device="cuda")
        ops.silu_and_mul_quant(out, x.cuda(), torch.tensor([scale]).cuda())
        codes_close(out.cpu(), oracle.silu_and_mul_quant(x, scale), 3e-3)


@pytest.mark.parametrize("tag", ["bf16", "fp16", "fp32"])
def test_fused_quant_golden_gpu(ops, tag):
    """The committed fixtures (torch restatement of the reference tests' native side)."""
    d = torch.load(os.path.join(os.path.dirname(__file__), "golden", f"fused_quant_{tag}.pt"))
    x, res, w = d["x"].cuda(), d["res"], d["w"].cuda()
    eps, scale = float(d["eps"]), d["scale"].reshape(1).cuda()
    out = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device="cuda")
    ops.scaled_rms_norm(out, x, w, scale, eps)
    codes_close(out.cpu(), d["static_q"], 3e-3)
    r = res.cuda()
    ops.scaled_fused_add_rms_norm(out, x, r, w, scale, eps)
    assert torch.equal(r.cpu(), d["add_static_res"])
    codes_close(out.cpu(), d["add_static_q"], 3e-3)
    dt, H = d["x"].dtype, d["x"].shape[1]
    q, s = ops.rms_norm_dynamic_per_token_quant(x, w, eps, torch.float8_e4m3fn)
    check_dynamic(q.cpu(), s.cpu(), d["dyn_q"], d["dyn_s"], dt, H)
    r = res.cuda()
    q, s = ops.rms_norm_dynamic_per_token_quant(x, w, eps, torch.float8_e4m3fn, None, r)
    assert torch.equal(r.cpu(), d["dynres_res"])
    check_dynamic(q.cpu(), s.cpu(), d["dynres_q"], d["dynres_s"], dt, H)
    if tag != "fp32":
        xs = d["silu_x"].cuda()
        o = torch.empty((xs.shape[0], xs.shape[1] // 2), dtype=torch.float8_e4m3fn, device="cuda")
        ops.silu_and_mul_quant(o, xs, d["silu_scale"].reshape(1).cuda())
        codes_close(o.cpu(), d["silu_q"], 3e-3)


# ---- slot mapping / BlockTable --------------------------------------------------------------
def test_slot_mapping_golden_gpu():
    from vllm_amd import _torch_ops
    d = torch.load(os.path.join(os.path.dirname(__file__), "golden", "slot_mapping.pt"))
    out = torch.full((d["req"].shape[0] + 5,), -7, dtype=torch.int64, device="cuda")
    _torch_ops.compute_slot_mapping(out, d["bt"].cuda(), d["req"].cuda(), d["pos"].cuda(),
                                    int(d["block_size"]))
    assert torch.equal(out[:-5].cpu(), d["slots"])
    assert bool((out[-5:] == -7).all())   # nothing written past num_tokens


@pytest.mark.parametrize("block_size", [16, 32, 24])
def test_block_table_device_slot_mapping(oracle, block_size):
    """BlockTable mirror: host row edits + commit, device compute_slot_mapping, vs the numpy
    arithmetic of the reference (block_table.py:92-99) restated in the oracle."""
    from vllm_amd.worker import BlockTable
    g = torch.Generator().manual_seed(block_size)
    max_reqs, max_model_len, max_tokens = 9, 1000, 4096
    mb = -(-max_model_len // block_size)   # not divisible by block_size on purpose (:89-91)
    bt = BlockTable(block_size, max_reqs, mb, max_tokens, False, torch.device("cuda"))
    free = torch.randperm(max_reqs * mb, generator=g).tolist()
    for r in range(max_reqs):
        n = int(torch.randint(1, mb + 1, (1,), generator=g))
        ids = [free.pop() for _ in range(n)]
        bt.add_row(ids[: n // 2], r)
        bt.append_row(ids[n // 2:], r)
    bt.swap_row(0, 3)
    bt.move_row(5, 1)
    bt.commit_block_table(max_reqs)
    # a mixed prefill/decode batch: positions inside each request's allocated range
    req, pos = [], []
    for r in range(max_reqs):
        ntok = int(bt.num_blocks_per_row[r]) * block_size
        k = int(torch.randint(1, 200, (1,), generator=g))
        p = torch.randint(0, min(ntok, max_model_len), (k,), generator=g)
        req += [r] * k
        pos += p.tolist()
    req_t = torch.tensor(req, dtype=torch.int32)
    pos_t = torch.tensor(pos, dtype=torch.int64)
    bt.compute_slot_mapping(req_t.cuda(), pos_t.cuda())
    bt.commit_slot_mapping(len(req))
    ref = oracle.compute_slot_mapping(bt.get_cpu_tensor(), req_t, pos_t, block_size)
    assert torch.equal(bt.slot_mapping[: len(req)].cpu(), ref)
    assert torch.equal(bt.get_device_tensor().cpu(), bt.get_cpu_tensor())


# ---- swap_blocks (tests/kernels/attention/test_cache.py:353-431) ------------------------
@pytest.mark.parametrize("direction", [("cuda", "cpu"), ("cpu", "cuda"), ("cuda", "cuda")])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.uint8])
def test_swap_blocks(ops, direction, dtype):
    src_dev, dst_dev = direction
    g = torch.Generator().manual_seed(3)
    NB, shape = 40, (8, 16, 16, 8)
    src = torch.randint(0, 200, (NB,) + shape, generator=g).to(dtype)
    dst = torch.randint(0, 200, (NB,) + shape, generator=g).to(dtype)
    src_d = src.to(src_dev).pin_memory() if src_dev == "cpu" else src.to(src_dev)
    dst_d = dst.to(dst_dev).pin_memory() if dst_dev == "cpu" else dst.to(dst_dev)
    srcs = torch.randperm(NB, generator=g)[:11].tolist()
    dsts = torch.randperm(NB, generator=g)[:11].tolist()
    if src_dev == dst_dev:   # in the same tensor space the reference test keeps them disjoint
        dsts = [d for d in range(NB) if d not in srcs][:11]
    mapping = torch.tensor(list(zip(srcs, dsts)), dtype=torch.int64)
    ops.swap_blocks(src_d, dst_d, mapping)
    torch.cuda.synchronize()
    expect = dst.clone()
    for s, d_ in zip(srcs, dsts):
        expect[d_] = src[s]
    assert torch.equal(dst_d.cpu(), expect)
    assert torch.equal(src_d.cpu(), src)

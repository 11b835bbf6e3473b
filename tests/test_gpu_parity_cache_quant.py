"""GPU parity, bit-exact: KV-cache writes, convert_fp8, copy_blocks and the three
fp8 quantisation modes vs the CPU oracle.  Shapes follow the reference's
tests/kernels/attention/test_cache.py:121-350 and
tests/kernels/quantization/test_fp8_quant.py:14-117."""
import pytest
import torch

from helpers import x_of

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float16, torch.float32]


def _kv_inputs(T, H, D, dtype, seed, strided):
    g = torch.Generator().manual_seed(seed)
    if strided:  # key/value as slices of a fused qkv tensor: stride(0) != H*D
        qkv = torch.randn((T, 3, H, D), generator=g).to(dtype)
        return qkv[:, 1], qkv[:, 2]
    return torch.randn((T, H, D), generator=g).to(dtype), torch.randn((T, H, D), generator=g).to(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kv", ["auto", "fp8"])
@pytest.mark.parametrize("D", [64, 80, 120, 128, 256])
@pytest.mark.parametrize("BS", [8, 16, 32])
def test_reshape_and_cache(ops, oracle, dtype, kv, D, BS):
    if kv != "auto" and D % 16:
        pytest.skip("fp8 cache needs head_size % 16 == 0 (x = 16), as in the reference tests")
    T, H, NB = 42, 8, 64
    key, value = _kv_inputs(T, H, D, dtype, D + BS, strided=(D == 128))
    x = x_of(dtype, kv)
    cdt = dtype if kv == "auto" else torch.uint8
    g = torch.Generator().manual_seed(5)
    slots = torch.randperm(NB * BS, generator=g)[:T].to(torch.int64)
    slots[3] = -1
    slots[17] = -1
    ks = (key.float().abs().max() / 64.0).item() if kv != "auto" else 1.0
    vs = (value.float().abs().max() / 64.0).item() if kv != "auto" else 1.0
    kc0 = torch.zeros((NB, H, D // x, BS, x), dtype=cdt)
    vc0 = torch.zeros((NB, H, D, BS), dtype=cdt)
    kc_o, vc_o = kc0.clone(), vc0.clone()
    oracle.reshape_and_cache(key, value, kc_o, vc_o, slots, kv, ks, vs)
    kc_d, vc_d = kc0.cuda(), vc0.cuda()
    ops.reshape_and_cache(key.cuda(), value.cuda(), kc_d, vc_d, slots.cuda(), kv,
                          torch.tensor([ks], device="cuda"), torch.tensor([vs], device="cuda"))
    assert torch.equal(kc_d.cpu().view(torch.uint8), kc_o.view(torch.uint8))
    assert torch.equal(vc_d.cpu().view(torch.uint8), vc_o.view(torch.uint8))


def _prefill_slots(T, NB, BS, seed):
    """Slot mapping of a mixed batch: runs of consecutive slots that start and end anywhere
    in a block (chunked prefill), a stretch of scattered decode slots, padding (-1)."""
    g = torch.Generator().manual_seed(seed)
    blocks = torch.randperm(NB, generator=g).tolist()
    slots, bi = [], 0
    starts = [0, 5, 16, 9, 31, 1, 0, 13]
    lens = [64, 37, 16, 50, 7, 100, 3, 29]
    i = 0
    while len(slots) < T:
        start, ln = starts[i % 8] % BS, lens[i % 8] + (i // 8)
        i += 1
        if i % 5 == 4:   # decode-like: scattered single slots and padding
            for j in range(ln // 4):
                slots.append(-1 if j % 6 == 5 else blocks[bi] * BS + (j * 7) % BS)
                bi += 1
            continue
        pos = start
        for _ in range(ln):   # one sequence: consecutive offsets, next block when one is full
            slots.append(blocks[bi] * BS + pos)
            pos += 1
            if pos == BS:
                pos, bi = 0, bi + 1
        bi += 1
    return torch.tensor(slots[:T], dtype=torch.int64)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kv", ["auto", "fp8"])
@pytest.mark.parametrize("D,BS,T", [(128, 16, 1000), (128, 32, 517), (64, 16, 333), (256, 16, 260),
                                    (80, 16, 300), (112, 32, 256)])
def test_reshape_and_cache_prefill_runs(ops, oracle, dtype, kv, D, BS, T):
    """num_tokens >= 256 takes the run-aware kernel (LDS-transposed V rows); the slot mapping
    mixes every case it distinguishes.  Also checks that nothing outside the slots is touched."""
    if kv != "auto" and D % 16:
        pytest.skip("fp8 cache needs head_size % 16 == 0 (x = 16), as in the reference tests")
    H, NB = 4, 400
    key, value = _kv_inputs(T, H, D, dtype, D + BS + T, strided=(D == 128))
    x = x_of(dtype, kv)
    cdt = dtype if kv == "auto" else torch.uint8
    slots = _prefill_slots(T, NB, BS, T)
    ks = (key.float().abs().max() / 64.0).item() if kv != "auto" else 1.0
    vs = (value.float().abs().max() / 64.0).item() if kv != "auto" else 1.0
    g = torch.Generator().manual_seed(9)
    kc0 = torch.randint(0, 120, (NB, H, D // x, BS, x), generator=g).to(cdt)
    vc0 = torch.randint(0, 120, (NB, H, D, BS), generator=g).to(cdt)
    kc_o, vc_o = kc0.clone(), vc0.clone()
    oracle.reshape_and_cache(key, value, kc_o, vc_o, slots, kv, ks, vs)
    kc_d, vc_d = kc0.cuda(), vc0.cuda()
    ops.reshape_and_cache(key.cuda(), value.cuda(), kc_d, vc_d, slots.cuda(), kv,
                          torch.tensor([ks], device="cuda"), torch.tensor([vs], device="cuda"))
    assert torch.equal(kc_d.cpu().view(torch.uint8), kc_o.view(torch.uint8))
    assert torch.equal(vc_d.cpu().view(torch.uint8), vc_o.view(torch.uint8))


@pytest.mark.parametrize("kv", ["auto", "fp8"])
@pytest.mark.parametrize("lens", [[1000, 777, 1333, 15, 1, 16, 17, 530], [31] * 40, [4100]])
@pytest.mark.parametrize("first", [0, 5])
def test_reshape_and_cache_batched_prompts(ops, oracle, kv, lens, first):
    """A batch of prompts of odd lengths: every sequence starts at its own block (the first one `first` slots in, a
    chunked-prefill continuation) but at a token index that is no multiple of 16 -- the run kernel's workgroups
    follow the slots' 16-groups, whole groups inside a run, partial ones at its two ends."""
    D, BS, H, dtype = 128, 16, 8, torch.bfloat16
    T = sum(lens)
    NB = sum((ln + first + BS - 1) // BS for ln in lens) + 3
    g = torch.Generator().manual_seed(T + first)
    blocks = torch.randperm(NB, generator=g).tolist()
    slots, bi = [], 0
    for si, ln in enumerate(lens):
        for pos in range(first if si == 0 else 0, ln + (first if si == 0 else 0)):
            slots.append(blocks[bi + pos // BS] * BS + pos % BS)
        bi += (ln + first + BS - 1) // BS
    slots = torch.tensor(slots, dtype=torch.int64)
    key, value = _kv_inputs(T, H, D, dtype, T, strided=False)
    x = x_of(dtype, kv)
    cdt = dtype if kv == "auto" else torch.uint8
    ks, vs = (0.05, 0.04) if kv != "auto" else (1.0, 1.0)
    kc0 = torch.randint(0, 120, (NB, H, D // x, BS, x), generator=g).to(cdt)
    vc0 = torch.randint(0, 120, (NB, H, D, BS), generator=g).to(cdt)
    kc_o, vc_o = kc0.clone(), vc0.clone()
    oracle.reshape_and_cache(key, value, kc_o, vc_o, slots, kv, ks, vs)
    kc_d, vc_d = kc0.cuda(), vc0.cuda()
    ops.reshape_and_cache(key.cuda(), value.cuda(), kc_d, vc_d, slots.cuda(), kv,
                          torch.tensor([ks], device="cuda"), torch.tensor([vs], device="cuda"))
    assert torch.equal(kc_d.cpu().view(torch.uint8), kc_o.view(torch.uint8))
    assert torch.equal(vc_d.cpu().view(torch.uint8), vc_o.view(torch.uint8))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kv", ["auto", "fp8"])
@pytest.mark.parametrize("layout", ["NHD", "HND"])
@pytest.mark.parametrize("D", [64, 120, 128])
def test_reshape_and_cache_flash(ops, oracle, dtype, kv, layout, D):
    T, H, NB, BS = 50, 4, 40, 16
    key, value = _kv_inputs(T + 6, H, D, dtype, D, strided=False)   # key longer than slot_mapping
    cdt = dtype if kv == "auto" else torch.uint8
    g = torch.Generator().manual_seed(6)
    slots = torch.randperm(NB * BS, generator=g)[:T].to(torch.int64)
    slots[0] = -1
    ks, vs = (0.37, 0.21) if kv != "auto" else (1.0, 1.0)
    if layout == "NHD":
        mk = lambda: torch.zeros((NB, BS, H, D), dtype=cdt)
        view = lambda t: t
    else:
        mk = lambda: torch.zeros((NB, H, BS, D), dtype=cdt)
        view = lambda t: t.permute(0, 2, 1, 3)
    kc_o, vc_o = mk(), mk()
    oracle.reshape_and_cache_flash(key, value, view(kc_o), view(vc_o), slots, kv, ks, vs)
    kc_d, vc_d = mk().cuda(), mk().cuda()
    ops.reshape_and_cache_flash(key.cuda(), value.cuda(), view(kc_d), view(vc_d), slots.cuda(), kv,
                                torch.tensor([ks], device="cuda"), torch.tensor([vs], device="cuda"))
    assert torch.equal(kc_d.cpu().view(torch.uint8), kc_o.view(torch.uint8))
    assert torch.equal(vc_d.cpu().view(torch.uint8), vc_o.view(torch.uint8))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("scales", [(0.37, 0.21), (1e-30, 3.0), (-0.5, 0.125)])
@pytest.mark.parametrize("op", ["flash", "x_runs", "x_tokens"])
def test_cache_write_special_values(ops, oracle, dtype, scales, op):
    """fp8 cache writes of rows that hold +-0, +-inf, NaN, the largest finite values and values
    whose quotient overflows (tiny scale), next to ordinary ones (the vectorised conversion takes
    its scalar path for such a vector, wave-uniformly): byte-exact with the oracle, NaN codes up
    to their sign; also a dense sweep of ordinary values through the vector path."""
    T, H, D, NB, BS = 64, 4, 128, 16, 16
    g = torch.Generator().manual_seed(17)
    key = (torch.randn((T, H, D), generator=g) * 30).to(dtype)
    value = (torch.randn((T, H, D), generator=g) * 0.3).to(dtype)
    big = torch.finfo(dtype).max
    spec = torch.tensor([0.0, -0.0, float("inf"), float("-inf"), float("nan"), big, -big, 1e-7, -448.0, 448.0])
    for t in range(0, T, 5):      # some rows ordinary, some special
        key[t, t % H, (7 * t) % (D - 10):(7 * t) % (D - 10) + 10] = spec.to(dtype)
        value[t, (t + 1) % H, (3 * t) % (D - 10):(3 * t) % (D - 10) + 10] = spec.flip(0).to(dtype)
    ks, vs = scales
    slots = torch.arange(T, dtype=torch.int64) + 16 if op == "x_runs" else \
        torch.randperm(NB * BS, generator=g)[:T].to(torch.int64)
    if op == "flash":
        mk = lambda: torch.zeros((NB, BS, H, D), dtype=torch.uint8)
        fo, fd = oracle.reshape_and_cache_flash, ops.reshape_and_cache_flash
    else:
        mk = None
        fo, fd = oracle.reshape_and_cache, ops.reshape_and_cache
    if op == "x_runs":           # >= 256 tokens take the run kernel
        reps = 5
        key, value = key.repeat(reps, 1, 1), value.repeat(reps, 1, 1)
        slots = torch.arange(T * reps, dtype=torch.int64) + 16
        NB = T * reps // BS + 2
    if mk is None:
        kc_o = torch.zeros((NB, H, D // 16, BS, 16), dtype=torch.uint8)
        vc_o = torch.zeros((NB, H, D, BS), dtype=torch.uint8)
    else:
        kc_o, vc_o = mk(), mk()
    kc_d, vc_d = kc_o.clone().cuda(), vc_o.clone().cuda()
    fo(key, value, kc_o, vc_o, slots, "fp8", ks, vs)
    fd(key.cuda(), value.cuda(), kc_d, vc_d, slots.cuda(), "fp8", torch.tensor([ks], device="cuda"),
       torch.tensor([vs], device="cuda"))
    for got, want in ((kc_d.cpu(), kc_o), (vc_d.cpu(), vc_o)):
        nan = (want & 0x7f) == 0x7f
        assert torch.equal(got[~nan], want[~nan])
        assert ((got[nan] & 0x7f) == 0x7f).all()


@pytest.mark.parametrize("dtype", DTYPES)
def test_convert_fp8_roundtrip_and_exhaustive(ops, oracle, dtype):
    # every e4m3 code decodes exactly like the oracle's table
    codes = torch.arange(256, dtype=torch.uint8).repeat(4)
    dec = torch.empty(codes.shape, dtype=dtype, device="cuda")
    ops.convert_fp8(dec, codes.cuda(), 1.0, "fp8")
    tab = oracle.fp8_decode_table().repeat(4).to(dtype)
    got = dec.cpu()
    nan = tab.float().isnan()
    it = torch.int16 if dtype != torch.float32 else torch.int32
    # the sign of a zero result is not checked (hipcc's fp32->fp16 path returns +0 for code 0x80)
    bad = (got.view(it) != tab.view(it)) & ~nan & ~((got.float() == 0) & (tab.float() == 0))
    assert not bad.any(), (codes[bad][:8], got[bad][:8], tab[bad][:8])
    assert got[nan].float().isnan().all()
    # encode: dense sweep incl. ties, subnormals, saturation
    g = torch.Generator().manual_seed(0)
    t = oracle.fp8_decode_table()
    fin = t[~t.isnan()].sort().values
    mids = (fin[:-1] + fin[1:]) / 2
    x = torch.cat([torch.randn(50000, generator=g) * 100, torch.randn(50000, generator=g) * 0.02,
                   torch.linspace(-500, 500, 40001), fin, mids, mids * (1 + 2 ** -10),
                   mids * (1 - 2 ** -10), torch.tensor([0.0, -0.0, 1e-30, -1e-30, 3e38, -3e38])]).to(dtype)
    for scale in (1.0, 0.37):
        out_o = torch.empty(x.shape, dtype=torch.uint8)
        oracle.convert_fp8(out_o, x, scale, "fp8")
        out = torch.empty(x.shape, dtype=torch.uint8, device="cuda")
        ops.convert_fp8(out, x.cuda(), scale, "fp8")
        assert torch.equal(out.cpu(), out_o)


def test_fp8_nan_inf_encoding(ops, oracle):
    x = torch.tensor([float("nan"), float("inf"), float("-inf"), -float("nan")], dtype=torch.float32)
    out = torch.empty(4, dtype=torch.uint8, device="cuda")
    ops.convert_fp8(out, x.cuda(), 1.0, "fp8")
    got = out.cpu()
    out_o = torch.empty(4, dtype=torch.uint8)
    oracle.convert_fp8(out_o, x, 1.0, "fp8")
    # NaN/Inf bypass the clamp (amd_hip_fp8.h:561-575) and become e4m3 NaN (no inf in e4m3fn)
    assert ((got & 0x7f) == 0x7f).all(), got
    assert torch.equal(got & 0x7f, out_o & 0x7f)


@pytest.mark.parametrize("tag", ["bf16", "fp16", "fp32"])
def test_reshape_and_cache_reference_executed_golden(ops, tag):
    """HIP reshape_and_cache / reshape_and_cache_flash straight against the caches the reference's own CPU
    kernel wrote (csrc/cpu/cache.cpp compiled as oracle/_ref; tests/golden/cache_ref_*.pt): exact
    (test_cache.py:212-213)."""
    import os
    d = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"cache_ref_{tag}.pt"),
                   weights_only=True)
    key, value, slots = d["key"].cuda(), d["value"].cuda(), d["slots"].cuda()
    kc, vc = torch.zeros_like(d["kc"]).cuda(), torch.zeros_like(d["vc"]).cuda()
    one = torch.ones(1, device="cuda")
    ops.reshape_and_cache(key, value, kc, vc, slots, "auto", one, one)
    assert torch.equal(kc.cpu(), d["kc"]) and torch.equal(vc.cpu(), d["vc"])
    NB, H, _, BS, _ = d["kc"].shape
    D = d["vc"].shape[2]
    kf = torch.zeros((NB, BS, H, D), dtype=kc.dtype, device="cuda"); vf = torch.zeros_like(kf)
    ops.reshape_and_cache_flash(key, value, kf, vf, slots, "auto", one, one)
    assert torch.equal(vf.permute(0, 2, 3, 1).cpu(), d["vc"])
    assert torch.equal(kf.permute(0, 2, 3, 1).cpu(), d["kc"].permute(0, 1, 2, 4, 3).reshape(NB, H, D, BS))


def test_copy_blocks(ops):
    NB, L = 32, 3
    g = torch.Generator().manual_seed(2)
    kcs = [torch.randn((NB, 8, 16, 16, 8), generator=g).to(torch.bfloat16) for _ in range(L)]
    vcs = [torch.randn((NB, 8, 128, 16), generator=g).to(torch.bfloat16) for _ in range(L)]
    mapping = torch.tensor([[0, 5], [0, 9], [3, 30], [7, 1]], dtype=torch.int64)
    dk, dv = [t.cuda() for t in kcs], [t.cuda() for t in vcs]
    ops.copy_blocks(dk, dv, mapping.cuda())
    for l in range(L):
        for s, d in mapping.tolist():
            kcs[l][d] = kcs[l][s]
            vcs[l][d] = vcs[l][s]
        assert torch.equal(dk[l].cpu(), kcs[l]) and torch.equal(dv[l].cpu(), vcs[l])


def test_swap_blocks(ops):
    src = torch.randn((8, 1024), device="cuda")
    dst = torch.zeros((8, 1024))
    ops.swap_blocks(src, dst, torch.tensor([[1, 2], [3, 0]], dtype=torch.int64))
    torch.cuda.synchronize()
    assert torch.equal(dst[2], src[1].cpu()) and torch.equal(dst[0], src[3].cpu())
    with pytest.raises(RuntimeError, match="block_mapping must be on CPU"):
        ops.swap_blocks(src, dst, torch.tensor([[1, 2]], dtype=torch.int64, device="cuda"))


HIDDEN = [1, 2, 3, 4, 16, 67, 768, 2048, 5120, 5137, 8192, 8193, 1024, 1025, 1026, 1031, 1032, 14336]
TOKENS = [1, 7, 83, 256]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hidden", HIDDEN)
@pytest.mark.parametrize("tokens", [1, 83])
@pytest.mark.parametrize("ub", [False, True])
def test_dynamic_per_token_fp8_quant(ops, oracle, dtype, hidden, tokens, ub):
    g = torch.Generator().manual_seed(hidden * 7 + tokens)
    x = (torch.rand((tokens, hidden), generator=g) * 1000 + 1e-6).to(dtype)
    x[0, 0] = 0 if hidden > 1 else x[0, 0]
    scale_ub = torch.tensor([x.float().abs().mean().item()], dtype=torch.float32) if ub else None
    q_o, s_o = oracle.scaled_fp8_quant(x, scale_ub=scale_ub, use_per_token_if_dynamic=True)
    q, s = ops.scaled_fp8_quant(x.cuda(), scale_ub=None if scale_ub is None else scale_ub.cuda(),
                                use_per_token_if_dynamic=True)
    assert q.dtype == torch.float8_e4m3fn and s.shape == (tokens, 1)
    assert torch.equal(s.cpu().view(torch.int32), s_o.view(torch.int32))
    assert torch.equal(q.cpu().view(torch.uint8), q_o)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hidden", [1, 17, 1024, 1031, 5137, 8193])
@pytest.mark.parametrize("tokens", TOKENS)
def test_dynamic_per_tensor_and_static_fp8_quant(ops, oracle, dtype, hidden, tokens):
    g = torch.Generator().manual_seed(hidden + tokens)
    x = (torch.randn((tokens, hidden), generator=g) * 30).to(dtype)
    q_o, s_o = oracle.scaled_fp8_quant(x)
    q, s = ops.scaled_fp8_quant(x.cuda())
    assert torch.equal(s.cpu().view(torch.int32), s_o.view(torch.int32))
    assert torch.equal(q.cpu().view(torch.uint8), q_o)
    st = torch.tensor([0.0123], dtype=torch.float32)
    q_o2, _ = oracle.scaled_fp8_quant(x, st)
    q2, _ = ops.scaled_fp8_quant(x.cuda(), st.cuda())
    assert torch.equal(q2.cpu().view(torch.uint8), q_o2)


def test_fp8_quant_padding_and_output_arg(ops, oracle):
    x = torch.randn((5, 64)).to(torch.bfloat16)
    q, s = ops.scaled_fp8_quant(x.cuda(), num_token_padding=17, use_per_token_if_dynamic=True)
    assert q.shape == (17, 64) and s.shape == (17, 1)
    q_o, s_o = oracle.scaled_fp8_quant(x, use_per_token_if_dynamic=True)
    assert torch.equal(q[:5].cpu().view(torch.uint8), q_o)
    out = torch.empty((5, 64), dtype=torch.float8_e4m3fn, device="cuda")
    q2, _ = ops.scaled_fp8_quant(x.cuda(), use_per_token_if_dynamic=True, output=out)
    assert q2.data_ptr() == out.data_ptr()


def test_fp8_quant_large_int64_index(ops):
    """int64 indexing regression of the reference (test_fp8_quant.py:98-117)."""
    x = torch.full((2 ** 31 // 8192 + 2, 8192), 1.0, dtype=torch.bfloat16, device="cuda")
    q, s = ops.scaled_fp8_quant(x, torch.tensor([1.0], device="cuda"))
    assert (q[-1].view(torch.uint8) == 0x38).all() and (q[0].view(torch.uint8) == 0x38).all()

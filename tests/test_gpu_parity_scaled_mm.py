"""GPU parity: fp8 w8a8 scaled GEMM vs the CPU oracle (fp64-accumulated, scales
then cast, bias in the output dtype: triton_scaled_mm.py:87-110).

Stated tolerance (tighter than the reference's rtol=atol=1e-1 of
test_triton_scaled_mm.py:110-114): products are exact and accumulation is fp32,
so the only differences are fp32 summation order (<= K * 2^-24 relative to
sum|a*b|) and one output rounding: rtol 1.6e-2 (2 bf16 ulp), atol 1e-2*|sa*sb| scale.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def make(M, N, K, seed, per_token, per_channel, out_dtype, bias):
    g = torch.Generator().manual_seed(seed)
    a = (torch.randn((M, K), generator=g)).clamp(-448, 448).to(torch.float8_e4m3fn)
    w = (torch.randn((N, K), generator=g)).clamp(-448, 448).to(torch.float8_e4m3fn)
    sa = (torch.rand((M, 1) if per_token else (1, 1), generator=g) + 0.5) * 0.02
    sb = (torch.rand((N, 1) if per_channel else (1, 1), generator=g) + 0.5) * 0.02
    b = (torch.randn(N, generator=g) * 0.5).to(out_dtype) if bias else None
    return a, w, sa.float(), sb.float(), b


def check(ops, oracle, M, N, K, per_token, per_channel, out_dtype, bias, seed=0):
    a, w, sa, sb, b = make(M, N, K, seed, per_token, per_channel, out_dtype, bias)
    ref = oracle.scaled_mm(a.view(torch.uint8), w.view(torch.uint8).t(), sa, sb, out_dtype, b)
    out = ops.cutlass_scaled_mm(a.cuda(), w.cuda().t(), sa.cuda(), sb.cuda(), out_dtype,
                                None if b is None else b.cuda())
    assert out.shape == (M, N) and out.dtype == out_dtype
    rtol = 1.6e-2 if out_dtype == torch.bfloat16 else 2e-3
    torch.testing.assert_close(out.cpu().float(), ref.float(), rtol=rtol, atol=5e-3)


@pytest.mark.parametrize("M", [1, 33, 64, 128, 257, 512])
@pytest.mark.parametrize("N", [256, 971, 4096])
@pytest.mark.parametrize("K", [128, 496, 1024])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float16])
def test_scaled_mm_shapes(ops, oracle, M, N, K, out_dtype):
    check(ops, oracle, M, N, K, True, True, out_dtype, bias=(N == 971), seed=M + N + K)


@pytest.mark.parametrize("per_token", [False, True])
@pytest.mark.parametrize("per_channel", [False, True])
@pytest.mark.parametrize("bias", [False, True])
def test_scaled_mm_scale_kinds(ops, oracle, per_token, per_channel, bias):
    check(ops, oracle, 96, 384, 512, per_token, per_channel, torch.bfloat16, bias, seed=3)


def test_scaled_mm_integer_exact(ops, oracle):
    """Asymmetric small-integer operands: every product and sum is exact, so the
    result must match to the bit (catches any operand-layout / transpose error)."""
    M, N, K = 128, 256, 256
    g = torch.Generator().manual_seed(1)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-2, 3, (N, K), generator=g).float()
    a[:, 0] = torch.arange(M).float() % 7
    w[:, 1] = torch.arange(N).float() % 5
    a8, w8 = a.to(torch.float8_e4m3fn), w.to(torch.float8_e4m3fn)
    one = torch.ones((1, 1))
    out = ops.cutlass_scaled_mm(a8.cuda(), w8.cuda().t(), one.cuda(), one.cuda(), torch.float16)
    ref = (a @ w.t())
    assert torch.equal(out.cpu().float(), ref.to(torch.float16).float())


@pytest.mark.parametrize("KN", [(4096, 6144), (14336, 4096)])
@pytest.mark.parametrize("M", [1, 16, 256])
def test_scaled_mm_llama_shapes(ops, oracle, KN, M):
    K, N = KN
    check(ops, oracle, M, N, K, True, True, torch.bfloat16, False, seed=K + M)


@pytest.mark.parametrize("K", [1024, 3584])
def test_scaled_mm_configs3_row_parallel_shapes(ops, oracle, K):
    """configs[3]: the per-GPU row-parallel GEMMs of the TP=8 Llama-3-70B shard (o_proj K 1024,
    down_proj K 3584 -> hidden 8192, M 256; weight_shapes.py:58-63 / 8), per-token x per-channel."""
    check(ops, oracle, 256, 8192, K, True, True, torch.bfloat16, False, seed=K)


@pytest.mark.parametrize("M", [1, 16])
def test_scaled_mm_gate_up_shape_small_m(ops, oracle, M):
    """configs[2]: (K 4096, N 28672) at decode batch sizes."""
    check(ops, oracle, M, 28672, 4096, True, True, torch.bfloat16, False, seed=M)


@pytest.mark.parametrize("n", [1, 2, 3, 4])
@pytest.mark.parametrize("mk", [(256, 1024), (4096, 4096), (1000, 512)])
def test_wvSplitKQ(ops, oracle, n, mk):
    """vs the oracle's scaled_mm restatement with per-tensor scales
    (reference test: tests/kernels/quantization/test_rocm_skinny_gemms.py:57-82, rtol 1e-2)."""
    m, k = mk
    g = torch.Generator().manual_seed(n + m)
    A = (torch.rand((n, k), generator=g) - 0.5).to(torch.float8_e4m3fn)   # activations
    B = (torch.rand((m, k), generator=g) - 0.5).to(torch.float8_e4m3fn)   # weights
    sa, sb = torch.tensor([0.7]), torch.tensor([1.3])
    out = ops.wvSplitKQ(B.cuda(), A.cuda(), torch.bfloat16, sb.cuda(), sa.cuda(), 256)
    assert out.shape == (n, m)
    ref = oracle.scaled_mm(A.view(torch.uint8), B.view(torch.uint8).t(), sa, sb, torch.bfloat16)
    torch.testing.assert_close(out.cpu().float(), ref.float(), rtol=1e-2, atol=1e-3)


def test_torch_scaled_mm_pertensor_golden(ops):
    """HIP wvSplitKQ and per-tensor cutlass_scaled_mm straight against outputs of CPU torch._scaled_mm
    (tests/golden/scaled_mm_torch_pertensor.pt): the reference's own comparison for wvSplitKQ
    (test_rocm_skinny_gemms.py:65-82), at its rtol 1e-2."""
    import os
    d = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                "scaled_mm_torch_pertensor.pt"), weights_only=True)
    for i in range(int(d["n"])):
        a = d[f"a{i}"].view(torch.float8_e4m3fn).cuda()      # activations [n, K]
        w = d[f"w{i}"].view(torch.float8_e4m3fn).cuda()      # weights [m, K]
        sa, sb, want = d[f"sa{i}"].cuda(), d[f"sb{i}"].cuda(), d[f"out{i}"]
        out = ops.cutlass_scaled_mm(a, w.t(), sa.reshape(1, 1), sb.reshape(1, 1), want.dtype)
        torch.testing.assert_close(out.cpu().float(), want.float(), rtol=1e-2, atol=1e-3)
        if a.shape[0] <= 4:
            out = ops.wvSplitKQ(w, a, want.dtype, sb, sa, 256)
            torch.testing.assert_close(out.cpu().float(), want.float(), rtol=1e-2, atol=1e-3)


@pytest.mark.parametrize("KN", [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)])
@pytest.mark.parametrize("M", [1024, 4096])
def test_scaled_mm_configs2_large_m_sampled_oracle(ops, oracle, KN, M):
    """configs[2] at large M on RANDOM data (per-token x per-channel scales, random mantissas over the full
    K): the oracle on a sample of 64 rows x 64 columns of the output, same tolerance as the small shapes.
    Twice: repeated launches of the persistent kernel agree to the bit."""
    K, N = KN
    g = torch.Generator(device="cuda").manual_seed(K + N + M)
    a = torch.randn((M, K), generator=g, device="cuda").to(torch.float8_e4m3fn)
    w = torch.randn((N, K), generator=g, device="cuda").to(torch.float8_e4m3fn)
    sa = (torch.rand((M, 1), generator=g, device="cuda") + 0.5) * 0.02
    sb = (torch.rand((N, 1), generator=g, device="cuda") + 0.5) * 0.02
    out = ops.cutlass_scaled_mm(a, w.t(), sa, sb, torch.bfloat16)
    again = ops.cutlass_scaled_mm(a, w.t(), sa, sb, torch.bfloat16)
    assert torch.equal(out, again)
    cg = torch.Generator().manual_seed(M + N)
    rows = torch.cat([torch.tensor([0, 255, 256, M - 1]), torch.randint(0, M, (60,), generator=cg)])
    cols = torch.cat([torch.tensor([0, 255, 256, N - 1]), torch.randint(0, N, (60,), generator=cg)])
    ref = oracle.scaled_mm(a[rows.cuda()].cpu().view(torch.uint8), w[cols.cuda()].cpu().view(torch.uint8).t(),
                           sa[rows.cuda()].cpu(), sb[cols.cuda()].cpu(), torch.bfloat16)
    got = out[rows.cuda()][:, cols.cuda()].cpu()
    torch.testing.assert_close(got.float(), ref.float(), rtol=1.6e-2, atol=5e-3)


@pytest.mark.parametrize("MNK", [(4096, 4096, 1024), (1000, 1288, 1024), (2304, 2816, 512), (4096, 10240, 384)])
def test_scaled_mm_pp_variants_identical(ops, MNK):
    """The staggered / persistent forms of the 256 x 256 kernel (scaled_mm_pp.hip, "mm_kernel" 18-25; 29-36: the same
    with 256 x 192 tiles) keep
    the K order and the MFMA of the kernel of scaled_mm_large.hip ("mm_kernel" 2): bit-identical outputs on
    random operands, with bias, ragged edges, one and several tiles per workgroup."""
    from vllm_amd import _lib
    M, N, K = MNK
    a, w, sa, sb, b = make(M, N, K, M + N, True, True, torch.bfloat16, True)
    a, w, sa, sb, b = a.cuda(), w.cuda(), sa.cuda(), sb.cuda(), b.cuda()
    try:
        outs = {}
        for k in (2, 18, 19, 20, 21, 23, 25, 29, 30, 31, 32, 34, 36, 0):
            _lib.set_tuning("mm_kernel", k)
            outs[k] = ops.cutlass_scaled_mm(a, w.t(), sa, sb, torch.bfloat16, b)
    finally:
        _lib.set_tuning("mm_kernel", 0)
    for k, o in outs.items():
        assert torch.equal(o, outs[2]), k


@pytest.fixture
def force_kernel():
    """Pin the GEMM kernel choice (vllm_amd_set_tuning "mm_kernel"); back to auto afterwards."""
    from vllm_amd import _lib

    def force(k, splits=0):
        _lib.set_tuning("mm_kernel", k)
        _lib.set_tuning("mm_splitk", splits)
    yield force
    _lib.set_tuning("mm_kernel", 0)
    _lib.set_tuning("mm_splitk", 0)


@pytest.mark.parametrize("kernel", list(range(1, 37)))
@pytest.mark.parametrize("MNK", [(300, 520, 256), (256, 256, 1024), (257, 971, 384), (640, 1288, 2048),
                                 (40, 8, 128 * 3), (5, 264, 512), (32, 1040, 1024)])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float16])
def test_scaled_mm_every_tile_kernel(ops, oracle, force_kernel, kernel, MNK, out_dtype):
    """The auto dispatch picks a tile shape from the problem size; force each one on ragged small shapes (edge tiles, odd N -> scalar epilogue,
    odd number of K blocks, bias) so the oracle still finishes in seconds."""
    M, N, K = MNK
    force_kernel(kernel)
    check(ops, oracle, M, N, K, True, True, out_dtype, bias=(N % 2 == 0), seed=M + N + K + kernel)
    check(ops, oracle, M, N, K, False, False, out_dtype, bias=False, seed=M + N + kernel)


@pytest.mark.parametrize("kernel", list(range(1, 11)) + list(range(12, 29)))
def test_scaled_mm_integer_exact_every_kernel(ops, force_kernel, kernel):
    M, N, K = 512, 768, 640
    g = torch.Generator().manual_seed(kernel)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-2, 3, (N, K), generator=g).float()
    a[:, 0] = torch.arange(M).float() % 7
    w[:, 1] = torch.arange(N).float() % 5
    one = torch.ones((1, 1))
    force_kernel(kernel)
    out = ops.cutlass_scaled_mm(a.to(torch.float8_e4m3fn).cuda(), w.to(torch.float8_e4m3fn).cuda().t(),
                                one.cuda(), one.cuda(), torch.float16)
    assert torch.equal(out.cpu().float(), (a @ w.t()).to(torch.float16).float())


def test_scaled_mm_large_auto(ops):
    """Full-size shape on the auto-dispatched large kernel, checked through a size-independent
    property instead of the oracle: with power-of-two operands every product and partial
    sum is exact, so out == K * sa * sb * pattern regardless of summation order."""
    M, N, K = 4096, 4096, 4096
    a = torch.full((M, K), 0.5).to(torch.float8_e4m3fn).cuda()
    w = torch.full((N, K), 2.0).to(torch.float8_e4m3fn).cuda()
    sa = (2.0 ** -(torch.arange(M) % 5).float()).reshape(M, 1).cuda()
    sb = (2.0 ** -(torch.arange(N) % 3).float()).reshape(N, 1).cuda()
    out = ops.cutlass_scaled_mm(a, w.t(), sa, sb, torch.bfloat16)
    ref = (K * sa * sb.t()).to(torch.bfloat16)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("MN", [(4096, 6144), (4000, 6144), (3584, 6144), (3500, 6144), (2048, 10240), (4096, 4096)])
@pytest.mark.parametrize("per_channel", [True, False])
def test_scaled_mm_several_tiles_per_workgroup(ops, MN, per_channel):
    """Shapes with more 256 x 256 tiles than CUs: a workgroup of the persistent kernel (scaled_mm_pp.hip)
    walks several tiles, unevenly (384 tiles = 1.5 per workgroup), with ragged M: small-integer operands
    and power-of-two scales make every sum exact, so the result equals the fp32 matmul bit for bit, with
    per-channel scales and bias addressed per tile."""
    M, N = MN
    K = 384
    g = torch.Generator().manual_seed(M + N)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-2, 3, (N, K), generator=g).float()
    w[:, 0] = torch.arange(N).float() % 7                    # column identity
    sa = (2.0 ** -(torch.arange(M) % 3).float()).reshape(M, 1)
    sb = (2.0 ** -(torch.arange(N) % 4).float()).reshape(N, 1) if per_channel else torch.full((1, 1), 0.5)
    bias = (torch.arange(N) % 9).to(torch.bfloat16)
    out = ops.cutlass_scaled_mm(a.to(torch.float8_e4m3fn).cuda(), w.to(torch.float8_e4m3fn).cuda().t(),
                                sa.cuda(), sb.cuda(), torch.bfloat16, bias.cuda())
    ref = ((a.cuda() @ w.cuda().t()) * sa.cuda() * (sb.cuda().t() if per_channel else 0.5)).to(torch.bfloat16)
    ref = (ref.float() + bias.cuda().float()).to(torch.bfloat16)
    assert torch.equal(out, ref)


SPLIT_KERNELS = [3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 26, 27, 28]


@pytest.mark.parametrize("kernel", SPLIT_KERNELS)
@pytest.mark.parametrize("splits", [2, 3, 8])
@pytest.mark.parametrize("MNK", [(300, 520, 1024), (33, 264, 128 * 7), (70, 1288, 2048), (16, 4096, 4096)])
def test_scaled_mm_split_k(ops, oracle, force_kernel, kernel, splits, MNK):
    """Split-K (K slices reduced in-kernel through the fp32 workspace): every tile shape,
    slice counts that do and do not divide the number of K blocks, ragged edge tiles."""
    M, N, K = MNK
    force_kernel(kernel, splits)
    check(ops, oracle, M, N, K, True, True, torch.bfloat16, bias=(kernel % 2 == 0), seed=M + kernel + splits)
    check(ops, oracle, M, N, K, False, True, torch.float16, bias=False, seed=N + kernel + splits)


@pytest.mark.parametrize("kernel", SPLIT_KERNELS)
def test_scaled_mm_split_k_integer_exact_and_repeatable(ops, force_kernel, kernel):
    """Integer operands: bit-exact for any slice count; random operands: the slices are
    summed in slice order whoever arrives last, so repeated launches (which also reuse the
    self-resetting ticket counters) agree to the bit."""
    M, N, K = 200, 776, 2560
    g = torch.Generator().manual_seed(kernel)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-2, 3, (N, K), generator=g).float()
    one = torch.ones((1, 1)).cuda()
    a8, w8 = a.to(torch.float8_e4m3fn).cuda(), w.to(torch.float8_e4m3fn).cuda()
    ref = (a @ w.t()).to(torch.float16).float()
    for splits in (2, 4, 5, 20):
        force_kernel(kernel, splits)
        out = ops.cutlass_scaled_mm(a8, w8.t(), one, one, torch.float16)
        assert torch.equal(out.cpu().float(), ref), splits
    ar, wr, sa, sb, _ = make(M, N, K, 7, True, True, torch.bfloat16, False)
    ar, wr, sa, sb = ar.cuda(), wr.cuda(), sa.cuda(), sb.cuda()
    force_kernel(kernel, 4)
    outs = [ops.cutlass_scaled_mm(ar, wr.t(), sa, sb, torch.bfloat16) for _ in range(6)]
    torch.cuda.synchronize()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


def test_scaled_mm_split_k_two_streams(ops, force_kernel):
    """Launches on different streams use different workspaces."""
    M, N, K = 64, 2048, 4096
    ar, wr, sa, sb, _ = make(M, N, K, 11, True, True, torch.bfloat16, False)
    ar, wr, sa, sb = ar.cuda(), wr.cuda(), sa.cuda(), sb.cuda()
    force_kernel(9, 1)
    ref = ops.cutlass_scaled_mm(ar, wr.t(), sa, sb, torch.bfloat16)
    force_kernel(9, 4)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    res = []
    for _ in range(8):
        for st in (s1, s2):
            with torch.cuda.stream(st):
                res.append(ops.cutlass_scaled_mm(ar, wr.t(), sa, sb, torch.bfloat16))
    torch.cuda.synchronize()
    for o in res:
        torch.testing.assert_close(o.float(), ref.float(), rtol=1.6e-2, atol=5e-3)
        assert torch.equal(o, res[0])


def test_scaled_mm_split_k_more_streams_than_slots(ops, force_kernel):
    """The registered workspace has 4 stream slots: launches from 6 streams at once stay correct
    (the fifth and sixth stream run unsplit) and the slot owners agree to the bit."""
    M, N, K = 64, 2048, 4096
    ar, wr, sa, sb, _ = make(M, N, K, 12, True, True, torch.bfloat16, False)
    ar, wr, sa, sb = ar.cuda(), wr.cuda(), sa.cuda(), sb.cuda()
    force_kernel(9, 1)
    ref = ops.cutlass_scaled_mm(ar, wr.t(), sa, sb, torch.bfloat16)
    force_kernel(9, 4)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(6)]
    res = []
    for _ in range(6):
        for st in streams:
            with torch.cuda.stream(st):
                res.append(ops.cutlass_scaled_mm(ar, wr.t(), sa, sb, torch.bfloat16))
    torch.cuda.synchronize()
    for o in res:
        torch.testing.assert_close(o.float(), ref.float(), rtol=1.6e-2, atol=5e-3)


def test_scaled_mm_split_k_graph_capture(ops, force_kernel):
    """A launch under stream capture never takes a workspace slot (it runs unsplit), so a
    replayed graph cannot meet eager split launches in the workspace; nothing is allocated or
    synchronised inside the captured call."""
    M, N, K = 48, 1024, 14336
    ar, wr, sa, sb, _ = make(M, N, K, 13, True, True, torch.bfloat16, False)
    ar, wr, sa, sb = ar.cuda(), wr.cuda(), sa.cuda(), sb.cuda()
    force_kernel(9, 1)
    ref = ops.cutlass_scaled_mm(ar, wr.t(), sa, sb, torch.bfloat16)     # also registers the workspace
    force_kernel(9, 2)
    eager = ops.cutlass_scaled_mm(ar, wr.t(), sa, sb, torch.bfloat16)   # split
    torch.cuda.synchronize()
    out = torch.empty_like(ref)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            torch.ops._C.cutlass_scaled_mm(out, ar, wr.t(), sa, sb, None)
        for _ in range(3):
            g.replay()
            again = ops.cutlass_scaled_mm(ar, wr.t(), sa, sb, torch.bfloat16)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)          # captured = unsplit
    assert torch.equal(again, eager)
    torch.testing.assert_close(eager.float(), ref.float(), rtol=1.6e-2, atol=5e-3)

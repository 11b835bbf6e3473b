"""GPU test of vllm_amd_scaled_mm_static, the one entry point that allocates nothing (no
counterpart in the reference): split-K partials and tickets live in the library's static
__device__ array.  Called straight through the C ABI and compared with the existing entry point
vllm_amd_scaled_mm on two calls in a row, the second with a smaller input.  Bar: the GEMM's
stated tolerance against the unsplit existing entry (rtol 1.6e-2 / atol 5e-3, bf16), and
bit-exact against the existing entry's own split-K path at the same slice count (both sum the
slices in slice order)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _static_fn():
    from vllm_amd._lib import lib
    return lib.vllm_amd_scaled_mm_static


def _make(M, N, K, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn((M, K), generator=g).clamp(-448, 448).to(torch.float8_e4m3fn).cuda()
    w = torch.randn((N, K), generator=g).clamp(-448, 448).to(torch.float8_e4m3fn).cuda()
    sa = ((torch.rand((M, 1), generator=g) + 0.5) * 0.02).float().cuda()
    sb = ((torch.rand((N, 1), generator=g) + 0.5) * 0.02).float().cuda()
    return a, w, sa, sb


def _call_static(a, w, sa, sb, splits):
    from vllm_amd import _lib
    M, K = a.shape
    N = w.shape[0]
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    rc = _static_fn()(out.data_ptr(), a.data_ptr(), w.data_ptr(), sa.data_ptr(), sb.data_ptr(), None,
                      M, N, K, K, K, N, M, N, _lib.BF16, splits, st)
    if rc != 0:
        raise RuntimeError(_lib.lib.vllm_amd_last_error().decode())
    return out


@pytest.mark.parametrize("splits", [0, 2, 4])
def test_scaled_mm_static_matches_existing_entry(ops, splits):
    from vllm_amd import _lib
    # first call: the shape class the heuristic splits (M <= 64, K >= 12288); second: smaller
    for M, N, K in ((48, 2048, 14336), (17, 520, 12288)):
        a, w, sa, sb = _make(M, N, K, M + N)
        try:
            _lib.set_tuning("mm_splitk", 1)
            unsplit = ops.cutlass_scaled_mm(a, w.t(), sa, sb, torch.bfloat16)
            _lib.set_tuning("mm_splitk", splits)     # 0 = the same heuristic the static entry applies
            same = ops.cutlass_scaled_mm(a, w.t(), sa, sb, torch.bfloat16)
        finally:
            _lib.set_tuning("mm_splitk", 0)
        out = _call_static(a, w, sa, sb, splits)
        torch.cuda.synchronize()
        assert not out.isnan().any()
        torch.testing.assert_close(out.float(), unsplit.float(), rtol=1.6e-2, atol=5e-3)
        assert torch.equal(out, same)


def test_scaled_mm_static_dirty_scratch_and_limit(ops):
    """The array keeps the last call's contents (tickets are cleared per call): a big split right
    before a small one must not leak; a split beyond the stated maximum is an error, not a
    fallback."""
    a, w, sa, sb = _make(64, 4096, 4096, 1)
    ref = ops.cutlass_scaled_mm(a, w.t(), sa, sb, torch.bfloat16)
    o1 = _call_static(a, w, sa, sb, 8)
    a2, w2, sa2, sb2 = _make(8, 264, 1024, 2)
    ref2 = ops.cutlass_scaled_mm(a2, w2.t(), sa2, sb2, torch.bfloat16)
    o2 = _call_static(a2, w2, sa2, sb2, 3)
    torch.cuda.synchronize()
    torch.testing.assert_close(o1.float(), ref.float(), rtol=1.6e-2, atol=5e-3)
    torch.testing.assert_close(o2.float(), ref2.float(), rtol=1.6e-2, atol=5e-3)
    a3, w3, sa3, sb3 = _make(512, 8192, 1024, 3)    # 4 x 128 tiles of 128 x 64, 8 slices: 128 MiB of partials
    from vllm_amd import _lib
    try:
        _lib.set_tuning("mm_kernel", 5)
        with pytest.raises(RuntimeError, match="static scratch holds"):
            _call_static(a3, w3, sa3, sb3, 8)
    finally:
        _lib.set_tuning("mm_kernel", 0)

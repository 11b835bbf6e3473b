#!/usr/bin/env python3
"""Generates tests/golden/*.pt (inputs + expected outputs).  Run in the authoring
container (needs oracle/_ref/librefcpu.so built from /root/reference by oracle/Makefile):

    python tests/golden/make_golden.py

Sources of the expected outputs
  attn_ref_*.pt     the reference's OWN CPU kernels (csrc/cpu/attention.cpp via oracle/_ref):
                    paged_attention_v1 and _v2, bf16 / fp16 / fp32, block 16, with and without alibi
  attnfp8_ref_*.pt  fp8-e4m3 KV (the headline workload): the cache is dequantised element by
                    element to the query dtype, scalar_t(float(fp8) * scale) -- what
                    quant_utils.cuh:338-345, 436-457 does inside the GPU kernel and what the
                    reference's own test does with convert_fp8 (test_attention.py:306-322) -- and
                    the reference's CPU kernels run on the dequantised 16-bit cache.  Stored:
                    fp8 codes, k_scale != v_scale, and the reference's v1 / v2 outputs.
  cache_ref_*.pt    the reference's own CPU reshape_and_cache (csrc/cpu/cache.cpp via oracle/_ref)
  fp8_quant_*.pt    torch CPU restatement of the formulas in the reference's test helpers
                    (tests/kernels/quant_utils.py:22-97: ref_dynamic_per_token_quant,
                    ref_dynamic_per_tensor_fp8_quant) using torch.float8_e4m3fn casts
  scaled_mm_*.pt    torch CPU restatement of baseline_scaled_mm (tests/kernels/utils.py:1212-1251)
  scaled_mm_torch_pertensor.pt  outputs of CPU torch._scaled_mm (scalar scales): the reference of the
                    reference's own wvSplitKQ test (test_rocm_skinny_gemms.py:65-82)
  fp8_codec.pt      all 256 e4m3fn codes decoded by torch + an encode sweep by torch casts
  fused_quant_*.pt  torch CPU restatement of the "native" side of the reference's tests:
                    RMSNorm.forward_native (vllm/model_executor/layers/layernorm.py) followed by
                    ops.scaled_fp8_quant (tests/kernels/core/test_fused_quant_layernorm.py:40-74),
                    SiluAndMul.forward_native + static quant
                    (tests/kernels/test_fused_quant_activation.py:21-25)
  slot_mapping.pt   numpy arithmetic of BlockTable.compute_slot_mapping (block_table.py:92-99)
The fixtures are DATA only; nothing of the reference's source text is stored.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
from helpers import make_attn_case  # noqa: E402

FP8_MAX = 448.0


def save(name, d):
    torch.save(d, os.path.join(HERE, name))
    print(name, sum(v.numel() * v.element_size() for v in d.values() if torch.is_tensor(v)) // 1024, "KiB")


def attention():
    assert O.load_ref(), "build oracle/_ref first (make -C oracle ref)"
    one = torch.ones(1)
    for dtype, tag in ((torch.bfloat16, "bf16"), (torch.float16, "fp16"), (torch.float32, "fp32")):
        for H, KVH in (((4, 2), (4, 1)) if tag != "fp32" else ((4, 1),)):
            lens = [1, 16, 17, 130, 513]
            c = make_attn_case(len(lens), H, KVH, 128, 16, dtype, "auto", lens, seed=H + KVH,
                               alibi=(KVH == 1), num_blocks=48)
            out1 = torch.empty_like(c["q"])
            torch.ops.refcpu.paged_attention_v1(out1, c["q"], c["kc"], c["vc"], KVH, c["scale"],
                                                c["bt"], c["sl"], 16, c["max_len"], c["slopes"],
                                                "auto", one, one, 0, 0, 0, 64, 0)
            P = (c["max_len"] + 511) // 512
            S, _, D = c["q"].shape
            es = torch.zeros((S, H, P)); ml = torch.zeros_like(es)
            tmp = torch.zeros((S, H, P, D), dtype=dtype)
            out2 = torch.empty_like(c["q"])
            torch.ops.refcpu.paged_attention_v2(out2, es, ml, tmp, c["q"], c["kc"], c["vc"], KVH,
                                                c["scale"], c["bt"], c["sl"], 16, c["max_len"],
                                                c["slopes"], "auto", one, one, 0, 0, 0, 64, 0)
            d = dict(q=c["q"], kc=c["kc"], vc=c["vc"], bt=c["bt"], sl=c["sl"], out_v1=out1,
                     out_v2=out2, scale=torch.tensor(c["scale"]), num_kv_heads=torch.tensor(KVH))
            if c["slopes"] is not None:
                d["slopes"] = c["slopes"]
            save(f"attn_ref_{tag}_h{H}kv{KVH}.pt", d)


def _dequant_cache(kc8, vc8, dtype, k_scale, v_scale):
    """fp8 paged cache (x = 16) -> 16-bit paged cache (x = 8) holding scalar_t(float(fp8) * scale)."""
    NB, KVH, DX, BS, X = kc8.shape
    D = DX * X
    k = (kc8.view(torch.float8_e4m3fn).float() * torch.tensor(k_scale, dtype=torch.float32)).to(dtype)
    v = (vc8.view(torch.float8_e4m3fn).float() * torch.tensor(v_scale, dtype=torch.float32)).to(dtype)
    k = k.permute(0, 1, 2, 4, 3).reshape(NB, KVH, D, BS)            # [NB, KVH, d, tok]
    x = 16 // k.element_size()
    k = k.reshape(NB, KVH, D // x, x, BS).permute(0, 1, 2, 4, 3).contiguous()
    return k, v.contiguous()


def attention_fp8kv():
    assert O.load_ref(), "build oracle/_ref first (make -C oracle ref)"
    one = torch.ones(1)
    cases = (("bf16", torch.bfloat16, 16, 4, [1, 15, 16, 17, 300, 513, 1029], 0.037, 0.021, False),
             ("fp16", torch.float16, 8, 1, [1, 64, 65, 257, 770], 0.05, 0.11, True),
             ("bf16", torch.bfloat16, 16, 2, [5, 511, 512, 1100], 1.0, 0.5, False))
    for tag, dtype, H, KVH, lens, ks, vs, alibi in cases:
        # codes of U(-d^-.5, d^-.5) / scale: spread over the e4m3 range like a calibrated cache
        c = make_attn_case(len(lens), H, KVH, 128, 16, dtype, "fp8", lens, seed=H + 3 * KVH,
                           kv_scale=ks, alibi=alibi)
        vc8 = make_attn_case(len(lens), H, KVH, 128, 16, dtype, "fp8", lens, seed=H + 3 * KVH,
                             kv_scale=vs)["vc"]
        kd, vd = _dequant_cache(c["kc"], vc8, dtype, ks, vs)
        out1 = torch.empty_like(c["q"])
        torch.ops.refcpu.paged_attention_v1(out1, c["q"], kd, vd, KVH, c["scale"], c["bt"], c["sl"],
                                            16, c["max_len"], c["slopes"], "auto", one, one,
                                            0, 0, 0, 64, 0)
        P = (c["max_len"] + 511) // 512
        S, _, D = c["q"].shape
        es = torch.zeros((S, H, P)); ml = torch.zeros_like(es)
        tmp = torch.zeros((S, H, P, D), dtype=dtype)
        out2 = torch.empty_like(c["q"])
        torch.ops.refcpu.paged_attention_v2(out2, es, ml, tmp, c["q"], kd, vd, KVH, c["scale"],
                                            c["bt"], c["sl"], 16, c["max_len"], c["slopes"], "auto",
                                            one, one, 0, 0, 0, 64, 0)
        d = dict(q=c["q"], kc=c["kc"], vc=vc8, bt=c["bt"], sl=c["sl"], out_v1=out1, out_v2=out2,
                 scale=torch.tensor(c["scale"]), num_kv_heads=torch.tensor(KVH),
                 k_scale=torch.tensor(ks), v_scale=torch.tensor(vs))
        if c["slopes"] is not None:
            d["slopes"] = c["slopes"]
        save(f"attnfp8_ref_{tag}_h{H}kv{KVH}.pt", d)


def cache():
    assert O.load_ref()
    one = torch.ones(1)
    for dtype, tag in ((torch.bfloat16, "bf16"), (torch.float16, "fp16"), (torch.float32, "fp32")):
        g = torch.Generator().manual_seed(11)
        T, H, D, BS, NB = 23, 4, 64, 16, 12
        x = 16 // torch.tensor([], dtype=dtype).element_size()
        key = torch.randn((T, H, D), generator=g).to(dtype)
        value = torch.randn((T, H, D), generator=g).to(dtype)
        slots = torch.randperm(NB * BS, generator=g)[:T].to(torch.int64)
        kc = torch.zeros((NB, H, D // x, BS, x), dtype=dtype)
        vc = torch.zeros((NB, H, D, BS), dtype=dtype)
        torch.ops.refcpu.reshape_and_cache(key, value, kc, vc, slots, "auto", one, one)
        save(f"cache_ref_{tag}.pt", dict(key=key, value=value, slots=slots, kc=kc, vc=vc))


def fp8_quant():
    g = torch.Generator().manual_seed(5)
    min_scale = 1.0 / (FP8_MAX * 512.0)
    for dtype, tag in ((torch.bfloat16, "bf16"), (torch.float16, "fp16"), (torch.float32, "fp32")):
        x = (torch.rand((9, 1031), generator=g) * 1000 + 1e-6).to(dtype)
        xf = x.float()
        # per token (quant_utils.py:22-61): scale = max(amax / 448, 1/(448*512)); q = sat(x / scale)
        amax = xf.abs().amax(dim=-1, keepdim=True)
        ub = torch.tensor([amax.mean().item()])
        out = {}
        for name, am in (("pt", amax), ("ptub", amax.clamp(max=ub))):
            s = (am / FP8_MAX).clamp(min=min_scale)
            q = (xf / s).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).view(torch.uint8)
            out[name + "_q"] = q
            out[name + "_s"] = s
        # per tensor (quant_utils.py:66-97): scale = amax / 448 ; q = sat(x * (1 / scale))
        s = (xf.abs().max() / FP8_MAX).reshape(1)
        q = (xf * (1.0 / s)).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).view(torch.uint8)
        out.update(x=x, ub=ub, tensor_q=q, tensor_s=s)
        save(f"fp8_quant_{tag}.pt", out)


def scaled_mm():
    g = torch.Generator().manual_seed(9)
    M, N, K = 33, 200, 496
    a = torch.randn((M, K), generator=g).to(torch.float8_e4m3fn)
    w = torch.randn((N, K), generator=g).to(torch.float8_e4m3fn)
    sa = torch.rand((M, 1), generator=g) * 0.05 + 0.01
    sb = torch.rand((N, 1), generator=g) * 0.05 + 0.01
    bias = torch.randn(N, generator=g).to(torch.bfloat16)
    # baseline_scaled_mm: (scale_a * a.f32) @ (scale_b * b.f32) -> out dtype (+ bias)
    ref = torch.mm(sa * a.float(), (sb * w.float()).t())
    save("scaled_mm_bf16.pt", dict(a=a.view(torch.uint8), w=w.view(torch.uint8), sa=sa, sb=sb,
                                   bias=bias, out=ref.to(torch.bfloat16),
                                   out_bias=(ref + bias.float()).to(torch.bfloat16)))


def scaled_mm_torch():
    """Per-tensor scales: outputs of CPU torch._scaled_mm itself -- the reference its own skinny-GEMM test
    compares wvSplitKQ with (tests/kernels/quantization/test_rocm_skinny_gemms.py:65-82, rtol 1e-2).  Cases
    0-3: the wvSplitKQ call shapes (n <= 4 activation rows); 4-5: per-tensor cutlass_scaled_mm shapes."""
    g = torch.Generator().manual_seed(21)
    d = {}
    cases = [(1, 256, 1024, torch.bfloat16), (2, 1000, 512, torch.bfloat16), (3, 256, 1024, torch.float16),
             (4, 2048, 256, torch.bfloat16), (33, 208, 496, torch.bfloat16), (128, 256, 640, torch.float16)]
    for i, (M, N, K, dt) in enumerate(cases):
        a = (torch.rand((M, K), generator=g) - 0.5).to(torch.float8_e4m3fn)
        w = (torch.rand((N, K), generator=g) - 0.5).to(torch.float8_e4m3fn)
        sa, sb = torch.rand(1, generator=g) + 0.5, torch.rand(1, generator=g) + 0.5
        out = torch._scaled_mm(a, w.t(), scale_a=sa.reshape(()), scale_b=sb.reshape(()), out_dtype=dt)
        d.update({f"a{i}": a.view(torch.uint8), f"w{i}": w.view(torch.uint8), f"sa{i}": sa, f"sb{i}": sb,
                  f"out{i}": out})
    d["n"] = torch.tensor(len(cases))
    save("scaled_mm_torch_pertensor.pt", d)


def fp8_codec():
    codes = torch.arange(256, dtype=torch.uint8)
    dec = codes.view(torch.float8_e4m3fn).float()
    fin = dec[~dec.isnan()].sort().values
    mids = (fin[:-1] + fin[1:]) / 2
    g = torch.Generator().manual_seed(1)
    x = torch.cat([fin, mids, mids * (1 + 2 ** -12), mids * (1 - 2 ** -12),
                   torch.randn(4096, generator=g) * 50, torch.randn(4096, generator=g) * 0.01,
                   torch.tensor([449.0, 1e6, -1e6, 464.0, 465.0, -464.0])])
    enc = x.clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).view(torch.uint8)
    save("fp8_codec.pt", dict(dec=dec, x=x, enc=enc))


def _rms_native(x, weight, eps, residual=None):
    """RMSNorm.forward_native: fp32 maths, one rounding to the input dtype before the weight."""
    dt = x.dtype
    xf = x.float()
    new_res = None
    if residual is not None:
        xf = xf + residual.float()
        new_res = xf.to(dt)
    var = xf.pow(2).mean(dim=-1, keepdim=True)
    y = (xf * torch.rsqrt(var + eps)).to(dt) * weight
    return y, new_res


def _q_static(y, scale):
    # static_scaled_fp8_quant (common.cu:21-25): the reciprocal is taken in fp32
    inv = 1.0 / torch.tensor(scale, dtype=torch.float32)
    return (y.float() * inv).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).view(torch.uint8)


def fused_quant():
    g = torch.Generator().manual_seed(21)
    min_scale = 1.0 / (FP8_MAX * 512.0)
    eps = 1e-6
    for dtype, tag in ((torch.bfloat16, "bf16"), (torch.float16, "fp16"), (torch.float32, "fp32")):
        T, H = 7, 1032
        x = (torch.randn((T, H), generator=g) * 0.7).to(dtype)
        res = (torch.randn((T, H), generator=g) * 0.7).to(dtype)
        w = (1.0 + 0.1 * torch.randn(H, generator=g)).to(dtype)
        scale = 0.02
        out = dict(x=x, res=res, w=w, scale=torch.tensor(scale), eps=torch.tensor(eps))
        y, _ = _rms_native(x, w, eps)
        out["static_q"] = _q_static(y, scale)
        # fused add (static): the reference kernel rounds input + residual to the dtype first
        z = (x.float() + res.float()).to(dtype)
        y, _ = _rms_native(z, w, eps)
        out["add_static_q"], out["add_static_res"] = _q_static(y, scale), z
        for name, r in (("dyn", None), ("dynres", res)):
            y, new_res = _rms_native(x, w, eps, r)
            yf = y.float()
            amax = yf.abs().amax(dim=-1, keepdim=True)
            ub = torch.tensor([amax.mean().item()])
            out["ub"] = ub
            for ubname, am in (("", amax), ("_ub", amax.clamp(max=ub))):
                sc = (am / FP8_MAX).clamp(min=min_scale)
                out[f"{name}{ubname}_q"] = (yf / sc).clamp(-FP8_MAX, FP8_MAX).to(
                    torch.float8_e4m3fn).view(torch.uint8)
                out[f"{name}{ubname}_s"] = sc
            if new_res is not None:
                out["dynres_res"] = new_res
        if dtype != torch.float32:
            xs = (torch.randn((T, 2 * 520), generator=g) * 2).to(dtype)
            d = 520
            act = torch.nn.functional.silu(xs[:, :d]) * xs[:, d:]
            out["silu_x"] = xs
            out["silu_q"] = _q_static(act, 0.03)
            out["silu_scale"] = torch.tensor(0.03)
        save(f"fused_quant_{tag}.pt", out)


def slot_mapping():
    import numpy as np
    rng = np.random.default_rng(3)
    R, MB, BS = 6, 40, 16
    bt = rng.permutation(R * MB).astype(np.int32).reshape(R, MB)
    req = rng.integers(0, R, size=300).astype(np.int32)
    pos = rng.integers(0, MB * BS, size=300).astype(np.int64)
    idx = req * MB + pos // BS
    slots = bt.flatten()[idx].astype(np.int64) * BS + pos % BS
    save("slot_mapping.pt", dict(bt=torch.from_numpy(bt), req=torch.from_numpy(req),
                                 pos=torch.from_numpy(pos), slots=torch.from_numpy(slots),
                                 block_size=torch.tensor(BS)))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        for name in sys.argv[1:]:
            globals()[name]()
    else:
        attention(); attention_fp8kv(); cache(); fp8_quant(); scaled_mm(); scaled_mm_torch(); fp8_codec(); fused_quant(); slot_mapping()

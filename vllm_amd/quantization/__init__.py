"""fp8 (e4m3fn) linear path of the reference, host side:

  QuantFP8       vllm/model_executor/layers/quantization/input_quant_fp8.py:23-103
  Fp8LinearOp    vllm/model_executor/layers/quantization/utils/w8a8_utils.py:297-377
  PTPCFp8LinearMethod.process_weights / apply
                 vllm/model_executor/layers/quantization/ptpc_fp8.py:78-129

On gfx950 every dispatch branch of the reference (torch._scaled_mm per-tensor /
row-wise, wvSplitKQ skinny, Triton scaled_mm) lands on ONE operator here:
``ops.cutlass_scaled_mm`` = the hand-written MFMA kernel (csrc/scaled_mm.hip)."""
from __future__ import annotations

from typing import Optional

import torch

from .. import _custom_ops as ops


class QuantFP8:
    """Quantize input to fp8, static or dynamic, per-tensor or per-token."""

    def __init__(self, static: bool, per_token: bool, num_token_padding: Optional[int] = None):
        assert not (static and per_token), "static quantization is per-tensor only"
        self.static = static
        self.per_token = per_token
        self.num_token_padding = num_token_padding

    def __call__(self, x: torch.Tensor, scale: Optional[torch.Tensor] = None,
                 scale_ub: Optional[torch.Tensor] = None):
        assert (scale is not None) == self.static
        assert scale_ub is None or (not self.static and self.per_token
                                    and scale_ub.numel() == 1)
        return ops.scaled_fp8_quant(x, scale, num_token_padding=self.num_token_padding,
                                    scale_ub=scale_ub, use_per_token_if_dynamic=self.per_token)

    forward = forward_cuda = __call__


class Fp8LinearOp:
    """w8a8_utils.py:297-377: quantize the activation (unless already fp8), run the
    scaled GEMM, restore the leading dims."""

    def __init__(self, act_quant_static: bool = False, use_per_token_if_dynamic: bool = True,
                 pad_output: bool = False):
        self.act_quant_static = act_quant_static
        self.use_per_token_if_dynamic = use_per_token_if_dynamic
        # the reference pads to 17 rows for torch._scaled_mm's benefit; our kernel
        # does not need it
        self.quant_fp8 = QuantFP8(static=act_quant_static,
                                  per_token=use_per_token_if_dynamic and not act_quant_static,
                                  num_token_padding=17 if pad_output else None)

    def apply(self, input: torch.Tensor, weight: torch.Tensor, weight_scale: torch.Tensor,
              out_dtype: Optional[torch.dtype] = None, input_scale: Optional[torch.Tensor] = None,
              input_scale_ub: Optional[torch.Tensor] = None,
              bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        # weight: fp8 [K, N] column-major (ptpc_fp8.py:114-115: weight.t())
        input_2d = input.view(-1, input.shape[-1])
        output_shape = [*input.shape[:-1], weight.shape[1]]
        if out_dtype is None:
            out_dtype = input.dtype
        if input.dtype != ops.FP8_DTYPE:
            qinput, x_scale = self.quant_fp8(input_2d, input_scale, input_scale_ub)
        else:
            qinput, x_scale = input_2d, input_scale
        m = input_2d.shape[0]
        if qinput.shape[0] != m:           # num_token_padding rows
            qinput = qinput[:m]
        # per-token scales may carry the padding rows; a per-tensor scale (static quantisation:
        # a 0-dim or 1-element tensor, w8a8_utils.py:370-377) passes through untouched
        if x_scale.numel() > 1 and x_scale.shape[0] != m:
            x_scale = x_scale[:m]
        out = ops.cutlass_scaled_mm(qinput, weight, x_scale, weight_scale, out_dtype, bias)
        return out.view(*output_shape)


class PTPCFp8LinearMethod:
    """Per-token activation x per-channel weight fp8 linear (ptpc_fp8.py:78-129)."""

    def __init__(self):
        self.fp8_linear = Fp8LinearOp(act_quant_static=False, use_per_token_if_dynamic=True)

    @staticmethod
    def process_weights_after_loading(weight: torch.Tensor):
        """bf16 [N, K] -> (fp8 [K, N] column-major view, per-channel scale [N, 1])."""
        assert weight.dtype == torch.bfloat16, "PTPC fp8 needs bfloat16 weights (ptpc_fp8.py:96-99)"
        qweight, weight_scale = ops.scaled_fp8_quant(weight, scale=None,
                                                     use_per_token_if_dynamic=True)
        return qweight.t(), weight_scale

    def apply(self, weight_t: torch.Tensor, weight_scale: torch.Tensor, x: torch.Tensor,
              bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        return self.fp8_linear.apply(input=x, weight=weight_t, weight_scale=weight_scale,
                                     input_scale=None, input_scale_ub=None, bias=bias)

"""vllm_amd -- MI355X (gfx950) native implementation of vLLM's quantized decode
hot path: paged attention over bf16 / fp8-e4m3 paged KV caches, KV-cache writes,
fp8 quantisation and the w8a8 fp8 scaled GEMM, as hand-written HIP behind the
reference's own operator surface (``vllm._custom_ops``).

Layout:
  vllm_amd/csrc/            HIP kernels + the C ABI (include/vllm_amd.h) -> libvllm_amd.so
  vllm_amd/_lib.py          ctypes binding of the C ABI (fails loudly if the .so is missing)
  vllm_amd/_torch_ops.py    thin torch.library layer: torch.ops._C / _C_cache_ops / _rocm_C
  vllm_amd/_custom_ops.py   python wrappers with the reference's signatures
  vllm_amd/attention/       PagedAttention helper + attention-backend plugin contract
  vllm_amd/quantization/    QuantFP8 / Fp8LinearOp / PTPC fp8 linear method
  vllm_amd/distributed/     KV-head sharding, column/row-parallel fp8 linear, RCCL all-reduce
"""
__version__ = "0.1.0"

"""vLLM general-plugin hook (vllm/plugins/__init__.py:19-81, entry-point group
``vllm.general_plugins``): importing the op layer overrides the hot-path operators of
torch.ops._C / _C_cache_ops / _rocm_C with the gfx950 HIP kernels.  See INTEGRATION.md."""


def register() -> None:
    from . import _torch_ops  # noqa: F401

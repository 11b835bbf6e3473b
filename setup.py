"""Packaging of the drop-in.  `make -C vllm_amd/csrc && pip install -e . --no-build-isolation`
makes vLLM load the gfx950 kernels through its plugin loader: the entry-point group
`vllm.general_plugins` is what vllm/plugins/__init__.py:19-81 scans at start-up (packaging
pattern of the reference: tests/plugins/vllm_add_dummy_platform/setup.py).  The HIP library is
built by the Makefile (plain hipcc, gfx950 only) and shipped as package data."""
from setuptools import find_packages, setup

setup(
    name="vllm_amd",
    version="0.2.0",
    description="MI355X (gfx950) native quantized decode hot path for vLLM: paged attention, "
                "KV-cache writes, fp8 quantisation, w8a8 fp8 scaled GEMM",
    packages=find_packages(include=["vllm_amd", "vllm_amd.*"]),
    package_data={"vllm_amd": ["csrc/libvllm_amd.so", "csrc/libvllm_amd_torch.so"]},
    python_requires=">=3.10",
    entry_points={"vllm.general_plugins": ["vllm_amd = vllm_amd.plugin:register"]},
)

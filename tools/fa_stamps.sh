#!/bin/bash
# Probe builds of the 256-row prefill kernel with compile-time ablations (run HERE, before gpurun):
#   tools/fa_stamps.sh 0 12 2 31   ->  vllm_amd/csrc/libvllm_amd_probe_abl<N>.so
# then on the GPU: VLLM_AMD_LIB=vllm_amd/csrc/libvllm_amd_probe_abl12.so python tools/fa_stamps.py
set -e
cd "$(dirname "$0")/../vllm_amd/csrc"
make libvllm_amd_probe.so -j8 > /dev/null
for a in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DNDEBUG -DVA_PROBES -DVA_FA_ABL=$a -c prefill_fa.hip -o build/probe_prefill_fa_abl$a.o
  objs=$(ls build/probe_*.o | grep -v "probe_prefill_fa")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libvllm_amd_probe_abl$a.so $objs build/probe_prefill_fa_abl$a.o
done

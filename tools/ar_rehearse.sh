export VLLM_AMD_AR_BENCH_VERBOSE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29641 WORLD_SIZE=2 LOCAL_RANK=0 VLLM_AMD_AR_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
RANK=1 timeout -k 10 60 python -m vllm_amd.distributed.ar_bench > gpurun_out/ar_child1.txt 2>&1 &
RANK=0 timeout -k 10 60 python -m vllm_amd.distributed.ar_bench > gpurun_out/ar_child0.txt 2>&1
wait
tail -5 gpurun_out/ar_child0.txt; tail -3 gpurun_out/ar_child1.txt

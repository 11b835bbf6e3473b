"""Child process of `bench.py --gpus N`: one per rank, on the rank's GPU, with process groups of its own (RCCL +
gloo on MASTER_PORT + 1), running vllm_amd.distributed.bench_custom_all_reduce.  The peer-to-peer all-reduce has
never run across GPUs; in a process of its own a fault in it cannot take the bench's JSON line down.  Rank 0 prints
one JSON object on stdout.

    RANK=.. LOCAL_RANK=.. WORLD_SIZE=.. MASTER_ADDR=127.0.0.1 MASTER_PORT=.. python -m vllm_amd.distributed.ar_bench
"""
import json
import os
import sys


def main() -> int:
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("VLLM_AMD_AR_BENCH_BACKEND", "nccl")     # (tests: "gloo")
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    from vllm_amd.distributed import bench_custom_all_reduce
    try:
        res = bench_custom_all_reduce(device, dist)
    except Exception as e:   # reported, not raised: the parent only relays
        res = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0:
        print(json.dumps(res), flush=True)
    try:
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        pass
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""Host side of the peer-to-peer all-reduce: mirror of
vllm/distributed/device_communicators/custom_all_reduce.py:49-305 (`CustomAllreduce`: same
constructor arguments, `should_custom_ar`, `all_reduce`, `custom_all_reduce`, `close`,
`create_shared_buffer` / `free_shared_buffer`) over the `vllm_amd_ar_*` entry points
(csrc/custom_all_reduce.hip).  The IPC handles of the shared buffers travel through the CPU
process group exactly as in the reference (`dist.all_gather_object`, so the group must not be
NCCL).

HIP graphs as in the reference (:175-215, 253-285): inside `with ar.capture():` an input met under
stream capture is used in place (no copy); the pointer tuples the captured kernels read are filled by
`register_graph_buffers()` when the context ends (IPC handles through the CPU group).  Outside a
capture the input is copied into the pre-registered buffer.  The P2P capability probe
(`gpu_p2p_access_check`) is `torch.cuda.can_device_access_peer`.

A peer that does not arrive within the time limit (30 s; `set_timeout_ms`) never hangs the GPU: the
launch's output is NaN and every later call of this object raises.

STATUS: exercised with simulated ranks on one GPU and with two processes on one GPU; never across
GPUs.  `tensor_model_parallel_all_reduce` only uses it when a `CustomAllreduce` is passed in."""
from __future__ import annotations

import ctypes as C
from contextlib import contextmanager
from typing import Optional, Union

import torch

from .. import _lib
from .._lib import check, lib
from .._torch_ops import _dt

_HANDLE_BYTES = 64


def meta_size() -> int:
    return int(lib.vllm_amd_ar_meta_size())


def is_weak_contiguous(inp: torch.Tensor) -> bool:
    return inp.is_contiguous() or (inp.untyped_storage().nbytes() - inp.storage_offset() * inp.element_size()
                                   == inp.numel() * inp.element_size())


def _ptr_array(ptrs):
    arr = (C.c_void_p * len(ptrs))(*ptrs)
    return arr


class CustomAllreduce:
    _SUPPORTED_WORLD_SIZES = [2, 4, 6, 8]

    def __init__(self, group, device: Union[int, str, torch.device], max_size: int = 2 * 8192 * 1024) -> None:
        import torch.distributed as dist
        self._IS_CAPTURING = False
        self.disabled = True
        self._ptr = None
        self.group = group
        assert dist.get_backend(group) != dist.Backend.NCCL, (
            "CustomAllreduce should be attached to a non-NCCL group.")
        rank = dist.get_rank(group=self.group)
        world_size = dist.get_world_size(group=self.group)
        if world_size == 1 or world_size not in CustomAllreduce._SUPPORTED_WORLD_SIZES:
            return   # no need / unsupported: stays disabled, as in the reference (:92-104)
        if isinstance(device, int):
            device = torch.device(f"cuda:{device}")
        elif isinstance(device, str):
            device = torch.device(device)
        assert isinstance(device, torch.device)
        self.device = device
        # MI355X: the GPUs of a node are fully connected over xGMI; what is checked is that every
        # peer is reachable at all (two ranks may also share one device, as the tests do)
        ids = [None] * world_size
        dist.all_gather_object(ids, device.index or 0, group=self.group)
        fully_connected = all(i == ids[rank] or torch.cuda.can_device_access_peer(ids[rank], i) for i in ids)
        if world_size > 2 and not fully_connected:
            return
        self.disabled = False
        with torch.cuda.device(device):
            self.meta_ptrs = self.create_shared_buffer(meta_size() + max_size, group=group, uncached=True)
            self.buffer_ptrs = self.create_shared_buffer(max_size, group=group)
            # tuples of input pointers the kernels read: 64 bytes per registered / graph-captured input
            # (the reference allocates the same 8 MiB, custom_all_reduce.py:168-176)
            self.rank_data = torch.empty(8 * 1024 * 1024, dtype=torch.uint8, device=self.device)
            self.max_size = max_size
            self.rank = rank
            self.world_size = world_size
            self.fully_connected = fully_connected
            h = C.c_void_p()
            check(lib.vllm_amd_ar_init(C.byref(h), _ptr_array(self.meta_ptrs), max_size,
                                       self.rank_data.data_ptr(), self.rank_data.numel(), world_size, rank,
                                       int(fully_connected)))
            self._ptr = h
            check(lib.vllm_amd_ar_register_buffer(self._ptr, _ptr_array(self.buffer_ptrs)))

    def set_timeout_ms(self, ms: int) -> None:
        check(lib.vllm_amd_ar_set_timeout_ms(self._ptr, int(ms)))

    @contextmanager
    def capture(self):
        """Wrap HIP-graph captures: the addresses used under capture are registered on every rank when
        the context ends (custom_all_reduce.py:175-190)."""
        try:
            self._IS_CAPTURING = True
            yield
        finally:
            self._IS_CAPTURING = False
            if not self.disabled:
                self.register_graph_buffers()

    def register_graph_buffers(self) -> None:
        import torch.distributed as dist
        n = C.c_int(0)
        check(lib.vllm_amd_ar_get_graph_buffer_ipc_meta(self._ptr, None, None, 0, C.byref(n)))
        cap = n.value
        handles = (C.c_ubyte * (_HANDLE_BYTES * max(cap, 1)))()
        offsets = (C.c_int64 * max(cap, 1))()
        if cap:
            check(lib.vllm_amd_ar_get_graph_buffer_ipc_meta(self._ptr, handles, offsets, cap, C.byref(n)))
        mine = [bytes(handles)[:_HANDLE_BYTES * cap], list(offsets)[:cap]]
        world = dist.get_world_size(group=self.group)
        all_data = [None] * world
        dist.all_gather_object(all_data, mine, group=self.group)
        assert all(len(d[1]) == cap for d in all_data), "ranks captured different numbers of all-reduce inputs"
        if cap == 0:
            return
        hb = b"".join(d[0] for d in all_data)
        ob = (C.c_int64 * (world * cap))(*[o for d in all_data for o in d[1]])
        check(lib.vllm_amd_ar_register_graph_buffers(self._ptr, (C.c_ubyte * len(hb)).from_buffer_copy(hb), ob, cap))

    def should_custom_ar(self, inp: torch.Tensor) -> bool:
        if self.disabled:
            return False
        inp_size = inp.numel() * inp.element_size()
        if inp_size % 16 != 0 or not is_weak_contiguous(inp):
            return False
        if self.world_size == 2 or self.fully_connected:
            return inp_size < self.max_size
        return False

    def all_reduce(self, inp: torch.Tensor, *, out: Optional[torch.Tensor] = None,
                   registered: bool = False) -> torch.Tensor:
        """Out-of-place all-reduce; `registered`: inp's pointer was registered on every rank,
        otherwise inp is first copied into the pre-registered buffer."""
        if out is None:
            out = torch.empty_like(inp)
        stream = torch.cuda.current_stream(inp.device).cuda_stream
        with torch.cuda.device(inp.device):
            check(lib.vllm_amd_ar_all_reduce(
                self._ptr, inp.data_ptr(), out.data_ptr(), inp.numel(), _dt(inp),
                None if registered else self.buffer_ptrs[self.rank], 0 if registered else self.max_size,
                stream))
        return out

    def custom_all_reduce(self, input: torch.Tensor) -> Optional[torch.Tensor]:
        """custom_all_reduce.py:253-285: under a graph capture inside `capture()` the input is used in place
        (registered when the context ends); during the warm-up runs of that context only the output is
        allocated; otherwise the input goes through the pre-registered buffer."""
        if self.disabled or not self.should_custom_ar(input):
            return None
        if self._IS_CAPTURING:
            if torch.cuda.is_current_stream_capturing():
                return self.all_reduce(input, registered=True)
            return torch.empty_like(input)      # warm-up: mimic the allocation pattern
        return self.all_reduce(input, registered=False)

    def timed_out(self) -> bool:
        """True if a barrier of this rank ever gave up waiting for a peer (reads a pinned host word: no
        synchronisation; every all_reduce after a time-out raises by itself)."""
        v = C.c_int(0)
        check(lib.vllm_amd_ar_error(self._ptr, C.byref(v)))
        return bool(v.value)

    def close(self) -> None:
        if not self.disabled and self._ptr:
            check(lib.vllm_amd_ar_dispose(self._ptr))
            self._ptr = None
            self.free_shared_buffer(self.meta_ptrs, rank=self.rank)
            self.free_shared_buffer(self.buffer_ptrs, rank=self.rank)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def create_shared_buffer(size_in_bytes: int, group=None, uncached: Optional[bool] = False) -> list[int]:
        import torch.distributed as dist
        p = C.c_void_p()
        handle = (C.c_ubyte * _HANDLE_BYTES)()
        check(lib.vllm_amd_ar_alloc_shared(size_in_bytes, C.byref(p), handle))
        world_size = dist.get_world_size(group=group)
        rank = dist.get_rank(group=group)
        handles = [None] * world_size
        dist.all_gather_object(handles, bytes(handle), group=group)
        pointers: list[int] = []
        for i, h in enumerate(handles):
            if i == rank:
                pointers.append(p.value)
            else:
                q = C.c_void_p()
                buf = (C.c_ubyte * _HANDLE_BYTES).from_buffer_copy(h)
                check(lib.vllm_amd_ar_open_handle(buf, C.byref(q)))
                pointers.append(q.value)
        return pointers

    @staticmethod
    def free_shared_buffer(pointers: list[int], group=None, rank: Optional[int] = 0) -> None:
        import torch.distributed as dist
        if rank is None:
            rank = dist.get_rank(group=group)
        for i, p in enumerate(pointers):
            if i != rank:
                lib.vllm_amd_ar_close_handle(p)
        check(lib.vllm_amd_ar_free_shared(pointers[rank]))

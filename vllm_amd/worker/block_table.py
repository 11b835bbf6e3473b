"""Producer of the integer tensors the hot path consumes: the per-request block table
and the per-token slot mapping (mirror of vllm/v1/worker/block_table.py:13-175).

Same class and method names as the reference, different data flow: the reference
computes the slot mapping with numpy on the host (`compute_slot_mapping`, :84-99) and
copies it to the GPU each step (`commit_slot_mapping`, :105-107).  Here
`compute_slot_mapping` takes DEVICE `req_indices` / `positions` and runs
`vllm_amd_compute_slot_mapping` on the current stream against the committed device block
table, so `commit_slot_mapping` has nothing left to copy.  Row edits (`append_row`,
`add_row`, `move_row`, `swap_row`) stay on the host copy and reach the device through
`commit_block_table`, exactly as in the reference.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _torch_ops


def cdiv(a: int, b: int) -> int:
    return -(a // -b)


class BlockTable:

    def __init__(self, block_size: int, max_num_reqs: int, max_num_blocks_per_req: int,
                 max_num_batched_tokens: int, pin_memory: bool, device: torch.device):
        self.block_size = block_size
        self.max_num_reqs = max_num_reqs
        self.max_num_blocks_per_req = max_num_blocks_per_req
        self.max_num_batched_tokens = max_num_batched_tokens
        self.pin_memory = pin_memory
        self.device = device

        self.block_table = torch.zeros((max_num_reqs, max_num_blocks_per_req), device=device,
                                       dtype=torch.int32)
        self.block_table_cpu = torch.zeros((max_num_reqs, max_num_blocks_per_req), device="cpu",
                                           dtype=torch.int32, pin_memory=pin_memory)
        self.block_table_np = self.block_table_cpu.numpy()
        self.num_blocks_per_row = np.zeros(max_num_reqs, dtype=np.int32)
        self.slot_mapping = torch.zeros(max_num_batched_tokens, dtype=torch.int64, device=device)
        # kept for callers that read the reference's host copies (gpu_model_runner); they are
        # filled only when compute_slot_mapping is handed host (numpy / CPU) inputs
        self.slot_mapping_cpu = torch.zeros(max_num_batched_tokens, dtype=torch.int64, device="cpu",
                                            pin_memory=pin_memory)
        self.slot_mapping_np = self.slot_mapping_cpu.numpy()
        self._dirty = False   # host rows edited since the last commit_block_table

    # ---- host-side row edits (block_table.py:54-82) ------------------------------------
    def append_row(self, block_ids: list[int], row_idx: int) -> None:
        if not block_ids:
            return
        num_blocks = len(block_ids)
        start = self.num_blocks_per_row[row_idx]
        self.num_blocks_per_row[row_idx] += num_blocks
        self.block_table_np[row_idx, start:start + num_blocks] = block_ids
        self._dirty = True

    def add_row(self, block_ids: list[int], row_idx: int) -> None:
        self.num_blocks_per_row[row_idx] = 0
        self.append_row(block_ids, row_idx)

    def move_row(self, src: int, tgt: int) -> None:
        num_blocks = self.num_blocks_per_row[src]
        self.block_table_np[tgt, :num_blocks] = self.block_table_np[src, :num_blocks]
        self.num_blocks_per_row[tgt] = num_blocks
        self._dirty = True

    def swap_row(self, src: int, tgt: int) -> None:
        n_src, n_tgt = self.num_blocks_per_row[src], self.num_blocks_per_row[tgt]
        self.num_blocks_per_row[src], self.num_blocks_per_row[tgt] = n_tgt, n_src
        self.block_table_np[[src, tgt]] = self.block_table_np[[tgt, src]]
        self._dirty = True

    # ---- device side ------------------------------------------------------------------------
    def commit_block_table(self, num_reqs: int) -> None:
        self.block_table[:num_reqs].copy_(self.block_table_cpu[:num_reqs], non_blocking=True)
        self._dirty = False

    def compute_slot_mapping(self, req_indices, positions) -> None:
        """slot_mapping[:n] from req_indices[n] (int32) and positions[n] (int64).

        DEVICE tensors: the device kernel runs against the COMMITTED device block table, so row
        edits must have been committed first (asserted; the reference's host computation has no
        such ordering because it reads block_table_cpu).  numpy arrays / CPU tensors (what the
        reference's gpu_model_runner passes, block_table.py:84-99): computed on the host from
        block_table_np exactly as the reference does, slot_mapping_np / slot_mapping_cpu are
        filled, and commit_slot_mapping copies them to the device."""
        if isinstance(req_indices, torch.Tensor) and req_indices.is_cuda:
            if self._dirty:
                raise RuntimeError("BlockTable.compute_slot_mapping on device tensors needs "
                                   "commit_block_table() after the last row edit")
            _torch_ops.compute_slot_mapping(self.slot_mapping, self.block_table, req_indices,
                                            positions, self.block_size)
            self._host_slots = 0
            return
        req = req_indices.numpy() if isinstance(req_indices, torch.Tensor) else np.asarray(req_indices)
        pos = positions.numpy() if isinstance(positions, torch.Tensor) else np.asarray(positions)
        idx = req.astype(np.int64) * self.max_num_blocks_per_req + pos // self.block_size
        blocks = self.block_table_np.ravel()[idx].astype(np.int64)
        n = req.shape[0]
        np.add(blocks * self.block_size, pos % self.block_size, out=self.slot_mapping_np[:n])
        self._host_slots = n

    def commit_slot_mapping(self, num_tokens: int) -> None:
        """Device inputs: nothing to copy, compute_slot_mapping wrote the device tensor.  Host
        inputs: the reference's copy (block_table.py:105-107)."""
        if getattr(self, "_host_slots", 0):
            self.slot_mapping[:num_tokens].copy_(self.slot_mapping_cpu[:num_tokens], non_blocking=True)

    def clear(self) -> None:
        self.block_table.fill_(0)
        self.block_table_cpu.fill_(0)

    def get_device_tensor(self) -> torch.Tensor:
        return self.block_table

    def get_cpu_tensor(self) -> torch.Tensor:
        return self.block_table_cpu

    def get_numpy_array(self) -> np.ndarray:
        return self.block_table_np


class MultiGroupBlockTable:
    """The BlockTables of each KV cache group (block_table.py:126-175)."""

    def __init__(self, max_num_reqs: int, max_model_len: int, max_num_batched_tokens: int,
                 pin_memory: bool, device: torch.device, block_sizes: list[int]) -> None:
        self.block_tables = [
            BlockTable(bs, max_num_reqs, cdiv(max_model_len, bs), max_num_batched_tokens,
                       pin_memory, device) for bs in block_sizes
        ]

    def append_row(self, block_ids, row_idx: int) -> None:
        for i, bt in enumerate(self.block_tables):
            bt.append_row(block_ids[i], row_idx)

    def add_row(self, block_ids, row_idx: int) -> None:
        for i, bt in enumerate(self.block_tables):
            bt.add_row(block_ids[i], row_idx)

    def move_row(self, src: int, tgt: int) -> None:
        for bt in self.block_tables:
            bt.move_row(src, tgt)

    def swap_row(self, src: int, tgt: int) -> None:
        for bt in self.block_tables:
            bt.swap_row(src, tgt)

    def compute_slot_mapping(self, req_indices: torch.Tensor, positions: torch.Tensor) -> None:
        for bt in self.block_tables:
            bt.compute_slot_mapping(req_indices, positions)

    def commit_block_table(self, num_reqs: int) -> None:
        for bt in self.block_tables:
            bt.commit_block_table(num_reqs)

    def commit_slot_mapping(self, num_tokens: int) -> None:
        for bt in self.block_tables:
            bt.commit_slot_mapping(num_tokens)

    def clear(self) -> None:
        for bt in self.block_tables:
            bt.clear()

    def __getitem__(self, idx: int) -> BlockTable:
        return self.block_tables[idx]

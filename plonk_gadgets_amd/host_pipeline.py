"""For callers whose consumer sits on the host: emit in chunks on the GPU and stream every finished chunk to pinned
host memory over PCIe while the next one is being emitted (two device buffers, two pinned buffers, one copy stream).
PCIe, not HBM, bounds this path ("PCIe note") -- it exists so that the bound is a measured number,
and so that a host-side consumer has something to call; the columns are normally left resident."""
from __future__ import annotations

import torch

from .distributed import columns_in, packed_layout


class HostPipeline:
    """range_check over a long witness array, chunk by chunk; `consume(host_columns, chunk_index, first_item)` sees the
    chunk in pinned host memory (a Columns view that is reused two chunks later)."""

    def __init__(self, engine, min_range, max_range, chunk: int):
        self.engine, self.mn, self.mx, self.chunk = engine, min_range, max_range, chunk
        self.lay = engine.range_check_layout(min_range, max_range, chunk)
        _, _, self.words = packed_layout(self.lay.n_gates, self.lay.n_vars)
        dev = engine.device
        self.dev_buf = [torch.empty(self.words, dtype=torch.int64, device=dev) for _ in range(2)]
        self.host_buf = [torch.empty(self.words, dtype=torch.int64, pin_memory=True) for _ in range(2)]
        self.dev_cols = [columns_in(b, self.lay.n_gates, self.lay.n_vars) for b in self.dev_buf]
        self.host_cols = [columns_in(b, self.lay.n_gates, self.lay.n_vars) for b in self.host_buf]
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.emitted = [torch.cuda.Event() for _ in range(2)]
        self.copied = [torch.cuda.Event() for _ in range(2)]
        self.res = torch.empty((chunk,), dtype=torch.int64, device=dev)

    def bytes_per_chunk(self) -> int:
        return self.words * 8

    def run(self, witness: torch.Tensor, gate_base: int = 0, var_base: int = 0, consume=None):
        total = witness.shape[0]
        assert total % self.chunk == 0
        G, V = self.lay.gates_per_item, self.lay.vars_per_item
        compute = torch.cuda.current_stream(self.engine.device)
        n = total // self.chunk
        for k in range(n + 1):
            if k < n:
                b = k & 1
                if k >= 2:
                    compute.wait_event(self.copied[b])  # the buffer's previous chunk has left the device
                first = k * self.chunk
                self.engine.range_check_batch(self.mn, self.mx, witness[first:first + self.chunk], gate_base + first * G,
                                              var_base + first * V, out=self.dev_cols[b], result_vars=self.res)
                self.emitted[b].record(compute)
                with torch.cuda.stream(self.copy_stream):
                    self.copy_stream.wait_event(self.emitted[b])
                    self.host_buf[b].copy_(self.dev_buf[b], non_blocking=True)
                    self.copied[b].record(self.copy_stream)
            if k >= 1 and consume is not None:
                pb = (k - 1) & 1
                self.copied[pb].synchronize()
                consume(self.host_cols[pb], k - 1, (k - 1) * self.chunk)
        self.copy_stream.synchronize()

"""Host-side handle on the gfx950 engine: owns a pg_engine, allocates output columns as torch tensors (device
memory + streams are torch's; nothing else of torch is used) and launches the batched gadgets through the C ABI."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from . import _lib
from .scalar import BlsScalar


class PgError(RuntimeError):
    def __init__(self, status: int, where: str):
        lib = _lib.load()
        self.status = status
        super().__init__(f"{where}: {lib.pg_status_string(status).decode()} ({lib.pg_last_error().decode()})")


class NonExistingInverse(PgError):
    """Error::NonExistingInverse, /root/reference/src/errors.rs:17"""


@dataclass
class Layout:
    num_bits: int
    gates_per_item: int
    vars_per_item: int
    n_gates: int
    n_vars: int


@dataclass
class Columns:
    """The 8 live columns + variable table of a batch (device tensors, int64 storage of the u64 limbs/indices).
    Row r is gate gate_base + r; var_values[v] is Variable(var_base + v)."""
    q_m: torch.Tensor
    q_l: torch.Tensor
    q_r: torch.Tensor
    q_o: torch.Tensor
    q_c: torch.Tensor
    w_l: torch.Tensor
    w_r: torch.Tensor
    w_o: torch.Tensor
    var_values: torch.Tensor
    gate_base: int = 0
    var_base: int = 0

    SCALAR_COLS = ("q_m", "q_l", "q_r", "q_o", "q_c")
    WIRE_COLS = ("w_l", "w_r", "w_o")

    @staticmethod
    def allocate(n_gates: int, n_vars: int, device, gate_base: int = 0, var_base: int = 0) -> "Columns":
        sc = [torch.empty((n_gates, 4), dtype=torch.int64, device=device) for _ in range(5)]
        wc = [torch.empty((n_gates,), dtype=torch.int64, device=device) for _ in range(3)]
        vv = torch.empty((n_vars, 4), dtype=torch.int64, device=device)
        return Columns(*sc, *wc, vv, gate_base, var_base)

    def as_c(self) -> _lib.ColumnsC:
        return _lib.ColumnsC(*[getattr(self, n).data_ptr() for n in
                               ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o", "var_values")])

    def nbytes(self) -> int:
        return sum(getattr(self, n).numel() * 8 for n in self.SCALAR_COLS + self.WIRE_COLS + ("var_values",))

    def to_numpy(self) -> dict:
        import numpy as np
        return {n: getattr(self, n).cpu().numpy().view(np.uint64)
                for n in self.SCALAR_COLS + self.WIRE_COLS + ("var_values",)}


class Engine:
    """pg_engine: one per GPU per host thread."""

    def __init__(self, device: int | torch.device | None = None):
        if not torch.cuda.is_available():
            raise RuntimeError("plonk_gadgets_amd needs a gfx950 GPU: there is no CPU path")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self._lib = _lib.load()
        torch.cuda.init()
        h = C.c_void_p()
        st = self._lib.pg_engine_create(self.device.index or 0, C.byref(h))
        if st != 0:
            raise PgError(st, "pg_engine_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pg_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- range_check ---------------------------------------------------
    def range_check_layout(self, min_range: BlsScalar, max_range: BlsScalar, batch: int) -> Layout:
        lay = _lib.LayoutC()
        st = self._lib.pg_range_check_layout(C.byref(min_range.c), C.byref(max_range.c), batch, C.byref(lay))
        if st != 0:
            raise PgError(st, "pg_range_check_layout")
        return Layout(*[int(getattr(lay, f)) for f in ("num_bits", "gates_per_item", "vars_per_item", "n_gates", "n_vars")])

    def range_check_batch(self, min_range: BlsScalar, max_range: BlsScalar, witness: torch.Tensor,
                          gate_base: int = 0, var_base: int = 0, out: Columns | None = None,
                          result_vars: torch.Tensor | None = None, want_result_vars: bool = True):
        """for each witness: AllocatedScalar::allocate + range_check (/root/reference/src/range.rs:27-43).
        witness: int64[batch, 4] device tensor of Montgomery limbs.  Returns (Columns, result_vars)."""
        assert witness.is_cuda and witness.dtype == torch.int64 and witness.dim() == 2 and witness.shape[1] == 4
        assert witness.is_contiguous()
        batch = witness.shape[0]
        lay = self.range_check_layout(min_range, max_range, batch)
        if out is None:
            out = Columns.allocate(lay.n_gates, lay.n_vars, self.device, gate_base, var_base)
        if result_vars is None and want_result_vars:
            result_vars = torch.empty((batch,), dtype=torch.int64, device=self.device)
        cols = out.as_c()
        st = self._lib.pg_range_check_batch(self._h, C.byref(min_range.c), C.byref(max_range.c), witness.data_ptr(),
                                            batch, gate_base, var_base, C.byref(cols),
                                            result_vars.data_ptr() if result_vars is not None else None, self._stream())
        if st != 0:
            raise PgError(st, "pg_range_check_batch")
        return out, result_vars

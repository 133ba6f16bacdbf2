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
    slab: torch.Tensor | None = None  # allocate(spread_gib=...): the one allocation the nine arrays are views of

    SCALAR_COLS = ("q_m", "q_l", "q_r", "q_o", "q_c")
    WIRE_COLS = ("w_l", "w_r", "w_o")

    @staticmethod
    def allocate(n_gates: int, n_vars: int, device, gate_base: int = 0, var_base: int = 0, spread_gib: float = 0) -> "Columns":
        """spread_gib = 0: nine allocations, one after the other.
        spread_gib > 0: ONE allocation, the five selector columns spread_gib GiB apart, the wire columns and the variable
        table behind the last.  The emitters write the same row of all five selector columns at once; on MI355X five such streams
        inside one stretch of a few GiB of physical memory run 10-15 % slower than five streams tens of GiB apart (the 12-high
        stacks' ranks lie one after the other in the address space: streams in one rank share its banks).  A circuit of a few
        GB therefore does better in a slab that spans much of the card (C3: 0.575 -> 0.50 ms per step with the columns 16 GiB
        and more apart, tools/placement_sweep.py); columns of tens of GB each lie that far apart anyway.  The memory between the
        arrays belongs to the slab: the caller's to use for whatever else it streams (Columns.slab), or the price of the layout."""
        if spread_gib <= 0:
            sc = [torch.empty((n_gates, 4), dtype=torch.int64, device=device) for _ in range(5)]
            wc = [torch.empty((n_gates,), dtype=torch.int64, device=device) for _ in range(3)]
            vv = torch.empty((n_vars, 4), dtype=torch.int64, device=device)
            return Columns(*sc, *wc, vv, gate_base, var_base)
        # q_m q_l q_r q_o q_c a stride apart, then w_l w_r w_o var_values back to back behind q_c (of the layouts measured --
        # the others between the selector columns, before them, a stride apart themselves -- the best: tools/placement_policy.py);
        # the arithmetic is the library's (pg_columns_slab_layout: what a caller of the C ABI uses for its own block)
        off = (C.c_uint64 * 9)()
        total = C.c_uint64()
        st = _lib.load().pg_columns_slab_layout(n_gates, n_vars, int(spread_gib * (1 << 30)), off, C.byref(total))
        if st != 0:
            raise PgError(st, "pg_columns_slab_layout")
        align = 2 << 20
        slab = torch.empty(((total.value + align) // 8,), dtype=torch.int64, device=device)
        first = ((-slab.data_ptr()) % align) // 8
        at = [first + o // 8 for o in off]
        sel = [slab[at[c]:at[c] + n_gates * 4].view(n_gates, 4) for c in range(5)]
        wc = [slab[at[5 + c]:at[5 + c] + n_gates] for c in range(3)]
        vv = slab[at[8]:at[8] + n_vars * 4].view(n_vars, 4)
        cols = Columns(*sel, *wc, vv, gate_base, var_base)
        cols.slab = slab
        return cols

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

    def range_check_sharded_batch(self, min_range: BlsScalar, max_range: BlsScalar, witness_local: torch.Tensor, total: int,
                                  rank: int, world: int, gate_base: int = 0, var_base: int = 0, out: Columns | None = None,
                                  result_vars: torch.Tensor | None = None):
        """this rank's shard of a `total`-item range_check batch, emitted at its global numbering
        (pg_range_check_sharded_batch; no communication).  Returns (Columns, result_vars)."""
        self._check_scalars(witness_local)
        shard = _lib.ShardC()
        st = self._lib.pg_range_check_shard_layout(C.byref(min_range.c), C.byref(max_range.c), total, rank, world, gate_base,
                                                   var_base, C.byref(shard))
        if st != 0:
            raise PgError(st, "pg_range_check_shard_layout")
        assert witness_local.shape[0] == shard.hi - shard.lo, (witness_local.shape, shard.lo, shard.hi)
        if out is None:
            out = Columns.allocate(shard.n_gates, shard.n_vars, self.device, shard.gate_base, shard.var_base)
        if result_vars is None:
            result_vars = torch.empty((witness_local.shape[0],), dtype=torch.int64, device=self.device)
        cols = out.as_c()
        st = self._lib.pg_range_check_sharded_batch(self._h, C.byref(min_range.c), C.byref(max_range.c), witness_local.data_ptr(),
                                                    total, rank, world, gate_base, var_base, C.byref(cols),
                                                    result_vars.data_ptr(), C.byref(shard), self._stream())
        if st != 0:
            raise PgError(st, "pg_range_check_sharded_batch")
        return out, result_vars

    # ---- encodings ------------------------------------------------------------------
    def scalars_from_canonical(self, raw: torch.Tensor):
        """int64[batch, 4] canonical little-endian values (BlsScalar::to_bytes) -> (Montgomery limbs, bad mask, bad count);
        values >= q come out as 0 and are flagged (BlsScalar::from_bytes would return Err)"""
        assert raw.is_cuda and raw.dtype == torch.int64 and raw.dim() == 2 and raw.shape[1] == 4 and raw.is_contiguous()
        out = torch.empty_like(raw)
        bad = torch.zeros((raw.shape[0],), dtype=torch.uint8, device=raw.device)
        n = C.c_uint64()
        st = self._lib.pg_scalars_from_canonical_batch(self._h, raw.data_ptr(), raw.shape[0], out.data_ptr(), bad.data_ptr(),
                                                       C.byref(n), self._stream())
        if st not in (0, 6):  # 6 = PG_ERR_BAD_ENCODING: flagged items, reported through the mask and the count
            raise PgError(st, "pg_scalars_from_canonical_batch")
        return out, bad, int(n.value)

    def scalars_to_canonical(self, scalars: torch.Tensor) -> torch.Tensor:
        assert scalars.is_cuda and scalars.dtype == torch.int64 and scalars.dim() == 2 and scalars.shape[1] == 4
        assert scalars.is_contiguous()
        out = torch.empty_like(scalars)
        st = self._lib.pg_scalars_to_canonical_batch(self._h, scalars.data_ptr(), scalars.shape[0], out.data_ptr(), self._stream())
        if st != 0:
            raise PgError(st, "pg_scalars_to_canonical_batch")
        return out

    def range_check_structure_batch(self, min_range: BlsScalar, max_range: BlsScalar, batch: int, gate_base: int,
                                    var_base: int, out: Columns):
        """selectors and wire indices of range_check_batch's rows -- no witnesses, `out.var_values` is left alone"""
        cols = out.as_c()
        st = self._lib.pg_range_check_structure_batch(self._h, C.byref(min_range.c), C.byref(max_range.c), batch, gate_base,
                                                      var_base, C.byref(cols), self._stream())
        if st != 0:
            raise PgError(st, "pg_range_check_structure_batch")
        return out

    # ---- helpers -------------------------------------------------------------
    def _layout(self, lay: "_lib.LayoutC") -> Layout:
        return Layout(*[int(getattr(lay, f)) for f in ("num_bits", "gates_per_item", "vars_per_item", "n_gates", "n_vars")])

    @staticmethod
    def _check_scalars(t: torch.Tensor, batch: int | None = None):
        assert t.is_cuda and t.dtype == torch.int64 and t.dim() == 2 and t.shape[1] == 4 and t.is_contiguous()
        assert batch is None or t.shape[0] == batch

    @staticmethod
    def _check_vars(t: torch.Tensor, batch: int):
        assert t.is_cuda and t.dtype == torch.int64 and t.dim() == 1 and t.shape[0] == batch and t.is_contiguous()

    def _out(self, out, n_gates, n_vars, gate_base, var_base):
        return out if out is not None else Columns.allocate(n_gates, n_vars, self.device, gate_base, var_base)

    # ---- max_bound -------------------------------------------------------------
    def max_bound_layout(self, max_range: BlsScalar, batch: int) -> Layout:
        lay = _lib.LayoutC()
        st = self._lib.pg_max_bound_layout(C.byref(max_range.c), batch, C.byref(lay))
        if st != 0:
            raise PgError(st, "pg_max_bound_layout")
        return self._layout(lay)

    def max_bound_batch(self, max_range: BlsScalar, witness: torch.Tensor, gate_base: int = 0, var_base: int = 0,
                        out: Columns | None = None):
        """for each witness: allocate + max_bound(composer, max_range, w) (/root/reference/src/range.rs:82-113).
        Returns (Columns, result_vars, num_bits)."""
        self._check_scalars(witness)
        batch = witness.shape[0]
        lay = self.max_bound_layout(max_range, batch)
        out = self._out(out, lay.n_gates, lay.n_vars, gate_base, var_base)
        res = torch.empty((batch,), dtype=torch.int64, device=self.device)
        cols = out.as_c()
        st = self._lib.pg_max_bound_batch(self._h, C.byref(max_range.c), witness.data_ptr(), batch, gate_base, var_base,
                                          C.byref(cols), res.data_ptr(), self._stream())
        if st != 0:
            raise PgError(st, "pg_max_bound_batch")
        return out, res, lay.num_bits

    def max_bound_ragged_batch(self, max_range: torch.Tensor, witness: torch.Tensor, gate_base: int = 0,
                               var_base: int = 0):
        """one public bound PER ITEM (device tensor): plan (ladder bits + prefix sums on the device) then emit.
        Returns (Columns, result_vars, num_bits[int32 tensor], layout)."""
        self._check_scalars(witness)
        batch = witness.shape[0]
        self._check_scalars(max_range, batch)
        nb = torch.empty((batch,), dtype=torch.int32, device=self.device)
        roff = torch.empty((batch + 1,), dtype=torch.int64, device=self.device)
        voff = torch.empty((batch + 1,), dtype=torch.int64, device=self.device)
        lay = _lib.LayoutC()
        st = self._lib.pg_max_bound_ragged_plan(self._h, max_range.data_ptr(), batch, nb.data_ptr(), roff.data_ptr(),
                                                voff.data_ptr(), C.byref(lay), self._stream())
        if st != 0:
            raise PgError(st, "pg_max_bound_ragged_plan")
        lay = self._layout(lay)
        out = Columns.allocate(lay.n_gates, lay.n_vars, self.device, gate_base, var_base)
        res = torch.empty((batch,), dtype=torch.int64, device=self.device)
        cols = out.as_c()
        st = self._lib.pg_max_bound_ragged_batch(self._h, max_range.data_ptr(), witness.data_ptr(), batch, nb.data_ptr(),
                                                 roff.data_ptr(), voff.data_ptr(), gate_base, var_base, C.byref(cols),
                                                 res.data_ptr(), self._stream())
        if st != 0:
            raise PgError(st, "pg_max_bound_ragged_batch")
        return out, res, nb, lay

    # ---- scalar gadgets ----------------------------------------------------------
    def _scalar2(self, fn_name, per_item, a_var, a_val, b_var, b_val, gate_base, var_base, out):
        batch = a_var.shape[0]
        self._check_vars(a_var, batch)
        self._check_vars(b_var, batch)
        self._check_scalars(a_val, batch)
        self._check_scalars(b_val, batch)
        out = self._out(out, per_item * batch, per_item * batch, gate_base, var_base)
        res = torch.empty((batch,), dtype=torch.int64, device=self.device)
        cols = out.as_c()
        st = getattr(self._lib, fn_name)(self._h, a_var.data_ptr(), a_val.data_ptr(), b_var.data_ptr(), b_val.data_ptr(),
                                         batch, gate_base, var_base, C.byref(cols), res.data_ptr(), self._stream())
        if st != 0:
            raise PgError(st, fn_name)
        return out, res

    def conditionally_select_zero_batch(self, x_var, x_val, select_var, select_val, gate_base=0, var_base=0, out=None):
        """/root/reference/src/scalar.rs:21-27, per item"""
        return self._scalar2("pg_conditionally_select_zero_batch", 1, x_var, x_val, select_var, select_val, gate_base,
                             var_base, out)

    def conditionally_select_one_batch(self, y_var, y_val, selector_var, selector_val, gate_base=0, var_base=0, out=None):
        """/root/reference/src/scalar.rs:36-59, per item"""
        return self._scalar2("pg_conditionally_select_one_batch", 4, y_var, y_val, selector_var, selector_val, gate_base,
                             var_base, out)

    def maybe_equal_batch(self, a_var, a_val, b_var, b_val, gate_base=0, var_base=0, out=None):
        """/root/reference/src/scalar.rs:105-140, per item"""
        return self._scalar2("pg_maybe_equal_batch", 3, a_var, a_val, b_var, b_val, gate_base, var_base, out)

    def _error_plan(self, fn_name, values, batch):
        roff = torch.empty((batch + 1,), dtype=torch.int64, device=self.device)
        voff = torch.empty((batch + 1,), dtype=torch.int64, device=self.device)
        err = torch.zeros((max(batch, 1),), dtype=torch.uint8, device=self.device)
        lay, nerr = _lib.LayoutC(), C.c_uint64()
        st = getattr(self._lib, fn_name)(self._h, values.data_ptr(), batch, roff.data_ptr(), voff.data_ptr(),
                                         err.data_ptr(), C.byref(lay), C.byref(nerr), self._stream())
        if st not in (0, 1):
            raise PgError(st, fn_name)
        return roff, voff, err[:batch], self._layout(lay), int(nerr.value)

    def is_non_zero_batch(self, var, value_assigned, gate_base=0, var_base=0, zero_var=0):
        """/root/reference/src/scalar.rs:63-97, per item.  Returns (Columns, err_mask[uint8], err_count): items whose
        value is 0 (Err(NonExistingInverse)) keep their partial emission (1 row, 1 variable)."""
        batch = var.shape[0]
        self._check_vars(var, batch)
        self._check_scalars(value_assigned, batch)
        roff, voff, err, lay, nerr = self._error_plan("pg_is_non_zero_plan", value_assigned, batch)
        out = Columns.allocate(lay.n_gates, lay.n_vars, self.device, gate_base, var_base)
        cols = out.as_c()
        st = self._lib.pg_is_non_zero_batch(self._h, var.data_ptr(), value_assigned.data_ptr(), batch, roff.data_ptr(),
                                            voff.data_ptr(), gate_base, var_base, zero_var, C.byref(cols), self._stream())
        if st != 0:
            raise PgError(st, "pg_is_non_zero_batch")
        return out, err, nerr

    def scalar_mix_batch(self, v, y, s, a, b, gate_base=0, var_base=0, zero_var=0):
        """BASELINE config 3, one launch: per item 5 x add_input, is_non_zero(v), conditionally_select_one(y, s),
        maybe_equal(a, b).  Returns (Columns, result_vars[batch,2], err_mask, err_count, layout)."""
        batch = v.shape[0]
        for t in (v, y, s, a, b):
            self._check_scalars(t, batch)
        roff, voff, err, lay, nerr = self._error_plan("pg_scalar_mix_plan", v, batch)
        out = Columns.allocate(lay.n_gates, lay.n_vars, self.device, gate_base, var_base)
        res = torch.empty((batch, 2), dtype=torch.int64, device=self.device)
        cols = out.as_c()
        st = self._lib.pg_scalar_mix_batch(self._h, v.data_ptr(), y.data_ptr(), s.data_ptr(), a.data_ptr(), b.data_ptr(),
                                           batch, roff.data_ptr(), voff.data_ptr(), gate_base, var_base, zero_var,
                                           C.byref(cols), res.data_ptr(), self._stream())
        if st != 0:
            raise PgError(st, "pg_scalar_mix_batch")
        return out, res, err, nerr, lay

    # ---- diagnostics -----------------------------------------------------------------
    def fill_columns(self, cols: "Columns", n_gates: int | None = None, n_vars: int | None = None, rows_per_tile: int = 0,
                     pattern: int = 0x0123456789ABCDEF):
        """pg_fill_columns: the emitters' store stream with nothing behind it, over these nine arrays (the store ceiling of
        a workload ON ITS OWN ARRAYS; bench.py times it beside every workload)"""
        n_gates = cols.q_m.shape[0] if n_gates is None else n_gates
        n_vars = cols.var_values.shape[0] if n_vars is None else n_vars
        cc = cols.as_c()
        st = self._lib.pg_fill_columns(self._h, C.byref(cc), n_gates, n_vars, rows_per_tile, pattern, self._stream())
        if st != 0:
            raise PgError(st, "pg_fill_columns")

    def fill_bytes(self, dst: torch.Tensor, streams: int = 5, pattern: int = 0x0123456789ABCDEF):
        """bare 16-B-per-lane streaming fill of `dst` as `streams` concurrent parts (the write ceiling bench.py quotes)"""
        nbytes = dst.numel() * dst.element_size()
        st = self._lib.pg_fill_bytes(self._h, dst.data_ptr(), nbytes, streams, pattern, self._stream())
        if st != 0:
            raise PgError(st, "pg_fill_bytes")

    # ---- gadgets on witnesses that are already allocated (the reference's exact argument: an AllocatedScalar) ----
    def range_check_allocated_batch(self, min_range: BlsScalar, max_range: BlsScalar, witness_var: torch.Tensor,
                                    witness: torch.Tensor, gate_base: int = 0, var_base: int = 0):
        """range_check(composer, min, max, AllocatedScalar{var, scalar}) per item: no allocate, 2n+523 variables"""
        self._check_scalars(witness)
        batch = witness.shape[0]
        self._check_vars(witness_var, batch)
        lay = self.range_check_layout(min_range, max_range, batch)
        out = Columns.allocate(lay.n_gates, lay.n_vars - batch, self.device, gate_base, var_base)
        res = torch.empty((batch,), dtype=torch.int64, device=self.device)
        cols = out.as_c()
        st = self._lib.pg_range_check_allocated_batch(self._h, C.byref(min_range.c), C.byref(max_range.c),
                                                      witness_var.data_ptr(), witness.data_ptr(), batch, gate_base,
                                                      var_base, C.byref(cols), res.data_ptr(), self._stream())
        if st != 0:
            raise PgError(st, "pg_range_check_allocated_batch")
        return out, res

    def max_bound_allocated_batch(self, max_range: BlsScalar, witness_var: torch.Tensor, witness: torch.Tensor,
                                  gate_base: int = 0, var_base: int = 0):
        self._check_scalars(witness)
        batch = witness.shape[0]
        self._check_vars(witness_var, batch)
        lay = self.max_bound_layout(max_range, batch)
        out = Columns.allocate(lay.n_gates, lay.n_vars - batch, self.device, gate_base, var_base)
        res = torch.empty((batch,), dtype=torch.int64, device=self.device)
        cols = out.as_c()
        st = self._lib.pg_max_bound_allocated_batch(self._h, C.byref(max_range.c), witness_var.data_ptr(),
                                                    witness.data_ptr(), batch, gate_base, var_base, C.byref(cols),
                                                    res.data_ptr(), self._stream())
        if st != 0:
            raise PgError(st, "pg_max_bound_allocated_batch")
        return out, res, lay.num_bits

    def check_rows(self, cols: Columns, var_base: int | None = None, zero_var: int = 0) -> int:
        """every row of a self-contained batch satisfied?  -1, or the first failing row (device-side check)"""
        bad = C.c_int64()
        cc = cols.as_c()
        st = self._lib.pg_check_rows(self._h, C.byref(cc), cols.q_m.shape[0], cols.var_base if var_base is None else var_base,
                                     cols.var_values.shape[0], zero_var, C.byref(bad), self._stream())
        if st != 0:
            raise PgError(st, "pg_check_rows")
        return bad.value

    # ---- two-step forms of the ragged batches (plan once into caller-owned buffers, emit many times) ------------
    def ragged_buffers(self, batch: int):
        """(num_bits int32[batch], row_off int64[batch+1], var_off int64[batch+1]) for the *_plan calls"""
        return (torch.empty((batch,), dtype=torch.int32, device=self.device),
                torch.empty((batch + 1,), dtype=torch.int64, device=self.device),
                torch.empty((batch + 1,), dtype=torch.int64, device=self.device))

    def max_bound_ragged_plan(self, max_range: torch.Tensor, num_bits, row_off, var_off) -> Layout:
        lay = _lib.LayoutC()
        st = self._lib.pg_max_bound_ragged_plan(self._h, max_range.data_ptr(), max_range.shape[0], num_bits.data_ptr(),
                                                row_off.data_ptr(), var_off.data_ptr(), C.byref(lay), self._stream())
        if st != 0:
            raise PgError(st, "pg_max_bound_ragged_plan")
        return self._layout(lay)

    def max_bound_ragged_emit(self, max_range, witness, num_bits, row_off, var_off, out: Columns, result_vars=None,
                              gate_base: int = 0, var_base: int = 0):
        cols = out.as_c()
        st = self._lib.pg_max_bound_ragged_batch(self._h, max_range.data_ptr(), witness.data_ptr(), witness.shape[0],
                                                 num_bits.data_ptr(), row_off.data_ptr(), var_off.data_ptr(), gate_base,
                                                 var_base, C.byref(cols),
                                                 result_vars.data_ptr() if result_vars is not None else None, self._stream())
        if st != 0:
            raise PgError(st, "pg_max_bound_ragged_batch")

    def scalar_mix_plan(self, v: torch.Tensor, row_off, var_off, err_mask=None):
        """-> (Layout, err_count); PG_ERR_NON_EXISTING_INVERSE is reported through err_count, not raised"""
        lay, nerr = _lib.LayoutC(), C.c_uint64()
        st = self._lib.pg_scalar_mix_plan(self._h, v.data_ptr(), v.shape[0], row_off.data_ptr(), var_off.data_ptr(),
                                          err_mask.data_ptr() if err_mask is not None else None, C.byref(lay),
                                          C.byref(nerr), self._stream())
        if st not in (0, 1):
            raise PgError(st, "pg_scalar_mix_plan")
        return self._layout(lay), int(nerr.value)

    def scalar_mix_emit(self, v, y, s, a, b, row_off, var_off, out: Columns, result_vars=None, gate_base: int = 0,
                        var_base: int = 0, zero_var: int = 0):
        cols = out.as_c()
        st = self._lib.pg_scalar_mix_batch(self._h, v.data_ptr(), y.data_ptr(), s.data_ptr(), a.data_ptr(), b.data_ptr(),
                                           v.shape[0], row_off.data_ptr(), var_off.data_ptr(), gate_base, var_base, zero_var,
                                           C.byref(cols), result_vars.data_ptr() if result_vars is not None else None,
                                           self._stream())
        if st != 0:
            raise PgError(st, "pg_scalar_mix_batch")

    def scalar_mix_planned(self, v, y, s, a, b, row_off, var_off, out: Columns, result_vars=None, err_mask=None,
                           gate_base: int = 0, var_base: int = 0, zero_var: int = 0):
        """plan + emit in one call (pg_scalar_mix_planned_batch): `out` must hold the worst case, 10 rows and 15 variables
        per item; the totals are read with plan_result() after a synchronisation"""
        cols = out.as_c()
        st = self._lib.pg_scalar_mix_planned_batch(self._h, v.data_ptr(), y.data_ptr(), s.data_ptr(), a.data_ptr(), b.data_ptr(),
                                                   v.shape[0], row_off.data_ptr(), var_off.data_ptr(),
                                                   err_mask.data_ptr() if err_mask is not None else None, gate_base, var_base,
                                                   zero_var, C.byref(cols),
                                                   result_vars.data_ptr() if result_vars is not None else None, self._stream())
        if st != 0:
            raise PgError(st, "pg_scalar_mix_planned_batch")

    # ---- witness refresh: the variable assignments of a call, no rows (pg_*_values_batch) ----------------------------------
    def range_check_values_batch(self, min_range: BlsScalar, max_range: BlsScalar, witness: torch.Tensor,
                                 var_values: torch.Tensor | None = None) -> torch.Tensor:
        """what range_check_batch writes into Columns.var_values, and nothing else: the same circuit rebuilt with other
        witnesses (prover.clear_witness() and the calls again, /root/reference/tests/scalar_gadgets_tests.rs:108-119)"""
        self._check_scalars(witness)
        batch = witness.shape[0]
        lay = self.range_check_layout(min_range, max_range, batch)
        if var_values is None:
            var_values = torch.empty((lay.n_vars, 4), dtype=torch.int64, device=self.device)
        assert var_values.is_contiguous() and var_values.shape == (lay.n_vars, 4)
        st = self._lib.pg_range_check_values_batch(self._h, C.byref(min_range.c), C.byref(max_range.c), witness.data_ptr(), batch,
                                                   var_values.data_ptr(), self._stream())
        if st != 0:
            raise PgError(st, "pg_range_check_values_batch")
        return var_values

    def max_bound_values_batch(self, max_range: BlsScalar, witness: torch.Tensor, var_values: torch.Tensor | None = None) -> torch.Tensor:
        self._check_scalars(witness)
        batch = witness.shape[0]
        lay = self.max_bound_layout(max_range, batch)
        if var_values is None:
            var_values = torch.empty((lay.n_vars, 4), dtype=torch.int64, device=self.device)
        assert var_values.is_contiguous() and var_values.shape == (lay.n_vars, 4)
        st = self._lib.pg_max_bound_values_batch(self._h, C.byref(max_range.c), witness.data_ptr(), batch, var_values.data_ptr(),
                                                 self._stream())
        if st != 0:
            raise PgError(st, "pg_max_bound_values_batch")
        return var_values

    def max_bound_ragged_values(self, max_range, witness, num_bits, row_off, var_off, var_values: torch.Tensor):
        """the ragged call's assignments under the plan of its (public) bounds"""
        st = self._lib.pg_max_bound_ragged_values_batch(self._h, max_range.data_ptr(), witness.data_ptr(), witness.shape[0],
                                                        num_bits.data_ptr(), row_off.data_ptr(), var_off.data_ptr(),
                                                        var_values.data_ptr(), self._stream())
        if st != 0:
            raise PgError(st, "pg_max_bound_ragged_values_batch")
        return var_values

    def scalar_mix_values(self, v, y, s, a, b, row_off, var_off, var_values: torch.Tensor, err_mask=None):
        """the fused mix's assignments AND its plan for these witnesses (row_off / var_off / err_mask are outputs; totals:
        plan_result()): an item's shape depends on its witness, the caller compares the plan with the circuit it has"""
        st = self._lib.pg_scalar_mix_values_batch(self._h, v.data_ptr(), y.data_ptr(), s.data_ptr(), a.data_ptr(), b.data_ptr(),
                                                  v.shape[0], row_off.data_ptr(), var_off.data_ptr(),
                                                  err_mask.data_ptr() if err_mask is not None else None, var_values.data_ptr(),
                                                  self._stream())
        if st != 0:
            raise PgError(st, "pg_scalar_mix_values_batch")
        return var_values

    # ---- asynchronous plans (no host round trip between plan and emit; totals read back later) -----------------
    def max_bound_ragged_plan_async(self, max_range: torch.Tensor, num_bits, row_off, var_off):
        st = self._lib.pg_max_bound_ragged_plan_async(self._h, max_range.data_ptr(), max_range.shape[0], num_bits.data_ptr(),
                                                      row_off.data_ptr(), var_off.data_ptr(), self._stream())
        if st != 0:
            raise PgError(st, "pg_max_bound_ragged_plan_async")

    def scalar_mix_plan_async(self, v: torch.Tensor, row_off, var_off, err_mask=None):
        st = self._lib.pg_scalar_mix_plan_async(self._h, v.data_ptr(), v.shape[0], row_off.data_ptr(), var_off.data_ptr(),
                                                err_mask.data_ptr() if err_mask is not None else None, self._stream())
        if st != 0:
            raise PgError(st, "pg_scalar_mix_plan_async")

    def plan_result(self):
        """(Layout, err_count) of the most recent plan; call after synchronising the stream it ran on"""
        lay, nerr = _lib.LayoutC(), C.c_uint64()
        st = self._lib.pg_plan_result(self._h, C.byref(lay), C.byref(nerr))
        if st not in (0, 1):
            raise PgError(st, "pg_plan_result")
        return self._layout(lay), int(nerr.value)

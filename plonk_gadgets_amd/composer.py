"""Host mirror of the reference's public interface on a device-resident composer.

Same names, argument order and error behaviour as /root/reference/src/lib.rs:42-45 re-exports, so circuits written
against the reference read the same:

    reference (tests/range_gadgets_tests.rs:36-43)                 here
    let w = AllocatedScalar::allocate(composer, witness);          w = AllocatedScalar.allocate(composer, witness)
    let r = range_check(composer, min, max, w);                    r = range_check(composer, mn, mx, w)
    composer.constrain_to_constant(r, outcome, None);              composer.constrain_to_constant(r, outcome, None)

Every call appends on the GPU through the C ABI (pg_composer_*, pg_range_check, ...); nothing is computed here.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from . import _lib
from .engine import Columns, Engine, NonExistingInverse, PgError
from .scalar import BlsScalar

Variable = int


def _chk(st: int, where: str):
    if st == 1:
        raise NonExistingInverse(st, where)
    if st != 0:
        raise PgError(st, where)


def _opt(pi: BlsScalar | None):
    return C.byref(pi.c) if pi is not None else None


class StandardComposer:
    """pg_composer: dusk-plonk's StandardComposer slice used by the gadgets, columns resident in HBM."""

    def __init__(self, engine: Engine, gate_capacity: int = 1 << 16, var_capacity: int = 1 << 16, with_dummy: bool = True):
        self.engine = engine
        self._lib = engine._lib
        h = C.c_void_p()
        _chk(self._lib.pg_composer_create(engine._h, gate_capacity, var_capacity, int(with_dummy), engine._stream(),
                                          C.byref(h)), "pg_composer_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pg_composer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- capacity ---------------------------------------------------------------------------
    def reserve(self, gate_capacity: int, var_capacity: int):
        """room for at least that many rows / Variables in total (re-allocates and copies the live part)"""
        _chk(self._lib.pg_composer_reserve(self._h, gate_capacity, var_capacity), "pg_composer_reserve")

    def auto_grow(self, on: bool = True):
        """like the reference's Vecs: an append that does not fit doubles the capacity instead of failing"""
        _chk(self._lib.pg_composer_auto_grow(self._h, int(on)), "pg_composer_auto_grow")

    def spread_columns(self, gib: float):
        """the composer's nine arrays in ONE allocation, the selector columns `gib` GiB apart (0: nine allocations again); the
        live part is copied.  For a composer of a few GB on a card with room to spare (pg_composer_spread_columns)"""
        _chk(self._lib.pg_composer_spread_columns(self._h, int(gib * (1 << 30))), "pg_composer_spread_columns")

    def capacity(self) -> tuple:
        return int(self._lib.pg_composer_gate_capacity(self._h)), int(self._lib.pg_composer_var_capacity(self._h))

    # -- the command queue -----------------------------------------------------------------
    def queue(self, on: bool = True):
        """single calls are recorded and flushed as few launches (default) / one launch per call"""
        _chk(self._lib.pg_composer_queue(self._h, int(on)), "pg_composer_queue")

    def sync(self):
        """flush what is recorded and wait for the composer's stream"""
        _chk(self._lib.pg_composer_sync(self._h), "pg_composer_sync")

    def flush(self):
        _chk(self._lib.pg_composer_flush(self._h), "pg_composer_flush")

    def queue_stats(self) -> tuple:
        """(entries waiting, flushes so far, launches those flushes took)"""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _chk(self._lib.pg_composer_queue_stats(self._h, C.byref(a), C.byref(b), C.byref(c)), "pg_composer_queue_stats")
        return int(a.value), int(b.value), int(c.value)

    # -- witness refresh ---------------------------------------------------------------------
    def clear_witness(self):
        """prover.clear_witness() of the reference's tests (tests/scalar_gadgets_tests.rs:110): the composer counts from
        StandardComposer::new()'s state again; the same calls with other witnesses then find their rows in place and
        write only assignments (pg_composer_clear_witness)"""
        _chk(self._lib.pg_composer_clear_witness(self._h), "pg_composer_clear_witness")

    def refresh_stats(self) -> tuple:
        """(rows found in place, rows written again, still matching the previous build) since the last clear_witness"""
        a, b, r = C.c_uint64(), C.c_uint64(), C.c_int()
        _chk(self._lib.pg_composer_refresh_stats(self._h, C.byref(a), C.byref(b), C.byref(r)), "pg_composer_refresh_stats")
        return int(a.value), int(b.value), bool(r.value)

    # -- state ------------------------------------------------------------------------------
    def circuit_size(self) -> int:
        return int(self._lib.pg_composer_circuit_size(self._h))

    def num_variables(self) -> int:
        return int(self._lib.pg_composer_num_variables(self._h))

    @property
    def zero_var(self) -> Variable:
        return int(self._lib.pg_composer_zero_var(self._h))

    # -- composer calls (dusk-plonk 0.8 argument order) ---------------------------------------------
    def add_input(self, s: BlsScalar) -> Variable:
        out = C.c_uint64()
        _chk(self._lib.pg_composer_add_input(self._h, C.byref(s.c), C.byref(out)), "add_input")
        return out.value

    def add_witness_to_circuit_description(self, value: BlsScalar) -> Variable:
        out = C.c_uint64()
        _chk(self._lib.pg_composer_add_witness_to_circuit_description(self._h, C.byref(value.c), C.byref(out)),
             "add_witness_to_circuit_description")
        return out.value

    def constrain_to_constant(self, a: Variable, constant: BlsScalar, pi: BlsScalar | None = None):
        _chk(self._lib.pg_composer_constrain_to_constant(self._h, a, C.byref(constant.c), _opt(pi)), "constrain_to_constant")

    def assert_equal(self, a: Variable, b: Variable):
        _chk(self._lib.pg_composer_assert_equal(self._h, a, b), "assert_equal")

    def poly_gate(self, a, b, c, q_m, q_l, q_r, q_o, q_c, pi=None):
        _chk(self._lib.pg_composer_poly_gate(self._h, a, b, c, C.byref(q_m.c), C.byref(q_l.c), C.byref(q_r.c),
                                             C.byref(q_o.c), C.byref(q_c.c), _opt(pi)), "poly_gate")

    def add(self, q_l_a, q_r_b, q_c: BlsScalar, pi=None) -> Variable:
        (q_l, a), (q_r, b) = q_l_a, q_r_b
        out = C.c_uint64()
        _chk(self._lib.pg_composer_add(self._h, C.byref(q_l.c), a, C.byref(q_r.c), b, C.byref(q_c.c), _opt(pi),
                                       C.byref(out)), "add")
        return out.value

    def mul(self, q_m: BlsScalar, a: Variable, b: Variable, q_c: BlsScalar, pi=None) -> Variable:
        out = C.c_uint64()
        _chk(self._lib.pg_composer_mul(self._h, C.byref(q_m.c), a, b, C.byref(q_c.c), _opt(pi), C.byref(out)), "mul")
        return out.value

    def mul_gate(self, a, b, c, q_m, q_o, q_c, pi=None):
        _chk(self._lib.pg_composer_mul_gate(self._h, a, b, c, C.byref(q_m.c), C.byref(q_o.c), C.byref(q_c.c), _opt(pi)),
             "mul_gate")

    def boolean_gate(self, a: Variable) -> Variable:
        _chk(self._lib.pg_composer_boolean_gate(self._h, a), "boolean_gate")
        return a

    # -- batched append -------------------------------------------------------------------------------
    def range_check_batch(self, min_range: BlsScalar, max_range: BlsScalar, witness: torch.Tensor) -> torch.Tensor:
        """for w in witness: allocate(w); range_check(min, max, w) -- appended at the composer's end"""
        assert witness.is_cuda and witness.dtype == torch.int64 and witness.dim() == 2 and witness.is_contiguous()
        res = torch.empty((witness.shape[0],), dtype=torch.int64, device=witness.device)
        _chk(self._lib.pg_composer_range_check_batch(self._h, C.byref(min_range.c), C.byref(max_range.c),
                                                     witness.data_ptr(), witness.shape[0], res.data_ptr()),
             "pg_composer_range_check_batch")
        return res

    def add_input_batch(self, scalars: torch.Tensor) -> int:
        """for s in scalars: add_input(s) -- returns the first Variable (the others follow it)"""
        assert scalars.is_cuda and scalars.dtype == torch.int64 and scalars.dim() == 2 and scalars.is_contiguous()
        first = C.c_uint64()
        _chk(self._lib.pg_composer_add_input_batch(self._h, scalars.data_ptr(), scalars.shape[0], C.byref(first)),
             "pg_composer_add_input_batch")
        return int(first.value)

    def range_check_allocated_batch(self, min_range: BlsScalar, max_range: BlsScalar, witness_vars: torch.Tensor,
                                    witness: torch.Tensor) -> torch.Tensor:
        """for i: range_check(min, max, AllocatedScalar(witness_vars[i], witness[i])) on witnesses allocated before"""
        assert witness.is_cuda and witness.dtype == torch.int64 and witness.dim() == 2 and witness.is_contiguous()
        assert witness_vars.is_cuda and witness_vars.dtype == torch.int64 and witness_vars.shape == (witness.shape[0],)
        res = torch.empty((witness.shape[0],), dtype=torch.int64, device=witness.device)
        _chk(self._lib.pg_composer_range_check_allocated_batch(self._h, C.byref(min_range.c), C.byref(max_range.c),
                                                               witness_vars.data_ptr(), witness.data_ptr(), witness.shape[0],
                                                               res.data_ptr()), "pg_composer_range_check_allocated_batch")
        return res

    def max_bound_batch(self, max_range: BlsScalar, witness: torch.Tensor):
        """for w in witness: allocate(w); max_bound(max_range, w) -> (result Variables, num_bits)"""
        assert witness.is_cuda and witness.dtype == torch.int64 and witness.dim() == 2 and witness.is_contiguous()
        res = torch.empty((witness.shape[0],), dtype=torch.int64, device=witness.device)
        nb = C.c_uint64()
        _chk(self._lib.pg_composer_max_bound_batch(self._h, C.byref(max_range.c), witness.data_ptr(), witness.shape[0],
                                                   res.data_ptr(), C.byref(nb)), "pg_composer_max_bound_batch")
        return res, int(nb.value)

    def max_bound_allocated_batch(self, max_range: BlsScalar, witness_vars: torch.Tensor, witness: torch.Tensor):
        assert witness.is_cuda and witness.dtype == torch.int64 and witness.dim() == 2 and witness.is_contiguous()
        assert witness_vars.is_cuda and witness_vars.dtype == torch.int64 and witness_vars.shape == (witness.shape[0],)
        res = torch.empty((witness.shape[0],), dtype=torch.int64, device=witness.device)
        nb = C.c_uint64()
        _chk(self._lib.pg_composer_max_bound_allocated_batch(self._h, C.byref(max_range.c), witness_vars.data_ptr(),
                                                             witness.data_ptr(), witness.shape[0], res.data_ptr(), C.byref(nb)),
             "pg_composer_max_bound_allocated_batch")
        return res, int(nb.value)

    def scalar_decomposition_batch(self, num_bits: int, witness_vars: torch.Tensor, witness: torch.Tensor) -> torch.Tensor:
        """for i: scalar_decomposition_gadget(num_bits, AllocatedScalar(witness_vars[i], witness[i])) -> is_equal Variables"""
        assert witness.is_cuda and witness.dtype == torch.int64 and witness.dim() == 2 and witness.is_contiguous()
        assert witness_vars.is_cuda and witness_vars.dtype == torch.int64 and witness_vars.shape == (witness.shape[0],)
        res = torch.empty((witness.shape[0],), dtype=torch.int64, device=witness.device)
        _chk(self._lib.pg_composer_scalar_decomposition_batch(self._h, num_bits, witness_vars.data_ptr(), witness.data_ptr(),
                                                              witness.shape[0], res.data_ptr()),
             "pg_composer_scalar_decomposition_batch")
        return res

    def _two_input_batch(self, fn: str, a_vars: torch.Tensor, b_vars: torch.Tensor) -> torch.Tensor:
        assert a_vars.is_cuda and a_vars.dtype == torch.int64 and a_vars.dim() == 1 and a_vars.is_contiguous()
        assert b_vars.is_cuda and b_vars.dtype == torch.int64 and b_vars.shape == a_vars.shape and b_vars.is_contiguous()
        res = torch.empty_like(a_vars)
        _chk(getattr(self._lib, fn)(self._h, a_vars.data_ptr(), b_vars.data_ptr(), a_vars.shape[0], res.data_ptr()), fn)
        return res

    def conditionally_select_zero_batch(self, x_vars: torch.Tensor, select_vars: torch.Tensor) -> torch.Tensor:
        return self._two_input_batch("pg_composer_conditionally_select_zero_batch", x_vars, select_vars)

    def conditionally_select_one_batch(self, y_vars: torch.Tensor, selector_vars: torch.Tensor) -> torch.Tensor:
        return self._two_input_batch("pg_composer_conditionally_select_one_batch", y_vars, selector_vars)

    def maybe_equal_batch(self, a_vars: torch.Tensor, b_vars: torch.Tensor) -> torch.Tensor:
        return self._two_input_batch("pg_composer_maybe_equal_batch", a_vars, b_vars)

    def max_bound_ragged_batch(self, max_range: torch.Tensor, witness: torch.Tensor):
        """for i: allocate(witness[i]); max_bound(max_range[i], w) -> (result Variables, num_bits per item)"""
        for x in (max_range, witness):
            assert x.is_cuda and x.dtype == torch.int64 and x.dim() == 2 and x.is_contiguous()
        assert max_range.shape == witness.shape
        res = torch.empty((witness.shape[0],), dtype=torch.int64, device=witness.device)
        nb = torch.empty((witness.shape[0],), dtype=torch.int32, device=witness.device)
        _chk(self._lib.pg_composer_max_bound_ragged_batch(self._h, max_range.data_ptr(), witness.data_ptr(), witness.shape[0],
                                                          res.data_ptr(), nb.data_ptr()), "pg_composer_max_bound_ragged_batch")
        return res, nb

    def is_non_zero_batch(self, vars_: torch.Tensor):
        """for i: is_non_zero(vars[i], its value) -> (error mask, error count); items whose value is 0 stop early and are
        reported here instead of raising (the reference returns Err(NonExistingInverse) for them)"""
        assert vars_.is_cuda and vars_.dtype == torch.int64 and vars_.dim() == 1 and vars_.is_contiguous()
        err = torch.zeros((vars_.shape[0],), dtype=torch.uint8, device=vars_.device)
        nerr = C.c_uint64()
        st = self._lib.pg_composer_is_non_zero_batch(self._h, vars_.data_ptr(), vars_.shape[0], err.data_ptr(), C.byref(nerr))
        if st not in (0, 1):
            _chk(st, "pg_composer_is_non_zero_batch")
        return err, int(nerr.value)

    def scalar_mix_batch(self, v, y, s, a, b):
        """the fused item of BASELINE config 3 -> (result Variables [batch, 2], error mask, error count)"""
        for x in (v, y, s, a, b):
            assert x.is_cuda and x.dtype == torch.int64 and x.dim() == 2 and x.is_contiguous() and x.shape == v.shape
        res = torch.empty((v.shape[0], 2), dtype=torch.int64, device=v.device)
        err = torch.zeros((v.shape[0],), dtype=torch.uint8, device=v.device)
        nerr = C.c_uint64()
        st = self._lib.pg_composer_scalar_mix_batch(self._h, v.data_ptr(), y.data_ptr(), s.data_ptr(), a.data_ptr(), b.data_ptr(),
                                                    v.shape[0], res.data_ptr(), err.data_ptr(), C.byref(nerr))
        if st not in (0, 1):
            _chk(st, "pg_composer_scalar_mix_batch")
        return res, err, int(nerr.value)

    # -- the gate calls over arrays of Variables (one set of selectors per batch, no public inputs) --------------------
    def _vars(self, *ts):
        for t in ts:
            assert t.is_cuda and t.dtype == torch.int64 and t.dim() == 1 and t.is_contiguous() and t.shape == ts[0].shape
        return ts[0].shape[0]

    def poly_gate_batch(self, a, b, c, q_m, q_l, q_r, q_o, q_c):
        n = self._vars(a, b, c)
        _chk(self._lib.pg_composer_poly_gate_batch(self._h, a.data_ptr(), b.data_ptr(), c.data_ptr(), C.byref(q_m.c), C.byref(q_l.c),
                                                   C.byref(q_r.c), C.byref(q_o.c), C.byref(q_c.c), n), "pg_composer_poly_gate_batch")

    def add_batch(self, q_l: BlsScalar, a, q_r: BlsScalar, b, q_c: BlsScalar) -> torch.Tensor:
        n = self._vars(a, b)
        out = torch.empty_like(a)
        _chk(self._lib.pg_composer_add_batch(self._h, C.byref(q_l.c), a.data_ptr(), C.byref(q_r.c), b.data_ptr(), C.byref(q_c.c), n,
                                             out.data_ptr()), "pg_composer_add_batch")
        return out

    def mul_batch(self, q_m: BlsScalar, a, b, q_c: BlsScalar) -> torch.Tensor:
        n = self._vars(a, b)
        out = torch.empty_like(a)
        _chk(self._lib.pg_composer_mul_batch(self._h, C.byref(q_m.c), a.data_ptr(), b.data_ptr(), C.byref(q_c.c), n, out.data_ptr()),
             "pg_composer_mul_batch")
        return out

    def constrain_to_constant_batch(self, a, constant: BlsScalar):
        _chk(self._lib.pg_composer_constrain_to_constant_batch(self._h, a.data_ptr(), C.byref(constant.c), self._vars(a)),
             "pg_composer_constrain_to_constant_batch")

    def boolean_gate_batch(self, a):
        _chk(self._lib.pg_composer_boolean_gate_batch(self._h, a.data_ptr(), self._vars(a)), "pg_composer_boolean_gate_batch")

    # -- read-back ----------------------------------------------------------------------------------------
    def value(self, v: Variable) -> BlsScalar:
        out = _lib.Scalar()
        _chk(self._lib.pg_composer_read_value(self._h, v, C.byref(out)), "read_value")
        return BlsScalar(out)

    def check(self) -> int:
        """-1 when every row is satisfied, else the first failing row"""
        bad = C.c_int64()
        _chk(self._lib.pg_composer_check(self._h, C.byref(bad)), "check")
        return bad.value

    def construct_dense_pi_vec(self) -> torch.Tensor:
        out = torch.empty((self.circuit_size(), 4), dtype=torch.int64, device=self.engine.device)
        _chk(self._lib.pg_composer_dense_pi(self._h, out.data_ptr()), "dense_pi")
        return out

    def device_columns(self) -> Columns:
        """the composer's OWN columns as tensors (no copy): rows [0, circuit_size), Variables [0, num_variables); valid until
        the composer grows or is destroyed; what is recorded is flushed to the stream first (pg_composer_columns)"""
        cc = _lib.ColumnsC()
        _chk(self._lib.pg_composer_columns(self._h, C.byref(cc)), "pg_composer_columns")
        n, nv, dev = self.circuit_size(), self.num_variables(), self.engine.device

        class _Mem:
            def __init__(self, ptr, words):
                self.__cuda_array_interface__ = {"shape": (words,), "typestr": "<i8", "data": (int(ptr), False), "version": 2}

        def words(ptr, count):
            return torch.as_tensor(_Mem(ptr, count), device=dev)
        sc = [words(getattr(cc, k), n * 4).view(n, 4) for k in Columns.SCALAR_COLS]
        wc = [words(getattr(cc, k), n) for k in Columns.WIRE_COLS]
        return Columns(*sc, *wc, words(cc.var_values, nv * 4).view(nv, 4))

    def export(self, gate_base: int = 0, var_base: int = 0) -> dict:
        """numpy copy of the live columns (rows >= gate_base, variables >= var_base)"""
        n, nv = self.circuit_size() - gate_base, self.num_variables() - var_base
        cols = Columns.allocate(n, nv, self.engine.device, gate_base, var_base)
        cc = cols.as_c()
        _chk(self._lib.pg_composer_copy_out(self._h, gate_base, n, var_base, nv, C.byref(cc)), "copy_out")
        _chk(self._lib.pg_composer_sync(self._h), "sync")
        return cols.to_numpy()

    def permutation_reserve(self, sparse_positions: int):
        """first-pass size of the permutation's sorted list of 'foreign' wire positions (0: the composer's estimate)"""
        _chk(self._lib.pg_composer_permutation_reserve(self._h, sparse_positions), "pg_composer_permutation_reserve")

    def permutation(self, padded_n: int | None = None) -> torch.Tensor:
        """sigma as int64[4, padded_n]; entry [w, i] = w' * padded_n + i' (next position of the Variable)"""
        n = self.circuit_size()
        padded_n = padded_n or n
        out = torch.empty((4, padded_n), dtype=torch.int64, device=self.engine.device)
        _chk(self._lib.pg_composer_permutation(self._h, padded_n, out.data_ptr()), "permutation")
        return out

    def materialize(self) -> dict:
        """constant columns, w_4 and the wire-value columns as device tensors"""
        n, dev = self.circuit_size(), self.engine.device
        names = ("q_4", "q_arith", "q_range", "q_logic", "q_fixed_group_add", "q_variable_group_add")
        vals = ("w_l_value", "w_r_value", "w_o_value", "w_4_value")
        t = {k: torch.empty((n, 4), dtype=torch.int64, device=dev) for k in names + vals}
        t["w_4"] = torch.empty((n,), dtype=torch.int64, device=dev)
        fc = _lib.FullColumnsC(**{k: v.data_ptr() for k, v in t.items()})
        _chk(self._lib.pg_composer_materialize(self._h, C.byref(fc)), "materialize")
        return t


@dataclass
class AllocatedScalar:
    """/root/reference/src/allocated_scalar.rs:17-30"""
    var: Variable
    scalar: BlsScalar

    @staticmethod
    def allocate(composer: StandardComposer, scalar: BlsScalar) -> "AllocatedScalar":
        out = _lib.AllocatedScalarC()
        _chk(composer._lib.pg_allocated_scalar_allocate(composer._h, C.byref(scalar.c), C.byref(out)), "allocate")
        return AllocatedScalar(int(out.var), scalar)

    def _c(self) -> _lib.AllocatedScalarC:
        return _lib.AllocatedScalarC(self.var, self.scalar.c)


# ---- RangeGadgets (/root/reference/src/range.rs) ---------------------------------------------------------
def range_check(composer: StandardComposer, min_range: BlsScalar, max_range: BlsScalar, witness: AllocatedScalar) -> Variable:
    """src/range.rs:27-43"""
    out, w = C.c_uint64(), witness._c()
    _chk(composer._lib.pg_range_check(composer._h, C.byref(min_range.c), C.byref(max_range.c), C.byref(w), C.byref(out)),
         "range_check")
    return out.value


def max_bound(composer: StandardComposer, max_range: BlsScalar, witness: AllocatedScalar):
    """src/range.rs:82-113 -> (Variable, num_bits)"""
    out, nb, w = C.c_uint64(), C.c_uint64(), witness._c()
    _chk(composer._lib.pg_max_bound(composer._h, C.byref(max_range.c), C.byref(w), C.byref(out), C.byref(nb)), "max_bound")
    return out.value, nb.value


def scalar_decomposition_gadget(composer: StandardComposer, num_bits: int, witness: AllocatedScalar):
    """src/range.rs:119-123 (private in the reference; its unit test at :205-233 calls it) -> (is_equal, bit Variables)"""
    out = C.c_uint64()
    bits = (C.c_uint64 * max(num_bits, 1))()
    w = witness._c()
    _chk(composer._lib.pg_scalar_decomposition_gadget(composer._h, num_bits, C.byref(w), C.byref(out), bits),
         "scalar_decomposition_gadget")
    return out.value, [int(bits[i]) for i in range(num_bits)]


# ---- ScalarGadgets (/root/reference/src/scalar.rs) --------------------------------------------------------
def conditionally_select_zero(composer: StandardComposer, x: Variable, select: Variable) -> Variable:
    """src/scalar.rs:21-27"""
    out = C.c_uint64()
    _chk(composer._lib.pg_conditionally_select_zero(composer._h, x, select, C.byref(out)), "conditionally_select_zero")
    return out.value


def conditionally_select_one(composer: StandardComposer, y: Variable, selector: Variable) -> Variable:
    """src/scalar.rs:36-59"""
    out = C.c_uint64()
    _chk(composer._lib.pg_conditionally_select_one(composer._h, y, selector, C.byref(out)), "conditionally_select_one")
    return out.value


def is_non_zero(composer: StandardComposer, var: Variable, value_assigned: BlsScalar) -> None:
    """src/scalar.rs:63-97: raises NonExistingInverse (after the partial emission) when value_assigned is zero"""
    _chk(composer._lib.pg_is_non_zero(composer._h, var, C.byref(value_assigned.c)), "is_non_zero")


def maybe_equal(composer: StandardComposer, a: AllocatedScalar, b: AllocatedScalar) -> Variable:
    """src/scalar.rs:105-140"""
    out, ca, cb = C.c_uint64(), a._c(), b._c()
    _chk(composer._lib.pg_maybe_equal(composer._h, C.byref(ca), C.byref(cb), C.byref(out)), "maybe_equal")
    return out.value

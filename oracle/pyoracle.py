"""oracle/pyoracle.py -- TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/liboracle.so.

Scalars cross this boundary as numpy uint64 arrays of shape [..., 4]
(Montgomery limbs, little-endian), the same memory image the HIP path emits.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


class Fr(C.Structure):
    _fields_ = [("l", C.c_uint64 * 4)]


class AllocatedScalar(C.Structure):
    _fields_ = [("var", C.c_uint64), ("scalar", Fr)]


class Columns(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o", "var_values")]


FULL_SCALAR_COLS = ("q_4", "q_arith", "q_range", "q_logic", "q_fixed_group_add", "q_variable_group_add")
FULL_VALUE_COLS = ("w_l_value", "w_r_value", "w_o_value", "w_4_value")


class FullColumns(C.Structure):
    """oracle_full_columns_t: what pg_composer_materialize produces beyond the eight live columns"""
    _fields_ = [(n, C.c_void_p) for n in FULL_SCALAR_COLS + ("w_4",) + FULL_VALUE_COLS]


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("fr.c", "composer.c", "gadgets.c", "fast.c", "fr.h", "composer.h", "gadgets.h")]
    stale = force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    u64, vp, sz = C.c_uint64, C.c_void_p, C.c_size_t
    P = C.POINTER
    sig = {
        "fr_add": (Fr, [Fr, Fr]), "fr_sub": (Fr, [Fr, Fr]), "fr_neg": (Fr, [Fr]), "fr_mul": (Fr, [Fr, Fr]),
        "fr_square": (Fr, [Fr]), "fr_from_u64": (Fr, [u64]), "fr_from_raw": (Fr, [P(u64)]), "fr_reduce": (Fr, [Fr]),
        "fr_to_bytes": (None, [Fr, P(C.c_uint8)]), "fr_pow": (Fr, [Fr, P(u64)]), "fr_pow_of_2": (Fr, [u64]),
        "fr_invert": (C.c_int, [Fr, P(Fr)]),
        "composer_new": (vp, []), "composer_new_without_dummy": (vp, []), "composer_free": (None, [vp]),
        "composer_circuit_size": (sz, [vp]), "composer_num_variables": (sz, [vp]), "composer_zero_var": (u64, [vp]),
        "composer_add_input": (u64, [vp, Fr]), "composer_add_witness_to_circuit_description": (u64, [vp, Fr]),
        "composer_constrain_to_constant": (None, [vp, u64, Fr, P(Fr)]), "composer_assert_equal": (None, [vp, u64, u64]),
        "composer_poly_gate": (None, [vp, u64, u64, u64, Fr, Fr, Fr, Fr, Fr, P(Fr)]),
        "composer_add": (u64, [vp, Fr, u64, Fr, u64, Fr, P(Fr)]), "composer_mul": (u64, [vp, Fr, u64, u64, Fr, P(Fr)]),
        "composer_mul_gate": (None, [vp, u64, u64, u64, Fr, Fr, Fr, P(Fr)]), "composer_boolean_gate": (u64, [vp, u64]),
        "composer_value": (Fr, [vp, u64]), "composer_selector": (P(Fr), [vp, C.c_int]),
        "composer_wire": (P(u64), [vp, C.c_int]), "composer_values_dense": (None, [vp, vp]),
        "composer_perm_count": (sz, [vp, u64]), "composer_dense_pi": (None, [vp, vp]), "composer_check": (C.c_long, [vp]),
        "composer_sigma": (None, [vp, sz, vp]),
        "allocated_scalar_allocate": (AllocatedScalar, [vp, Fr]),
        "range_check": (u64, [vp, Fr, Fr, AllocatedScalar]), "max_bound": (u64, [vp, Fr, AllocatedScalar, P(u64)]),
        "min_bound": (u64, [vp, Fr, AllocatedScalar, u64]), "range_proof": (u64, [vp, AllocatedScalar, u64]),
        "scalar_decomposition_gadget": (u64, [vp, sz, AllocatedScalar, P(u64)]),
        "scalar_to_bits": (None, [Fr, P(C.c_uint8)]), "bits_count": (u64, [Fr]),
        "num_bits_closest_power_of_two": (u64, [Fr]),
        "conditionally_select_zero": (u64, [vp, u64, u64]), "conditionally_select_one": (u64, [vp, u64, u64]),
        "is_non_zero": (C.c_int, [vp, u64, Fr]), "maybe_equal": (u64, [vp, AllocatedScalar, AllocatedScalar]),
        "oracle_range_check_batch": (C.c_int, [Fr, Fr, vp, sz, C.c_int, P(Columns), vp, P(u64), P(u64), P(u64), P(u64)]),
        "oracle_range_check_fast": (C.c_int, [Fr, Fr, vp, sz, u64, C.c_int, P(Columns), vp]),
        "oracle_range_check_allocated_fast": (C.c_int, [Fr, Fr, vp, vp, sz, u64, C.c_int, P(Columns), vp]),
        "oracle_is_non_zero_plan": (C.c_int, [vp, vp, sz, vp, vp, vp]),
        "oracle_small_batch_fast": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, vp, vp, sz, sz, u64, u64, C.c_int, P(Columns), vp]),
        "oracle_sigma_fast_begin": (vp, [vp, vp, vp, vp, sz, sz, sz, C.c_int]),
        "oracle_sigma_fast_chunk": (C.c_int, [vp, sz, sz, P(vp)]), "oracle_sigma_fast_end": (None, [vp]),
        "oracle_materialize_fast": (C.c_int, [vp, vp, vp, vp, vp, sz, vp, vp, sz, sz, sz, C.c_int, P(FullColumns)]),
        "oracle_max_bound_plan": (C.c_int, [vp, sz, C.c_int, vp, vp, vp]),
        "oracle_max_bound_fast": (C.c_int, [vp, vp, vp, vp, vp, sz, sz, u64, C.c_int, P(Columns), vp]),
        "oracle_scalar_mix_plan": (C.c_int, [vp, sz, vp, vp, vp]),
        "oracle_scalar_mix_fast": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, sz, sz, u64, u64, C.c_int, P(Columns), vp]),
        "oracle_max_bound_batch": (C.c_int, [vp, vp, sz, C.c_int, P(Columns), vp, vp, P(u64), P(u64), P(u64), P(u64)]),
        "oracle_scalar_mix_batch": (C.c_int, [vp, vp, vp, vp, vp, sz, C.c_int, P(Columns), vp, vp, P(u64), P(u64),
                                            P(u64), P(u64)]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _lib = L
    return L


# ---- scalar helpers ---------------------------------------------------------

def fr(limbs) -> Fr:
    f = Fr()
    for i in range(4):
        f.l[i] = int(limbs[i])
    return f


def limbs(f: Fr) -> list[int]:
    return [int(f.l[i]) for i in range(4)]


def fr_from_int(x: int) -> Fr:
    """canonical integer -> Montgomery scalar, through the C code (fr_from_raw)."""
    from .model import Q
    x %= Q
    raw = (C.c_uint64 * 4)(*[(x >> (64 * i)) & ((1 << 64) - 1) for i in range(4)])
    return lib().fr_from_raw(raw)


def fr_to_int(f: Fr) -> int:
    r = lib().fr_reduce(f)
    return sum(int(r.l[i]) << (64 * i) for i in range(4))


def ints_to_mont_array(xs) -> np.ndarray:
    """list of canonical ints -> uint64[N,4] Montgomery limbs (via the big-int model: independent of fr.c)."""
    from .model import mont_limbs
    return np.array([mont_limbs(int(x)) for x in xs], dtype=np.uint64).reshape(-1, 4)


def _alloc_columns(n_gates: int, n_vars: int):
    arrs = {k: np.zeros((n_gates, 4), dtype=np.uint64) for k in ("q_m", "q_l", "q_r", "q_o", "q_c")}
    arrs.update({k: np.zeros(n_gates, dtype=np.uint64) for k in ("w_l", "w_r", "w_o")})
    arrs["var_values"] = np.zeros((n_vars, 4), dtype=np.uint64)
    cols = Columns(**{k: v.ctypes.data for k, v in arrs.items()})
    return arrs, cols


def _columns_into(out: dict | None, n_gates: int, n_vars: int):
    """the nine arrays of a call: fresh ones, or views of the caller's (e.g. pinned) buffers `out[name]`, which must be
    C-contiguous uint64 arrays at least that long -- the oracle writes straight into them"""
    if out is None:
        return _alloc_columns(n_gates, n_vars)
    arrs = {}
    sel = ("q_m", "q_l", "q_r", "q_o", "q_c")
    have_sel = [k in out for k in sel]
    assert all(have_sel) or not any(have_sel), "all five selector columns or none (wires and assignments only)"
    for k in (sel if all(have_sel) else ()) + ("var_values",):
        n = n_vars if k == "var_values" else n_gates
        a = out[k]
        assert a.dtype == np.uint64 and a.flags.c_contiguous and a.size >= 4 * n, k
        arrs[k] = a.reshape(-1)[:4 * n].reshape(n, 4)
    have_w = [k in out for k in ("w_l", "w_r", "w_o")]
    assert all(have_w) or not any(have_w), "all three wire columns or none (assignments only)"
    for k in ("w_l", "w_r", "w_o") if all(have_w) else ():
        a = out[k]
        assert a.dtype == np.uint64 and a.flags.c_contiguous and a.size >= n_gates, k
        arrs[k] = a.reshape(-1)[:n_gates]
    return arrs, Columns(**{k: v.ctypes.data for k, v in arrs.items()})  # (absent selector columns: NULL, not written)


def _as_fr_array(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.ndim == 2 and a.shape[1] == 4
    return a


def range_check_batch(min_mont, max_mont, witness: np.ndarray, check: bool = True, want_columns: bool = True):
    """allocate + range_check per witness on a fresh composer -> dict of columns (rows after the initial state)."""
    L = lib()
    witness = _as_fr_array(witness)
    batch = witness.shape[0]
    mn, mx = fr(min_mont), fr(max_mont)
    n = int(L.num_bits_closest_power_of_two(L.fr_sub(mx, L.fr_from_u64(1))))
    G, V = 4 * n + 11, 2 * n + 524
    res = np.zeros(batch, dtype=np.uint64)
    gb, vb, ng, nv = (C.c_uint64() for _ in range(4))
    arrs, cols = _alloc_columns(G * batch if want_columns else 0, V * batch if want_columns else 0)
    rc = L.oracle_range_check_batch(mn, mx, witness.ctypes.data, batch, int(check),
                                    C.byref(cols) if want_columns else None, res.ctypes.data, C.byref(gb),
                                    C.byref(vb), C.byref(ng), C.byref(nv))
    assert ng.value == G * batch and nv.value == V * batch, (ng.value, nv.value, G, V, batch)
    arrs.update(result_vars=res, gate_base=gb.value, var_base=vb.value, n_gates=ng.value, n_vars=nv.value,
                num_bits=n, satisfied=(rc == 0))
    return arrs


def range_check_fast(min_mont, max_mont, witness: np.ndarray, threads: int = 1, var_base: int = 5, timed_passes: int = 1,
                     out: dict | None = None):
    """oracle/fast.c: same columns as range_check_batch, flat arrays + mont(2^i) table + threads.  `seconds` is the LAST of
    1 + timed_passes... passes when timed_passes > 1: the first pass lets every thread touch the output pages it writes
    (first touch = placement on the thread's own NUMA node), the timed one then measures the loop, not page faults."""
    L = lib()
    witness = _as_fr_array(witness)
    batch = witness.shape[0]
    mn, mx = fr(min_mont), fr(max_mont)
    n = int(L.num_bits_closest_power_of_two(L.fr_sub(mx, L.fr_from_u64(1))))
    G, V = 4 * n + 11, 2 * n + 524
    arrs, cols = _columns_into(out, G * batch, V * batch)
    res = np.zeros(batch, dtype=np.uint64)
    import time
    if timed_passes <= 1 and out is None:
        for a in arrs.values():
            a.fill(0)  # touch the pages: first-touch faults are not the algorithm's time
    for _ in range(max(1, timed_passes)):
        t0 = time.perf_counter()
        rc = L.oracle_range_check_fast(mn, mx, witness.ctypes.data, batch, var_base, threads, C.byref(cols), res.ctypes.data)
        seconds = time.perf_counter() - t0
        assert rc == 0
    arrs.update(result_vars=res, n_gates=G * batch, n_vars=V * batch, num_bits=n, var_base=var_base, seconds=seconds)
    return arrs


def range_check_allocated_fast(min_mont, max_mont, witness: np.ndarray, witness_vars: np.ndarray, threads: int = 1,
                               var_base: int = 5, out: dict | None = None):
    """oracle/fast.c: `for i: range_check(min, max, AllocatedScalar(witness_vars[i], witness[i]))` on witnesses allocated before
    the loop (/root/reference/src/range.rs:27-32): 4n + 11 rows, 2n + 523 new Variables per item, from var_base"""
    L = lib()
    witness = _as_fr_array(witness)
    witness_vars = np.ascontiguousarray(witness_vars, dtype=np.uint64)
    batch = witness.shape[0]
    assert witness_vars.shape == (batch,)
    mn, mx = fr(min_mont), fr(max_mont)
    n = int(L.num_bits_closest_power_of_two(L.fr_sub(mx, L.fr_from_u64(1))))
    G, V = 4 * n + 11, 2 * n + 523
    arrs, cols = _columns_into(out, G * batch, V * batch)
    res = np.zeros(batch, dtype=np.uint64)
    rc = L.oracle_range_check_allocated_fast(mn, mx, witness.ctypes.data, witness_vars.ctypes.data, batch, var_base, threads,
                                             C.byref(cols), res.ctypes.data)
    assert rc == 0
    arrs.update(result_vars=res, n_gates=G * batch, n_vars=V * batch, num_bits=n, var_base=var_base)
    return arrs


SMALL_KINDS = {"select_zero": (0, 1, 1), "select_one": (1, 4, 4), "maybe_equal": (2, 3, 3), "is_non_zero": (3, 3, 3), "add": (4, 1, 1),
               "mul": (5, 1, 1), "rows": (6, 1, 0)}   # name -> (ORACLE_* kind, rows, Variables of a full item)


def is_non_zero_plan(vars_: np.ndarray, table: np.ndarray):
    """prefix sums of rows / Variables of `for v in vars: is_non_zero(v, value of v)` and its error mask (scalar.rs:73-80)"""
    vars_ = np.ascontiguousarray(vars_, dtype=np.uint64)
    table = _as_fr_array(table)
    batch = vars_.shape[0]
    roff, voff = np.zeros(batch + 1, dtype=np.uint64), np.zeros(batch + 1, dtype=np.uint64)
    err = np.zeros(batch, dtype=np.uint8)
    lib().oracle_is_non_zero_plan(vars_.ctypes.data, table.ctypes.data, batch, roff.ctypes.data, voff.ctypes.data, err.ctypes.data)
    return roff, voff, err


def small_batch_fast(kind: str, a, b, c, table: np.ndarray, lo: int, hi: int, var_base: int, zero_var: int = 0, selectors=None,
                     plan=None, threads: int = 1, out: dict | None = None):
    """oracle/fast.c: items [lo, hi) of `for i: gadget(a[i], b[i])` on existing Variables (kind: SMALL_KINDS; selectors: the five
    Montgomery scalars q_m, q_l, q_r, q_o, q_c of a gate batch; plan: is_non_zero_plan's, for is_non_zero)"""
    k, rows, nvars = SMALL_KINDS[kind]
    arr = [None if x is None else np.ascontiguousarray(x, dtype=np.uint64) for x in (a, b, c)]
    table = _as_fr_array(table)
    if plan is not None:
        roff, voff = plan[0], plan[1]
        G, V = int(roff[hi] - roff[lo]), int(voff[hi] - voff[lo])
    else:
        roff = voff = None
        G, V = rows * (hi - lo), nvars * (hi - lo)
    if out is not None and "var_values" not in out:
        out = dict(out, var_values=np.zeros((0, 4), dtype=np.uint64))
    arrs, cols = _columns_into(out, G, V)
    res = np.zeros(hi - lo, dtype=np.uint64)
    sel = None if selectors is None else np.ascontiguousarray(selectors, dtype=np.uint64).reshape(5, 4)
    rc = lib().oracle_small_batch_fast(k, *[None if x is None else x.ctypes.data for x in arr], table.ctypes.data,
                                       None if sel is None else sel.ctypes.data, None if roff is None else roff.ctypes.data,
                                       None if voff is None else voff.ctypes.data, lo, hi, var_base, zero_var, threads, C.byref(cols),
                                       res.ctypes.data)
    assert rc == 0, "oracle_small_batch_fast: bad arguments"
    arrs.update(result_vars=res, n_gates=G, n_vars=V)
    return arrs


class SigmaFast:
    """oracle/fast.c: sigma of a whole circuit from its four wire columns (uint64[n] each), chunk by chunk from the LAST row
    of the padded domain to the first.  == Composer.sigma(padded_n) (tests/test_oracle_fast.py)."""

    def __init__(self, w_l, w_r, w_o, w_4, padded_n: int, n_vars: int, threads: int = 1):
        self.L = lib()
        self.wires = [np.ascontiguousarray(w, dtype=np.uint64) for w in (w_l, w_r, w_o, w_4)]  # (kept alive)
        self.n = self.wires[0].shape[0]
        assert all(w.shape == (self.n,) for w in self.wires) and padded_n >= self.n
        self.padded_n = padded_n
        self.plan = self.L.oracle_sigma_fast_begin(*[w.ctypes.data for w in self.wires], self.n, padded_n, n_vars, threads)
        assert self.plan, "oracle_sigma_fast_begin"
        self.next_r1 = padded_n

    def chunk(self, r0: int, r1: int, out=None) -> np.ndarray:
        """sigma[:, r0:r1] as uint64[4, r1 - r0] (or into `out`: four C-contiguous uint64 arrays at least that long)"""
        assert r1 == self.next_r1 and 0 <= r0 < r1, (r0, r1, self.next_r1)
        m = r1 - r0
        if out is None:
            res = np.empty((4, m), dtype=np.uint64)
            out = [res[w] for w in range(4)]
        else:
            res = None
            for a in out:
                assert a.dtype == np.uint64 and a.flags.c_contiguous and a.size >= m
        ptrs = (C.c_void_p * 4)(*[a.ctypes.data for a in out])
        rc = self.L.oracle_sigma_fast_chunk(self.plan, r0, r1, ptrs)
        assert rc == 0, "oracle_sigma_fast_chunk: inconsistent wires or a chunk out of order"
        self.next_r1 = r0
        return res

    def whole(self, chunk_rows: int = 1 << 20) -> np.ndarray:
        res = np.empty((4, self.padded_n), dtype=np.uint64)
        r1 = self.padded_n
        while r1 > 0:
            r0 = max(0, r1 - chunk_rows)
            res[:, r0:r1] = self.chunk(r0, r1)
            r1 = r0
        return res

    def close(self):
        if self.plan:
            self.L.oracle_sigma_fast_end(self.plan)
            self.plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def materialize_fast(w_l, w_r, w_o, w_4, values: np.ndarray, r0: int, r1: int, q4: dict | None = None, threads: int = 1,
                     out: dict | None = None) -> dict:
    """oracle/fast.c: rows [r0, r1) of the eleven arrays pg_composer_materialize produces, from the wire columns and the dense
    assignment table; q4 = {row: limbs} lists the rows whose q_4 is not zero (composer_new()'s first dummy row)"""
    wires = [np.ascontiguousarray(w, dtype=np.uint64) for w in (w_l, w_r, w_o, w_4)]
    values = _as_fr_array(values)
    m = r1 - r0
    arrs = {}
    for k in FULL_SCALAR_COLS + FULL_VALUE_COLS:
        if out is None:
            arrs[k] = np.empty((m, 4), dtype=np.uint64)
        else:
            a = out[k]
            assert a.dtype == np.uint64 and a.flags.c_contiguous and a.size >= 4 * m, k
            arrs[k] = a.reshape(-1)[:4 * m].reshape(m, 4)
    if out is None:
        arrs["w_4"] = np.empty(m, dtype=np.uint64)
    else:
        assert out["w_4"].dtype == np.uint64 and out["w_4"].flags.c_contiguous and out["w_4"].size >= m
        arrs["w_4"] = out["w_4"].reshape(-1)[:m]
    fc = FullColumns(**{k: v.ctypes.data for k, v in arrs.items()})
    q4 = q4 or {}
    rows = np.array(sorted(q4), dtype=np.uint64)
    vals = np.array([q4[r] for r in sorted(q4)], dtype=np.uint64).reshape(-1, 4)
    rc = lib().oracle_materialize_fast(*[w.ctypes.data for w in wires], values.ctypes.data, values.shape[0], rows.ctypes.data,
                                       vals.ctypes.data, len(q4), r0, r1, threads, C.byref(fc))
    assert rc == 0, "oracle_materialize_fast: a wire holds a Variable beyond the table"
    return arrs


def max_bound_batch(max_mont: np.ndarray, witness: np.ndarray, check: bool = True):
    L = lib()
    max_mont, witness = _as_fr_array(max_mont), _as_fr_array(witness)
    batch = witness.shape[0]
    one = L.fr_from_u64(1)
    ns = [int(L.num_bits_closest_power_of_two(L.fr_sub(fr(m), one))) for m in max_mont]
    G, V = sum(2 * n + 5 for n in ns), sum(n + 262 for n in ns)
    res, nb = np.zeros(batch, dtype=np.uint64), np.zeros(batch, dtype=np.uint64)
    gb, vb, ng, nv = (C.c_uint64() for _ in range(4))
    arrs, cols = _alloc_columns(G, V)
    rc = L.oracle_max_bound_batch(max_mont.ctypes.data, witness.ctypes.data, batch, int(check), C.byref(cols),
                                  res.ctypes.data, nb.ctypes.data, C.byref(gb), C.byref(vb), C.byref(ng), C.byref(nv))
    assert ng.value == G and nv.value == V
    arrs.update(result_vars=res, num_bits=nb, gate_base=gb.value, var_base=vb.value, n_gates=G, n_vars=V,
                satisfied=(rc == 0))
    return arrs


def max_bound_plan(max_mont: np.ndarray, threads: int = 1):
    """ladder length per item and the prefix sums of rows / variables (batch + 1 entries each) of
    `for i: allocate(w_i); max_bound(bound_i, w_i)`"""
    max_mont = _as_fr_array(max_mont)
    batch = max_mont.shape[0]
    nb = np.zeros(batch, dtype=np.uint64)
    roff, voff = np.zeros(batch + 1, dtype=np.uint64), np.zeros(batch + 1, dtype=np.uint64)
    lib().oracle_max_bound_plan(max_mont.ctypes.data, batch, threads, nb.ctypes.data, roff.ctypes.data, voff.ctypes.data)
    return nb, roff, voff


def max_bound_fast(max_mont: np.ndarray, witness: np.ndarray, plan, lo: int, hi: int, var_base: int = 5, threads: int = 1,
                   out: dict | None = None):
    """oracle/fast.c: items [lo, hi) of max_bound_batch's columns at the numbering of the whole batch (`plan` from
    max_bound_plan over the whole batch); rows relative to item lo's first row"""
    max_mont, witness = _as_fr_array(max_mont), _as_fr_array(witness)
    nb, roff, voff = plan
    G, V = int(roff[hi] - roff[lo]), int(voff[hi] - voff[lo])
    arrs, cols = _columns_into(out, G, V)
    res = np.zeros(hi - lo, dtype=np.uint64)
    rc = lib().oracle_max_bound_fast(max_mont.ctypes.data, witness.ctypes.data, nb.ctypes.data, roff.ctypes.data,
                                     voff.ctypes.data, lo, hi, var_base, threads, C.byref(cols), res.ctypes.data)
    assert rc == 0
    arrs.update(result_vars=res, n_gates=G, n_vars=V, num_bits=nb[lo:hi])
    return arrs


def scalar_mix_plan(v: np.ndarray):
    v = _as_fr_array(v)
    batch = v.shape[0]
    roff, voff = np.zeros(batch + 1, dtype=np.uint64), np.zeros(batch + 1, dtype=np.uint64)
    err = np.zeros(batch, dtype=np.uint8)
    lib().oracle_scalar_mix_plan(v.ctypes.data, batch, roff.ctypes.data, voff.ctypes.data, err.ctypes.data)
    return roff, voff, err


def scalar_mix_fast(v, y, s, a, b, plan, lo: int, hi: int, var_base: int = 5, zero_var: int = 0, threads: int = 1,
                    out: dict | None = None):
    """oracle/fast.c: items [lo, hi) of scalar_mix_batch's columns at the numbering of the whole batch"""
    v, y, s, a, b = (_as_fr_array(x) for x in (v, y, s, a, b))
    roff, voff = plan[0], plan[1]
    G, V = int(roff[hi] - roff[lo]), int(voff[hi] - voff[lo])
    arrs, cols = _columns_into(out, G, V)
    res = np.zeros(2 * (hi - lo), dtype=np.uint64)
    rc = lib().oracle_scalar_mix_fast(v.ctypes.data, y.ctypes.data, s.ctypes.data, a.ctypes.data, b.ctypes.data,
                                      roff.ctypes.data, voff.ctypes.data, lo, hi, var_base, zero_var, threads,
                                      C.byref(cols), res.ctypes.data)
    assert rc == 0
    arrs.update(result_vars=res.reshape(-1, 2), n_gates=G, n_vars=V)
    return arrs


def scalar_mix_batch(v, y, s, a, b, check: bool = True):
    L = lib()
    v, y, s, a, b = (_as_fr_array(x) for x in (v, y, s, a, b))
    batch = v.shape[0]
    nerr = int((v == 0).all(axis=1).sum())
    G, V = 10 * batch - 2 * nerr, 15 * batch - 2 * nerr
    res, err = np.zeros(2 * batch, dtype=np.uint64), np.zeros(batch, dtype=np.uint8)
    gb, vb, ng, nv = (C.c_uint64() for _ in range(4))
    arrs, cols = _alloc_columns(G, V)
    rc = L.oracle_scalar_mix_batch(v.ctypes.data, y.ctypes.data, s.ctypes.data, a.ctypes.data, b.ctypes.data, batch,
                                   int(check), C.byref(cols), res.ctypes.data, err.ctypes.data, C.byref(gb),
                                   C.byref(vb), C.byref(ng), C.byref(nv))
    assert ng.value == G and nv.value == V, (ng.value, nv.value, G, V)
    arrs.update(result_vars=res.reshape(batch, 2), err_mask=err, gate_base=gb.value, var_base=vb.value, n_gates=G,
                n_vars=V, satisfied=(rc == 0))
    return arrs


class Composer:
    """RAII handle on the C oracle's composer with numpy export (rows >= gate_base, variables >= var_base)."""

    def __init__(self, dummy: bool = True):
        self.L = lib()
        self.c = self.L.composer_new() if dummy else self.L.composer_new_without_dummy()

    def __del__(self):
        try:
            self.L.composer_free(self.c)
        except Exception:
            pass

    @property
    def n(self):
        return int(self.L.composer_circuit_size(self.c))

    @property
    def num_vars(self):
        return int(self.L.composer_num_variables(self.c))

    def add_input(self, limbs_) -> int:
        return int(self.L.composer_add_input(self.c, fr(limbs_)))

    def allocate(self, limbs_) -> AllocatedScalar:
        return self.L.allocated_scalar_allocate(self.c, fr(limbs_))

    def check(self) -> int:
        return int(self.L.composer_check(self.c))

    def sigma(self, padded_n: int) -> np.ndarray:
        out = np.zeros(4 * padded_n, dtype=np.uint64)
        self.L.composer_sigma(self.c, padded_n, out.ctypes.data)
        return out.reshape(4, padded_n)

    def full_columns(self) -> dict:
        """the columns beyond the eight live ones: q_4, q_arith, w_4 and the dense public-input vector"""
        n = self.n
        out = {}
        for name, col in (("q_4", 5), ("q_arith", 6)):
            p = self.L.composer_selector(self.c, col)
            out[name] = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(n, 4)).copy()
        out["w_4"] = np.ctypeslib.as_array(self.L.composer_wire(self.c, 3), shape=(n,)).copy()
        pi = np.zeros((n, 4), dtype=np.uint64)
        self.L.composer_dense_pi(self.c, pi.ctypes.data)
        out["dense_pi"] = pi
        return out

    def export(self, gate_base: int = 0, var_base: int = 0) -> dict:
        n, nv = self.n, self.num_vars
        out = {}
        for name, col in (("q_m", 0), ("q_l", 1), ("q_r", 2), ("q_o", 3), ("q_c", 4)):
            p = self.L.composer_selector(self.c, col)
            a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(n, 4)) if n else np.zeros((0, 4), np.uint64)
            out[name] = a[gate_base:].copy()
        for name, col in (("w_l", 0), ("w_r", 1), ("w_o", 2)):
            p = self.L.composer_wire(self.c, col)
            a = np.ctypeslib.as_array(p, shape=(n,)) if n else np.zeros((0,), np.uint64)
            out[name] = a[gate_base:].copy()
        vals = np.zeros((nv, 4), dtype=np.uint64)
        self.L.composer_values_dense(self.c, vals.ctypes.data)
        out["var_values"] = vals[var_base:].copy()
        return out

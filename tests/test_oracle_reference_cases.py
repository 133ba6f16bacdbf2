"""Pins the oracle (C restatement and big-int model) on everything the reference's tests pin:
outcome bits and satisfiability of every case, plus C-vs-model limb-for-limb agreement."""
import ctypes as C
import random

import numpy as np
import pytest

from oracle import model
from oracle import pyoracle as po
from oracle.model import Q, mont_limbs
from tests.refcases import MAX_BOUND_CASES, MAYBE_EQUAL_CASES, RANGE_CHECK_CASES

L = po.lib()


class CComposer:
    def __init__(self, dummy=True):
        self.c = L.composer_new() if dummy else L.composer_new_without_dummy()

    def __del__(self):
        L.composer_free(self.c)

    def export(self):
        n, nv = L.composer_circuit_size(self.c), L.composer_num_variables(self.c)
        out = {}
        for name, col in (("q_m", 0), ("q_l", 1), ("q_r", 2), ("q_o", 3), ("q_c", 4)):
            p = L.composer_selector(self.c, col)
            out[name] = [po.limbs(p[i]) for i in range(n)]
        for name, col in (("w_l", 0), ("w_r", 1), ("w_o", 2)):
            p = L.composer_wire(self.c, col)
            out[name] = [int(p[i]) for i in range(n)]
        out["var_values"] = [po.limbs(L.composer_value(self.c, v)) for v in range(nv)]
        return out

    def value(self, v):
        return po.fr_to_int(L.composer_value(self.c, v))


def test_initial_state():
    c, m = CComposer(), model.Composer()
    assert L.composer_circuit_size(c.c) == 3 == m.n
    assert L.composer_num_variables(c.c) == 5 == len(m.variables)
    assert L.composer_zero_var(c.c) == 0
    assert c.export() == model.export(m)
    assert L.composer_check(c.c) == -1 and m.check() == -1
    c2 = CComposer(dummy=False)
    assert L.composer_circuit_size(c2.c) == 1 and L.composer_num_variables(c2.c) == 1


@pytest.mark.parametrize("max_range,witness,expected", MAX_BOUND_CASES)
def test_max_bound_reference_cases(max_range, witness, expected):
    c, m = CComposer(), model.Composer()
    w = L.allocated_scalar_allocate(c.c, po.fr_from_int(witness))
    nb = C.c_uint64()
    res = L.max_bound(c.c, po.fr_from_int(max_range), w, C.byref(nb))
    mw = model.AllocatedScalar.allocate(m, witness)
    mres, mn = model.max_bound(m, max_range, mw)
    assert (res, nb.value) == (mres, mn)
    assert c.value(res) == int(expected) == m.variables[mres]
    assert L.composer_check(c.c) == -1 and m.check() == -1
    assert c.export() == model.export(m)
    n = nb.value
    assert L.composer_circuit_size(c.c) == 3 + 2 * n + 5
    assert L.composer_num_variables(c.c) == 5 + 1 + n + 261
    # tests/range_gadgets_tests.rs:26 -- constrain_to_constant(res, outcome) must hold
    L.composer_constrain_to_constant(c.c, res, po.fr_from_int(int(expected)), None)
    assert L.composer_check(c.c) == -1
    L.composer_constrain_to_constant(c.c, res, po.fr_from_int(1 - int(expected)), None)
    assert L.composer_check(c.c) == L.composer_circuit_size(c.c) - 1


@pytest.mark.parametrize("min_range,max_range,witness,expected", RANGE_CHECK_CASES)
def test_range_check_reference_cases(min_range, max_range, witness, expected):
    c, m = CComposer(), model.Composer()
    w = L.allocated_scalar_allocate(c.c, po.fr_from_int(witness))
    res = L.range_check(c.c, po.fr_from_int(min_range), po.fr_from_int(max_range), w)
    mres = model.range_check(m, min_range, max_range, model.AllocatedScalar.allocate(m, witness))
    assert res == mres
    assert c.value(res) == int(expected) == m.variables[mres]
    assert L.composer_check(c.c) == -1 and m.check() == -1
    assert c.export() == model.export(m)
    n = model.num_bits_closest_power_of_two(max_range - 1)
    assert L.composer_circuit_size(c.c) == 3 + 4 * n + 11
    assert L.composer_num_variables(c.c) == 5 + 2 * n + 524


def test_scalar_decomposition_reference_case():
    """src/range.rs:205-233: -100 does not fit 8 bits -> is_eq == 0; verifier side uses witness 1"""
    for witness, expected in ((Q - 100, 0), (1, 1)):
        c, m = CComposer(), model.Composer()
        w = L.allocated_scalar_allocate(c.c, po.fr_from_int(witness))
        bits = (C.c_uint64 * 8)()
        is_eq = L.scalar_decomposition_gadget(c.c, 8, w, bits)
        mis_eq, mbits = model.scalar_decomposition_gadget(m, 8, model.AllocatedScalar.allocate(m, witness))
        assert is_eq == mis_eq and list(bits) == mbits
        assert c.value(is_eq) == expected
        assert L.composer_check(c.c) == -1
        assert c.export() == model.export(m)
    # structure (selectors + wires) is witness independent
    a, b = CComposer(), CComposer()
    for cc, wv in ((a, Q - 100), (b, 1)):
        L.scalar_decomposition_gadget(cc.c, 8, L.allocated_scalar_allocate(cc.c, po.fr_from_int(wv)), None)
    ea, eb = a.export(), b.export()
    for k in ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o"):
        assert ea[k] == eb[k]
    # num_bits > 256: the reference panics (range.rs:134)
    cc = CComposer()
    assert L.scalar_decomposition_gadget(cc.c, 257, L.allocated_scalar_allocate(cc.c, po.fr_from_int(5)), None) == 2**64 - 1


@pytest.mark.parametrize("a,b,expected", MAYBE_EQUAL_CASES)
def test_maybe_equal_reference_cases(a, b, expected):
    c, m = CComposer(), model.Composer()
    aa, bb = (L.allocated_scalar_allocate(c.c, po.fr_from_int(x)) for x in (a, b))
    bit = L.maybe_equal(c.c, aa, bb)
    mbit = model.maybe_equal(m, model.AllocatedScalar.allocate(m, a), model.AllocatedScalar.allocate(m, b))
    assert bit == mbit and c.value(bit) == int(expected)
    assert L.composer_check(c.c) == -1
    assert c.export() == model.export(m)
    assert L.composer_circuit_size(c.c) == 3 + 3 and L.composer_num_variables(c.c) == 5 + 2 + 3


def test_conditionally_select_zero_cases():
    """tests/scalar_gadgets_tests.rs:70-122"""
    rng = random.Random(3)
    for sel in (0, 1):
        value = rng.randrange(1, Q)
        c, m = CComposer(), model.Composer()
        v, s = L.composer_add_input(c.c, po.fr_from_int(value)), L.composer_add_input(c.c, po.fr_from_int(sel))
        res = L.conditionally_select_zero(c.c, v, s)
        mres = model.conditionally_select_zero(m, m.add_input(value), m.add_input(sel))
        assert res == mres and c.value(res) == (value if sel else 0)
        assert c.export() == model.export(m)
        L.composer_constrain_to_constant(c.c, res, po.fr_from_int(0), None)
        # selector 0 -> constraining the result to 0 holds; selector 1 -> it must fail (:119)
        assert (L.composer_check(c.c) == -1) == (sel == 0)


def test_conditionally_select_one_cases():
    """tests/scalar_gadgets_tests.rs:124-178 -- expected value injected as public input"""
    rng = random.Random(4)
    for sel in (0, 1):
        value = rng.randrange(1, Q)
        expected = value if sel else 1
        c, m = CComposer(), model.Composer()
        v, s = L.composer_add_input(c.c, po.fr_from_int(value)), L.composer_add_input(c.c, po.fr_from_int(sel))
        res = L.conditionally_select_one(c.c, v, s)
        mres = model.conditionally_select_one(m, m.add_input(value), m.add_input(sel))
        assert res == mres and c.value(res) == expected
        pi = po.fr_from_int(-expected)
        L.composer_constrain_to_constant(c.c, res, po.fr_from_int(0), C.byref(pi))
        m.constrain_to_constant(mres, 0, -expected)
        assert L.composer_check(c.c) == -1 and m.check() == -1
        assert c.export() == model.export(m)
        n = L.composer_circuit_size(c.c)
        dense = np.zeros((n, 4), dtype=np.uint64)
        L.composer_dense_pi(c.c, dense.ctypes.data)
        assert dense[:-1].sum() == 0 and [int(x) for x in dense[-1]] == mont_limbs(-expected)
        assert n == 3 + 4 + 1 and L.composer_num_variables(c.c) == 5 + 2 + 4


def test_is_non_zero_cases():
    """tests/scalar_gadgets_tests.rs:180-236"""
    rng = random.Random(5)
    # (0, 0) -> Err after one variable and one row were already pushed (scalar.rs:69-79)
    c, m = CComposer(), model.Composer()
    v = L.composer_add_input(c.c, po.fr_from_int(0))
    assert L.is_non_zero(c.c, v, po.fr_from_int(0)) == 1
    with pytest.raises(model.NonExistingInverse):
        model.is_non_zero(m, m.add_input(0), 0)
    assert L.composer_circuit_size(c.c) == 3 + 1 and L.composer_num_variables(c.c) == 5 + 1 + 1
    assert c.export() == model.export(m)
    # mismatching var / value_assigned -> Ok but unsatisfied (:224)
    c = CComposer()
    v = L.composer_add_input(c.c, po.fr_from_int(rng.randrange(1, Q)))
    assert L.is_non_zero(c.c, v, po.fr_from_int(rng.randrange(1, Q))) == 0
    assert L.composer_check(c.c) >= 0
    # equal and non-zero -> satisfied (:235)
    r = rng.randrange(1, Q)
    c, m = CComposer(), model.Composer()
    v = L.composer_add_input(c.c, po.fr_from_int(r))
    assert L.is_non_zero(c.c, v, po.fr_from_int(r)) == 0
    model.is_non_zero(m, m.add_input(r), r)
    assert L.composer_check(c.c) == -1 and m.check() == -1
    assert c.export() == model.export(m)
    assert L.composer_circuit_size(c.c) == 3 + 3 and L.composer_num_variables(c.c) == 5 + 1 + 3


def test_batch_driver_matches_model_and_counts():
    """the batch driver the HIP path is compared with == a model loop; gate/variable counts of SURVEY 3.1"""
    rng = random.Random(11)
    for (mn, mx) in ((0, 2**64), (50_000, 250_000), (3, 2**254), (0, 2)):
        ws = [rng.randrange(Q) if i % 2 else rng.randrange(max(1, min(Q, 2 * mx))) for i in range(5)] + [mn, mx - 1, mx]
        out = po.range_check_batch(mont_limbs(mn), mont_limbs(mx), po.ints_to_mont_array(ws))
        assert out["satisfied"]
        m = model.Composer()
        res = [model.range_check(m, mn, mx, model.AllocatedScalar.allocate(m, w)) for w in ws]
        assert m.check() == -1
        exp = model.export(m, 3, 5)
        for k in ("q_m", "q_l", "q_r", "q_o", "q_c", "var_values"):
            assert out[k].tolist() == exp[k], k
        for k in ("w_l", "w_r", "w_o"):
            assert out[k].tolist() == exp[k], k
        assert out["result_vars"].tolist() == res
        n = out["num_bits"]
        # what the gadget decides: both (max-1-w) mod q and (w-min) mod q fit n bits.  For n <= 253 that is the
        # true predicate min <= w < max; at n = 255 every field element fits (the reference's own behaviour).
        assert [m.variables[r] for r in res] == [int((mx - 1 - w) % Q < 2**n and (w - mn) % Q < 2**n) for w in ws]
        if n <= 253:
            assert [m.variables[r] for r in res] == [int(mn <= w < mx) for w in ws]
        # permutation bookkeeping: the witness variable sits on 4 wire positions (w_l, w_r of the two bound rows)
        assert out["n_gates"] == len(ws) * (4 * n + 11) and out["n_vars"] == len(ws) * (2 * n + 524)

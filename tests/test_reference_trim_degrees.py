"""The one thing about ROWS that the reference itself pins: every one of its test circuits is preprocessed, proven and verified
under a commit key trimmed to a fixed degree (tests/refcases.py, TRIM_LOG2, with the reference lines), so each circuit --
StandardComposer::new()'s three rows, the gadget's rows, the closing constrain_to_constant -- padded to a power of two must
fit that degree.  An upper bound, not row parity; checked here on the C oracle and on the big-int model (and on the device
composer in tests/test_gpu_composer.py), together with what it excludes."""
import ctypes as C

import pytest

from oracle import model
from oracle import pyoracle as po
from oracle.model import Q
from tests.refcases import (MAX_BOUND_CASES, MAYBE_EQUAL_CASES, RANGE_CHECK_CASES, TRIM_LOG2, fits_trim_degree,
                            padded_circuit_size)

L = po.lib()
F = po.fr_from_int


def sizes(build_c, build_m):
    c, m = po.Composer(), model.Composer()
    build_c(c)
    build_m(m)
    assert c.check() == -1 and m.check() == -1
    assert c.n == m.n
    return c.n


@pytest.mark.parametrize("min_range,max_range,witness,expected", RANGE_CHECK_CASES)
def test_range_check_circuits_fit_their_key(min_range, max_range, witness, expected):
    """tests/range_gadgets_tests.rs:29-44 under trim(1 << 10) (:111-112)"""
    def on_c(c):
        res = L.range_check(c.c, F(min_range), F(max_range), c.allocate(po.limbs(F(witness))))
        L.composer_constrain_to_constant(c.c, res, F(int(expected)), None)

    def on_m(m):
        res = model.range_check(m, min_range, max_range, model.AllocatedScalar.allocate(m, witness))
        m.constrain_to_constant(res, int(expected), None)
    n_rows = sizes(on_c, on_m)
    n = model.num_bits_closest_power_of_two(max_range - 1)
    assert n_rows == 3 + (4 * n + 11) + 1
    assert fits_trim_degree(n_rows, "range_check"), (n_rows, padded_circuit_size(n_rows))


def test_what_the_range_check_key_excludes():
    """the bound is not vacuous: for the case over [2^126, 2^127 + 1) the key of degree 2^10 admits ladders up to n = 252 and
    no longer -- the restatement's n = 129 (bit length of 2^127, plus one) is inside, the whole 253..255 family
    (e.g. a num_bits_closest_power_of_two that returned the field's width) is outside; and the circuit is padded to exactly
    the key's degree, so one more block of rows per ladder step would not fit either"""
    n = model.num_bits_closest_power_of_two(2**127 + 1 - 1)
    assert n == 129 == int(L.num_bits_closest_power_of_two(F(2**127)))
    rows = lambda k: 3 + (4 * k + 11) + 1
    assert fits_trim_degree(rows(n), "range_check") and padded_circuit_size(rows(n)) == 1 << TRIM_LOG2["range_check"]
    assert fits_trim_degree(rows(252), "range_check") and not fits_trim_degree(rows(253), "range_check")
    # two ladders of FOUR rows per bit instead of two (8n + 11) would already be outside at this case's n
    assert not fits_trim_degree(3 + (8 * n + 11) + 1, "range_check")


@pytest.mark.parametrize("max_range,witness,expected", MAX_BOUND_CASES)
def test_max_bound_circuits_fit_their_key(max_range, witness, expected):
    """tests/range_gadgets_tests.rs:13-27 under trim(1 << 10) (:49-50)"""
    def on_c(c):
        res = L.max_bound(c.c, F(max_range), c.allocate(po.limbs(F(witness))), None)
        L.composer_constrain_to_constant(c.c, res, F(int(expected)), None)

    def on_m(m):
        res, _ = model.max_bound(m, max_range, model.AllocatedScalar.allocate(m, witness))
        m.constrain_to_constant(res, int(expected), None)
    n_rows = sizes(on_c, on_m)
    n = model.num_bits_closest_power_of_two(max_range - 1)
    assert n_rows == 3 + (2 * n + 5) + 1 and fits_trim_degree(n_rows, "max_bound")
    # (2^128 - 1: n = 129, 267 rows, padded to 512 -- a key of degree 2^10 admits n <= 507, i.e. every ladder there is)
    assert fits_trim_degree(3 + (2 * 255 + 5) + 1, "max_bound")


def test_scalar_decomposition_circuit_fits_its_key():
    """src/range.rs:205-233 under trim(1 << 10) (:208-209): 8 bits of -100 (prover) / of 1 (verifier), is_eq constrained to 0"""
    for witness in (Q - 100, 1):
        c, m = po.Composer(), model.Composer()
        is_eq = L.scalar_decomposition_gadget(c.c, 8, c.allocate(po.limbs(F(witness))), None)
        L.composer_constrain_to_constant(c.c, is_eq, F(0), None)
        mis_eq, _ = model.scalar_decomposition_gadget(m, 8, model.AllocatedScalar.allocate(m, witness))
        m.constrain_to_constant(mis_eq, 0, None)
        assert c.n == m.n == 3 + (2 * 8 + 4) + 1 and fits_trim_degree(c.n, "scalar_decomposition")


@pytest.mark.parametrize("a,b,expected", MAYBE_EQUAL_CASES)
def test_maybe_equal_circuits_fit_their_key(a, b, expected):
    """tests/scalar_gadgets_tests.rs:19-31 under trim(1 << 9) (:16-17)"""
    def on_c(c):
        bit = L.maybe_equal(c.c, c.allocate(po.limbs(F(a))), c.allocate(po.limbs(F(b))))
        L.composer_constrain_to_constant(c.c, bit, F(int(expected)), None)

    def on_m(m):
        bit = model.maybe_equal(m, model.AllocatedScalar.allocate(m, a), model.AllocatedScalar.allocate(m, b))
        m.constrain_to_constant(bit, int(expected), None)
    assert fits_trim_degree(sizes(on_c, on_m), "maybe_equal")


def test_select_and_is_non_zero_circuits_fit_their_keys():
    """tests/scalar_gadgets_tests.rs:70-236, all under trim(1 << 7) (:82-83, :139-140, :193-194)"""
    value = Q - 12345
    for sel in (0, 1):
        c, m = po.Composer(), model.Composer()
        res = L.conditionally_select_zero(c.c, c.add_input(po.limbs(F(value))), c.add_input(po.limbs(F(sel))))
        L.composer_constrain_to_constant(c.c, res, F(0), None)
        mres = model.conditionally_select_zero(m, m.add_input(value), m.add_input(sel))
        m.constrain_to_constant(mres, 0, None)
        assert c.n == m.n and fits_trim_degree(c.n, "select_zero")
        c, m = po.Composer(), model.Composer()
        expected = value if sel else 1
        res = L.conditionally_select_one(c.c, c.add_input(po.limbs(F(value))), c.add_input(po.limbs(F(sel))))
        pi = F(-expected)
        L.composer_constrain_to_constant(c.c, res, F(0), C.byref(pi))
        mres = model.conditionally_select_one(m, m.add_input(value), m.add_input(sel))
        m.constrain_to_constant(mres, 0, -expected)
        assert c.check() == -1 and m.check() == -1
        assert c.n == m.n and fits_trim_degree(c.n, "select_one")
    c, m = po.Composer(), model.Composer()
    assert L.is_non_zero(c.c, c.add_input(po.limbs(F(value))), F(value)) == 0
    model.is_non_zero(m, m.add_input(value), value)
    assert c.check() == -1 and c.n == m.n and fits_trim_degree(c.n, "is_non_zero")

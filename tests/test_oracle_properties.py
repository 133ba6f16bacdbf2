"""Property tests (hypothesis) that cross-pin the three CPU statements of the path: big-int model, faithful C
restatement, fast C form -- on random public bounds and witnesses, including the structure-is-witness-independent
property the reference's tests rely on (tests/scalar_gadgets_tests.rs:36 vs :43)."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import model
from oracle import pyoracle as po
from oracle.model import Q, mont_limbs

scalars = st.one_of(st.integers(0, Q - 1), st.integers(0, 2**64), st.sampled_from([0, 1, 2, Q - 1, Q - 2, 2**254, 2**255 % Q]))
small_bounds = st.integers(0, 2**40)


@settings(max_examples=25, deadline=None, derandomize=True)
@given(mn=small_bounds, span=st.integers(1, 2**40), ws=st.lists(scalars, min_size=1, max_size=4))
def test_range_check_three_statements_agree(mn, span, ws):
    mx = mn + span
    wit = po.ints_to_mont_array(ws)
    slow = po.range_check_batch(mont_limbs(mn), mont_limbs(mx), wit)
    fast = po.range_check_fast(mont_limbs(mn), mont_limbs(mx), wit, threads=2)
    assert slow["satisfied"]
    m = model.Composer()
    res = [model.range_check(m, mn, mx, model.AllocatedScalar.allocate(m, w)) for w in ws]
    exp = model.export(m, 3, 5)
    for k in ("q_m", "q_l", "q_r", "q_o", "q_c", "var_values"):
        assert slow[k].tolist() == exp[k] and np.array_equal(slow[k], fast[k]), k
    for k in ("w_l", "w_r", "w_o"):
        assert slow[k].tolist() == exp[k] and np.array_equal(slow[k], fast[k]), k
    assert slow["result_vars"].tolist() == res == fast["result_vars"].tolist()
    # bounds below 2^41 give ladders of at most 42 bits: the gadget decides the true predicate
    assert [m.variables[r] for r in res] == [int(mn <= w < mx) for w in ws]


@settings(max_examples=20, deadline=None, derandomize=True)
@given(mx=st.one_of(st.integers(0, 2**254), st.integers(0, 2**20)), w1=scalars, w2=scalars)
def test_max_bound_structure_is_witness_independent(mx, w1, w2):
    a = po.max_bound_batch(po.ints_to_mont_array([mx]), po.ints_to_mont_array([w1]))
    b = po.max_bound_batch(po.ints_to_mont_array([mx]), po.ints_to_mont_array([w2]))
    assert a["satisfied"] and b["satisfied"]
    for k in ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o"):
        assert np.array_equal(a[k], b[k]), k
    n = int(a["num_bits"][0])
    assert n == model.num_bits_closest_power_of_two(mx - 1)
    out = po.fr_to_int(po.fr(a["var_values"][int(a["result_vars"][0]) - 5]))
    assert out == int((mx - 1 - w1) % Q < 2**n)


@settings(max_examples=30, deadline=None, derandomize=True)
@given(v=scalars, y=scalars, s=st.integers(0, 1), a=scalars, b=scalars, same=st.booleans())
def test_scalar_mix_model_vs_c(v, y, s, a, b, same):
    if same:
        b = a
    arr = [po.ints_to_mont_array([x]) for x in (v, y, s, a, b)]
    ora = po.scalar_mix_batch(*arr)
    assert ora["satisfied"]
    m = model.Composer()
    vv, yv, sv = m.add_input(v), m.add_input(y), m.add_input(s)
    aa, bb = model.AllocatedScalar.allocate(m, a), model.AllocatedScalar.allocate(m, b)
    err = 0
    try:
        model.is_non_zero(m, vv, v)
    except model.NonExistingInverse:
        err = 1
    sel = model.conditionally_select_one(m, yv, sv)
    eq = model.maybe_equal(m, aa, bb)
    exp = model.export(m, 3, 5)
    for k in ("q_m", "q_l", "q_r", "q_o", "q_c", "var_values", "w_l", "w_r", "w_o"):
        assert ora[k].tolist() == exp[k], k
    assert ora["err_mask"].tolist() == [err] and ora["result_vars"].tolist() == [[sel, eq]]
    assert m.variables[sel] == (y if s else 1) % Q and m.variables[eq] == int(a % Q == b % Q)

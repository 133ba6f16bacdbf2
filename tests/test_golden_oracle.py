"""CPU: the C oracle reproduces every committed golden fixture (tests/golden/*.npz, made by the big-int model)."""
import glob
import os

import numpy as np
import pytest

from oracle import pyoracle as po

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
COLS = ("q_m", "q_l", "q_r", "q_o", "q_c", "w_l", "w_r", "w_o", "var_values")


def load(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


def same(ora, g):
    for k in COLS:
        assert np.array_equal(ora[k], g[k]), k


@pytest.mark.parametrize("name", ["range_check_ref_50k_250k", "range_check_ref_2p126_2p127", "range_check_c1_n65",
                                  "range_check_c2_n255"])
def test_range_check_golden(name):
    g = load(name)
    ora = po.range_check_batch(g["min_range"][0], g["max_range"][0], g["witness"])
    assert ora["satisfied"]
    same(ora, g)
    assert np.array_equal(ora["result_vars"], g["result_vars"])
    vals = [po.fr_to_int(po.fr(ora["var_values"][int(r) - 5])) for r in g["result_vars"]]
    assert vals == g["expected"].tolist()


def test_max_bound_golden():
    g = load("max_bound_ref")
    ora = po.max_bound_batch(g["max_range"], g["witness"])
    assert ora["satisfied"]
    same(ora, g)
    assert np.array_equal(ora["result_vars"], g["result_vars"]) and np.array_equal(ora["num_bits"], g["num_bits"])


def test_scalar_mix_golden():
    g = load("scalar_mix")
    ora = po.scalar_mix_batch(g["v"], g["y"], g["s"], g["a"], g["b"])
    assert ora["satisfied"]
    same(ora, g)
    assert np.array_equal(ora["result_vars"], g["result_vars"]) and np.array_equal(ora["err_mask"], g["err_mask"])


class OracleOps:
    """tests/refcases.py:full_circuit on the C oracle"""

    def __init__(self, c):
        self.c, self.L = c, c.L

    @staticmethod
    def m(v):
        from oracle.model import mont_limbs
        return np.array(mont_limbs(v), dtype=np.uint64)

    @classmethod
    def f(cls, v): return po.fr(cls.m(v))

    def add_input(self, v): return self.c.add_input(self.m(v))
    def allocate(self, v): return self.c.allocate(self.m(v))
    def range_check_loop(self, mn, mx, ws): return [int(self.L.range_check(self.c.c, self.f(mn), self.f(mx), self.allocate(w))) for w in ws]
    def max_bound(self, mx, a): return int(self.L.max_bound(self.c.c, self.f(mx), a, None))
    def maybe_equal(self, a, b): return int(self.L.maybe_equal(self.c.c, a, b))
    def is_non_zero(self, var, value): assert self.L.is_non_zero(self.c.c, var, self.f(value)) == 0
    def conditionally_select_one(self, y, s): return int(self.L.conditionally_select_one(self.c.c, y, s))
    def conditionally_select_zero(self, x, s): return int(self.L.conditionally_select_zero(self.c.c, x, s))
    def boolean_gate(self, a): self.L.composer_boolean_gate(self.c.c, a)

    def constrain_to_constant(self, a, c, pi):
        import ctypes as C
        p = self.f(pi) if pi is not None else None
        self.L.composer_constrain_to_constant(self.c.c, a, self.f(c), C.byref(p) if p is not None else None)


def test_full_composer_golden():
    """every gadget once + a public input, from row 0: live columns, fourth wire, q_4 / q_arith, dense PI and sigma of
    the C oracle's composer against the fixture frozen from the big-int model"""
    from tests.refcases import full_circuit
    g = load("composer_full")
    c = po.Composer()
    full_circuit(OracleOps(c))
    assert c.check() == -1
    same(c.export(), g)
    full = c.full_columns()
    for k in ("q_4", "q_arith", "w_4", "dense_pi"):
        assert np.array_equal(full[k], g[k]), k
    padded = int(g["padded_n"][0])
    assert np.array_equal(c.sigma(padded).reshape(-1), g["sigma"])


def test_all_fixtures_are_covered():
    names = {os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz"))}
    assert names == {"range_check_ref_50k_250k", "range_check_ref_2p126_2p127", "range_check_c1_n65",
                     "range_check_c2_n255", "max_bound_ref", "scalar_mix", "maybe_equal_ref", "composer_full"}

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


def test_all_fixtures_are_covered():
    names = {os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz"))}
    assert names == {"range_check_ref_50k_250k", "range_check_ref_2p126_2p127", "range_check_c1_n65",
                     "range_check_c2_n255", "max_bound_ref", "scalar_mix", "maybe_equal_ref"}

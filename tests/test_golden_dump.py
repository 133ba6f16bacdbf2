"""CPU: the f4 hand-off.  tests/golden/compare_dump.py must accept a dump written in the layout the Rust side
(bindings/rust/fork/dump_columns.rs + examples/dump_golden.rs) writes, and name what a difference falsifies.  The dump
here comes from oracle/model.py -- the fixtures' own source -- because the reference cannot run in this environment."""
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def test_model_dump_matches_the_fixtures_and_differences_are_named(tmp_path):
    import compare_dump as cd
    root = str(tmp_path / "dump")
    cd.write_from_model(root)
    out = io.StringIO()
    assert cd.compare(root, out) == 0, out.getvalue()
    assert "parity PINNED" in out.getvalue() and out.getvalue().count("identical") == len(cd.CASES)

    def poke(case, name, word):
        """flip one word of one array of the dump; returns a function that restores the file"""
        p = os.path.join(root, case, name + ".u64")
        before = open(p, "rb").read()
        a = np.frombuffer(before, dtype="<u8").copy()
        a[word] += np.uint64(1)
        a.tofile(p)
        return lambda: open(p, "wb").write(before)

    # a wire index off by one in one circuit: that circuit alone is reported, with the numbering recollection
    undo = poke("max_bound_ref", "w_o", 40)
    out = io.StringIO()
    assert cd.compare(root, out) == 1
    text = out.getvalue()
    assert "max_bound_ref" in text and "DIFFERS: w_o" in text and "Variable numbering" in text and "NOT pinned: 1 of 8" in text
    undo()
    # a different initial state shows up in EVERY circuit and is named as such
    for case in cd.CASES:
        poke(case, "q_c", 4 * 1 + 0)   # row 1 = the first dummy constraint
    out = io.StringIO()
    assert cd.compare(root, out) == len(cd.CASES)
    assert out.getvalue().count("the initial composer state") == len(cd.CASES)


def test_rust_hand_off_sources_cover_the_fixtures():
    """the Rust example dumps exactly the circuits compare_dump.py expects, and the fork modules use no invented accessor"""
    import compare_dump as cd
    ex = open(os.path.join(ROOT, "bindings", "rust", "examples", "dump_golden.rs")).read()
    for case in cd.CASES:
        assert '"%s"' % case in ex, case
    dump = open(os.path.join(ROOT, "bindings", "rust", "fork", "dump_columns.rs")).read()
    for name in ("q_m", "q_l", "q_r", "q_o", "q_c", "q_4", "q_arith", "w_l", "w_r", "w_o", "w_4", "var_values", "dense_pi", "sigma", "meta"):
        assert '"%s"' % name in dump, name
    shim = open(os.path.join(ROOT, "bindings", "rust", "fork", "hip_composer.rs")).read()
    for invented in ("Variable::new(", ".index()", "from_raw_unchecked", "internal_repr"):
        assert invented not in shim, invented

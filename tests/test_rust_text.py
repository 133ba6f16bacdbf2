"""The Rust side of the hand-off has never met a compiler (no toolchain in this image): bindings/rust/fork/*.rs and
bindings/rust/examples/dump_golden.rs are hand-written text, only src/ffi.rs is generated.  The one person who can pin row
parity -- a maintainer with cargo and the dusk crates -- should not hit typos first.  What can be checked without a compiler:

  * every file tokenises (comments, strings, raw strings, char literals vs lifetimes) and its (), [], {} balance;
  * every `pg_*` symbol the hand-written files call is declared in the generated src/ffi.rs, with the same number of arguments;
    every PG_* constant and Pg* type they name exists there too;
  * examples/dump_golden.rs builds exactly the eight circuits tests/golden/make_golden.py freezes: the same names as the .npz files,
    and THE SAME INPUTS -- the Rust expressions are evaluated (s(x), pow2(k), q_minus(k), BlsScalar::one()/zero(), + and - in the
    field) and compared with the input arrays stored in the fixtures; the whole-composer circuit is compared call for call with
    tests/refcases.py:full_circuit.
"""
import os
import re

import numpy as np
import pytest

from oracle.model import Q, mont_limbs
from tests.refcases import MAX_BOUND_CASES, MAYBE_EQUAL_CASES, RANGE_CHECK_CASES, full_circuit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUST = os.path.join(ROOT, "bindings", "rust")
GOLD = os.path.join(ROOT, "tests", "golden")
HAND_WRITTEN = ["fork/hip_composer.rs", "fork/dump_columns.rs", "examples/dump_golden.rs"]
ALL_RS = HAND_WRITTEN + ["src/ffi.rs", "build.rs"]


def read(rel):
    return open(os.path.join(RUST, rel)).read()


# ---- a tokeniser that knows what is code ---------------------------------------------------------------------------------

def strip_non_code(src: str, where: str) -> str:
    """the source with comments, string / byte-string / raw-string and char literals blanked out (same length, newlines kept);
    raises on an unterminated one"""
    out, i, n = [], 0, len(src)

    def blank(j):
        out.append("".join(c if c == "\n" else " " for c in src[i:j]))

    while i < n:
        c = src[i]
        if src.startswith("//", i):
            j = src.find("\n", i)
            j = n if j < 0 else j
            blank(j)
            i = j
        elif src.startswith("/*", i):
            depth, j = 1, i + 2
            while depth and j < n:  # Rust block comments nest
                if src.startswith("/*", j):
                    depth, j = depth + 1, j + 2
                elif src.startswith("*/", j):
                    depth, j = depth - 1, j + 2
                else:
                    j += 1
            assert depth == 0, f"{where}: unterminated block comment at offset {i}"
            blank(j)
            i = j
        elif c == '"' or (c == "b" and src.startswith('b"', i)):
            j = i + (2 if c == "b" else 1)
            while j < n and src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            assert j < n, f"{where}: unterminated string literal at line {src.count(chr(10), 0, i) + 1}"
            blank(j + 1)
            i = j + 1
        elif c == "r" and re.match(r'r#*"', src[i:]) and (i == 0 or not (src[i - 1].isalnum() or src[i - 1] == "_")):
            hashes = len(re.match(r"r(#*)", src[i:]).group(1))
            end = src.find('"' + "#" * hashes, i + 2 + hashes)
            assert end >= 0, f"{where}: unterminated raw string at line {src.count(chr(10), 0, i) + 1}"
            blank(end + 1 + hashes)
            i = end + 1 + hashes
        elif c == "'":
            m = re.match(r"'(\\.[^']*|[^'\\])'", src[i:])  # a char literal; otherwise a lifetime / loop label
            if m:
                blank(i + m.end())
                i += m.end()
            else:
                out.append(c)
                i += 1
        else:
            out.append(c)
            i += 1
    return "".join(out)


@pytest.mark.parametrize("rel", ALL_RS)
def test_brackets_balance(rel):
    code = strip_non_code(read(rel), rel)
    stack = []
    pairs = {")": "(", "]": "[", "}": "{"}
    for k, ch in enumerate(code):
        if ch in "([{":
            stack.append((ch, k))
        elif ch in ")]}":
            line = code.count("\n", 0, k) + 1
            assert stack, f"{rel}:{line}: '{ch}' closes nothing"
            op, at = stack.pop()
            assert op == pairs[ch], f"{rel}:{line}: '{ch}' closes the '{op}' of line {code.count(chr(10), 0, at) + 1}"
    assert not stack, f"{rel}: '{stack[-1][0]}' of line {code.count(chr(10), 0, stack[-1][1]) + 1} is never closed"
    assert code.count(";") > 3  # (the blanking did not eat the file)


def test_the_tokeniser_itself():
    src = 'fn f<\'a>(x: &\'a str) -> char { let s = "}\\"{"; let r = r#"("#; /* { /* nested ( */ */ \'}\' } // )'
    code = strip_non_code(src, "self-test")
    assert code.count("{") == code.count("}") == 1 and code.count("(") == code.count(")") == 1
    with pytest.raises(AssertionError):
        strip_non_code('let s = "never closed;', "self-test")


# ---- symbols: what the hand-written files call must be what the generated block declares -----------------------------------

def split_args(text: str):
    """top-level comma-separated pieces of an argument list (text between the outer parentheses)"""
    parts, depth, cur = [], 0, []
    for ch in text:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
    last = "".join(cur).strip()
    if last:
        parts.append(last)
    return parts


def call_sites(code: str, pattern: str):
    """(name, [arguments], line) of every call `name(...)` whose name matches pattern"""
    for m in re.finditer(r"\b(" + pattern + r")\s*\(", code):
        depth, j = 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(code[j], 0)
            j += 1
        yield m.group(1), split_args(code[m.end():j - 1]), code.count("\n", 0, m.start()) + 1


def ffi_declarations():
    code = strip_non_code(read("src/ffi.rs"), "src/ffi.rs")
    fns = {}
    for m in re.finditer(r"pub fn (pg_\w+)\s*\(", code):
        depth, j = 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(code[j], 0)
            j += 1
        fns[m.group(1)] = len(split_args(code[m.end():j - 1]))
    consts = set(re.findall(r"pub const (PG_\w+)", code))
    types = set(re.findall(r"pub (?:struct|type) (Pg\w+)", code))
    return fns, consts, types


def test_ffi_block_is_what_the_header_exports():
    fns, consts, types = ffi_declarations()
    import subprocess
    lib = os.path.join(ROOT, "plonk_gadgets_amd", "libplonk_gadgets_hip.so")
    if os.path.exists(lib):  # (built by __graft_entry__.build(); tests/test_capi_exports.py keeps header, library and ffi.rs in step)
        exported = {ln.split()[-1] for ln in subprocess.check_output(["nm", "-D", "--defined-only", lib], text=True).splitlines()
                    if " T " in ln and ln.split()[-1].startswith("pg_")}
        assert set(fns) == exported, (sorted(set(fns) ^ exported))
    assert len(fns) > 100 and {"PG_OK", "PG_ERR_NON_EXISTING_INVERSE"} <= consts and {"PgScalar", "PgComposer", "PgEngine"} <= types


@pytest.mark.parametrize("rel", HAND_WRITTEN)
def test_hand_written_files_call_what_is_declared(rel):
    fns, consts, types = ffi_declarations()
    code = strip_non_code(read(rel), rel)
    used = list(call_sites(code, r"pg_\w+"))
    if rel == "fork/hip_composer.rs":
        assert len(used) >= 14  # engine, composer, the seven gadget signatures
    for name, args, line in used:
        assert name in fns, f"{rel}:{line}: {name} is not declared in src/ffi.rs"
        assert len(args) == fns[name], f"{rel}:{line}: {name} called with {len(args)} arguments, src/ffi.rs declares {fns[name]}: {args}"
    for name in set(re.findall(r"\bPG_[A-Z_]+\b", code)):
        assert name in consts, f"{rel}: constant {name} is not in src/ffi.rs"
    for name in set(re.findall(r"\bPg[A-Z]\w+\b", code)):
        assert name in types, f"{rel}: type {name} is not in src/ffi.rs"


def test_hip_composer_mirrors_the_seven_reference_signatures():
    """names and argument order of src/lib.rs:42-45's re-exports (src/range.rs:27-32, :82-86; src/scalar.rs:21-25, :36-40, :63-67,
    :105-109; src/allocated_scalar.rs:27), with HipComposer where the reference takes StandardComposer"""
    code = strip_non_code(read("fork/hip_composer.rs"), "hip_composer.rs")
    sigs = {m.group(1): [a.split(":")[0].strip() for a in split_args(m.group(2))]
            for m in re.finditer(r"pub fn (\w+)\s*\(([^)]*)\)", code)}
    assert sigs["range_check"] == ["composer", "min_range", "max_range", "witness"]
    assert sigs["max_bound"] == ["composer", "max_range", "witness"]
    assert sigs["conditionally_select_zero"] == ["composer", "x", "select"]
    assert sigs["conditionally_select_one"] == ["composer", "y", "selector"]
    assert sigs["is_non_zero"] == ["composer", "var", "value_assigned"]
    assert sigs["maybe_equal"] == ["composer", "a", "b"]
    assert sigs["allocate"] == ["composer", "scalar"]
    assert re.search(r"pub fn is_non_zero[^{]*->\s*Result<\(\),\s*HipError>", code)
    assert re.search(r"pub fn max_bound[^{]*->\s*\(Variable,\s*u64\)", code)


# ---- dump_golden.rs builds the circuits of tests/golden/ -------------------------------------------------------------------

def rust_value(expr: str) -> int:
    """value in the field of one of dump_golden.rs's scalar expressions"""
    e = re.sub(r"(\d)(?:u128|u64|usize)\b", r"\1", expr.strip())
    e = e.replace("BlsScalar::one()", "one()").replace("BlsScalar::zero()", "zero()")
    assert re.fullmatch(r"[\w\s()+\-,]*", e), expr
    env = {"s": lambda x: x % Q, "pow2": lambda k: pow(2, k, Q), "q_minus": lambda k: (Q - k) % Q, "one": lambda: 1, "zero": lambda: 0,
           "__builtins__": {}}
    return eval(e, env) % Q  # noqa: S307 -- the alphabet is checked above, the names are these five


def limbs_of(values):
    return np.array([mont_limbs(v) for v in values], dtype=np.uint64).reshape(-1, 4)


def bracket_list(text: str):
    """elements of the first [...] in text"""
    i = text.index("[")
    depth, j = 1, i + 1
    while depth:
        depth += {"[": 1, "]": -1}.get(text[j], 0)
        j += 1
    return split_args(text[i + 1:j - 1])


def tuple_items(el: str):
    assert el.startswith("(") and el.endswith(")"), el
    return split_args(el[1:-1])


@pytest.fixture(scope="module")
def dump():
    return strip_non_code(read("examples/dump_golden.rs"), "dump_golden.rs")


def block_before(code: str, name: str) -> str:
    """the code (literals blanked out) from `let mut c = StandardComposer::new();` to the dump_columns call that names `name`
    (offsets are those of the raw source: the blanking keeps lengths)"""
    raw = read("examples/dump_golden.rs")
    at = raw.index('c.dump_columns(&dir.join("%s"))' % name)
    start = code.rindex("let mut c = StandardComposer::new();", 0, at)
    return code[start:at]


def test_dump_golden_names_are_the_fixture_names(dump):
    src = read("examples/dump_golden.rs")
    names = set(re.findall(r'range_cases\(&dir,\s*"(\w+)"', src)) | set(re.findall(r'dir\.join\("(\w+)"\)', src))
    fixtures = {f[:-4] for f in os.listdir(GOLD) if f.endswith(".npz")}
    assert names == fixtures and len(names) == 8
    assert f"dumped {len(names)} circuits" in src
    made = set(re.findall(r'"(\w+)\.npz"', open(os.path.join(GOLD, "make_golden.py")).read())) | \
        set(re.findall(r'\("(range_check_\w+)"', open(os.path.join(GOLD, "make_golden.py")).read()))
    assert made == fixtures


def test_dump_golden_range_check_inputs(dump):
    ws_def = re.search(r"let ws: Vec<BlsScalar> = (\[[^\]]*\])", dump).group(1)
    ws = [rust_value("s(%s)" % w) for w in bracket_list(ws_def)]
    seen = {}
    for name, args, _ in call_sites(dump, "range_cases"):
        if args[0] != "&dir":
            continue  # (the function's own definition)
        wit = ws if args[4] == "&ws" else [rust_value(w) for w in bracket_list(args[4])]
        seen[read("examples/dump_golden.rs").split("range_cases(&dir,")[len(seen) + 1].split('"')[1]] = (rust_value(args[2]), rust_value(args[3]), wit)
    assert set(seen) == {"range_check_ref_50k_250k", "range_check_ref_2p126_2p127", "range_check_c1_n65", "range_check_c2_n255"}
    for name, (mn, mx, wit) in seen.items():
        g = np.load(os.path.join(GOLD, name + ".npz"))
        assert np.array_equal(g["min_range"], limbs_of([mn])) and np.array_equal(g["max_range"], limbs_of([mx])), name
        assert np.array_equal(g["witness"], limbs_of(wit)), name
    # and those are the reference's cases (tests/range_gadgets_tests.rs:120-169)
    assert seen["range_check_ref_50k_250k"][2] == [c[2] for c in RANGE_CHECK_CASES if c[0] == 50_000]
    assert seen["range_check_ref_2p126_2p127"] == (2**126, 2**127 + 1, [2**127 - 1])


def test_dump_golden_max_bound_maybe_equal_and_mix_inputs(dump):
    blk = block_before(dump, "max_bound_ref")
    pairs = [tuple(rust_value(x) for x in tuple_items(el)) for el in bracket_list(blk[blk.index("for (max, w) in"):])]
    assert pairs == [(c[0], c[1]) for c in MAX_BOUND_CASES]
    g = np.load(os.path.join(GOLD, "max_bound_ref.npz"))
    assert np.array_equal(g["max_range"], limbs_of([p[0] for p in pairs])) and np.array_equal(g["witness"], limbs_of([p[1] for p in pairs]))
    assert "RangeGadgets::max_bound(&mut c, max, a)" in blk and "AllocatedScalar::allocate(&mut c, w)" in blk

    blk = block_before(dump, "maybe_equal_ref")
    pairs = [tuple(rust_value("s(%s)" % x) for x in tuple_items(el)) for el in bracket_list(blk[blk.index("for (a, b) in"):])]
    assert pairs == [(c[0], c[1]) for c in MAYBE_EQUAL_CASES]
    g = np.load(os.path.join(GOLD, "maybe_equal_ref.npz"))
    assert np.array_equal(g["a"], limbs_of([p[0] for p in pairs])) and np.array_equal(g["b"], limbs_of([p[1] for p in pairs]))

    blk = block_before(dump, "scalar_mix")
    items = [tuple(rust_value(x) for x in tuple_items(el)) for el in bracket_list(blk[blk.index("let items ="):])]
    g = np.load(os.path.join(GOLD, "scalar_mix.npz"))
    for k, name in enumerate(("v", "y", "s", "a", "b")):
        assert np.array_equal(g[name], limbs_of([it[k] for it in items])), name
    # the order of the calls per item: three add_input, two allocate, is_non_zero, select_one, maybe_equal (make_golden.py section 4)
    body = blk[blk.index("for (v, y, sel, a, b) in items"):]
    order = [m.group(0) for m in re.finditer(r"c\.add_input|AllocatedScalar::allocate|ScalarGadgets::\w+", body)]
    assert order == ["c.add_input"] * 3 + ["AllocatedScalar::allocate"] * 2 + ["ScalarGadgets::is_non_zero", "ScalarGadgets::conditionally_select_one",
                                                                               "ScalarGadgets::maybe_equal"]
    assert g["err_mask"].tolist() == [int(it[0] == 0) for it in items]


class Recorder:
    """tests/refcases.py:full_circuit as a list of calls with their scalar arguments"""

    def __init__(self):
        self.calls, self.k = [], 0

    def _var(self):
        self.k += 1
        return "var%d" % self.k

    def add_input(self, v):
        self.calls.append(("add_input", v))
        return self._var()

    def allocate(self, v):
        self.calls.append(("allocate", v))
        return self._var()

    def range_check_loop(self, mn, mx, ws):
        out = []
        for w in ws:
            self.calls += [("allocate", w), ("range_check", mn, mx)]
            out.append(self._var())
        return out

    def max_bound(self, mx, a):
        self.calls.append(("max_bound", mx))
        return self._var()

    def maybe_equal(self, a, b):
        self.calls.append(("maybe_equal",))
        return self._var()

    def is_non_zero(self, var, value):
        self.calls.append(("is_non_zero", value))

    def conditionally_select_one(self, y, s):
        self.calls.append(("conditionally_select_one",))
        return self._var()

    def conditionally_select_zero(self, x, s):
        self.calls.append(("conditionally_select_zero",))
        return self._var()

    def constrain_to_constant(self, a, c, pi):
        self.calls.append(("constrain_to_constant", c % Q, None if pi is None else pi % Q))

    def boolean_gate(self, a):
        self.calls.append(("boolean_gate",))


def test_dump_golden_whole_composer_is_full_circuit_call_for_call(dump):
    rec = Recorder()
    full_circuit(rec)
    blk = block_before(dump, "composer_full")
    loop = re.search(r"for w in (\[[^\]]*\])\s*\{", blk)
    ws = [rust_value("s(%s)" % w) for w in bracket_list(loop.group(1))]
    depth, j = 1, loop.end()
    while depth:
        depth += {"{": 1, "}": -1}.get(blk[j], 0)
        j += 1
    body, before, after = blk[loop.end():j - 1], blk[:loop.start()], blk[j:]

    def calls_of(text, w=None):
        out = []
        pat = r"c\.add_input|AllocatedScalar::allocate|RangeGadgets::\w+|ScalarGadgets::\w+|c\.constrain_to_constant|c\.boolean_gate"
        for name, args, _ in call_sites(text, pat):
            val = lambda e: w if e == "s(w)" else rust_value(e)  # noqa: E731
            if name == "c.add_input":
                out.append(("add_input", val(args[0])))
            elif name == "AllocatedScalar::allocate":
                out.append(("allocate", val(args[1])))
            elif name == "RangeGadgets::range_check":
                out.append(("range_check", val(args[1]), val(args[2])))
            elif name == "RangeGadgets::max_bound":
                out.append(("max_bound", val(args[1])))
            elif name == "ScalarGadgets::is_non_zero":
                out.append(("is_non_zero", val(args[2])))
            elif name == "c.constrain_to_constant":
                pi = None if args[2] == "None" else rust_value(re.fullmatch(r"Some\((.*)\)", args[2]).group(1))
                out.append(("constrain_to_constant", val(args[1]), pi))
            elif name == "c.boolean_gate":
                out.append(("boolean_gate",))
            else:
                out.append((name.split("::")[1],))
        return out
    rust_calls = calls_of(before) + [c for w in ws for c in calls_of(body, w)] + calls_of(after)
    assert rust_calls == rec.calls, [(a, b) for a, b in zip(rust_calls, rec.calls) if a != b][:3]
    assert len(rust_calls) == len(rec.calls) == 20

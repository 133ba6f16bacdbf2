"""CPU: the C-ABI library loads and exports every symbol include/plonk_gadgets_hip.h declares (no compute)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "plonk_gadgets_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from plonk_gadgets_amd import _lib
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 28
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    # and the Python binding table covers the header exactly
    assert sorted(_lib.SIGNATURES) == names


def test_host_scalar_helpers_and_layouts():
    """host-only entry points: no GPU needed"""
    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import _lib
    from oracle.model import Q, mont_limbs, num_bits_closest_power_of_two
    lib = _lib.load()
    assert lib.pg_build_arch() == b"gfx950"
    for x in (0, 1, 2, 2**64, 2**254 - 1, Q - 1):
        assert pg.BlsScalar.from_int(x).limbs() == mont_limbs(x)
        assert pg.BlsScalar.from_int(x).to_int() == x
        assert pg.num_bits_closest_power_of_two(pg.BlsScalar.from_int(x)) == num_bits_closest_power_of_two(x)
    a, b = pg.BlsScalar.from_int(12345), pg.BlsScalar.from_int(Q - 7)
    assert (a + b).to_int() == 12338 and (a - b).to_int() == 12352 and (a * b).to_int() == (12345 * (Q - 7)) % Q
    assert (-a).to_int() == Q - 12345
    # the reference's only pure KAT (src/range.rs:196-203) through the product's own host code
    assert pg.bits_count(pg.BlsScalar.zero()) == 1 and pg.bits_count(pg.BlsScalar.one()) == 1
    assert pg.bits_count(pg.BlsScalar.from_u64(3)) == 2 and pg.bits_count(pg.BlsScalar.pow_of_2(128)) == 129
    lay = _lib.LayoutC()
    mn, mx = pg.BlsScalar.from_int(0), pg.BlsScalar.from_int(2**254)
    assert lib.pg_range_check_layout(C.byref(mn.c), C.byref(mx.c), 1 << 20, C.byref(lay)) == 0
    assert (lay.num_bits, lay.gates_per_item, lay.vars_per_item) == (255, 1031, 1034)
    assert lay.n_gates == 1031 << 20 and lay.n_vars == 1034 << 20
    assert lib.pg_max_bound_layout(C.byref(pg.BlsScalar.from_int(200).c), 3, C.byref(lay)) == 0
    assert (lay.num_bits, lay.gates_per_item, lay.vars_per_item, lay.n_gates) == (9, 23, 271, 69)
    # a non-reduced "scalar" is rejected, not computed with
    bad = _lib.Scalar.of([2**64 - 1] * 4)
    assert lib.pg_range_check_layout(C.byref(mn.c), C.byref(bad), 1, C.byref(lay)) == 2
    assert b"reduced" in lib.pg_last_error()


def test_columns_slab_layout():
    """pg_columns_slab_layout (host arithmetic): the nine arrays of a circuit in one block -- disjoint, inside it, on 2-MiB
    boundaries, the five selector columns a stride apart (at least their own size), the rest behind the last"""
    from plonk_gadgets_amd import _lib
    lib = _lib.load()
    MiB2 = 2 << 20
    for n_gates, n_vars, stride in ((10 << 20, 15 << 20, 24 << 30), (1000, 1500, 0), (1000, 1500, 5 << 20), (1 << 27, 1 << 27, 1 << 30),
                                    (0, 0, 0), (1, 1, 1)):
        off = (C.c_uint64 * 9)()
        total = C.c_uint64()
        assert lib.pg_columns_slab_layout(n_gates, n_vars, stride, off, C.byref(total)) == 0
        off = list(off)
        sizes = [n_gates * 32] * 5 + [n_gates * 8] * 3 + [n_vars * 32]
        assert all(o % MiB2 == 0 for o in off) and off[0] == 0
        iv = sorted(zip(off, sizes))
        assert all(a + n <= b for (a, n), (b, _) in zip(iv, iv[1:])) and iv[-1][0] + iv[-1][1] <= total.value
        want = max((stride + MiB2 - 1) // MiB2 * MiB2, (n_gates * 32 + MiB2 - 1) // MiB2 * MiB2)
        assert [b - a for a, b in zip(off[:5], off[1:5])] == [want] * 4
        assert off[5] >= off[4] + n_gates * 32 and off[5] < off[6] < off[7] < off[8] or n_gates == 0
    assert lib.pg_columns_slab_layout(1, 1, 0, None, C.byref(total)) == 2
    assert lib.pg_columns_slab_layout(1 << 41, 1, 0, off if False else (C.c_uint64 * 9)(), C.byref(total)) == 2


def test_engine_fails_loudly_without_gpu():
    import pytest
    import torch
    import plonk_gadgets_amd as pg
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU path"):
        pg.Engine(0)
    from plonk_gadgets_amd import _lib
    h = C.c_void_p()
    assert _lib.load().pg_engine_create(0, C.byref(h)) == 3  # PG_ERR_NO_DEVICE


def test_host_inversion_matches_fermat_and_pow():
    """pg_scalar_invert (division steps, the routine the device pre-pass runs) against a^(q-2) in the library and
    against Python's pow, on edge values and 3000 random scalars; zero -> PG_ERR_NON_EXISTING_INVERSE and 0"""
    import ctypes as C
    import random
    import plonk_gadgets_amd as pg
    from plonk_gadgets_amd import _lib
    Q = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    L = _lib.load()
    rnd = random.Random(11)
    vals = [1, 2, 3, Q - 1, Q - 2, (Q + 1) // 2, 2**254, 2**255 % Q, 2**32, 2**30 - 1, 2**30, 2**60 + 1]
    vals += [rnd.randrange(1, Q) for _ in range(3000)] + [rnd.randrange(1, 2**64) for _ in range(200)]
    out = _lib.Scalar()
    for x in vals:
        s = pg.BlsScalar.from_int(x)
        assert s.invert().to_int() == pow(x, -1, Q), hex(x)
        assert (s * s.invert()).to_int() == 1
    for x in vals[:300]:
        s = pg.BlsScalar.from_int(x)
        assert L.pg_scalar_invert_fermat(C.byref(s.c), C.byref(out)) == 0 and pg.BlsScalar(out) == s.invert()
    z = pg.BlsScalar.from_int(0)
    assert z.invert() is None
    assert L.pg_scalar_invert(C.byref(z.c), C.byref(out)) == 1 and pg.BlsScalar(out).to_int() == 0


def test_generated_rust_declarations_cover_the_header():
    """bindings/rust/src/ffi.rs (tools/gen_rust_ffi.py; not compiled here -- no Rust toolchain) is current and declares
    every function of the header exactly once"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_ffi as g
    text, names = g.generate()
    assert sorted(names) == declared_functions() and len(set(names)) == len(names)
    assert open(os.path.join(ROOT, "bindings", "rust", "src", "ffi.rs")).read() == text, "run python tools/gen_rust_ffi.py"
    for n in names:
        assert len(re.findall(r"\bpub fn %s\(" % n, text)) == 1

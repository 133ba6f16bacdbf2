"""BlsScalar on the host: a thin value type over the library's pg_scalar helpers (Montgomery limbs).

Mirrors the constructors/operators the reference's callers use to form public arguments:
BlsScalar::from(u64), one(), zero(), -, +, *, neg, pow of two (tests/range_gadgets_tests.rs:58-76,121-168)."""
from __future__ import annotations

import ctypes as C

from . import _lib

Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


class BlsScalar:
    __slots__ = ("c",)

    def __init__(self, c: _lib.Scalar):
        self.c = c

    # -- constructors --------------------------------------------------
    @staticmethod
    def from_u64(v: int) -> "BlsScalar":
        out = _lib.Scalar()
        _lib.load().pg_scalar_from_u64(v, C.byref(out))
        return BlsScalar(out)

    @staticmethod
    def from_int(x: int) -> "BlsScalar":
        """canonical integer (reduced mod q on the way in) -> scalar"""
        x %= Q
        raw = (C.c_uint64 * 4)(*[(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)])
        out = _lib.Scalar()
        _lib.load().pg_scalar_from_canonical(raw, C.byref(out))
        return BlsScalar(out)

    @staticmethod
    def from_limbs(limbs) -> "BlsScalar":
        return BlsScalar(_lib.Scalar.of(limbs))

    @staticmethod
    def zero() -> "BlsScalar":
        return BlsScalar.from_u64(0)

    @staticmethod
    def one() -> "BlsScalar":
        return BlsScalar.from_u64(1)

    @staticmethod
    def pow_of_2(by: int) -> "BlsScalar":
        return BlsScalar.from_int(pow(2, by, Q))

    # -- views -----------------------------------------------------------
    def limbs(self):
        return self.c.limbs()

    def to_int(self) -> int:
        raw = (C.c_uint64 * 4)()
        _lib.load().pg_scalar_to_canonical(C.byref(self.c), raw)
        return sum(int(raw[i]) << (64 * i) for i in range(4))

    # -- arithmetic --------------------------------------------------------
    def _bin(self, other, fn):
        out = _lib.Scalar()
        getattr(_lib.load(), fn)(C.byref(self.c), C.byref(other.c), C.byref(out))
        return BlsScalar(out)

    def __add__(self, o): return self._bin(o, "pg_scalar_add")
    def __sub__(self, o): return self._bin(o, "pg_scalar_sub")
    def __mul__(self, o): return self._bin(o, "pg_scalar_mul")

    def invert(self):
        """BlsScalar::invert: None for zero (the reference's CtOption)"""
        out = _lib.Scalar()
        return BlsScalar(out) if _lib.load().pg_scalar_invert(C.byref(self.c), C.byref(out)) == 0 else None

    def __neg__(self):
        out = _lib.Scalar()
        _lib.load().pg_scalar_neg(C.byref(self.c), C.byref(out))
        return BlsScalar(out)

    def __eq__(self, o):
        return isinstance(o, BlsScalar) and self.limbs() == o.limbs()

    def __hash__(self):
        return hash(tuple(self.limbs()))

    def __repr__(self):
        return f"BlsScalar({self.to_int():#x})"


def bits_count(s: BlsScalar) -> int:
    """/root/reference/src/range.rs:173-181"""
    return int(_lib.load().pg_bits_count(C.byref(s.c)))


def num_bits_closest_power_of_two(s: BlsScalar) -> int:
    """/root/reference/src/range.rs:185-189"""
    return int(_lib.load().pg_num_bits_closest_power_of_two(C.byref(s.c)))

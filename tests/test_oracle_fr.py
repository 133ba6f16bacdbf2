"""Pins oracle/fr.c against Python big-int arithmetic and the constants of SURVEY.md section 8a."""
import ctypes as C
import random

from oracle import pyoracle as po
from oracle.model import Q, R, mont_limbs

L = po.lib()


def mont_int(f):
    return sum(int(f.l[i]) << (64 * i) for i in range(4))


def test_constants():
    assert po.limbs(L.fr_from_u64(1)) == [0x00000001FFFFFFFE, 0x5884B7FA00034802, 0x998C4FEFECBC4FF5, 0x1824B159ACC5056F]
    assert po.limbs(L.fr_neg(L.fr_from_u64(1))) == [0xFFFFFFFD00000003, 0xFB38EC08FFFB13FC, 0x99AD88181CE5880F,
                                                    0x5BC8F5F97CD877D8]
    assert po.limbs(L.fr_from_u64(2)) == [0x00000003FFFFFFFC, 0xB1096FF400069004, 0x33189FDFD9789FEA, 0x304962B3598A0ADF]
    assert po.limbs(po.fr_from_int(2**254 - 1)) == [0x32667A617CFCA71E, 0x7124BE7B21E01406, 0xCE53D73BB7122072,
                                                    0x409FC29B1A47984A]
    assert po.limbs(L.fr_from_u64(0)) == [0, 0, 0, 0]


def test_random_ops_vs_bigint():
    rng = random.Random(1234)
    edge = [0, 1, 2, Q - 1, Q - 2, (Q - 1) // 2, 2**64 - 1, 2**64, 2**128, 2**192, 2**254, 2**255 % Q, R, Q - R]
    vals = edge + [rng.randrange(Q) for _ in range(300)]
    for i in range(len(vals)):
        a, b = vals[i], vals[(i * 7 + 3) % len(vals)]
        fa, fb = po.fr_from_int(a), po.fr_from_int(b)
        assert po.limbs(fa) == mont_limbs(a)
        assert po.limbs(L.fr_add(fa, fb)) == mont_limbs(a + b)
        assert po.limbs(L.fr_sub(fa, fb)) == mont_limbs(a - b)
        assert po.limbs(L.fr_mul(fa, fb)) == mont_limbs(a * b)
        assert po.limbs(L.fr_square(fa)) == mont_limbs(a * a)
        assert po.limbs(L.fr_neg(fa)) == mont_limbs(-a)
        assert po.fr_to_int(fa) == a
        out = (C.c_uint8 * 32)()
        L.fr_to_bytes(fa, out)
        assert bytes(out) == a.to_bytes(32, "little")
        inv = po.Fr()
        ok = L.fr_invert(fa, C.byref(inv))
        if a == 0:
            assert ok == 0
        else:
            assert ok == 1 and po.limbs(inv) == mont_limbs(pow(a, -1, Q))


def test_pow_and_pow_of_2():
    two = L.fr_from_u64(2)
    for p in (0, 1, 2, 63, 64, 65, 127, 128, 200, 254, 255, 256, 300):
        e = (C.c_uint64 * 4)(p, 0, 0, 0)
        assert po.limbs(L.fr_pow(two, e)) == mont_limbs(pow(2, p, Q))
        assert po.limbs(L.fr_pow_of_2(p)) == mont_limbs(pow(2, p, Q))
    big = (C.c_uint64 * 4)(0x1234, 0x5678, 0x9ABC, 0x0DEF)
    e = 0x1234 | (0x5678 << 64) | (0x9ABC << 128) | (0x0DEF << 192)
    assert po.limbs(L.fr_pow(po.fr_from_int(7), big)) == mont_limbs(pow(7, e, Q))


def test_bits_count_reference_kat():
    """the reference's only pure-arithmetic KAT: /root/reference/src/range.rs:196-203"""
    assert L.bits_count(L.fr_from_u64(0)) == 1
    assert L.bits_count(L.fr_from_u64(1)) == 1
    assert L.bits_count(L.fr_from_u64(3)) == 2
    e = (C.c_uint64 * 4)(128, 0, 0, 0)
    assert L.bits_count(L.fr_pow(L.fr_from_u64(2), e)) == 129


def test_num_bits_closest_power_of_two():
    from oracle import model
    rng = random.Random(7)
    cases = [0, 1, 2, 3, 4, 2**64 - 1, 2**64, 2**127, 2**128 - 2, 2**253, 2**254 - 1, 2**254, Q - 1] + \
            [rng.randrange(2**k) for k in range(1, 256, 9)]
    for s in cases:
        got = L.num_bits_closest_power_of_two(po.fr_from_int(s))
        assert got == model.num_bits_closest_power_of_two(s)
        bl = max(1, (s % Q).bit_length())
        assert got == (252 if bl == 255 else bl + 1)

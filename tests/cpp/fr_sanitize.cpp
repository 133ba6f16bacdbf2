// tests/cpp/fr_sanitize.cpp -- the host build of csrc/fr.hpp (the code the device runs, compiled by g++) under
// -fsanitize=undefined,address: Montgomery multiplication, both inversions, conversions over 200 000 values.
// Built and run by tests/test_fr_host_sanitizers.py (CPU suite).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "fr.hpp"
using namespace pg;
static uint64_t s = 0x9e3779b97f4a7c15ull;
static uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main() {
    int bad = 0;
    for (int i = 0; i < 200000; i++) {
        Fr raw{{rnd(), rnd(), rnd(), rnd() % 0x73eda753299d7d48ull}};
        if (i < 64) raw = Fr{{(uint64_t)i, 0, 0, 0}};
        if (i >= 64 && i < 128) raw = Fr{{0, 0, 0, 1ull << (i - 64) % 62}};
        const Fr a = fr_to_mont(raw);
        const Fr inv = fr_invert_or_zero(a);
        if (fr_is_zero(a)) { bad += !fr_is_zero(inv); continue; }
        bad += !fr_eq(fr_mul(a, inv), fr_one());
        if (i % 64 == 0) bad += !fr_eq(inv, fr_invert_fermat(a));
        bad += !fr_eq(fr_from_mont(a), raw);
        (void)bits_count(a);
    }
    std::printf("bad = %d\n", bad);
    return bad != 0;
}

// printf("%.9g") of a float, by integer arithmetic: the bytes dge_write_vec puts out for every vector element (J/DeepWalk.java:134-140 writes the vectors as text;
// nine significant digits give every float32 back).  A float is m * 2^q with a 24-bit m, so m * 10^k fits 128 bits for every k the range below needs and the
// rounding to nine digits is exact (half to even on the exact binary value, as glibc's printf rounds); values outside ~[6e-11, 8e9), zeros, infinities and NaNs
// take the caller's slow path.  Checked against snprintf on every exponent and 2e8 random floats (dge_selftest_fmt_g9, tests/test_vec_format.py).
#pragma once
#include <cstdint>
#include <cstring>

// returns the end of the text written at p (at most 16 characters), or nullptr: not a value of the fast range
static inline char* dge_fmt_g9(float f, char* p) {
    uint32_t u; memcpy(&u, &f, 4);
    const uint32_t be = (u >> 23) & 0xFFu;
    if (be < 127 - 34 || be > 127 + 33) return nullptr;                 // |f| < ~5.8e-11 (zero and denormals too) or >= ~8.6e9, inf, nan
    const uint64_t m = (u & 0x7FFFFFu) | 0x800000u;
    const int q = (int)be - 150;                                        // |f| = m * 2^q
    static const uint64_t P10[20] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull, 1000000000ull, 10000000000ull,
                                     100000000000ull, 1000000000000ull, 10000000000000ull, 100000000000000ull, 1000000000000000ull, 10000000000000000ull,
                                     100000000000000000ull, 1000000000000000000ull, 10000000000000000000ull};
    int E = (((int)be - 127) * 1233) >> 12;                             // floor(log10(2^e)) within one
    uint64_t N = 0;
    for (int tries = 0; tries < 3; tries++) {
        const int k = 8 - E;                                            // N = round(m * 2^q * 10^k), half to even
        if (k > 19 || k < -2) return nullptr;
        unsigned __int128 num = (unsigned __int128)m, n;
        if (k >= 0 && q < 0) {                                          // the common case: the divisor is a power of two
            num *= P10[k];
            const int sft = -q;                                         // 1 .. 57
            n = num >> sft;
            const unsigned __int128 r = num & (((unsigned __int128)1 << sft) - 1), half = (unsigned __int128)1 << (sft - 1);
            if (r > half || (r == half && (n & 1))) n++;
        } else {
            unsigned __int128 den = 1;
            if (k >= 0) num *= P10[k]; else den = P10[-k];
            if (q >= 0) num <<= q; else den <<= -q;                     // (den = 10^-k or 10^-k * 2^-q: far below 2^128 in the range admitted)
            n = num / den;
            const unsigned __int128 r = num - n * den;
            if (2 * r > den || (2 * r == den && (n & 1))) n++;
        }
        N = (uint64_t)n;
        if (N >= 1000000000ull) { E++; continue; }
        if (N < 100000000ull) { E--; continue; }
        break;
    }
    if (N >= 1000000000ull || N < 100000000ull) return nullptr;
    if (u >> 31) *p++ = '-';
    char d[9];
    { uint32_t t = (uint32_t)N; for (int i = 8; i >= 0; i--) { d[i] = (char)('0' + t % 10u); t /= 10u; } }
    int nd = 9; while (nd > 1 && d[nd - 1] == '0') nd--;               // %g: trailing zeros go
    if (E >= -4 && E < 9) {
        if (E >= 0) {
            for (int i = 0; i <= E; i++) *p++ = d[i];                   // (digits behind nd are zeros: d[] holds them)
            if (nd > E + 1) { *p++ = '.'; for (int i = E + 1; i < nd; i++) *p++ = d[i]; }
        } else {
            *p++ = '0'; *p++ = '.';
            for (int i = 0; i < -E - 1; i++) *p++ = '0';
            for (int i = 0; i < nd; i++) *p++ = d[i];
        }
    } else {
        *p++ = d[0];
        if (nd > 1) { *p++ = '.'; for (int i = 1; i < nd; i++) *p++ = d[i]; }
        *p++ = 'e';
        int x = E; if (x < 0) { *p++ = '-'; x = -x; } else *p++ = '+';
        *p++ = (char)('0' + x / 10); *p++ = (char)('0' + x % 10);
    }
    return p;
}

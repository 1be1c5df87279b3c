// ff_fmt_core.hpp -- one distance as the reference prints it: fmt.Fprintln(w, f) for a float64
// (frcfrc/frcfrc.go:58-62), i.e. strconv.FormatFloat(f, 'g', -1, 64) -- the shortest decimal digits that
// read back as the same binary64 (the closest to the value among them), laid out as %e when the decimal
// exponent is below -4 or at least 6 and as %f otherwise.
//
// The same code for the host (ff_format_float) and for the device (ff_kernels_fmt.hpp): the text of a
// run does not depend on where it was formatted.  The digits come from the Schubfach construction
// (R. Giulietti, "The Schubfach way to render doubles", 2020): with v = c * 2^q and k = floor(log10 2^q)
// the three integers 4v, 4v -/+ half an ulp scaled by 10^-k are computed rounded to odd from one 128-bit
// multiplication each by a tabulated g(-k) ~ 10^-k (ff_fmt_pow10.inc, exact integer arithmetic,
// tools/gen_pow10_table.py), and the shortest / closest decimal in the rounding interval is read off them.
// Checked against std::to_chars (host_selftest.cpp: edge cases + random bit patterns) and, through
// ff_format_float, against the oracle's restatement of Go's rule (tests/test_host_cpu.py).
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FF_FMT_HD __host__ __device__ inline
#else
#define FF_FMT_HD inline
#endif

namespace ff {
namespace fmt {

struct U128 {
    uint64_t hi, lo;
};

constexpr int POW10_KMIN = -292, POW10_KMAX = 324;
constexpr int MAX_CHARS = 24;  // "-1.2345678901234567e-308"

#if defined(__HIP_DEVICE_COMPILE__)
__device__ const U128 POW10_TABLE[POW10_KMAX - POW10_KMIN + 1] = {
#include "ff_fmt_pow10.inc"
};
#else
static const U128 POW10_TABLE[POW10_KMAX - POW10_KMIN + 1] = {
#include "ff_fmt_pow10.inc"
};
#endif

FF_FMT_HD uint64_t mulhi64(uint64_t a, uint64_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// floor(cp * g / 2^128) with the bits below it folded into bit 0 ("round to odd")
FF_FMT_HD uint64_t round_to_odd(U128 g, uint64_t cp)
{
    const uint64_t x_hi = mulhi64(cp, g.lo);
    const uint64_t y_lo = cp * g.hi;
    const uint64_t y0 = y_lo + x_hi;
    const uint64_t y1 = mulhi64(cp, g.hi) + (y0 < y_lo ? 1u : 0u);
    return y1 | (y0 > 1 ? 1u : 0u);
}

// A finite non-zero |f|: digits (no trailing zero), their count, and the decimal exponent of the FIRST digit.
struct Decimal {
    uint64_t digits;
    int nd, x;
};

FF_FMT_HD Decimal shortest(uint64_t bits)
{
    const uint64_t frac = bits & 0x000FFFFFFFFFFFFFull;
    const int bexp = (int)((bits >> 52) & 0x7FF);
    uint64_t c;
    int q;
    uint64_t d;
    int k;
    bool done = false;
    if (bexp != 0) {
        c = frac | 0x0010000000000000ull;
        q = bexp - 1075;
        if (0 <= -q && -q < 53 && (c & ((1ull << -q) - 1)) == 0) {  // an integer below 2^53
            d = c >> -q;
            k = 0;
            done = true;
        }
    } else {
        c = frac;
        q = -1074;
    }
    if (!done) {
        const bool even = (c & 1) == 0;
        const bool lower_closer = frac == 0 && bexp > 1;
        const uint64_t cbl = 4 * c - 2 + (lower_closer ? 1 : 0), cb = 4 * c, cbr = 4 * c + 2;
        k = (q * 1262611 - (lower_closer ? 524031 : 0)) >> 22;       // floor(log10(2^q)) / floor(log10(3/4 2^q))
        const int h = q + ((-k * 1741647) >> 19) + 1;                  // q + floor(log2(10^-k)) + 1, in [1, 4]
        const U128 g = POW10_TABLE[-k - POW10_KMIN];
        const uint64_t vbl = round_to_odd(g, cbl << h), vb = round_to_odd(g, cb << h), vbr = round_to_odd(g, cbr << h);
        const uint64_t lower = vbl + (even ? 0 : 1), upper = vbr - (even ? 0 : 1);
        const uint64_t s = vb >> 2;
        bool found = false;
        if (s >= 10) {
            const uint64_t sp = s / 10;
            const bool up_in = lower <= 40 * sp, wp_in = 40 * sp + 40 <= upper;
            if (up_in != wp_in) {
                d = sp + (wp_in ? 1 : 0);
                k += 1;
                found = true;
            }
        }
        if (!found) {
            const bool u_in = lower <= 4 * s, w_in = 4 * s + 4 <= upper;
            if (u_in != w_in) {
                d = s + (w_in ? 1 : 0);
            } else {
                const uint64_t mid = 4 * s + 2;
                const bool up = vb > mid || (vb == mid && (s & 1) != 0);
                d = s + (up ? 1 : 0);
            }
        }
    }
    while (d % 10 == 0) {  // (d > 0)
        d /= 10;
        ++k;
    }
    int nd = 1;
    for (uint64_t t = d; t >= 10; t /= 10) ++nd;
    Decimal r;
    r.digits = d;
    r.nd = nd;
    r.x = nd + k - 1;
    return r;
}

// What kind of text a value gets, and how long it is.
struct Shape {
    Decimal dec;
    int len;      // characters, sign included, no newline
    int special;  // 0: digits; 1: NaN; 2: +Inf; 3: -Inf; 4: 0; 5: -0
};

FF_FMT_HD Shape shape_of(uint64_t bits)
{
    Shape s;
    s.dec.digits = 0;
    s.dec.nd = 0;
    s.dec.x = 0;
    const bool neg = (bits >> 63) != 0;
    const uint64_t mag = bits & 0x7FFFFFFFFFFFFFFFull;
    if (mag > 0x7FF0000000000000ull) {
        s.special = 1;
        s.len = 3;
        return s;
    }
    if (mag == 0x7FF0000000000000ull) {
        s.special = neg ? 3 : 2;
        s.len = 4;
        return s;
    }
    if (mag == 0) {
        s.special = neg ? 5 : 4;
        s.len = neg ? 2 : 1;
        return s;
    }
    s.special = 0;
    s.dec = shortest(mag);
    const int nd = s.dec.nd, x = s.dec.x;
    int len = neg ? 1 : 0;
    if (x < -4 || x >= 6) {  // %e: strconv's shortest-%g rule (eprec = 6)
        const int ax = x < 0 ? -x : x;
        len += nd + (nd > 1 ? 1 : 0) + 2 + (ax < 100 ? 2 : 3);
    } else {
        const int dp = x + 1;
        len += dp <= 0 ? 2 - dp + nd : dp >= nd ? dp : nd + 1;
    }
    s.len = len;
    return s;
}

// Writes the s.len characters of the value whose shape is s.
template <typename Ptr> FF_FMT_HD void write_shape(const Shape &s, bool neg, Ptr o)
{
    if (s.special) {
        const char *t = s.special == 1 ? "NaN" : s.special == 2 ? "+Inf" : s.special == 3 ? "-Inf" : s.special == 4 ? "0" : "-0";
        for (int i = 0; i < s.len; ++i) o[i] = t[i];
        return;
    }
    const int nd = s.dec.nd, x = s.dec.x;
    int at = 0;
    if (neg) o[at++] = '-';
    uint64_t d = s.dec.digits;
    if (x < -4 || x >= 6) {
        // d[.ddd]e[+-]XX: digits from the back
        const int first = at, last = at + nd + (nd > 1 ? 1 : 0) - 1;
        for (int p = last; p > first + 1; --p) {
            o[p] = (char)('0' + (int)(d % 10));
            d /= 10;
        }
        if (nd > 1) o[first + 1] = '.';
        o[first] = (char)('0' + (int)d);
        at = last + 1;
        o[at++] = 'e';
        o[at++] = x < 0 ? '-' : '+';
        const int ax = x < 0 ? -x : x;
        if (ax >= 100) o[at++] = (char)('0' + ax / 100);
        o[at++] = (char)('0' + (ax / 10) % 10);
        o[at++] = (char)('0' + ax % 10);
        return;
    }
    const int dp = x + 1;  // digits before the decimal point
    if (dp <= 0) {
        o[at++] = '0';
        o[at++] = '.';
        for (int i = 0; i < -dp; ++i) o[at++] = '0';
        for (int p = at + nd - 1; p >= at; --p) {
            o[p] = (char)('0' + (int)(d % 10));
            d /= 10;
        }
    } else if (dp >= nd) {
        for (int p = at + nd - 1; p >= at; --p) {
            o[p] = (char)('0' + (int)(d % 10));
            d /= 10;
        }
        for (int i = nd; i < dp; ++i) o[at + i] = '0';
    } else {
        for (int p = at + nd; p > at + dp; --p) {
            o[p] = (char)('0' + (int)(d % 10));
            d /= 10;
        }
        o[at + dp] = '.';
        for (int p = at + dp - 1; p >= at; --p) {
            o[p] = (char)('0' + (int)(d % 10));
            d /= 10;
        }
    }
}

// The text of one value (no newline, no terminator); returns its length (<= MAX_CHARS).
FF_FMT_HD int format_bits(uint64_t bits, char *o)
{
    const Shape s = shape_of(bits);
    write_shape(s, (bits >> 63) != 0, o);
    return s.len;
}

}  // namespace fmt
}  // namespace ff

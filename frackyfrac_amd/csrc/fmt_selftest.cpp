// fmt_selftest.cpp -- ff_fmt_core.hpp (the formatter of the host AND of the device kernels) against
// std::to_chars' shortest round-trip digits laid out by Go's %v rule: every edge of the binary64 format,
// short decimals (the cases whose scaled value is exact), and random bit patterns.
// Usage: fmt_selftest [random patterns, default 20000000] [threads, default 4]
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "ff_fmt_core.hpp"

// the rule restated with the library's digits (what ff_format_float was until round 5)
static int reference(double f, char *buf)
{
    if (std::isnan(f)) return memcpy(buf, "NaN", 3), 3;
    if (std::isinf(f)) return memcpy(buf, f > 0 ? "+Inf" : "-Inf", 4), 4;
    char *o = buf;
    if (std::signbit(f)) {
        *o++ = '-';
        f = -f;
    }
    if (f == 0) {
        *o++ = '0';
        return (int)(o - buf);
    }
    char s[40];
    auto r = std::to_chars(s, s + sizeof s - 1, f, std::chars_format::scientific);
    *r.ptr = 0;
    const char *epos = s;
    while (epos < r.ptr && *epos != 'e') ++epos;
    char digs[24];
    int nd = 0;
    for (const char *p = s; p < epos; ++p)
        if (*p != '.') digs[nd++] = *p;
    const int x = atoi(epos + 1);
    while (nd > 1 && digs[nd - 1] == '0') --nd;
    if (x < -4 || x >= 6) {
        *o++ = digs[0];
        if (nd > 1) {
            *o++ = '.';
            memcpy(o, digs + 1, (size_t)nd - 1);
            o += nd - 1;
        }
        *o++ = 'e';
        *o++ = x < 0 ? '-' : '+';
        o += snprintf(o, 8, "%02d", x < 0 ? -x : x);
        return (int)(o - buf);
    }
    const int dp = x + 1;
    if (dp <= 0) {
        *o++ = '0';
        *o++ = '.';
        for (int i = 0; i < -dp; ++i) *o++ = '0';
        memcpy(o, digs, (size_t)nd);
        o += nd;
    } else if (dp >= nd) {
        memcpy(o, digs, (size_t)nd);
        o += nd;
        for (int i = nd; i < dp; ++i) *o++ = '0';
    } else {
        memcpy(o, digs, (size_t)dp);
        o += dp;
        *o++ = '.';
        memcpy(o, digs + dp, (size_t)(nd - dp));
        o += nd - dp;
    }
    return (int)(o - buf);
}

static std::atomic<long> fails{0}, checked{0};

static void check_bits(uint64_t bits)
{
    double d;
    memcpy(&d, &bits, 8);
    char a[48], b[48];
    const int na = ff::fmt::format_bits(bits, a), nb = reference(d, b);
    ++checked;
    if (na != nb || memcmp(a, b, (size_t)na) != 0 || na > ff::fmt::MAX_CHARS) {
        a[na] = 0;
        b[nb] = 0;
        if (fails++ < 20) fprintf(stderr, "MISMATCH bits %016llx: core \"%s\" reference \"%s\"\n", (unsigned long long)bits, a, b);
    }
}

static void check(double d)
{
    uint64_t bits;
    memcpy(&bits, &d, 8);
    check_bits(bits);
    check_bits(bits ^ 0x8000000000000000ull);
    check_bits(bits + 1);
    check_bits(bits - 1);
}

int main(int argc, char **argv)
{
    const long n_random = argc > 1 ? atol(argv[1]) : 20000000;
    const int n_threads = argc > 2 ? atoi(argv[2]) : 4;
    // every power of two and its neighbours; the largest / smallest of each kind
    for (int e = -1074; e <= 1023; ++e) check(std::ldexp(1.0, e));
    for (uint64_t b : {0ull, 1ull, 2ull, 0x000FFFFFFFFFFFFFull, 0x0010000000000000ull, 0x7FEFFFFFFFFFFFFFull, 0x7FF0000000000000ull,
                       0x7FF0000000000001ull, 0x7FF8000000000000ull, 0xFFF0000000000000ull, 0xFFFFFFFFFFFFFFFFull, 0x8000000000000000ull})
        check_bits(b);
    // short decimals m * 10^e (exact scaled values: where the sticky bit decides) and their neighbours
    for (int m = 1; m < 2000; ++m)
        for (int e = -330; e <= 310; ++e) {
            char t[32];
            snprintf(t, sizeof t, "%de%d", m, e);
            check(strtod(t, nullptr));
        }
    // integers around 2^53 and small ones
    for (uint64_t i = 1; i < 200000; ++i) check((double)i);
    for (int64_t i = -2000; i < 2000; ++i) check((double)((int64_t)1 << 53) + (double)i * 2);
    for (int e = 0; e < 64; ++e) check((double)(1ull << e) * 10.0), check(1e22 * (double)(e + 1)), check(1e23 * (double)(e + 1));
    // distances as the pair kernels produce them: quotients of sums
    {
        uint64_t x = 12345;
        for (int i = 0; i < 2000000; ++i) {
            x = x * 6364136223846793005ull + 1442695040888963407ull;
            const double a = (double)(x >> 33), b = (double)((x & 0xFFFFFFFF) + 1);
            check(a / (a + b));
        }
    }
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t)
        th.emplace_back([=] {
            uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(t + 1);
            for (long i = t; i < n_random; i += n_threads) {
                uint64_t z = (s += 0x9E3779B97F4A7C15ull);
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                check_bits(z ^ (z >> 31));
            }
        });
    for (auto &t : th) t.join();
    printf("%ld values checked, %ld mismatches\n", checked.load(), fails.load());
    if (fails) return 1;
    puts("fmt selftest ok");
    return 0;
}

// ff_synth.cpp -- the synthetic abundance tables of bench.py and the tests, in C so
// that any host can regenerate them and a 8192 x 50k-leaf table takes a second, not half a minute.
// The recipe (frackyfrac_amd/synth.py spells it out and holds the numpy twin the tests compare with):
//   stream(seed, k): xoshiro256** whose state words are four consecutive outputs of splitmix64
//                    started at seed XOR (0xD1B54A32D192ED03 * k); double() = (next() >> 11) * 2^-53
//   sample s = stream(seed, s + 1): per leaf (pre-order) two draws u1, u2: present iff u1 < density,
//                    count = 1 + floor(999 * u2 * u2); then one draw u3: a sample with no leaf present
//                    holds leaf floor(u3 * n_leaves) with the count that leaf drew.
#include <cmath>
#include <vector>

#include "ff_host.hpp"

namespace {

struct Xoshiro {
    uint64_t s[4];
    Xoshiro(uint64_t seed, uint64_t stream)
    {
        uint64_t st = seed ^ (0xD1B54A32D192ED03ull * stream);
        for (int k = 0; k < 4; ++k) {
            st += 0x9E3779B97F4A7C15ull;
            uint64_t z = st;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            s[k] = z ^ (z >> 31);
        }
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next()
    {
        const uint64_t res = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return res;
    }
    double dbl() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

// One sample's stream: calls emit(leaf ordinal, count) for every leaf it holds; returns how many.
template <typename Emit> int64_t run_sample(int64_t n_leaves, double density, uint64_t seed, int64_t sample, Emit emit)
{
    Xoshiro g(seed, (uint64_t)sample + 1);
    int64_t n = 0;
    std::vector<uint16_t> count;  // only needed when no leaf turns out present
    count.resize((size_t)n_leaves);
    for (int64_t k = 0; k < n_leaves; ++k) {
        const double u1 = g.dbl(), u2 = g.dbl();
        const double c = 1.0 + std::floor(999.0 * u2 * u2);
        count[(size_t)k] = (uint16_t)c;
        if (u1 < density) {
            emit(k, c);
            ++n;
        }
    }
    const double u3 = g.dbl();
    if (n == 0) {
        int64_t k = (int64_t)(u3 * (double)n_leaves);
        if (k > n_leaves - 1) k = n_leaves - 1;
        emit(k, (double)count[(size_t)k]);
        n = 1;
    }
    return n;
}

}  // namespace

extern "C" int ff_synth_counts(int64_t n_leaves, double density, uint64_t seed, int64_t sample_begin, int64_t sample_end,
                               int threads, int64_t *counts)
{
    if (n_leaves < 1 || sample_end < sample_begin || !counts) return FF_ERR_ARG;
    ff::parallel_for(sample_end - sample_begin, ff::clamp_threads(threads), [&](unsigned, int64_t a, int64_t b) {
        for (int64_t r = a; r < b; ++r)
            counts[r] = run_sample(n_leaves, density, seed, sample_begin + r, [](int64_t, double) {});
    });
    return FF_OK;
}

extern "C" int ff_synth_fill(int64_t n_leaves, double density, uint64_t seed, int64_t sample_begin, int64_t sample_end,
                             int threads, const int64_t *ptr, int64_t *leaf_ordinal, double *value)
{
    if (n_leaves < 1 || sample_end < sample_begin || !ptr || !leaf_ordinal || !value) return FF_ERR_ARG;
    ff::parallel_for(sample_end - sample_begin, ff::clamp_threads(threads), [&](unsigned, int64_t a, int64_t b) {
        for (int64_t r = a; r < b; ++r) {
            int64_t at = ptr[r];
            run_sample(n_leaves, density, seed, sample_begin + r, [&](int64_t k, double c) {
                leaf_ordinal[at] = k;
                value[at] = c;
                ++at;
            });
        }
    });
    return FF_OK;
}

// ff_dither.hpp -- the rounding offset of FIXED32 staging, one per BRANCH, shared by all samples.
//
// FIXED32 stages q_s(b) = floor(v_s(b) + u_b) with v = l_b * x_s(b) * 2^e and u_b in [0, 1) a
// hash of the branch id.  Because floor(. + u_b) is monotone, |q_i(b) - q_j(b)| =
// floor(max + u_b) - floor(min + u_b), whose mean over u_b is exactly |v_i(b) - v_j(b)|: every
// term of U(i,j) is an unbiased estimate with an error inside (-1, 1), independent from branch
// to branch WHATEVER the input looks like -- equal branch lengths and repeated counts included,
// which under round-to-nearest gave thousands of branches the same residual and an error
// growing like k instead of sqrt(k).  Samples with equal values on a branch still get equal
// integers, so identical samples are at distance exactly 0.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define FF_HOST_DEVICE __host__ __device__
#else
#define FF_HOST_DEVICE
#endif

namespace ff {

FF_HOST_DEVICE inline double branch_dither(int64_t branch)
{
    uint64_t x = (uint64_t)branch + 1u;
    x *= 0x9E3779B97F4A7C15ull;
    x ^= x >> 32;
    x *= 0xD6E8FEB86659FD93ull;
    x ^= x >> 32;
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);  // top 53 bits / 2^53
}

}  // namespace ff

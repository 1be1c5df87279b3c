// ff_kernels_finish_pair.hpp -- integer sums of ONE pair -> its distance (shared by finish_fixed32_kernel and by the
// epilogues that finish in place).  A fragment of ff_dev_run.hip: included there, once, inside its anonymous namespace.

// Integer sums -> distances (unifrac.go:169 and :204), IEEE binary64 division.
//
// U(i,j) is a sum of k <= k_i + k_j terms (k_s = flat nodes of sample s), each an unbiased
// estimate of l_b * |x_i(b) - x_j(b)| * 2^e with an error inside (-1, 1) that is independent
// from branch to branch (the per-branch offset of the staging, ff_dither.hpp).  Hoeffding's
// bound for such a sum: P(|error| >= t) <= 2 exp(-2 t^2 / k).  The denominator is binary64
// (exact_weight_kernel), so the relative error of a distance is that of U alone.  A pair whose
// U is so small that REFINE_C * sqrt(k) units could exceed 1e-6 of it -- nearly identical
// samples -- is queued for refine_exact_kernel, which recomputes it with the reference's own
// binary64 merge walk.  For every other pair the 1e-6 bar of BASELINE.json is missed with
// probability <= 2 exp(-2 * 25) = 4e-22, and half of it (what the run-time audit checks) with
// probability <= 2 exp(-12.5) per pair in the worst case of the bound, ~1e-9 for typical
// variances -- and that only for the pairs right at the threshold.
constexpr double REFINE_BAR = 1e-6;   // the relative bar of BASELINE.json ("within 1e-6 relative for weighted")
constexpr double REFINE_C = 5.0;
// HEADROOM of a pair that is not queued: (U * 1e-6 - 2) / (C sqrt(k)) >= 1 -- how many times over the rule's bound its
// numerator stands.  The pairs the statistical argument protects least are those just above 1: every run hands the
// ones below RISK_HEADROOM (up to RISK_CAP of them) to audit_risk_kernel, which computes them in binary64 and holds
// what was delivered to the audit's bar; the smallest headroom of the run is kept for the caller (ff_plan_audit_detail).
constexpr float RISK_HEADROOM = 1.25f;
constexpr unsigned long long RISK_CAP = 4096;
// slots of FinishArgs::refine_count (one allocation, reset before every run by reset_counters_kernel)
enum { CNT_QUEUED = 0, CNT_AUDIT_FAILED = 1, CNT_AUDIT_WORST = 2, CNT_RISK_FOUND = 3, CNT_RISK_CHECKED = 4, CNT_N = 5 };
// behind the counters: HEADROOM_SLOTS 32-bit words, the smallest squared headroom (float bits) seen by the waves that
// hash to each -- ONE word for all waves made finish_fixed32_kernel three times as long (every wave's request to the
// same address queues at its memory channel: 148 us instead of 45 at C3); the reader takes the minimum over the slots
constexpr int HEADROOM_SLOTS = 1024;

struct FinishArgs {
    const unsigned long long *W;   // integer column sums
    const double *wex;             // binary64 weights (null: the integer sums are exact)
    double *out;                   // the shard's distances, out[t] for local slot t
    const int32_t *n_nodes;        // flat nodes per sample; null: no refinement
    unsigned long long *refine_list, *refine_count;
    unsigned long long refine_cap;
    unsigned long long *risk_list;  // local slots of the pairs just above the refinement rule's bound (null: none kept)
    int scale_log2, weighted;
    // sparse tables (ff_kernels_low.hpp): the numerator is num + wl[i] + wl[j] - 2 mlow[t]; null: num alone
    const uint32_t *mlow, *wl;
};

// Local slot t = pair (i, j), integer numerator u, w = W[i] + W[j] (the caller has loaded them).
// h2min: the caller's running minimum of the squared headroom over the pairs it finishes (a register; the caller hands
// it to finish_note_headroom once, when it is done: a memory operation per PAIR on the one word all threads share cost
// finish_fixed32_kernel 100 of its 148 us at C3).
__device__ __forceinline__ void finish_pair_w(const FinishArgs &f, int64_t t, int64_t i, int64_t j, unsigned long long u,
                                              unsigned long long w, float &h2min)
{
    double d;
    if (u == w || !f.wex) {
        // u == w: no branch carries both samples (integer identity): exactly 1, or 0/0 = NaN
        // when both are empty (unifrac.go:169,204)
        if (f.weighted) {
            d = (double)u / (double)w;                 // numer / denom
        } else {
            const unsigned long long common = (w - u) >> 1;  // exact: w - u = 2 * common
            d = (double)u / (double)(u + common);      // result / (result + common)
        }
    } else {
        const double s = ldexp(f.wex[i] + f.wex[j], f.scale_log2);
        d = f.weighted ? (double)u / s                   // numer / denom
                       : 2.0 * (double)u / (s + (double)u);  // result / (result + common), common = (s - result) / 2
        d = fmin(d, 1.0);  // the integer numerator may pass the binary64 denominator by its rounding
    }
    f.out[t] = d;
    if (f.n_nodes && w != 0) {
        const double k = (double)f.n_nodes[i] + (double)f.n_nodes[j];
        const double a = (double)u * REFINE_BAR - 2.0;  // u * 1e-6 < C sqrt(k) + 2, without the square root
        const double c2k = REFINE_C * REFINE_C * k;
        if (a < 0.0 || a * a < c2k) {
            const unsigned long long at = atomicAdd(&f.refine_count[CNT_QUEUED], 1ull);
            if (at < f.refine_cap) f.refine_list[at] = (unsigned long long)t;
        } else if (f.risk_list) {
            const float h2 = (float)(a * a) / (float)c2k;  // headroom squared, >= 1 (single precision: it ranks pairs, no more)
            if (h2 < RISK_HEADROOM * RISK_HEADROOM) {
                const unsigned long long at = atomicAdd(&f.refine_count[CNT_RISK_FOUND], 1ull);
                if (at < RISK_CAP) f.risk_list[at] = (unsigned long long)t;
            }
            h2min = fminf(h2min, h2);
        }
    }
}

// Local slot t = pair (i, j), integer numerator u.
__device__ __forceinline__ void finish_pair(const FinishArgs &f, int64_t t, int64_t i, int64_t j, unsigned long long u, float &h2min)
{
    finish_pair_w(f, t, i, j, u, f.W[i] + f.W[j], h2min);
}

// Once per thread, behind its last pair: the smallest squared headroom of the wave's lanes that are still here into
// the wave's slot (non-negative floats order like their bit patterns).  h2min starts at +infinity, so a lane that
// finished no pair takes no part in the minimum.
__device__ __forceinline__ void finish_note_headroom(const FinishArgs &f, float h2min)
{
    if (!f.risk_list) return;
    // (lanes that left the kernel earlier are not here: their registers hold anything)
    const unsigned long long here = __ballot(1);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        const float other = __shfl_xor(h2min, m);
        h2min = fminf(h2min, ((here >> (lane ^ m)) & 1ull) ? other : INFINITY);
    }
    if (lane == __ffsll((long long)here) - 1 && h2min < INFINITY) {
        uint32_t *slots = reinterpret_cast<uint32_t *>(f.refine_count + CNT_N);
        const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        atomicMin(&slots[(wave * 2654435761u) >> 22], __float_as_uint(h2min));  // (top 10 bits: HEADROOM_SLOTS = 1024)
    }
}
static_assert(HEADROOM_SLOTS == 1024, "finish_note_headroom: ten bits of the hash");

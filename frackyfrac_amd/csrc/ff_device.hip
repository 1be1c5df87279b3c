// ff_device.hip -- the UniFrac path on gfx950 (MI355X): stage A on the device, staging
// of the flat-node vectors into dense matrices in HBM, the pair reduction kernels, and
// the plan/run C ABI around them.
//
// Replaces the per-pair merge walks unifracDistWeighted / unifracDistUnweighted
// (frcfrc/unifrac.go:144-205) and their driver unifracDists (unifrac.go:209-228).
// See DESIGN.md for the derivation; in short, with q_s(b) the staged value of
// sample s on branch b (0 where the sample has no flat node):
//
//   FIXED32  q_s(b) = round(l_b * abnd_s(b) * 2^e)   (weighted)
//            q_s(b) = k_b * [present], k_b = round(l_b * 2^e)   (unweighted)
//            U(i,j) = sum_b |q_i(b) - q_j(b)|,  W_s = sum_b q_s(b)   -- exact integers
//            weighted   d = U / (W_i + W_j)          U by v_sad_u32, one per term
//            unweighted d = U / (U + C), C = (W_i + W_j - U) / 2
//                       C = sum_b k_b [i present][j present] is a contraction: int8 MFMA
//   EXACT64  binary64 running sums over b ascending, with the reference's own
//            operations (no contraction), so every rounding is the reference's.
//
// Layouts.  v_sad_u32 path: QT[b][s], row = branch (pre-order id), column = sample, so
// that the 64 lanes of a wave read 4 x 64 consecutive samples of one branch (coalesced)
// while the other side of the pair tile -- 32 samples of the same branch -- arrives
// through the scalar cache as SGPR operands; no LDS, no cross-lane traffic.  MFMA path:
// sample-major int8 planes (a lane's 16 consecutive branches are one MFMA fragment).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "ff_dither.hpp"
#include "ff_host.hpp"
#include "ff_schedule.hpp"

namespace {

using namespace ff::sched;

// ----------------------------------------------------------------------------
// Device code (kernels live in the ff_kernels_*.hpp fragments)
// ----------------------------------------------------------------------------
#include "ff_kernels_stage.hpp"
#include "ff_kernels_pair_sad.hpp"
#include "ff_kernels_finish_pair.hpp"
#include "ff_kernels_mfma.hpp"
#include "ff_kernels_mfma_small.hpp"
#include "ff_kernels_stage_a.hpp"
#include "ff_kernels_finish.hpp"
#include "ff_kernels_exact_unw.hpp"

}  // namespace

// ----------------------------------------------------------------------------
// Plan
// ----------------------------------------------------------------------------

struct ff_plan {
    ff_plan_info info{};
    int device = 0;
    int weighted = 0;
    // FIXED32
    uint32_t *d_QT = nullptr;
    unsigned long long *d_W = nullptr;
    uint32_t *d_num = nullptr;
    int n_planes = 1;            // planes of accumulators in d_num (the ranges of a split tile own one each)
    int64_t plane_stride = 0;
    Item *d_items = nullptr;
    int32_t *d_item_ptr = nullptr;
    int32_t shard_rank = 0, shard_world = 1;
    double *d_host_out = nullptr;  // ff_plan_run_host's device buffer
    int64_t host_out_cap = 0;
    int n_workgroups = 0;
    int waves_per_wg = WAVES_PER_WG;
    size_t lds_bytes = 0;
    unsigned long long *d_stamps = nullptr;  // FF_STAMPS=1 diagnostics
    // sparse-aware variant: activity bits per (i-block, 16-row trip)
    bool sparse = false;
    uint32_t *d_arows = nullptr, *d_aptr16 = nullptr, *d_cs16 = nullptr;
    int64_t aptr_stride = 0;
    int32_t zero_row = 0;
    // refinement of nearly-equal pairs: the flat nodes stay on the device
    bool refine = false;
    int64_t *d_indptr = nullptr;
    int32_t *d_ids = nullptr;
    double *d_abnd = nullptr;
    unsigned long long *d_refine_list = nullptr;
    int32_t *d_n_nodes = nullptr;                  // flat nodes per sample (the refinement rule's k)
    unsigned long long *d_refine_count = nullptr;  // CNT_N counters of a run (ff_kernels_finish_pair.hpp: pairs queued, audit verdicts, risk list)
    unsigned long long *d_risk_list = nullptr;     // the run's pairs just above the refinement rule's bound (RISK_CAP slots)
    unsigned long long refine_cap = 0;
    double *d_wex = nullptr;          // binary64 weights of the samples (exact_weight_kernel); null: integer denominators
    int64_t *d_audit_slots = nullptr;  // run-time audit: sampled slots of the shard and their binary64 distances
    double *d_audit_exact = nullptr;
    int n_audit = 0;
    // FIXED32 unweighted on the matrix cores
    bool mfma = false;
    unsigned long long *d_Pbits = nullptr;  // presence, one 64-bit word per (64-branch slab, sample), slab-major
    int8_t *d_Kd = nullptr;                 // base-128 digits of the integer branch lengths, [digit][row]
    int8_t *d_Kt = nullptr;                 // graded staging: three signed digit planes of the rows (stage_for_mfma), or null
    int64_t m_ldb = 0, m_n8 = 0;
    int m_digits = 0;
    MItem *d_mitems = nullptr;
    int32_t *d_mitem_ptr = nullptr;
    uint32_t *d_partial = nullptr;  // small problems: private partial tiles of the ranges
    int32_t *d_ptiles = nullptr, *d_ptile_ptr = nullptr;
    int n_ptiles = 0;
    bool m_all_private = false;  // every item has a private partial tile
    bool m_any_atomic = true;    // some item adds into num[] atomically: num[] has to be zero before a run
    bool m_fused = false;        // the kernels that hold a slot's final sum write its distance (no num[] round trip, no finish launch)
    int n_mitems = 0, n_mgroups = 0;
    bool m_graded = false;       // the rows are staged graded (stage_for_mfma): sorted by length, three signed planes in d_Kt
    int m_duo_from_slab = 0;     //   the first slab from which two of them do
    bool m_small = false;        // a shard smaller than one round: pair_common_small_kernel, one launch per pass
    int n_stiles = 0;            // its 32 x 32 tiles (= workgroups)
    int64_t stile_c0 = 0;        // position of the shard's first tile in the triangle of 32 x 32 blocks
    // EXACT64
    double *d_DT = nullptr;
    double *d_len = nullptr;
    double *d_len_rows = nullptr;  // EXACT64 with compacted rows: treeDists by staged row
    XTile *d_xtiles = nullptr;
    int n_xtiles = 0;
    int x_tile_h = 0;  // EXACT64 tile height in use (0: not chosen yet)
    bool walk = false;  // FF_FLAG_UNSORTED_WALK: no staging at all, pair_walk_kernel over the flat nodes as they stand
    // EXACT64 unweighted (pair_exact_unw_kernel): presence bits, lengths by staged row, tiles
    bool xu = false;
    uint32_t *d_Xbits = nullptr;
    int64_t xu_ldx = 0;
    int xu_slabs = 0;
    XUTile *d_xutiles = nullptr;
    int n_xutiles = 0;
    // timing: one event pair per timed run since the last collect
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
};

namespace {

#define FF_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            (void)hipGetLastError(); /* do not leave it for a later call's launch check */    \
            return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: %s failed: %s", #call,           \
                            hipGetErrorString(e_));                                           \
        }                                                                                     \
    } while (0)


// The large buffers of a plan: say what did not fit and what to do about it.
#define FF_ALLOC(ptr, bytes, what)                                                                 \
    do {                                                                                           \
        hipError_t e_ = hipMalloc(&(ptr), (bytes));                                                \
        if (e_ != hipSuccess) {                                                                    \
            (void)hipGetLastError();                                                               \
            return ff::fail(FF_ERR_DEVICE, err, errlen,                                            \
                            "HIP: %s: cannot allocate %.2f GB for %s (shard %d of %d; more shards " \
                            "make it smaller)", hipGetErrorString(e_), (double)(bytes) / 1e9, what, \
                            (int)pl->shard_rank, (int)pl->shard_world);                            \
        }                                                                                          \
    } while (0)

// Device scratch that lives for one function: freed on every return path.
template <typename T> struct Scratch {
    T *p = nullptr;
    Scratch() = default;
    Scratch(const Scratch &) = delete;
    Scratch &operator=(const Scratch &) = delete;
    ~Scratch() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    hipError_t alloc(size_t count) { return hipMalloc(&p, sizeof(T) * std::max<size_t>(count, 1)); }
};

// Makes `device` current for a scope and gives the caller its own device back on every way out.
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    hipError_t enter(int device)
    {
        hipError_t e = hipGetDevice(&prev);
        if (e == hipSuccess && prev != device) {
            e = hipSetDevice(device);
            switched = e == hipSuccess;
        }
        return e;
    }
    ~DeviceScope()
    {
        if (switched) (void)hipSetDevice(prev);
    }
};

int env_int(const char *name, int dflt)
{
    const auto v = ff::tuning(name);
    if (!v || v->empty()) return dflt;
    return atoi(v->c_str());
}

// Host threads for a pass over `work` flat nodes: one per 2 M, at most 8 (and never more than the machine has).
unsigned host_threads(int64_t work)
{
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, hw), work / 2000000));
}

int validate_problem(const ff_problem *p, char *err, size_t errlen, bool unsorted_ok = false)
{
    if (!p) return ff::fail(FF_ERR_ARG, err, errlen, "null problem");
    if (p->n_samples < 0 || p->n_branches < 0 || p->n_branches > (int64_t)INT32_MAX - 64)
        return ff::fail(FF_ERR_ARG, err, errlen, "bad problem size N=%lld B=%lld",
                        (long long)p->n_samples, (long long)p->n_branches);
    if (p->n_samples > 0 && !p->indptr) return ff::fail(FF_ERR_ARG, err, errlen, "null indptr");
    if (p->n_branches > 0 && !p->branch_len) return ff::fail(FF_ERR_ARG, err, errlen, "null branch_len");
    if (p->n_samples > (int64_t)1 << 22)
        return ff::fail(FF_ERR_ARG, err, errlen, "too many samples (%lld)", (long long)p->n_samples);
    if (p->n_samples == 0) return FF_OK;
    if (p->indptr[0] != 0) return ff::fail(FF_ERR_ARG, err, errlen, "indptr[0] != 0");
    for (int64_t s = 0; s < p->n_samples; ++s)
        if (p->indptr[s + 1] < p->indptr[s])
            return ff::fail(FF_ERR_ARG, err, errlen, "indptr not monotone at sample %lld", (long long)s);
    // the flat nodes themselves: 21 M of them at C3 -- on a few host threads (this pass and the weights below were
    // 40 of the 50 ms of a one-shot ff_unifrac_dists there); the first offending sample in sample order is reported
    const unsigned nt = host_threads(p->indptr[p->n_samples]);
    std::vector<int64_t> bad_sample(nt, -1);
    std::vector<int> bad_kind(nt, 0);
    std::vector<int32_t> bad_id(nt, 0);
    ff::parallel_for(p->n_samples, nt, [&](unsigned th, int64_t s0, int64_t s1) {
        for (int64_t s = s0; s < s1 && bad_sample[th] < 0; ++s) {
            const int64_t b = p->indptr[s], e = p->indptr[s + 1];
            for (int64_t t = b; t < e; ++t) {
                const int32_t id = p->branch_id[t];
                int kind = 0;
                if (id < 0 || id >= p->n_branches) kind = 1;
                else if (!unsorted_ok && t > b && id <= p->branch_id[t - 1]) kind = 2;
                else if (!(p->abnd[t] > 0) || !std::isfinite(p->abnd[t])) kind = 3;
                if (kind) {
                    bad_sample[th] = s;
                    bad_kind[th] = kind;
                    bad_id[th] = id;
                    break;
                }
            }
        }
    });
    for (unsigned th = 0; th < nt; ++th) {  // (threads hold ascending sample ranges)
        if (bad_sample[th] < 0) continue;
        const long long s = (long long)bad_sample[th];
        if (bad_kind[th] == 1) return ff::fail(FF_ERR_ARG, err, errlen, "sample %lld: branch id %d out of range", s, bad_id[th]);
        if (bad_kind[th] == 2) return ff::fail(FF_ERR_ARG, err, errlen, "sample %lld: branch ids not strictly ascending", s);
        return ff::fail(FF_ERR_ARG, err, errlen, "sample %lld: abundance must be finite and > 0", s);
    }
    return FF_OK;
}

// The inputs of unifracDists resident on the device, plus the small host-side facts the
// staging decisions need.  Filled either from a host ff_problem (csr_from_host) or by
// stage A on the device (csr_from_leaves).
struct DeviceCsr {
    int64_t N = 0, B = 0, nnz = 0;
    int64_t *d_indptr = nullptr;
    int32_t *d_ids = nullptr;
    double *d_abnd = nullptr;
    double *d_len = nullptr;
    std::vector<int64_t> h_indptr;  // [N+1]
    std::vector<double> h_len;      // [B] treeDists
    std::vector<double> h_weight;   // [N] sum_b l_b * x_s(b)
    void release()
    {
        (void)hipFree(d_indptr);
        (void)hipFree(d_ids);
        (void)hipFree(d_abnd);
        (void)hipFree(d_len);
        d_indptr = nullptr;
        d_ids = nullptr;
        d_abnd = nullptr;
        d_len = nullptr;
    }
};

// Chooses the arithmetic and, for FIXED32, the binary scale and the integer branch lengths.
struct Quant {
    bool fixed_ok = false;
    int e = 0;
    int lengths_exact = 0;
    std::vector<uint32_t> klen;  // unweighted: round(l_b * 2^e)
    std::string why_not;
};

// (d_indptr, d_ids, d_len: the flat nodes and treeDists on the device -- the plan owns them by now)
Quant choose_quant(const DeviceCsr &c, bool weighted, const int64_t *d_indptr, const int32_t *d_ids, const double *d_abnd,
                   const double *d_len)
{
    Quant q;
    const int64_t B = c.B, N = c.N;
    for (int64_t b = 0; b < B; ++b)
        if (!std::isfinite(c.h_len[(size_t)b]) || c.h_len[(size_t)b] < 0) {
            q.why_not = "negative or non-finite branch length";
            return q;
        }
    const double LIMIT = 2147483647.0;  // every W_s must stay below 2^31 so that U < 2^32
    if (weighted) {
        double wmax = 0, wmin_pos = INFINITY;
        int64_t nnz_max = 0;
        for (int64_t s = 0; s < N; ++s) {
            const double w = c.h_weight[(size_t)s];
            if (!std::isfinite(w)) {
                q.why_not = "non-finite sample weight";
                return q;
            }
            wmax = std::max(wmax, w);
            if (w > 0) wmin_pos = std::min(wmin_pos, w);
            nnz_max = std::max(nnz_max, c.h_indptr[(size_t)s + 1] - c.h_indptr[(size_t)s]);
        }
        if (wmax == 0) {  // every distance is 0/0
            q.fixed_ok = true;
            q.e = 0;
            return q;
        }
        // a sample far lighter than the heaviest one would keep too few bits
        if (wmin_pos < wmax * 0x1p-10) {
            q.why_not = "sample weights span more than 2^10";
            return q;
        }
        int ex;
        std::frexp((LIMIT - (double)nnz_max - 2.0) / wmax, &ex);  // 2^(ex-1) <= ratio < 2^ex
        q.e = ex - 1;
        q.fixed_ok = true;
        return q;
    }
    // unweighted: smallest e making every length an integer, if every sample's sum still fits
    double lmax = 0;
    int e_exact = -2000;
    for (int64_t b = 0; b < B; ++b) {
        const double l = c.h_len[(size_t)b];
        lmax = std::max(lmax, l);
        if (l == 0) continue;
        int ex;
        const double m = std::frexp(l, &ex);  // l = m * 2^ex, 0.5 <= m < 1
        const uint64_t mi = (uint64_t)std::ldexp(m, 53);
        const int tz = __builtin_ctzll(mi);
        const int lowbit = ex - 53 + tz;  // l is a multiple of 2^lowbit
        e_exact = std::max(e_exact, -lowbit);
    }
    q.klen.assign((size_t)B, 0);
    // What has to stay below 2^31 is a SAMPLE's sum of integer lengths (U = W_i + W_j - 2 common), not the
    // tree's: a sample reaches a fraction of the tree, and the bits this leaves go to the resolution.
    // (Scaled by the tree's total, C3's shape with inexact lengths kept so few bits per pair that most
    // pairs failed the refinement rule and went to the binary64 walk: 55 ms a pass instead of 0.5.)
    double wl = 0;  // max over samples of sum_b l_b over the sample's flat nodes
    int64_t nnz_max = 0;
    if (N > 0 && c.nnz > 0) {
        double *d_w = nullptr;
        std::vector<double> hw((size_t)N);
        bool ok = hipMalloc(&d_w, sizeof(double) * (size_t)N) == hipSuccess;
        if (ok) {
            exact_weight_kernel<<<dim3((unsigned)N), dim3(256)>>>(d_indptr, d_ids, d_abnd, d_len, 0, d_w);
            ok = hipGetLastError() == hipSuccess &&
                 hipMemcpy(hw.data(), d_w, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost) == hipSuccess;
        }
        (void)hipFree(d_w);
        if (!ok) {
            q.why_not = "device error while summing the samples' branch lengths";
            return q;
        }
        for (int64_t s = 0; s < N; ++s) {
            wl = std::max(wl, hw[(size_t)s]);
            nnz_max = std::max(nnz_max, c.h_indptr[(size_t)s + 1] - c.h_indptr[(size_t)s]);
        }
    }
    if (wl == 0) {  // no sample has a branch of positive length: every distance is 0/0
        q.fixed_ok = true;
        q.e = 0;
        q.lengths_exact = 1;
        return q;
    }
    if (e_exact > -2000 && e_exact < 1000 && std::ldexp(std::max(wl, lmax), e_exact) <= LIMIT - 2.0) {
        q.e = e_exact;
        q.lengths_exact = 1;
    } else {
        // every sample's sum (each length rounded up by less than 1) below 2^31.  (Until round 3 every length was
        // also kept below 2^28 -- four base-128 digits, two sweeps of the matrix-core kernel -- which cost a tree
        // with a few very long branches most of its resolution: log-normal lengths of sigma 2.5 sent four pairs in
        // five to the binary64 walk.  Graded staging, stage_for_mfma, multiplies a long branch as several rows.)
        int ex;
        std::frexp((LIMIT - (double)nnz_max - 2.0) / wl, &ex);
        q.e = ex - 1;
        q.lengths_exact = 0;
    }
    // the branch's shared rounding offset (ff_dither.hpp); an exact length is its own integer.  (A branch no
    // sample has a flat node on may be longer than any sample's sum: its integer is never used, only kept in range.)
    for (int64_t b = 0; b < B; ++b)
        q.klen[(size_t)b] = (uint32_t)(int64_t)std::min(LIMIT, std::floor(std::ldexp(c.h_len[(size_t)b], q.e) + ff::branch_dither(b)));
    q.fixed_ok = true;
    return q;
}

void plan_free_device(ff_plan *pl)
{
    if (!pl) return;
    (void)hipFree(pl->d_QT);
    (void)hipFree(pl->d_W);
    (void)hipFree(pl->d_num);
    (void)hipFree(pl->d_items);
    (void)hipFree(pl->d_item_ptr);
    (void)hipFree(pl->d_stamps);
    (void)hipFree(pl->d_arows);
    (void)hipFree(pl->d_aptr16);
    (void)hipFree(pl->d_cs16);
    (void)hipFree(pl->d_indptr);
    (void)hipFree(pl->d_ids);
    (void)hipFree(pl->d_abnd);
    (void)hipFree(pl->d_refine_list);
    (void)hipFree(pl->d_refine_count);
    (void)hipFree(pl->d_risk_list);
    (void)hipFree(pl->d_n_nodes);
    (void)hipFree(pl->d_wex);
    (void)hipFree(pl->d_audit_slots);
    (void)hipFree(pl->d_audit_exact);
    (void)hipFree(pl->d_Pbits);
    (void)hipFree(pl->d_Kd);
    (void)hipFree(pl->d_Kt);
    (void)hipFree(pl->d_mitems);
    (void)hipFree(pl->d_mitem_ptr);
    (void)hipFree(pl->d_partial);
    (void)hipFree(pl->d_ptiles);
    (void)hipFree(pl->d_ptile_ptr);
    (void)hipFree(pl->d_DT);
    (void)hipFree(pl->d_len);
    (void)hipFree(pl->d_len_rows);
    (void)hipFree(pl->d_host_out);
    (void)hipFree(pl->d_xtiles);
    (void)hipFree(pl->d_Xbits);
    (void)hipFree(pl->d_xutiles);
    for (auto &e : pl->events) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
}

// Picks the device (it must be a gfx950) and fills the shard geometry.
int set_shard_geometry(ff_plan *pl, int32_t rank, int32_t world, char *err, size_t errlen);

int plan_begin(const ff_options *o, int64_t N, int64_t B, ff_plan *pl, hipDeviceProp_t *prop, char *err,
               size_t errlen)
{
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev <= 0)
        return ff::fail(FF_ERR_DEVICE, err, errlen,
                        "no HIP device available (%s); this engine has no CPU path",
                        he != hipSuccess ? hipGetErrorString(he) : "device count is 0");
    if (o->device >= 0) {
        if (o->device >= ndev) return ff::fail(FF_ERR_DEVICE, err, errlen, "device %d of %d", o->device, ndev);
        FF_HIP(hipSetDevice(o->device));
    }
    FF_HIP(hipGetDevice(&pl->device));
    (void)hipGetLastError();  // a stale error of the caller's (or of a failed plan) is not ours
    FF_HIP(hipGetDeviceProperties(prop, pl->device));
    if (strncmp(prop->gcnArchName, "gfx950", 6) != 0)
        return ff::fail(FF_ERR_DEVICE, err, errlen, "device %d is %s; this engine is built for gfx950 only",
                        pl->device, prop->gcnArchName);
    pl->weighted = o->weighted != 0;
    ff_plan_info &inf = pl->info;
    inf.audit_min_headroom = INFINITY;
    inf.n_samples = N;
    inf.n_branches = B;
    inf.n_compute_units = prop->multiProcessorCount;
    pl->shard_rank = o->rank;
    pl->shard_world = o->world;
    return set_shard_geometry(pl, o->rank, o->world, err, errlen);
}

// Host flat nodes -> device (the inner-seam entry: ff_plan_create / ff_unifrac_dists).
int csr_from_host(const ff_problem *p, DeviceCsr *c, char *err, size_t errlen)
{
    const int64_t N = p->n_samples, B = p->n_branches;
    c->N = N;
    c->B = B;
    c->nnz = N > 0 ? p->indptr[N] : 0;
    c->h_indptr.assign((size_t)N + 1, 0);
    if (N > 0) memcpy(c->h_indptr.data(), p->indptr, sizeof(int64_t) * (size_t)(N + 1));
    c->h_len.assign(p->branch_len, p->branch_len + B);
    c->h_weight.assign((size_t)N, 0.0);
    ff::parallel_for(N, host_threads(c->nnz), [&](unsigned, int64_t s0, int64_t s1) {
        for (int64_t s = s0; s < s1; ++s) {
            double w = 0;
            for (int64_t t = p->indptr[s]; t < p->indptr[s + 1]; ++t) w += p->branch_len[p->branch_id[t]] * p->abnd[t];
            c->h_weight[(size_t)s] = w;
        }
    });
    FF_HIP(hipMalloc(&c->d_indptr, sizeof(int64_t) * (size_t)(N + 1)));
    FF_HIP(hipMalloc(&c->d_ids, sizeof(int32_t) * (size_t)std::max<int64_t>(c->nnz, 1)));
    FF_HIP(hipMalloc(&c->d_abnd, sizeof(double) * (size_t)std::max<int64_t>(c->nnz, 1)));
    FF_HIP(hipMalloc(&c->d_len, sizeof(double) * (size_t)std::max<int64_t>(B, 1)));
    FF_HIP(hipMemcpy(c->d_indptr, c->h_indptr.data(), sizeof(int64_t) * (size_t)(N + 1), hipMemcpyHostToDevice));
    if (c->nnz > 0) {
        FF_HIP(hipMemcpy(c->d_ids, p->branch_id, sizeof(int32_t) * (size_t)c->nnz, hipMemcpyHostToDevice));
        FF_HIP(hipMemcpy(c->d_abnd, p->abnd, sizeof(double) * (size_t)c->nnz, hipMemcpyHostToDevice));
    }
    if (B > 0) FF_HIP(hipMemcpy(c->d_len, p->branch_len, sizeof(double) * (size_t)B, hipMemcpyHostToDevice));
    return FF_OK;
}

// Stage A on the device: leaf values -> flat nodes.  Returns
// FF_ERR_INTERNAL + *too_deep when the tree has more levels than it is worth launching
// kernels for (a caterpillar); the caller then flattens on the host.
int csr_from_leaves(const ff_tree *t, int64_t N, const int64_t *leaf_ptr, const int64_t *leaf_idx,
                    const double *leaf_val, bool normalize, DeviceCsr *c, bool *too_deep, char *err,
                    size_t errlen)
{
    *too_deep = false;
    const int64_t B = (int64_t)t->size.size();
    c->N = N;
    c->B = B;
    c->h_len = t->dist;
    const int64_t n_leafvals = leaf_ptr[N];
    for (int64_t k = 0; k < n_leafvals; ++k)
        if (leaf_idx[k] < 0 || leaf_idx[k] >= B)
            return ff::fail(FF_ERR_ARG, err, errlen, "leaf index %lld out of range", (long long)leaf_idx[k]);
    // The tree the sums run over: the whole tree, or -- when the samples touch less than 90 % of
    // it -- the tree induced by the leaves that carry abundance and their ancestors (W nodes,
    // pre-order kept; node_of[w] = original id).  An absent child adds +0.0 to its parent's sum,
    // which changes no bit, so the flat nodes are the same; the dense S matrix is W x N
    // instead of B x N.
    std::vector<int32_t> node_of;   // empty: identity
    std::vector<int64_t> w_parent, w_size, w_lidx;
    const int64_t *parentP = t->parent.data(), *sizeP = t->size.data(), *lidxP = leaf_idx;
    int64_t W = B;
    if (env_int("FF_COMPACT", 1) != 0 && B > 1 && n_leafvals > 0) {
        std::vector<unsigned char> used((size_t)B, 0);
        for (int64_t k = 0; k < n_leafvals; ++k)
            if (t->size[(size_t)leaf_idx[k]] == 1 && leaf_val[k] > 0) used[(size_t)leaf_idx[k]] = 1;
        int64_t cnt = 0;
        for (int64_t id = B - 1; id >= 1; --id)  // parent[id] < id
            if (used[(size_t)id]) {
                used[(size_t)t->parent[(size_t)id]] = 1;
                ++cnt;
            }
        cnt += used[0];
        if (cnt > 1 && cnt * 10 <= B * 9) {
            W = cnt;
            std::vector<int32_t> row_of((size_t)B, 0);
            node_of.reserve((size_t)W);
            for (int64_t id = 0; id < B; ++id)
                if (used[(size_t)id]) {
                    row_of[(size_t)id] = (int32_t)node_of.size();
                    node_of.push_back((int32_t)id);
                }
            w_parent.assign((size_t)W, -1);
            w_size.assign((size_t)W, 1);
            for (int64_t w = 1; w < W; ++w) w_parent[(size_t)w] = row_of[(size_t)t->parent[(size_t)node_of[(size_t)w]]];
            for (int64_t w = W - 1; w >= 1; --w) w_size[(size_t)w_parent[(size_t)w]] += w_size[(size_t)w];
            // an entry that is not a leaf with abundance goes to the root, which is internal here
            w_lidx.resize((size_t)n_leafvals);
            for (int64_t k = 0; k < n_leafvals; ++k)
                w_lidx[(size_t)k] = used[(size_t)leaf_idx[k]] && t->size[(size_t)leaf_idx[k]] == 1
                                        ? row_of[(size_t)leaf_idx[k]] : 0;
            parentP = w_parent.data();
            sizeP = w_size.data();
            lidxP = w_lidx.data();
        }
    }
    // levels: depth of every node; internal nodes grouped by level, deepest first
    std::vector<int32_t> depth((size_t)W, 0);
    int32_t max_depth = 0;
    for (int64_t id = 1; id < W; ++id) {
        depth[(size_t)id] = depth[(size_t)parentP[(size_t)id]] + 1;
        max_depth = std::max(max_depth, depth[(size_t)id]);
    }
    if (max_depth > 4096) {
        *too_deep = true;
        return FF_ERR_INTERNAL;
    }
    std::vector<int64_t> child_ptr((size_t)W + 1, 0);
    std::vector<int32_t> child_idx;
    child_idx.reserve((size_t)W);
    std::vector<std::vector<int32_t>> by_level((size_t)max_depth + 1);
    for (int64_t id = 0; id < W; ++id) {
        const int64_t end = id + sizeP[(size_t)id];
        for (int64_t ch = id + 1; ch < end; ch += sizeP[(size_t)ch]) child_idx.push_back((int32_t)ch);  // ascending
        child_ptr[(size_t)id + 1] = (int64_t)child_idx.size();
        if (sizeP[(size_t)id] > 1) by_level[(size_t)depth[(size_t)id]].push_back((int32_t)id);
    }
    std::vector<int32_t> order;
    std::vector<int> level_ptr{0};
    for (int32_t L = max_depth; L >= 0; --L) {
        order.insert(order.end(), by_level[(size_t)L].begin(), by_level[(size_t)L].end());
        level_ptr.push_back((int)order.size());
    }
    const int64_t ld = round_up(std::max<int64_t>(N, 1), 64);
    double *d_S = nullptr, *d_lval = nullptr, *d_div = nullptr, *d_weight = nullptr;
    int64_t *d_lptr = nullptr, *d_lidx = nullptr, *d_size = nullptr, *d_cptr = nullptr, *d_count = nullptr;
    int32_t *d_cidx = nullptr, *d_order = nullptr, *d_node_of = nullptr;
    auto cleanup = [&] {
        (void)hipFree(d_node_of);
        (void)hipFree(d_S); (void)hipFree(d_lval); (void)hipFree(d_div); (void)hipFree(d_weight);
        (void)hipFree(d_lptr); (void)hipFree(d_lidx); (void)hipFree(d_size); (void)hipFree(d_cptr);
        (void)hipFree(d_count); (void)hipFree(d_cidx); (void)hipFree(d_order);
    };
#define FF_HIP_C(call)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (void)hipGetLastError();                                                            \
            cleanup();                                                                          \
            return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: %s failed: %s", #call,             \
                            hipGetErrorString(e_));                                             \
        }                                                                                       \
    } while (0)
    const size_t s_bytes = sizeof(double) * (size_t)std::max<int64_t>(W, 1) * (size_t)ld;
    FF_HIP_C(hipMalloc(&d_S, s_bytes));
    FF_HIP_C(hipMemset(d_S, 0, s_bytes));
    FF_HIP_C(hipMalloc(&d_lptr, sizeof(int64_t) * (size_t)(N + 1)));
    FF_HIP_C(hipMalloc(&d_lidx, sizeof(int64_t) * (size_t)std::max<int64_t>(n_leafvals, 1)));
    FF_HIP_C(hipMalloc(&d_lval, sizeof(double) * (size_t)std::max<int64_t>(n_leafvals, 1)));
    FF_HIP_C(hipMalloc(&d_size, sizeof(int64_t) * (size_t)std::max<int64_t>(W, 1)));
    FF_HIP_C(hipMalloc(&d_cptr, sizeof(int64_t) * (size_t)(W + 1)));
    if (!node_of.empty()) {
        FF_HIP_C(hipMalloc(&d_node_of, sizeof(int32_t) * (size_t)W));
        FF_HIP_C(hipMemcpy(d_node_of, node_of.data(), sizeof(int32_t) * (size_t)W, hipMemcpyHostToDevice));
    }
    FF_HIP_C(hipMalloc(&d_cidx, sizeof(int32_t) * std::max<size_t>(child_idx.size(), 1)));
    FF_HIP_C(hipMalloc(&d_order, sizeof(int32_t) * std::max<size_t>(order.size(), 1)));
    FF_HIP_C(hipMalloc(&d_count, sizeof(int64_t) * (size_t)std::max<int64_t>(N, 1)));
    FF_HIP_C(hipMalloc(&d_div, sizeof(double) * (size_t)std::max<int64_t>(N, 1)));
    FF_HIP_C(hipMalloc(&d_weight, sizeof(double) * (size_t)std::max<int64_t>(N, 1)));
    FF_HIP_C(hipMalloc(&c->d_len, sizeof(double) * (size_t)std::max<int64_t>(B, 1)));
    FF_HIP_C(hipMemcpy(d_lptr, leaf_ptr, sizeof(int64_t) * (size_t)(N + 1), hipMemcpyHostToDevice));
    if (n_leafvals > 0) {
        FF_HIP_C(hipMemcpy(d_lidx, lidxP, sizeof(int64_t) * (size_t)n_leafvals, hipMemcpyHostToDevice));
        FF_HIP_C(hipMemcpy(d_lval, leaf_val, sizeof(double) * (size_t)n_leafvals, hipMemcpyHostToDevice));
    }
    if (B > 0) {
        FF_HIP_C(hipMemcpy(d_size, sizeP, sizeof(int64_t) * (size_t)W, hipMemcpyHostToDevice));
        FF_HIP_C(hipMemcpy(c->d_len, t->dist.data(), sizeof(double) * (size_t)B, hipMemcpyHostToDevice));
    }
    FF_HIP_C(hipMemcpy(d_cptr, child_ptr.data(), sizeof(int64_t) * (size_t)(W + 1), hipMemcpyHostToDevice));
    if (!child_idx.empty())
        FF_HIP_C(hipMemcpy(d_cidx, child_idx.data(), sizeof(int32_t) * child_idx.size(), hipMemcpyHostToDevice));
    if (!order.empty())
        FF_HIP_C(hipMemcpy(d_order, order.data(), sizeof(int32_t) * order.size(), hipMemcpyHostToDevice));
    if (N > 0 && B > 0) {
        if (n_leafvals > 0)
            stage_a_scatter_kernel<<<dim3((unsigned)N), dim3(256)>>>(d_lptr, d_lidx, d_lval, d_size, d_S, ld);
        const unsigned sblocks = (unsigned)((N + 255) / 256);
        for (size_t L = 0; L + 1 < level_ptr.size(); ++L) {
            int b0 = level_ptr[L], b1 = level_ptr[L + 1];
            while (b0 < b1) {  // grid.y is limited to 65535
                const int chunk = std::min(b1 - b0, 65535);
                stage_a_level_kernel<<<dim3(sblocks, (unsigned)chunk), dim3(256)>>>(d_order, b0, b0 + chunk, d_cptr,
                                                                                    d_cidx, d_S, ld, N);
                b0 += chunk;
            }
        }
        stage_a_count_kernel<<<dim3((unsigned)((N + 63) / 64)), dim3(64)>>>(d_S, ld, W, N, d_count, d_div);
    }
    FF_HIP_C(hipGetLastError());
    std::vector<int64_t> cnt((size_t)N, 0);
    if (N > 0 && B > 0) FF_HIP_C(hipMemcpy(cnt.data(), d_count, sizeof(int64_t) * (size_t)N, hipMemcpyDeviceToHost));
    c->h_indptr.assign((size_t)N + 1, 0);
    for (int64_t s2 = 0; s2 < N; ++s2) c->h_indptr[(size_t)s2 + 1] = c->h_indptr[(size_t)s2] + cnt[(size_t)s2];
    c->nnz = c->h_indptr[(size_t)N];
    FF_HIP_C(hipMalloc(&c->d_indptr, sizeof(int64_t) * (size_t)(N + 1)));
    FF_HIP_C(hipMalloc(&c->d_ids, sizeof(int32_t) * (size_t)std::max<int64_t>(c->nnz, 1)));
    FF_HIP_C(hipMalloc(&c->d_abnd, sizeof(double) * (size_t)std::max<int64_t>(c->nnz, 1)));
    FF_HIP_C(hipMemcpy(c->d_indptr, c->h_indptr.data(), sizeof(int64_t) * (size_t)(N + 1), hipMemcpyHostToDevice));
    c->h_weight.assign((size_t)N, 0.0);
    if (N > 0 && B > 0) {
        stage_a_fill_kernel<<<dim3((unsigned)((N + 63) / 64)), dim3(64)>>>(d_S, ld, W, N, c->d_indptr, d_div,
                                                                           normalize ? 1 : 0, c->d_len, d_node_of,
                                                                           c->d_ids, c->d_abnd, d_weight);
        FF_HIP_C(hipGetLastError());
        FF_HIP_C(hipMemcpy(c->h_weight.data(), d_weight, sizeof(double) * (size_t)N, hipMemcpyDeviceToHost));
    }
#undef FF_HIP_C
    cleanup();
    return FF_OK;
}

// ---- The shard-dependent part of a plan: work schedule and accumulators --------------------
// (rebuilt by ff_plan_set_shard; the staged matrix does not depend on the shard)

template <typename T> void free_and_null(T *&p)
{
    (void)hipFree(p);
    p = nullptr;
}

int set_shard_geometry(ff_plan *pl, int32_t rank, int32_t world, char *err, size_t errlen)
{
    ff_plan_info &inf = pl->info;
    int rc = ff_shard_rows(inf.n_samples, rank, world, &inf.row_begin, &inf.row_end);
    if (rc) return ff::fail(FF_ERR_ARG, err, errlen, "bad shard %d of %d", rank, world);
    inf.slot_begin = inf.row_begin > 0 ? inf.row_begin * (inf.row_begin - 1) / 2 : 0;
    inf.slot_end = inf.row_end > 0 ? inf.row_end * (inf.row_end - 1) / 2 : 0;
    return FF_OK;
}

int schedule_sad(ff_plan *pl, char *err, size_t errlen)
{
    ff_plan_info &inf = pl->info;
    const int64_t N = inf.n_samples, rows = inf.rows_padded, n_slots = inf.slot_end - inf.slot_begin;
    free_and_null(pl->d_items);
    free_and_null(pl->d_item_ptr);
    free_and_null(pl->d_num);
    free_and_null(pl->d_stamps);
    std::vector<Tile> tiles;
    build_tiles(N, inf.row_begin, inf.row_end, TILE_I, TILE_J, true, &tiles);
    inf.n_tiles = (int64_t)tiles.size();
    // 8 waves per workgroup (two per SIMD) for pair_sad_kernel and the sparse-aware kernel, 12 (three per SIMD,
    // paid for with half the vector prefetch) for pair_sad_kernel12.  FF_WAVES_PER_WG = 8 / 12 forces one; otherwise
    // the 12-wave variant takes
    //   * a shard that BEGINS AT ROW 0 -- a whole problem, the first rank's shard: a triangle -- and holds more than
    //     2.25 tiles per workgroup (3,072 samples up on 256 CUs): tools/shape_sweep.py, 4,096 samples 4.98 -> 4.91 ms,
    //     8,192 19.9 -> 19.2, 8,192 x 50k leaves 100.3 -> 95.9; it ties at 2,560 and loses below 2,048;
    //   * ANY shard with 200,000 or more (tile, branch row) units per workgroup -- about 11 ms of kernel: on the
    //     trapezoid of a later row shard the third wave pays once a shard is several rounds long, and not before
    //     (tools/shard_balance.py at HEAD, profiles/r04_shard_balance.txt, max over ranks in ms, 8 waves / 12 waves:
    //     C4 over 2 GPUs 40.2 / 38.5, over 4 19.9 / 19.3, over 8 10.12 / 10.42 (one rank 4 % behind the others);
    //     C5 over 2 50.3 / 48.4, over 4 25.0 / 24.5, over 8 13.06 / 12.73; the weak problem, C3's pairs per rank,
    //     5.00-5.13 / 5.04-5.26).  Round 3's rule gave the first rank alone the 12-wave kernel whatever the shard's
    //     size, and said otherwise in this comment.
    pl->waves_per_wg = pl->sparse ? WAVES_PER_WG : waves_per_wg();
    if (!pl->sparse && !ff::tuning("FF_WAVES_PER_WG").has_value() &&
        ((inf.row_begin == 0 && inf.n_tiles * 4 >= (int64_t)pl->n_workgroups * 9) ||
         (double)inf.n_tiles * (double)rows >= 200000.0 * (double)pl->n_workgroups))
        pl->waves_per_wg = L_WAVES_PER_WG;
    pl->lds_bytes = 96 * 1024;  // unused dynamic LDS sized so that exactly one workgroup fits a CU
    const int U = pl->n_workgroups * pl->waves_per_wg;
    inf.n_wave_slots = U;
    std::vector<Item> items;
    std::vector<int32_t> item_ptr;
    // up to 255 planes of accumulators (FF_PLANES; 1 = atomics only), within 1 GiB
    int max_planes = std::min(255, std::max(1, env_int("FF_PLANES", 255)));
    while (max_planes > 1 && (double)max_planes * 4.0 * (double)std::max<int64_t>(n_slots, 1) > 1073741824.0) --max_planes;
    build_schedule(tiles, rows, U, &items, &item_ptr, &inf.elements, xcd_slices(), pl->waves_per_wg,
                   max_planes > 1 ? max_planes : 0, inf.row_begin > 0);
    inf.n_items = (int64_t)items.size();
    pl->n_planes = 1;
    for (const Item &it : items) pl->n_planes = std::max(pl->n_planes, (int)((it.flags >> 3) & 255u) + 1);
    pl->plane_stride = round_up(std::max<int64_t>(n_slots, 1), FINISH_RUN);  // (planes start 16-byte aligned: finish_fixed32_kernel)
    FF_HIP(hipMalloc(&pl->d_items, sizeof(Item) * std::max<size_t>(items.size(), 1)));
    FF_HIP(hipMalloc(&pl->d_item_ptr, sizeof(int32_t) * item_ptr.size()));
    if (!items.empty())
        FF_HIP(hipMemcpy(pl->d_items, items.data(), sizeof(Item) * items.size(), hipMemcpyHostToDevice));
    FF_HIP(hipMemcpy(pl->d_item_ptr, item_ptr.data(), sizeof(int32_t) * item_ptr.size(), hipMemcpyHostToDevice));
    FF_ALLOC(pl->d_num, sizeof(uint32_t) * (size_t)pl->plane_stride * (size_t)pl->n_planes, "the pair accumulators");
    // slots of tiles that are not split that way are never written in planes 1..: zero once
    FF_HIP(hipMemset(pl->d_num, 0, sizeof(uint32_t) * (size_t)pl->plane_stride * (size_t)pl->n_planes));
#ifdef FF_MFMA_DIAG  // (diagnostic build: per-wave clock stamps, tools/wave_stamps.py)
    if (env_int("FF_STAMPS", 0)) {
        FF_HIP(hipMalloc(&pl->d_stamps, sizeof(unsigned long long) * 4 * (size_t)U));
        FF_HIP(hipMemset(pl->d_stamps, 0, sizeof(unsigned long long) * 4 * (size_t)U));
    }
#endif
    if (pl->waves_per_wg == L_WAVES_PER_WG) {
        FF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_sad_kernel12),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes));
    } else if (pl->sparse) {
        FF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_sad_sparse_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes));
    } else {
        FF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_sad_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes));
    }
    return FF_OK;
}

int schedule_mfma(ff_plan *pl, char *err, size_t errlen)
{
    ff_plan_info &inf = pl->info;
    const int64_t N = inf.n_samples, n_slots = inf.slot_end - inf.slot_begin;
    free_and_null(pl->d_mitems);
    free_and_null(pl->d_mitem_ptr);
    free_and_null(pl->d_num);
    free_and_null(pl->d_partial);
    free_and_null(pl->d_ptiles);
    free_and_null(pl->d_ptile_ptr);
    pl->n_ptiles = 0;
    pl->m_small = false;
    pl->n_stiles = 0;
    pl->n_mitems = 0;
    const int64_t slabs = pl->m_ldb / M_KSLAB;
    const int G = inf.n_compute_units * M_WGS_PER_CU;  // one 8-wave workgroup per CU
    pl->n_mgroups = G;
    {
        // A shard with fewer 256 x 128 tiles than workgroups is all "remainder" for the persistent kernel --
        // every tile cut into branch ranges that each pay its 10 us of prologue and write-out, plus a reduce
        // launch.  Below S_MAX_WORK (32 x 32 tiles x k-steps; calibrated with tools/mfma_small_sweep.py) such a
        // shard takes pair_common_small_kernel instead: one 32 x 32 tile per workgroup over all branches, the
        // sum over the waves' ranges and the division inside the same launch.  FF_MFMA_SMALL=1 / 0 forces.
        // its tiles: row blocks of 32 in ascending order, block I with the column blocks 0 .. I (the kernel maps a
        // tile's ordinal to (I, J) by itself: small_tile_of)
        int64_t n_st = 0;
        const int64_t ib0 = inf.row_begin / S_TILE;
        for (int64_t i0 = ib0 * S_TILE; i0 < inf.row_end; i0 += S_TILE) {
            const int64_t w = std::min<int64_t>(std::min<int64_t>(i0 + S_TILE, inf.row_end) - 1, N);  // valid columns: j < w
            n_st += (w + S_TILE - 1) / S_TILE;  // (= I + 1, or I for a last block of one row)
        }
        int64_t big_tiles = 0;
        for (int64_t i0 = inf.row_begin / M_TILE_I * M_TILE_I; i0 < inf.row_end; i0 += M_TILE_I)
            big_tiles += (std::min<int64_t>(std::min<int64_t>(i0 + M_TILE_I, inf.row_end) - 1, N) + M_TILE_J - 1) / M_TILE_J;
        const int force = env_int("FF_MFMA_SMALL", -1);
        const bool fits = n_st > 0 && pl->m_digits <= S_MAX_DIGITS && n_st < ((int64_t)1 << 30) &&
                          pl->m_ldb * pl->m_digits <= S_TABLE_BYTES;  // (its digit planes live in LDS)
        const bool small = fits && (force >= 0 ? force != 0
                                               : big_tiles < G && (double)n_st * 2.0 * (double)slabs <= S_MAX_WORK);
        if (small) {
            pl->m_small = true;
            pl->n_stiles = (int)n_st;
            pl->stile_c0 = ib0 * (ib0 + 1) / 2;
            inf.kernel = FF_KERNEL_MFMA_I8_SMALL;
            inf.n_sweeps = 1;  // (every digit plane in its one pass)
            inf.planes_per_sweep = pl->m_digits;
            inf.rows_three_planes = 0;
            inf.n_tiles = inf.n_items = n_st;
            inf.n_wave_slots = n_st * S_WAVES;
            inf.elements = (double)n_st * S_TILE * S_TILE * (double)pl->m_ldb * pl->m_digits;
            pl->m_all_private = false;
            pl->m_any_atomic = false;
            pl->m_fused = env_int("FF_MFMA_FUSED_FINISH", -1) != 0;  // every slot has one writer: it can write the distance
#define FF_S_ATTR(ND)                                                                                                   \
    FF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_common_small_kernel<ND>),                            \
                               hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES));
            FF_S_ATTR(1) FF_S_ATTR(2) FF_S_ATTR(3) FF_S_ATTR(4) FF_S_ATTR(5)
#undef FF_S_ATTR
            FF_ALLOC(pl->d_num, sizeof(uint32_t) * (size_t)std::max<int64_t>(n_slots, 1), "the pair accumulators");
            return FF_OK;
        }
        inf.kernel = FF_KERNEL_MFMA_I8;
    }
    // graded rows: ONE sweep, three planes up to m_duo_from_slab and two from there on (one digit group per tile,
    // cut by cost); else base-128 digits, two planes per sweep
    const int sched_digits = pl->m_graded ? 2 : pl->m_digits;
    const int64_t duo_from_quad = pl->m_graded ? (pl->m_duo_from_slab + M_QUAD_SLABS - 1) / M_QUAD_SLABS : -1;
    inf.n_sweeps = (sched_digits + M_ND - 1) / M_ND;
    inf.planes_per_sweep = pl->m_graded ? 3 : std::min(pl->m_digits, M_ND);
    inf.rows_three_planes = pl->m_graded ? std::min<int64_t>((int64_t)pl->m_duo_from_slab * M_KSLAB, pl->m_ldb) : 0;
    std::vector<MItem> mi;
    std::vector<int32_t> mptr;
    std::vector<int32_t> ptiles, pptr;
    const bool want_partials = true;
    // Up to FF_MFMA_PRIVATE_MB of partial tiles (128 KiB each), every item gets its own: the kernel's
    // copy-out is then aligned 512-byte rows into a contiguous tile (2.8 us a tile at C3) instead of 4-byte
    // stores into rows of the triangle that start anywhere (12.8 us), and reduce_partials_kernel writes the
    // distances straight from the sums -- no num[] round trip, no finish launch.
    int64_t private_tiles = (int64_t)env_int("FF_MFMA_PRIVATE_MB", 2048) * (1 << 20) / (M_TILE_I * M_TILE_J * 4);
    int64_t n_mtiles = 0;
    for (;;) {
        n_mtiles = build_mfma_schedule(N, inf.row_begin, inf.row_end, slabs, sched_digits, G, &mi, &mptr,
                                       want_partials ? &ptiles : nullptr, want_partials ? &pptr : nullptr, private_tiles,
                                       duo_from_quad);
        if (pptr.empty()) break;
        if (hipMalloc(&pl->d_partial, sizeof(uint32_t) * (size_t)pptr.back() * M_TILE_I * M_TILE_J) == hipSuccess) break;
        (void)hipGetLastError();  // (the device is short of memory: only the remainder's ranges get private tiles)
        pl->d_partial = nullptr;
        if (private_tiles == 0)
            return ff::fail(FF_ERR_DEVICE, err, errlen, "out of device memory for %lld partial tiles of the matrix-core schedule",
                            (long long)pptr.back());
        private_tiles = 0;
    }
    if (!pptr.empty()) {
        pl->n_ptiles = (int)pptr.size() - 1;
        FF_HIP(hipMalloc(&pl->d_ptiles, sizeof(int32_t) * ptiles.size()));
        FF_HIP(hipMalloc(&pl->d_ptile_ptr, sizeof(int32_t) * pptr.size()));
        FF_HIP(hipMemcpy(pl->d_ptiles, ptiles.data(), sizeof(int32_t) * ptiles.size(), hipMemcpyHostToDevice));
        FF_HIP(hipMemcpy(pl->d_ptile_ptr, pptr.data(), sizeof(int32_t) * pptr.size(), hipMemcpyHostToDevice));
    }
    pl->n_mitems = (int)mi.size();
    inf.n_tiles = n_mtiles;
    inf.n_items = (int64_t)mi.size();
    inf.n_wave_slots = (int64_t)G * (M_THREADS / 64);
    inf.elements = (double)n_mtiles * M_TILE_I * M_TILE_J *
                   (pl->m_graded ? 3.0 * std::min<double>(pl->m_duo_from_slab * M_KSLAB, pl->m_ldb) +
                                       2.0 * std::max<double>(0.0, (double)pl->m_ldb - pl->m_duo_from_slab * M_KSLAB)
                                 : (double)pl->m_ldb * pl->m_digits);
    FF_HIP(hipMalloc(&pl->d_mitems, sizeof(MItem) * std::max<size_t>(mi.size(), 1)));
    FF_HIP(hipMalloc(&pl->d_mitem_ptr, sizeof(int32_t) * mptr.size()));
    if (!mi.empty()) FF_HIP(hipMemcpy(pl->d_mitems, mi.data(), sizeof(MItem) * mi.size(), hipMemcpyHostToDevice));
    FF_HIP(hipMemcpy(pl->d_mitem_ptr, mptr.data(), sizeof(int32_t) * mptr.size(), hipMemcpyHostToDevice));
    pl->lds_bytes = (size_t)M_LDS_BYTES;
    pl->m_all_private = !mi.empty();
    pl->m_any_atomic = false;
    for (const MItem &it : mi) {
        pl->m_all_private = pl->m_all_private && it.pad > 0;
        pl->m_any_atomic = pl->m_any_atomic || it.pad == 0;
    }
    // The matrix-core path can finish in place when every slot has exactly one writer (its tile's
    // only item, or a reduce kernel): the integer sums then never go through num[], and there is
    // neither a memset nor a finish launch.  FF_MFMA_FUSED_FINISH=1 / 0 forces either where possible;
    // by default whenever every item owns a private partial tile.  Read here, once per schedule.
    {
        const int fuse_env = env_int("FF_MFMA_FUSED_FINISH", -1);
        pl->m_fused = !pl->m_any_atomic && (fuse_env < 0 ? pl->m_all_private : fuse_env != 0);
    }
    FF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_common_mfma_kernel<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes));
    FF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_common_mfma_kernel<true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes));
    FF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_common_mfma_kernel<false, 0, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes));
    FF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_common_mfma_kernel<true, 0, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes));
    FF_ALLOC(pl->d_num, sizeof(uint32_t) * (size_t)std::max<int64_t>(n_slots, 1), "the pair accumulators");
    return FF_OK;
}

// Workgroups of refine_exact_kernel a compute unit holds at once: its grid is one round of them (a workgroup walks
// its pairs one after the other; a second round of workgroups would wait for the first to finish all of theirs).
int refine_blocks_per_cu()
{
    static const int n = [] {
        int b = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, refine_exact_kernel, REFINE_THREADS, 0) != hipSuccess || b < 1) b = 2;
        return b;
    }();
    return n;
}

// One launch of the EXACT64 pair kernel with the plan's tile height.
int launch_exact64(ff_plan *pl, hipStream_t st, double *d_out, char *err, size_t errlen)
{
    const ff_plan_info &inf = pl->info;
    if (pl->xu) {
        if (pl->n_xutiles > 0)
            pair_exact_unw_kernel<<<dim3((unsigned)pl->n_xutiles), dim3(64), 0, st>>>(pl->d_Xbits, pl->xu_ldx, pl->d_len_rows, pl->xu_slabs,
                                                                                     pl->d_xutiles, inf.row_begin, inf.row_end,
                                                                                     inf.slot_begin, d_out);
        FF_HIP(hipGetLastError());
        return FF_OK;
    }
    if (pl->n_xtiles <= 0) return FF_OK;
    const unsigned nb = (unsigned)((pl->n_xtiles + 3) / 4);
    const double *len = pl->d_len_rows ? pl->d_len_rows : pl->d_len;
#define FF_X_CASE(H)                                                                                              \
    case H:                                                                                                       \
        if (pl->weighted)                                                                                         \
            pair_exact64_kernel<true, H><<<dim3(nb), dim3(256), 0, st>>>(pl->d_DT, inf.ld, len, inf.n_rows, pl->d_xtiles, \
                                                                        pl->n_xtiles, inf.row_begin, inf.row_end,  \
                                                                        inf.slot_begin, d_out);                    \
        else                                                                                                      \
            pair_exact64_kernel<false, H><<<dim3(nb), dim3(256), 0, st>>>(pl->d_DT, inf.ld, len, inf.n_rows, pl->d_xtiles, \
                                                                         pl->n_xtiles, inf.row_begin, inf.row_end, \
                                                                         inf.slot_begin, d_out);                   \
        break;
    switch (pl->x_tile_h) {
        FF_X_CASE(4)
        FF_X_CASE(8)
        FF_X_CASE(10)
        FF_X_CASE(12)
        FF_X_CASE(14)
        FF_X_CASE(16)
    default: return ff::fail(FF_ERR_ARG, err, errlen, "EXACT64: no kernel for tile height %d", pl->x_tile_h);
    }
#undef FF_X_CASE
    FF_HIP(hipGetLastError());
    return FF_OK;
}

// Tiles of height h for the plan's shard.
int upload_exact64_tiles(ff_plan *pl, int h, char *err, size_t errlen)
{
    ff_plan_info &inf = pl->info;
    free_and_null(pl->d_xtiles);
    std::vector<Tile> tiles;
    build_tiles(inf.n_samples, inf.row_begin, inf.row_end, h, X_TILE_J, false, &tiles);
    inf.n_tiles = inf.n_items = (int64_t)tiles.size();
    inf.elements = (double)tiles.size() * h * X_TILE_J * (double)inf.n_rows;
    std::vector<XTile> xt(tiles.size());
    for (size_t k = 0; k < tiles.size(); ++k) xt[k] = {tiles[k].i0, tiles[k].j0};
    // one wave per tile, 4 per block: a launch carries fewer than 2^32 threads
    if (xt.size() >= ((size_t)1 << 26))
        return ff::fail(FF_ERR_ARG, err, errlen, "EXACT64: %zu pair tiles in one shard, at most %zu (use more shards)",
                        xt.size(), ((size_t)1 << 26) - 1);
    pl->n_xtiles = (int)xt.size();
    pl->x_tile_h = h;
    FF_HIP(hipMalloc(&pl->d_xtiles, sizeof(XTile) * std::max<size_t>(xt.size(), 1)));
    if (!xt.empty()) FF_HIP(hipMemcpy(pl->d_xtiles, xt.data(), sizeof(XTile) * xt.size(), hipMemcpyHostToDevice));
    inf.n_wave_slots = (int64_t)xt.size();
    return FF_OK;
}

// The tile height is picked once per plan.  Every height gives every pair the same operations in the
// same order; what differs is the number of waves and how their count divides into rounds of resident
// waves (C3: 33.9 ms with 16 rows, 29.8 with 12; 2,500 samples x 20,000 leaves: 34.1 with 16, 27.2 with 10).
// The default is 12 rows (the best or second best at the three shapes above); FF_X_TILE_H forces another,
// and with FF_X_CALIBRATE=1 -- for a host that runs a plan many times -- a shard big enough for it to
// matter (a quarter as many tiles of 16 rows as waves fit the device, or more) is timed with each height when it is
// scheduled, results into a scratch array, and keeps the fastest (eleven extra launches at plan time).
// The tiles of pair_exact_unw_kernel for the plan's shard.  Two column groups per tile (64 accumulator registers, six
// waves per SIMD) is what the vector ALU wants; a shard with fewer such tiles than SIMDs is bound by one wave's chain
// of steps and takes single groups (twice the waves, shorter steps).
int schedule_exact_unw(ff_plan *pl, char *err, size_t errlen)
{
    ff_plan_info &inf = pl->info;
    free_and_null(pl->d_xutiles);
    std::vector<XUTile> tiles;
    int jmax = XU_JMAX;
    build_xu_tiles(inf.n_samples, inf.row_begin, inf.row_end, jmax, &tiles);
    const int forced = env_int("FF_XU_JMAX", 0);
    if (forced == 1 || forced == 2) jmax = forced;
    else if ((int64_t)tiles.size() < (int64_t)inf.n_compute_units * 4) jmax = 1;
    if (jmax != XU_JMAX) build_xu_tiles(inf.n_samples, inf.row_begin, inf.row_end, jmax, &tiles);
    // one 64-thread workgroup per tile: a launch carries fewer than 2^31 of them
    if (tiles.size() >= ((size_t)1 << 31))
        return ff::fail(FF_ERR_ARG, err, errlen, "EXACT64: %zu pair tiles in one shard, at most %zu (use more shards)",
                        tiles.size(), ((size_t)1 << 31) - 1);
    inf.n_tiles = inf.n_items = inf.n_wave_slots = (int64_t)tiles.size();
    double cols = 0;
    for (const XUTile &t : tiles) cols += 64.0 * t.jn;
    inf.elements = cols * XU_TILE_H * (double)inf.n_rows;
    pl->n_xutiles = (int)tiles.size();
    FF_HIP(hipMalloc(&pl->d_xutiles, sizeof(XUTile) * std::max<size_t>(tiles.size(), 1)));
    if (!tiles.empty()) FF_HIP(hipMemcpy(pl->d_xutiles, tiles.data(), sizeof(XUTile) * tiles.size(), hipMemcpyHostToDevice));
    return FF_OK;
}

int schedule_exact64(ff_plan *pl, char *err, size_t errlen)
{
    ff_plan_info &inf = pl->info;
    if (pl->xu) return schedule_exact_unw(pl, err, errlen);
    if (pl->x_tile_h == 0) {
        const int forced = env_int("FF_X_TILE_H", 0);
        int h = X_TILE_H_DEFAULT;
        // A shard whose waves all fit the device at once is bound by one wave's chain of trips, not by the
        // vector ALU: the lowest tile that still keeps them all resident (C2: 1.31 ms with 16 rows, 0.84 with 8,
        // 0.56 with 4; 2 rows and 16 two-value scalar loads per trip are slower again: 0.75).
        const int64_t resident = (int64_t)inf.n_compute_units * 4 * 8;
        bool small = false;
        for (int cand : {4, 8}) {
            std::vector<Tile> count;
            build_tiles(inf.n_samples, inf.row_begin, inf.row_end, cand, X_TILE_J, false, &count);
            if ((int64_t)count.size() <= resident) {
                h = cand;
                small = true;
                break;
            }
        }
        // Larger shards are bound by the vector ALU, every SIMD working through the tiles it is dealt one after the
        // other (interleaved): a SIMD gets floor or ceil of tiles / SIMDs of them, and the kernel ends with the SIMDs
        // that got the ceiling -- so the height decides how much of the last "tile per SIMD" is idle.  With
        // avg = tiles(h) / SIMDs the efficiency is avg / ceil(avg), times what the height itself is worth (scalar
        // operands per trip, waves per SIMD; from the sweep's largest sizes).  This ranks the five heights as
        // measured at every size of tools/exact64_sweep.py (round 3; the fixed 12 rows of round 2 lost 6 % at 3,072
        // samples, 4 % at 2,048 and 3,584); 4,096 samples keep their 12 rows.
        if (!small) {
            const double simds = (double)inf.n_compute_units * 4.0;
            double best_score = 0;
            for (int cand : {8, 10, 12, 14, 16}) {
                std::vector<Tile> count;
                build_tiles(inf.n_samples, inf.row_begin, inf.row_end, cand, X_TILE_J, false, &count);
                const double avg = (double)count.size() / simds;
                const double worth = cand == 8 ? 0.95 : cand == 10 ? 0.97 : cand == 16 ? 0.985 : 1.0;
                const double score = worth * avg / std::ceil(avg);
                if (score > best_score + 1e-12) {
                    best_score = score;
                    h = cand;
                }
            }
        }
        for (int cand : X_TILE_HEIGHTS)
            if (cand == forced) h = forced;
        const int64_t n_slots = inf.slot_end - inf.slot_begin;
        std::vector<Tile> probe;
        build_tiles(inf.n_samples, inf.row_begin, inf.row_end, 16, X_TILE_J, false, &probe);
        const bool big = (int64_t)probe.size() * 4 > (int64_t)inf.n_compute_units * 4 * 6 && inf.n_rows > 0;
        if (!forced && big && env_int("FF_X_CALIBRATE", 0)) {
            // Anything that goes wrong here (no room for the scratch array, a failed launch or event) only
            // costs the calibration: the plan keeps the default height.  Scratch and events are released on
            // every path out.
            Scratch<double> scratch;
            struct Events {
                hipEvent_t e0 = nullptr, e1 = nullptr;
                ~Events()
                {
                    if (e0) (void)hipEventDestroy(e0);
                    if (e1) (void)hipEventDestroy(e1);
                }
            } ev;
            bool ok = scratch.alloc((size_t)std::max<int64_t>(n_slots, 1)) == hipSuccess &&
                      hipEventCreate(&ev.e0) == hipSuccess && hipEventCreate(&ev.e1) == hipSuccess;
            float best = 0;
            int best_h = h;
            bool warm = false;
            for (int cand : X_TILE_HEIGHTS) {
                if (!ok) break;
                ok = upload_exact64_tiles(pl, cand, err, errlen) == FF_OK;
                if (ok && !warm) ok = launch_exact64(pl, nullptr, scratch.p, err, errlen) == FF_OK;  // (clocks up)
                warm = true;
                for (int rep = 0; rep < 2 && ok; ++rep) {
                    float ms = 0;
                    ok = hipEventRecord(ev.e0, nullptr) == hipSuccess &&
                         launch_exact64(pl, nullptr, scratch.p, err, errlen) == FF_OK &&
                         hipEventRecord(ev.e1, nullptr) == hipSuccess && hipEventSynchronize(ev.e1) == hipSuccess &&
                         hipEventElapsedTime(&ms, ev.e0, ev.e1) == hipSuccess;
                    if (ok && (best == 0 || ms < best)) {
                        best = ms;
                        best_h = cand;
                    }
                }
            }
            if (ok) h = best_h;
            else (void)hipGetLastError();
        }
        pl->x_tile_h = h;
    }
    return upload_exact64_tiles(pl, pl->x_tile_h, err, errlen);
}

// The queue of pairs to recompute exactly holds up to an eighth of the shard (at least 2^20);
// next to it the run-time audit's sample of the shard and its binary64 distances.
int alloc_refine_queue(ff_plan *pl, char *err, size_t errlen)
{
    const int64_t n_slots = pl->info.slot_end - pl->info.slot_begin;
    free_and_null(pl->d_refine_list);
    free_and_null(pl->d_audit_slots);
    free_and_null(pl->d_audit_exact);
    pl->n_audit = 0;
    pl->refine_cap = (unsigned long long)std::min<int64_t>(n_slots, std::max<int64_t>(1 << 20, n_slots / 8));
    FF_HIP(hipMalloc(&pl->d_refine_list, sizeof(unsigned long long) * (size_t)std::max<unsigned long long>(pl->refine_cap, 1)));
    if (!pl->d_refine_count) FF_HIP(hipMalloc(&pl->d_refine_count, sizeof(unsigned long long) * CNT_N));
    if (!pl->d_risk_list && env_int("FF_AUDIT", 1) != 0) FF_HIP(hipMalloc(&pl->d_risk_list, sizeof(unsigned long long) * RISK_CAP));
    if (!pl->d_n_nodes) {
        const int64_t ns = pl->info.n_samples;
        FF_HIP(hipMalloc(&pl->d_n_nodes, sizeof(int32_t) * (size_t)std::max<int64_t>(ns, 1)));
        if (ns > 0) node_counts_kernel<<<dim3((unsigned)((ns + 255) / 256)), dim3(256)>>>(pl->d_indptr, ns, pl->d_n_nodes);
        FF_HIP(hipGetLastError());
        FF_HIP(hipDeviceSynchronize());  // (runs may come on any stream)
    }
    reset_counters_kernel<<<dim3(1), dim3(64)>>>(pl->d_refine_count);
    FF_HIP(hipGetLastError());
    if (n_slots > 0 && env_int("FF_AUDIT", 1) != 0) {
        // the uniform sample grows with the shard: AUDIT_PAIRS per 2^23 pairs of it (C3 as a whole: 4,096; C4: 65,536)
        const int64_t want_n = std::min<int64_t>(AUDIT_PAIRS_MAX, AUDIT_PAIRS * ((n_slots + ((int64_t)1 << 23) - 1) >> 23));
        const int n = (int)std::min<int64_t>(want_n, n_slots);
        std::vector<int64_t> slots((size_t)n);
        uint64_t x = 0x5EEDF4ACull ^ (uint64_t)pl->info.slot_begin;
        for (int q = 0; q < n; ++q) {  // splitmix64
            uint64_t z = (x += 0x9E3779B97F4A7C15ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            slots[(size_t)q] = n_slots <= n ? q : (int64_t)(z % (uint64_t)n_slots);
        }
        FF_HIP(hipMalloc(&pl->d_audit_slots, sizeof(int64_t) * (size_t)n));
        FF_HIP(hipMalloc(&pl->d_audit_exact, sizeof(double) * (size_t)n));
        FF_HIP(hipMemcpy(pl->d_audit_slots, slots.data(), sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice));
        audit_exact_kernel<<<dim3((unsigned)n), dim3(64)>>>(pl->d_audit_slots, pl->d_indptr, pl->d_ids, pl->d_abnd,
                                                             pl->d_len, pl->weighted, pl->info.slot_begin,
                                                             pl->d_audit_exact);
        FF_HIP(hipGetLastError());
        FF_HIP(hipDeviceSynchronize());  // runs may come on any stream
        pl->n_audit = n;
    }
    return FF_OK;
}

int schedule_for_shard(ff_plan *pl, char *err, size_t errlen)
{
    if (pl->walk) return FF_OK;  // (a grid-stride loop over the shard's slots: nothing to build)
    int rc = pl->mfma ? schedule_mfma(pl, err, errlen)
             : pl->info.precision == FF_PRECISION_FIXED32 ? schedule_sad(pl, err, errlen)
                                                          : schedule_exact64(pl, err, errlen);
    if (rc == FF_OK && pl->refine) rc = alloc_refine_queue(pl, err, errlen);
    return rc;
}

// What the staging steps of a plan share.
struct StageCtx {
    const ff_options *o;
    DeviceCsr *c;
    const hipDeviceProp_t *prop;
    ff_plan *pl;
    int64_t R = 0;                       // staged rows: B, or the branches in use (compaction)
    Scratch<int32_t> row_of;             // branch id -> staged row (null: identity)
    std::vector<int32_t> branch_of_row;  // staged row -> branch id (empty: identity)
    std::vector<unsigned char> branch_used;  // [B] 1: some sample has a flat node on the branch (empty: not known)
    Quant q;
};

// The names the staging code is written in.
#define FF_STAGE_NAMES                                                                      \
    const ff_options *o = x.o;                                                              \
    DeviceCsr *c = x.c;                                                                     \
    const hipDeviceProp_t &prop = *x.prop;                                                  \
    ff_plan *pl = x.pl;                                                                     \
    const int64_t N = c->N, B = c->B, nnz = c->nnz, R = x.R;                                \
    const bool weighted = pl->weighted != 0;                                                \
    ff_plan_info &inf = pl->info;                                                           \
    const int64_t n_slots = inf.slot_end - inf.slot_begin;                                  \
    int64_t *d_indptr = pl->d_indptr;                                                       \
    int32_t *d_ids = pl->d_ids;                                                             \
    double *d_abnd = pl->d_abnd, *d_len = pl->d_len;                                        \
    Scratch<int32_t> &row_of = x.row_of;                                                    \
    std::vector<int32_t> &branch_of_row = x.branch_of_row;                                  \
    Quant &q = x.q;                                                                         \
    (void)o; (void)prop; (void)N; (void)B; (void)nnz; (void)R; (void)weighted; (void)n_slots; \
    (void)d_indptr; (void)d_ids; (void)d_abnd; (void)d_len; (void)row_of; (void)branch_of_row; (void)q


int compact_branches(StageCtx &x, char *err, size_t errlen)
{
    x.R = x.c->B;
    FF_STAGE_NAMES;
    // Branch compaction.  A branch no sample has a flat node on is a zero row of the staged
    // matrix and adds |0 - 0| (or +0.0) to every pair: with a reference phylogeny much larger
    // than what the samples cover, most rows are like that.  Rows are renumbered over the
    // branches in use (ascending, so EXACT64 keeps the reference's order) when that drops
    // at least a tenth of them.  R = staged rows.
    std::vector<int32_t> h_row_of;
    if (env_int("FF_COMPACT", 1) != 0 && B > 0 && nnz > 0) {
        Scratch<unsigned char> mark;
        FF_HIP(mark.alloc((size_t)B));
        FF_HIP(hipMemset(mark.p, 0, (size_t)B));
        mark_branches_kernel<<<dim3((unsigned)std::min<int64_t>((nnz + 255) / 256, 1 << 20)), dim3(256)>>>(d_ids, nnz, mark.p);
        FF_HIP(hipGetLastError());
        std::vector<unsigned char> hm((size_t)B);
        FF_HIP(hipMemcpy(hm.data(), mark.p, (size_t)B, hipMemcpyDeviceToHost));
        int64_t used = 0;
        for (unsigned char m : hm) used += m;
        x.branch_used = hm;
        if (used * 10 <= B * 9) {
            h_row_of.assign((size_t)B, 0);
            branch_of_row.reserve((size_t)used);
            for (int64_t b = 0; b < B; ++b)
                if (hm[(size_t)b]) {
                    h_row_of[(size_t)b] = (int32_t)branch_of_row.size();
                    branch_of_row.push_back((int32_t)b);
                }
            x.R = used;
            FF_HIP(row_of.alloc((size_t)B));
            FF_HIP(hipMemcpy(row_of.p, h_row_of.data(), sizeof(int32_t) * (size_t)B, hipMemcpyHostToDevice));
        }
    }
    return FF_OK;
}

// FIXED32 unweighted on the matrix cores: presence / digit planes, sample-major, and the MFMA schedule.
int stage_for_mfma(StageCtx &x, char *err, size_t errlen)
{
    FF_STAGE_NAMES;
    // presence bits (64-bit words, pairs of slabs major) and per-row digits, zero padded to whole tiles and quads of slabs
    pl->mfma = true;
    inf.kernel = FF_KERNEL_MFMA_I8;
    inf.lengths_exact = q.lengths_exact;
    inf.scale_log2 = q.e;
    // A branch no sample has a flat node on multiplies presence bits that are all zero: its integer length is never
    // used (choose_quant only keeps it in range, up to 2^31), so it must not decide the digits of the sweep or be cut
    // into hundreds of all-zero rows when fewer than a tenth of the branches are like that and the rows stay as they are.
    if (!x.branch_used.empty())
        for (int64_t b = 0; b < B; ++b)
            if (!x.branch_used[(size_t)b]) q.klen[(size_t)b] = 0;
    uint32_t kmax = 0;
    for (uint32_t k : q.klen) kmax = std::max(kmax, k);
    auto digits_of = [](uint32_t k) {
        int d = 1;
        while (d < 5 && (k >> (7 * d)) != 0) ++d;
        return d;
    };
    int digits = digits_of(kmax);
    // The staged rows: (branch, integer length of the row).  Up to two base-128 digits -- short binary fractions,
    // C3's generator -- a row is a branch in use, in ascending order, and a sweep multiplies both digit planes.
    // Longer lengths (any real phylogeny: the integers then take the 31-bit budget of a sample's sum) are staged
    // GRADED: three signed digits d0 + 128 d1 + 32768 d2 cover a length up to TRI_KMAX in one sweep of three
    // MFMAs per block (pair_common_mfma_kernel<.., GRADED>), two of them one up to DUO_KMAX; common(i, j) is
    // linear in the lengths, so a longer branch becomes several rows with the same presence bits whose lengths
    // add up to its own, and the order of the rows is free, so they are sorted by length, longest first: the
    // sweep multiplies three planes up to the first slab without a third digit and two from there on.  With
    // lengths spread over orders of magnitude most rows are of the second kind.  FF_MFMA_GRADED=0: base-128
    // digits in branch order, two planes per sweep, as many sweeps as it takes.
    struct StagedRow {
        int32_t branch;
        uint32_t k;
    };
    std::vector<StagedRow> rows;
    bool graded = false;
    if (digits > 2 && env_int("FF_MFMA_GRADED", 1) != 0) {
        int64_t pieces = 0;
        for (int64_t r = 0; r < R; ++r) {
            const int64_t k = q.klen[(size_t)(branch_of_row.empty() ? r : branch_of_row[(size_t)r])];
            pieces += std::max<int64_t>(1, (k + TRI_KMAX - 1) / TRI_KMAX);
        }
        // (a few long branches, not a tree of them: three planes over x times the rows against two sweeps of two
        // planes, or three sweeps from five base-128 digits)
        if (pieces <= (digits < 5 ? R + R / 4 : R + R * 4 / 5) + 1024 && pieces < ((int64_t)1 << 30)) {
            graded = true;
            rows.reserve((size_t)pieces);
            for (int64_t r = 0; r < R; ++r) {
                const int32_t b = (int32_t)(branch_of_row.empty() ? r : branch_of_row[(size_t)r]);
                const int64_t k = q.klen[(size_t)b], n = std::max<int64_t>(1, (k + TRI_KMAX - 1) / TRI_KMAX);
                for (int64_t j = 0; j < n; ++j) rows.push_back({b, (uint32_t)(k / n + (j < k % n ? 1 : 0))});
            }
            std::stable_sort(rows.begin(), rows.end(), [](const StagedRow &u, const StagedRow &v) { return u.k > v.k; });
            digits = digits_of(rows.empty() ? 0u : rows[0].k);  // (of the rows: what the small-shard kernel multiplies)
        }
    }
    if (!graded) {
        rows.reserve((size_t)R);
        for (int64_t r = 0; r < R; ++r) {
            const int32_t b = (int32_t)(branch_of_row.empty() ? r : branch_of_row[(size_t)r]);
            rows.push_back({b, q.klen[(size_t)b]});
        }
    }
    const int64_t Rs = (int64_t)rows.size();  // staged rows
    pl->m_digits = digits;
    pl->m_graded = graded;
    inf.n_digits = digits;
    const int64_t n8 = round_up(N, M_TILE_I);
    const int64_t n_slabs = mfma_staged_slabs(Rs);  // whole quads of slabs
    const int64_t ldb = n_slabs * M_KSLAB;
    pl->m_ldb = ldb;
    pl->m_n8 = n8;
    inf.ld = n8;
    inf.rows_padded = ldb;
    // (+ M_PAD_SLABS slabs of zeros behind the arrays: the kernel's prefetches run past an item's end)
    const size_t bits_bytes = sizeof(unsigned long long) * (size_t)mfma_alloc_slabs(Rs) * (size_t)n8;
    const size_t plane_alloc = (size_t)(mfma_alloc_slabs(Rs) * M_KSLAB);
    const size_t digit_bytes = plane_alloc * (size_t)(digits + 1);  // (+1: a single-digit item reads its plane twice)
    inf.staged_bytes = (double)bits_bytes + (double)digit_bytes + (graded ? 3.0 * (double)plane_alloc : 0.0);
    FF_ALLOC(pl->d_Pbits, bits_bytes, "the presence bits");
    FF_HIP(hipMalloc(&pl->d_Kd, digit_bytes));
    FF_HIP(hipMemset(pl->d_Pbits, 0, bits_bytes));
    FF_HIP(hipMalloc(&pl->d_W, sizeof(unsigned long long) * (size_t)n8));
    FF_HIP(hipMemset(pl->d_W, 0, sizeof(unsigned long long) * (size_t)n8));
    {
        // digits in the kernel's order of the 64 rows of a slab: chunk C, dword kk, byte q holds
        // row 32 * (C >> 1) + 8 * q + 4 * (C & 1) + kk (ff_kernels_mfma.hpp)
        auto pos_of = [](int64_t r) {
            const int64_t slab = r / M_KSLAB, w = r % M_KSLAB;  // w = 32 * h + 8 * q + 4 * c1 + kk
            const int64_t h = w >> 5, qq = (w >> 3) & 3, c1 = (w >> 2) & 1, kk = w & 3;
            return (size_t)(slab * M_KSLAB + (2 * h + c1) * 16 + kk * 4 + qq);
        };
        std::vector<int8_t> kd(digit_bytes, 0);
        for (int64_t r = 0; r < Rs; ++r) {
            const uint32_t k = rows[(size_t)r].k;
            const size_t pos = pos_of(r);
            for (int d = 0; d < digits; ++d) kd[(size_t)d * (size_t)ldb + pos] = (int8_t)((k >> (7 * d)) & 127u);
        }
        FF_HIP(hipMemcpy(pl->d_Kd, kd.data(), digit_bytes, hipMemcpyHostToDevice));
        pl->m_duo_from_slab = 0;
        if (graded) {
            std::vector<int8_t> kt(plane_alloc * 3, 0);
            int64_t first_duo = 0;  // the first row whose length (and every later one's) needs no third digit
            for (int64_t r = 0; r < Rs; ++r) {
                int8_t d[3];
                tri_digits((int64_t)rows[(size_t)r].k, d);
                const size_t pos = pos_of(r);
                kt[pos] = d[0];
                kt[(size_t)ldb + pos] = d[1];
                kt[2 * (size_t)ldb + pos] = d[2];
                if ((int64_t)rows[(size_t)r].k > DUO_KMAX) first_duo = r + 1;
            }
            pl->m_duo_from_slab = (int)((std::min(first_duo, Rs) + M_KSLAB - 1) / M_KSLAB);
            FF_HIP(hipMalloc(&pl->d_Kt, kt.size()));
            FF_HIP(hipMemcpy(pl->d_Kt, kt.data(), kt.size(), hipMemcpyHostToDevice));
        }
    }
    Scratch<uint32_t> klen;
    Scratch<int32_t> row_ptr, row_list;  // graded: the rows of a branch (it may have several)
    FF_HIP(klen.alloc((size_t)B));
    FF_HIP(hipMemcpy(klen.p, q.klen.data(), sizeof(uint32_t) * (size_t)B, hipMemcpyHostToDevice));
    if (graded) {
        std::vector<int32_t> ptr((size_t)B + 1, 0), list((size_t)std::max<int64_t>(Rs, 1));
        for (const StagedRow &sr : rows) ++ptr[(size_t)sr.branch + 1];
        for (int64_t b = 0; b < B; ++b) ptr[(size_t)b + 1] += ptr[(size_t)b];
        std::vector<int32_t> at(ptr.begin(), ptr.end() - 1);
        for (int64_t r = 0; r < Rs; ++r) list[(size_t)at[(size_t)rows[(size_t)r].branch]++] = (int32_t)r;
        FF_HIP(row_ptr.alloc(ptr.size()));
        FF_HIP(row_list.alloc(list.size()));
        FF_HIP(hipMemcpy(row_ptr.p, ptr.data(), sizeof(int32_t) * ptr.size(), hipMemcpyHostToDevice));
        FF_HIP(hipMemcpy(row_list.p, list.data(), sizeof(int32_t) * list.size(), hipMemcpyHostToDevice));
    }
    if (nnz > 0)
        stage_mfma_bits_kernel<<<dim3((unsigned)N), dim3(256)>>>(d_indptr, d_ids, klen.p, graded ? nullptr : row_of.p,
                                                                  row_ptr.p, row_list.p, pl->d_Pbits, n8, n_slabs, pl->d_W);
    FF_HIP(hipGetLastError());
    FF_HIP(hipDeviceSynchronize());
    klen.release();
    return schedule_mfma(pl, err, errlen);
}

// FIXED32 on the vector ALU: the branch-major u32 matrix, column sums, the sparse decision, the wave schedule.
int stage_for_sad(StageCtx &x, char *err, size_t errlen)
{
    FF_STAGE_NAMES;
    const int64_t ld = round_up(std::max<int64_t>(N, 1), TILE_J);
    const int64_t rows = sad_staged_rows(R);
    inf.ld = ld;
    inf.rows_padded = rows;
    inf.lengths_exact = weighted ? 0 : q.lengths_exact;
    const size_t qt_bytes = sizeof(uint32_t) * (size_t)sad_alloc_rows(R) * (size_t)ld;
    inf.staged_bytes = (double)qt_bytes;
    FF_ALLOC(pl->d_QT, qt_bytes, "the staged branch x sample matrix");
    FF_HIP(hipMalloc(&pl->d_W, sizeof(unsigned long long) * (size_t)ld));
    Scratch<uint32_t> klen;
    if (!weighted) {
        FF_HIP(klen.alloc((size_t)B));
        if (B > 0) FF_HIP(hipMemcpy(klen.p, q.klen.data(), sizeof(uint32_t) * (size_t)B, hipMemcpyHostToDevice));
    }
    std::vector<unsigned long long> hW((size_t)ld);
    int e = q.e;
    for (int attempt = 0;; ++attempt) {
        FF_HIP(hipMemset(pl->d_QT, 0, qt_bytes));
        FF_HIP(hipMemset(pl->d_W, 0, sizeof(unsigned long long) * (size_t)ld));
        if (N > 0 && nnz > 0)
            stage_fixed32_kernel<<<dim3((unsigned)N), dim3(256)>>>(d_indptr, d_ids, d_abnd, d_len, klen.p,
                                                                    weighted ? 1 : 0, e, row_of.p, pl->d_QT, ld);
        if (rows > 0) {
            const int64_t rpb = std::max<int64_t>(64, round_up(rows, 256) / 256);
            dim3 grid((unsigned)(ld / 64), (unsigned)((rows + rpb - 1) / rpb));
            colsum_kernel<<<grid, dim3(64)>>>(pl->d_QT, ld, rows, rpb, pl->d_W);
        }
        FF_HIP(hipGetLastError());
        FF_HIP(hipMemcpy(hW.data(), pl->d_W, sizeof(unsigned long long) * (size_t)ld, hipMemcpyDeviceToHost));
        unsigned long long wmax = 0;
        for (auto w : hW) wmax = std::max(wmax, w);
        if (wmax <= 2147483647ull) break;
        if (!weighted || attempt >= 3)
            return ff::fail(FF_ERR_INTERNAL, err, errlen, "FIXED32 staging overflow (max column sum %llu)", wmax);
        --e;  // rounding pushed a column over the bound: drop one bit
    }
    klen.release();
    inf.scale_log2 = e;
    pl->n_workgroups = prop.multiProcessorCount;  // persistent: one workgroup per CU
    pl->lds_bytes = 96 * 1024;  // unused dynamic LDS sized so that exactly one workgroup fits a CU
    // activity of every (i-block, branch row): decides between the dense and the
    // sparse-aware kernel
    if (env_int("FF_SPARSE", 1) != 0 && rows > 0 && N > 0) {
        const int64_t n_iblocks = ld / TILE_I, words = (rows + SLACK_ROWS + 63) / 64;
        Scratch<unsigned long long> act64;
        FF_HIP(act64.alloc((size_t)(n_iblocks * words)));
        build_activity_kernel<<<dim3((unsigned)words, (unsigned)n_iblocks), dim3(64)>>>(pl->d_QT, ld, rows, words,
                                                                                        act64.p);
        FF_HIP(hipGetLastError());
        std::vector<unsigned long long> a64((size_t)(n_iblocks * words));
        FF_HIP(hipMemcpy(a64.data(), act64.p, sizeof(unsigned long long) * a64.size(), hipMemcpyDeviceToHost));
        act64.release();
        // only the i-blocks this shard's tiles use count for the decision
        const int64_t ib0 = inf.row_begin / TILE_I, ib1 = (inf.row_end + TILE_I - 1) / TILE_I;
        int64_t active = 0;
        for (int64_t ib = ib0; ib < ib1; ++ib)
            for (int64_t w = 0; w < words; ++w) active += __builtin_popcountll(a64[(size_t)(ib * words + w)]);
        const double total = (double)std::max<int64_t>(1, (ib1 - ib0) * rows);
        const double inactive = 1.0 - (double)active / total;
        const auto thr = ff::tuning("FF_SPARSE_MIN");
        // the list walk runs at about 0.77 of the dense loop's rate per row (shallower
        // prefetch, per-row address arithmetic), so it pays from about a quarter upwards
        if (inactive >= (thr && !thr->empty() ? atof(thr->c_str()) : 0.28)) {
            // per i-block: the list of active rows and, every 16 rows, where the list stands
            const int64_t marks = rows / (2 * KSTEP) + 1;
            pl->aptr_stride = marks;
            std::vector<uint32_t> arows, aptr((size_t)(n_iblocks * marks), 0u);
            arows.reserve((size_t)active + SPARSE_LIST_PAD);
            for (int64_t ib = 0; ib < n_iblocks; ++ib)
                for (int64_t r = 0; r <= rows; ++r) {
                    if (r % (2 * KSTEP) == 0) aptr[(size_t)(ib * marks + r / (2 * KSTEP))] = (uint32_t)arows.size();
                    if (r < rows && ((a64[(size_t)(ib * words + r / 64)] >> (r % 64)) & 1ull)) arows.push_back((uint32_t)r);
                }
            if (arows.size() >= 0xFFFFFFF0ull)
                return ff::fail(FF_ERR_INTERNAL, err, errlen, "active-row list too long");
            arows.resize(arows.size() + SPARSE_LIST_PAD, (uint32_t)rows);  // (spare entries: the batch prefetch, ff_schedule.hpp)
            pl->zero_row = (int32_t)rows;  // first slack row: zero in every column
            FF_HIP(hipMalloc(&pl->d_arows, sizeof(uint32_t) * arows.size()));
            FF_HIP(hipMemcpy(pl->d_arows, arows.data(), sizeof(uint32_t) * arows.size(), hipMemcpyHostToDevice));
            FF_HIP(hipMalloc(&pl->d_aptr16, sizeof(uint32_t) * aptr.size()));
            FF_HIP(hipMemcpy(pl->d_aptr16, aptr.data(), sizeof(uint32_t) * aptr.size(), hipMemcpyHostToDevice));
            FF_HIP(hipMalloc(&pl->d_cs16, sizeof(uint32_t) * (size_t)(marks * ld)));
            prefix16_kernel<<<dim3((unsigned)((ld + 63) / 64)), dim3(64)>>>(pl->d_QT, ld, rows, pl->d_cs16);
            FF_HIP(hipGetLastError());
            pl->sparse = true;
            inf.kernel = FF_KERNEL_SAD_U32_SPARSE;
            FF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_sad_sparse_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes));
        }
    }
    return schedule_sad(pl, err, errlen);
}

// EXACT64 unweighted: presence bits (a word per 32 staged rows and sample), the lengths by staged row, the tiles.
int stage_for_exact_unw(StageCtx &x, char *err, size_t errlen)
{
    FF_STAGE_NAMES;
    pl->xu = true;
    inf.kernel = FF_KERNEL_EXACT_F64_UNW;
    const int64_t ldx = xu_ld(N);
    pl->xu_ldx = ldx;
    pl->xu_slabs = (int)xu_slabs(R);
    inf.ld = ldx;
    inf.rows_padded = xu_slabs(R) * XU_SLAB;
    const size_t bits_bytes = sizeof(uint32_t) * (size_t)xu_alloc_slabs(R) * (size_t)ldx;
    const size_t len_count = (size_t)xu_alloc_lengths(R);
    inf.staged_bytes = (double)bits_bytes + 8.0 * (double)len_count;
    FF_ALLOC(pl->d_Xbits, bits_bytes, "the presence bits");
    FF_HIP(hipMemset(pl->d_Xbits, 0, bits_bytes));
    if (N > 0 && nnz > 0)
        stage_xbits_kernel<<<dim3((unsigned)N), dim3(256)>>>(d_indptr, d_ids, row_of.p, pl->d_Xbits, ldx);
    FF_HIP(hipGetLastError());
    std::vector<double> lr(len_count, 0.0);  // treeDists by staged row, zeros behind
    for (int64_t r = 0; r < R; ++r) lr[(size_t)r] = c->h_len[(size_t)(branch_of_row.empty() ? r : branch_of_row[(size_t)r])];
    FF_HIP(hipMalloc(&pl->d_len_rows, sizeof(double) * len_count));
    FF_HIP(hipMemcpy(pl->d_len_rows, lr.data(), sizeof(double) * len_count, hipMemcpyHostToDevice));
    return schedule_exact_unw(pl, err, errlen);
}

// EXACT64: the branch-major binary64 matrix and its tiles.
int stage_for_exact64(StageCtx &x, char *err, size_t errlen)
{
    FF_STAGE_NAMES;
    if (!weighted && N > 0 && B > 0 && env_int("FF_EXACT_UNW", 1) != 0) return stage_for_exact_unw(x, err, errlen);
    const int64_t ld = round_up(std::max<int64_t>(N, 1), X_TILE_J);
    inf.ld = ld;
    inf.rows_padded = R;
    // (+ X_VALUES_PAD values: a tile whose height does not divide 64 reads up to H - 1 operands past the last row's end)
    const size_t dt_bytes = sizeof(double) * ((size_t)std::max<int64_t>(R, 1) * (size_t)ld + X_VALUES_PAD);
    inf.staged_bytes = (double)dt_bytes;
    FF_ALLOC(pl->d_DT, dt_bytes, "the staged binary64 matrix");
    FF_HIP(hipMemset(pl->d_DT, 0, dt_bytes));
    if (N > 0 && nnz > 0)
        stage_exact64_kernel<<<dim3((unsigned)N), dim3(256)>>>(d_indptr, d_ids, d_abnd, weighted ? 1 : 0,
                                                                row_of.p, pl->d_DT, ld);
    FF_HIP(hipGetLastError());
    if (row_of.p) {  // the walk reads treeDists by staged row
        std::vector<double> lr((size_t)R);
        for (int64_t r = 0; r < R; ++r) lr[(size_t)r] = c->h_len[(size_t)branch_of_row[(size_t)r]];
        FF_HIP(hipMalloc(&pl->d_len_rows, sizeof(double) * (size_t)R));
        FF_HIP(hipMemcpy(pl->d_len_rows, lr.data(), sizeof(double) * (size_t)R, hipMemcpyHostToDevice));
    }
    return schedule_exact64(pl, err, errlen);
}

// Stages the device-resident flat nodes and builds the schedule.  Takes ownership of *c.
int plan_build(const ff_options *o, DeviceCsr *c, const hipDeviceProp_t &prop, ff_plan *pl, char *err,
               size_t errlen)
{
    const int64_t N = c->N, B = c->B;
    const bool weighted = pl->weighted != 0;
    ff_plan_info &inf = pl->info;
    pl->d_len = c->d_len;  // the plan owns the device arrays from here on
    pl->d_indptr = c->d_indptr;
    pl->d_ids = c->d_ids;
    pl->d_abnd = c->d_abnd;
    c->d_len = nullptr;
    c->d_indptr = nullptr;
    c->d_ids = nullptr;
    c->d_abnd = nullptr;

    if (o->flags & FF_FLAG_UNSORTED_WALK) {
        // nothing to stage: the walk reads the flat nodes as they stand, and every reformulation above (dense rows,
        // integer sums, presence bits) assumes lists a merge pairs up correctly
        pl->walk = true;
        inf.n_rows = B;
        inf.rows_padded = B;
        inf.precision = FF_PRECISION_EXACT64;
        inf.kernel = FF_KERNEL_WALK_F64;
        inf.staged_bytes = 12.0 * (double)c->nnz;
        inf.n_tiles = inf.n_items = 0;
        inf.n_wave_slots = (int64_t)inf.n_compute_units * 8 * 4;
        inf.elements = 0;
        FF_HIP(hipDeviceSynchronize());
        return FF_OK;
    }
    StageCtx x;
    x.o = o;
    x.c = c;
    x.prop = &prop;
    x.pl = pl;
    int rc = compact_branches(x, err, errlen);
    if (rc) return rc;
    const int64_t R = x.R;
    Quant &q = x.q;
    inf.n_rows = R;

    int prec = o->precision;
    const bool is_auto = prec == FF_PRECISION_AUTO;
    // AUTO: problems small enough that the binary64 walk costs about a millisecond
    // get the reference's exact roundings (this covers all of the reference's own
    // test data); everything larger takes the fixed-point path -- except UNWEIGHTED with
    // a branch length off the binary grid (below).
    if (is_auto && (double)ff_num_pairs(N) * (double)R <= 4294967296.0)
        prec = FF_PRECISION_EXACT64;
    if (prec != FF_PRECISION_EXACT64) {
        q = choose_quant(*c, weighted, pl->d_indptr, pl->d_ids, pl->d_abnd, pl->d_len);
        if (!q.fixed_ok) {
            if (prec == FF_PRECISION_FIXED32)
                return ff::fail(FF_ERR_ARG, err, errlen, "FIXED32 not applicable: %s", q.why_not.c_str());
            prec = FF_PRECISION_EXACT64;
        } else if (is_auto && !weighted && !q.lengths_exact) {
            // The reference's unweighted value is what its two chains of additions round to (unifrac.go:144-171), and
            // the bar for unweighted is its bits, not a tolerance: integer lengths that carry a rounding (any real
            // phylogeny) cannot give them, pair_exact_unw_kernel does (C3's shape: 10 ms a pass against 0.3 on the
            // matrix cores -- a thirtieth of what the command spends reading the table and writing the distances).
            // FIXED32 on such lengths stays available on request: within 1e-6, with refinement and audit.
            prec = FF_PRECISION_EXACT64;
        } else {
            prec = FF_PRECISION_FIXED32;
        }
    }
    inf.precision = prec;
    inf.kernel = prec == FF_PRECISION_EXACT64 ? FF_KERNEL_EXACT_F64 : FF_KERNEL_SAD_U32;

    const bool use_mfma = prec == FF_PRECISION_FIXED32 && !weighted && env_int("FF_UNWEIGHTED_MFMA", 1) != 0 && N > 0 && B > 0;
    if (use_mfma) rc = stage_for_mfma(x, err, errlen);
    else if (prec == FF_PRECISION_FIXED32) rc = stage_for_sad(x, err, errlen);
    else rc = stage_for_exact64(x, err, errlen);
    if (rc) return rc;
    FF_HIP(hipDeviceSynchronize());
    // FIXED32 whose integers carry a rounding (weighted; unweighted with lengths off the binary
    // grid) divides by binary64 weights, so that only the numerator's rounding reaches a distance
    if (prec == FF_PRECISION_FIXED32 && (weighted || !inf.lengths_exact) && N > 0) {
        FF_HIP(hipMalloc(&pl->d_wex, sizeof(double) * (size_t)N));
        exact_weight_kernel<<<dim3((unsigned)N), dim3(256)>>>(pl->d_indptr, pl->d_ids, pl->d_abnd, pl->d_len,
                                                               weighted ? 1 : 0, pl->d_wex);
        FF_HIP(hipGetLastError());
        FF_HIP(hipDeviceSynchronize());
    }
    // FIXED32 keeps the flat nodes resident for refine_exact_kernel unless the integer
    // sums are exact already (unweighted with lengths on the binary grid)
    if (prec == FF_PRECISION_FIXED32 && (weighted || !inf.lengths_exact) && env_int("FF_REFINE", 1)) {
        pl->refine = true;
        rc = alloc_refine_queue(pl, err, errlen);
        if (rc) return rc;
    } else {
        (void)hipFree(pl->d_indptr);
        (void)hipFree(pl->d_ids);
        (void)hipFree(pl->d_abnd);
        pl->d_indptr = nullptr;
        pl->d_ids = nullptr;
        pl->d_abnd = nullptr;
    }
    return FF_OK;
}

int plan_create_impl(const ff_problem *p, const ff_options *o, ff_plan *pl, char *err, size_t errlen)
{
    hipDeviceProp_t prop;
    int rc = plan_begin(o, p->n_samples, p->n_branches, pl, &prop, err, errlen);
    if (rc) return rc;
    DeviceCsr c;
    rc = csr_from_host(p, &c, err, errlen);
    if (rc == FF_OK) rc = plan_build(o, &c, prop, pl, err, errlen);
    c.release();
    return rc;
}

int plan_run_impl(ff_plan *pl, hipStream_t st, double *d_out, bool timed, char *err, size_t errlen)
{
    const ff_plan_info &inf = pl->info;
    const int64_t n_slots = inf.slot_end - inf.slot_begin;
    if (n_slots <= 0) return FF_OK;
    if (!d_out) return ff::fail(FF_ERR_ARG, err, errlen, "null output pointer");
    DeviceScope scope;  // (the caller's current device is its own again when this returns: a host that drives several
    FF_HIP(scope.enter(pl->device));  // plans on several devices from one thread does not find it changed under it)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (timed) {
        if (pl->events_used == pl->events.size()) {
            hipEvent_t a, b;
            FF_HIP(hipEventCreate(&a));
            FF_HIP(hipEventCreate(&b));
            pl->events.push_back({a, b});
        }
        ev0 = pl->events[pl->events_used].first;
        ev1 = pl->events[pl->events_used].second;
        ++pl->events_used;
    }
    if (pl->walk) {
        if (timed) FF_HIP(hipEventRecord(ev0, st));
        pair_walk_kernel<<<dim3((unsigned)std::min<int64_t>((n_slots + 255) / 256, (int64_t)inf.n_compute_units * 8)), dim3(256), 0, st>>>(
            pl->d_indptr, pl->d_ids, pl->d_abnd, pl->d_len, pl->weighted, inf.slot_begin, n_slots, d_out);
        if (timed) FF_HIP(hipEventRecord(ev1, st));
        FF_HIP(hipGetLastError());
        return FF_OK;
    }
    if (inf.precision == FF_PRECISION_FIXED32) {
        FinishArgs fin;
        fin.W = pl->d_W;
        fin.wex = pl->d_wex;
        fin.out = d_out;
        fin.n_nodes = pl->refine ? pl->d_n_nodes : nullptr;
        fin.refine_list = pl->d_refine_list;
        fin.refine_count = pl->d_refine_count;
        fin.refine_cap = pl->refine_cap;
        fin.risk_list = pl->refine ? pl->d_risk_list : nullptr;
        fin.scale_log2 = inf.scale_log2;
        fin.weighted = pl->weighted;
        const bool fused = pl->mfma && pl->m_fused;  // (decided when the shard was scheduled: schedule_mfma)
        if (pl->refine) reset_counters_kernel<<<dim3(1), dim3(64), 0, st>>>(pl->d_refine_count);
        if (!fused && (!pl->mfma || pl->m_any_atomic))
            FF_HIP(hipMemsetAsync(pl->d_num, 0, sizeof(uint32_t) * (size_t)n_slots, st));
        if (timed) FF_HIP(hipEventRecord(ev0, st));
        if (pl->mfma && pl->m_small) {
            FinishArgs none = fin;
            none.out = nullptr;  // null: integer sums into num[]
            const dim3 grid((unsigned)pl->n_stiles), block(S_THREADS);
            const uint4 *bits = reinterpret_cast<const uint4 *>(pl->d_Pbits);
            const int n_slab_pairs = (int)(pl->m_ldb / (2 * M_KSLAB));
#define FF_S_CASE(ND)                                                                                                  \
    case ND:                                                                                                           \
        pair_common_small_kernel<ND><<<grid, block, (size_t)(pl->m_ldb * ND + S_RED_BYTES), st>>>(bits, pl->m_n8, pl->d_Kd, pl->m_ldb, n_slab_pairs,        \
                                                             pl->stile_c0, pl->d_W, pl->d_num, inf.row_begin,          \
                                                             inf.row_end, inf.slot_begin, fused ? fin : none);         \
        break;
            switch (pl->m_digits) {
                FF_S_CASE(1)
                FF_S_CASE(2)
                FF_S_CASE(3)
                FF_S_CASE(4)
                FF_S_CASE(5)
            default: return ff::fail(FF_ERR_INTERNAL, err, errlen, "no small-shard kernel for %d digits", pl->m_digits);
            }
#undef FF_S_CASE
            if (timed) FF_HIP(hipEventRecord(ev1, st));
        } else if (pl->mfma) {
            auto kern = pl->m_graded ? (pl->m_all_private ? pair_common_mfma_kernel<true, 0, true> : pair_common_mfma_kernel<false, 0, true>)
                                     : (pl->m_all_private ? pair_common_mfma_kernel<true> : pair_common_mfma_kernel<false>);
#ifdef FF_MFMA_DIAG  // ablations for timing only (wrong results): see the kernel's DIAG parameter
            switch (env_int("FF_MFMA_DIAG", 0)) {
            case 2: kern = pair_common_mfma_kernel<false, 2>; break;
            case 4: kern = pair_common_mfma_kernel<false, 4>; break;
            case 8: kern = pair_common_mfma_kernel<false, 8>; break;
            case 6: kern = pair_common_mfma_kernel<false, 6>; break;
            case 14: kern = pair_common_mfma_kernel<false, 14>; break;
            default: break;
            }
            if (env_int("FF_MFMA_DIAG", 0))
                FF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes));
#endif
            FinishArgs none = fin;
            none.out = nullptr;  // null: the kernels leave integer sums in num[]
            if (pl->n_mitems > 0)
                kern<<<dim3((unsigned)pl->n_mgroups), dim3(M_THREADS), pl->lds_bytes, st>>>(
                    reinterpret_cast<const uint4 *>(pl->d_Pbits), pl->m_n8, pl->m_graded ? pl->d_Kt : pl->d_Kd, pl->m_ldb, pl->d_mitems, pl->d_mitem_ptr, pl->d_W, pl->d_num,
                    pl->d_partial, inf.row_begin, inf.row_end, inf.slot_begin, pl->m_duo_from_slab, fused ? fin : none);
            if (pl->n_ptiles > 0 && pl->m_all_private)
                reduce_private_kernel<<<dim3(M_REDUCE_BLOCKS, (unsigned)pl->n_ptiles), dim3(M_REDUCE_THREADS), 0, st>>>(
                    pl->d_partial, pl->d_ptiles, pl->d_ptile_ptr, pl->d_W, pl->d_num, inf.row_begin, inf.row_end,
                    inf.slot_begin, fused ? fin : none);
            else if (pl->n_ptiles > 0)
                reduce_partials_kernel<<<dim3(M_REDUCE_BLOCKS, (unsigned)pl->n_ptiles), dim3(M_REDUCE_THREADS), 0, st>>>(
                    pl->d_partial, pl->d_ptiles, pl->d_ptile_ptr, pl->d_num, inf.row_begin, inf.row_end, inf.slot_begin,
                    fused ? fin : none);
            // the timed region is the pair kernel AND the reduction of its partial tiles (sums, W_i + W_j,
            // divisions: work that round 1's pair kernel did itself)
            if (timed) FF_HIP(hipEventRecord(ev1, st));
        } else if (inf.n_items > 0 && pl->sparse)
            pair_sad_sparse_kernel<<<dim3((unsigned)pl->n_workgroups), dim3(WAVES_PER_WG * 64), pl->lds_bytes, st>>>(
                pl->d_QT, inf.ld, pl->d_items, pl->d_item_ptr, pl->d_arows, pl->d_aptr16, pl->aptr_stride, pl->d_cs16,
                pl->zero_row, pl->d_num, pl->plane_stride, inf.row_begin, inf.row_end, inf.slot_begin);
        else if (inf.n_items > 0)
            (pl->waves_per_wg == L_WAVES_PER_WG ? pair_sad_kernel12 : pair_sad_kernel)
                <<<dim3((unsigned)pl->n_workgroups), dim3((unsigned)pl->waves_per_wg * 64), pl->lds_bytes, st>>>(
                pl->d_QT, inf.ld, pl->d_items, pl->d_item_ptr, pl->d_num, pl->plane_stride, inf.row_begin, inf.row_end,
                inf.slot_begin, pl->d_stamps, SYNC_TRIPS);
        if (timed && !pl->mfma) FF_HIP(hipEventRecord(ev1, st));
        if (!fused) {
            const unsigned nb = (unsigned)std::min<int64_t>((n_slots + 256 * FINISH_RUN - 1) / (256 * FINISH_RUN), 1 << 22);
            finish_fixed32_kernel<<<dim3(nb), dim3(256), 0, st>>>(pl->d_num, pl->n_planes, pl->plane_stride, fin, inf.slot_begin, n_slots);
        }
        if (pl->refine)
            refine_exact_kernel<<<dim3((unsigned)(inf.n_compute_units * refine_blocks_per_cu())), dim3(REFINE_THREADS), 0, st>>>(
                pl->d_refine_list, pl->d_refine_count, pl->refine_cap, pl->d_indptr, pl->d_ids, pl->d_abnd,
                pl->d_len, pl->weighted, inf.slot_begin, d_out);
        if (pl->refine && pl->n_audit > 0)
            audit_compare_kernel<<<dim3((unsigned)((pl->n_audit + 255) / 256)), dim3(256), 0, st>>>(
                pl->d_audit_slots, pl->d_audit_exact, pl->n_audit, d_out, pl->d_refine_count);
        if (pl->refine && pl->d_risk_list)
            audit_risk_kernel<<<dim3((unsigned)RISK_CAP), dim3(64), 0, st>>>(pl->d_risk_list, pl->d_refine_count, pl->d_indptr, pl->d_ids,
                                                                             pl->d_abnd, pl->d_len, pl->weighted, inf.slot_begin, d_out);
    } else {
        if (timed) FF_HIP(hipEventRecord(ev0, st));
        {
            const int rc = launch_exact64(pl, st, d_out, err, errlen);
            if (rc != FF_OK) return rc;
        }
        if (timed) FF_HIP(hipEventRecord(ev1, st));
    }
    FF_HIP(hipGetLastError());
    return FF_OK;
}

// After a completed FIXED32 run: did it deliver what the tolerance promises?  Not when more pairs
// were queued for the binary64 walk than the queue holds, or when a pair of the audit sample is
// further than AUDIT_REL from its binary64 value.  `why` gets the sentence for the caller.
int plan_fixed32_verdict(ff_plan *pl, bool *ok, std::string *why)
{
    *ok = true;
    if (!pl->refine) return FF_OK;
    unsigned long long c[CNT_N] = {};
    if (hipMemcpy(c, pl->d_refine_count, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) return FF_ERR_DEVICE;
    char buf[256];
    if (c[0] > pl->refine_cap) {
        snprintf(buf, sizeof buf, "%llu nearly identical pairs, %llu can be re-computed exactly", c[0], pl->refine_cap);
        *ok = false;
    } else if (c[1] > 0) {
        double worst;
        memcpy(&worst, &c[2], sizeof worst);
        snprintf(buf, sizeof buf, "%llu of %llu audited pairs are further than %.1e from their binary64 value (worst %.2e)",
                 c[1], (unsigned long long)pl->n_audit + c[CNT_RISK_CHECKED], AUDIT_REL, worst);
        *ok = false;
    }
    if (!*ok && why) *why = buf;
    return FF_OK;
}

// The audit's verdict on the run that has just completed, into an ff_plan_info that is handed back to a caller.
void fill_audit_info(ff_plan *pl, ff_plan_info *info)
{
    info->audit_checked = info->audit_failed = 0;
    info->audit_worst_rel_err = 0.0;
    info->audit_min_headroom = INFINITY;
    int64_t uniform = 0, found = 0, chk = 0;
    (void)ff_plan_audit(pl, &info->audit_checked, &info->audit_failed, &info->audit_worst_rel_err);
    (void)ff_plan_audit_detail(pl, &uniform, &found, &chk, &info->audit_min_headroom);
}

}  // namespace

extern "C" {

int ff_plan_create(const ff_problem *p, const ff_options *o, ff_plan **plan, char *err, size_t errlen)
{
    if (!plan) return ff::fail(FF_ERR_ARG, err, errlen, "null plan pointer");
    *plan = nullptr;
    ff_options dflt;
    ff_options_default(&dflt);
    if (!o) o = &dflt;
    if (o->world < 1 || o->rank < 0 || o->rank >= o->world)
        return ff::fail(FF_ERR_ARG, err, errlen, "bad shard %d of %d", o->rank, o->world);
    if (o->precision < FF_PRECISION_AUTO || o->precision > FF_PRECISION_EXACT64)
        return ff::fail(FF_ERR_ARG, err, errlen, "bad precision %d", o->precision);
    if ((o->flags & ~FF_FLAG_UNSORTED_WALK) != 0) return ff::fail(FF_ERR_ARG, err, errlen, "unknown flags %d", o->flags);
    int rc = validate_problem(p, err, errlen, (o->flags & FF_FLAG_UNSORTED_WALK) != 0);
    if (rc) return rc;
    auto *pl = new ff_plan();
    rc = plan_create_impl(p, o, pl, err, errlen);
    if (rc) {
        plan_free_device(pl);
        delete pl;
        return rc;
    }
    *plan = pl;
    return FF_OK;
}

void ff_plan_destroy(ff_plan *pl)
{
    if (!pl) return;
    plan_free_device(pl);
    delete pl;
}

int ff_plan_info_get(const ff_plan *pl, ff_plan_info *info)
{
    if (!pl || !info) return FF_ERR_ARG;
    *info = pl->info;
    return FF_OK;
}

int ff_plan_set_shard(ff_plan *pl, int32_t rank, int32_t world, char *err, size_t errlen)
{
    if (!pl) return ff::fail(FF_ERR_ARG, err, errlen, "null plan");
    if (world < 1 || rank < 0 || rank >= world) return ff::fail(FF_ERR_ARG, err, errlen, "bad shard %d of %d", rank, world);
    int cur = -1;
    FF_HIP(hipGetDevice(&cur));
    if (cur != pl->device) FF_HIP(hipSetDevice(pl->device));
    FF_HIP(hipDeviceSynchronize());  // runs of the old shard may still read the schedule
    int rc = set_shard_geometry(pl, rank, world, err, errlen);
    if (rc == FF_OK) {
        pl->shard_rank = rank;
        pl->shard_world = world;
        rc = schedule_for_shard(pl, err, errlen);
    }
    if (rc == FF_OK) FF_HIP(hipDeviceSynchronize());
    if (cur != pl->device) (void)hipSetDevice(cur);
    return rc;
}

int ff_plan_run(ff_plan *pl, void *stream, double *d_out, char *err, size_t errlen)
{
    if (!pl) return ff::fail(FF_ERR_ARG, err, errlen, "null plan");
    return plan_run_impl(pl, (hipStream_t)stream, d_out, false, err, errlen);
}

int ff_plan_run_host(ff_plan *pl, double *out, char *err, size_t errlen)
{
    if (!pl) return ff::fail(FF_ERR_ARG, err, errlen, "null plan");
    const int64_t n_slots = pl->info.slot_end - pl->info.slot_begin;
    if (n_slots <= 0) return FF_OK;
    if (!out) return ff::fail(FF_ERR_ARG, err, errlen, "null output pointer");
    int cur = -1;
    FF_HIP(hipGetDevice(&cur));
    if (cur != pl->device) FF_HIP(hipSetDevice(pl->device));
    if (n_slots > pl->host_out_cap) {
        (void)hipFree(pl->d_host_out);
        pl->d_host_out = nullptr;
        pl->host_out_cap = 0;
        FF_ALLOC(pl->d_host_out, sizeof(double) * (size_t)n_slots, "the results");
        pl->host_out_cap = n_slots;
    }
    int rc = plan_run_impl(pl, nullptr, pl->d_host_out, false, err, errlen);
    if (rc == FF_OK) {
        FF_HIP(hipMemcpy(out, pl->d_host_out, sizeof(double) * (size_t)n_slots, hipMemcpyDeviceToHost));
        bool ok = true;
        std::string why;
        if (plan_fixed32_verdict(pl, &ok, &why) == FF_OK && !ok)
            rc = ff::fail(FF_ERR_PRECISION, err, errlen, "%s: stage this problem with FF_PRECISION_EXACT64", why.c_str());
    }
    if (cur != pl->device) (void)hipSetDevice(cur);
    return rc;
}

int ff_plan_run_timed(ff_plan *pl, void *stream, double *d_out, char *err, size_t errlen)
{
    if (!pl) return ff::fail(FF_ERR_ARG, err, errlen, "null plan");
    return plan_run_impl(pl, (hipStream_t)stream, d_out, true, err, errlen);
}

int ff_plan_refined_pairs(ff_plan *pl, int64_t *queued, int64_t *capacity)
{
    if (!pl || !queued || !capacity) return FF_ERR_ARG;
    *queued = 0;
    *capacity = (int64_t)pl->refine_cap;
    if (!pl->refine) return FF_OK;
    unsigned long long n = 0;  // (CNT_QUEUED is the first counter)
    if (hipMemcpy(&n, pl->d_refine_count, sizeof n, hipMemcpyDeviceToHost) != hipSuccess) return FF_ERR_DEVICE;
    *queued = (int64_t)n;
    return FF_OK;
}

#ifdef FF_MFMA_DIAG
// Diagnostic build only.  First call (host_out == null): allocates the stamp array for n_workgroups
// and arms the kernel.  Later calls copy the stamps out ([workgroup][4 items][8] 100 MHz ticks).
int ff_debug_mfma_stamps(unsigned long long *host_out, int64_t n_workgroups)
{
    static unsigned long long *d = nullptr;
    const size_t bytes = (size_t)n_workgroups * 4 * 8 * sizeof(unsigned long long);
    if (!host_out) {
        if (hipMalloc(&d, bytes) != hipSuccess || hipMemset(d, 0, bytes) != hipSuccess) return FF_ERR_DEVICE;
        return hipMemcpyToSymbol(HIP_SYMBOL(g_mfma_stamps), &d, sizeof(d)) == hipSuccess ? FF_OK : FF_ERR_DEVICE;
    }
    if (!d) return FF_ERR_ARG;
    return hipMemcpy(host_out, d, bytes, hipMemcpyDeviceToHost) == hipSuccess ? FF_OK : FF_ERR_DEVICE;
}
#endif

#ifdef FF_MFMA_DIAG
// Diagnostic build only: as ff_debug_mfma_stamps, for pair_common_small_kernel ([workgroup][8] 100 MHz ticks).
int ff_debug_small_stamps(unsigned long long *host_out, int64_t n_workgroups)
{
    static unsigned long long *d = nullptr;
    const size_t bytes = (size_t)n_workgroups * 8 * sizeof(unsigned long long);
    if (!host_out) {
        if (hipMalloc(&d, bytes) != hipSuccess || hipMemset(d, 0, bytes) != hipSuccess) return FF_ERR_DEVICE;
        return hipMemcpyToSymbol(HIP_SYMBOL(g_small_stamps), &d, sizeof(d)) == hipSuccess ? FF_OK : FF_ERR_DEVICE;
    }
    if (!d) return FF_ERR_ARG;
    return hipMemcpy(host_out, d, bytes, hipMemcpyDeviceToHost) == hipSuccess ? FF_OK : FF_ERR_DEVICE;
}
#endif

int ff_plan_audit(ff_plan *pl, int64_t *checked, int64_t *failed, double *max_rel_err)
{
    if (!pl || !checked || !failed || !max_rel_err) return FF_ERR_ARG;
    *checked = 0;
    *failed = 0;
    *max_rel_err = 0.0;
    if (!pl->refine || (pl->n_audit <= 0 && !pl->d_risk_list)) return FF_OK;
    unsigned long long c[CNT_N] = {};
    if (hipMemcpy(c, pl->d_refine_count, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) return FF_ERR_DEVICE;
    *checked = pl->n_audit + (int64_t)c[CNT_RISK_CHECKED];
    *failed = (int64_t)c[CNT_AUDIT_FAILED];
    memcpy(max_rel_err, &c[CNT_AUDIT_WORST], sizeof(double));
    return FF_OK;
}

int ff_plan_audit_detail(ff_plan *pl, int64_t *uniform_checked, int64_t *risk_found, int64_t *risk_checked, double *min_headroom)
{
    if (!pl || !uniform_checked || !risk_found || !risk_checked || !min_headroom) return FF_ERR_ARG;
    *uniform_checked = *risk_found = *risk_checked = 0;
    *min_headroom = INFINITY;
    if (!pl->refine) return FF_OK;
    unsigned long long c[CNT_N] = {};
    if (hipMemcpy(c, pl->d_refine_count, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) return FF_ERR_DEVICE;
    *uniform_checked = pl->n_audit;
    *risk_found = (int64_t)c[CNT_RISK_FOUND];
    *risk_checked = (int64_t)c[CNT_RISK_CHECKED];
    const uint32_t bits = (uint32_t)c[CNT_MIN_HEADROOM2];
    float h2;
    memcpy(&h2, &bits, sizeof h2);
    *min_headroom = pl->d_risk_list ? std::sqrt((double)h2) : INFINITY;
    return FF_OK;
}

// ---- device buffers shared between processes (include/frackyfrac_amd.h, "Device buffers ...") ----

static_assert(sizeof(hipIpcMemHandle_t) == sizeof(ff_ipc_handle), "ff_ipc_handle must hold a hipIpcMemHandle_t");

int ff_device_alloc(int32_t device, size_t bytes, void **dptr, char *err, size_t errlen)
{
    if (!dptr) return ff::fail(FF_ERR_ARG, err, errlen, "null pointer argument");
    *dptr = nullptr;
    int cur = -1;
    FF_HIP(hipGetDevice(&cur));
    if (device >= 0 && device != cur) FF_HIP(hipSetDevice(device));
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, std::max<size_t>(bytes, 1));
    if (e == hipSuccess) e = hipMemset(p, 0, std::max<size_t>(bytes, 1));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (device >= 0 && device != cur) (void)hipSetDevice(cur);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(p);
        return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: cannot allocate %.2f GB of device memory: %s", (double)bytes / 1e9,
                        hipGetErrorString(e));
    }
    *dptr = p;
    return FF_OK;
}

int ff_device_free(void *dptr, char *err, size_t errlen)
{
    if (dptr) FF_HIP(hipFree(dptr));
    return FF_OK;
}

int ff_ipc_export(void *dptr, ff_ipc_handle *handle, char *err, size_t errlen)
{
    if (!dptr || !handle) return ff::fail(FF_ERR_ARG, err, errlen, "null pointer argument");
    hipIpcMemHandle_t h;
    FF_HIP(hipIpcGetMemHandle(&h, dptr));
    memcpy(handle->bytes, &h, sizeof h);
    return FF_OK;
}

int ff_ipc_open(const ff_ipc_handle *handle, int32_t device, void **dptr, char *err, size_t errlen)
{
    if (!dptr || !handle) return ff::fail(FF_ERR_ARG, err, errlen, "null pointer argument");
    *dptr = nullptr;
    int cur = -1;
    FF_HIP(hipGetDevice(&cur));
    if (device >= 0 && device != cur) FF_HIP(hipSetDevice(device));
    hipIpcMemHandle_t h;
    memcpy(&h, handle->bytes, sizeof h);
    void *p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (device >= 0 && device != cur) (void)hipSetDevice(cur);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: hipIpcOpenMemHandle failed: %s", hipGetErrorString(e));
    }
    *dptr = p;
    return FF_OK;
}

int ff_ipc_close(void *dptr, char *err, size_t errlen)
{
    if (dptr) FF_HIP(hipIpcCloseMemHandle(dptr));
    return FF_OK;
}

int ff_device_copy_async(void *dst, const void *src, size_t bytes, void *stream, char *err, size_t errlen)
{
    if (bytes == 0) return FF_OK;
    if (!dst || !src) return ff::fail(FF_ERR_ARG, err, errlen, "null pointer argument");
    FF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return FF_OK;
}

// Diagnostics, not part of the public header: copies the per-wave start/end stamps
// of the last pair_sad_kernel launch (FF_STAMPS=1) into out[4 * n_wave_slots]: 2 per wave on the 100 MHz
// wall clock, then 2 per wave on the shader clock.
int ff_debug_read_stamps(ff_plan *pl, unsigned long long *out)
{
    if (!pl || !pl->d_stamps || !out) return FF_ERR_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return FF_ERR_DEVICE;
    if (hipMemcpy(out, pl->d_stamps, sizeof(unsigned long long) * 4 * (size_t)pl->info.n_wave_slots,
                  hipMemcpyDeviceToHost) != hipSuccess)
        return FF_ERR_DEVICE;
    return FF_OK;
}

int ff_plan_timing_collect(ff_plan *pl, double *total_ms, int32_t *launches)
{
    if (!pl || !total_ms || !launches) return FF_ERR_ARG;
    double sum = 0;
    int32_t n = 0;
    for (size_t k = 0; k < pl->events_used; ++k) {
        float ms = 0.f;
        if (hipEventSynchronize(pl->events[k].second) != hipSuccess ||
            hipEventElapsedTime(&ms, pl->events[k].first, pl->events[k].second) != hipSuccess)
            return FF_ERR_DEVICE;
        sum += (double)ms;
        ++n;
    }
    pl->events_used = 0;
    *total_ms = sum;
    *launches = n;
    return FF_OK;
}

int ff_unifrac_dists(const ff_problem *p, const ff_options *o, double *out, char *err, size_t errlen)
{
    return ff::unifrac_dists_info(p, o, out, nullptr, err, errlen);
}

// ---- flat-argument forms (hosts whose FFI may not pass a struct of managed pointers: cgo) ----

static ff_problem problem_of(int64_t n_samples, int64_t n_branches, const double *branch_len, const int64_t *indptr,
                             const int32_t *branch_id, const double *abnd)
{
    ff_problem p;
    p.n_samples = n_samples;
    p.n_branches = n_branches;
    p.branch_len = branch_len;
    p.indptr = indptr;
    p.branch_id = branch_id;
    p.abnd = abnd;
    return p;
}

int ff_plan_create_csr(int64_t n_samples, int64_t n_branches, const double *branch_len, const int64_t *indptr,
                       const int32_t *branch_id, const double *abnd, const ff_options *o, ff_plan **plan, char *err,
                       size_t errlen)
{
    const ff_problem p = problem_of(n_samples, n_branches, branch_len, indptr, branch_id, abnd);
    return ff_plan_create(&p, o, plan, err, errlen);
}

int ff_unifrac_dists_csr(int64_t n_samples, int64_t n_branches, const double *branch_len, const int64_t *indptr,
                         const int32_t *branch_id, const double *abnd, const ff_options *o, double *out, char *err,
                         size_t errlen)
{
    const ff_problem p = problem_of(n_samples, n_branches, branch_len, indptr, branch_id, abnd);
    return ff_unifrac_dists(&p, o, out, err, errlen);
}

int ff_unifrac_dists_stream_csr(int64_t n_samples, int64_t n_branches, const double *branch_len, const int64_t *indptr,
                                const int32_t *branch_id, const double *abnd, const ff_options *o,
                                int64_t max_pairs_per_chunk, ff_dists_fn fn, void *user, char *err, size_t errlen)
{
    const ff_problem p = problem_of(n_samples, n_branches, branch_len, indptr, branch_id, abnd);
    return ff_unifrac_dists_stream(&p, o, max_pairs_per_chunk, fn, user, err, errlen);
}

// ---- unifracDists as a lazy ordered sequence (frcfrc/unifrac.go:209-228) ----

namespace {

// Two device result buffers and their host twins (pinned when the host lets us), one stream: sub-shard k + 1
// is reduced and copied out while the consumer is handed sub-shard k.
struct StreamBuffers {
    double *d[2] = {nullptr, nullptr}, *h[2] = {nullptr, nullptr};
    bool pinned[2] = {false, false};
    int64_t cap[2] = {0, 0};
    hipStream_t st = nullptr;
    ~StreamBuffers()
    {
        for (int b = 0; b < 2; ++b) {
            (void)hipFree(d[b]);
            if (pinned[b]) (void)hipHostFree(h[b]);
            else free(h[b]);
        }
        if (st) (void)hipStreamDestroy(st);
    }
    int reserve(int b, int64_t n, char *err, size_t errlen)
    {
        if (n <= cap[b]) return FF_OK;
        (void)hipFree(d[b]);
        d[b] = nullptr;
        if (pinned[b]) (void)hipHostFree(h[b]);
        else free(h[b]);
        h[b] = nullptr;
        cap[b] = 0;
        const size_t bytes = sizeof(double) * (size_t)n;
        if (hipMalloc(&d[b], bytes) != hipSuccess) {
            (void)hipGetLastError();
            return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: cannot allocate %.2f GB for a sub-shard's results (a smaller "
                            "max_pairs_per_chunk makes it smaller)", (double)bytes / 1e9);
        }
        pinned[b] = hipHostMalloc(reinterpret_cast<void **>(&h[b]), bytes, hipHostMallocDefault) == hipSuccess;
        if (!pinned[b]) {
            (void)hipGetLastError();
            h[b] = static_cast<double *>(malloc(bytes));
            if (!h[b]) return ff::fail(FF_ERR_INTERNAL, err, errlen, "out of host memory for %.2f GB of results", (double)bytes / 1e9);
        }
        cap[b] = n;
        return FF_OK;
    }
};

}  // namespace

int ff_unifrac_dists_stream(const ff_problem *p, const ff_options *o, int64_t max_pairs, ff_dists_fn fn, void *user,
                            char *err, size_t errlen)
{
    if (!fn) return ff::fail(FF_ERR_ARG, err, errlen, "null callback");
    ff_options base;
    ff_options_default(&base);
    if (o) base = *o;
    if (base.world < 1 || base.rank < 0 || base.rank >= base.world)
        return ff::fail(FF_ERR_ARG, err, errlen, "bad shard %d of %d", base.rank, base.world);
    if (max_pairs <= 0) max_pairs = (int64_t)1 << 25;
    int rc = validate_problem(p, err, errlen, (base.flags & FF_FLAG_UNSORTED_WALK) != 0);
    if (rc) return rc;
    int64_t rb = 0, re = 0;
    if (ff_shard_rows(p->n_samples, base.rank, base.world, &rb, &re) != FF_OK)
        return ff::fail(FF_ERR_ARG, err, errlen, "bad shard %d of %d", base.rank, base.world);
    const int64_t shard_pairs = (re > 0 ? re * (re - 1) / 2 : 0) - (rb > 0 ? rb * (rb - 1) / 2 : 0);
    if (shard_pairs <= 0) return FF_OK;  // lazily: nothing to deliver, nothing staged
    // Sub-shard k of c of shard (rank, world) is shard rank * c + k of world * c: ff_shard_rows' boundaries
    // N sqrt(k / world) nest, so the sub-shards tile this shard's contiguous slot range exactly.  Shards are
    // whole 32-row blocks; a sub-shard that comes out above max_pairs is delivered in pieces.
    int64_t c = (shard_pairs + max_pairs - 1) / max_pairs;
    c = std::min<int64_t>(c, std::max<int64_t>(1, (re - rb + 31) / 32));
    c = std::min<int64_t>(c, (int64_t)INT32_MAX / base.world);
    ff_options o2 = base;
    o2.rank = (int32_t)(base.rank * c);
    o2.world = (int32_t)(base.world * c);
    ff_plan *pl = nullptr;
    rc = ff_plan_create(p, &o2, &pl, err, errlen);
    if (rc) return rc;
    struct PlanGuard {
        ff_plan *&pl;
        ~PlanGuard() { ff_plan_destroy(pl); }
    } guard{pl};
    StreamBuffers buf;
    if (hipStreamCreateWithFlags(&buf.st, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: cannot create a stream");
    }
    auto deliver = [&](const double *h, int64_t slot0, int64_t n) -> bool {
        for (int64_t off = 0; off < n; off += max_pairs)
            if (!fn(user, slot0 + off, h + off, std::min(max_pairs, n - off))) return false;
        return true;
    };
    int prev = -1;  // buffer holding a finished sub-shard that has not been delivered yet
    int64_t prev_slot0 = 0, prev_n = 0;
    int cur = 0;
    for (int64_t k = 0; k < c; ++k) {
        if (k > 0) {
            o2.rank = (int32_t)(base.rank * c + k);
            rc = ff_plan_set_shard(pl, o2.rank, o2.world, err, errlen);
            if (rc) return rc;
        }
        const int64_t slot0 = pl->info.slot_begin, n = pl->info.slot_end - pl->info.slot_begin;
        if (n <= 0) continue;
        rc = buf.reserve(cur, n, err, errlen);
        if (rc) return rc;
        rc = plan_run_impl(pl, buf.st, buf.d[cur], false, err, errlen);
        if (rc) return rc;
        FF_HIP(hipMemcpyAsync(buf.h[cur], buf.d[cur], sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, buf.st));
        bool go_on = true;
        if (prev >= 0) go_on = deliver(buf.h[prev], prev_slot0, prev_n);  // (the device works on sub-shard k meanwhile)
        prev = -1;
        FF_HIP(hipStreamSynchronize(buf.st));
        if (!go_on) return FF_OK;  // the consumer stopped: sub-shard k is dropped, the rest never computed
        bool ok = true;
        if (plan_fixed32_verdict(pl, &ok, nullptr) == FF_OK && !ok) {
            // replicates, or a failed audit (ff_plan_audit): this sub-shard again, and all later ones, in binary64
            ff_plan_destroy(pl);
            pl = nullptr;
            o2.precision = FF_PRECISION_EXACT64;
            rc = ff_plan_create(p, &o2, &pl, err, errlen);
            if (rc == FF_OK) rc = plan_run_impl(pl, buf.st, buf.d[cur], false, err, errlen);
            if (rc) return rc;
            FF_HIP(hipMemcpyAsync(buf.h[cur], buf.d[cur], sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, buf.st));
            FF_HIP(hipStreamSynchronize(buf.st));
        }
        prev = cur;
        prev_slot0 = slot0;
        prev_n = n;
        cur ^= 1;
    }
    if (prev >= 0) (void)deliver(buf.h[prev], prev_slot0, prev_n);
    return FF_OK;
}

int ff_flatten_device(const ff_tree *tree, int64_t n_samples, const int64_t *leaf_ptr, const int64_t *leaf_idx,
                      const double *leaf_val, int leave_unnormalized, ff_flat **flat, char *err, size_t errlen)
{
    if (!tree || !leaf_ptr || !flat || n_samples < 0 || (leaf_ptr[n_samples] > 0 && (!leaf_idx || !leaf_val)))
        return ff::fail(FF_ERR_ARG, err, errlen, "ff_flatten_device: bad argument");
    if (leave_unnormalized == FF_L_REFERENCE)  // (the recursion's order is made on the host)
        return ff_flatten_leaf_csr(tree, n_samples, leaf_ptr, leaf_idx, leaf_val, leave_unnormalized, flat, err, errlen);
    ff_options o;
    ff_options_default(&o);
    ff_plan tmp;  // only to run the device checks of plan_begin
    hipDeviceProp_t prop;
    int rc = plan_begin(&o, n_samples, (int64_t)tree->size.size(), &tmp, &prop, err, errlen);
    if (rc) return rc;
    DeviceCsr c;
    bool too_deep = false;
    rc = csr_from_leaves(tree, n_samples, leaf_ptr, leaf_idx, leaf_val, !leave_unnormalized, &c, &too_deep, err,
                         errlen);
    if (too_deep) {
        c.release();
        return ff_flatten_leaf_csr(tree, n_samples, leaf_ptr, leaf_idx, leaf_val, leave_unnormalized, flat, err,
                                   errlen);
    }
    if (rc) {
        c.release();
        return rc;
    }
    auto *f = new ff_flat();
    f->n_samples = c.N;
    f->n_branches = c.B;
    f->branch_len = c.h_len;
    f->indptr = c.h_indptr;
    f->branch_id.resize((size_t)c.nnz);
    f->abnd.resize((size_t)c.nnz);
    hipError_t he = hipSuccess;
    if (c.nnz > 0) {
        he = hipMemcpy(f->branch_id.data(), c.d_ids, sizeof(int32_t) * (size_t)c.nnz, hipMemcpyDeviceToHost);
        if (he == hipSuccess)
            he = hipMemcpy(f->abnd.data(), c.d_abnd, sizeof(double) * (size_t)c.nnz, hipMemcpyDeviceToHost);
    }
    c.release();
    if (he != hipSuccess) {
        delete f;
        return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: download of flat nodes failed: %s", hipGetErrorString(he));
    }
    *flat = f;
    return FF_OK;
}

int ff_plan_create_from_leaves(const ff_tree *tree, int64_t n_samples, const int64_t *leaf_ptr,
                               const int64_t *leaf_idx, const double *leaf_val, int leave_unnormalized,
                               const ff_options *o, ff_plan **plan, char *err, size_t errlen)
{
    if (!plan) return ff::fail(FF_ERR_ARG, err, errlen, "null plan pointer");
    *plan = nullptr;
    if (!tree || !leaf_ptr || n_samples < 0 || (leaf_ptr[n_samples] > 0 && (!leaf_idx || !leaf_val)))
        return ff::fail(FF_ERR_ARG, err, errlen, "ff_plan_create_from_leaves: bad argument");
    ff_options dflt;
    ff_options_default(&dflt);
    if (!o) o = &dflt;
    if (o->world < 1 || o->rank < 0 || o->rank >= o->world)
        return ff::fail(FF_ERR_ARG, err, errlen, "bad shard %d of %d", o->rank, o->world);
    if (o->precision < FF_PRECISION_AUTO || o->precision > FF_PRECISION_EXACT64)
        return ff::fail(FF_ERR_ARG, err, errlen, "bad precision %d", o->precision);
    const int64_t B = (int64_t)tree->size.size();
    if (B > (int64_t)INT32_MAX - 64 || n_samples > (int64_t)1 << 22)
        return ff::fail(FF_ERR_ARG, err, errlen, "bad problem size N=%lld B=%lld", (long long)n_samples, (long long)B);
    if (leave_unnormalized < 0 || leave_unnormalized > FF_L_REFERENCE)
        return ff::fail(FF_ERR_ARG, err, errlen, "leave_unnormalized must be 0, 1 or FF_L_REFERENCE");
    // FF_L_REFERENCE: the lists as the reference's -l leaves them (unsorted) and the literal walk over them
    ff_options o_walk = *o;
    const bool reference_l = leave_unnormalized == FF_L_REFERENCE;
    if (reference_l) {
        o_walk.flags |= FF_FLAG_UNSORTED_WALK;
        o = &o_walk;
    }
    auto *pl = new ff_plan();
    hipDeviceProp_t prop;
    int rc = plan_begin(o, n_samples, B, pl, &prop, err, errlen);
    DeviceCsr c;
    if (rc == FF_OK) {
        bool too_deep = reference_l;  // (the recursion's order is made on the host)
        if (!reference_l)
            rc = csr_from_leaves(tree, n_samples, leaf_ptr, leaf_idx, leaf_val, !leave_unnormalized, &c, &too_deep, err,
                                 errlen);
        if (too_deep) {  // a caterpillar: flatten on the host instead
            c.release();
            ff_flat *flat = nullptr;
            rc = ff_flatten_leaf_csr(tree, n_samples, leaf_ptr, leaf_idx, leaf_val, leave_unnormalized, &flat, err,
                                     errlen);
            if (rc == FF_OK) {
                ff_problem p;
                ff_flat_problem(flat, &p);
                rc = csr_from_host(&p, &c, err, errlen);
                ff_flat_free(flat);
            }
        }
    }
    if (rc == FF_OK) rc = plan_build(o, &c, prop, pl, err, errlen);
    c.release();
    if (rc) {
        plan_free_device(pl);
        delete pl;
        return rc;
    }
    *plan = pl;
    return FF_OK;
}

}  // extern "C"

// Runs a freshly created plan, copies the shard's slots into out (host) and destroys the
// plan.  If FIXED32's refinement queue overflowed, `recreate` builds an EXACT64 plan and
// the shard is repeated with it.
int ff::run_plan_to_host(ff_plan *pl, const std::function<int(ff_plan **)> &recreate_exact64, double *out,
                         ff_plan_info *info_out, char *err, size_t errlen, bool shard_local)
{
    int rc = FF_OK;
    const int64_t n_slots = pl->info.slot_end - pl->info.slot_begin;
    if (n_slots > 0) {
        double *d_out = nullptr;
        hipError_t he = hipMalloc(&d_out, sizeof(double) * (size_t)n_slots);
        if (he != hipSuccess) {
            ff_plan_destroy(pl);
            return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: hipMalloc(out) failed: %s", hipGetErrorString(he));
        }
        rc = ff_plan_run(pl, nullptr, d_out, err, errlen);
        bool ok = true;
        if (rc == FF_OK && plan_fixed32_verdict(pl, &ok, nullptr) == FF_OK && !ok) {
            // more nearly-equal pairs than the refinement queue holds (the data set is mostly
            // replicates), or the audit sample missed its bar: run the whole shard in binary64
            ff_plan_destroy(pl);
            pl = nullptr;
            rc = recreate_exact64(&pl);
            if (rc == FF_OK) rc = ff_plan_run(pl, nullptr, d_out, err, errlen);
        }
        if (rc == FF_OK) {
            he = hipMemcpy(out + (shard_local ? 0 : pl->info.slot_begin), d_out, sizeof(double) * (size_t)n_slots,
                           hipMemcpyDeviceToHost);
            if (he != hipSuccess)
                rc = ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: copy of results failed: %s", hipGetErrorString(he));
        }
        (void)hipFree(d_out);
    }
    if (info_out && pl) {
        *info_out = pl->info;
        if (rc == FF_OK) fill_audit_info(pl, info_out);
    }
    ff_plan_destroy(pl);
    return rc;
}

ff::ShardRunner::ShardRunner(const ff_tree *tree, int64_t n_samples, const int64_t *leaf_ptr, const int64_t *leaf_idx,
                             const double *leaf_val, int leave_unnormalized, const ff_options &opt)
    : tree_(tree), n_(n_samples), lp_(leaf_ptr), li_(leaf_idx), lv_(leaf_val), unnorm_(leave_unnormalized), opt_(opt)
{
}

ff::ShardRunner::~ShardRunner()
{
    if (pl_) {
        int cur = -1;
        const bool sw = hipGetDevice(&cur) == hipSuccess && cur != pl_->device && hipSetDevice(pl_->device) == hipSuccess;
        (void)hipFree(d_out_);
        ff_plan_destroy(pl_);
        if (sw) (void)hipSetDevice(cur);
    }
}

int ff::ShardRunner::create(int32_t rank, int32_t world, int precision, char *err, size_t errlen)
{
    if (pl_) ff_plan_destroy(pl_);
    pl_ = nullptr;
    ff_options o = opt_;
    o.rank = rank;
    o.world = world;
    o.precision = precision;
    return ff_plan_create_from_leaves(tree_, n_, lp_, li_, lv_, unnorm_, &o, &pl_, err, errlen);
}

int ff::ShardRunner::run(int32_t rank, int32_t world, double *out, ff_plan_info *info, char *err, size_t errlen)
{
    int rc = pl_ ? ff_plan_set_shard(pl_, rank, world, err, errlen) : create(rank, world, opt_.precision, err, errlen);
    if (rc) return rc;
    // (a pass runs on a thread of its own: the buffers below belong on the plan's device)
    if (hipSetDevice(pl_->device) != hipSuccess) return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: cannot select device %d", pl_->device);
    const int64_t n_slots = pl_->info.slot_end - pl_->info.slot_begin;
    if (n_slots > 0) {
        if (!out) return ff::fail(FF_ERR_ARG, err, errlen, "null output pointer");
        if (n_slots > d_out_cap_) {
            (void)hipFree(d_out_);
            d_out_ = nullptr;
            d_out_cap_ = 0;
            hipError_t he = hipMalloc(&d_out_, sizeof(double) * (size_t)n_slots);
            if (he != hipSuccess) {
                (void)hipGetLastError();
                return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: %s: cannot allocate %.2f GB for the results of shard %d of %d",
                                hipGetErrorString(he), 8e-9 * (double)n_slots, (int)rank, (int)world);
            }
            d_out_cap_ = n_slots;
        }
        rc = ff_plan_run(pl_, nullptr, d_out_, err, errlen);
        bool ok = true;
        if (rc == FF_OK && plan_fixed32_verdict(pl_, &ok, nullptr) == FF_OK && !ok) {
            // mostly replicates, or a failed audit: binary64 from here on (run_plan_to_host does the same for one shard)
            opt_.precision = FF_PRECISION_EXACT64;
            rc = create(rank, world, FF_PRECISION_EXACT64, err, errlen);
            if (rc == FF_OK) rc = ff_plan_run(pl_, nullptr, d_out_, err, errlen);
        }
        if (rc) return rc;
        hipError_t he = hipMemcpy(out, d_out_, sizeof(double) * (size_t)n_slots, hipMemcpyDeviceToHost);
        if (he != hipSuccess) {
            (void)hipGetLastError();
            return ff::fail(FF_ERR_DEVICE, err, errlen, "HIP: copy of results failed: %s", hipGetErrorString(he));
        }
    }
    if (info) {
        *info = pl_->info;
        fill_audit_info(pl_, info);
    }
    return FF_OK;
}

int ff::device_count()
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

size_t ff::device_free_bytes(int device)
{
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return 0;
    size_t free_b = 0, total_b = 0;
    if (device >= 0 && device != cur && hipSetDevice(device) != hipSuccess) return 0;
    const bool ok = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
    if (device >= 0 && device != cur) (void)hipSetDevice(cur);
    return ok ? free_b : 0;
}

void ff::device_warmup(int want)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return;
    n = std::min(n, std::max(want, 1));
    for (int d = 0; d < n; ++d)
        if (hipSetDevice(d) == hipSuccess) (void)hipFree(nullptr);
    if (n > 0) (void)hipSetDevice(0);
}

// ff_unifrac_dists that also reports what the staging decided (used by the CLI's -stats).
int ff::unifrac_dists_info(const ff_problem *p, const ff_options *o, double *out, ff_plan_info *info_out,
                           char *err, size_t errlen)
{
    if (!out && p && p->n_samples > 1) return ff::fail(FF_ERR_ARG, err, errlen, "null output pointer");
    ff_plan *pl = nullptr;
    int rc = ff_plan_create(p, o, &pl, err, errlen);
    if (rc) return rc;
    auto again = [&](ff_plan **np) {
        ff_options o2;
        if (o) o2 = *o; else ff_options_default(&o2);
        o2.precision = FF_PRECISION_EXACT64;
        return ff_plan_create(p, &o2, np, err, errlen);
    };
    return ff::run_plan_to_host(pl, again, out, info_out, err, errlen);
}

// unifrac() with stage A on the device (used by ff_unifrac and the CLI).
int ff::unifrac_leaves_info(const ff_tree *tree, int64_t n_samples, const int64_t *leaf_ptr,
                            const int64_t *leaf_idx, const double *leaf_val, int leave_unnormalized,
                            const ff_options *o, double *out, ff_plan_info *info_out, char *err, size_t errlen,
                            bool shard_local)
{
    if (!out && n_samples > 1 && !shard_local) return ff::fail(FF_ERR_ARG, err, errlen, "null output pointer");
    ff_plan *pl = nullptr;
    int rc = ff_plan_create_from_leaves(tree, n_samples, leaf_ptr, leaf_idx, leaf_val, leave_unnormalized, o, &pl,
                                        err, errlen);
    if (rc) return rc;
    auto again = [&](ff_plan **np) {
        ff_options o2;
        if (o) o2 = *o; else ff_options_default(&o2);
        o2.precision = FF_PRECISION_EXACT64;
        return ff_plan_create_from_leaves(tree, n_samples, leaf_ptr, leaf_idx, leaf_val, leave_unnormalized, &o2, np,
                                          err, errlen);
    };
    return ff::run_plan_to_host(pl, again, out, info_out, err, errlen, shard_local);
}

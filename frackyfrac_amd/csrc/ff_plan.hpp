// ff_plan.hpp -- the plan of the device path and what its three translation units share.
//
//   ff_device.hip      the C ABI around a plan: create / destroy / run, the lazy ordered sequence, device buffers
//                      shared between processes, and the C++ helpers of the frcfrc command.  No kernels.
//   ff_dev_stage.hip   inputs -> HBM: stage A on the device, the flat nodes from the host, the choice of arithmetic,
//                      branch compaction, the staged matrices / presence bits of every kernel
//                      (ff_kernels_stage.hpp, ff_kernels_stage_a.hpp).
//   ff_dev_run.hip     the pair kernels (ff_kernels_pair_sad / mfma / mfma_small / finish / exact_unw.hpp), their work
//                      schedules per shard, the refinement queue and the audit, one pass of a plan.
//
// Every kernel lives in exactly one of the two files that launch kernels, inside its anonymous namespace (the
// kernels stay internal and need no relocatable device code); the host functions declared below cross them.
#pragma once

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
    ff::sched::Item *d_items = nullptr;
    int32_t *d_item_ptr = nullptr;
    int32_t shard_rank = 0, shard_world = 1;
    double *d_host_out = nullptr;  // ff_plan_run_host's device buffer
    int64_t host_out_cap = 0;
    int n_workgroups = 0;
    int waves_per_wg = ff::sched::WAVES_PER_WG;
    size_t lds_bytes = 0;
    unsigned long long *d_stamps = nullptr;  // FF_STAMPS=1 diagnostics
    // sparse-aware variant: activity bits per (i-block, 16-row trip)
    bool sparse = false;
    uint32_t *d_arows = nullptr, *d_aptr16 = nullptr, *d_cs16 = nullptr;
    int64_t aptr_stride = 0;
    int32_t zero_row = 0;
    // sparse tables: the rare rows live outside the staged matrix, as lists grouped by blocks of low_tile samples
    bool split = false;
    int low_tile = 0;                     // one of LOW_TILES
    int64_t low_rows = 0;                 // rare rows
    int low_blocks = 0;                   // sample blocks
    int64_t low_words = 0;                // 64-bit words of a block's row bitmap
    uint32_t *d_low_ptr = nullptr;        // [low_blocks][low_rows + 1]: where row r's entries of block k begin
    uint2 *d_low_ent = nullptr;           // entries: (sample's index within its block, staged value)
    unsigned long long *d_low_bits = nullptr;  // [low_blocks][low_words]: rows with an entry in the block
    uint32_t *d_Wl = nullptr;             // [ld] column sums over the rare rows
    uint32_t *d_mlow = nullptr;           // [slots of the shard] sum over the rare rows of min(q_i, q_j)
    ff::sched::LowTile *d_low_tiles = nullptr;
    int n_low_tiles = 0;
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
    ff::sched::MItem *d_mitems = nullptr;
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
    ff::sched::XTile *d_xtiles = nullptr;
    int n_xtiles = 0;
    int x_tile_h = 0;  // EXACT64 tile height in use (0: not chosen yet)
    bool x_skip = false;  // weighted EXACT64 on pair_exact64_skip_kernel (heights 8, 12, 16; read once per schedule)
    bool walk = false;  // FF_FLAG_UNSORTED_WALK: no staging at all, pair_walk_kernel over the flat nodes as they stand
    // EXACT64 unweighted (pair_exact_unw_kernel): presence bits, lengths by staged row, tiles
    bool xu = false;
    uint32_t *d_Xbits = nullptr;
    int64_t xu_ldx = 0;
    int xu_slabs = 0;
    ff::sched::XUTile *d_xutiles = nullptr;
    int n_xutiles = 0;
    // timing: the events of every timed run since the last collect -- around the pair reduction, and between the matrix
    // rows' kernel and the rare rows' (recorded only where the plan is split: `split` says which)
    struct TimedRun { hipEvent_t first, second, mid; };
    std::vector<TimedRun> events;
    size_t events_used = 0;
};

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

namespace ff {
namespace dev {

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

inline int env_int(const char *name, int dflt)
{
    const auto v = ff::tuning(name);
    if (!v || v->empty()) return dflt;
    return atoi(v->c_str());
}

// Host threads for a pass over `work` flat nodes: one per 2 M, at most 8 (and never more than the machine has).
inline unsigned host_threads(int64_t work)
{
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, hw), work / 2000000));
}

template <typename T> void free_and_null(T *&p)
{
    (void)hipFree(p);
    p = nullptr;
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

// ---- ff_device.hip
// The caller's arrays are what an ff_problem promises (sizes, ascending ids unless unsorted_ok, positive abundances).
int validate_problem(const ff_problem *p, char *err, size_t errlen, bool unsorted_ok = false);
// Picks the device (it must be a gfx950) and fills the shard geometry.
int set_shard_geometry(ff_plan *pl, int32_t rank, int32_t world, char *err, size_t errlen);

// ---- ff_dev_stage.hip
int csr_from_host(const ff_problem *p, DeviceCsr *c, char *err, size_t errlen);
int csr_from_leaves(const ff_tree *t, int64_t N, const int64_t *leaf_ptr, const int64_t *leaf_idx, const double *leaf_val,
                    bool normalize, DeviceCsr *c, bool *too_deep, char *err, size_t errlen);
int plan_build(const ff_options *o, DeviceCsr *c, const hipDeviceProp_t &prop, ff_plan *pl, char *err, size_t errlen);

// ---- ff_dev_run.hip
int schedule_sad(ff_plan *pl, char *err, size_t errlen);
int schedule_mfma(ff_plan *pl, char *err, size_t errlen);
int schedule_exact64(ff_plan *pl, char *err, size_t errlen);
int schedule_exact_unw(ff_plan *pl, char *err, size_t errlen);
int alloc_refine_queue(ff_plan *pl, char *err, size_t errlen);
int schedule_for_shard(ff_plan *pl, char *err, size_t errlen);
int plan_run_impl(ff_plan *pl, hipStream_t st, double *d_out, bool timed, char *err, size_t errlen);
int plan_fixed32_verdict(ff_plan *pl, bool *ok, std::string *why);

}  // namespace dev
}  // namespace ff

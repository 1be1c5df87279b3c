// ff_kernels_finish.hpp -- integer sums -> distances, exact refinement of nearly equal pairs, the EXACT64 pair kernel.
// A fragment of ff_dev_run.hip: included there, once, inside its anonymous namespace
// (every kernel lives in exactly one translation unit, so the kernels stay internal and need no relocatable device code).

// Integer sums -> distances: finish_pair (ff_kernels_finish_pair.hpp) for every slot of the shard.  A workgroup
// takes FINISH_RUN x 256 consecutive slots, thread t of it slots t, t + 256, ...: every access is a wave's
// contiguous 256 / 512 bytes, and the slot -> (i, j) conversion (a square root) is done once per thread, the
// later ones by stepping j and wrapping into the next row.
constexpr int FINISH_RUN = 4;
__global__ __launch_bounds__(256)
void finish_fixed32_kernel(const uint32_t *__restrict__ num,
                           int n_planes, int64_t plane_stride,  // the ranges of a split tile own a plane each
                           const FinishArgs f, int64_t slot_begin, int64_t n_slots)
{
    float h2min = INFINITY;  // smallest squared headroom over this thread's pairs (finish_note_headroom)
    // grid-stride: a launch carries at most 2^32 - 1 threads, a shard can have more slots
    for (int64_t t = (int64_t)blockIdx.x * (256 * FINISH_RUN) + threadIdx.x; t < n_slots;
         t += ((int64_t)gridDim.x - 1) * (256 * FINISH_RUN)) {
        int64_t i, j;
        slot_to_pair(slot_begin + t, &i, &j);
        // the sums of the thread's FINISH_RUN slots first, all their loads in flight together (a slot past the
        // shard's end reads the last one's: never used)
        uint32_t u32[FINISH_RUN];
#pragma unroll
        for (int e = 0; e < FINISH_RUN; ++e) {
            const int64_t te = t + 256 * e < n_slots ? t + 256 * e : n_slots - 1;
            u32[e] = num[te];
            if (f.mlow) u32[e] -= 2u * f.mlow[te];  // (modulo 2^32: the sum below is the pair's U < 2^32)
        }
        for (int q = 1; q < n_planes; ++q) {
#pragma unroll
            for (int e = 0; e < FINISH_RUN; ++e) {
                const int64_t te = t + 256 * e < n_slots ? t + 256 * e : n_slots - 1;
                u32[e] += num[(int64_t)q * plane_stride + te];
            }
        }
#pragma unroll
        for (int e = 0; e < FINISH_RUN; ++e, t += 256) {
            if (t < n_slots) finish_pair(f, t, i, j, f.mlow ? u32[e] + f.wl[i] + f.wl[j] : u32[e], h2min);
            j += 256;
            while (j >= i) {  // (rows are shorter than 256 only at the top of the triangle)
                j -= i;
                ++i;
            }
        }
    }
    finish_note_headroom(f, h2min);
}

// Flat nodes per sample (what the refinement rule of finish_pair_w wants of a pair's two samples).
__global__ void node_counts_kernel(const int64_t *__restrict__ indptr, int64_t n_samples, int32_t *__restrict__ n_nodes)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_samples) n_nodes[s] = (int32_t)(indptr[s + 1] - indptr[s]);
}

// ---- Run-time audit of FIXED32 ------------------------------------------------------------
// A fixed pseudo-random sample of the shard's pairs is computed in binary64 when the shard is
// scheduled (audit_exact_kernel: the inputs of a plan do not change between runs) and compared
// with what every run delivers (audit_compare_kernel, a few microseconds).  A sampled pair
// further than AUDIT_REL from its binary64 value fails the run: the blocking entry points then
// repeat the shard in EXACT64 or return FF_ERR_PRECISION (ff_plan_audit for asynchronous
// callers).  The staging's error model makes that a < 1e-9 event per sampled pair; the audit is
// there for what a model cannot promise.
constexpr double AUDIT_REL = 0.5e-6;
constexpr int AUDIT_PAIRS = 4096;        // per 2^23 pairs of the shard
constexpr int AUDIT_PAIRS_MAX = 65536;   // and at most this many

// Position of id b in ids[lo, hi) (ascending), or -1.
__device__ __forceinline__ int64_t find_branch(const int32_t *__restrict__ ids, int64_t lo, int64_t hi, int32_t b)
{
    const int64_t end = hi;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (ids[mid] < b) lo = mid + 1;
        else hi = mid;
    }
    return (lo < end && ids[lo] == b) ? lo : -1;
}

// A wave computes one pair (samples si > sj) in binary64: the sums of unifrac.go:144-205, lanes striding over
// the flat nodes of either sample and looking the branch up in the other.  Not the reference's
// ORDER of additions (the differences are ~1e-15 relative, nine orders below what is checked).  Every lane returns
// the distance.
__device__ __forceinline__ double exact_pair_wave(int64_t si, int64_t sj, const int64_t *__restrict__ indptr,
                                                  const int32_t *__restrict__ branch_id, const double *__restrict__ abnd,
                                                  const double *__restrict__ tree_dists, int weighted)
{
    const int lane = threadIdx.x & 63;
    const int64_t i0 = indptr[si], i1 = indptr[si + 1], j0 = indptr[sj], j1 = indptr[sj + 1];
    double x = 0.0, y = 0.0;  // numer/denom or result/common
    for (int64_t t = i0 + lane; t < i1; t += 64) {
        const int32_t b = branch_id[t];
        const double l = tree_dists[b];
        const int64_t p = find_branch(branch_id, j0, j1, b);
        if (weighted) {
            const double a = abnd[t];
            if (p >= 0) { x += l * fabs(a - abnd[p]); y += l * (a + abnd[p]); }
            else { x += l * a; y += l * a; }
        } else {
            if (p >= 0) y += l; else x += l;
        }
    }
    for (int64_t t = j0 + lane; t < j1; t += 64) {
        const int32_t b = branch_id[t];
        if (find_branch(branch_id, i0, i1, b) >= 0) continue;  // counted above
        const double l = tree_dists[b];
        if (weighted) { x += l * abnd[t]; y += l * abnd[t]; } else { x += l; }
    }
    for (int m = 32; m > 0; m >>= 1) {
        x += __shfl_xor(x, m);
        y += __shfl_xor(y, m);
    }
    return weighted ? x / y : x / (x + y);
}

// The uniform sample: one wave per sampled pair, once per plan and shard.
__global__ __launch_bounds__(64)
void audit_exact_kernel(const int64_t *__restrict__ slots, const int64_t *__restrict__ indptr,
                        const int32_t *__restrict__ branch_id, const double *__restrict__ abnd,
                        const double *__restrict__ tree_dists, int weighted, int64_t slot_begin,
                        double *__restrict__ exact)
{
    int64_t si, sj;
    slot_to_pair(slot_begin + slots[blockIdx.x], &si, &sj);
    const double d = exact_pair_wave(si, sj, indptr, branch_id, abnd, tree_dists, weighted);
    if (threadIdx.x == 0) exact[blockIdx.x] = d;
}

// |got - want| / |want| as the audit counts it (NaN where the reference has NaN: 0; anything else off a NaN or a
// zero: infinite).
__device__ __forceinline__ double audit_rel_err(double got, double want)
{
    if (want != want) return got != got ? 0.0 : INFINITY;
    if (want == 0.0) return got == 0.0 ? 0.0 : INFINITY;
    const double rel = fabs(got - want) / fabs(want);
    return rel != rel ? INFINITY : rel;
}

// counters[CNT_AUDIT_FAILED] += sampled pairs further than AUDIT_REL from their binary64 value,
// counters[CNT_AUDIT_WORST] = max over the sample of the relative error (bits of a non-negative double).
__global__ void audit_compare_kernel(const int64_t *__restrict__ slots, const double *__restrict__ exact, int n,
                                     const double *__restrict__ out, unsigned long long *__restrict__ counters)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const double rel = audit_rel_err(out[slots[q]], exact[q]);
    if (!(rel <= AUDIT_REL)) atomicAdd(&counters[CNT_AUDIT_FAILED], 1ull);
    atomicMax(&counters[CNT_AUDIT_WORST], (unsigned long long)__double_as_longlong(rel));
}

// The pairs of THIS run that stand just above the refinement rule's bound (finish_pair_w's risk list): one wave each
// computes the pair in binary64 and holds what was delivered to the same bar.  A grid of RISK_CAP workgroups whatever
// the list holds (usually nothing: a few microseconds).
__global__ __launch_bounds__(64)
void audit_risk_kernel(const unsigned long long *__restrict__ risk_list, unsigned long long *__restrict__ counters,
                       const int64_t *__restrict__ indptr, const int32_t *__restrict__ branch_id,
                       const double *__restrict__ abnd, const double *__restrict__ tree_dists, int weighted,
                       int64_t slot_begin, const double *__restrict__ out)
{
    unsigned long long n = counters[CNT_RISK_FOUND];
    if (n > RISK_CAP) n = RISK_CAP;
    if (blockIdx.x >= n) return;
    const int64_t t = (int64_t)risk_list[blockIdx.x];
    int64_t si, sj;
    slot_to_pair(slot_begin + t, &si, &sj);
    const double want = exact_pair_wave(si, sj, indptr, branch_id, abnd, tree_dists, weighted);
    if (threadIdx.x != 0) return;
    const double rel = audit_rel_err(out[t], want);
    if (!(rel <= AUDIT_REL)) atomicAdd(&counters[CNT_AUDIT_FAILED], 1ull);
    atomicMax(&counters[CNT_AUDIT_WORST], (unsigned long long)__double_as_longlong(rel));
    atomicAdd(&counters[CNT_RISK_CHECKED], 1ull);
}

// Before every run of a plan that refines: counters to zero, the headroom slots to +infinity (HEADROOM_SLOTS threads).
__global__ void reset_counters_kernel(unsigned long long *__restrict__ counters)
{
    if (threadIdx.x < CNT_N) counters[threadIdx.x] = 0ull;
    reinterpret_cast<uint32_t *>(counters + CNT_N)[threadIdx.x] = 0x7F800000u;
}

// The reference's merge walk (unifrac.go:144-205) for the queued pairs, in binary64 and in the reference's
// order: bit-for-bit the reference's value.
//
// A walk is a chain of dependent loads -- id, then the branch's length -- and with one thread per pair it takes
// about a microsecond a step whatever the queue holds: 3.9 ms for C3's samples (4,000 flat nodes each), which
// a handful of queued pairs added to a pass of 0.35 ms (unweighted) or 5 ms.  Up to REFINE_BLOCK_PAIRS queued
// pairs a WORKGROUP walks a pair instead: a window of each list's ids in LDS; every element finds its place in
// the merged sequence by a binary search in the other window (ties: the first list's copy first, which then knows
// the branch is shared), its thread loads its length and abundances and leaves its terms -- every product rounded
// on its own, as the reference's are (-ffp-contract=off); +0.0 where the reference adds nothing -- at that place
// of a table in LDS; then one wave adds up the table's column for each of the reference's two sums, in order.  Same
// terms, same order, same bits (x + 0.0 = x); what is left of the waiting is the chain of additions itself.
// Longer queues keep one thread per pair (more pairs in flight than workgroups could hold).
// unifracDistWeighted / unifracDistUnweighted (unifrac.go:144-205) as written: two pointers, three cases, the tails;
// a = sample si (the higher index: IterPairs yields {s[i], s[j]}, common.go:24-26), b = sample sj.
__device__ __forceinline__ double literal_walk(int64_t si, int64_t sj, const int64_t *__restrict__ indptr,
                                               const int32_t *__restrict__ branch_id, const double *__restrict__ abnd,
                                               const double *__restrict__ tree_dists, int weighted)
{
    int64_t i = indptr[si], ie = indptr[si + 1];
    int64_t j = indptr[sj], je = indptr[sj + 1];
    double x = 0.0, y = 0.0;                      // numer/denom or result/common
    while (i < ie && j < je) {
        const int32_t ia = branch_id[i], ib = branch_id[j];
        if (ia < ib) {
            const double l = tree_dists[ia];
            if (weighted) { x += l * abnd[i]; y += l * abnd[i]; } else { x += l; }
            ++i;
        } else if (ia > ib) {
            const double l = tree_dists[ib];
            if (weighted) { x += l * abnd[j]; y += l * abnd[j]; } else { x += l; }
            ++j;
        } else {
            const double l = tree_dists[ia];
            if (weighted) { x += l * fabs(abnd[i] - abnd[j]); y += l * (abnd[i] + abnd[j]); } else { y += l; }
            ++i;
            ++j;
        }
    }
    for (; i < ie; ++i) {
        const double l = tree_dists[branch_id[i]];
        if (weighted) { x += l * abnd[i]; y += l * abnd[i]; } else { x += l; }
    }
    for (; j < je; ++j) {
        const double l = tree_dists[branch_id[j]];
        if (weighted) { x += l * abnd[j]; y += l * abnd[j]; } else { x += l; }
    }
    return weighted ? x / y : x / (x + y);
}

constexpr int REFINE_THREADS = 256;
constexpr int REFINE_WINDOW = 1024;                 // ids of each list in LDS at a time
constexpr int REFINE_BATCH = 32;                    // table entries an adding wave reads at a time
constexpr unsigned long long REFINE_BLOCK_PAIRS = 20000;   // (C3's samples: 11 pairs per workgroup 1.6 ms; a thread's walk 3.9)

__global__ __launch_bounds__(REFINE_THREADS)
void refine_exact_kernel(const unsigned long long *__restrict__ refine_list,
                         const unsigned long long *__restrict__ refine_count,
                         unsigned long long refine_cap,
                         const int64_t *__restrict__ indptr,
                         const int32_t *__restrict__ branch_id,
                         const double *__restrict__ abnd,
                         const double *__restrict__ tree_dists, int weighted,
                         int64_t slot_begin, double *__restrict__ out)
{
    __shared__ int32_t sa[REFINE_WINDOW], sb[REFINE_WINDOW];
    // what the p-th merged element adds to x (column 0) and to y (column 1)
    __shared__ __attribute__((aligned(16))) double terms[2][2 * REFINE_WINDOW + REFINE_BATCH];
    unsigned long long n = *refine_count;
    if (n > refine_cap) n = refine_cap;
    if (n <= REFINE_BLOCK_PAIRS) {
        const int tid = threadIdx.x, wave = tid >> 6;
        // (workgroups b, b + gridDim / 3 or / 4, ... share a compute unit: one round of the grid)
        const int wave_x = (int)((uint64_t)blockIdx.x * 4 / gridDim.x) & 3, wave_y = (wave_x + 2) & 3;
#ifdef FF_MFMA_DIAG  // diagnostic build: 100 MHz ticks workgroup 0 spends in each phase, over all its pairs
        unsigned long long ph[5] = {0, 0, 0, 0, 0}, t_last = __builtin_amdgcn_s_memrealtime();
#define FF_RPHASE(k) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); ph[k] += now - t_last; t_last = now; }
#else
#define FF_RPHASE(k)
#endif
        for (unsigned long long q = blockIdx.x; q < n; q += gridDim.x) {
            const int64_t t = (int64_t)refine_list[q];
            int64_t si, sj;
            slot_to_pair(slot_begin + t, &si, &sj);
            int64_t i = indptr[si], j = indptr[sj];  // a = sample i (the higher index), b = sample j
            const int64_t ie = indptr[si + 1], je = indptr[sj + 1];
            double x = 0.0, y = 0.0;                 // numer/denom or result/common (the first wave's count)
            while (i < ie || j < je) {
                const int na = (int)(ie - i < REFINE_WINDOW ? ie - i : REFINE_WINDOW);
                const int nb = (int)(je - j < REFINE_WINDOW ? je - j : REFINE_WINDOW);
                __syncthreads();  // (everybody is done with the previous window and its table)
                FF_RPHASE(4);
                for (int k = tid; k < na; k += REFINE_THREADS) sa[k] = branch_id[i + k];
                for (int k = tid; k < nb; k += REFINE_THREADS) sb[k] = branch_id[j + k];
                __syncthreads();
                FF_RPHASE(0);
                // What can be merged now: everything up to the smaller of the windows' last ids (a list that ends
                // inside its window does not limit).  ca / cb = elements of the windows not above that.
                int ca = na, cb = nb;
                if (na > 0 && nb > 0) {
                    const int32_t la = i + na == ie ? INT32_MAX : sa[na - 1], lb = j + nb == je ? INT32_MAX : sb[nb - 1];
                    const int32_t lim = la < lb ? la : lb;
                    auto count_le = [lim](const int32_t *w, int m) {
                        int lo = 0, hi = m;
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if (w[mid] <= lim) lo = mid + 1;
                            else hi = mid;
                        }
                        return lo;
                    };
                    ca = count_le(sa, na);
                    cb = count_le(sb, nb);
                }
                const int total = ca + cb, total_r = (total + REFINE_BATCH - 1) / REFINE_BATCH * REFINE_BATCH;
                FF_RPHASE(1);
                // Every element finds its own place in the merged sequence: a's element ai stands behind the lb
                // elements of b below it (ties: a's copy first), b's element bi behind the ub elements of a not above
                // it.  A thread's eight searches advance together -- selects, not branches: one round of LDS reads for
                // all of them per halving -- and then all its loads are in flight at once.
                constexpr int EPT = 2 * REFINE_WINDOW / REFINE_THREADS, HALF = EPT / 2;  // a's elements: u < HALF
                int lo[EPT], hi[EPT];
                int32_t id[EPT];
#pragma unroll
                for (int u = 0; u < EPT; ++u) {
                    const bool is_a = u < HALF;
                    const int e = tid + (u % HALF) * REFINE_THREADS;  // index in its window
                    const bool in = e < (is_a ? ca : cb);
                    id[u] = is_a ? sa[in ? e : 0] : sb[in ? e : 0];
                    lo[u] = 0;
                    hi[u] = in ? (is_a ? cb : ca) : 0;
                }
                for (int it = 0; it < 11; ++it) {  // (2^10 = REFINE_WINDOW: eleven halvings empty any range)
#pragma unroll
                    for (int u = 0; u < EPT; ++u) {
                        const bool is_a = u < HALF, go = lo[u] < hi[u];
                        const int mid = (lo[u] + hi[u]) >> 1;
                        const int32_t other = is_a ? sb[go ? mid : 0] : sa[go ? mid : 0];
                        const bool right = is_a ? other < id[u] : other <= id[u];  // lower bound in b / upper bound in a
                        lo[u] = go && right ? mid + 1 : lo[u];
                        hi[u] = go && !right ? mid : hi[u];
                    }
                }
                static_assert(REFINE_WINDOW == 1024, "refine_exact_kernel: eleven halvings per search");
                int kind[EPT];  // 0: nothing (b's copy of a shared branch, or no element), 1: only a's, 2: only b's, 3: shared
                double l[EPT], aa[EPT], ab[EPT];
#pragma unroll
                for (int u = 0; u < EPT; ++u) {
                    const bool is_a = u < HALF;
                    const int e = tid + (u % HALF) * REFINE_THREADS, r = lo[u];
                    const bool in = e < (is_a ? ca : cb);
                    if (is_a) {
                        const bool shared = in && r < cb && sb[in && r < cb ? r : 0] == id[u];
                        kind[u] = !in ? 0 : shared ? 3 : 1;
                        aa[u] = abnd[weighted && in ? i + e : 0];
                        ab[u] = abnd[weighted && shared ? j + r : 0];
                    } else {
                        const bool second = in && r > 0 && sa[in && r > 0 ? r - 1 : 0] == id[u];
                        kind[u] = in && !second ? 2 : 0;
                        aa[u] = 0.0;
                        ab[u] = abnd[weighted && kind[u] ? j + e : 0];
                    }
                    l[u] = tree_dists[kind[u] ? id[u] : 0];
                }
#pragma unroll
                for (int u = 0; u < EPT; ++u) {
                    const int e = tid + (u % HALF) * REFINE_THREADS, p = e + lo[u];
                    const bool in = e < (u < HALF ? ca : cb);
                    const double a1 = kind[u] == 2 ? ab[u] : aa[u];               // the one abundance of a lone node
                    const double wx = kind[u] == 3 ? l[u] * fabs(aa[u] - ab[u]) : l[u] * a1;
                    const double wy = kind[u] == 3 ? l[u] * (aa[u] + ab[u]) : l[u] * a1;
                    const double vx = kind[u] == 0 ? 0.0 : weighted ? wx : kind[u] == 3 ? 0.0 : l[u];
                    const double vy = kind[u] == 0 ? 0.0 : weighted ? wy : kind[u] == 3 ? l[u] : 0.0;
                    if (in) {
                        terms[0][p] = vx;
                        terms[1][p] = vy;
                    }
                }
                if (tid < total_r - total) terms[0][total + tid] = terms[1][total + tid] = 0.0;  // (+0.0 up to a whole batch)
                __syncthreads();
                FF_RPHASE(2);
                // One wave adds up x's column of the table, another y's -- two chains of dependent additions that do
                // not wait for each other, on different SIMDs, and not the same ones in workgroups that share a compute
                // unit; every lane of a wave alike (the reads are broadcasts, two entries each), a batch of entries at
                // a time (the table is padded with +0.0 to a whole batch).
                if (wave == wave_x || wave == wave_y) {
                    const double2 *col = reinterpret_cast<const double2 *>(terms[wave == wave_x ? 0 : 1]);
                    double acc = wave == wave_x ? x : y;
                    for (int k = 0; k < total; k += REFINE_BATCH) {
                        double2 v[REFINE_BATCH / 2];
#pragma unroll
                        for (int e = 0; e < REFINE_BATCH / 2; ++e) v[e] = col[k / 2 + e];
#pragma unroll
                        for (int e = 0; e < REFINE_BATCH / 2; ++e) {
                            acc += v[e].x;
                            acc += v[e].y;
                        }
                    }
                    if (wave == wave_x) x = acc;
                    else y = acc;
                }
                FF_RPHASE(3);
                i += ca;
                j += cb;
            }
            // (x lives in one wave, y in another)
            __syncthreads();
            if (tid == wave_y * 64) terms[1][0] = y;
            __syncthreads();
            if (tid == wave_x * 64) {
                y = terms[1][0];
                out[t] = weighted ? x / y : x / (x + y);
            }
        }
#ifdef FF_MFMA_DIAG
        if (g_small_stamps && tid == 0)
            for (int k = 0; k < 5; ++k) g_small_stamps[(int64_t)blockIdx.x * 8 + k] = ph[k];
#endif
#undef FF_RPHASE
        return;
    }
    for (unsigned long long q = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; q < n;
         q += (unsigned long long)gridDim.x * blockDim.x) {
        const int64_t t = (int64_t)refine_list[q];
        int64_t si, sj;
        slot_to_pair(slot_begin + t, &si, &sj);
        out[t] = literal_walk(si, sj, indptr, branch_id, abnd, tree_dists, weighted);
    }
}

// The reference's driver as it stands (unifrac.go:209-228): every pair of the shard by the literal walk, a thread
// per pair.  For lists that are NOT ascending -- what the reference's -l leaves behind (FF_FLAG_UNSORTED_WALK): the
// walk's result then depends on the order of the lists, and no reformulation of it exists.
__global__ __launch_bounds__(256)
void pair_walk_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ branch_id,
                      const double *__restrict__ abnd, const double *__restrict__ tree_dists, int weighted,
                      int64_t slot_begin, int64_t n_slots, double *__restrict__ out)
{
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_slots; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t si, sj;
        slot_to_pair(slot_begin + t, &si, &sj);
        out[t] = literal_walk(si, sj, indptr, branch_id, abnd, tree_dists, weighted);
    }
}

// EXACT64: each lane owns the pairs (i0..i0+H-1, j0+lane) and walks every branch in
// ascending id with the reference's operations.  Compiled with -ffp-contract=off.
// The tile height H changes nothing in any pair's arithmetic; it sets how many waves a shard makes and how
// many fit a SIMD (the row's H scalar operands cannot be prefetched, so a wave issues about a third of the
// time and the kernel lives on occupancy): which H is fastest depends on how the shard's tiles divide
// into rounds of resident waves, and the plan measures it (schedule_exact64).
template <bool WEIGHTED, int H>
__global__ __launch_bounds__(256)
void pair_exact64_kernel(const double *__restrict__ DT, int64_t ld,
                         const double *__restrict__ branch_len, int64_t n_branches,
                         const XTile *__restrict__ tiles, int n_tiles, int64_t row_begin,
                         int64_t row_end, int64_t slot_begin, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = blockIdx.x * 4 + wave;
    if (t >= n_tiles) return;
    const XTile tile = tiles[t];
    double a[H], c[H];  // numer/denom, or result/common
#pragma unroll
    for (int r = 0; r < H; ++r) {
        a[r] = 0.0;
        c[r] = 0.0;
    }
    const double *pj = DT + tile.j0 + lane;
    const double *pi = DT + tile.i0;
    // (the trip's H x ROWS_PER_TRIP scalar operands must fit the SGPRs: 64 of them; a shard of few waves is bound by
    // the chain of trips -- each waits out its loads -- so low tiles, many rows per trip, are what small problems want)
    constexpr int ROWS_PER_TRIP = H <= 4 ? 8 : H <= 8 ? 4 : 2;
#pragma unroll ROWS_PER_TRIP
    for (int64_t k = 0; k < n_branches; ++k) {
        const double l = branch_len[k];
        const double y = pj[k * ld];
        const double *row = pi + k * ld;
#pragma unroll
        for (int r = 0; r < H; ++r) {
            const double x = row[r];
            if (WEIGHTED) {
                a[r] = a[r] + l * fabs(x - y);  // numer += treeDists[id] * |a-b|  (:191)
                c[r] = c[r] + l * (x + y);      // denom += treeDists[id] * (a+b)  (:192)
            } else {
                a[r] = a[r] + l * fabs(x - y);  // result += treeDists[id] iff exactly one present
                c[r] = c[r] + l * (x * y);      // common += treeDists[id] iff both present
            }
        }
    }
    const int64_t j = tile.j0 + lane;
#pragma unroll
    for (int r = 0; r < H; ++r) {
        const int64_t i = tile.i0 + r;
        if (i < row_begin || i >= row_end || j >= i) continue;
        const double d = WEIGHTED ? a[r] / c[r] : a[r] / (a[r] + c[r]);
        out[i * (i - 1) / 2 - slot_begin + j] = d;
    }
}

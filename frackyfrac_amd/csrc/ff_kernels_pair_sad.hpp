// ff_kernels_pair_sad.hpp -- the v_sad_u32 pair-tile kernels (register-buffered with two or three waves per SIMD, sparse-aware) and their helpers.
// A fragment of ff_dev_run.hip: included there, once, inside its anonymous namespace
// (every kernel lives in exactly one translation unit, so the kernels stay internal and need no relocatable device code).

// D = |a - b| + c on 32-bit unsigned integers, `a` wave-uniform (SGPR).
__device__ __forceinline__ uint32_t sad_u32(uint32_t s, uint32_t v, uint32_t acc)
{
    uint32_t r;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "s"(s), "v"(v), "v"(acc));
    return r;
}

// The same in place: the accumulator keeps its register (what a kernel with no VGPR to spare needs).
__device__ __forceinline__ void sad_u32_acc(uint32_t s, uint32_t v, uint32_t &acc)
{
    asm("v_sad_u32 %0, %1, %2, %0" : "+v"(acc) : "s"(s), "v"(v));
}

// NC 32-bit values per lane of one branch row: one 16-byte (NC = 4) or 8-byte (NC = 2) load.
template <int NC> struct RowVec { uint32_t v[NC]; };
template <int NC> __device__ __forceinline__ RowVec<NC> load_row(const uint32_t *p)
{
    RowVec<NC> r;
    if constexpr (NC == 4) {
        const uint4 t = *(const uint4 *)p;
        r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    } else {
        const uint2 t = *(const uint2 *)p;
        r.v[0] = t.x; r.v[1] = t.y;
    }
    return r;
}

// One work item: a 32 x (64*NC) pair tile over the branch rows [k0, k1).
template <int NC, int KS = KSTEP>
__device__ __forceinline__ void run_item(const uint32_t *__restrict__ QT, int64_t ld, const Item item,
                                         uint32_t *__restrict__ num, int64_t plane_stride,
                                         int64_t row_begin, int64_t row_end,
                                         int64_t slot_begin, int sync_trips, int lane)
{
    const uint32_t *pj = QT + (int64_t)item.k0 * ld + item.j0 + NC * lane;
    // constant address space: the staged matrix is read-only for the whole launch, and
    // loads from it with a wave-uniform address become s_load (scalar cache) without
    // depending on the compiler's clobber analysis
    typedef const uint32_t __attribute__((address_space(4))) *const_u32_ptr;
    const_u32_ptr ps = (const_u32_ptr)(QT + (int64_t)item.k0 * ld + item.i0);
    uint32_t acc[NC][TILE_I];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < TILE_I; ++r) acc[c][r] = 0;
    // Two vector buffers of KSTEP rows each: while the rows of one are consumed the other
    // is refilled in one burst, a full KSTEP steps ahead of its first use, so the loads
    // still in flight at the loop's back edge are always old (the compiler drains vmcnt
    // there).  The 32 scalars of the next row are fetched one step ahead into the idle
    // one of two SGPR sets.
    // Rows read: the trip at k refills vB with rows k0 + k + KS .. and vA with rows k0 + k + 2 KS .. + KS - 1, so the
    // last trip (k = nk - 2 KS) reads up to row k1 + KS - 1; the scalar operands stop at row k1.  An item's length
    // is a multiple of 2 * KSTEP rows (ff_schedule.cpp), which every KS used here divides.
    static_assert(KS <= SAD_ROWS_AHEAD && (2 * KSTEP) % (2 * KS) == 0, "run_item: prefetch depth vs the matrix's slack rows (ff_schedule.hpp)");
    RowVec<NC> vA[KS], vB[KS];
#pragma unroll
    for (int d = 0; d < KS; ++d) vA[d] = load_row<NC>(pj + (int64_t)d * ld);
    const uint32_t *pv = pj + (int64_t)KS * ld;
    uint32_t sA[TILE_I], sB[TILE_I];
#pragma unroll
    for (int r = 0; r < TILE_I; ++r) sA[r] = ps[r];
    const int nk = item.k1 - item.k0;
#define FF_STEP(SCUR, SNXT, V, PREFETCH)                                        \
    {                                                                          \
        sad_u32_acc(SCUR[0], (V).v[0], acc[0][0]);                                \
        __builtin_amdgcn_sched_barrier(0);                                     \
        ps += ld;                                                              \
        _Pragma("unroll") for (int r = 0; r < TILE_I; ++r) SNXT[r] = ps[r];    \
        PREFETCH;                                                              \
        __builtin_amdgcn_sched_barrier(0);                                     \
        _Pragma("unroll") for (int r = 0; r < TILE_I; ++r) {                   \
            _Pragma("unroll") for (int c = 0; c < NC; ++c) {                   \
                if (r || c) sad_u32_acc(SCUR[r], (V).v[c], acc[c][r]);         \
            }                                                                  \
        }                                                                      \
    }
#define FF_FILL(BUF)                                                            \
    _Pragma("unroll") for (int q = 0; q < KS; ++q) {                        \
        BUF[q] = load_row<NC>(pv);                                             \
        pv += ld;                                                              \
    }
    const int sync_every = (item.flags & 2u) ? sync_trips : 0;
    int trips_left = sync_every;
    for (int k = 0; k < nk; k += 2 * KS) {
        // Items of a main round have the same length on all 8 waves of the workgroup
        // (flag bit 1): a barrier every few trips keeps them on the same rows, so the
        // older wave of each SIMD (which wins VALU arbitration) cannot run ahead and the
        // vector rows the waves share stay hot in L1/L2.
        if (sync_every && --trips_left == 0) {
            __builtin_amdgcn_s_barrier();
            trips_left = sync_every;
        }
        FF_STEP(sA, sB, vA[0], FF_FILL(vB))
#pragma unroll
        for (int d = 1; d < KS; d += 2) {
            FF_STEP(sB, sA, vA[d], )
            if (d + 1 < KS) FF_STEP(sA, sB, vA[d + 1], )
        }
        FF_STEP(sA, sB, vB[0], FF_FILL(vA))
#pragma unroll
        for (int d = 1; d < KS; d += 2) {
            FF_STEP(sB, sA, vB[d], )
            if (d + 1 < KS) FF_STEP(sA, sB, vB[d + 1], )
        }
    }
#undef FF_STEP
#undef FF_FILL
    // epilogue: slot of (i, j) is i(i-1)/2 + j (common.IterPairs, common.go:21-31)
    const int64_t j = item.j0 + NC * lane;
    const bool atomic = item.flags & 1u;
    // flag bits 3..10: the plane of accumulators this range owns among the ranges of its tile
    // (plain stores instead of memory-side atomics; finish_fixed32_kernel adds the planes)
    uint32_t *dst = num + (int64_t)((item.flags >> 3) & 255u) * plane_stride;
#pragma unroll
    for (int r = 0; r < TILE_I; ++r) {
        const int64_t i = item.i0 + r;
        if (i < row_begin || i >= row_end) continue;
        const int64_t base = i * (i - 1) / 2 - slot_begin + j;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (j + c >= i) continue;
            if (atomic) {
                if (acc[c][r]) atomicAdd(&num[base + c], acc[c][r]);
            } else {
                dst[base + c] = acc[c][r];
            }
        }
    }
}

// The pair-tile reduction.  Persistent: wave slot w runs items[item_ptr[w] .. item_ptr[w+1]).
// Per branch row a wave issues 1 coalesced 1-KiB vector load (4 samples per lane),
// 2 scalar 64-B loads (32 samples, wave-uniform) and 128 v_sad_u32.  The tile shape
// is set by the scalar path: it delivers a row's 32 operands about once per 500
// cycles per wave, so each operand has to feed 4 lanes' worth of v_sad_u32 (16
// cycles of SIMD time) for the vector ALU, not the scalar cache, to be the limit
// (measured: 32x128 tiles 27 T, 32x256 tiles 34.7 T |a-b| terms/s).
// Tiles that overhang the diagonal by more than half run as 32x128 (flag bit 2).
__global__ __launch_bounds__(WAVES_PER_WG * 64, 2)
void pair_sad_kernel(const uint32_t *__restrict__ QT, int64_t ld,
                     const Item *__restrict__ items, const int32_t *__restrict__ item_ptr,
                     uint32_t *__restrict__ num, int64_t plane_stride, int64_t row_begin, int64_t row_end,
                     int64_t slot_begin, unsigned long long *__restrict__ stamps, int sync_trips)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = blockIdx.x * WAVES_PER_WG + wave;
    const int it_begin = item_ptr[slot], it_end = item_ptr[slot + 1];
    // diagnostics (FF_STAMPS=1): start/end of every wave on the 100 MHz wall clock and, behind those, on the shader clock
    if (stamps && lane == 0) {
        stamps[2 * slot] = __builtin_amdgcn_s_memrealtime();
        stamps[2 * (gridDim.x * (blockDim.x / 64) + slot)] = __builtin_amdgcn_s_memtime();  // (the SIMD's own clock)
    }
    for (int it = it_begin; it < it_end; ++it) {
        const Item item = items[it];
        if (item.flags & 4u)
            run_item<2>(QT, ld, item, num, plane_stride, row_begin, row_end, slot_begin, sync_trips, lane);
        else
            run_item<4>(QT, ld, item, num, plane_stride, row_begin, row_end, slot_begin, sync_trips, lane);
    }
    if (stamps && lane == 0) {
        stamps[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
        stamps[2 * (gridDim.x * (blockDim.x / 64) + slot) + 1] = __builtin_amdgcn_s_memtime();
    }
}

// The same with half the vector buffers (2 x 4 rows): 168 VGPRs, three waves per SIMD.
__global__ __launch_bounds__(L_WAVES_PER_WG * 64)
void pair_sad_kernel12(const uint32_t *__restrict__ QT, int64_t ld,
                     const Item *__restrict__ items, const int32_t *__restrict__ item_ptr,
                     uint32_t *__restrict__ num, int64_t plane_stride, int64_t row_begin, int64_t row_end,
                     int64_t slot_begin, unsigned long long *__restrict__ stamps, int sync_trips)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = blockIdx.x * L_WAVES_PER_WG + wave;
    const int it_begin = item_ptr[slot], it_end = item_ptr[slot + 1];
    // diagnostics (FF_STAMPS=1): start/end of every wave on the 100 MHz wall clock and, behind those, on the shader clock
    if (stamps && lane == 0) {
        stamps[2 * slot] = __builtin_amdgcn_s_memrealtime();
        stamps[2 * (gridDim.x * (blockDim.x / 64) + slot)] = __builtin_amdgcn_s_memtime();  // (the SIMD's own clock)
    }
    for (int it = it_begin; it < it_end; ++it) {
        const Item item = items[it];
        if (item.flags & 4u)
            run_item<2, 4>(QT, ld, item, num, plane_stride, row_begin, row_end, slot_begin, sync_trips, lane);
        else
            run_item<4, 4>(QT, ld, item, num, plane_stride, row_begin, row_end, slot_begin, sync_trips, lane);
    }
    if (stamps && lane == 0) {
        stamps[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
        stamps[2 * (gridDim.x * (blockDim.x / 64) + slot) + 1] = __builtin_amdgcn_s_memtime();
    }
}

// ---- Sparse-aware variant of the pair-tile reduction ----------------------------------
//
// A branch row on which none of a tile's 32 i-samples has a flat node ("inactive" for that
// i-block) contributes |0 - q_j| = q_j to each of the tile's sums, whatever the row of the
// tile.  The wave therefore walks only the ACTIVE rows of its i-block (a precomputed list
// of row numbers), and accounts for the others in closed form:
//     U(i,j) = sum_{b active} |q_i(b) - q_j(b)|  +  R_j - sum_{b active} q_j(b),
// R_j = sum of column j over the item's branch range, from prefix sums kept every 16 rows.
// Same integers, same results.  At 10 % leaf density 2 % of the (i-block, row) cells are
// inactive, at 5 % 10 %, at 1 % 51 %, at 0.2 % 82 %: the plan picks this kernel
// when at least FF_SPARSE_MIN (default 28 %) are.  Rows past the end of the list are replaced by a
// zero slack row (|0 - 0| = 0), so the loop has no tail and no branches.
template <int NC>
__device__ __forceinline__ void run_item_sparse(const uint32_t *__restrict__ QT, int64_t ld, const Item item,
                                                const uint32_t *__restrict__ arows,
                                                const uint32_t *__restrict__ aptr16, int64_t aptr_stride,
                                                const uint32_t *__restrict__ cs16, int32_t zero_row,
                                                uint32_t *__restrict__ num, int64_t plane_stride, int64_t row_begin, int64_t row_end,
                                                int64_t slot_begin, int lane)
{
    typedef const uint32_t __attribute__((address_space(4))) *const_u32_ptr;
    const int64_t ib = item.i0 / TILE_I;
    const_u32_ptr pp = (const_u32_ptr)(aptr16 + ib * aptr_stride);
    const uint32_t a0 = pp[item.k0 / (2 * KSTEP)], a1 = pp[item.k1 / (2 * KSTEP)];
    const_u32_ptr pr = (const_u32_ptr)arows;
    const uint32_t *colj = QT + item.j0 + NC * lane;          // per-lane column base
    const_u32_ptr coli = (const_u32_ptr)(QT + item.i0);      // wave-uniform column base
    uint32_t acc[NC][TILE_I], z[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        z[c] = 0;
#pragma unroll
        for (int r = 0; r < TILE_I; ++r) acc[c][r] = 0;
    }
    // Row numbers travel in batches of four, two batches ahead of their use, so that the
    // operand loads never wait on a load of their own address: the trip at list position t (< a1) reads
    // entries t + 8 .. t + 11, i.e. up to SPARSE_LIST_AHEAD past the last real entry; the list carries
    // SPARSE_LIST_PAD spare entries (ff_schedule.hpp), and positions past the item's segment read the zero
    // slack row instead.
    constexpr int BATCH = 4, BATCHES_AHEAD = 2;
    static_assert(BATCHES_AHEAD * BATCH + BATCH - 1 == SPARSE_LIST_AHEAD, "run_item_sparse: batch prefetch vs the list's spare entries");
    auto batch = [&](uint32_t pos) -> uint4 {
        const_u32_ptr p4 = pr + pos;  // four adjacent scalar loads (one s_load_dwordx4)
        uint4 b;
        b.x = p4[0];
        b.y = p4[1];
        b.z = p4[2];
        b.w = p4[3];
        return b;
    };
    auto pick = [&](const uint4 &c, const uint4 &n, int o, uint32_t pos) -> int64_t {
        // entry o (0..7) of the two batches {c, n}; position `pos` decides whether it is real
        const uint32_t r = o == 0 ? c.x : o == 1 ? c.y : o == 2 ? c.z : o == 3 ? c.w
                         : o == 4 ? n.x : o == 5 ? n.y : o == 6 ? n.z : n.w;
        return pos < a1 ? (int64_t)r : (int64_t)zero_row;
    };
    uint4 cur = batch(a0), nxt = batch(a0 + 4);
    // vector ring of 4 active rows, scalars double-buffered one active row ahead
    RowVec<NC> v[4];
#pragma unroll
    for (int q = 0; q < 3; ++q) v[q] = load_row<NC>(colj + pick(cur, nxt, q, a0 + q) * ld);
    uint32_t sA[TILE_I], sB[TILE_I];
    {
        const_u32_ptr p0 = coli + pick(cur, nxt, 0, a0) * ld;
#pragma unroll
        for (int r = 0; r < TILE_I; ++r) sA[r] = p0[r];
    }
#define FF_ASTEP(Q, SCUR, SNXT)                                                  \
    {                                                                          \
        acc[0][0] = sad_u32(SCUR[0], v[Q].v[0], acc[0][0]);                    \
        __builtin_amdgcn_sched_barrier(0);                                     \
        {                                                                      \
            const_u32_ptr pn = coli + pick(cur, nxt, (Q) + 1, t + (Q) + 1) * ld; \
            _Pragma("unroll") for (int r = 0; r < TILE_I; ++r) SNXT[r] = pn[r]; \
            v[((Q) + 3) & 3] = load_row<NC>(colj + pick(cur, nxt, (Q) + 3, t + (Q) + 3) * ld); \
        }                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                     \
        _Pragma("unroll") for (int r = 0; r < TILE_I; ++r) {                   \
            _Pragma("unroll") for (int c = 0; c < NC; ++c) {                   \
                if (r || c) acc[c][r] = sad_u32(SCUR[r], v[Q].v[c], acc[c][r]); \
            }                                                                  \
        }                                                                      \
        _Pragma("unroll") for (int c = 0; c < NC; ++c) z[c] += v[Q].v[c];      \
    }
    for (uint32_t t = a0; t < a1; t += BATCH) {
        const uint4 nn = batch(t + BATCHES_AHEAD * BATCH);
        FF_ASTEP(0, sA, sB)
        FF_ASTEP(1, sB, sA)
        FF_ASTEP(2, sA, sB)
        FF_ASTEP(3, sB, sA)
        cur = nxt;
        nxt = nn;
    }
#undef FF_ASTEP
    // R_j over [k0, k1) from the 16-row prefix sums
    const int64_t j = item.j0 + NC * lane;
    uint32_t rj[NC];
    {
        const RowVec<NC> hi = load_row<NC>(cs16 + (int64_t)(item.k1 / (2 * KSTEP)) * ld + j);
        const RowVec<NC> lo = load_row<NC>(cs16 + (int64_t)(item.k0 / (2 * KSTEP)) * ld + j);
#pragma unroll
        for (int c = 0; c < NC; ++c) rj[c] = hi.v[c] - lo.v[c] - z[c];
    }
    const bool atomic = item.flags & 1u;
    uint32_t *dst = num + (int64_t)((item.flags >> 3) & 255u) * plane_stride;  // the range's plane
#pragma unroll
    for (int r = 0; r < TILE_I; ++r) {
        const int64_t i = item.i0 + r;
        if (i < row_begin || i >= row_end) continue;
        const int64_t base = i * (i - 1) / 2 - slot_begin + j;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (j + c >= i) continue;
            const uint32_t val = acc[c][r] + rj[c];
            if (atomic) {
                if (val) atomicAdd(&num[base + c], val);
            } else {
                dst[base + c] = val;
            }
        }
    }
}

__global__ __launch_bounds__(WAVES_PER_WG * 64, 2)
void pair_sad_sparse_kernel(const uint32_t *__restrict__ QT, int64_t ld,
                            const Item *__restrict__ items, const int32_t *__restrict__ item_ptr,
                            const uint32_t *__restrict__ arows, const uint32_t *__restrict__ aptr16,
                            int64_t aptr_stride, const uint32_t *__restrict__ cs16, int32_t zero_row,
                            uint32_t *__restrict__ num, int64_t plane_stride, int64_t row_begin, int64_t row_end,
                            int64_t slot_begin)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = blockIdx.x * WAVES_PER_WG + wave;
    const int it_begin = item_ptr[slot], it_end = item_ptr[slot + 1];
    for (int it = it_begin; it < it_end; ++it) {
        const Item item = items[it];
        if (item.flags & 4u)
            run_item_sparse<2>(QT, ld, item, arows, aptr16, aptr_stride, cs16, zero_row, num, plane_stride, row_begin, row_end,
                               slot_begin, lane);
        else
            run_item_sparse<4>(QT, ld, item, arows, aptr16, aptr_stride, cs16, zero_row, num, plane_stride, row_begin, row_end,
                               slot_begin, lane);
    }
}

__device__ __forceinline__ void slot_to_pair(int64_t k, int64_t *pi, int64_t *pj)
{
    int64_t i = (int64_t)((1.0 + sqrt(1.0 + 8.0 * (double)k)) * 0.5);
    while (i * (i - 1) / 2 > k) --i;
    while ((i + 1) * i / 2 <= k) ++i;
    *pi = i;
    *pj = k - i * (i - 1) / 2;
}

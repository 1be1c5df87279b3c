// pair_sad_lds_variant.hpp -- NOT part of the product.  The LDS-staged three-waves-per-SIMD variant of the
// weighted pair kernel as it shipped in rounds 1-2 (frackyfrac_amd/csrc/ff_kernels_pair_sad.hpp, selectable
// with FF_WAVES_PER_WG=12 without FF_REG12), removed in round 3 because it was slower on every configuration
// measured (C3 5.32 vs 4.99 ms, C4 81.0 vs 78.1, C5 108 vs 100): the LDS-DMA path costs more than
// the third wave returns.  Kept as a record of what was tried; it compiles as a fragment of ff_device.hip
// (same includes, L_RING = 8 rows of per-wave LDS ring) and is launched like pair_sad_kernel12 with
// 12 * L_RING KiB of dynamic LDS.
constexpr int L_RING = 8;  // rows of the per-wave LDS ring (1 KiB each)

// ---- LDS-staged variant: three waves per SIMD ----------------------------------------------
//
// The register-buffered kernel above keeps 16 branch rows of its vector operand in 64 VGPRs,
// which with the 128 accumulators allows two waves per SIMD.  Measured there (SQ counters):
// a wave spends 45 % of its cycles issuing v_sad_u32, 28 % in s_waitcnt (the row's scalar
// operands, an L2 round trip away) and the rest waiting for the other wave's turn; both waves
// of a SIMD wait at once 9 % of the time, and that is the idle vector ALU.  A third wave fills
// most of it, but only fits if the kernel stays under 168 VGPRs.  Here the vector rows travel
// global -> LDS by LDS-DMA (no registers) into a ring of L_RING rows per wave, 7 rows ahead,
// and come back one row ahead of their use with a single ds_read_b128: 8 VGPRs instead of 64.
// Same tiles, same integers.
template <int NC>
__device__ __forceinline__ void run_item_lds(const uint32_t *__restrict__ QT, int64_t ld, const Item item,
                                             uint32_t *__restrict__ num, int64_t plane_stride,
                                             int64_t row_begin, int64_t row_end,
                                             int64_t slot_begin, int sync_trips, int lane,
                                             uint32_t __attribute__((address_space(3))) *ring)
{
    typedef const uint32_t __attribute__((address_space(4))) *const_u32_ptr;
    typedef const void __attribute__((address_space(1))) *gptr;
    typedef void __attribute__((address_space(3))) *lptr;
    constexpr int ROW_WORDS = 64 * NC;          // one ring row: 256 (or 128) samples
    constexpr int DMA_PER_ROW = NC == 4 ? 1 : 2;  // dwordx4 per lane, or two dwords
    // per-lane source of row 0; NC == 2 has no 8-byte DMA: two dword pieces, lanes 0..63 | 64..127
    const uint32_t *src = QT + (int64_t)item.k0 * ld + item.j0 + (NC == 4 ? 4 * lane : lane);
    const_u32_ptr ps = (const_u32_ptr)(QT + (int64_t)item.k0 * ld + item.i0);
    uint32_t acc[NC][TILE_I];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < TILE_I; ++r) acc[c][r] = 0;
    auto dma = [&](int slot_, const uint32_t *g) {  // one row from g into ring slot slot_
        uint32_t __attribute__((address_space(3))) *dst = ring + slot_ * ROW_WORDS;
        if constexpr (NC == 4) {
            __builtin_amdgcn_global_load_lds((gptr)g, (lptr)dst, 16, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds((gptr)g, (lptr)dst, 4, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr)(g + 64), (lptr)(dst + 64), 4, 0, 0);
        }
    };
    auto fetch = [&](int slot_) -> RowVec<NC> {  // the lane's NC samples of a landed row
        RowVec<NC> v;
        const uint32_t __attribute__((address_space(3))) *p = ring + slot_ * ROW_WORDS + NC * lane;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        if constexpr (NC == 4) {  // one ds_read_b128
            const u32x4 t = *(const u32x4 __attribute__((address_space(3))) *)p;
            v.v[0] = t.x; v.v[1] = t.y; v.v[2] = t.z; v.v[3] = t.w;
        } else {
            const u32x2 t = *(const u32x2 __attribute__((address_space(3))) *)p;
            v.v[0] = t.x; v.v[1] = t.y;
        }
        return v;
    };
    // the previous item's ring reads are complete (their values were consumed); start the ring
    const uint32_t *pv = src;  // source of the next row to request; rows k..k+7 live in slots 0..7
#pragma unroll
    for (int q = 0; q < L_RING; ++q) {
        dma(q, pv);
        pv += ld;
    }
    uint32_t sA[TILE_I], sB[TILE_I];
#pragma unroll
    for (int r = 0; r < TILE_I; ++r) sA[r] = ps[r];
    // row 0 has landed when at most the L_RING - 1 younger rows are still in flight
    if constexpr (DMA_PER_ROW == 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    RowVec<NC> vA = fetch(0), vB;
    const int nk = item.k1 - item.k0;
    // One step (row k + D, ring slot D): the row's vector (VCUR) and scalars (SCUR) are here.  First make sure the
    // next row's DMA has landed and read it back (VNXT), request the next scalars (SNXT), and
    // refill this row's ring slot -- its ds_read completed before the step began -- with
    // the row L_RING ahead; then the 32 x NC v_sad_u32.
#define FF_LSTEP(D, SCUR, SNXT, VCUR, VNXT)                                     \
    {                                                                          \
        sad_u32_acc(SCUR[0], (VCUR).v[0], acc[0][0]);                             \
        __builtin_amdgcn_sched_barrier(0);                                     \
        /* at most the DMAs of the 6 rows after row k+D+1 may still be in flight */ \
        if constexpr (DMA_PER_ROW == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); \
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                 \
        VNXT = fetch(((D) + 1) % L_RING);                                      \
        ps += ld;                                                              \
        _Pragma("unroll") for (int r = 0; r < TILE_I; ++r) SNXT[r] = ps[r];    \
        dma(D, pv);                                                            \
        pv += ld;                                                              \
        __builtin_amdgcn_sched_barrier(0);                                     \
        _Pragma("unroll") for (int r = 0; r < TILE_I; ++r) {                   \
            _Pragma("unroll") for (int c = 0; c < NC; ++c) {                   \
                if (r || c) sad_u32_acc(SCUR[r], (VCUR).v[c], acc[c][r]);         \
            }                                                                  \
        }                                                                      \
    }
    const int sync_every = (item.flags & 2u) ? sync_trips : 0;
    int trips_left = sync_every;
    for (int k = 0; k < nk; k += L_RING) {
        if (sync_every && --trips_left == 0) {
            __builtin_amdgcn_s_barrier();
            trips_left = sync_every;
        }
#pragma unroll
        for (int d = 0; d < L_RING; d += 2) {
            FF_LSTEP(d, sA, sB, vA, vB)
            FF_LSTEP(d + 1, sB, sA, vB, vA)
        }
    }
#undef FF_LSTEP
    // drain: the ring still holds prefetched rows past k1 (slack rows of the matrix); nothing
    // may overwrite a slot while its DMA is in flight
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

__global__ __launch_bounds__(L_WAVES_PER_WG * 64)
void pair_sad_lds_kernel(const uint32_t *__restrict__ QT, int64_t ld,
                         const Item *__restrict__ items, const int32_t *__restrict__ item_ptr,
                         uint32_t *__restrict__ num, int64_t plane_stride, int64_t row_begin, int64_t row_end,
                         int64_t slot_begin, unsigned long long *__restrict__ stamps, int sync_trips)
{
    extern __shared__ uint32_t lds_ring[];  // L_WAVES_PER_WG rings of L_RING KiB
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = blockIdx.x * L_WAVES_PER_WG + wave;
    const int it_begin = item_ptr[slot], it_end = item_ptr[slot + 1];
    uint32_t __attribute__((address_space(3))) *ring =
        (uint32_t __attribute__((address_space(3))) *)lds_ring + wave * (L_RING * 256);
    if (stamps && lane == 0) {
        stamps[2 * slot] = __builtin_amdgcn_s_memrealtime();
        stamps[2 * (gridDim.x * (blockDim.x / 64) + slot)] = __builtin_amdgcn_s_memtime();  // (the SIMD's own clock)
    }
    for (int it = it_begin; it < it_end; ++it) {
        const Item item = items[it];
        if (item.flags & 4u)
            run_item_lds<2>(QT, ld, item, num, plane_stride, row_begin, row_end, slot_begin, sync_trips, lane, ring);
        else
            run_item_lds<4>(QT, ld, item, num, plane_stride, row_begin, row_end, slot_begin, sync_trips, lane, ring);
    }
    if (stamps && lane == 0) {
        stamps[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
        stamps[2 * (gridDim.x * (blockDim.x / 64) + slot) + 1] = __builtin_amdgcn_s_memtime();
    }
}


// ff_kernels_mfma_small.hpp -- unweighted UniFrac on the int8 matrix cores for shards SMALLER THAN ONE ROUND of
// pair_common_mfma_kernel: one launch does the pair reduction, the sum over the branch ranges and the division.
// A fragment of ff_dev_run.hip: included there, once, inside its anonymous namespace
// (every kernel lives in exactly one translation unit, so the kernels stay internal and need no relocatable device code).
//
// pair_common_mfma_kernel is built for throughput: 256 x 128 tiles, one persistent workgroup per CU, 8 slabs of
// prefetch, a prologue and an accumulator write-out of about 10 us per item, partial tiles summed by a second
// launch.  A shard with fewer tiles than workgroups (BASELINE configs[1]: 512 samples = 6 tiles on 256 CUs) is all
// "remainder" there: every tile cut into a dozen branch ranges, each paying the 10 us for 2 us of loop, and a pass
// is two launches -- 23.9 us for about 1 us of matrix work (round 2).  What such a shard wants is LATENCY: many
// small tiles, the whole branch sweep inside one workgroup, nothing left for a second kernel.
//
// Here a workgroup owns ONE 32 x 32 pair tile (one MFMA tile) over ALL branches: its eight waves cut the branch
// sweep into eight ranges of whole pairs of slabs (the sums are integers: any split, any order, same bits), each
// wave keeps one 32 x 32 accumulator tile per digit plane, the eight partial tiles meet in LDS, and the
// workgroup's 512 threads add them up, bring W_i + W_j and write the distances (finish_pair) -- no num[] round
// trip, no partial tiles in HBM, no atomics, no second launch.  The operands are the ones the big kernel reads
// (Pbits pairs of slabs, Kd digits in its slab order: ff_kernels_mfma.hpp), so the staging is shared and the
// results are the same integers.  C2: 136 workgroups, 16 k-steps per wave.
// Per k-step a wave spends 8 + 12 + 4 ND vector instructions on fragments for ND MFMAs (no operand reuse across
// MFMA tiles -- that is the price of the small tile), so the kernel is right where a shard leaves most SIMDs
// idle anyway; the plan takes it when tiles x k-steps is below S_MAX_WORK (schedule_mfma), FF_MFMA_SMALL forces.

constexpr int S_TILE = 32;      // pairs tile of a workgroup: one 32 x 32 MFMA tile
constexpr int S_WAVES = 8;      // branch ranges = waves of the workgroup
constexpr int S_THREADS = S_WAVES * 64;
constexpr int S_MAX_DIGITS = 5; // (stage_for_mfma: at most five base-128 digits)
constexpr double S_MAX_WORK = 300000.0;  // 32 x 32 tiles x k-steps up to which the plan prefers this kernel (tools/mfma_small_sweep.py:
                                         // 1024 x 2k leaves 12 vs 28 us, 1536 x 2k 19 vs 33, 768 x 10k 24 vs 31, 1024 x 10k 33 vs 34; beyond: the persistent kernel)
constexpr int S_DEPTH = 4;               // pairs of slabs of presence words in flight per wave
constexpr int S_TABLE_BYTES = 64 * 1024; // LDS for the digit planes (all slabs, all digits); + 32 KiB of partial tiles
constexpr int S_RED_BYTES = S_WAVES * S_TILE * S_TILE * 4;
constexpr int S_LDS_BYTES = S_TABLE_BYTES + S_RED_BYTES;  // the most a launch asks for
static_assert(M_PAD_SLABS >= 2 * S_DEPTH, "pair_common_small_kernel requests S_DEPTH pairs of slabs behind a wave's range");

// Tile t of a shard whose first row block is ib0 (rows 32 ib0 ..): row blocks in ascending order, block I with its
// I + 1 column blocks 0 .. I (the diagonal one last); t + c0 with c0 = ib0 (ib0 + 1) / 2 is the tile's position in
// the triangle of 32 x 32 blocks, so (I, J) follow from a square root -- no table of tiles, no load before the
// first operand request.  (A shard's last row block may lack its diagonal tile -- a single row whose pairs all lie
// left of it: that tile is the last of the enumeration and is simply not launched.)
__device__ __forceinline__ void small_tile_of(int64_t t, int64_t c0, int32_t *i0, int32_t *j0)
{
    const int64_t k = t + c0;
    int64_t I = (int64_t)((sqrt(8.0 * (double)k + 1.0) - 1.0) * 0.5);
    while (I * (I + 1) / 2 > k) --I;
    while ((I + 1) * (I + 2) / 2 <= k) ++I;
    *i0 = (int32_t)(I * S_TILE);
    *j0 = (int32_t)((k - I * (I + 1) / 2) * S_TILE);
}

#ifdef FF_MFMA_DIAG
// Diagnostic build: thread 0 of every workgroup stamps the phases with the 100 MHz real-time clock
// (tools/small_stamps.py): [workgroup][8].
__device__ unsigned long long *g_small_stamps = nullptr;
#define FF_SSTAMP(slot) if (g_small_stamps && tid == 0) g_small_stamps[(int64_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime()
#else
#define FF_SSTAMP(slot)
#endif
template <int NDIG>
__global__ __launch_bounds__(S_THREADS)
void pair_common_small_kernel(const uint4 *__restrict__ Pbits, int64_t n8, const int8_t *__restrict__ Kd, int64_t ldb,
                              int n_slab_pairs, int64_t tile_c0, const unsigned long long *__restrict__ W,
                              uint32_t *__restrict__ num, int64_t row_begin, int64_t row_end, int64_t slot_begin,
                              const FinishArgs fin)  // fin.out != null: distances; else integer sums into num[]
{
    // LDS (NDIG * ldb + S_RED_BYTES, given at launch): the digit planes of ALL slabs (ldb bytes per plane; the plan
    // takes this kernel only when they fit S_TABLE_BYTES), and behind them the eight partial 32 x 32 tiles of the waves.
    extern __shared__ __attribute__((aligned(16))) int8_t small_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, row = lane & 31;
    FF_SSTAMP(0);
    struct {
        int32_t i0, j0;
    } tile;
    small_tile_of(blockIdx.x, tile_c0, &tile.i0, &tile.j0);
    // this wave's share of the branch sweep: pairs of slabs [p0, p1)
    const int p0 = (int)((int64_t)n_slab_pairs * wave / S_WAVES), p1 = (int)((int64_t)n_slab_pairs * (wave + 1) / S_WAVES);
    const uint4 *pa = Pbits + (int64_t)p0 * n8 + tile.i0 + row;   // the presence words of i-row / j-row `row`
    const uint4 *pb = Pbits + (int64_t)p0 * n8 + tile.j0 + row;
    // The words of the first S_DEPTH pairs are requested before anything else; pair p + S_DEPTH is requested when
    // pair p is done.  (Requests past a wave's range stay inside the arrays or their zero padding: S_DEPTH pairs =
    // 2 * S_DEPTH slabs <= M_PAD_SLABS behind the last slab.)
    uint4 wa[S_DEPTH], wb[S_DEPTH];
#pragma unroll
    for (int q = 0; q < S_DEPTH; ++q) {
        wa[q] = pa[(int64_t)q * n8];
        wb[q] = pb[(int64_t)q * n8];
    }
    pa += (int64_t)S_DEPTH * n8;
    pb += (int64_t)S_DEPTH * n8;
    // W_i, W_j of the two pairs this thread finishes (elements tid and tid + 512 of the tile): requested now, used
    // after the sweep (W has a zero entry for every padded sample)
    const unsigned long long w_j = W[tile.j0 + (tid & 31)];
    const unsigned long long w_i0 = W[tile.i0 + (tid >> 5)], w_i1 = W[tile.i0 + (tid >> 5) + S_THREADS / S_TILE];
    static_assert(S_TILE * S_TILE == 2 * S_THREADS, "a thread finishes two pairs of the tile");
    // the digit planes -> LDS, 16 bytes per thread and trip: table[d * ldb + position]
    {
        const int pieces = (int)(ldb / 16);
        for (int c = tid; c < pieces * NDIG; c += S_THREADS) {
            const int d = c / pieces, at = c - d * pieces;
            *(mfma_v4i *)(small_lds + (int64_t)d * ldb + 16 * at) = *(const mfma_v4i *)(Kd + (int64_t)d * ldb + 16 * at);
        }
    }
    __syncthreads();
    FF_SSTAMP(1);
    // digits of (slab, k-step kt, half-wave): 16 bytes at slab * 64 + kt * 32 + half * 16 of the digit's plane
    const int8_t *pd = small_lds + (int64_t)p0 * (2 * M_KSLAB) + half * 16;
    mfma_v16i acc[NDIG];
#pragma unroll
    for (int d = 0; d < NDIG; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[d][r] = 0;
    uint32_t shk[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) shk[kk] = (uint32_t)(4 * half + kk);
    auto pair_steps = [&](const uint4 &va, const uint4 &vb) {  // the four k-steps of one pair of slabs
        const uint32_t xa[4] = {va.x, va.y, va.z, va.w}, xb[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {  // k-step c of the pair: slab 2 p + (c >> 1), k-step c & 1 of it = 32 c bytes on
            mfma_v4i fa;
            uint32_t mask[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                fa[kk] = (int)((xa[c] >> shk[kk]) & 0x01010101u);
                const uint32_t one = (xb[c] >> shk[kk]) & 0x01010101u;
                const mfma_u16x2 m16 = __builtin_bit_cast(mfma_u16x2, one) * (unsigned short)0x00FF;  // bytes 0/1 -> 0x00/0xFF
                mask[kk] = __builtin_bit_cast(uint32_t, m16);
            }
#pragma unroll
            for (int d = 0; d < NDIG; ++d) {
                const mfma_v4i dg = *(const mfma_v4i *)(pd + (int64_t)d * ldb + c * 32);
                mfma_v4i fb;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) fb[kk] = (int)((uint32_t)dg[kk] & mask[kk]);
                acc[d] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc[d], 0, 0, 0);
            }
        }
        pd += 2 * M_KSLAB;
    };
    int p = p0;
    for (; p + S_DEPTH <= p1; p += S_DEPTH) {
#pragma unroll
        for (int q = 0; q < S_DEPTH; ++q) {
            pair_steps(wa[q], wb[q]);
            wa[q] = *pa;
            wb[q] = *pb;
            pa += n8;
            pb += n8;
        }
    }
#pragma unroll
    for (int q = 0; q < S_DEPTH - 1; ++q)  // the last p1 - p < S_DEPTH pairs (their words are here already)
        if (p + q < p1) pair_steps(wa[q], wb[q]);
    FF_SSTAMP(2);
    // common = sum_d 128^d acc_d (modulo 2^32, like every sum here: the final U < 2^32) -> LDS, one tile per wave
    // D[r] of a lane: row (r & 3) + 8 (r >> 2) + 4 half, column lane & 31
    uint32_t *red = (uint32_t *)(small_lds + (int64_t)NDIG * ldb);  // [wave][32 x 32], behind the table (ldb is a multiple of 256)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        uint32_t c = 0;
#pragma unroll
        for (int d = 0; d < NDIG; ++d) c += (uint32_t)acc[d][r] << (7 * d);
        red[wave * (S_TILE * S_TILE) + ((r & 3) + 8 * (r >> 2) + 4 * half) * S_TILE + row] = c;
    }
    __syncthreads();
    FF_SSTAMP(3);
    float h2min = INFINITY;
#pragma unroll
    for (int e = tid; e < S_TILE * S_TILE; e += S_THREADS) {
        const int64_t i = tile.i0 + (e >> 5), j = tile.j0 + (e & 31);
        if (i < row_begin || i >= row_end || j >= i) continue;
        uint32_t c = 0;
#pragma unroll
        for (int w = 0; w < S_WAVES; ++w) c += red[w * (S_TILE * S_TILE) + e];
        const unsigned long long w_i = e < S_THREADS ? w_i0 : w_i1;
        const uint32_t u = (uint32_t)w_i + (uint32_t)w_j - 2u * c;  // result = W_i + W_j - 2 common
        const int64_t slot = i * (i - 1) / 2 - slot_begin + j;
        if (fin.out) finish_pair_w(fin, slot, i, j, u, w_i + w_j, h2min);
        else num[slot] = u;
    }
    finish_note_headroom(fin, h2min);
    FF_SSTAMP(4);
}
#undef FF_SSTAMP

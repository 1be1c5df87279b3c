// ff_kernels_exact_unw.hpp -- EXACT64 for UNWEIGHTED UniFrac: the reference's two running sums of
// unifracDistUnweighted (frcfrc/unifrac.go:144-171), bit for bit, from presence BITS.
// A fragment of ff_dev_run.hip: included there, once, inside its anonymous namespace
// (every kernel lives in exactly one translation unit, so the kernels stay internal and need no relocatable device code).
//
// The reference walks two ascending id lists and does, per id,
//     result += treeDists[id]    if exactly one sample has it     (:151-152,155-156,163-167)
//     common += treeDists[id]    if both have it                  (:158-159)
// and nothing for an id neither has.  A pair's two sums are two chains of binary64 ADDITIONS of branch lengths, each in
// ascending id -- no products, no |a - b|: the weighted kernel's six operations per (pair, branch)
// (pair_exact64_kernel) are four too many here.  This kernel does the additions the reference does, and on the lanes
// whose own sample has nothing to add an addition of +0.0 (x + (+0.0) = x for every x such a chain can hold: it starts
// at +0.0 and never reaches -0.0):
//
//   * a lane owns J columns j (one per 64-sample group of the tile) of H = 8 rows i; the row side is wave-uniform.
//   * per branch b a lane prepares, once for all rows, the two operands its pairs can need,
//       lj  = the bits of l_b if sample j has b, else +0.0        lnj = the bits of l_b if it has not, else +0.0
//     by AND / AND-NOT of the length's words with a mask: exact for EVERY length, no arithmetic touches them
//     (5 vector instructions per column group and branch, shared by the 8 rows);
//   * per (row, branch) the row's bit -- scalar -- picks the code:
//       row has b:      result += lnj ; common += lj            row has not:     result += lj
//     so a branch a row has not costs ONE addition per column: 1 + (density of the rows) additions per term and
//     0.625 of preparation, against 6 operations in pair_exact64_kernel.
//
// The code of the eight rows of a branch is ONE asm block (the compiler turns an if / else of additions into copies of
// the operands and a single addition behind the join: four v_mov per row) in two copies, chain X ("the row before has
// not the branch") and chain Y ("it has"); each row's code ends with the test of the next row and falls through into
// its own kind, so a branch is TAKEN only where consecutive rows differ (a taken branch costs a wave about 45 cycles,
// tools/microbench/branch_skip.hip).  tools/microbench/exact_unw_variants.hip has what was measured on the way
// (C3's shape, ms: compiler's if / else 12.9, asm cell per row 11.9, rows in one block with the present path out of
// line 10.9, scalar operands requested by hand 10.2, two chains 9.98; with EXEC-masked scalar operands instead of lj /
// lnj 18; tiles of 12 and 16 rows 9.8 - 11.2: fewer preparations, but their waves wait on more branches).
//
// Presence is staged as Xbits[slab][sample]: one 32-bit word per (32 staged rows, sample), bit k = row 32 slab + k
// (ascending branch id, after compaction), sample-minor: a wave's column words are one coalesced 256-byte load per
// 64-sample group, its 8 row words one scalar load.  C3: 10 MB instead of the 655 MB binary64 matrix of the
// weighted kernel.  Padding (ff_schedule.hpp): XU_PAD_SLABS zero slabs behind the end -- the next slab's words are
// requested while the current ones are worked on -- and the lengths padded with zeros to whole slabs.

#define XU_ABS1(R) "v_add_f64 %[a" #R "_0], %[a" #R "_0], %[p0]\n"
#define XU_ABS2(R) XU_ABS1(R) "v_add_f64 %[a" #R "_1], %[a" #R "_1], %[p1]\n"
#define XU_PRE1(R) "v_add_f64 %[a" #R "_0], %[a" #R "_0], %[n0]\n" "v_add_f64 %[c" #R "_0], %[c" #R "_0], %[p0]\n"
#define XU_PRE2(R) XU_PRE1(R) "v_add_f64 %[a" #R "_1], %[a" #R "_1], %[n1]\n" "v_add_f64 %[c" #R "_1], %[c" #R "_1], %[p1]\n"
#define XU_TEST(R) "s_bitcmp1_b32 %[w" #R "], %[k]\n"
#define XU_X(R, N, J) ".Lx" #R "_%=:\n" XU_ABS##J(R) XU_TEST(N) "s_cbranch_scc1 .Ly" #N "_%=\n"
#define XU_Y(R, N, J) ".Ly" #R "_%=:\n" XU_PRE##J(R) XU_TEST(N) "s_cbranch_scc0 .Lx" #N "_%=\n"
#define XU_XLAST(R, J) ".Lx" #R "_%=:\n" XU_ABS##J(R) "s_branch .Lend_%=\n"
#define XU_YLAST(R, J) ".Ly" #R "_%=:\n" XU_PRE##J(R)
#define XU_CHAIN(M, L, J) M(0, 1, J) M(1, 2, J) M(2, 3, J) M(3, 4, J) M(4, 5, J) M(5, 6, J) M(6, 7, J) L(7, J)
#define XU_BLOCK(J) XU_TEST(0) "s_cbranch_scc1 .Ly0_%=\n" XU_CHAIN(XU_X, XU_XLAST, J) XU_CHAIN(XU_Y, XU_YLAST, J) ".Lend_%=:"
#define XU_ROWS(M, J) M(0, J) M(1, J) M(2, J) M(3, J) M(4, J) M(5, J) M(6, J) M(7, J)
#define XU_ACC1(R, J) , [a##R##_0] "+v"(res[R][0]), [c##R##_0] "+v"(com[R][0])
#define XU_ACC2(R, J) XU_ACC1(R, J), [a##R##_1] "+v"(res[R][1]), [c##R##_1] "+v"(com[R][1])
#define XU_ACC(R, J) XU_ACC##J(R, J)
#define XU_W(R, J) , [w##R] "s"(wi[R])
#define XU_OPS1 , [n0] "v"(lnj[0])
#define XU_OPS2 XU_OPS1, [p1] "v"(lj[1]), [n1] "v"(lnj[1])

// The eight rows of branch k (bit k of the rows' words wi) for a lane's J columns.
template <int J> struct XuBranch;
// (p0, the first operand, heads the output list -- every other list starts with a comma; the block leaves it as it is)
#define XU_DEFINE(J)                                                                                                      \
    template <> struct XuBranch<J> {                                                                                      \
        static __device__ __forceinline__ void run(double (&res)[XU_TILE_H][J], double (&com)[XU_TILE_H][J],              \
                                                   const double (&lj)[J], const double (&lnj)[J],                          \
                                                   const uint32_t (&wi)[XU_TILE_H], uint32_t k)                            \
        {                                                                                                                 \
            double p0 = lj[0];                                                                                            \
            asm volatile(XU_BLOCK(J) : [p0] "+v"(p0) XU_ROWS(XU_ACC, J) : [k] "s"(k) XU_ROWS(XU_W, J) XU_OPS##J : "scc");  \
        }                                                                                                                 \
    };
XU_DEFINE(1)
XU_DEFINE(2)
static_assert(XU_TILE_H == 8, "pair_exact_unw_kernel: the asm block spells out eight rows");

typedef const __attribute__((address_space(4))) uint32_t *xu_c32;  // (the constant address space: the compiler
typedef const __attribute__((address_space(4))) uint64_t *xu_c64;  //  then fetches uniform addresses by scalar loads)

template <int J>
__device__ __forceinline__ void exact_unw_tile(const uint32_t *__restrict__ Xb, int64_t ldx,
                                               const double *__restrict__ len, int n_slabs, int i0, int j0,
                                               int64_t row_begin, int64_t row_end, int64_t slot_begin,
                                               double *__restrict__ out)
{
    constexpr int H = XU_TILE_H;
    const int lane = threadIdx.x & 63;
    double res[H][J], com[H][J];
#pragma unroll
    for (int r = 0; r < H; ++r)
#pragma unroll
        for (int t = 0; t < J; ++t) res[r][t] = com[r][t] = 0.0;
    const uint32_t *pj = Xb + j0 + lane;
    xu_c32 pi = (xu_c32)(Xb + i0);
    xu_c64 pl = (xu_c64)len;
    uint32_t wj[J], wi[H];
#pragma unroll
    for (int t = 0; t < J; ++t) wj[t] = pj[64 * t];
#pragma unroll
    for (int r = 0; r < H; ++r) wi[r] = pi[r];
    for (int s = 0; s < n_slabs; ++s) {
        // the next slab's words are requested a slab ahead (a zero slab stands behind the last one)
        uint32_t wjn[J], win[H];
        const int64_t nx = (int64_t)(s + 1) * ldx;
#pragma unroll
        for (int t = 0; t < J; ++t) wjn[t] = pj[nx + 64 * t];
#pragma unroll
        for (int r = 0; r < H; ++r) win[r] = pi[nx + r];
        // The lengths: XU_LEN_STEP at a time into scalar registers, requested and waited for on the spot -- one
        // round trip of the scalar cache per 8 branches and wave, which the other five waves of the SIMD cover.
        // (Requested a step ahead by hand -- the compiler's wait stands right behind its request -- they saved 7 %
        // at four lengths per step, tools/microbench/exact_unw_variants.hip; but an asm statement whose outputs land
        // LATER is not something the compiler can be told about: it copied a requested tuple before it had landed.)
#pragma unroll 1
        for (uint32_t k0 = 0; k0 < XU_SLAB; k0 += XU_LEN_STEP) {
            uint64_t l[XU_LEN_STEP];
#pragma unroll
            for (uint32_t q = 0; q < XU_LEN_STEP; ++q) l[q] = pl[(int64_t)s * XU_SLAB + k0 + q];
#pragma unroll
            for (uint32_t q = 0; q < XU_LEN_STEP; ++q) {
                const uint32_t k = k0 + q, llo = (uint32_t)l[q], lhi = (uint32_t)(l[q] >> 32);
                double lj[J], lnj[J];
#pragma unroll
                for (int t = 0; t < J; ++t) {
                    const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)wj[t], k, 1);  // all ones iff j has the branch
                    lj[t] = __hiloint2double((int)(lhi & m), (int)(llo & m));
                    lnj[t] = __hiloint2double((int)(lhi & ~m), (int)(llo & ~m));
                }
                XuBranch<J>::run(res, com, lj, lnj, wi, k);
            }
        }
#pragma unroll
        for (int t = 0; t < J; ++t) wj[t] = wjn[t];
#pragma unroll
        for (int r = 0; r < H; ++r) wi[r] = win[r];
    }
    static_assert(XU_SLAB % XU_LEN_STEP == 0, "exact_unw_tile: whole steps per slab");
#pragma unroll
    for (int r = 0; r < H; ++r) {
        const int64_t i = (int64_t)i0 + r;
        if (i < row_begin || i >= row_end) continue;
#pragma unroll
        for (int t = 0; t < J; ++t) {
            const int64_t j = (int64_t)j0 + 64 * t + lane;
            if (j < i) out[i * (i - 1) / 2 - slot_begin + j] = res[r][t] / (res[r][t] + com[r][t]);  // :169
        }
    }
}

// One wave per tile (64-thread workgroups: the hardware hands the next tile to whichever SIMD has room; the list is
// sorted widest first).
// Six waves per SIMD (80 registers: 64 of them accumulators): what it takes to keep the vector ALU busy across the
// waves' branches; left to itself the compiler schedules a step's preparations ahead into 108.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 8)))
void pair_exact_unw_kernel(const uint32_t *__restrict__ Xb, int64_t ldx, const double *__restrict__ len, int n_slabs,
                           const XUTile *__restrict__ tiles, int64_t row_begin, int64_t row_end, int64_t slot_begin,
                           double *__restrict__ out)
{
    const XUTile tile = tiles[blockIdx.x];
    const int i0 = __builtin_amdgcn_readfirstlane(tile.i0), j0 = __builtin_amdgcn_readfirstlane(tile.j0);
    if (__builtin_amdgcn_readfirstlane(tile.jn) == 2) exact_unw_tile<2>(Xb, ldx, len, n_slabs, i0, j0, row_begin, row_end, slot_begin, out);
    else exact_unw_tile<1>(Xb, ldx, len, n_slabs, i0, j0, row_begin, row_end, slot_begin, out);
}

// Candidates for the unweighted EXACT64 pair kernel (the reference's two chains of binary64 additions per pair,
// frcfrc/unifrac.go:144-171), timed alone on the whole triangle of N samples and checked bit for bit on sampled pairs.
//   V: lanes own columns, per-lane operands lj / lnj prepared per branch (5 vector instructions per column group),
//      the row's scalar bit picks the code path of a (row, branch) cell (inline asm: the compiler turns an if / else
//      of additions into copies of the operands and one addition behind the join).
//   E: no per-lane operands at all: the branch's presence over a column group is a 64-bit LANE MASK in scalar
//      registers (bit matrix stored branch-major), the operand is the scalar length, and EXEC selects the lanes that
//      add: result under (row has it ? ~P : P), common under P for rows that have it (skipped by a branch otherwise).
//      No vector loads, no vector instruction other than the additions.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off exact_unw_variants.hip -o exact_unw_variants
//   ./exact_unw_variants [N=4096] [rows=19999] [density=0.27]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef const __attribute__((address_space(4))) uint32_t *c32;
typedef const __attribute__((address_space(4))) uint64_t *c64;

struct XUTile { int32_t i0, j0, jn, pad; };

// ------------------------------------------------------------------ V: per-lane operands, branch per cell
template <int J> struct CellV;
template <> struct CellV<4> {
    static __device__ __forceinline__ void run(double (&res)[4], double (&com)[4], const double (&lj)[4], const double (&lnj)[4], uint32_t w, uint32_t k)
    {
        asm volatile("s_bitcmp1_b32 %16, %17\n s_cbranch_scc1 1f\n"
                     "v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %9\n v_add_f64 %2, %2, %10\n v_add_f64 %3, %3, %11\n s_branch 2f\n"
                     "1: v_add_f64 %0, %0, %12\n v_add_f64 %1, %1, %13\n v_add_f64 %2, %2, %14\n v_add_f64 %3, %3, %15\n"
                     "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %9\n v_add_f64 %6, %6, %10\n v_add_f64 %7, %7, %11\n2:"
                     : "+v"(res[0]), "+v"(res[1]), "+v"(res[2]), "+v"(res[3]), "+v"(com[0]), "+v"(com[1]), "+v"(com[2]), "+v"(com[3])
                     : "v"(lj[0]), "v"(lj[1]), "v"(lj[2]), "v"(lj[3]), "v"(lnj[0]), "v"(lnj[1]), "v"(lnj[2]), "v"(lnj[3]), "s"(w), "s"(k) : "scc");
    }
};
template <> struct CellV<2> {
    static __device__ __forceinline__ void run(double (&res)[2], double (&com)[2], const double (&lj)[2], const double (&lnj)[2], uint32_t w, uint32_t k)
    {
        asm volatile("s_bitcmp1_b32 %8, %9\n s_cbranch_scc1 1f\n"
                     "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %5\n s_branch 2f\n"
                     "1: v_add_f64 %0, %0, %6\n v_add_f64 %1, %1, %7\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %5\n2:"
                     : "+v"(res[0]), "+v"(res[1]), "+v"(com[0]), "+v"(com[1])
                     : "v"(lj[0]), "v"(lj[1]), "v"(lnj[0]), "v"(lnj[1]), "s"(w), "s"(k) : "scc");
    }
};
template <> struct CellV<1> {
    static __device__ __forceinline__ void run(double (&res)[1], double (&com)[1], const double (&lj)[1], const double (&lnj)[1], uint32_t w, uint32_t k)
    {
        asm volatile("s_bitcmp1_b32 %4, %5\n s_cbranch_scc1 1f\n"
                     "v_add_f64 %0, %0, %2\n s_branch 2f\n"
                     "1: v_add_f64 %0, %0, %3\n v_add_f64 %1, %1, %2\n2:"
                     : "+v"(res[0]), "+v"(com[0]) : "v"(lj[0]), "v"(lnj[0]), "s"(w), "s"(k) : "scc");
    }
};

template <int H, int J>
__device__ __forceinline__ void tile_v(const uint32_t *__restrict__ Xb, int64_t ldx, const double *__restrict__ len, int n_slabs,
                                       int i0, int j0, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    double res[H][J], com[H][J];
#pragma unroll
    for (int r = 0; r < H; ++r)
#pragma unroll
        for (int t = 0; t < J; ++t) res[r][t] = com[r][t] = 0.0;
    const uint32_t *pj = Xb + j0 + lane;
    c32 pi = (c32)(Xb + i0);
    c64 pl = (c64)len;
    uint32_t wj[J], wi[H];
#pragma unroll
    for (int t = 0; t < J; ++t) wj[t] = pj[64 * t];
#pragma unroll
    for (int r = 0; r < H; ++r) wi[r] = pi[r];
    for (int s = 0; s < n_slabs; ++s) {
        uint32_t wjn[J], win[H];
        const int64_t nx = (int64_t)(s + 1) * ldx;
#pragma unroll
        for (int t = 0; t < J; ++t) wjn[t] = pj[nx + 64 * t];
#pragma unroll
        for (int r = 0; r < H; ++r) win[r] = pi[nx + r];
        uint64_t lcur = pl[(int64_t)s * 32];
#pragma unroll 2
        for (uint32_t k = 0; k < 32; ++k) {
            const uint64_t lnext = pl[(int64_t)s * 32 + k + 1];
            const uint32_t llo = (uint32_t)lcur, lhi = (uint32_t)(lcur >> 32);
            double lj[J], lnj[J];
#pragma unroll
            for (int t = 0; t < J; ++t) {
                const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)wj[t], k, 1);  // all ones iff j has the branch
                const uint32_t alo = llo & m, ahi = lhi & m;
                lj[t] = __hiloint2double((int)ahi, (int)alo);
                lnj[t] = __hiloint2double((int)(ahi ^ lhi), (int)(alo ^ llo));
            }
#pragma unroll
            for (int r = 0; r < H; ++r) CellV<J>::run(res[r], com[r], lj, lnj, wi[r], k);
            lcur = lnext;
        }
#pragma unroll
        for (int t = 0; t < J; ++t) wj[t] = wjn[t];
#pragma unroll
        for (int r = 0; r < H; ++r) wi[r] = win[r];
    }
#pragma unroll
    for (int r = 0; r < H; ++r) {
        const int64_t i = (int64_t)i0 + r;
#pragma unroll
        for (int t = 0; t < J; ++t) {
            const int64_t j = (int64_t)j0 + 64 * t + lane;
            if (j < i) out[i * (i - 1) / 2 + j] = res[r][t] / (res[r][t] + com[r][t]);
        }
    }
}

template <int H, int JMAX>
__global__ __launch_bounds__(64) void kernel_v(const uint32_t *__restrict__ Xb, int64_t ldx, const double *__restrict__ len, int n_slabs,
                                               const XUTile *__restrict__ tiles, double *__restrict__ out)
{
    const XUTile tile = tiles[blockIdx.x];
    const int i0 = __builtin_amdgcn_readfirstlane(tile.i0), j0 = __builtin_amdgcn_readfirstlane(tile.j0);
    const int jn = __builtin_amdgcn_readfirstlane(tile.jn);
    if (JMAX >= 4 && jn == 4) tile_v<H, JMAX >= 4 ? 4 : 1>(Xb, ldx, len, n_slabs, i0, j0, out);
    else if (JMAX >= 2 && jn == 2) tile_v<H, JMAX >= 2 ? 2 : 1>(Xb, ldx, len, n_slabs, i0, j0, out);
    else tile_v<H, 1>(Xb, ldx, len, n_slabs, i0, j0, out);
}

// ------------------------------------------------------------------ A: as V, all rows of a branch in ONE asm block,
// the code of a row that HAS the branch out of line behind the block: a row that has not falls through (no taken branch)
#define XA_ABS1(R) "v_add_f64 %[a" #R "_0], %[a" #R "_0], %[p0]\n"
#define XA_ABS2(R) XA_ABS1(R) "v_add_f64 %[a" #R "_1], %[a" #R "_1], %[p1]\n"
#define XA_ABS4(R) XA_ABS2(R) "v_add_f64 %[a" #R "_2], %[a" #R "_2], %[p2]\n" "v_add_f64 %[a" #R "_3], %[a" #R "_3], %[p3]\n"
#define XA_PRE1(R) "v_add_f64 %[a" #R "_0], %[a" #R "_0], %[n0]\n" "v_add_f64 %[c" #R "_0], %[c" #R "_0], %[p0]\n"
#define XA_PRE2(R) XA_PRE1(R) "v_add_f64 %[a" #R "_1], %[a" #R "_1], %[n1]\n" "v_add_f64 %[c" #R "_1], %[c" #R "_1], %[p1]\n"
#define XA_PRE4(R) XA_PRE2(R) "v_add_f64 %[a" #R "_2], %[a" #R "_2], %[n2]\n" "v_add_f64 %[c" #R "_2], %[c" #R "_2], %[p2]\n" \
                   "v_add_f64 %[a" #R "_3], %[a" #R "_3], %[n3]\n" "v_add_f64 %[c" #R "_3], %[c" #R "_3], %[p3]\n"
// Two copies of the chain of rows: X_r = "row r has not the branch" (one addition per column), Y_r = "it has" (two);
// each ends with the test of row r + 1 and falls through into its own kind: a branch is TAKEN only where
// consecutive rows differ (a taken branch costs a wave about 45 cycles, one not taken about 8).
#define XA_TEST(R) "s_bitcmp1_b32 %[w" #R "], %[k]\n"
#define XA_X(R, N, J) ".Lx" #R "_%=:\n" XA_ABS##J(R) XA_TEST(N) "s_cbranch_scc1 .Ly" #N "_%=\n"
#define XA_Y(R, N, J) ".Ly" #R "_%=:\n" XA_PRE##J(R) XA_TEST(N) "s_cbranch_scc0 .Lx" #N "_%=\n"
#define XA_XLAST(R, J) ".Lx" #R "_%=:\n" XA_ABS##J(R) "s_branch .Lend_%=\n"
#define XA_YLAST(R, J) ".Ly" #R "_%=:\n" XA_PRE##J(R)
#define XA_CHAIN8(M, L, J) M(0, 1, J) M(1, 2, J) M(2, 3, J) M(3, 4, J) M(4, 5, J) M(5, 6, J) M(6, 7, J) L(7, J)
#define XA_CHAIN12(M, L, J) M(0, 1, J) M(1, 2, J) M(2, 3, J) M(3, 4, J) M(4, 5, J) M(5, 6, J) M(6, 7, J) M(7, 8, J) M(8, 9, J) M(9, 10, J) \
                            M(10, 11, J) L(11, J)
#define XA_CHAIN16(M, L, J) M(0, 1, J) M(1, 2, J) M(2, 3, J) M(3, 4, J) M(4, 5, J) M(5, 6, J) M(6, 7, J) M(7, 8, J) M(8, 9, J) M(9, 10, J) \
                            M(10, 11, J) M(11, 12, J) M(12, 13, J) M(13, 14, J) M(14, 15, J) L(15, J)
#define XA_ROWS8(M, J) M(0, J) M(1, J) M(2, J) M(3, J) M(4, J) M(5, J) M(6, J) M(7, J)
#define XA_ROWS16(M, J) XA_ROWS8(M, J) M(8, J) M(9, J) M(10, J) M(11, J) M(12, J) M(13, J) M(14, J) M(15, J)
#define XA_BLOCK(H, J) XA_TEST(0) "s_cbranch_scc1 .Ly0_%=\n" XA_CHAIN##H(XA_X, XA_XLAST, J) XA_CHAIN##H(XA_Y, XA_YLAST, J) ".Lend_%=:"
#define XA_ACC1(R, J) , [a##R##_0] "+v"(res[R][0]), [c##R##_0] "+v"(com[R][0])
#define XA_ACC2(R, J) XA_ACC1(R, J), [a##R##_1] "+v"(res[R][1]), [c##R##_1] "+v"(com[R][1])
#define XA_ACC4(R, J) XA_ACC2(R, J), [a##R##_2] "+v"(res[R][2]), [c##R##_2] "+v"(com[R][2]), [a##R##_3] "+v"(res[R][3]), [c##R##_3] "+v"(com[R][3])
#define XA_ACC(R, J) XA_ACC##J(R, J)
#define XA_W(R, J) , [w##R] "s"(wi[R])
#define XA_OPS1 , [n0] "v"(lnj[0])
#define XA_OPS2 XA_OPS1, [p1] "v"(lj[1]), [n1] "v"(lnj[1])
#define XA_OPS4 XA_OPS2, [p2] "v"(lj[2]), [n2] "v"(lnj[2]), [p3] "v"(lj[3]), [n3] "v"(lnj[3])
template <int H, int J> struct BlockA;
// (p0, the first operand, heads the output list -- every other list starts with a comma; the block leaves it as it is)
#define XA_DEFINE(H, J)                                                                                                   \
    template <> struct BlockA<H, J> {                                                                                     \
        static __device__ __forceinline__ void run(double (&res)[H][J], double (&com)[H][J], const double (&lj)[J],        \
                                                   const double (&lnj)[J], const uint32_t (&wi)[H], uint32_t k)            \
        {                                                                                                                 \
            double p0 = lj[0];                                                                                            \
            asm volatile(XA_BLOCK(H, J) : [p0] "+v"(p0) XA_ROWS##H(XA_ACC, J)                                              \
                         : [k] "s"(k) XA_ROWS##H(XA_W, J) XA_OPS##J : "scc");                                             \
        }                                                                                                                 \
    };
#define XA_ROWS12(M, J) XA_ROWS8(M, J) M(8, J) M(9, J) M(10, J) M(11, J)
XA_DEFINE(8, 1)
XA_DEFINE(8, 2)
XA_DEFINE(8, 4)
XA_DEFINE(16, 1)
XA_DEFINE(16, 2)
XA_DEFINE(12, 1)
XA_DEFINE(12, 2)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

// The H row words of a slab: requested by hand (no wait), valid after the next xa_wait that names them.
template <int H> struct RowWords;
template <> struct RowWords<8> {
    u32x8 v;
    __device__ __forceinline__ void request(const uint32_t *p) { asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(v) : "s"(p)); }
    __device__ __forceinline__ void wait() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v)); }
    __device__ __forceinline__ uint32_t get(int r) const { return v[r]; }
};
template <> struct RowWords<12> {
    u32x8 v;
    u32x4 u;
    __device__ __forceinline__ void request(const uint32_t *p)
    {
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(v) : "s"(p));
        asm volatile("s_load_dwordx4 %0, %1, 0x20" : "=&s"(u) : "s"(p));
    }
    __device__ __forceinline__ void wait() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v), "+s"(u)); }
    __device__ __forceinline__ uint32_t get(int r) const { return r < 8 ? v[r] : u[r - 8]; }
};
template <> struct RowWords<16> {
    u32x16 v;
    __device__ __forceinline__ void request(const uint32_t *p) { asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(v) : "s"(p)); }
    __device__ __forceinline__ void wait() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v)); }
    __device__ __forceinline__ uint32_t get(int r) const { return v[r]; }
};

template <int H, int J>
__device__ __forceinline__ void tile_a(const uint32_t *__restrict__ Xb, int64_t ldx, const double *__restrict__ len, int n_slabs,
                                       int i0, int j0, int64_t n_samples, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    double res[H][J], com[H][J];
#pragma unroll
    for (int r = 0; r < H; ++r)
#pragma unroll
        for (int t = 0; t < J; ++t) res[r][t] = com[r][t] = 0.0;
    const uint32_t *pj = Xb + j0 + lane;
    const uint32_t *pi = Xb + i0;
    uint32_t wj[J], wi[H];
#pragma unroll
    for (int t = 0; t < J; ++t) wj[t] = pj[64 * t];
    // Scalar operands -- the rows' words of the next slab, four lengths at a time one step (four branches) ahead --
    // are requested by hand: SMEM returns out of order, so the only wait there is waits for everything, and it has
    // to stand at the END of a step, behind the work (the compiler's own stands in front of the first use).
    RowWords<H> wn;
    u32x8 lc, ln;
    const double *lp = len;
    wn.request(pi);
    asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(lc) : "s"(lp));
    wn.wait();
    asm volatile("" : "+s"(lc));
#pragma unroll
    for (int r = 0; r < H; ++r) wi[r] = wn.get(r);
    for (int s = 0; s < n_slabs; ++s) {
        uint32_t wjn[J];
        const int64_t nx = (int64_t)(s + 1) * ldx;
#pragma unroll
        for (int t = 0; t < J; ++t) wjn[t] = pj[nx + 64 * t];
        wn.request(pi + nx);
#pragma unroll 1
        for (uint32_t k4 = 0; k4 < 32; k4 += 4) {
            lp += 4;
            asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(ln) : "s"(lp));
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                const uint32_t k = k4 + q, llo = lc[2 * q], lhi = lc[2 * q + 1];
                double lj[J], lnj[J];
#pragma unroll
                for (int t = 0; t < J; ++t) {
                    const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)wj[t], k, 1);  // all ones iff j has the branch
                    const uint32_t alo = llo & m, ahi = lhi & m;
                    lj[t] = __hiloint2double((int)ahi, (int)alo);
                    lnj[t] = __hiloint2double((int)(ahi ^ lhi), (int)(alo ^ llo));
                }
                BlockA<H, J>::run(res, com, lj, lnj, wi, k);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ln));
            lc = ln;
        }
        wn.wait();  // (nothing is outstanding by now: only the dependence)
#pragma unroll
        for (int t = 0; t < J; ++t) wj[t] = wjn[t];
#pragma unroll
        for (int r = 0; r < H; ++r) wi[r] = wn.get(r);
    }
#pragma unroll
    for (int r = 0; r < H; ++r) {
        const int64_t i = (int64_t)i0 + r;
#pragma unroll
        for (int t = 0; t < J; ++t) {
            const int64_t j = (int64_t)j0 + 64 * t + lane;
            if (j < i && i < n_samples) out[i * (i - 1) / 2 + j] = res[r][t] / (res[r][t] + com[r][t]);
        }
    }
}

template <int H, int JMAX>
__global__ __launch_bounds__(64) void kernel_a(const uint32_t *__restrict__ Xb, int64_t ldx, const double *__restrict__ len, int n_slabs,
                                               const XUTile *__restrict__ tiles, int64_t n_samples, double *__restrict__ out)
{
    const XUTile tile = tiles[blockIdx.x];
    const int i0 = __builtin_amdgcn_readfirstlane(tile.i0), j0 = __builtin_amdgcn_readfirstlane(tile.j0);
    const int jn = __builtin_amdgcn_readfirstlane(tile.jn);
    if (JMAX >= 4 && jn == 4) tile_a<H, JMAX >= 4 ? 4 : 1>(Xb, ldx, len, n_slabs, i0, j0, n_samples, out);
    else if (JMAX >= 2 && jn == 2) tile_a<H, JMAX >= 2 ? 2 : 1>(Xb, ldx, len, n_slabs, i0, j0, n_samples, out);
    else tile_a<H, 1>(Xb, ldx, len, n_slabs, i0, j0, n_samples, out);
}

// ------------------------------------------------------------------ E: lane masks in scalar registers, EXEC selects
template <int J> struct CellE;
template <> struct CellE<4> {
    template <int R> static __device__ __forceinline__ void run(double (&res)[4], double (&com)[4], const uint64_t (&P)[4], const uint64_t (&nP)[4], uint64_t l, uint32_t w)
    {
        asm volatile("s_bitcmp1_b32 %17, %18\n"
                     "s_cselect_b64 exec, %12, %8\n v_add_f64 %0, %0, %16\n"
                     "s_cselect_b64 exec, %13, %9\n v_add_f64 %1, %1, %16\n"
                     "s_cselect_b64 exec, %14, %10\n v_add_f64 %2, %2, %16\n"
                     "s_cselect_b64 exec, %15, %11\n v_add_f64 %3, %3, %16\n"
                     "s_cbranch_scc0 1f\n"
                     "s_mov_b64 exec, %8\n v_add_f64 %4, %4, %16\n"
                     "s_mov_b64 exec, %9\n v_add_f64 %5, %5, %16\n"
                     "s_mov_b64 exec, %10\n v_add_f64 %6, %6, %16\n"
                     "s_mov_b64 exec, %11\n v_add_f64 %7, %7, %16\n"
                     "1: s_mov_b64 exec, -1"
                     : "+v"(res[0]), "+v"(res[1]), "+v"(res[2]), "+v"(res[3]), "+v"(com[0]), "+v"(com[1]), "+v"(com[2]), "+v"(com[3])
                     : "s"(P[0]), "s"(P[1]), "s"(P[2]), "s"(P[3]), "s"(nP[0]), "s"(nP[1]), "s"(nP[2]), "s"(nP[3]), "s"(l), "s"(w), "n"(R) : "scc");
    }
};
template <> struct CellE<2> {
    template <int R> static __device__ __forceinline__ void run(double (&res)[2], double (&com)[2], const uint64_t (&P)[2], const uint64_t (&nP)[2], uint64_t l, uint32_t w)
    {
        asm volatile("s_bitcmp1_b32 %9, %10\n"
                     "s_cselect_b64 exec, %6, %4\n v_add_f64 %0, %0, %8\n"
                     "s_cselect_b64 exec, %7, %5\n v_add_f64 %1, %1, %8\n"
                     "s_cbranch_scc0 1f\n"
                     "s_mov_b64 exec, %4\n v_add_f64 %2, %2, %8\n"
                     "s_mov_b64 exec, %5\n v_add_f64 %3, %3, %8\n"
                     "1: s_mov_b64 exec, -1"
                     : "+v"(res[0]), "+v"(res[1]), "+v"(com[0]), "+v"(com[1])
                     : "s"(P[0]), "s"(P[1]), "s"(nP[0]), "s"(nP[1]), "s"(l), "s"(w), "n"(R) : "scc");
    }
};
template <> struct CellE<1> {
    template <int R> static __device__ __forceinline__ void run(double (&res)[1], double (&com)[1], const uint64_t (&P)[1], const uint64_t (&nP)[1], uint64_t l, uint32_t w)
    {
        asm volatile("s_bitcmp1_b32 %5, %6\n"
                     "s_cselect_b64 exec, %3, %2\n v_add_f64 %0, %0, %4\n"
                     "s_cbranch_scc0 1f\n"
                     "s_mov_b64 exec, %2\n v_add_f64 %1, %1, %4\n"
                     "1: s_mov_b64 exec, -1"
                     : "+v"(res[0]), "+v"(com[0]) : "s"(P[0]), "s"(nP[0]), "s"(l), "s"(w), "n"(R) : "scc");
    }
};
template <int H, int J, int R = 0> struct RowsE {
    static __device__ __forceinline__ void run(double (&res)[H][J], double (&com)[H][J], const uint64_t (&P)[J], const uint64_t (&nP)[J], uint64_t l, uint32_t w)
    {
        CellE<J>::template run<R>(res[R], com[R], P, nP, l, w);
        RowsE<H, J, R + 1>::run(res, com, P, nP, l, w);
    }
};
template <int H, int J> struct RowsE<H, J, H> {
    static __device__ __forceinline__ void run(double (&)[H][J], double (&)[H][J], const uint64_t (&)[J], const uint64_t (&)[J], uint64_t, uint32_t) {}
};

// Pm[k][ldq]: bit s of row k = sample s has staged row k (64-bit words; the same words read as pairs of 32-bit ones)
template <int H, int J>
__device__ __forceinline__ void tile_e(const uint64_t *__restrict__ Pm, int64_t ldq, const double *__restrict__ len, int n_rows,
                                       int i0, int j0, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    double res[H][J], com[H][J];
#pragma unroll
    for (int r = 0; r < H; ++r)
#pragma unroll
        for (int t = 0; t < J; ++t) res[r][t] = com[r][t] = 0.0;
    c64 pq = (c64)(Pm + j0 / 64);
    c32 pw = (c32)(reinterpret_cast<const uint32_t *>(Pm) + i0 / 32);
    c64 pl = (c64)len;
    const uint32_t sh = (uint32_t)i0 & 31u;
    uint64_t P[J], l;
    uint32_t w;
#pragma unroll
    for (int t = 0; t < J; ++t) P[t] = pq[t];
    w = pw[0];
    l = pl[0];
#pragma unroll 2
    for (int k = 0; k < n_rows; ++k) {
        uint64_t Pn[J], ln;
        uint32_t wn;
        const int64_t nx = (int64_t)(k + 1) * ldq;
#pragma unroll
        for (int t = 0; t < J; ++t) Pn[t] = pq[nx + t];
        wn = pw[2 * nx];
        ln = pl[k + 1];
        uint64_t nP[J];
#pragma unroll
        for (int t = 0; t < J; ++t) nP[t] = ~P[t];
        RowsE<H, J>::run(res, com, P, nP, l, w >> sh);
#pragma unroll
        for (int t = 0; t < J; ++t) P[t] = Pn[t];
        w = wn;
        l = ln;
    }
#pragma unroll
    for (int r = 0; r < H; ++r) {
        const int64_t i = (int64_t)i0 + r;
#pragma unroll
        for (int t = 0; t < J; ++t) {
            const int64_t j = (int64_t)j0 + 64 * t + lane;
            if (j < i) out[i * (i - 1) / 2 + j] = res[r][t] / (res[r][t] + com[r][t]);
        }
    }
}

template <int H, int JMAX>
__global__ __launch_bounds__(64) void kernel_e(const uint64_t *__restrict__ Pm, int64_t ldq, const double *__restrict__ len, int n_rows,
                                               const XUTile *__restrict__ tiles, double *__restrict__ out)
{
    const XUTile tile = tiles[blockIdx.x];
    const int i0 = __builtin_amdgcn_readfirstlane(tile.i0), j0 = __builtin_amdgcn_readfirstlane(tile.j0);
    const int jn = __builtin_amdgcn_readfirstlane(tile.jn);
    if (JMAX >= 4 && jn == 4) tile_e<H, JMAX >= 4 ? 4 : 1>(Pm, ldq, len, n_rows, i0, j0, out);
    else if (JMAX >= 2 && jn == 2) tile_e<H, JMAX >= 2 ? 2 : 1>(Pm, ldq, len, n_rows, i0, j0, out);
    else tile_e<H, 1>(Pm, ldq, len, n_rows, i0, j0, out);
}

// ------------------------------------------------------------------ host
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double unif() { return (double)(rnd() >> 11) * 0x1p-53; }

static std::vector<XUTile> make_tiles(int64_t N, int H, int jmax)
{
    std::vector<XUTile> tiles;
    for (int64_t i0 = 0; i0 < N; i0 += H) {
        const int64_t w = std::min<int64_t>(i0 + H - 1, N);  // columns j < w wanted
        for (int64_t j = 0; j < w;) {
            int g = jmax;
            while (g > 1 && w - j <= 64 * g / 2) g /= 2;  // the narrowest group count that covers what is left
            tiles.push_back({(int32_t)i0, (int32_t)j, g, 0});
            j += 64 * g;
        }
    }
    std::stable_sort(tiles.begin(), tiles.end(), [](const XUTile &a, const XUTile &b) { return a.jn > b.jn; });
    // XU_WIDE=n: only the first n tiles stay wider than one column group, the others are cut into single groups
    if (const char *e = getenv("XU_WIDE")) {
        const size_t keep = (size_t)atoll(e);
        std::vector<XUTile> out;
        for (size_t q = 0; q < tiles.size(); ++q) {
            if (q < keep || tiles[q].jn == 1) out.push_back(tiles[q]);
            else
                for (int t = 0; t < tiles[q].jn; ++t) out.push_back({tiles[q].i0, tiles[q].j0 + 64 * t, 1, 0});
        }
        std::stable_sort(out.begin(), out.end(), [](const XUTile &a, const XUTile &b) { return a.jn > b.jn; });
        tiles = out;
    }
    return tiles;
}

struct Ctx {
    int64_t N, R, ldx, ldq, P;
    int n_slabs;
    std::vector<uint32_t> X;
    std::vector<uint64_t> Pm;
    std::vector<double> len, got;
    uint32_t *dX;
    uint64_t *dPm;
    double *dlen, *dout;
};

static void check(Ctx &c, const char *what)
{
    CK(hipMemcpy(c.got.data(), c.dout, sizeof(double) * (size_t)c.P, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int q = 0; q < 3000; ++q) {
        int64_t i = 1 + (int64_t)(rnd() % (uint64_t)(c.N - 1)), j = (int64_t)(rnd() % (uint64_t)i);
        if (q < 200 && i > 1) j = i - 1 - (q % 2);  // next to the diagonal
        double res = 0.0, com = 0.0;
        for (int64_t b = 0; b < c.R; ++b) {
            const bool a = (c.X[(size_t)(b / 32) * c.ldx + i] >> (b % 32)) & 1u;
            const bool d = (c.X[(size_t)(b / 32) * c.ldx + j] >> (b % 32)) & 1u;
            if (a != d) res += c.len[(size_t)b];
            else if (a) com += c.len[(size_t)b];
        }
        const double want = res / (res + com), g = c.got[(size_t)(i * (i - 1) / 2 + j)];
        if (memcmp(&want, &g, 8) != 0 && bad++ < 3) printf("  MISMATCH %s (%lld,%lld): %.17g vs %.17g\n", what, (long long)i, (long long)j, g, want);
    }
    printf("  %s: %d of 3000 sampled pairs differ\n", what, bad);
    fflush(stdout);
    CK(hipMemset(c.dout, 0xFF, sizeof(double) * (size_t)c.P));
}

template <typename F> static void timeit(const char *what, size_t n_tiles, F launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    CK(hipGetLastError());
    printf("%-14s %6zu tiles  %.3f ms\n", what, n_tiles, best);
    fflush(stdout);
}

template <int H, int JMAX> static void run_v(Ctx &c)
{
    auto tiles = make_tiles(c.N, H, JMAX);
    XUTile *dt;
    CK(hipMalloc(&dt, sizeof(XUTile) * tiles.size()));
    CK(hipMemcpy(dt, tiles.data(), sizeof(XUTile) * tiles.size(), hipMemcpyHostToDevice));
    char name[64];
    snprintf(name, sizeof name, "V H=%d J<=%d", H, JMAX);
    timeit(name, tiles.size(), [&] { kernel_v<H, JMAX><<<dim3((unsigned)tiles.size()), dim3(64)>>>(c.dX, c.ldx, c.dlen, c.n_slabs, dt, c.dout); });
    check(c, name);
    CK(hipFree(dt));
}
template <int H, int JMAX> static void run_a(Ctx &c)
{
    auto tiles = make_tiles(c.N, H, JMAX);
    XUTile *dt;
    CK(hipMalloc(&dt, sizeof(XUTile) * tiles.size()));
    CK(hipMemcpy(dt, tiles.data(), sizeof(XUTile) * tiles.size(), hipMemcpyHostToDevice));
    char name[64];
    snprintf(name, sizeof name, "A H=%d J<=%d", H, JMAX);
    timeit(name, tiles.size(), [&] { kernel_a<H, JMAX><<<dim3((unsigned)tiles.size()), dim3(64)>>>(c.dX, c.ldx, c.dlen, c.n_slabs, dt, c.N, c.dout); });
    check(c, name);
    CK(hipFree(dt));
}
template <int H, int JMAX> static void run_e(Ctx &c)
{
    auto tiles = make_tiles(c.N, H, JMAX);
    XUTile *dt;
    CK(hipMalloc(&dt, sizeof(XUTile) * tiles.size()));
    CK(hipMemcpy(dt, tiles.data(), sizeof(XUTile) * tiles.size(), hipMemcpyHostToDevice));
    char name[64];
    snprintf(name, sizeof name, "E H=%d J<=%d", H, JMAX);
    timeit(name, tiles.size(), [&] { kernel_e<H, JMAX><<<dim3((unsigned)tiles.size()), dim3(64)>>>(c.dPm, c.ldq, c.dlen, (int)c.R, dt, c.dout); });
    check(c, name);
    CK(hipFree(dt));
}

int main(int argc, char **argv)
{
    Ctx c;
    c.N = argc > 1 ? atoll(argv[1]) : 4096;
    c.R = argc > 2 ? atoll(argv[2]) : 19999;
    const double dens = argc > 3 ? atof(argv[3]) : 0.27;
    c.n_slabs = (int)((c.R + 31) / 32);
    c.ldx = (c.N + 255) / 256 * 256 + 256;
    c.ldq = c.ldx / 64;
    c.P = c.N * (c.N - 1) / 2;
    c.X.assign((size_t)(c.n_slabs + 1) * c.ldx, 0u);
    c.Pm.assign((size_t)(c.R + 2) * c.ldq, 0ull);
    c.len.assign((size_t)c.n_slabs * 32 + 64, 0.0);
    for (int64_t b = 0; b < c.R; ++b) c.len[(size_t)b] = exp(3.0 * (unif() + unif() + unif() - 1.5));
    double ones = 0;
    for (int64_t b = 0; b < c.R; ++b) {  // densities vary by row as in a tree (dense near the root, sparse near the leaves); same mean
        const double d = std::min(1.0, dens * 2.0 * unif());
        for (int64_t s = 0; s < c.N; ++s)
            if (unif() < d) {
                c.X[(size_t)(b / 32) * c.ldx + s] |= 1u << (b % 32);
                c.Pm[(size_t)b * c.ldq + s / 64] |= 1ull << (s % 64);
                ones += 1;
            }
    }
    printf("N=%lld rows=%lld density %.3f\n", (long long)c.N, (long long)c.R, ones / ((double)c.N * c.R));
    c.got.resize((size_t)c.P);
    CK(hipMalloc(&c.dX, c.X.size() * 4));
    CK(hipMalloc(&c.dPm, c.Pm.size() * 8));
    CK(hipMalloc(&c.dlen, c.len.size() * 8));
    CK(hipMalloc(&c.dout, sizeof(double) * (size_t)c.P));
    CK(hipMemcpy(c.dX, c.X.data(), c.X.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(c.dPm, c.Pm.data(), c.Pm.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(c.dlen, c.len.data(), c.len.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(c.dout, 0xFF, sizeof(double) * (size_t)c.P));
    const char *only = getenv("XU_ONLY");  // e.g. "12,2"
    auto want = [&](int h, int j) { int a = 0, b = 0; return !only || (sscanf(only, "%d,%d", &a, &b) == 2 && a == h && b == j); };
    if (want(8, 4)) run_a<8, 4>(c);
    if (want(8, 2)) run_a<8, 2>(c);
    if (want(8, 1)) run_a<8, 1>(c);
    if (want(16, 2)) run_a<16, 2>(c);
    if (want(16, 1)) run_a<16, 1>(c);
    if (want(12, 2)) run_a<12, 2>(c);
    if (want(12, 1)) run_a<12, 1>(c);
    if (getenv("XU_ALL")) run_v<8, 4>(c);
    run_v<8, 2>(c);
    if (getenv("XU_ALL")) run_v<16, 2>(c);
    if (getenv("XU_ALL")) run_v<16, 1>(c);
    if (getenv("XU_ALL")) run_v<4, 4>(c);
    if (getenv("XU_ALL")) run_e<8, 4>(c);
    if (getenv("XU_ALL")) run_e<8, 2>(c);
    if (getenv("XU_ALL")) run_e<16, 2>(c);
    if (getenv("XU_ALL")) run_e<16, 1>(c);
    if (getenv("XU_ALL")) run_e<4, 4>(c);
    return 0;
}

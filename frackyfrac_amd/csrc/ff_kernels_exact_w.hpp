// ff_kernels_exact_w.hpp -- EXACT64 weighted with the reference's shortcut for branches a sample has not.
// A fragment of ff_dev_run.hip: included there, once, inside its anonymous namespace.
//
// unifracDistWeighted (frcfrc/unifrac.go:174-205) does per id
//     only one sample has it:   numer += l * x ;          denom += l * x                  (:180-181, :186-187, :196-203)
//     both have it:             numer += l * |x - y| ;    denom += l * (x + y)            (:191-192)
// pair_exact64_kernel runs the second line for every (pair, branch) -- with an absent side it gives the first line's
// bits (|0 - y| = y, 0 + y = y), and an absent / absent branch adds +0 -- : six binary64 operations per term.  Three
// quarters of a tile's (row, branch) cells belong to rows that have NOT the branch (C3: 73 %); for those the two
// products are the SAME number p = l * y, a property of the column alone: one multiplication per column and branch,
// shared by all rows, and two additions per term.  This kernel tests the row's value -- a scalar -- and takes
//     row has not the branch:   numer += p ;   denom += p              (p = l * y, once per branch)
//     row has it:               the six operations of :191-192
// 1 / H + 2 (1 - d) + 6 d operations per term at row density d: 3.2 instead of 6 at C3.  Same products, same
// additions in the same order as pair_exact64_kernel, hence as the reference: bit for bit.
//
// As in pair_exact_unw_kernel (ff_kernels_exact_unw.hpp) the H rows of a branch are ONE asm block in two chains,
// X ("the row before has not the branch") and Y ("it has"), each row's code ending with the test of the next row and
// falling through into its own kind: a branch instruction is taken only where consecutive rows differ.

#define XW_X(R, N) ".Lx" #R "_%=:\n" "v_add_f64 %[a" #R "], %[a" #R "], %[p]\n" "v_add_f64 %[c" #R "], %[c" #R "], %[p]\n" \
                   "s_cmp_eq_u64 %[x" #N "], 0\n" "s_cbranch_scc0 .Ly" #N "_%=\n"
#define XW_BOTH(R) "v_add_f64 %[t], %[x" #R "], -%[y]\n" "v_mul_f64 %[t], %[l], |%[t]|\n" "v_add_f64 %[a" #R "], %[a" #R "], %[t]\n" \
                   "v_add_f64 %[t], %[x" #R "], %[y]\n" "v_mul_f64 %[t], %[l], %[t]\n" "v_add_f64 %[c" #R "], %[c" #R "], %[t]\n"
#define XW_Y(R, N) ".Ly" #R "_%=:\n" XW_BOTH(R) "s_cmp_eq_u64 %[x" #N "], 0\n" "s_cbranch_scc1 .Lx" #N "_%=\n"
#define XW_XLAST(R) ".Lx" #R "_%=:\n" "v_add_f64 %[a" #R "], %[a" #R "], %[p]\n" "v_add_f64 %[c" #R "], %[c" #R "], %[p]\n" "s_branch .Lend_%=\n"
#define XW_YLAST(R) ".Ly" #R "_%=:\n" XW_BOTH(R)
#define XW_CHAIN8(M, L) M(0, 1) M(1, 2) M(2, 3) M(3, 4) M(4, 5) M(5, 6) M(6, 7) L(7)
#define XW_CHAIN12(M, L) M(0, 1) M(1, 2) M(2, 3) M(3, 4) M(4, 5) M(5, 6) M(6, 7) M(7, 8) M(8, 9) M(9, 10) M(10, 11) L(11)
#define XW_CHAIN16(M, L) M(0, 1) M(1, 2) M(2, 3) M(3, 4) M(4, 5) M(5, 6) M(6, 7) M(7, 8) M(8, 9) M(9, 10) M(10, 11) M(11, 12) \
                         M(12, 13) M(13, 14) M(14, 15) L(15)
#define XW_BLOCK(H) "s_cmp_eq_u64 %[x0], 0\n" "s_cbranch_scc0 .Ly0_%=\n" XW_CHAIN##H(XW_X, XW_XLAST) XW_CHAIN##H(XW_Y, XW_YLAST) ".Lend_%=:"
#define XW_ROWS8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define XW_ROWS12(M) XW_ROWS8(M) M(8) M(9) M(10) M(11)
#define XW_ROWS16(M) XW_ROWS12(M) M(12) M(13) M(14) M(15)
#define XW_ACC(R) , [a##R] "+v"(a[R]), [c##R] "+v"(c[R])
#define XW_XS(R) , [x##R] "s"(x[R])

// The H rows of one branch: a[], c[] = the lane's numerators and denominators, x[] = the rows' values (scalar),
// y = the lane's column value, l = the branch length, p = l * y.  ("memory": the block touches none, but the requests
// of the NEXT branch's operands, which stand in front of it in the source, must not sink behind it.)
template <int H> struct XwBranch;
#define XW_DEFINE(H)                                                                                                      \
    template <> struct XwBranch<H> {                                                                                      \
        static __device__ __forceinline__ void run(double (&a)[H], double (&c)[H], const double (&x)[H], double y, double l, \
                                                   double p)                                                               \
        {                                                                                                                 \
            double t;  /* (one temporary for both products: 64 registers at 12 rows, eight waves per SIMD) */               \
            asm volatile(XW_BLOCK(H) : [t] "=&v"(t) XW_ROWS##H(XW_ACC)                                                     \
                         : [y] "v"(y), [l] "s"(l), [p] "v"(p) XW_ROWS##H(XW_XS) : "scc", "memory");                        \
        }                                                                                                                 \
    };
XW_DEFINE(8)
XW_DEFINE(12)
XW_DEFINE(16)

// H x 64 tile per wave, one tile per wave, the full branch range in order (as pair_exact64_kernel<true, H>).
template <int H>
__global__ __launch_bounds__(256)
void pair_exact64_skip_kernel(const double *__restrict__ DT, int64_t ld, const double *__restrict__ branch_len,
                              int64_t n_branches, const XTile *__restrict__ tiles, int n_tiles, int64_t row_begin,
                              int64_t row_end, int64_t slot_begin, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = blockIdx.x * 4 + wave;
    if (t >= n_tiles) return;
    const XTile tile = tiles[t];
    double a[H], c[H];  // numer / denom
#pragma unroll
    for (int r = 0; r < H; ++r) {
        a[r] = 0.0;
        c[r] = 0.0;
    }
    typedef const __attribute__((address_space(4))) double *cdouble;  // (uniform addresses: scalar loads)
    const double *pj = DT + tile.j0 + lane;
    cdouble pi = (cdouble)(DT + tile.i0);
    cdouble pl = (cdouble)branch_len;
    // Two sets of operands -- the rows' H values and the length in scalar registers, the lane's column value -- taken
    // in turn: branch k + 1's are requested before branch k is worked on.  SMEM returns out of order, so the only wait
    // there is waits for everything outstanding; it has to stand BEFORE the next request (the empty asm statements
    // below use the current set, which makes the compiler wait there, and keep the requests behind them).
    double xa[H], xb[H], la, lb, ya, yb;
    auto request = [&](int64_t k, double (&x)[H], double &l, double &y) {
        const int64_t kk = k < n_branches ? k : n_branches - 1;  // (one past the end: the last row again, never used)
        l = pl[kk];
        y = pj[kk * ld];
#pragma unroll
        for (int r = 0; r < H; ++r) x[r] = pi[kk * ld + r];
    };
    request(0, xa, la, ya);  // (n_branches >= 1: a problem without branches never takes this kernel, upload_exact64_tiles)
    // (pairs of branches in ONE basic block -- an exit test between the two halves lets the compiler sink the first
    // half's requests behind its work, into the block that uses them --; an odd last branch behind the loop)
    const int64_t n_even = n_branches & ~(int64_t)1;
    for (int64_t k = 0; k < n_even; k += 2) {
        asm volatile("" : : "s"(xa[0]), "s"(xa[H - 1]), "s"(la), "v"(ya) : "memory");
        request(k + 1, xb, lb, yb);
        XwBranch<H>::run(a, c, xa, ya, la, la * ya);  // la * ya = treeDists[id] * b.abnd (:186): the product of the one-sided cases
        asm volatile("" : : "s"(xb[0]), "s"(xb[H - 1]), "s"(lb), "v"(yb) : "memory");
        request(k + 2, xa, la, ya);
        XwBranch<H>::run(a, c, xb, yb, lb, lb * yb);
    }
    if (n_even < n_branches) XwBranch<H>::run(a, c, xa, ya, la, la * ya);
    const int64_t j = tile.j0 + lane;
#pragma unroll
    for (int r = 0; r < H; ++r) {
        const int64_t i = tile.i0 + r;
        if (i < row_begin || i >= row_end || j >= i) continue;
        out[i * (i - 1) / 2 - slot_begin + j] = a[r] / c[r];  // :204
    }
}

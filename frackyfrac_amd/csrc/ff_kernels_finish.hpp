// ff_kernels_finish.hpp -- integer sums -> distances, exact refinement of nearly equal pairs, the EXACT64 pair kernel.
// A fragment of ff_device.hip: included there, once, inside its anonymous namespace
// (one translation unit, so the kernels stay internal and need no relocatable device code).

// Integer sums -> distances (unifrac.go:169 and :204), IEEE binary64 division.
//
// Rounding every staged value to an integer leaves U with an error of about
// sqrt((k_i + k_j) / 12) units (k = flat nodes of the sample).  Where U is so small
// that this could exceed REFINE_REL of U -- nearly identical samples -- the pair is
// queued for refine_exact_kernel, which recomputes it with the reference's own
// binary64 merge walk; all other pairs already meet the tolerance.
constexpr double REFINE_REL = 0.5e-6;   // half of the 1e-6 relative bar of BASELINE.json
constexpr double REFINE_SIGMAS = 6.0;

__global__ void finish_fixed32_kernel(const uint32_t *__restrict__ num,
                                      int n_planes, int64_t plane_stride,  // the ranges of a split tile own a plane each
                                      const unsigned long long *__restrict__ W, int weighted,
                                      int64_t slot_begin, int64_t n_slots,
                                      double *__restrict__ out,
                                      const int64_t *__restrict__ indptr,  // null: no refinement
                                      unsigned long long *__restrict__ refine_list,
                                      unsigned long long *__restrict__ refine_count,
                                      unsigned long long refine_cap)
{
    // grid-stride: a launch carries at most 2^32 - 1 threads, a shard can have more slots
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_slots;
         t += (int64_t)gridDim.x * blockDim.x) {
        int64_t i, j;
        slot_to_pair(slot_begin + t, &i, &j);
        uint32_t u32 = num[t];
        for (int q = 1; q < n_planes; ++q) u32 += num[(int64_t)q * plane_stride + t];
        const unsigned long long u = u32;
        const unsigned long long w = W[i] + W[j];
        double d;
        if (weighted) {
            d = (double)u / (double)w;                 // numer / denom
        } else {
            const unsigned long long common = (w - u) >> 1;  // exact: w - u = 2 * common
            d = (double)u / (double)(u + common);      // result / (result + common)
        }
        out[t] = d;
        if (indptr && w != 0) {
            const double k = (double)((indptr[i + 1] - indptr[i]) + (indptr[j + 1] - indptr[j]));
            const double err = REFINE_SIGMAS * sqrt(k * (1.0 / 12.0)) + 1.0;
            if ((double)u * REFINE_REL < err) {
                const unsigned long long at = atomicAdd(refine_count, 1ull);
                if (at < refine_cap) refine_list[at] = (unsigned long long)t;
            }
        }
    }
}

// The reference's merge walk (unifrac.go:144-205) for the queued pairs, one thread per
// pair, in binary64 and in the reference's order: bit-for-bit the reference's value.
__global__ void refine_exact_kernel(const unsigned long long *__restrict__ refine_list,
                                    const unsigned long long *__restrict__ refine_count,
                                    unsigned long long refine_cap,
                                    const int64_t *__restrict__ indptr,
                                    const int32_t *__restrict__ branch_id,
                                    const double *__restrict__ abnd,
                                    const double *__restrict__ tree_dists, int weighted,
                                    int64_t slot_begin, double *__restrict__ out)
{
    unsigned long long n = *refine_count;
    if (n > refine_cap) n = refine_cap;
    for (unsigned long long q = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; q < n;
         q += (unsigned long long)gridDim.x * blockDim.x) {
        const int64_t t = (int64_t)refine_list[q];
        int64_t si, sj;
        slot_to_pair(slot_begin + t, &si, &sj);
        int64_t i = indptr[si], ie = indptr[si + 1];  // a = sample i (the higher index)
        int64_t j = indptr[sj], je = indptr[sj + 1];  // b = sample j
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
        out[t] = weighted ? x / y : x / (x + y);
    }
}

// EXACT64: each lane owns the pairs (i0..i0+15, j0+lane) and walks every branch in
// ascending id with the reference's operations.  Compiled with -ffp-contract=off.
template <bool WEIGHTED>
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
    double a[X_TILE_I], c[X_TILE_I];  // numer/denom, or result/common
#pragma unroll
    for (int r = 0; r < X_TILE_I; ++r) {
        a[r] = 0.0;
        c[r] = 0.0;
    }
    const double *pj = DT + tile.j0 + lane;
    const double *pi = DT + tile.i0;
#pragma unroll 2
    for (int64_t k = 0; k < n_branches; ++k) {
        const double l = branch_len[k];
        const double y = pj[k * ld];
        const double *row = pi + k * ld;
#pragma unroll
        for (int r = 0; r < X_TILE_I; ++r) {
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
    for (int r = 0; r < X_TILE_I; ++r) {
        const int64_t i = tile.i0 + r;
        if (i < row_begin || i >= row_end || j >= i) continue;
        const double d = WEIGHTED ? a[r] / c[r] : a[r] / (a[r] + c[r]);
        out[i * (i - 1) / 2 - slot_begin + j] = d;
    }
}

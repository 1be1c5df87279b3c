// ff_kernels_stage_a.hpp -- stage A on the device: subtree sums level by level, normaliser, CSR fill.
// A fragment of ff_dev_stage.hip: included there, once, inside its anonymous namespace
// (every kernel lives in exactly one translation unit, so the kernels stay internal and need no relocatable device code).

// ---- Stage A on the device (frcfrc/unifrac.go:32-67): subtree sums and normaliser ----
//
// S[b][s] (binary64, branch-major) starts as the leaf values.  Internal nodes are then
// filled level by level from the deepest level up; a node adds its children's sums in
// ascending child order, which is the order of the reference's recursion (:35-37), so
// every sum carries the reference's roundings.

__global__ void stage_a_scatter_kernel(const int64_t *__restrict__ leaf_ptr,
                                       const int64_t *__restrict__ leaf_idx,
                                       const double *__restrict__ leaf_val,
                                       const int64_t *__restrict__ size, double *__restrict__ S,
                                       int64_t ld)
{
    const int64_t s = blockIdx.x;
    for (int64_t t = leaf_ptr[s] + threadIdx.x; t < leaf_ptr[s + 1]; t += blockDim.x) {
        const int64_t id = leaf_idx[t];
        const double a = leaf_val[t];
        if (size[id] == 1 && a > 0) S[id * ld + s] = a;  // leaves only, if a > 0 (:39-42)
    }
}

// One tree level: nodes[level_begin .. level_end) are the internal nodes of that level.
__global__ void stage_a_level_kernel(const int32_t *__restrict__ nodes, int level_begin, int level_end,
                                     const int64_t *__restrict__ child_ptr,
                                     const int32_t *__restrict__ child_idx, double *__restrict__ S,
                                     int64_t ld, int64_t n_samples)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n = level_begin + blockIdx.y;
    if (s >= n_samples || n >= level_end) return;
    const int32_t id = nodes[n];
    double sum = 0.0;
    for (int64_t c = child_ptr[id]; c < child_ptr[id + 1]; ++c) sum += S[(int64_t)child_idx[c] * ld + s];
    S[(int64_t)id * ld + s] = sum;
}

// Per sample: number of flat nodes (sum > 0, :49) and normalizeFlatNodes' divisor: the sum
// of ALL flat-node abundances in ascending id (:60-63).
__global__ void stage_a_count_kernel(const double *__restrict__ S, int64_t ld, int64_t n_branches,
                                     int64_t n_samples, int64_t *__restrict__ count,
                                     double *__restrict__ divisor)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_samples) return;
    double total = 0.0;
    int64_t k = 0;
#pragma unroll 8
    for (int64_t b = 0; b < n_branches; ++b) {
        const double v = S[b * ld + s];
        if (v > 0) {
            total += v;
            ++k;
        }
    }
    count[s] = k;
    divisor[s] = total;
}

// Dense sums -> CSR flat nodes in ascending id, normalised (:64-66) unless -l; also the
// sample's weight sum_b l_b * x_s(b), which only steers the choice of the fixed-point scale.
__global__ void stage_a_fill_kernel(const double *__restrict__ S, int64_t ld, int64_t n_branches,
                                    int64_t n_samples, const int64_t *__restrict__ indptr,
                                    const double *__restrict__ divisor, int normalize,
                                    const double *__restrict__ branch_len,
                                    const int32_t *__restrict__ node_of,  // row of S -> branch id (null: identity)
                                    int32_t *__restrict__ ids, double *__restrict__ abnd,
                                    double *__restrict__ weight)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_samples) return;
    int64_t pos = indptr[s];
    const double d = divisor[s];
    double w = 0.0;
#pragma unroll 8
    for (int64_t r = 0; r < n_branches; ++r) {
        const double v = S[r * ld + s];
        if (v > 0) {
            const double x = normalize ? v / d : v;
            const int32_t b = node_of ? node_of[r] : (int32_t)r;
            ids[pos] = b;
            abnd[pos] = x;
            w += branch_len[b] * x;
            ++pos;
        }
    }
    weight[s] = w;
}

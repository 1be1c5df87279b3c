// ff_kernels_stage.hpp -- staging of the flat nodes into the dense matrices (FIXED32 / EXACT64), column sums, branch marks.
// A fragment of ff_device.hip: included there, once, inside its anonymous namespace
// (one translation unit, so the kernels stay internal and need no relocatable device code).


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

// Stage FIXED32: one workgroup per sample scatters its flat nodes into column s.
// Weighted values are rounded with the branch's shared offset (ff_dither.hpp): q = floor(v + u_b).
__global__ void stage_fixed32_kernel(const int64_t *__restrict__ indptr,
                                     const int32_t *__restrict__ branch_id,
                                     const double *__restrict__ abnd,
                                     const double *__restrict__ branch_len,
                                     const uint32_t *__restrict__ klen, int weighted, int e,
                                     const int32_t *__restrict__ row_of,  // branch id -> staged row (null: identity)
                                     uint32_t *__restrict__ QT, int64_t ld)
{
    const int64_t s = blockIdx.x;
    const int64_t b0 = indptr[s], b1 = indptr[s + 1];
    for (int64_t t = b0 + threadIdx.x; t < b1; t += blockDim.x) {
        const int32_t b = branch_id[t];
        uint32_t q;
        if (weighted) {
            const double x = branch_len[b] * abnd[t];  // treeDists[id] * abnd (unifrac.go:180)
            q = (uint32_t)(unsigned long long)floor(ldexp(x, e) + ff::branch_dither(b));
        } else {
            q = klen[b];
        }
        QT[(int64_t)(row_of ? row_of[b] : b) * ld + s] = q;
    }
}

// The binary64 side of a FIXED32 distance: wex_s = sum_b l_b * x_s(b) (weighted: the sample's share
// of every denominator, unifrac.go:181,187,192) or sum_b l_b over its flat nodes (unweighted:
// result + common = (wex_i + wex_j + result) / 2).  One workgroup per sample; a fixed tree of
// additions, so the value does not depend on the launch.  Only the integer NUMERATOR carries the
// staging's rounding; the denominator comes from here (finish_fixed32_kernel).
__global__ __launch_bounds__(256)
void exact_weight_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ branch_id,
                         const double *__restrict__ abnd, const double *__restrict__ branch_len,
                         int weighted, double *__restrict__ wex)
{
    __shared__ double part[256];
    const int64_t s = blockIdx.x;
    double acc = 0.0;
    for (int64_t t = indptr[s] + threadIdx.x; t < indptr[s + 1]; t += 256)
        acc += branch_len[branch_id[t]] * (weighted ? abnd[t] : 1.0);
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) wex[s] = part[0];
}

// Branch compaction: which branches carry a flat node of any sample.
__global__ void mark_branches_kernel(const int32_t *__restrict__ branch_id, int64_t nnz, unsigned char *__restrict__ mark)
{
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * blockDim.x)
        mark[branch_id[t]] = 1;
}

// Stage EXACT64: the abundance itself (weighted) or 1.0 for presence (unweighted).
__global__ void stage_exact64_kernel(const int64_t *__restrict__ indptr,
                                     const int32_t *__restrict__ branch_id,
                                     const double *__restrict__ abnd, int weighted,
                                     const int32_t *__restrict__ row_of, double *__restrict__ DT, int64_t ld)
{
    const int64_t s = blockIdx.x;
    const int64_t b0 = indptr[s], b1 = indptr[s + 1];
    for (int64_t t = b0 + threadIdx.x; t < b1; t += blockDim.x) {
        const int32_t b = branch_id[t];
        DT[(int64_t)(row_of ? row_of[b] : b) * ld + s] = weighted ? abnd[t] : 1.0;
    }
}

// W_s = sum_b q_s(b): grid (ld/64, row chunks), one column per lane.
__global__ void colsum_kernel(const uint32_t *__restrict__ QT, int64_t ld, int64_t rows,
                              int64_t rows_per_block, unsigned long long *__restrict__ W)
{
    const int64_t s = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = min(rows, r0 + rows_per_block);
    unsigned long long acc = 0;
    for (int64_t r = r0; r < r1; ++r) acc += QT[r * ld + s];
    if (acc) atomicAdd(&W[s], acc);
}

// NC 32-bit values per lane of one branch row: one 16-byte (NC = 4) or 8-byte (NC = 2) load.

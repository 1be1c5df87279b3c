// ff_kernels_stage.hpp -- staging of the flat nodes into what the pair kernels read: the dense matrices (FIXED32 / EXACT64), presence
// words and bits, column sums, activity lists, branch marks.
// A fragment of ff_dev_stage.hip: included there, once, inside its anonymous namespace
// (every kernel lives in exactly one translation unit, so the kernels stay internal and need no relocatable device code).


// The staged value of one weighted flat node: floor(l_b x 2^e + u_b), u_b the branch's shared offset (ff_dither.hpp).
__device__ __forceinline__ uint32_t stage_q_weighted(double len, double x_abnd, int e, int32_t b)
{
    const double x = len * x_abnd;  // treeDists[id] * abnd (unifrac.go:180)
    return (uint32_t)(unsigned long long)floor(ldexp(x, e) + ff::branch_dither(b));
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
        const int64_t row = row_of ? row_of[b] : b;
        if (row < 0) continue;  // (a rare row of a sparse table: it lives in the lists of pair_low_kernel)
        uint32_t q;
        if (weighted) {
            q = stage_q_weighted(branch_len[b], abnd[t], e, b);
        } else {
            q = klen[b];
        }
        QT[row * ld + s] = q;
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

// cs16[t][s] = sum of column s over the rows [0, 16 t).  Three launches over chunks of PFX_CHUNK_ROWS rows -- the
// chunks' sums, their exclusive scan per column, the marks of every chunk from its base -- so that a column is walked
// by rows / PFX_CHUNK_ROWS threads instead of one (one thread per column took 38 ms for C5's 100,000 rows: more than
// a pass of the kernel the sums are for; two reads of the staged matrix at the memory's rate take 2).
constexpr int PFX_CHUNK_ROWS = 1024;  // a multiple of 16
__global__ void prefix16_sums_kernel(const uint32_t *__restrict__ QT, int64_t ld, int64_t rows,
                                     uint32_t *__restrict__ chunk_sums)  // [chunk][ld]
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ld) return;
    const int64_t r0 = (int64_t)blockIdx.y * PFX_CHUNK_ROWS, r1 = min(rows, r0 + PFX_CHUNK_ROWS);
    uint32_t run = 0;
    for (int64_t r = r0; r < r1; ++r) run += QT[r * ld + s];
    chunk_sums[(int64_t)blockIdx.y * ld + s] = run;
}

__global__ void prefix16_scan_kernel(uint32_t *__restrict__ chunk_sums, int64_t ld, int64_t n_chunks)  // -> exclusive, in place
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ld) return;
    uint32_t run = 0;
    for (int64_t c = 0; c < n_chunks; ++c) {
        const uint32_t v = chunk_sums[c * ld + s];
        chunk_sums[c * ld + s] = run;
        run += v;
    }
}

__global__ void prefix16_fill_kernel(const uint32_t *__restrict__ QT, int64_t ld, int64_t rows,
                                     const uint32_t *__restrict__ chunk_base, uint32_t *__restrict__ cs16)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ld) return;
    const int64_t r0 = (int64_t)blockIdx.y * PFX_CHUNK_ROWS, r1 = min(rows, r0 + PFX_CHUNK_ROWS);
    uint32_t run = chunk_base[(int64_t)blockIdx.y * ld + s];
    if (blockIdx.y == 0) cs16[s] = 0;
    for (int64_t r = r0; r < r1; ++r) {
        run += QT[r * ld + s];
        if ((r & 15) == 15) cs16[((r >> 4) + 1) * ld + s] = run;
    }
}

// act64[iblock][w] bit r: branch row 64 w + r has a non-zero value among the 32 samples of
// i-block `iblock`.  One wave per (i-block, 64 rows), lane = row.
__global__ void build_activity_kernel(const uint32_t *__restrict__ QT, int64_t ld, int64_t rows, int64_t words,
                                      unsigned long long *__restrict__ act64)
{
    const int64_t w = blockIdx.x, iblock = blockIdx.y;
    const int64_t row = w * 64 + threadIdx.x;
    uint32_t any = 0;
    if (row < rows) {
        const uint4 *p = (const uint4 *)(QT + row * ld + iblock * TILE_I);
#pragma unroll
        for (int q = 0; q < TILE_I / 4; ++q) {
            const uint4 t = p[q];
            any |= t.x | t.y | t.z | t.w;
        }
    }
    const unsigned long long mask = __ballot(any != 0);
    if (threadIdx.x == 0) act64[iblock * words + w] = mask;
}

// Presence bits from the flat nodes: one workgroup per sample builds the sample's bitmap in LDS,
// 65,536 branch rows at a time, and stores it slab by slab.  Also W_s = sum of the sample's
// integer branch lengths (what colsum_kernel gives the SAD path).
__global__ __launch_bounds__(256)
void stage_mfma_bits_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ branch_id,
                            const uint32_t *__restrict__ klen, const int32_t *__restrict__ row_of,
                            const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ row_list,
                            unsigned long long *__restrict__ Pbits, int64_t n8, int64_t n_slabs,
                            unsigned long long *__restrict__ W)
{
    // a branch's staged rows: row_list[row_ptr[b] .. row_ptr[b + 1]) (graded staging: one or more, anywhere), else
    // the one row row_of[b], else row b
    __shared__ uint32_t bm[2048];
    const int64_t s = blockIdx.x;
    const int64_t t0 = indptr[s], t1 = indptr[s + 1];
    unsigned long long w = 0;
    for (int64_t win = 0; win * 1024 < n_slabs; ++win) {
        for (int q = threadIdx.x; q < 2048; q += 256) bm[q] = 0;
        __syncthreads();
        for (int64_t t = t0 + threadIdx.x; t < t1; t += 256) {
            const int32_t b0 = branch_id[t];
            if (row_ptr) {
                for (int32_t p = row_ptr[b0]; p < row_ptr[b0 + 1]; ++p) {
                    const int64_t r = row_list[p];
                    if ((r >> 16) == win) atomicOr(&bm[(r & 65535) >> 5], 1u << (r & 31));
                }
                if (win == 0) w += klen[b0];
                continue;
            }
            const int64_t r = row_of ? row_of[b0] : b0;
            if ((r >> 16) == win) {
                atomicOr(&bm[(r & 65535) >> 5], 1u << (r & 31));
                w += klen[b0];
            }
        }
        __syncthreads();
        for (int64_t q = threadIdx.x; q < 1024 && win * 1024 + q < n_slabs; q += 256)
            Pbits[(((win * 1024 + q) >> 1) * n8 + s) * 2 + (q & 1)] =
                (unsigned long long)bm[2 * q] | ((unsigned long long)bm[2 * q + 1] << 32);
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) w += __shfl_down(w, off, 64);
    if ((threadIdx.x & 63) == 0 && w) atomicAdd(&W[s], w);
}

// Stage the presence bits of pair_exact_unw_kernel (ff_kernels_exact_unw.hpp): one workgroup per sample ORs bit (row & 31) into word [row / 32][s].
__global__ void stage_xbits_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ branch_id,
                                   const int32_t *__restrict__ row_of,  // branch id -> staged row (null: identity)
                                   uint32_t *__restrict__ Xb, int64_t ldx)
{
    const int64_t s = blockIdx.x;
    for (int64_t t = indptr[s] + threadIdx.x; t < indptr[s + 1]; t += blockDim.x) {
        const int32_t b = branch_id[t], row = row_of ? row_of[b] : b;
        atomicOr(&Xb[(int64_t)(row / XU_SLAB) * ldx + s], 1u << (row % XU_SLAB));
    }
}


// ---- rare rows of a sparse table: lists by sample block for pair_low_kernel (ff_schedule.hpp LOW_*) ----------------
// Layout: BLOCK-major.  ptr[block][r] .. ptr[block][r + 1] are the entries (sample, staged value) of rare row r among the
// `tile` samples of the block (rows1 = rare rows + 1 positions per block); the entries of a block are contiguous and
// ascend with the row, so a workgroup that joins two blocks streams two lists.  (Row-major -- ptr[r][block] -- was
// built first: every row of every tile cost four cache lines for a few bytes, 75 GB of fabric traffic at 1 % density.)

// Flat nodes per staged row.
__global__ void row_counts_kernel(const int32_t *__restrict__ branch_id, int64_t nnz, const int32_t *__restrict__ row_of,
                                  int32_t *__restrict__ counts)
{
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&counts[row_of ? row_of[branch_id[t]] : branch_id[t]], 1);
}

// cnt == null: Wl[s] = sum of the staged values of sample s on its rare rows, and the same added into W[s] (which
// holds the column sums of the dense part).  cnt != null: only cnt[block of s][r] += 1 for each of them.  One
// workgroup per sample.
__global__ __launch_bounds__(256)
void low_sums_counts_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ branch_id,
                            const double *__restrict__ abnd, const double *__restrict__ branch_len, int e,
                            const int32_t *__restrict__ low_of, int64_t rows1, int tile, uint32_t *__restrict__ cnt,
                            uint32_t *__restrict__ Wl, unsigned long long *__restrict__ W)
{
    __shared__ unsigned long long total;
    const int64_t s = blockIdx.x;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    unsigned long long mine = 0;
    for (int64_t t = indptr[s] + threadIdx.x; t < indptr[s + 1]; t += blockDim.x) {
        const int32_t b = branch_id[t], r = low_of[b];
        if (r < 0) continue;
        if (cnt) atomicAdd(&cnt[(s / tile) * rows1 + r], 1u);
        else mine += stage_q_weighted(branch_len[b], abnd[t], e, b);
    }
    if (mine) atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0 && !cnt) {
        Wl[s] = (uint32_t)total;  // (< 2^31 whenever the staging is accepted: the caller checks W)
        W[s] += total;
    }
}

// Segment blockIdx.x of `in` (n values at in + blockIdx.x * stride) -> its exclusive prefix sums in place, the
// segment's total into totals[blockIdx.x].  One workgroup per segment.
__global__ __launch_bounds__(1024)
void scan_segments_kernel(uint32_t *__restrict__ data, int64_t n, int64_t stride, uint32_t *__restrict__ totals)
{
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    uint32_t *seg = data + (int64_t)blockIdx.x * stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < n; b0 += 1024) {
        const int64_t k = b0 + threadIdx.x;
        const uint32_t v = k < n ? seg[k] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(inc, d, 64);
            if (lane >= d) inc += up;
        }
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        uint32_t before = carry_s;
        for (int w = 0; w < wave; ++w) before += wave_tot[w];
        if (k < n) seg[k] = before + (inc - v);
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry_s;
}

// ptr[block][k] += base[block] (k <= rows: the last position of a block is its end = the next block's begin).
__global__ void low_add_base_kernel(uint32_t *__restrict__ ptr, int64_t rows1, int n_blocks, const uint32_t *__restrict__ base)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= rows1 * n_blocks) return;
    ptr[q] += base[q / rows1];
}

// The entries: (sample's index within its block, staged value) of every rare flat node -- 8 bytes, one load -- at
// ptr[block][r] + its turn (the order within a row's block does not matter: integer sums).  cursor: zeroed, same shape
// as ptr.
__global__ __launch_bounds__(256)
void low_fill_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ branch_id,
                     const double *__restrict__ abnd, const double *__restrict__ branch_len, int e,
                     const int32_t *__restrict__ low_of, int64_t rows1, int tile, const uint32_t *__restrict__ ptr,
                     uint32_t *__restrict__ cursor, uint2 *__restrict__ entries)
{
    const int64_t s = blockIdx.x;
    for (int64_t t = indptr[s] + threadIdx.x; t < indptr[s + 1]; t += blockDim.x) {
        const int32_t b = branch_id[t], r = low_of[b];
        if (r < 0) continue;
        const int64_t cell = (s / tile) * rows1 + r;
        const uint32_t at = ptr[cell] + atomicAdd(&cursor[cell], 1u);
        entries[at] = uint2{(uint32_t)(s % tile) << LOW_ROW_SHIFT, stage_q_weighted(branch_len[b], abnd[t], e, b)};
    }
}

// bits[block][w] bit t: rare row 64 w + t has an entry in the sample block.  One wave per (word, block).
__global__ void low_bits_kernel(const uint32_t *__restrict__ ptr, int64_t n_rows, int64_t rows1, int64_t words,
                                unsigned long long *__restrict__ bits)
{
    const int64_t w = blockIdx.x, blk = blockIdx.y, r = w * 64 + threadIdx.x;
    bool any = false;
    if (r < n_rows) any = ptr[blk * rows1 + r + 1] > ptr[blk * rows1 + r];
    const unsigned long long mask = __ballot(any);
    if (threadIdx.x == 0) bits[blk * words + w] = mask;
}

// ff_kernels_mfma.hpp -- unweighted UniFrac on the int8 matrix cores: plane staging and the MFMA pair kernel.
// A fragment of ff_device.hip: included there, once, inside its anonymous namespace
// (one translation unit, so the kernels stay internal and need no relocatable device code).

// ---- Unweighted on the matrix cores -------------------------------------------------
//
// common(i,j) = sum_b k_b u_i(b) u_j(b) (unifrac.go:159) IS a contraction: with the
// presence bits P[s][b] (int8 0/1) and the integer branch lengths cut into base-128 digits
// K_d[s][b] = digit_d(k_b) * P[s][b] (int8 0..127), common = sum_d 128^d * (P . K_d^T), an
// int8 GEMM with exact int32 accumulation (v_mfma_i32_32x32x32_i8).  The distance then
// follows from U = W_i + W_j - 2*common exactly as on the v_sad_u32 path, so the results
// are identical bit for bit; only the unit that does the work changes.  Both operands are
// sample-major (a lane's 16 consecutive branches are one 16-byte load), staged through
// LDS in 128 x 64-byte slabs with a padded 80-byte row stride (conflict-free ds_read_b128).

typedef int mfma_v4i __attribute__((ext_vector_type(4)));
typedef int mfma_v16i __attribute__((ext_vector_type(16)));

// P8 / K8 planes from the flat nodes: one workgroup per sample.
__global__ void stage_mfma_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ branch_id,
                                  const uint32_t *__restrict__ klen, int n_digits,
                                  const int32_t *__restrict__ row_of, int8_t *__restrict__ P8,
                                  int8_t *__restrict__ K8, int64_t ldb, int64_t plane,
                                  unsigned long long *__restrict__ W)
{
    const int64_t s = blockIdx.x;
    unsigned long long w = 0;
    for (int64_t t = indptr[s] + threadIdx.x; t < indptr[s + 1]; t += blockDim.x) {
        const int32_t b0 = branch_id[t];
        const uint32_t k = klen[b0];
        const int64_t b = row_of ? row_of[b0] : b0;
        P8[s * ldb + b] = 1;
        for (int d = 0; d < n_digits; ++d) K8[d * plane + s * ldb + b] = (int8_t)((k >> (7 * d)) & 127u);
        w += k;
    }
    // W_s = sum of the sample's integer branch lengths (what colsum_kernel gives the SAD path)
    for (int off = 32; off > 0; off >>= 1) w += __shfl_down(w, off, 64);
    if ((threadIdx.x & 63) == 0 && w) atomicAdd(&W[s], w);
}

// Persistent: workgroup g runs items[item_ptr[g] .. item_ptr[g+1]).
//
// Operand slabs go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no
// ds_write), three slabs ahead of the multiply, into a ring of four 32-KiB stages; each
// wave issues four 1-KiB pieces per slab.  One raw s_barrier per slab: behind it every
// wave's pieces of slab S have landed (each wave first waits on its own vmcnt) and every
// wave is done reading slab S-1, whose stage the pieces of slab S+3 may now overwrite.
// LDS rows are 64 bytes, unpadded (the DMA writes linearly); a 16-byte chunk c of row r sits
// in slot c ^ ((r >> 2) & 3), applied on the global source address of the DMA and again on
// the fragment reads, which makes the 16-lane ds_read_b128 groups conflict-free.
__device__ __forceinline__ void mfma_wait_vmcnt(int pieces_in_flight_allowed)
{
    if (pieces_in_flight_allowed >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (pieces_in_flight_allowed >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ALL_PRIVATE: every item of the launch has a private partial tile (a problem smaller than one
// round); the epilogue is then the plain stores alone, which keeps that variant's code small --
// the kernel's speed on such problems turned out to depend on it (35 vs 50 us at C2).
template <bool ALL_PRIVATE>
__global__ __launch_bounds__(512, 2)
void pair_common_mfma_kernel(const int8_t *__restrict__ P8, const int8_t *__restrict__ K8, int64_t ldb,
                             int64_t plane, const MItem *__restrict__ items,
                             const int32_t *__restrict__ item_ptr, const unsigned long long *__restrict__ W,
                             uint32_t *__restrict__ num, uint32_t *__restrict__ partial, int64_t row_begin,
                             int64_t row_end, int64_t slot_begin)
{
    extern __shared__ __attribute__((aligned(16))) int8_t mfma_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int it_begin = item_ptr[blockIdx.x], it_end = item_ptr[blockIdx.x + 1];
    // fragment read offsets inside a stage (bytes), per m / n tile and k step, swizzled
    const int arow = wi * 64 + (lane & 31), brow = M_TILE_I + wj * 64 + (lane & 31);
    for (int it = it_begin; it < it_end; ++it) {
        const MItem item = items[it];
        const int nd = item.nd;
        // this wave's four DMA pieces per slab: piece q covers stage rows (4 * wave + q) * 16 .. +16;
        // lane l moves row + l / 4, slot l % 4, i.e. source chunk (l % 4) ^ ((row >> 2) & 3)
        const int8_t *src[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = (4 * wave + q) * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ ((row >> 2) & 3);
            const int8_t *base;
            if (row < M_TILE_I) base = P8 + (int64_t)(item.i0 + row) * ldb;
            else {
                const int p = (row - M_TILE_I) / M_TILE_J, jr = (row - M_TILE_I) % M_TILE_J;
                base = K8 + (int64_t)(item.d0 + (p < nd ? p : 0)) * plane + (int64_t)(item.j0 + jr) * ldb;
            }
            src[q] = base + item.k0 + chunk * 16;
        }
        mfma_v16i acc[M_ND][2][2];
#pragma unroll
        for (int d = 0; d < M_ND; ++d)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[d][m][n][r] = 0;
        const int nslab = (item.k1 - item.k0) / M_KSLAB;
        // previous item: its atomics are out of vmcnt, and every wave is done with the ring
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        auto issue = [&](int slab) {
            int8_t *dst = mfma_lds + (slab % M_STAGES) * M_STAGE + (4 * wave) * 16 * M_KSLAB;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src[q] + (int64_t)slab * M_KSLAB),
                                                 (void __attribute__((address_space(3))) *)(dst + q * 16 * M_KSLAB), 16, 0, 0);
        };
        for (int p = 0; p < 3 && p < nslab; ++p) issue(p);
        for (int sl = 0; sl < nslab; ++sl) {
            // pieces of later slabs this wave already has in flight: min(nslab - 1 - sl, 2) * 4
            const int later = nslab - 1 - sl;
            mfma_wait_vmcnt(later >= 2 ? 8 : later * 4);
            __builtin_amdgcn_s_barrier();
            if (sl + 3 < nslab) issue(sl + 3);
            const int8_t *st = mfma_lds + (sl % M_STAGES) * M_STAGE;
#pragma unroll
            for (int kt = 0; kt < M_KSLAB / 32; ++kt) {
                const int c = 2 * kt + (lane >> 5);
                mfma_v4i a[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int r = arow + m * 32;
                    a[m] = *(const mfma_v4i *)(st + r * M_KSLAB + ((c ^ ((r >> 2) & 3)) << 4));
                }
#pragma unroll
                for (int d = 0; d < M_ND; ++d) {
                    mfma_v4i b[2];
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const int r = brow + d * M_TILE_J + n * 32;
                        b[n] = *(const mfma_v4i *)(st + r * M_KSLAB + ((c ^ ((r >> 2) & 3)) << 4));
                    }
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[d][m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b[n], acc[d][m][n], 0, 0, 0);
                }
            }
        }
        // D[row][col]: row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31
        // pad > 0: this range's share goes to its private partial tile, every element, plain stores
        // (reduce_partials_kernel applies the shard and diagonal masks); pad < 0: the tile's only
        // item stores into num[]; pad == 0: atomic add into num[]
        const bool priv = ALL_PRIVATE || item.pad > 0;
        uint32_t *pt = partial + (int64_t)(priv ? item.pad - 1 : 0) * (M_TILE_I * M_TILE_J);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = wi * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int64_t i = item.i0 + lr;
                if (!priv && (i < row_begin || i >= row_end)) continue;
                const uint32_t wi_ = item.first ? (uint32_t)W[i] : 0u;
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int lc = wj * 64 + n * 32 + (lane & 31);
                    const int64_t j = item.j0 + lc;
                    if (!priv && j >= i) continue;
                    uint32_t common = (uint32_t)acc[0][m][n][r] << (7 * item.d0);
                    if (nd > 1) common += (uint32_t)acc[1][m][n][r] << (7 * (item.d0 + 1));
                    // this item's share of result = W_i + W_j - 2 * common, modulo 2^32
                    const uint32_t v = wi_ + (item.first ? (uint32_t)W[j] : 0u) - 2u * common;
                    if (priv) pt[lr * M_TILE_J + lc] = v;
                    else if constexpr (!ALL_PRIVATE) {
                        if (item.pad < 0) num[i * (i - 1) / 2 - slot_begin + j] = v;
                        else if (v) atomicAdd(&num[i * (i - 1) / 2 - slot_begin + j], v);
                    }
                }
            }
    }
}

// Sums the private partial tiles of a small problem: tile t = (tiles[2t], tiles[2t+1]) owns the
// partials tile_ptr[t] .. tile_ptr[t+1]; one thread per pair of the tile.
__global__ void reduce_partials_kernel(const uint32_t *__restrict__ partial, const int32_t *__restrict__ tiles,
                                       const int32_t *__restrict__ tile_ptr, uint32_t *__restrict__ num,
                                       int64_t row_begin, int64_t row_end, int64_t slot_begin)
{
    const int t = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // lr * M_TILE_J + lc
    const int64_t i = tiles[2 * t] + idx / M_TILE_J, j = tiles[2 * t + 1] + idx % M_TILE_J;
    if (i < row_begin || i >= row_end || j >= i) return;
    uint32_t s = 0;
    for (int p = tile_ptr[t]; p < tile_ptr[t + 1]; ++p) s += partial[(int64_t)p * (M_TILE_I * M_TILE_J) + idx];
    num[i * (i - 1) / 2 - slot_begin + j] = s;
}

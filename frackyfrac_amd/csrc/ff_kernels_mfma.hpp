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
// are identical bit for bit; only the unit that does the work changes.
//
// What travels is the INFORMATION, not the int8 operands: presence is one bit per (sample,
// branch) -- Pbits[slab][sample], a 64-bit word per 64-branch slab, sample-minor so that the words
// a wave needs are contiguous -- and a digit is a property of the BRANCH (Kd[digit][branch], one
// byte).  Every wave builds its own MFMA fragments in registers from those words: no int8 plane
// is ever stored, in HBM or in LDS.  History, all at C3 unweighted (tools/mfma_diag.py times the
// kernel with parts compiled out): int8 planes streamed by LDS-DMA 0.385 ms, bound by the stream
// (32 KiB per slab and workgroup, 13 B/clk/CU arriving, 32 needed); bit-packed words expanded
// once per workgroup into an LDS ring 0.32-0.36 ms in every variant tried (two or three stages,
// one 8-wave or two 4-wave workgroups per CU, expansion interleaved with the MFMAs by hand),
// bound by LDS itself: the 16-byte stores of the expansion cost 13-16 cycles each and, with the
// fragment reads, kept the LDS pipe busy 80-110 % of an MFMA-paced iteration, and the barrier
// per slab marched the waves in step.  Here LDS only holds the digits (128 bytes per slab,
// read-only), there is no barrier in the loop, and the waves of a CU drift apart freely, so that
// one wave's vector work runs under its SIMD partner's MFMAs.
//
// Order of the 64 branches inside a slab.  A contraction does not care in which order its index
// runs as long as both operands agree, and one order makes the expansion cheap: for a 32-bit
// presence word x, (x >> k) & 0x01010101 is a dword of four 0/1 bytes holding bits k, 8+k, 16+k
// and 24+k -- two instructions per dword, no multiply, no dependent chain.  In the MFMA's
// k-step kt (32 branches: half kt of the slab's word), the lanes of half-wave h hold 16 of them:
// dword kk, byte q = branch 32*kt + 8*q + 4*h + kk of the slab, and the digit arrays are stored in
// that order (chunk 2*kt + h, ff_device.hip stage_for_mfma).

typedef int mfma_v4i __attribute__((ext_vector_type(4)));
typedef int mfma_v16i __attribute__((ext_vector_type(16)));
typedef unsigned short mfma_u16x2 __attribute__((ext_vector_type(2)));

constexpr int M_PAD_SLABS = 4;      // slabs of zero padding behind the staged arrays: the loop's prefetch runs past an item's end
constexpr int M_TABLE_SLABS = 512;  // slabs of digits held in LDS at a time (128 bytes each: 64 KiB)

// Presence bits from the flat nodes: one workgroup per sample builds the sample's bitmap in LDS,
// 65,536 branch rows at a time, and stores it slab by slab.  Also W_s = sum of the sample's
// integer branch lengths (what colsum_kernel gives the SAD path).
__global__ __launch_bounds__(256)
void stage_mfma_bits_kernel(const int64_t *__restrict__ indptr, const int32_t *__restrict__ branch_id,
                            const uint32_t *__restrict__ klen, const int32_t *__restrict__ row_of,
                            unsigned long long *__restrict__ Pbits, int64_t n8, int64_t n_slabs,
                            unsigned long long *__restrict__ W)
{
    __shared__ uint32_t bm[2048];
    const int64_t s = blockIdx.x;
    const int64_t t0 = indptr[s], t1 = indptr[s + 1];
    unsigned long long w = 0;
    for (int64_t win = 0; win * 1024 < n_slabs; ++win) {
        for (int q = threadIdx.x; q < 2048; q += 256) bm[q] = 0;
        __syncthreads();
        for (int64_t t = t0 + threadIdx.x; t < t1; t += 256) {
            const int32_t b0 = branch_id[t];
            const int64_t r = row_of ? row_of[b0] : b0;
            if ((r >> 16) == win) {
                atomicOr(&bm[(r & 65535) >> 5], 1u << (r & 31));
                w += klen[b0];
            }
        }
        __syncthreads();
        for (int64_t q = threadIdx.x; q < 1024 && win * 1024 + q < n_slabs; q += 256)
            Pbits[(win * 1024 + q) * n8 + s] = (unsigned long long)bm[2 * q] | ((unsigned long long)bm[2 * q + 1] << 32);
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) w += __shfl_down(w, off, 64);
    if ((threadIdx.x & 63) == 0 && w) atomicAdd(&W[s], w);
}

// Persistent: workgroup g runs items[item_ptr[g] .. item_ptr[g+1]).
//
// 256 x 128 tile per workgroup, eight waves (4 x 2) of 64 x 64 = 2 x 2 MFMA tiles per digit plane.
// Per slab a lane loads the presence words of its two i-samples and its two j-samples (four 8-byte
// loads, one slab ahead) and, per k-step: reads the 2 x 16 digits of its half-wave from the LDS
// table (two ds_read_b128, two addresses per wave: broadcast), turns the i-words into the A
// fragments (2 instructions per dword), the j-words into byte masks 0x00 / 0xFF (3 per dword:
// shift, and, one packed 16-bit multiply by 0x00FF) and the digits under those masks into the B
// fragments of both planes, then issues the 8 MFMAs: about 7 vector instructions per MFMA, which
// two waves per SIMD hide under each other's MFMAs (tools/microbench/mfma_i8_rate.hip: up to 12
// per MFMA cost nothing).
//
// ALL_PRIVATE: every item of the launch has a private partial tile (a problem smaller than one
// round); the epilogue is then the plain stores alone, which keeps that variant's code small --
// the kernel's speed on such problems turned out to depend on it (35 vs 50 us at C2).
// DIAG (builds with -DFF_MFMA_DIAG only; results are then WRONG, the time is what is asked for):
// bit 1 drops the global loads inside the loop, bit 2 the expansion (vector work), bit 3 the
// digit reads, bit 4 the MFMAs.
template <bool ALL_PRIVATE, int DIAG = 0>
__global__ __launch_bounds__(M_THREADS, 2)
void pair_common_mfma_kernel(const uint2 *__restrict__ Pbits, int64_t n8,
                             const int8_t *__restrict__ Kd, int64_t ldb, const MItem *__restrict__ items,
                             const int32_t *__restrict__ item_ptr, const unsigned long long *__restrict__ W,
                             uint32_t *__restrict__ num, uint32_t *__restrict__ partial, int64_t row_begin,
                             int64_t row_end, int64_t slot_begin)
{
    extern __shared__ __attribute__((aligned(16))) int8_t mfma_lds[];  // [slab of the segment][plane][64 digits]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int it_begin = item_ptr[blockIdx.x], it_end = item_ptr[blockIdx.x + 1];
    const int half = lane >> 5, sh = 4 * half;
    for (int it = it_begin; it < it_end; ++it) {
        const MItem item = items[it];
        const int nd = item.nd;
        const int nslab = (item.k1 - item.k0) / M_KSLAB;
        // this lane's samples: one row of the 64 x 64 sub-tile on either side
        const uint2 *pa = Pbits + (int64_t)(item.k0 / M_KSLAB) * n8 + item.i0 + wi * 64 + lane;
        const uint2 *pb = Pbits + (int64_t)(item.k0 / M_KSLAB) * n8 + item.j0 + wj * 64 + lane;
        const int8_t *dig_src[2] = {Kd + (int64_t)item.d0 * ldb + item.k0,
                                    Kd + (int64_t)(item.d0 + (nd > 1 ? 1 : 0)) * ldb + item.k0};
        mfma_v16i acc[M_ND][2][2];
#pragma unroll
        for (int d = 0; d < M_ND; ++d)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[d][m][n][r] = 0;
        for (int seg = 0; seg < nslab; seg += M_TABLE_SLABS) {
            const int nseg = nslab - seg < M_TABLE_SLABS ? nslab - seg : M_TABLE_SLABS;
            // the segment's digits -> LDS: table[(slab * 2 + plane) * 64 + position]
            __syncthreads();  // (every wave is done with the previous table)
            for (int c = tid; c < nseg * 8; c += M_THREADS) {  // 16-byte pieces: 4 per (slab, plane)
                const int sp = c >> 2, piece = c & 3;
                *(mfma_v4i *)(mfma_lds + sp * 64 + piece * 16) =
                    *(const mfma_v4i *)(dig_src[sp & 1] + (int64_t)(seg + (sp >> 1)) * M_KSLAB + piece * 16);
            }
            __syncthreads();
            // One slab: the lane's word of i-sample `lane` and of j-sample `lane` of the wave's 64 + 64
            // (one 512-byte load each; v_permlane32_swap then gives every lane the words of rows
            // lane & 31 and 32 + (lane & 31), which is what the MFMA fragments hold), 16 MFMAs.
            const int8_t *tab = mfma_lds + (2 * 0 + half) * 16;
            auto slab_step = [&](const uint2 &wx, const uint2 &wy, int sl) {
                uint32_t xw[2][2], yw[2][2];  // [k-step][row block]
                if constexpr (!(DIAG & 4)) {
                    const auto x0 = __builtin_amdgcn_permlane32_swap(wx.x, wx.x, false, false);
                    const auto x1 = __builtin_amdgcn_permlane32_swap(wx.y, wx.y, false, false);
                    const auto y0 = __builtin_amdgcn_permlane32_swap(wy.x, wy.x, false, false);
                    const auto y1 = __builtin_amdgcn_permlane32_swap(wy.y, wy.y, false, false);
                    xw[0][0] = x0[0] >> sh; xw[0][1] = x0[1] >> sh; xw[1][0] = x1[0] >> sh; xw[1][1] = x1[1] >> sh;
                    yw[0][0] = y0[0] >> sh; yw[0][1] = y0[1] >> sh; yw[1][0] = y1[0] >> sh; yw[1][1] = y1[1] >> sh;
                } else {
                    xw[0][0] = wx.x; xw[0][1] = wx.y; xw[1][0] = wx.x ^ 1; xw[1][1] = wx.y ^ 1;
                    yw[0][0] = wy.x; yw[0][1] = wy.y; yw[1][0] = wy.x ^ 1; yw[1][1] = wy.y ^ 1;
                }
                mfma_v4i d0[2], d1[2];
                if constexpr (!(DIAG & 8)) {
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) {
                        d0[kt] = *(const mfma_v4i *)(tab + sl * 128 + kt * 32);
                        d1[kt] = *(const mfma_v4i *)(tab + sl * 128 + kt * 32 + 64);
                    }
                } else {
                    d0[0] = d0[1] = d1[0] = d1[1] = mfma_v4i{sl, half, lane, 3};
                }
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    mfma_v4i a[2], b0[2], b1[2];
                    if constexpr (!(DIAG & 4)) {
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int kk = 0; kk < 4; ++kk) {
                                a[q][kk] = (int)((xw[kt][q] >> kk) & 0x01010101u);
                                const uint32_t one = (yw[kt][q] >> kk) & 0x01010101u;
                                // bytes 0x00 / 0xFF: each 16-bit half (b0 + 256 b1) * 255 = 0x00FF b0 + 0xFF00 b1
                                const mfma_u16x2 m16 = __builtin_bit_cast(mfma_u16x2, one) * (unsigned short)0x00FF;
                                const uint32_t mask = __builtin_bit_cast(uint32_t, m16);
                                b0[q][kk] = (int)((uint32_t)d0[kt][kk] & mask);
                                b1[q][kk] = (int)((uint32_t)d1[kt][kk] & mask);
                            }
                    } else {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            a[q] = mfma_v4i{(int)xw[kt][q], (int)yw[kt][q], kt, q};
                            b0[q] = d0[kt];
                            b1[q] = d1[kt];
                        }
                    }
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            if constexpr (!(DIAG & 16)) {
                                acc[0][m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b0[n], acc[0][m][n], 0, 0, 0);
                                acc[1][m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[m], b1[n], acc[1][m][n], 0, 0, 0);
                            } else {
                                acc[0][m][n][0] += a[m][0] ^ b0[n][0];
                                acc[1][m][n][0] += a[m][1] ^ b1[n][1];
                            }
                        }
                }
            };
            // two slabs in flight: buffers A (even slabs) and B (odd), each refilled right after its use
            // (reads past the item's end on the last trips hit the arrays' padding and are never used)
            const uint2 *qa = pa + (int64_t)seg * n8, *qb = pb + (int64_t)seg * n8;
            uint2 ax = qa[0], ay = qb[0], bx = qa[n8], by = qb[n8];
            qa += 2 * n8;
            qb += 2 * n8;
            int sl = 0;
            for (; sl + 1 < nseg; sl += 2) {
                slab_step(ax, ay, sl);
                if constexpr (!(DIAG & 2)) {
                    ax = qa[0];
                    ay = qb[0];
                }
                slab_step(bx, by, sl + 1);
                if constexpr (!(DIAG & 2)) {
                    bx = qa[n8];
                    by = qb[n8];
                }
                qa += 2 * n8;
                qb += 2 * n8;
            }
            if (sl < nseg) slab_step(ax, ay, sl);
        }
        // D[row][col]: row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31.
        // This item's share of result = W_i + W_j - 2 * common (modulo 2^32) goes, by item.pad:
        //   > 0  to its private partial tile, every element, plain stores (reduce_partials_kernel applies
        //        the shard and diagonal masks);   < 0  plainly into num[] (the tile's only item);
        //   = 0  into num[] by atomic add.
        auto share = [&](int m, int n, int r, uint32_t wsum) {
            uint32_t common = (uint32_t)acc[0][m][n][r] << (7 * item.d0);
            if (nd > 1) common += (uint32_t)acc[1][m][n][r] << (7 * (item.d0 + 1));
            return wsum - 2u * common;
        };
        const int lc0 = wj * 64 + (lane & 31);                 // + 32 n
        const int lr0 = wi * 64 + 4 * half;                    // + 32 m + (r & 3) + 8 (r >> 2)
        uint32_t wj_[2] = {0u, 0u};
        if (item.first) {
            wj_[0] = (uint32_t)W[item.j0 + lc0];
            wj_[1] = (uint32_t)W[item.j0 + lc0 + 32];
        }
        if (ALL_PRIVATE || item.pad > 0) {
            uint32_t *pt = partial + (int64_t)(item.pad - 1) * (M_TILE_I * M_TILE_J) + lc0;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = lr0 + m * 32 + (r & 3) + 8 * (r >> 2);
                    const uint32_t wi_ = item.first ? (uint32_t)W[item.i0 + lr] : 0u;
#pragma unroll
                    for (int n = 0; n < 2; ++n) pt[lr * M_TILE_J + 32 * n] = share(m, n, r, wi_ + wj_[n]);
                }
        } else if constexpr (!ALL_PRIVATE) {
            const bool plain = item.pad < 0;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // four consecutive rows: slot base of row i + 1 = that of row i, plus i
                    const int64_t i_first = item.i0 + lr0 + m * 32 + 8 * g;
                    int64_t base = i_first * (i_first - 1) / 2 - slot_begin + item.j0 + lc0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e;
                        const int64_t i = i_first + e;
                        if (i >= row_begin && i < row_end) {
                            const uint32_t wi_ = item.first ? (uint32_t)W[i] : 0u;
#pragma unroll
                            for (int n = 0; n < 2; ++n) {
                                if (item.j0 + lc0 + 32 * n >= i) continue;
                                const uint32_t v = share(m, n, r, wi_ + wj_[n]);
                                if (plain) num[base + 32 * n] = v;
                                else if (v) atomicAdd(&num[base + 32 * n], v);
                            }
                        }
                        base += i;
                    }
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores and atomics of this item
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

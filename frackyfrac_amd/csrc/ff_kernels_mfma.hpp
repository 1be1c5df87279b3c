// ff_kernels_mfma.hpp -- unweighted UniFrac on the int8 matrix cores: plane staging and the MFMA pair kernel.
// A fragment of ff_dev_run.hip: included there, once, inside its anonymous namespace
// (every kernel lives in exactly one translation unit, so the kernels stay internal and need no relocatable device code).

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
// branch) -- a 64-bit word per 64-branch slab and sample, the words of two consecutive slabs side
// by side (Pbits[slab / 2][sample]), sample-minor so that the words a wave needs are contiguous --
// and a digit is a property of the BRANCH (Kd[digit][branch], one byte).  Every wave builds its own MFMA fragments in registers from those words: no int8 plane
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
// that order (chunk 2*kt + h, ff_dev_stage.hip stage_for_mfma).

typedef unsigned short mfma_u16x2 __attribute__((ext_vector_type(2)));
typedef int mfma_v4i __attribute__((ext_vector_type(4)));
typedef int mfma_v16i __attribute__((ext_vector_type(16)));

// The presence words are staged in PAIRS of slabs -- Pbits[slab / 2][sample] = the sample's 64-bit words of
// slabs 2p and 2p + 1 -- so that one 16-byte load per lane fetches two slabs; the kernel's loop keeps four
// pairs in flight.  An item is a whole number of M_QUAD_SLABS (ff_schedule.hpp: 4) slabs -- the branch rows are
// zero padded to that -- and starts at a multiple of it.
// Slabs of zero padding behind the staged arrays (M_PAD_SLABS, ff_schedule.hpp): the loop requests pair
// p + M_PAIRS_IN_FLIGHT when it is done with pair p, and its prologue that many pairs whatever the item's length:
// up to M_SLABS_AHEAD = 8 slabs past an item's end are read.  The loop below is written for four buffers:
static_assert(M_PAIRS_IN_FLIGHT == 4 && M_QUAD_SLABS == 4, "pair_common_mfma_kernel: four word buffers, k-steps in groups of a quad of slabs");
#ifndef FF_MFMA_DIRECT_WORDS
#define FF_MFMA_DIRECT_WORDS 1
#endif
// 1: every lane loads the words of the rows its fragments hold (rows lane & 31 of each 32-row block: six
// loads per pair of slabs, two lanes per address); 0: one row per lane (three loads) and a
// v_permlane32_swap per word and k-step to bring them there.
constexpr bool M_DIRECT_WORDS = FF_MFMA_DIRECT_WORDS != 0;
constexpr int M_TABLE_SLABS = 512;  // slabs of digits held in LDS at a time: 64 bytes per plane and slab -- 64 KiB with two
                                    // planes, 96 KiB with the graded sweep's three; a segment's last k-step reads one slab
                                    // past its table (static_assert in the kernel: inside the 128 KiB either way)

// Persistent: workgroup g runs items[item_ptr[g] .. item_ptr[g+1]).
//
// 256 x 128 tile per workgroup, FOUR waves (2 x 2), one per SIMD, each 128 x 64 = 4 x 2 MFMA tiles
// per digit plane: 256 accumulator registers per lane, which is why there is one wave per SIMD (up
// to 512 VGPRs).  The reason for the big per-wave tile is the vector ALU: a wave64 integer
// instruction occupies its SIMD's vector issue for about 4.4 cycles and an MFMA for 8 of its 32
// (tools/microbench/mfma_i8_rate.hip: beyond six vector instructions per MFMA the matrix pipe
// waits), and building the fragments costs about 1.75 instructions per A dword and 4.75 per pair of
// B dwords (both planes share the mask).  A 64 x 64 wave tile needs 7.5 per MFMA and ran at 45
// cycles per MFMA; with 4 x 2 tiles every B fragment serves four MFMAs per plane: 4.7 per MFMA.
//
// The loop runs over k-steps (two per slab).  K-step u issues the 16 MFMAs of fragment set u & 1
// while the vector ALU builds set (u + 1) & 1, a piece of four to six instructions behind each MFMA.
// The MFMAs are inline asm: volatile asm statements keep their program order, which is the only
// way to hold this interleave (the compiler's own schedule is "all vector work, then all MFMAs",
// and a pure intrinsic has no place of its own in the instruction selector's order).  The price is
// that the compiler no longer sees matrix instructions: nothing here reads an accumulator or
// rewrites a fragment register within 16 MFMAs of the instruction concerned, and the epilogue
// waits out the last MFMA with explicit s_nop.
// Inputs of a piece: the presence words of the rows the lane's fragments hold -- row lane & 31 of each of the
// wave's four 32-row i-blocks and two j-blocks, one 16-byte load per block and PAIR of slabs, four pairs in
// flight (first use 5.5 slabs after the request) -- and the half-wave's 2 x 16 digits from the LDS table (two
// k-steps ahead).
//
// ALL_PRIVATE: every item of the launch owns a partial tile (the schedule fits FF_MFMA_PRIVATE_MB); the way
// out is then 16-byte stores of the accumulators in their own order, which reduce_private_kernel reads.
// An item with a single digit plane runs its own instantiation of everything between the accumulators'
// declaration and the way out (run_item), without the second plane.
// GRADED: the rows are staged by descending length with SIGNED digits (ff_dev_stage.hip stage_for_mfma; Kd = their
// three planes): the sweep runs k-steps of three MFMAs per tile (TRI: X += A d0, X += (A << 7)(-d1), Y += A d2) up
// to slab duo_from_slab and of two (DUO: X alone) from there on -- kstep3 / kstep2 below; the accumulators meet
// again as common = X + (Y << 15).  In these k-steps an accumulator tile is written twice, four MFMAs apart, and
// the scaled A fragments are rewritten five or more MFMAs after their last use (the hardware reads an MFMA's A and
// B operands when it issues).
// DIAG (builds with -DFF_MFMA_DIAG only; results are then WRONG, the time is what is asked for):
// bit 1 drops the global loads inside the loop, bit 2 the expansion (vector work), bit 3 the
// digit reads, bit 4 the MFMAs.
#ifdef FF_MFMA_DIAG
// Diagnostic build: the first thread of every workgroup stamps the phases of its first four items
// with the 100 MHz real-time clock (tools/mfma_stamps.py): [workgroup][item][8].
__device__ unsigned long long *g_mfma_stamps = nullptr;
#define FF_STAMP(slot)                                                                                         \
    if (g_mfma_stamps && tid == 0 && it - it_begin < 4)                                                        \
    g_mfma_stamps[((int64_t)blockIdx.x * 4 + (it - it_begin)) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime()
#define FF_STAMP_CLOCK(slot)                                                                                   \
    if (g_mfma_stamps && tid == 0 && it - it_begin < 4)                                                        \
    g_mfma_stamps[((int64_t)blockIdx.x * 4 + (it - it_begin)) * 8 + (slot)] = __builtin_amdgcn_s_memtime()
#else
#define FF_STAMP(slot)
#define FF_STAMP_CLOCK(slot)
#endif
template <bool ALL_PRIVATE, int DIAG = 0, bool GRADED = false>
__global__ __launch_bounds__(M_THREADS, 1)
void pair_common_mfma_kernel(const uint4 *__restrict__ Pbits, int64_t n8,
                             const int8_t *__restrict__ Kd, int64_t ldb, const MItem *__restrict__ items,
                             const int32_t *__restrict__ item_ptr, const unsigned long long *__restrict__ W,
                             uint32_t *__restrict__ num, uint32_t *__restrict__ partial, int64_t row_begin,
                             int64_t row_end, int64_t slot_begin,
                             int duo_from_slab,     // GRADED: from this slab on no length needs the third plane
                             const FinishArgs fin)  // fin.out != null: a tile's only item writes distances, not sums
{
    constexpr bool TRI = GRADED;  // (the graded sweep's first kind of k-step: three planes)
    extern __shared__ __attribute__((aligned(16))) int8_t mfma_lds[];  // [slab of the segment][plane][64 digits]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int it_begin = item_ptr[blockIdx.x], it_end = item_ptr[blockIdx.x + 1];
    const int half = lane >> 5, sh = 4 * half;
    for (int it = it_begin; it < it_end; ++it) {
        const MItem item = items[it];
        FF_STAMP(0);
        const int nd = item.nd;
        const int nslab = (item.k1 - item.k0) / M_KSLAB;
        __builtin_assume(nslab >= M_QUAD_SLABS);  // (ff_schedule.cpp: whole quads of slabs, at least one)
        // this lane's samples: row lane & 31 of every 32-row block of the wave's 128 i-samples and 64 j-samples
        // (without M_DIRECT_WORDS: rows `lane` and 64 + `lane`, and row `lane`)
        const int wlane = M_DIRECT_WORDS ? (lane & 31) : lane;
        const uint4 *pa = Pbits + (int64_t)(item.k0 / (2 * M_KSLAB)) * n8 + item.i0 + wi * 128 + wlane;
        const uint4 *pb = Pbits + (int64_t)(item.k0 / (2 * M_KSLAB)) * n8 + item.j0 + wj * 64 + wlane;
        constexpr int NPL = TRI ? 3 : 2;  // digit planes of a slab in the LDS table
        // the table of a segment, plus the slab past it that the last k-step's read-ahead touches (read_digits(.., + 2)
        // / read_digits3(+ 1): never used), must lie inside the workgroup's LDS (the epilogue's tile sizes it)
        static_assert((M_TABLE_SLABS + 1) * NPL * M_KSLAB <= M_LDS_BYTES, "pair_common_mfma_kernel: digit table + read-ahead past LDS");
        const int8_t *dig_src[3] = {Kd + (int64_t)item.d0 * ldb + item.k0,
                                    Kd + (int64_t)(item.d0 + (nd > 1 ? 1 : 0)) * ldb + item.k0,
                                    Kd + (int64_t)(TRI ? 2 : 0) * ldb + item.k0};  // (TRI: Kd = the three signed planes, d0 = 0)
        // An item with a single digit plane (the last group of an odd number of digits) runs without the
        // second plane's MFMAs, digit reads and masks.  The whole item -- accumulators, loop, way out -- is
        // instantiated once per case: a branch inside the loop nest would join the two cases' accumulators,
        // and the compiler resolves such a join of 256 AGPR-pinned values through private memory.
        auto run_item = [&](auto two_planes) {
            mfma_v16i acc[M_ND][4][2];
#pragma unroll
            for (int d = 0; d < M_ND; ++d)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[d][m][n][r] = 0;
            for (int seg = 0; seg < nslab; seg += M_TABLE_SLABS) {
                const int nseg = nslab - seg < M_TABLE_SLABS ? nslab - seg : M_TABLE_SLABS;
                // the words of the segment's first four pairs are requested before anything else: their latency
                // passes while the digit table is copied
                // [pair & 3][block]: the words of i-rows / j-rows; component c = the 32 branches of k-step c of
                // the pair.  Direct: block m = rows 32 m + (lane & 31); else blocks 0, 1 = rows lane, 64 + lane.
                constexpr int NWA = M_DIRECT_WORDS ? 4 : 2, NWB = M_DIRECT_WORDS ? 2 : 1, WSTEP = M_DIRECT_WORDS ? 32 : 64;
                uint4 wa[4][NWA], wb[4][NWB];
                const uint4 *qa = pa + (int64_t)(seg / 2) * n8, *qb = pb + (int64_t)(seg / 2) * n8;
                auto request_words = [&](int buf) {
#pragma unroll
                    for (int m = 0; m < NWA; ++m) wa[buf][m] = qa[WSTEP * m];
#pragma unroll
                    for (int n = 0; n < NWB; ++n) wb[buf][n] = qb[WSTEP * n];
                };
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    request_words(q);
                    qa += n8;
                    qb += n8;
                }
                // the segment's digits -> LDS: table[(slab * 2 + plane) * 64 + position]
                __syncthreads();  // (every wave is done with the previous table)
                for (int c0 = tid; c0 < nseg * 4 * NPL; c0 += 4 * M_THREADS) {  // 16-byte pieces: 4 per (slab, plane);
                    mfma_v4i piece16[4];                                        // four loads in flight per thread
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int c = c0 + u * M_THREADS, sp = c >> 2, piece = c & 3;
                        if (c < nseg * 4 * NPL)
                            piece16[u] = *(const mfma_v4i *)(dig_src[sp % NPL] + (int64_t)(seg + sp / NPL) * M_KSLAB + piece * 16);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int c = c0 + u * M_THREADS;
                        if (c < nseg * 4 * NPL) *(mfma_v4i *)(mfma_lds + (c >> 2) * 64 + (c & 3) * 16) = piece16[u];
                    }
                }
                __syncthreads();
                const int8_t *tab = mfma_lds + half * 16;
                mfma_v4i fa[2][4], fb0[2][2], fb1[2][2];  // [set][row block]
                mfma_v4i dg0[2], dg1[2];                   // digits for the k-step set [s] is (being) built for
                mfma_v4i fb2[2][2], dg2, a128[2];          // TRI: the third plane, and the A fragments of two row blocks times -128
                uint32_t swx[4], swy[2];                   // the k-step being built: words of row blocks m / n, swapped
                uint32_t t[8];                             // its B masks in the making
                uint32_t shk[4];                           // shift of dword kk for this half-wave: 4 * half + kk
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) shk[kk] = (uint32_t)(sh + kk);
                auto load_words = [&](int buf) {  // the next pair not yet requested, into buffer `buf` (reads past the
                    if constexpr (!(DIAG & 2)) request_words(buf);  // item's end hit the arrays' padding and are never multiplied)
                    qa += n8;
                    qb += n8;
                };
                auto read_digits = [&](auto two_planes, int set, int kstep) {  // kstep = 2 * slab + kt, within the segment
                    if constexpr (!(DIAG & 8)) {
                        dg0[set] = *(const mfma_v4i *)(tab + kstep * 32 + (kstep >> 1) * 64);
                        if constexpr (decltype(two_planes)::value)
                            dg1[set] = *(const mfma_v4i *)(tab + kstep * 32 + (kstep >> 1) * 64 + 64);
                    } else {
                        dg0[set] = dg1[set] = mfma_v4i{kstep, half, lane, 3};
                    }
                };
                // component `c` of the words in buffer `buf` to where the fragments want them (3 swaps, or nothing)
                auto take_words = [&](int buf, int c) {
                    auto comp = [c](const uint4 &w) { return c == 0 ? w.x : c == 1 ? w.y : c == 2 ? w.z : w.w; };
                    if constexpr (M_DIRECT_WORDS) {
#pragma unroll
                        for (int m = 0; m < 4; ++m) swx[m] = comp(wa[buf][m % NWA]);
                        swy[0] = comp(wb[buf][0]);
                        swy[1] = comp(wb[buf][1 % NWB]);
                        return;
                    }
                    const uint32_t w0 = comp(wa[buf][0]), w1 = comp(wa[buf][1 % NWA]), wy = comp(wb[buf][0]);
                    if constexpr (!(DIAG & 4)) {
                        const auto s0 = __builtin_amdgcn_permlane32_swap(w0, w0, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(w1, w1, false, false);
                        const auto sy = __builtin_amdgcn_permlane32_swap(wy, wy, false, false);
                        swx[0] = s0[0]; swx[1] = s0[1]; swx[2] = s1[0]; swx[3] = s1[1];
                        swy[0] = sy[0]; swy[1] = sy[1];
                    } else {
                        swx[0] = w0; swx[1] = w1; swx[2] = w0 ^ 1; swx[3] = w1 ^ 1;
                        swy[0] = wy; swy[1] = wy ^ 1;
                    }
                };
                // bytes 0/1 -> 0x00/0xFF: each 16-bit half (b0 + 256 b1) * 255 = 0x00FF b0 + 0xFF00 b1
                auto ff_bytemask = [](uint32_t one) {
                    const mfma_u16x2 m16 = __builtin_bit_cast(mfma_u16x2, one) * (unsigned short)0x00FF;
                    return __builtin_bit_cast(uint32_t, m16);
                };
#define FF_MM(d, m, n, A, B)                                                                           \
        if constexpr ((d) == 0 || TWO) {                                                                   \
            if constexpr (!(DIAG & 16))                                                               \
                asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(acc[d][m][n]) : "v"(A), "v"(B)); \
            else acc[d][m][n][0] += A[0] ^ B[0];                                                           \
        }                                                                                                  \
        __builtin_amdgcn_sched_barrier(0)
#define FF_OP(stmt) if constexpr (!(DIAG & 4)) { stmt; }
#define FF_OP1(stmt) if constexpr (TWO && !(DIAG & 4)) { stmt; }  // (the second digit plane's)
#define FF_END_PIECE() __builtin_amdgcn_sched_barrier(0)
                // K-step `u` of the current group of eight slabs (u = 0..15; slab sl + (u >> 1), pair u >> 2, component u & 3): the 16 MFMAs of
                // set `cur`, and behind them the 72 vector operations that build set `nxt` for k-step u + 1
                // -- per B dword pair: shift, and, byte mask, two ands with the digits; per A dword: shift,
                // and -- dealt out LEVEL BY LEVEL (all shifts, then all ands, ...), so that no instruction
                // waits for the one in front of it: a wave alone on its SIMD has nobody to hide a
                // dependent chain behind.  The last slot swaps in the words of k-step u + 2.
                auto kstep = [&](auto two_planes, int u, int sl) {
                    constexpr bool TWO = decltype(two_planes)::value;
                    const int cur = u & 1, nxt = cur ^ 1;
                    const int bp2 = ((u + 2) >> 2) & 3, bkt2 = (u + 2) & 3;  // buffer and component of k-step u + 2
                    read_digits(two_planes, cur, 2 * sl + u + 2);  // set `cur` is rebuilt in the NEXT k-step, for u + 2
                    __builtin_amdgcn_sched_barrier(0);
                    FF_MM(0, 0, 0, fa[cur][0], fb0[cur][0]); FF_OP(t[0] = swy[0] >> shk[0]); FF_OP(t[1] = swy[0] >> shk[1]); FF_OP(t[2] = swy[0] >> shk[2]); FF_OP(t[3] = swy[0] >> shk[3]); FF_END_PIECE();
                    FF_MM(1, 0, 0, fa[cur][0], fb1[cur][0]); FF_OP(t[4] = swy[1] >> shk[0]); FF_OP(t[5] = swy[1] >> shk[1]); FF_OP(t[6] = swy[1] >> shk[2]); FF_OP(t[7] = swy[1] >> shk[3]); FF_OP(t[0] &= 0x01010101u); FF_END_PIECE();
                    FF_MM(0, 0, 1, fa[cur][0], fb0[cur][1]); FF_OP(t[1] &= 0x01010101u); FF_OP(t[2] &= 0x01010101u); FF_OP(t[3] &= 0x01010101u); FF_OP(t[4] &= 0x01010101u); FF_OP(t[5] &= 0x01010101u); FF_END_PIECE();
                    FF_MM(1, 0, 1, fa[cur][0], fb1[cur][1]); FF_OP(t[6] &= 0x01010101u); FF_OP(t[7] &= 0x01010101u); FF_OP(t[0] = ff_bytemask(t[0])); FF_OP(t[1] = ff_bytemask(t[1])); FF_OP(t[2] = ff_bytemask(t[2])); FF_END_PIECE();
                    FF_MM(0, 1, 0, fa[cur][1], fb0[cur][0]); FF_OP(t[3] = ff_bytemask(t[3])); FF_OP(t[4] = ff_bytemask(t[4])); FF_OP(t[5] = ff_bytemask(t[5])); FF_OP(t[6] = ff_bytemask(t[6])); FF_OP(t[7] = ff_bytemask(t[7])); FF_END_PIECE();
                    FF_MM(1, 1, 0, fa[cur][1], fb1[cur][0]); FF_OP(fb0[nxt][0][0] = (int)((uint32_t)dg0[nxt][0] & t[0])); FF_OP1(fb1[nxt][0][0] = (int)((uint32_t)dg1[nxt][0] & t[0])); FF_OP(fb0[nxt][0][1] = (int)((uint32_t)dg0[nxt][1] & t[1])); FF_OP1(fb1[nxt][0][1] = (int)((uint32_t)dg1[nxt][1] & t[1])); FF_END_PIECE();
                    FF_MM(0, 1, 1, fa[cur][1], fb0[cur][1]); FF_OP(fb0[nxt][0][2] = (int)((uint32_t)dg0[nxt][2] & t[2])); FF_OP1(fb1[nxt][0][2] = (int)((uint32_t)dg1[nxt][2] & t[2])); FF_OP(fb0[nxt][0][3] = (int)((uint32_t)dg0[nxt][3] & t[3])); FF_OP1(fb1[nxt][0][3] = (int)((uint32_t)dg1[nxt][3] & t[3])); FF_OP(fb0[nxt][1][0] = (int)((uint32_t)dg0[nxt][0] & t[4])); FF_END_PIECE();
                    FF_MM(1, 1, 1, fa[cur][1], fb1[cur][1]); FF_OP1(fb1[nxt][1][0] = (int)((uint32_t)dg1[nxt][0] & t[4])); FF_OP(fb0[nxt][1][1] = (int)((uint32_t)dg0[nxt][1] & t[5])); FF_OP1(fb1[nxt][1][1] = (int)((uint32_t)dg1[nxt][1] & t[5])); FF_OP(fb0[nxt][1][2] = (int)((uint32_t)dg0[nxt][2] & t[6])); FF_OP1(fb1[nxt][1][2] = (int)((uint32_t)dg1[nxt][2] & t[6])); FF_END_PIECE();
                    FF_MM(0, 2, 0, fa[cur][2], fb0[cur][0]); FF_OP(fb0[nxt][1][3] = (int)((uint32_t)dg0[nxt][3] & t[7])); FF_OP1(fb1[nxt][1][3] = (int)((uint32_t)dg1[nxt][3] & t[7])); FF_OP(fa[nxt][0][0] = (int)(swx[0] >> shk[0])); FF_OP(fa[nxt][0][1] = (int)(swx[0] >> shk[1])); FF_OP(fa[nxt][0][2] = (int)(swx[0] >> shk[2])); FF_END_PIECE();
                    FF_MM(1, 2, 0, fa[cur][2], fb1[cur][0]); FF_OP(fa[nxt][0][3] = (int)(swx[0] >> shk[3])); FF_OP(fa[nxt][1][0] = (int)(swx[1] >> shk[0])); FF_OP(fa[nxt][1][1] = (int)(swx[1] >> shk[1])); FF_OP(fa[nxt][1][2] = (int)(swx[1] >> shk[2])); FF_OP(fa[nxt][1][3] = (int)(swx[1] >> shk[3])); FF_END_PIECE();
                    FF_MM(0, 2, 1, fa[cur][2], fb0[cur][1]); FF_OP(fa[nxt][2][0] = (int)(swx[2] >> shk[0])); FF_OP(fa[nxt][2][1] = (int)(swx[2] >> shk[1])); FF_OP(fa[nxt][2][2] = (int)(swx[2] >> shk[2])); FF_OP(fa[nxt][2][3] = (int)(swx[2] >> shk[3])); FF_END_PIECE();
                    FF_MM(1, 2, 1, fa[cur][2], fb1[cur][1]); FF_OP(fa[nxt][3][0] = (int)(swx[3] >> shk[0])); FF_OP(fa[nxt][3][1] = (int)(swx[3] >> shk[1])); FF_OP(fa[nxt][3][2] = (int)(swx[3] >> shk[2])); FF_OP(fa[nxt][3][3] = (int)(swx[3] >> shk[3])); FF_OP(fa[nxt][0][0] &= 0x01010101); FF_END_PIECE();
                    FF_MM(0, 3, 0, fa[cur][3], fb0[cur][0]); FF_OP(fa[nxt][0][1] &= 0x01010101); FF_OP(fa[nxt][0][2] &= 0x01010101); FF_OP(fa[nxt][0][3] &= 0x01010101); FF_OP(fa[nxt][1][0] &= 0x01010101); FF_OP(fa[nxt][1][1] &= 0x01010101); FF_END_PIECE();
                    FF_MM(1, 3, 0, fa[cur][3], fb1[cur][0]); FF_OP(fa[nxt][1][2] &= 0x01010101); FF_OP(fa[nxt][1][3] &= 0x01010101); FF_OP(fa[nxt][2][0] &= 0x01010101); FF_OP(fa[nxt][2][1] &= 0x01010101); FF_OP(fa[nxt][2][2] &= 0x01010101); FF_END_PIECE();
                    FF_MM(0, 3, 1, fa[cur][3], fb0[cur][1]); FF_OP(fa[nxt][2][3] &= 0x01010101); FF_OP(fa[nxt][3][0] &= 0x01010101); FF_OP(fa[nxt][3][1] &= 0x01010101); FF_OP(fa[nxt][3][2] &= 0x01010101); FF_OP(fa[nxt][3][3] &= 0x01010101); FF_END_PIECE();
                    FF_MM(1, 3, 1, fa[cur][3], fb1[cur][1]); take_words(bp2, bkt2); FF_END_PIECE();
                };
                // TRI: three digit planes in one sweep, into the same two accumulator sets.  With signed digits
                // k = d0 + 128 d1 + 32768 d2 (d0 in [-64, 63], d1 in [-127, 128], d2 in [0, 127]; plane 1 holds -d1):
                //   X += A . (d0 & mask),  X += (A << 7) . (-d1 & mask)   -- the bytes 0x80 of A << 7 are -128 --
                //   Y += A . (d2 & mask),  common = X + (Y << 15),
                // 24 MFMAs per k-step behind 96 vector instructions (4.0 per MFMA).  The scaled A fragments of row
                // block m live in a128[m & 1] from four MFMAs before their use to their use; the digits are
                // single-buffered (read at the top of the k-step that masks them, eight MFMAs before the first
                // mask).  The two updates of an X tile are four MFMAs apart.
                auto read_digits3 = [&](int kstep) {  // kstep = 2 * slab + kt, within the segment; slab stride 192 bytes
                    const int8_t *at = tab + kstep * 32 + (kstep >> 1) * 128;
                    dg0[0] = *(const mfma_v4i *)at;
                    dg1[0] = *(const mfma_v4i *)(at + 64);
                    dg2 = *(const mfma_v4i *)(at + 128);
                };
#define FF_T_SH(i) FF_OP(t[i] = swy[(i) >> 2] >> shk[(i) & 3])
#define FF_T_AND(i) FF_OP(t[i] &= 0x01010101u)
#define FF_T_BM(i) FF_OP(t[i] = ff_bytemask(t[i]))
#define FF_B0(n, kk) FF_OP(fb0[nxt][n][kk] = (int)((uint32_t)dg0[0][kk] & t[4 * (n) + (kk)]))
#define FF_B1(n, kk) FF_OP(fb1[nxt][n][kk] = (int)((uint32_t)dg1[0][kk] & t[4 * (n) + (kk)]))
#define FF_B2(n, kk) FF_OP(fb2[nxt][n][kk] = (int)((uint32_t)dg2[kk] & t[4 * (n) + (kk)]))
#define FF_A_SH(m) FF_OP(fa[nxt][m][0] = (int)(swx[m] >> shk[0])); FF_OP(fa[nxt][m][1] = (int)(swx[m] >> shk[1])); FF_OP(fa[nxt][m][2] = (int)(swx[m] >> shk[2])); FF_OP(fa[nxt][m][3] = (int)(swx[m] >> shk[3]))
#define FF_A_AND(m) FF_OP(fa[nxt][m][0] &= 0x01010101); FF_OP(fa[nxt][m][1] &= 0x01010101); FF_OP(fa[nxt][m][2] &= 0x01010101); FF_OP(fa[nxt][m][3] &= 0x01010101)
#define FF_A128(s, m) FF_OP(a128[s][0] = (int)((uint32_t)fa[cur][m][0] << 7)); FF_OP(a128[s][1] = (int)((uint32_t)fa[cur][m][1] << 7)); FF_OP(a128[s][2] = (int)((uint32_t)fa[cur][m][2] << 7)); FF_OP(a128[s][3] = (int)((uint32_t)fa[cur][m][3] << 7))
                auto kstep3 = [&](int u, int sl) {
                    constexpr bool TWO = true;
                    const int cur = u & 1, nxt = cur ^ 1;
                    const int bp2 = ((u + 2) >> 2) & 3, bkt2 = (u + 2) & 3;  // buffer and component of k-step u + 2
                    read_digits3(2 * sl + u + 1);  // of the k-step whose fragments (set `nxt`) this one builds
                    __builtin_amdgcn_sched_barrier(0);
                    FF_MM(0, 0, 0, fa[cur][0], fb0[cur][0]); FF_A128(0, 0); FF_END_PIECE();
                    FF_MM(1, 0, 0, fa[cur][0], fb2[cur][0]); FF_T_SH(0); FF_T_SH(1); FF_T_SH(2); FF_T_SH(3); FF_END_PIECE();
                    FF_MM(0, 0, 1, fa[cur][0], fb0[cur][1]); FF_T_SH(4); FF_T_SH(5); FF_T_SH(6); FF_T_SH(7); FF_END_PIECE();
                    FF_MM(1, 0, 1, fa[cur][0], fb2[cur][1]); FF_T_AND(0); FF_T_AND(1); FF_T_AND(2); FF_T_AND(3); FF_END_PIECE();
                    FF_MM(0, 0, 0, a128[0], fb1[cur][0]); FF_A128(1, 1); FF_END_PIECE();
                    FF_MM(0, 0, 1, a128[0], fb1[cur][1]); FF_T_AND(4); FF_T_AND(5); FF_T_AND(6); FF_T_AND(7); FF_END_PIECE();
                    FF_MM(0, 1, 0, fa[cur][1], fb0[cur][0]); FF_T_BM(0); FF_T_BM(1); FF_T_BM(2); FF_T_BM(3); FF_END_PIECE();
                    FF_MM(1, 1, 0, fa[cur][1], fb2[cur][0]); FF_T_BM(4); FF_T_BM(5); FF_T_BM(6); FF_T_BM(7); FF_END_PIECE();
                    FF_MM(0, 1, 1, fa[cur][1], fb0[cur][1]); FF_B0(0, 0); FF_B1(0, 0); FF_B2(0, 0); FF_B0(0, 1); FF_END_PIECE();
                    FF_MM(1, 1, 1, fa[cur][1], fb2[cur][1]); FF_B1(0, 1); FF_B2(0, 1); FF_B0(0, 2); FF_B1(0, 2); FF_END_PIECE();
                    FF_MM(0, 1, 0, a128[1], fb1[cur][0]); FF_A128(0, 2); FF_END_PIECE();
                    FF_MM(0, 1, 1, a128[1], fb1[cur][1]); FF_B2(0, 2); FF_B0(0, 3); FF_B1(0, 3); FF_B2(0, 3); FF_END_PIECE();
                    FF_MM(0, 2, 0, fa[cur][2], fb0[cur][0]); FF_B0(1, 0); FF_B1(1, 0); FF_B2(1, 0); FF_B0(1, 1); FF_END_PIECE();
                    FF_MM(1, 2, 0, fa[cur][2], fb2[cur][0]); FF_B1(1, 1); FF_B2(1, 1); FF_B0(1, 2); FF_B1(1, 2); FF_END_PIECE();
                    FF_MM(0, 2, 1, fa[cur][2], fb0[cur][1]); FF_B2(1, 2); FF_B0(1, 3); FF_B1(1, 3); FF_B2(1, 3); FF_END_PIECE();
                    FF_MM(1, 2, 1, fa[cur][2], fb2[cur][1]); FF_A_SH(0); FF_END_PIECE();
                    FF_MM(0, 2, 0, a128[0], fb1[cur][0]); FF_A128(1, 3); FF_END_PIECE();
                    FF_MM(0, 2, 1, a128[0], fb1[cur][1]); FF_A_SH(1); FF_END_PIECE();
                    FF_MM(0, 3, 0, fa[cur][3], fb0[cur][0]); FF_A_SH(2); FF_END_PIECE();
                    FF_MM(1, 3, 0, fa[cur][3], fb2[cur][0]); FF_A_SH(3); FF_END_PIECE();
                    FF_MM(0, 3, 1, fa[cur][3], fb0[cur][1]); FF_A_AND(0); FF_END_PIECE();
                    FF_MM(1, 3, 1, fa[cur][3], fb2[cur][1]); FF_A_AND(1); FF_END_PIECE();
                    FF_MM(0, 3, 0, a128[1], fb1[cur][0]); FF_A_AND(2); FF_END_PIECE();
                    FF_MM(0, 3, 1, a128[1], fb1[cur][1]); FF_A_AND(3); take_words(bp2, bkt2); FF_END_PIECE();
                };
#define FF_A_SH1(m, kk) FF_OP(fa[nxt][m][kk] = (int)(swx[m] >> shk[kk]))
#define FF_A_AND1(m, kk) FF_OP(fa[nxt][m][kk] &= 0x01010101)
                // DUO: the same without the third plane, for the slabs whose lengths all fit d0 + 128 d1 (the rows
                // are staged in descending order of length): 16 MFMAs into X alone behind 88 vector instructions.
                // It builds fb0 / fb1 only; a DUO k-step never comes before a TRI one.
                auto kstep2 = [&](int u, int sl) {
                    constexpr bool TWO = true;
                    const int cur = u & 1, nxt = cur ^ 1;
                    const int bp2 = ((u + 2) >> 2) & 3, bkt2 = (u + 2) & 3;
                    {
                        const int8_t *at = tab + (2 * sl + u + 1) * 32 + ((2 * sl + u + 1) >> 1) * 128;
                        dg0[0] = *(const mfma_v4i *)at;
                        dg1[0] = *(const mfma_v4i *)(at + 64);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    FF_MM(0, 0, 0, fa[cur][0], fb0[cur][0]); FF_A128(0, 0); FF_T_SH(0); FF_T_SH(1); FF_END_PIECE();
                    FF_MM(0, 0, 1, fa[cur][0], fb0[cur][1]); FF_A128(1, 1); FF_T_SH(2); FF_T_SH(3); FF_END_PIECE();
                    FF_MM(0, 1, 0, fa[cur][1], fb0[cur][0]); FF_T_SH(4); FF_T_SH(5); FF_T_SH(6); FF_T_SH(7); FF_T_AND(0); FF_T_AND(1); FF_END_PIECE();
                    FF_MM(0, 1, 1, fa[cur][1], fb0[cur][1]); FF_T_AND(2); FF_T_AND(3); FF_T_AND(4); FF_T_AND(5); FF_T_AND(6); FF_T_AND(7); FF_END_PIECE();
                    FF_MM(0, 0, 0, a128[0], fb1[cur][0]); FF_T_BM(0); FF_T_BM(1); FF_T_BM(2); FF_T_BM(3); FF_T_BM(4); FF_T_BM(5); FF_END_PIECE();
                    FF_MM(0, 0, 1, a128[0], fb1[cur][1]); FF_T_BM(6); FF_T_BM(7); FF_B0(0, 0); FF_B1(0, 0); FF_B0(0, 1); FF_B1(0, 1); FF_END_PIECE();
                    FF_MM(0, 1, 0, a128[1], fb1[cur][0]); FF_B0(0, 2); FF_B1(0, 2); FF_B0(0, 3); FF_B1(0, 3); FF_B0(1, 0); FF_B1(1, 0); FF_END_PIECE();
                    FF_MM(0, 1, 1, a128[1], fb1[cur][1]); FF_B0(1, 1); FF_B1(1, 1); FF_B0(1, 2); FF_B1(1, 2); FF_B0(1, 3); FF_B1(1, 3); FF_END_PIECE();
                    FF_MM(0, 2, 0, fa[cur][2], fb0[cur][0]); FF_A128(0, 2); FF_A_SH1(0, 0); FF_END_PIECE();
                    FF_MM(0, 2, 1, fa[cur][2], fb0[cur][1]); FF_A128(1, 3); FF_A_SH1(0, 1); FF_END_PIECE();
                    FF_MM(0, 3, 0, fa[cur][3], fb0[cur][0]); FF_A_SH1(0, 2); FF_A_SH1(0, 3); FF_A_SH1(1, 0); FF_A_SH1(1, 1); FF_A_SH1(1, 2); FF_END_PIECE();
                    FF_MM(0, 3, 1, fa[cur][3], fb0[cur][1]); FF_A_SH1(1, 3); FF_A_SH1(2, 0); FF_A_SH1(2, 1); FF_A_SH1(2, 2); FF_A_SH1(2, 3); FF_END_PIECE();
                    FF_MM(0, 2, 0, a128[0], fb1[cur][0]); FF_A_SH1(3, 0); FF_A_SH1(3, 1); FF_A_SH1(3, 2); FF_A_SH1(3, 3); FF_A_AND1(0, 0); FF_END_PIECE();
                    FF_MM(0, 2, 1, a128[0], fb1[cur][1]); FF_A_AND1(0, 1); FF_A_AND1(0, 2); FF_A_AND1(0, 3); FF_A_AND1(1, 0); FF_A_AND1(1, 1); FF_END_PIECE();
                    FF_MM(0, 3, 0, a128[1], fb1[cur][0]); FF_A_AND1(1, 2); FF_A_AND1(1, 3); FF_A_AND1(2, 0); FF_A_AND1(2, 1); FF_A_AND1(2, 2); FF_END_PIECE();
                    FF_MM(0, 3, 1, a128[1], fb1[cur][1]); FF_A_AND1(2, 3); FF_A_AND1(3, 0); FF_A_AND1(3, 1); FF_A_AND1(3, 2); FF_A_AND1(3, 3); take_words(bp2, bkt2); FF_END_PIECE();
                };
#undef FF_A_SH1
#undef FF_A_AND1
#undef FF_T_SH
#undef FF_T_AND
#undef FF_T_BM
#undef FF_B0
#undef FF_B1
#undef FF_B2
#undef FF_A_SH
#undef FF_A_AND
#undef FF_A128
                // prologue: digits of k-steps 0 and 1, fragment set 0 for k-step 0, the words of k-step 1
                if constexpr (TRI) {
                    read_digits3(0);
                } else {
                    read_digits(std::true_type{}, 0, 0);
                    read_digits(std::true_type{}, 1, 1);
                }
                take_words(0, 0);
                if constexpr (TRI) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            const uint32_t mask = ff_bytemask((swy[n] >> shk[kk]) & 0x01010101u);
                            fb0[0][n][kk] = (int)((uint32_t)dg0[0][kk] & mask);
                            fb1[0][n][kk] = (int)((uint32_t)dg1[0][kk] & mask);
                            fb2[0][n][kk] = (int)((uint32_t)dg2[kk] & mask);
                        }
#pragma unroll
                        for (int m = 0; m < 4; ++m) fa[0][m][kk] = (int)((swx[m] >> shk[kk]) & 0x01010101u);
                    }
                } else if constexpr (!(DIAG & 4)) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            const uint32_t mask = ff_bytemask((swy[n] >> shk[kk]) & 0x01010101u);
                            fb0[0][n][kk] = (int)((uint32_t)dg0[0][kk] & mask);
                            fb1[0][n][kk] = (int)((uint32_t)dg1[0][kk] & mask);
                        }
#pragma unroll
                        for (int m = 0; m < 4; ++m) fa[0][m][kk] = (int)((swx[m] >> shk[kk]) & 0x01010101u);
                    }
                } else {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                        for (int n = 0; n < 2; ++n) { fb0[0][n][kk] = dg0[0][kk]; fb1[0][n][kk] = dg1[0][kk]; }
#pragma unroll
                        for (int m = 0; m < 4; ++m) fa[0][m][kk] = (int)swx[m];
                    }
                }
                take_words(0, 1);
                __builtin_amdgcn_sched_barrier(0);
                if (seg == 0) { FF_STAMP(1); FF_STAMP_CLOCK(5); }
                // Buffer b holds the words of the pair whose number is b mod 4.  K-step u takes the words of k-step
                // u + 2, so pair p's last component is gone after its second k-step; when its fourth is done the
                // buffer takes pair p + 4, whose first component is wanted eleven k-steps -- five and a half slabs,
                // about 7,000 cycles -- later.  With one wave per SIMD a late load stalls the matrix pipe outright
                // (single slabs requested two ahead left 14 % of the wave's life in s_waitcnt, four ahead 9 %),
                // and a 16-byte load costs the wave's instruction stream what an 8-byte one does.
                auto sweep = [&](auto two_planes) {
                    int sl = 0;
                    if constexpr (GRADED) {
                        // groups of eight slabs from the segment's start: three planes while a group begins before
                        // duo_from_slab (what it multiplies beyond is zero digits), two from there on
                        const int tri_until = duo_from_slab - (item.k0 / M_KSLAB + seg);
                        for (; sl + 7 < nseg && sl < tri_until; sl += 8) {
#pragma unroll
                            for (int p = 0; p < 4; ++p) {
                                kstep3(4 * p, sl);
                                kstep3(4 * p + 1, sl);
                                kstep3(4 * p + 2, sl);
                                kstep3(4 * p + 3, sl);
                                load_words(p);
                            }
                        }
                        if (sl + 7 < nseg) {
                            // two planes from here on: their k-steps do not build the third plane's fragments, which
                            // are all zero in this part of the sweep -- and wanted as such by the last quad below
#pragma unroll
                            for (int n = 0; n < 2; ++n) fb2[0][n] = fb2[1][n] = mfma_v4i{0, 0, 0, 0};
                        }
                        for (; sl + 7 < nseg; sl += 8) {
#pragma unroll
                            for (int p = 0; p < 4; ++p) {
                                kstep2(4 * p, sl);
                                kstep2(4 * p + 1, sl);
                                kstep2(4 * p + 2, sl);
                                kstep2(4 * p + 3, sl);
                                load_words(p);
                            }
                        }
                        // a last quad of slabs, always with three planes (a third digit that is zero adds nothing).  An
                        // if / else here -- two bodies that both write the accumulators and meet again -- makes the
                        // register allocator give the 256 values new homes and copy them through private memory.
                        if (sl < nseg) {
#pragma unroll
                            for (int u = 0; u < 8; ++u) kstep3(u, sl);
                        }
                        return;
                    }
                    for (; sl + 7 < nseg; sl += 8) {
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            kstep(two_planes, 4 * p, sl);
                            kstep(two_planes, 4 * p + 1, sl);
                            kstep(two_planes, 4 * p + 2, sl);
                            kstep(two_planes, 4 * p + 3, sl);
                            load_words(p);
                        }
                    }
                    // a last quad of slabs (what its k-steps build past the end is never used)
                    if (sl < nseg) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) kstep(two_planes, u, sl);
                    }
                };
                sweep(two_planes);
#undef FF_MM
#undef FF_OP
#undef FF_OP1
#undef FF_END_PIECE
            }
            // the last MFMAs (inline asm: the compiler inserts no wait) must have written the accumulators
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
            FF_STAMP(2);
            FF_STAMP_CLOCK(6);
            // Epilogue.  The 256 accumulator tiles of a lane sit in registers that only static code can name;
            // written out element by element with the shard / diagonal tests around each store that was
            // 9,000 instructions per item (instruction-cache misses made it cost more than the whole
            // loop).  Instead: common = sum of the planes goes to LDS as a plain 256 x 128 tile (the digit
            // table is dead by now), and a short rolled loop, one row per wave and trip, applies the
            // tests and writes whole 512-byte rows.
            // D[row][col] of an MFMA tile: row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31.
            if constexpr (ALL_PRIVATE) {
                // Every item owns a partial tile, and nobody but reduce_private_kernel reads it: it is written in
                // the order the accumulators have in the registers -- [wave][m][n][r >> 2][lane][r & 3], 16-byte
                // stores, a contiguous KiB per instruction -- without a detour through LDS, and W_i + W_j is the
                // reduce kernel's business.
                uint32_t *pt = partial + (int64_t)(item.pad - 1) * (M_TILE_I * M_TILE_J) + wave * (M_TILE_I * M_TILE_J / 4) + lane * 4;
                const int s0 = 7 * item.d0, s1 = TRI ? 15 : 7 * (item.d0 + 1);
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            uint32_t v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                uint32_t lo, hi;
                                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(lo) : "a"(acc[0][m][n][4 * g + e]));
                                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(hi) : "a"(acc[1][m][n][4 * g + e]));
                                v[e] = lo << s0;
                                if (nd > 1) v[e] += hi << s1;
                            }
                            *(uint4 *)(pt + ((m * 2 + n) * 4 + g) * 256) = uint4{v[0], v[1], v[2], v[3]};
                        }
                FF_STAMP(3);
            } else {
                __syncthreads();  // every wave is done with the digit table
                {
                    uint32_t *tile = (uint32_t *)mfma_lds + (wi * 128 + 4 * half) * M_TILE_J + wj * 64 + (lane & 31);
                    const int s0 = 7 * item.d0, s1 = TRI ? 15 : 7 * (item.d0 + 1);
        #pragma unroll
                    for (int m = 0; m < 4; ++m)
        #pragma unroll
                        for (int n = 0; n < 2; ++n)
        #pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                // (explicit reads, next to their stores: left to the compiler, all 256 accumulators
                                // are copied out of the AGPRs at the loop's exit, which spills -- and a kernel with
                                // private memory pays for it at dispatch, see tools/microbench/launch_cost.hip)
                                uint32_t lo, hi;
                                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(lo) : "a"(acc[0][m][n][r]));
                                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(hi) : "a"(acc[1][m][n][r]));
                                uint32_t common = lo << s0;
                                if (nd > 1) common += hi << s1;
                                tile[(m * 32 + (r & 3) + 8 * (r >> 2)) * M_TILE_J + 32 * n] = common;
                            }
                }
                __syncthreads();
                FF_STAMP(3);
                {
                    // This item's share of result = W_i + W_j - 2 * common (modulo 2^32) goes, by item.pad:
                    //   > 0  to its private partial tile, every element, plain stores (reduce_partials_kernel
                    //        applies the shard and diagonal masks);   < 0  plainly into num[] (the tile's only
                    //        item);   = 0  into num[] by atomic add.
                    constexpr int NW = M_THREADS / 64;
                    const bool priv = item.pad > 0;
                    // W_i of the 64 rows this wave writes (rows wave, wave + 4, ...): lane t holds trip t's, so that
                    // no trip waits for a load of its own
                    const uint32_t wrows = item.first ? (uint32_t)W[item.i0 + wave + NW * lane] : 0u;
                    if (priv) {  // a lane takes columns 2 * lane, 2 * lane + 1: aligned 8-byte stores
                        const uint2 *tile = (const uint2 *)mfma_lds;
                        const int64_t j = item.j0 + 2 * lane;
                        uint32_t wj0 = 0u, wj1 = 0u;
                        if (item.first) {
                            wj0 = (uint32_t)W[j];
                            wj1 = (uint32_t)W[j + 1];
                        }
                        uint32_t *pt = partial + (int64_t)(item.pad - 1) * (M_TILE_I * M_TILE_J) + 2 * lane;
        #pragma unroll 4
                        for (int trip = 0; trip < M_TILE_I / NW; ++trip) {
                            const int row = wave + NW * trip;
                            const uint2 c = tile[row * (M_TILE_J / 2) + lane];
                            const uint32_t wi_ = (uint32_t)__builtin_amdgcn_readlane((int)wrows, trip);
                            *(uint2 *)(pt + row * M_TILE_J) = uint2{wi_ + wj0 - 2u * c.x, wi_ + wj1 - 2u * c.y};
                        }
                    } else {
                        // a lane takes columns lane and 64 + lane: a row of num[] starts at slot i (i - 1) / 2, aligned
                        // to nothing, so the stores are 4 bytes each -- and a wave's 64 of them are contiguous
                        const uint32_t *tile = (const uint32_t *)mfma_lds;
                        const int64_t j = item.j0 + lane;
                        uint32_t wj0 = 0u, wj1 = 0u;
                        if (item.first) {
                            wj0 = (uint32_t)W[j];
                            wj1 = (uint32_t)W[j + 64];
                        }
                        float h2min = INFINITY;
        #pragma unroll 4
                        for (int trip = 0; trip < M_TILE_I / NW; ++trip) {
                            const int row = wave + NW * trip;
                            const int64_t i = item.i0 + row;  // (wave-uniform, like everything derived from it)
                            const uint32_t c0 = tile[row * M_TILE_J + lane], c1 = tile[row * M_TILE_J + 64 + lane];
                            const uint32_t wi_ = (uint32_t)__builtin_amdgcn_readlane((int)wrows, trip);
                            const uint32_t v0 = wi_ + wj0 - 2u * c0, v1 = wi_ + wj1 - 2u * c1;
                            if (i < row_begin || i >= row_end) continue;
                            const int64_t t0 = i * (i - 1) / 2 - slot_begin + j;
                            if (item.pad < 0 && fin.out) {  // the tile's only item: finish in place
                                if (j < i) finish_pair(fin, t0, i, j, v0, h2min);
                                if (j + 64 < i) finish_pair(fin, t0 + 64, i, j + 64, v1, h2min);
                                continue;
                            }
                            uint32_t *dst = num + t0;
                            if (item.pad < 0) {
                                if (j < i) dst[0] = v0;
                                if (j + 64 < i) dst[64] = v1;
                            } else {
                                if (j < i && v0) atomicAdd(dst, v0);
                                if (j + 64 < i && v1) atomicAdd(dst + 64, v1);
                            }
                        }
                        finish_note_headroom(fin, h2min);
                    }
                }
            }
        };
        if constexpr (TRI) run_item(std::true_type{});  // (its schedule has one digit group per tile, nd = 2)
        else if (nd > 1) run_item(std::true_type{});
        else run_item(std::false_type{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores and atomics of this item
        FF_STAMP(4);
    }
}

#undef FF_STAMP
#undef FF_STAMP_CLOCK

// Sums the private partial tiles of a small problem: tile t = (tiles[2t], tiles[2t+1]) owns the
// partials tile_ptr[t] .. tile_ptr[t+1]; one thread per four consecutive pairs of a row (16-byte
// loads, several partials in flight per thread).
constexpr int M_REDUCE_THREADS = 256;
constexpr int M_REDUCE_BLOCKS = M_TILE_I * M_TILE_J / 4 / M_REDUCE_THREADS;
__global__ __launch_bounds__(M_REDUCE_THREADS)
void reduce_partials_kernel(const uint32_t *__restrict__ partial, const int32_t *__restrict__ tiles,
                            const int32_t *__restrict__ tile_ptr, uint32_t *__restrict__ num,
                            int64_t row_begin, int64_t row_end, int64_t slot_begin,
                            const FinishArgs fin)  // fin.out != null: distances, not sums
{
    const int t = blockIdx.y;
    const int idx = (blockIdx.x * M_REDUCE_THREADS + threadIdx.x) * 4;  // lr * M_TILE_J + lc, lc a multiple of 4
    const int64_t i = tiles[2 * t] + idx / M_TILE_J, j = tiles[2 * t + 1] + idx % M_TILE_J;
    if (i < row_begin || i >= row_end || j >= i) return;
    uint4 s = {0u, 0u, 0u, 0u};
    const int p0 = tile_ptr[t], p1 = tile_ptr[t + 1];
#pragma unroll 4
    for (int p = p0; p < p1; ++p) {
        const uint4 v = *(const uint4 *)(partial + (int64_t)p * (M_TILE_I * M_TILE_J) + idx);
        s.x += v.x;
        s.y += v.y;
        s.z += v.z;
        s.w += v.w;
    }
    const int64_t slot = i * (i - 1) / 2 - slot_begin + j;
    const uint32_t sums[4] = {s.x, s.y, s.z, s.w};
    float h2min = INFINITY;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (j + e < i) {
            if (fin.out) finish_pair(fin, slot + e, i, j + e, sums[e], h2min);
            else num[slot + e] = sums[e];
        }
    finish_note_headroom(fin, h2min);
}

// The same for a schedule whose every item has a private tile in the accumulators' own order
// (pair_common_mfma_kernel<true>): thread q of a tile takes the 16 bytes at 4 q of each of its partials
// -- rows row0 .. row0 + 3 of one column -- and adds W_i + W_j, which those items leave out.
__global__ __launch_bounds__(M_REDUCE_THREADS)
void reduce_private_kernel(const uint32_t *__restrict__ partial, const int32_t *__restrict__ tiles,
                           const int32_t *__restrict__ tile_ptr, const unsigned long long *__restrict__ W,
                           uint32_t *__restrict__ num, int64_t row_begin, int64_t row_end, int64_t slot_begin,
                           const FinishArgs fin)  // fin.out != null: distances, not sums
{
    const int t = blockIdx.y;
    const int q = blockIdx.x * M_REDUCE_THREADS + threadIdx.x;
    const int lane = q & 63, g = (q >> 6) & 3, n = (q >> 8) & 1, m = (q >> 9) & 3, wave = q >> 11;
    const int row0 = (wave >> 1) * 128 + m * 32 + 8 * g + 4 * (lane >> 5), col = (wave & 1) * 64 + n * 32 + (lane & 31);
    const int64_t i0 = tiles[2 * t] + row0, j = tiles[2 * t + 1] + col;
    if (i0 + 3 < row_begin || i0 >= row_end || j >= i0 + 3) return;
    uint4 s = {0u, 0u, 0u, 0u};
    const int p0 = tile_ptr[t], p1 = tile_ptr[t + 1];
#pragma unroll 4
    for (int p = p0; p < p1; ++p) {
        const uint4 v = *(const uint4 *)(partial + (int64_t)p * (M_TILE_I * M_TILE_J) + 4 * q);
        s.x += v.x;
        s.y += v.y;
        s.z += v.z;
        s.w += v.w;
    }
    const uint32_t sums[4] = {s.x, s.y, s.z, s.w};
    const uint32_t wj = (uint32_t)W[j];
    float h2min = INFINITY;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t i = i0 + e;
        if (i < row_begin || i >= row_end || j >= i) continue;
        const uint32_t u = (uint32_t)W[i] + wj - 2u * sums[e];
        const int64_t slot = i * (i - 1) / 2 - slot_begin + j;
        if (fin.out) finish_pair(fin, slot, i, j, u, h2min);
        else num[slot] = u;
    }
    finish_note_headroom(fin, h2min);
}


// Raw v_mfma_i32_32x32x32_i8 rate: 8 accumulator tiles per wave (the product kernel's shape), one
// or two waves per SIMD, operands in registers; N plain 32-bit integer VALU instructions behind each
// MFMA (inline asm, so that none is folded away).  Prints cycles per MFMA per SIMD and the clock held:
// how much vector work a wave can carry per MFMA before the matrix pipe starts to wait.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int WAVES, int NVALU>
__global__ __launch_bounds__(WAVES * 64) void rate(int* out, int iters, int seed, unsigned long long* clk)
{
    v16i acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    v4i a[2], b[4];
    for (int q = 0; q < 2; ++q) a[q] = v4i{(int)threadIdx.x * 0x01010101 & 0x01010101, seed & 0x01000100, 0x00010001, q};
    for (int q = 0; q < 4; ++q) b[q] = v4i{(int)(threadIdx.x * 2654435761u) & 0x7f7f7f7f, seed * 77 & 0x7f7f7f7f, 0x01020304, q};
    uint32_t x = threadIdx.x, y = seed;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[t & 1], b[t >> 1], acc[t], 0, 0, 0);
            asm volatile("" : "+v"(acc[t]));  // pins the MFMA here
#pragma unroll
            for (int v = 0; v < NVALU; ++v) {  // 2 * NVALU vector instructions the compiler cannot fold
                asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(x) : "v"(y));
                asm volatile("v_and_b32 %0, 0x01010101, %1" : "=v"(y) : "v"(x));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        a[0][3] += (int)(y & 1);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    int t = (int)x + (int)y;
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
    if (threadIdx.x == 0 && blockIdx.x == 7) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// One wave per SIMD, 16 accumulator tiles in AGPRs, the MFMA as inline asm (what the product kernel does:
// volatile asm keeps its place, and nothing makes the wave wait for an MFMA's result), NV integer
// vector instructions behind each; KIND 0: shift/and pairs, 1: with a v_pk_mul_lo_u16 every third.
template <int NV, int KIND>
__global__ __launch_bounds__(256, 1) void rate_asm(int* out, int iters, int seed, unsigned long long* clk)
{
    v16i acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    v4i a[4], b[4];
    for (int q = 0; q < 4; ++q) {
        a[q] = v4i{(int)threadIdx.x * 0x01010101 & 0x01010101, seed & 0x01000100, 0x00010001, q};
        b[q] = v4i{(int)(threadIdx.x * 2654435761u) & 0x7f7f7f7f, seed * 77 & 0x7f7f7f7f, 0x01020304, q};
    }
    uint32_t x[4] = {threadIdx.x, threadIdx.x * 3u, (uint32_t)seed, 77u}, y[4] = {1u, 2u, 3u, 4u};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a[t & 3]), "v"(b[t >> 2]));
#pragma unroll
            for (int v = 0; v < NV; ++v) {  // four independent chains, so that no filler waits for its neighbour
                const int c = v & 3;
                if (KIND == 1 && v % 3 == 2) asm volatile("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(y[c]) : "v"(x[c]), "v"(y[(c + 1) & 3]));
                else if (KIND == 2) asm volatile("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(y[c]) : "v"(x[c]), "v"(y[(c + 1) & 3]));
                else if (KIND == 3) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(y[c]) : "v"(x[c]), "v"(y[(c + 1) & 3]), "v"(x[(c + 2) & 3]));
                else if (KIND == 4) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(y[c]), "+v"(x[c]));
                else if (KIND == 5) asm volatile("v_alignbit_b32 %0, %1, %1, %2" : "=v"(y[c]) : "v"(x[c]), "v"(x[(c + 1) & 3]));
                else if (KIND == 6) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(y[c]) : "v"(x[c]), "v"(y[(c + 1) & 3]), "v"(x[(c + 2) & 3]));
                else if (KIND == 7) asm volatile("v_and_b32 %0, %1, %2" : "=v"(y[c]) : "v"(x[c]), "v"(y[(c + 1) & 3]));
                else if (KIND == 8) asm volatile("v_lshrrev_b32 %0, %1, %2" : "=v"(y[c]) : "v"(x[c]), "v"(y[(c + 1) & 3]));
                else if (KIND == 9) asm volatile("v_and_b32 %0, 0x01010101, %1" : "=v"(y[c]) : "v"(x[c]));
                else if (KIND == 10) asm volatile("v_and_b32 %0, %1, %2" : "=v"(y[c]) : "s"(seed), "v"(x[c]));
                else if (KIND == 11) asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(y[c]) : "v"(x[c]));
                else if (KIND == 12) asm volatile("v_lshlrev_b32 %0, %1, %2" : "=v"(y[c]) : "v"(x[c]), "v"(y[(c + 1) & 3]));
                else if (KIND == 13) asm volatile("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(y[c]) : "v"(x[c]), "v"(y[(c + 1) & 3]));
                else if (KIND == 14) asm volatile("v_bfe_u32 %0, %1, %2, 1" : "=v"(y[c]) : "v"(x[c]), "v"(y[(c + 1) & 3]));
                else if (KIND == 15) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(y[c]) : "v"(x[c]), "v"(y[(c + 1) & 3]), "v"(x[(c + 2) & 3]));
                else if (v & 4) asm volatile("v_and_b32 %0, 0x01010101, %1" : "=v"(y[c]) : "v"(x[c]));
                else asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(x[c]) : "v"(y[c]));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    int t = (int)(x[0] + x[1] + x[2] + x[3] + y[0] + y[1] + y[2] + y[3]);
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
    if (threadIdx.x == 0 && blockIdx.x == 7) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// The product loop's shape: two fragment sets; the 16 MFMAs of a k-step read set `cur` while 64 vector
// instructions (4 behind each MFMA) build set `nxt` -- every dword of it written twice (shift, then
// and), level by level.  EXTRA 1: + the 3 v_permlane32_swap and 2 ds_read_b128 of a k-step.
template <int EXTRA>
__global__ __launch_bounds__(256, 1) void rate_sets(int* out, int iters, int seed, unsigned long long* clk)
{
    __shared__ v4i tab[512];
    tab[threadIdx.x] = v4i{seed, 1, 2, 3};
    tab[threadIdx.x + 256] = v4i{seed, 1, 2, 3};
    __syncthreads();
    v16i acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    v4i A[2][4], B[2][4];
    for (int s = 0; s < 2; ++s)
        for (int q = 0; q < 4; ++q) {
            A[s][q] = v4i{(int)threadIdx.x * 0x01010101 & 0x01010101, seed & 0x01000100, 0x00010001, q};
            B[s][q] = v4i{(int)(threadIdx.x * 2654435761u) & 0x7f7f7f7f, seed * 77 & 0x7f7f7f7f, 0x01020304, q};
        }
    uint32_t x[4] = {threadIdx.x, threadIdx.x * 3u, (uint32_t)seed, 77u}, shv = 3, c01;
    asm volatile("v_mov_b32 %0, 0x01010101" : "=v"(c01));
    v4i dg = v4i{1, 2, 3, 4}, dg2 = v4i{1, 2, 3, 4};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    auto kstep = [&](int cur) {
        const int nxt = cur ^ 1;
        if (EXTRA) {
            dg = tab[(threadIdx.x + x[3]) & 255];
            dg2 = tab[256 + ((threadIdx.x + x[3]) & 255)];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(A[cur][t >> 2]), "v"(B[cur][t & 3]));
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int o = 4 * t + v, id = o & 31, q = (id & 15) >> 2, kk = id & 3;
                if (o < 32) {
                    if (id < 16) asm volatile("v_lshrrev_b32 %0, %1, %2" : "=v"(A[nxt][q][kk]) : "v"(shv), "v"(x[kk]));
                    else asm volatile("v_lshrrev_b32 %0, %1, %2" : "=v"(B[nxt][q][kk]) : "v"(shv), "v"(x[kk]));
                } else {
                    if (id < 16) asm volatile("v_and_b32 %0, %1, %0" : "+v"(A[nxt][q][kk]) : "v"(c01));
                    else asm volatile("v_and_b32 %0, %1, %0" : "+v"(B[nxt][q][kk]) : "v"(EXTRA ? (uint32_t)(dg[kk] + dg2[kk]) : c01));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (EXTRA) {
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[0]), "+v"(x[1]));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[2]), "+v"(x[3]));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[1]), "+v"(x[2]));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int it = 0; it < iters; it += 2) {
        kstep(0);
        kstep(1);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    int t = (int)(x[0] + x[1] + x[2] + x[3]) + A[0][0][0] + B[0][0][0] + A[1][3][3] + B[1][3][3];
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
    if (threadIdx.x == 0 && blockIdx.x == 7) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int EXTRA> void run_sets(int* d, unsigned long long* dc)
{
    const int iters = 10000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        rate_sets<EXTRA><<<256, 256>>>(d, iters, 123, dc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    unsigned long long hc[2];
    CK(hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost));
    printf("1 wave/SIMD, two fragment sets, 4 VALU per MFMA writing the next set%s: %.3f ms, %.1f cycles per MFMA, clock %.2f GHz\n",
           EXTRA ? " + 3 swaps + 2 ds_read_b128 per 16" : "", ms, (double)hc[0] / ((double)iters * 16), (double)hc[0] / hc[1] * 0.1);
}

static const char* kinds[16] = {"shift/and", "every third a pk_mul", "v_pk_mul_lo_u16", "v_perm_b32", "v_permlane32_swap",
                               "v_alignbit_b32", "v_bfi_b32", "v_and_b32 reg,reg", "v_lshrrev_b32 reg,reg", "v_and_b32 literal",
                                "v_and_b32 sgpr", "v_lshrrev_b32 by 3", "v_lshlrev_b32 reg,reg", "v_pk_lshrrev_b16", "v_bfe_u32", "v_and_or_b32"};
template <int NV, int KIND> void run_asm(int* d, unsigned long long* dc)
{
    const int iters = 10000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        rate_asm<NV, KIND><<<256, 256>>>(d, iters, 123, dc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    unsigned long long hc[2];
    CK(hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost));
    const double mfma_per_simd = (double)iters * 16, ghz = (double)hc[0] / hc[1] * 0.1;
    printf("1 wave/SIMD, asm MFMA, %2d VALU per MFMA (%s): %.3f ms, %.1f cycles per MFMA (in-kernel clock), clock %.2f GHz, %.2f POP/s\n",
           NV, KIND < 16 ? kinds[KIND] : "?", ms, (double)hc[0] / mfma_per_simd, ghz, mfma_per_simd * 1024 * 65536.0 / ms / 1e12);
}

template <int WAVES, int NVALU> void run(int* d, unsigned long long* dc)
{
    const int iters = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        rate<WAVES, NVALU><<<256, WAVES * 64>>>(d, iters, 123, dc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    unsigned long long hc[2];
    CK(hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost));
    const double mfma_per_simd = (double)iters * 8 * (WAVES / 4);
    const double ghz = (double)hc[0] / hc[1] * 0.1;
    printf("%d waves/SIMD, %2d VALU per MFMA: %.3f ms, %.1f cycles per MFMA per SIMD (wall x clock), clock %.2f GHz, %.2f POP/s\n",
           WAVES / 4, 2 * NVALU, ms, ms * 1e-3 * ghz * 1e9 / mfma_per_simd, ghz, mfma_per_simd * 1024 * 65536.0 / ms / 1e12);
}

int main()
{
    int* d; CK(hipMalloc(&d, 256 * 512 * 4));
    unsigned long long* dc; CK(hipMalloc(&dc, 16));
    run<4, 0>(d, dc); run<8, 0>(d, dc);
    run<8, 1>(d, dc); run<8, 2>(d, dc); run<8, 3>(d, dc); run<8, 4>(d, dc); run<8, 5>(d, dc); run<8, 6>(d, dc);
    run_asm<0, 0>(d, dc); run_asm<2, 0>(d, dc); run_asm<4, 0>(d, dc); run_asm<5, 0>(d, dc); run_asm<6, 0>(d, dc);
    run_asm<7, 0>(d, dc); run_asm<8, 0>(d, dc); run_asm<5, 1>(d, dc); run_asm<6, 1>(d, dc);
    run_sets<0>(d, dc); run_sets<1>(d, dc);
    run_asm<4, 2>(d, dc); run_asm<4, 3>(d, dc); run_asm<4, 4>(d, dc); run_asm<4, 5>(d, dc); run_asm<4, 6>(d, dc); run_asm<4, 7>(d, dc);
    run_asm<4, 8>(d, dc); run_asm<4, 9>(d, dc); run_asm<4, 10>(d, dc); run_asm<4, 11>(d, dc); run_asm<4, 12>(d, dc); run_asm<4, 13>(d, dc);
    run_asm<4, 14>(d, dc); run_asm<4, 15>(d, dc); run_asm<6, 7>(d, dc); run_asm<6, 8>(d, dc); run_asm<6, 3>(d, dc);
    run_asm<2, 2>(d, dc); run_asm<2, 3>(d, dc); run_asm<2, 4>(d, dc); run_asm<8, 2>(d, dc); run_asm<8, 3>(d, dc); run_asm<8, 4>(d, dc);
    return 0;
}

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
           NV, KIND ? "with pk_mul" : "shift/and", ms, (double)hc[0] / mfma_per_simd, ghz, mfma_per_simd * 1024 * 65536.0 / ms / 1e12);
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
    return 0;
}

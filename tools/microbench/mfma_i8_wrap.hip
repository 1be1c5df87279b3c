// Does v_mfma_i32_32x32x32_i8 wrap or saturate when an int32 accumulator overflows?  One wave: accumulators start
// near INT32_MAX / INT32_MIN, one MFMA pushes them over; also a signed A operand (bytes 0x80 = -128) against
// signed B bytes, which is what the graded-plane sweep of pair_common_mfma_kernel multiplies.
// Prints what comes back next to the two's-complement expectation.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(64) void wrap(int *out, int a_byte, int b_byte, int start)
{
    v16i acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = start;
    const int ab = (a_byte & 255) * 0x01010101, bb = (b_byte & 255) * 0x01010101;
    const v4i a = {ab, ab, ab, ab}, b = {bb, bb, bb, bb};
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) out[threadIdx.x * 16 + r] = acc[r];
}

int main()
{
    int *d;
    CK(hipMalloc(&d, 64 * 16 * sizeof(int)));
    struct { int a, b, start; } cases[] = {
        {1, 127, 0x7FFFFF00}, {1, -128, (int)0x80000010}, {-128, -128, 0x7FFFFF00}, {-128, 127, (int)0x80000010},
        {-128, -1, 0}, {1, 1, 0}, {-128, 127, 0}};
    int bad = 0;
    for (auto c : cases) {
        wrap<<<1, 64>>>(d, c.a, c.b, c.start);
        CK(hipDeviceSynchronize());
        int h[64 * 16];
        CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
        const uint32_t want = (uint32_t)c.start + (uint32_t)(32 * c.a * c.b);
        int same = 1;
        for (int q = 0; q < 64 * 16; ++q) same &= (uint32_t)h[q] == want;
        printf("A byte %4d  B byte %4d  start %11d: got %11d  two's complement %11d  %s\n", c.a, c.b, c.start, h[0],
               (int)want, same ? "wraps (all 1024 elements)" : "DIFFERENT");
        bad += !same;
    }
    printf(bad ? "NOT two's-complement accumulation\n" : "int32 accumulation is two's complement: overflow wraps\n");
    return bad;
}

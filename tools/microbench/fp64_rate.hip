// What does the chip sustain in UNFUSED binary64 vector operations -- the six per term of pair_exact64_kernel
// (sub, |x| * l, add; add, mul, add) -- and at what clock?  256 x 256 or 512 threads, 16 independent pair
// accumulators per lane as in the kernel's 16-row tile, operands in registers, -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k(double* out, int iters, double seed, unsigned long long* clk)
{
    double a[16], c[16], x[16];
    for (int r = 0; r < 16; ++r) { a[r] = 0.0; c[r] = 0.0; x[r] = seed * (r + 1) + threadIdx.x; }
    double y = seed + threadIdx.x * 0.5, l = seed * 0.25;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            a[r] = a[r] + l * fabs(x[r] - y);
            c[r] = c[r] + l * (x[r] + y);
        }
        y += 1.0;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int r = 0; r < 16; ++r) s += a[r] + c[r];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 7) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int THREADS> void run(double* d, unsigned long long* dc, int wgs)
{
    const int iters = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        k<THREADS><<<wgs, THREADS>>>(d, iters, 1.25, dc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    unsigned long long hc[2];
    CK(hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost));
    const double ops = (double)wgs * THREADS * (double)iters * (16 * 6 + 1);
    printf("%4d workgroups x %3d threads: %.3f ms, %.1f T binary64 op/s, shader clock %.2f GHz (peak at that clock: %.1f)\n",
           wgs, THREADS, ms, ops / ms / 1e9, (double)hc[0] / hc[1] * 0.1, 1024 * 16 * (double)hc[0] / hc[1] * 0.1 / 1000.0);
}

int main()
{
    double* d; CK(hipMalloc(&d, sizeof(double) * 1024 * 512));
    unsigned long long* dc; CK(hipMalloc(&dc, 16));
    run<256>(d, dc, 256); run<512>(d, dc, 256); run<256>(d, dc, 1024); run<512>(d, dc, 1024);
    return 0;
}

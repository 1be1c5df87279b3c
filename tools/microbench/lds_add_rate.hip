// LDS atomic-add rate microbench: what pair_low_kernel's updates (one ds_add_u32 each, no return) are priced
// against in bench.py's roofline.parts (LDS_PEAK_TADDS).  16 waves per workgroup, 2 workgroups per CU, a 48-KiB
// accumulator per workgroup -- the product kernel's shape.  Modes:
//   0  ds_add_u32, conflict-free (lane l -> bank l mod 32, the two halves of the wave in turn)
//   1  ds_add_u32, addresses as the kernel's: (random row of 96) * 128 words + (random column of 96)
//   2  ds_write_b32, conflict-free (the guide's LDS table gives 64 B per CU and clock: calibration)
//   3  ds_add_u32, conflict-free, half of the lanes masked off (EXEC = the even lanes)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int UNROLL = 16;

template <int MODE>
__global__ __launch_bounds__(1024) void rate(uint32_t *out, int iters, uint32_t seed, unsigned long long *clk)
{
    __shared__ uint32_t acc[12288];
    for (int k = threadIdx.x; k < 12288; k += 1024) acc[k] = 0;
    __syncthreads();
    uint32_t addr[UNROLL];
    uint32_t h = threadIdx.x * 2654435761u + seed + blockIdx.x * 40503u;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        if (MODE == 1) {
            h ^= h << 13; h ^= h >> 17; h ^= h << 5;
            addr[u] = (((h >> 8) % 96u) * 128u + (h & 0xffu) % 96u) * 4u;
        } else {
            addr[u] = ((threadIdx.x & 255u) + 256u * (uint32_t)u) * 4u;   // (16 KiB: lane -> its own bank)
        }
    }
    const uint32_t v = seed | 1u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (MODE != 3 || (threadIdx.x & 1) == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                if (MODE == 2) asm volatile("ds_write_b32 %0, %1" ::"v"(addr[u]), "v"(v) : "memory");
                else asm volatile("ds_add_u32 %0, %1" ::"v"(addr[u]), "v"(v) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    uint32_t t = 0;
    for (int k = threadIdx.x; k < 12288; k += 1024) t += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
    if (threadIdx.x == 0 && blockIdx.x == 7) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t *d;
    CK(hipMalloc(&d, (size_t)cus * 2 * 1024 * 4));
    unsigned long long *dc, hc[2];
    CK(hipMalloc(&dc, 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char *names[] = {"ds_add_u32 conflict-free", "ds_add_u32 rows x columns of 96", "ds_write_b32 conflict-free",
                           "ds_add_u32 conflict-free, even lanes"};
    for (int mode = 0; mode < 4; ++mode)
        for (int wg = 1; wg <= 2; ++wg) {
            const int grid = cus * wg, iters = 2000;
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                switch (mode) {
                case 0: rate<0><<<grid, 1024>>>(d, iters, 123, dc); break;
                case 1: rate<1><<<grid, 1024>>>(d, iters, 123, dc); break;
                case 2: rate<2><<<grid, 1024>>>(d, iters, 123, dc); break;
                case 3: rate<3><<<grid, 1024>>>(d, iters, 123, dc); break;
                }
                CK(hipGetLastError());
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            CK(hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost));
            const double ghz = (double)hc[0] / (double)hc[1] * 0.1;
            const double lanes = (double)grid * 1024 * iters * UNROLL * (mode == 3 ? 0.5 : 1.0);
            const double insts = (double)grid * 16 * iters * UNROLL;
            printf("%-38s %d WG/CU: %7.3f ms  %6.2f T lane-ops/s  %5.2f lanes/clk/CU  %5.2f clk/wave-instruction/CU  (clock %.2f GHz)\n",
                   names[mode], wg, ms, lanes / ms / 1e9, lanes / (ms * 1e-3) / cus / (ghz * 1e9),
                   (ms * 1e-3) * ghz * 1e9 * cus / insts, ghz);
        }
    return 0;
}

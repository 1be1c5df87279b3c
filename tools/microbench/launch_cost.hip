// What does a dispatch cost before its first wave runs?  Back-to-back launches of near-empty kernels
// that differ in what the dispatcher has to set up: dynamic LDS, private (scratch) memory, a full
// register file.  Time per launch from HIP events around 200 launches.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_plain(int* out) { if (threadIdx.x == 0 && blockIdx.x == 999999) out[0] = 1; }

__global__ __launch_bounds__(512) void k_lds(int* out)
{
    extern __shared__ int lds[];
    if (threadIdx.x == 0 && blockIdx.x == 999999) out[0] = lds[5];
}

template <int WORDS>
__global__ __launch_bounds__(256) void k_scratch(int* out, int idx)
{
    extern __shared__ int lds[];
    volatile int priv[WORDS];  // dynamic indexing keeps it in private memory
    for (int k = 0; k < WORDS; ++k) priv[k] = k + threadIdx.x;
    if (threadIdx.x == 0 && blockIdx.x == 999999) out[0] = priv[idx & (WORDS - 1)] + lds[5];
}

// The register file is claimed by naming its last registers in an asm clobber list: the kernel
// descriptor then asks for 256 VGPRs + 256 AGPRs per lane, as pair_common_mfma_kernel does.
template <bool SCRATCH>
__global__ __launch_bounds__(256, 1) void k_regs(int* out, int idx)
{
    extern __shared__ int lds[];
    asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a255, 0" ::: "v255", "a255");
    if constexpr (SCRATCH) {
        volatile int priv[32];
        for (int k = 0; k < 32; ++k) priv[k] = k + threadIdx.x;
        if (threadIdx.x == 0 && blockIdx.x == 999999) out[1] = priv[idx & 31];
    }
    if (threadIdx.x == 0 && blockIdx.x == 999999) out[0] = lds[3];
}

template <typename F> void timeit(const char* name, F launch)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 20; ++w) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int w = 0; w < 200; ++w) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-58s %.2f us per launch\n", name, ms * 1000 / 200);
}

int main()
{
    int* d; CK(hipMalloc(&d, 64));
    const int L = 128 * 1024;
    CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, L));
    CK(hipFuncSetAttribute((const void*)k_scratch<32>, hipFuncAttributeMaxDynamicSharedMemorySize, L));
    CK(hipFuncSetAttribute((const void*)k_scratch<512>, hipFuncAttributeMaxDynamicSharedMemorySize, L));
    CK(hipFuncSetAttribute((const void*)k_regs<false>, hipFuncAttributeMaxDynamicSharedMemorySize, L));
    CK(hipFuncSetAttribute((const void*)k_regs<true>, hipFuncAttributeMaxDynamicSharedMemorySize, L));
    timeit("256 x 256 threads, nothing", [&] { k_plain<<<256, 256>>>(d); });
    timeit("+ 128 KiB dynamic LDS", [&] { k_lds<<<256, 256, L>>>(d); });
    timeit("+ 128 KiB LDS + 128 B/lane of scratch", [&] { k_scratch<32><<<256, 256, L>>>(d, 3); });
    timeit("+ 128 KiB LDS + 2 KiB/lane of scratch", [&] { k_scratch<512><<<256, 256, L>>>(d, 3); });
    timeit("+ 128 KiB LDS + 512 registers per lane", [&] { k_regs<false><<<256, 256, L>>>(d, 0); });
    timeit("+ 128 KiB LDS + 512 registers + 128 B/lane of scratch", [&] { k_regs<true><<<256, 256, L>>>(d, 0); });
    timeit("512 registers per lane, no LDS", [&] { k_regs<false><<<256, 256, 0>>>(d, 0); });
    timeit("512 registers + 128 B/lane of scratch, no LDS", [&] { k_regs<true><<<256, 256, 0>>>(d, 0); });
    timeit("2048 x 256 threads: 128 KiB LDS + 512 registers + scratch", [&] { k_regs<true><<<2048, 256, L>>>(d, 0); });
    timeit("2048 x 256 threads: 128 KiB LDS + 512 registers", [&] { k_regs<false><<<2048, 256, L>>>(d, 0); });
    timeit("256 x 512 threads, 96 KiB LDS", [&] { k_lds<<<256, 512, 96 * 1024>>>(d); });
    return 0;
}

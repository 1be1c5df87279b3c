// Raw VALU rate microbench: candidate instruction mixes for sum |a-b|.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint32_t sad_s(uint32_t s, uint32_t v, uint32_t acc) { uint32_t r; asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "s"(s), "v"(v), "v"(acc)); return r; }
__device__ __forceinline__ uint32_t min_s(uint32_t s, uint32_t v) { uint32_t r; asm("v_min_u32 %0, %1, %2" : "=v"(r) : "s"(s), "v"(v)); return r; }
__device__ __forceinline__ uint32_t add3(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ uint32_t xad_s(uint32_t s, uint32_t v, uint32_t acc) { uint32_t r; asm("v_xad_u32 %0, %1, %2, %3" : "=v"(r) : "s"(s), "v"(v), "v"(acc)); return r; }
__device__ __forceinline__ uint32_t sad16_s(uint32_t s, uint32_t v, uint32_t acc) { uint32_t r; asm("v_sad_u16 %0, %1, %2, %3" : "=v"(r) : "s"(s), "v"(v), "v"(acc)); return r; }
__device__ __forceinline__ uint32_t sad8_s(uint32_t s, uint32_t v, uint32_t acc) { uint32_t r; asm("v_sad_u8 %0, %1, %2, %3" : "=v"(r) : "s"(s), "v"(v), "v"(acc)); return r; }
__device__ __forceinline__ uint32_t sub_s(uint32_t s, uint32_t v) { uint32_t r; asm("v_sub_u32 %0, %1, %2" : "=v"(r) : "s"(s), "v"(v)); return r; }
__device__ __forceinline__ uint32_t max_v(uint32_t a, uint32_t b) { uint32_t r; asm("v_max_u32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <int MODE>
__global__ __launch_bounds__(512) void rate(uint32_t* out, int iters, uint32_t seed, unsigned long long* clk) {
  uint32_t acc[32];
  uint32_t v = threadIdx.x * 2654435761u + seed, v2 = v * 31u;
  uint32_t s = __builtin_amdgcn_readfirstlane(seed * 31u + blockIdx.x);
#pragma unroll
  for (int r = 0; r < 32; ++r) acc[r] = r;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      if (MODE == 0) { acc[r] = sad_s(s + r, v, acc[r]); acc[r] = sad_s(s + r + 64, v2, acc[r]); }
      else if (MODE == 1) { acc[r] = add3(acc[r], min_s(s + r, v), min_s(s + r + 64, v2)); }
      else if (MODE == 2) { acc[r] = xad_s(s + r, v, acc[r]); acc[r] = xad_s(s + r + 64, v2, acc[r]); }
      else if (MODE == 3) { acc[r] = sad16_s(s + r, v, acc[r]); acc[r] = sad16_s(s + r + 64, v2, acc[r]); }
      else if (MODE == 4) { acc[r] = sad8_s(s + r, v, acc[r]); acc[r] = sad8_s(s + r + 64, v2, acc[r]); }
      else if (MODE == 5) { uint32_t d1 = sub_s(s + r, v), d2 = sub_s(s + r + 64, v2); acc[r] = add3(acc[r], max_v(d1, 0u - d1), max_v(d2, 0u - d2)); }
    }
    v += acc[0] & 1; v2 += acc[1] & 1;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t t = 0;
#pragma unroll
  for (int r = 0; r < 32; ++r) t += acc[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
  if (threadIdx.x == 0 && blockIdx.x == 7) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
int main() {
  uint32_t* d; CK(hipMalloc(&d, 1024 * 512 * 4));
  unsigned long long* dc; CK(hipMalloc(&dc, 16)); unsigned long long hc[2];
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[] = {"2x sad_u32", "min,min,add3", "2x xad_u32", "2x sad_u16", "2x sad_u8", "sub,sub,max,max,add3(+2 neg)"};
  for (int mode = 0; mode < 6; ++mode) for (int wg = 1; wg <= 2; ++wg) {
    int grid = 256 * wg, iters = 20000;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      switch (mode) { case 0: rate<0><<<grid, 512>>>(d, iters, 123, dc); break; case 1: rate<1><<<grid, 512>>>(d, iters, 123, dc); break; case 2: rate<2><<<grid, 512>>>(d, iters, 123, dc); break;
        case 3: rate<3><<<grid, 512>>>(d, iters, 123, dc); break; case 4: rate<4><<<grid, 512>>>(d, iters, 123, dc); break; case 5: rate<5><<<grid, 512>>>(d, iters, 123, dc); break; }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost));
    double terms = (double)grid * 512 * iters * 64;
    printf("%-30s grid %4d: %.3f ms  %.2f T terms/s  clock %.2f GHz\n", names[mode], grid, ms, terms / ms / 1e9, (double)hc[0] / hc[1] * 0.1);
  }
  return 0;
}

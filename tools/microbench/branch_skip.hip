// Cost of skipping (row, sample) cells whose scalar operand is zero by a scalar bit test +
// branch around the vector ops, versus the dense kernel's unconditional v_sad_u32.
//   MODE 0: dense, IB*JV v_sad_u32 per row
//   MODE 1: per cell: s_bitcmp1 + s_cbranch around JV x (v_sad_u32 ; v_sub_u32)   [U = W_j + sum_active(|qi-qj| - qj)]
//   MODE 2: per cell: s_bitcmp1 + s_cbranch around JV x v_sad_u32 (lower bound for the branch cost)
// Tile shapes IB x (64*JV): 32x256 (JV=4) and 16x512 (JV=8); 128 accumulator VGPRs either way.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef const __attribute__((address_space(4))) uint32_t* cptr;

template <int JV, int MODE> struct Cell;
template <int MODE> struct Cell<4, MODE> {
  template <int BIT> static __device__ __forceinline__ void run(uint32_t (&a)[4], uint32_t m, uint32_t s, const uint32_t (&v)[4]) {
    if (MODE == 0)
      asm volatile("v_sad_u32 %0, %4, %5, %0\n v_sad_u32 %1, %4, %6, %1\n v_sad_u32 %2, %4, %7, %2\n v_sad_u32 %3, %4, %8, %3"
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "s"(s), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    else if (MODE == 1)
      asm volatile("s_bitcmp1_b32 %9, %10\n s_cbranch_scc0 .Lskip%=\n"
                   "v_sad_u32 %0, %4, %5, %0\n v_sad_u32 %1, %4, %6, %1\n v_sad_u32 %2, %4, %7, %2\n v_sad_u32 %3, %4, %8, %3\n"
                   "v_sub_u32 %0, %0, %5\n v_sub_u32 %1, %1, %6\n v_sub_u32 %2, %2, %7\n v_sub_u32 %3, %3, %8\n.Lskip%=:"
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "s"(s), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "s"(m), "n"(BIT) : "scc");
    else
      asm volatile("s_bitcmp1_b32 %9, %10\n s_cbranch_scc0 .Lskip%=\n"
                   "v_sad_u32 %0, %4, %5, %0\n v_sad_u32 %1, %4, %6, %1\n v_sad_u32 %2, %4, %7, %2\n v_sad_u32 %3, %4, %8, %3\n.Lskip%=:"
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "s"(s), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "s"(m), "n"(BIT) : "scc");
  }
};
template <int MODE> struct Cell<8, MODE> {
  template <int BIT> static __device__ __forceinline__ void run(uint32_t (&a)[8], uint32_t m, uint32_t s, const uint32_t (&v)[8]) {
#define SAD8 "v_sad_u32 %0, %8, %9, %0\n v_sad_u32 %1, %8, %10, %1\n v_sad_u32 %2, %8, %11, %2\n v_sad_u32 %3, %8, %12, %3\n" \
             "v_sad_u32 %4, %8, %13, %4\n v_sad_u32 %5, %8, %14, %5\n v_sad_u32 %6, %8, %15, %6\n v_sad_u32 %7, %8, %16, %7\n"
#define SUB8 "v_sub_u32 %0, %0, %9\n v_sub_u32 %1, %1, %10\n v_sub_u32 %2, %2, %11\n v_sub_u32 %3, %3, %12\n" \
             "v_sub_u32 %4, %4, %13\n v_sub_u32 %5, %5, %14\n v_sub_u32 %6, %6, %15\n v_sub_u32 %7, %7, %16\n"
#define OPS8 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
             : "s"(s), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "s"(m), "n"(BIT)
    if (MODE == 0) asm volatile(SAD8 OPS8);
    else if (MODE == 1) asm volatile("s_bitcmp1_b32 %17, %18\n s_cbranch_scc0 .Lskip%=\n" SAD8 SUB8 ".Lskip%=:" OPS8 : "scc");
    else asm volatile("s_bitcmp1_b32 %17, %18\n s_cbranch_scc0 .Lskip%=\n" SAD8 ".Lskip%=:" OPS8 : "scc");
  }
};

template <int IB, int JV, int MODE, int I = 0> struct Row {
  static __device__ __forceinline__ void run(uint32_t (&acc)[IB][JV], uint32_t m, const uint32_t (&s)[IB], const uint32_t (&v)[JV]) {
    Cell<JV, MODE>::template run<I>(acc[I], m, s[I], v);
    Row<IB, JV, MODE, I + 1>::run(acc, m, s, v);
  }
};
template <int IB, int JV, int MODE> struct Row<IB, JV, MODE, IB> {
  static __device__ __forceinline__ void run(uint32_t (&)[IB][JV], uint32_t, const uint32_t (&)[IB], const uint32_t (&)[JV]) {}
};

template <int IB, int JV, int MODE, int NT>
__global__ __launch_bounds__(NT) void skip(const uint32_t* masks_g, const uint32_t* svals_g, uint32_t* out, int R) {
  uint32_t acc[IB][JV], v[JV];
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < JV; ++j) acc[i][j] = i + j;
#pragma unroll
  for (int j = 0; j < JV; ++j) v[j] = threadIdx.x * 2654435761u + j;
  const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * NT + threadIdx.x) >> 6);
  cptr masks = (cptr)(masks_g) + (size_t)(wave & 1023) * R;
  cptr svals = (cptr)(svals_g);
  for (int r = 0; r < R; ++r) {
    uint32_t m = masks[r];
    uint32_t s[IB];
#pragma unroll
    for (int i = 0; i < IB; ++i) s[i] = svals[(size_t)r * IB + i];
    Row<IB, JV, MODE>::run(acc, m, s, v);
  }
  uint32_t t = 0;
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < JV; ++j) t += acc[i][j];
  out[blockIdx.x * NT + threadIdx.x] = t;
}

template <int IB, int JV, int MODE, int NT>
static void bench(const char* name, const uint32_t* dm, const uint32_t* ds, uint32_t* dout, int R, double density) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    skip<IB, JV, MODE, NT><<<256, NT>>>(dm, ds, dout, R);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  const double cells = 256.0 * (NT / 64) * R * IB * JV * 64;  // (pair, row) terms covered
  printf("%-28s %dx%d NT=%d density %.2f: %.3f ms  %.1f T covered terms/s\n", name, IB, JV * 64, NT, density, ms, cells / ms / 1e9);
}


// Group skip on the 32x256 tile: a group of 4 samples is skipped on a row where all four of
// its scalar operands are zero (s_or_b32 chain sets SCC); otherwise 16 v_sad_u32 + 4
// v_add_u32 into the group's column-sum accumulators z  (U = acc + W_j - z, exact).
template <int G0, bool SKIP>
__device__ __forceinline__ void grp4(uint32_t (&a)[32][4], uint32_t (&z)[8][4], const uint32_t (&s)[32], const uint32_t (&v)[4])
{
#define GBODY \
  "v_sad_u32 %0, %21, %25, %0\n v_sad_u32 %1, %21, %26, %1\n v_sad_u32 %2, %21, %27, %2\n v_sad_u32 %3, %21, %28, %3\n" \
  "v_sad_u32 %4, %22, %25, %4\n v_sad_u32 %5, %22, %26, %5\n v_sad_u32 %6, %22, %27, %6\n v_sad_u32 %7, %22, %28, %7\n" \
  "v_sad_u32 %8, %23, %25, %8\n v_sad_u32 %9, %23, %26, %9\n v_sad_u32 %10, %23, %27, %10\n v_sad_u32 %11, %23, %28, %11\n" \
  "v_sad_u32 %12, %24, %25, %12\n v_sad_u32 %13, %24, %26, %13\n v_sad_u32 %14, %24, %27, %14\n v_sad_u32 %15, %24, %28, %15\n"
#define GOPS \
  : "+v"(a[G0][0]), "+v"(a[G0][1]), "+v"(a[G0][2]), "+v"(a[G0][3]), "+v"(a[G0 + 1][0]), "+v"(a[G0 + 1][1]), "+v"(a[G0 + 1][2]), "+v"(a[G0 + 1][3]), \
    "+v"(a[G0 + 2][0]), "+v"(a[G0 + 2][1]), "+v"(a[G0 + 2][2]), "+v"(a[G0 + 2][3]), "+v"(a[G0 + 3][0]), "+v"(a[G0 + 3][1]), "+v"(a[G0 + 3][2]), "+v"(a[G0 + 3][3]), \
    "+v"(z[G0 / 4][0]), "+v"(z[G0 / 4][1]), "+v"(z[G0 / 4][2]), "+v"(z[G0 / 4][3]), "=&s"(t) \
  : "s"(s[G0]), "s"(s[G0 + 1]), "s"(s[G0 + 2]), "s"(s[G0 + 3]), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "scc"
  uint32_t t;
  if (SKIP)
    asm volatile("s_or_b32 %20, %21, %22\n s_or_b32 %20, %20, %23\n s_or_b32 %20, %20, %24\n s_cbranch_scc0 .Lskip%=\n" GBODY
                 "v_add_u32 %16, %16, %25\n v_add_u32 %17, %17, %26\n v_add_u32 %18, %18, %27\n v_add_u32 %19, %19, %28\n.Lskip%=:" GOPS);
  else
    asm volatile(GBODY GOPS);
}

template <bool SKIP, int NT>
__global__ __launch_bounds__(NT) void group_skip(const uint32_t* svals_g, uint32_t* out, int R) {
  uint32_t acc[32][4], z[8][4], v[4];
#pragma unroll
  for (int i = 0; i < 32; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = i + j;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) z[i][j] = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = threadIdx.x * 2654435761u + j;
  const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * NT + threadIdx.x) >> 6);
  cptr svals = (cptr)(svals_g) + (size_t)(wave & 127) * 32;   // [R][128 i-blocks][32]
  for (int r = 0; r < R; ++r) {
    uint32_t s[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) s[i] = svals[(size_t)r * 4096 + i];
    grp4<0, SKIP>(acc, z, s, v); grp4<4, SKIP>(acc, z, s, v); grp4<8, SKIP>(acc, z, s, v); grp4<12, SKIP>(acc, z, s, v);
    grp4<16, SKIP>(acc, z, s, v); grp4<20, SKIP>(acc, z, s, v); grp4<24, SKIP>(acc, z, s, v); grp4<28, SKIP>(acc, z, s, v);
  }
  uint32_t t = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) t += acc[i][j] - z[i / 4][j];
  out[blockIdx.x * NT + threadIdx.x] = t;
}

template <bool SKIP, int NT>
static void bench_group(const uint32_t* ds, uint32_t* dout, int R, double d0, double dens) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    group_skip<SKIP, NT><<<256, NT>>>(ds, dout, R);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  const double cells = 256.0 * (NT / 64) * R * 32 * 256;
  printf("group4 %-6s NT=%d leaf density %.3f (matrix %.3f): %.3f ms  %.1f T covered terms/s\n", SKIP ? "skip" : "dense", NT, d0, dens, ms, cells / ms / 1e9);
}

static void group_main() {
  // a 4096-sample matrix whose rows have the densities of a balanced tree's levels over
  // leaves of density d0: level l (probability 2^-(l+1)) has density 1-(1-d0)^(2^l)
  const int R = 8192;
  uint32_t *ds, *dout;
  CK(hipMalloc(&ds, (size_t)R * 4096 * 4)); CK(hipMalloc(&dout, 256 * 1024 * 4));
  std::vector<uint32_t> hs((size_t)R * 4096);
  const double d0s[] = {0.10, 0.05, 0.02, 0.01, 0.002};
  for (double d0 : d0s) {
    double nz = 0;
    for (int r = 0; r < R; ++r) {
      int l = 0; while (l < 14 && (rand() & 1)) ++l;
      double d = 1.0 - pow(1.0 - d0, (double)(1 << l));
      for (int i = 0; i < 4096; ++i) { bool on = rand() < d * RAND_MAX; hs[(size_t)r * 4096 + i] = on ? (uint32_t)rand() | 1u : 0u; nz += on; }
    }
    CK(hipMemcpy(ds, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    bench_group<false, 512>(ds, dout, R, d0, nz / hs.size());
    bench_group<true, 512>(ds, dout, R, d0, nz / hs.size());
    bench_group<true, 768>(ds, dout, R, d0, nz / hs.size());
  }
}

int main() {
  group_main();

  const int R = 8192, W = 1024;
  uint32_t *dm, *ds, *dout;
  CK(hipMalloc(&dm, (size_t)W * R * 4)); CK(hipMalloc(&ds, (size_t)R * 32 * 4)); CK(hipMalloc(&dout, 256 * 1024 * 4));
  std::vector<uint32_t> hs((size_t)R * 32);
  for (auto& x : hs) x = rand();
  CK(hipMemcpy(ds, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
  const double dens[] = {1.0, 0.25, 0.0};
  for (double d : dens) {
    std::vector<uint32_t> hm((size_t)W * R);
    for (auto& x : hm) { uint32_t m = 0; for (int b = 0; b < 32; ++b) if (rand() < d * RAND_MAX) m |= 1u << b; x = m; }
    CK(hipMemcpy(dm, hm.data(), hm.size() * 4, hipMemcpyHostToDevice));
    if (d == 1.0) {
      bench<32, 4, 0, 512>("dense", dm, ds, dout, R, d);
      bench<16, 8, 0, 512>("dense", dm, ds, dout, R, d);
    }
    bench<32, 4, 1, 512>("skip sad+sub", dm, ds, dout, R, d);
    bench<16, 8, 1, 512>("skip sad+sub", dm, ds, dout, R, d);
    bench<16, 8, 1, 768>("skip sad+sub", dm, ds, dout, R, d);
    bench<32, 4, 2, 512>("skip sad only", dm, ds, dout, R, d);
    bench<16, 8, 2, 512>("skip sad only", dm, ds, dout, R, d);
  }
  return 0;
}

// Scalar operands two rows ahead: a 16 x 512 pair tile needs 16 SGPRs per row, so four sets fit
// and the loads of rows r+2, r+3 can be issued together at the start of row r (SMEM returns out of
// order: only lgkmcnt(0) is usable, so what counts is the age of the YOUNGEST load when the wait
// comes -- two rows here, one in the 32 x 256 product kernel).  Same harness as tile_sweep.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ void sad_acc(uint32_t s, uint32_t v, uint32_t& acc) { asm("v_sad_u32 %0, %1, %2, %0" : "+v"(acc) : "s"(s), "v"(v)); }
struct Item { int32_t i0, j0, k0, k1; };
typedef const uint32_t __attribute__((address_space(4))) *cp;
template <int JPL> struct Vec { uint32_t v[JPL]; };
template <int JPL> __device__ __forceinline__ Vec<JPL> vload(const uint32_t* p) {
  Vec<JPL> r;
#pragma unroll
  for (int q = 0; q < JPL / 4; ++q) { uint4 t = *(const uint4*)(p + 4 * q); r.v[4 * q] = t.x; r.v[4 * q + 1] = t.y; r.v[4 * q + 2] = t.z; r.v[4 * q + 3] = t.w; }
  return r;
}

// TI x (64*JPL) tile; DIST = 1: scalars one row ahead (two sets); DIST = 2: bursts of two rows (four sets)
template <int TI, int JPL, int DIST, int KS, int WAVES>
__global__ __launch_bounds__(WAVES * 64)
void k(const uint32_t* __restrict__ QT, int64_t ld, const Item* __restrict__ items, uint32_t* __restrict__ num)
{
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slot = blockIdx.x * WAVES + wave;
  const Item item = items[slot];
  const uint32_t* pj = QT + (int64_t)item.k0 * ld + item.j0 + JPL * lane;
  cp ps = (cp)(QT + (int64_t)item.k0 * ld + item.i0);
  uint32_t acc[JPL][TI];
#pragma unroll
  for (int c = 0; c < JPL; ++c)
#pragma unroll
    for (int r = 0; r < TI; ++r) acc[c][r] = 0;
  Vec<JPL> vA[KS], vB[KS];
#pragma unroll
  for (int d = 0; d < KS; ++d) vA[d] = vload<JPL>(pj + (int64_t)d * ld);
  const uint32_t* pv = pj + (int64_t)KS * ld;
  const int nk = item.k1 - item.k0;
#define ROW(S, V) { _Pragma("unroll") for (int r = 0; r < TI; ++r) { _Pragma("unroll") for (int c = 0; c < JPL; ++c) sad_acc(S[r], (V).v[c], acc[c][r]); } }
#define FILL(BUF) { _Pragma("unroll") for (int q = 0; q < KS; ++q) { BUF[q] = vload<JPL>(pv); pv += ld; } }
  if constexpr (DIST == 1) {
    uint32_t sA[TI], sB[TI];
#pragma unroll
    for (int r = 0; r < TI; ++r) sA[r] = ps[r];
#define STEP1(SCUR, SNXT, V, PRE) { sad_acc(SCUR[0], (V).v[0], acc[0][0]); __builtin_amdgcn_sched_barrier(0); ps += ld; \
      _Pragma("unroll") for (int r = 0; r < TI; ++r) SNXT[r] = ps[r]; PRE; __builtin_amdgcn_sched_barrier(0); ROW(SCUR, V) }
    for (int kk = 0; kk < nk; kk += 2 * KS) {
      STEP1(sA, sB, vA[0], FILL(vB))
#pragma unroll
      for (int d = 1; d < KS; d += 2) { STEP1(sB, sA, vA[d], ) if (d + 1 < KS) STEP1(sA, sB, vA[d + 1], ) }
      STEP1(sA, sB, vB[0], FILL(vA))
#pragma unroll
      for (int d = 1; d < KS; d += 2) { STEP1(sB, sA, vB[d], ) if (d + 1 < KS) STEP1(sA, sB, vB[d + 1], ) }
    }
  } else {
    uint32_t s0[TI], s1[TI], s2[TI], s3[TI];
#pragma unroll
    for (int r = 0; r < TI; ++r) { s0[r] = ps[r]; s1[r] = ps[r + ld]; }
    // a pair of rows from (SA, SB); the next pair's 2 x TI scalars are requested first, into (NA, NB)
#define PAIR(SA, SB, NA, NB, V0, V1, PRE) { sad_acc(SA[0], (V0).v[0], acc[0][0]); __builtin_amdgcn_sched_barrier(0); ps += 2 * ld; \
      _Pragma("unroll") for (int r = 0; r < TI; ++r) { NA[r] = ps[r]; } _Pragma("unroll") for (int r = 0; r < TI; ++r) { NB[r] = ps[r + ld]; } PRE; \
      __builtin_amdgcn_sched_barrier(0); ROW(SA, V0) ROW(SB, V1) }
    static_assert(KS % 2 == 0 && (KS / 2) % 2 == 0 || KS == 2, "KS");
    for (int kk = 0; kk < nk; kk += 2 * KS) {
      PAIR(s0, s1, s2, s3, vA[0], vA[1], FILL(vB))
      if constexpr (KS >= 4) PAIR(s2, s3, s0, s1, vA[2], vA[3], )
      if constexpr (KS >= 8) { PAIR(s0, s1, s2, s3, vA[4], vA[5], ) PAIR(s2, s3, s0, s1, vA[6], vA[7], ) }
      if constexpr (KS == 2) { PAIR(s2, s3, s0, s1, vB[0], vB[1], FILL(vA)) }
      else {
        PAIR(s0, s1, s2, s3, vB[0], vB[1], FILL(vA))
        if constexpr (KS >= 4) PAIR(s2, s3, s0, s1, vB[2], vB[3], )
        if constexpr (KS >= 8) { PAIR(s0, s1, s2, s3, vB[4], vB[5], ) PAIR(s2, s3, s0, s1, vB[6], vB[7], ) }
      }
    }
  }
  uint32_t t = 0;
#pragma unroll
  for (int c = 0; c < JPL; ++c)
#pragma unroll
    for (int r = 0; r < TI; ++r) t += acc[c][r];
  num[(size_t)slot * 64 + lane] = t;
}

template <int TI, int JPL, int DIST, int KS, int WAVES>
void run(const uint32_t* dQ, int64_t ld, int N, int B) {
  const int TJ = JPL * 64;
  std::vector<std::pair<int,int>> tiles;
  for (int i0 = 0; i0 < N; i0 += TI) for (int j0 = 0; j0 < i0 + TI - 1 && j0 + TJ <= N; j0 += TJ) tiles.push_back({i0, j0});
  const int U = 256 * WAVES;
  std::vector<Item> items;
  size_t off = tiles.size() > (size_t)U ? tiles.size() - U : 0;
  for (int u = 0; u < U; ++u) { auto t = tiles[(off + u) % tiles.size()]; items.push_back({t.first, t.second, 0, B}); }
  Item* dI; CK(hipMalloc(&dI, items.size() * sizeof(Item))); CK(hipMemcpy(dI, items.data(), items.size() * sizeof(Item), hipMemcpyHostToDevice));
  uint32_t* dnum; CK(hipMalloc(&dnum, (size_t)U * 64 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto fn = k<TI, JPL, DIST, KS, WAVES>;
  CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    fn<<<256, WAVES * 64, 96 * 1024>>>(dQ, ld, dI, dnum);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
  }
  double sads = (double)U * TI * TJ * B;
  printf("tile %2d x %3d  scalars %d row(s) ahead  vector buffers 2 x %d rows  %2d waves/CU: %.3f ms  %.2f T sad/s\n", TI, TJ, DIST, KS, WAVES, best, sads / best / 1e9);
  CK(hipFree(dI)); CK(hipFree(dnum));
}

int main() {
  int N = 4096, B = 20000;
  int64_t ld = N; size_t rows = B + 64;
  std::vector<uint32_t> h(rows * ld);
  uint64_t s = 42;
  for (auto& x : h) { s = s * 6364136223846793005ull + 1442695040888963407ull; x = ((s >> 33) & 3) == 0 ? (uint32_t)(s >> 44) : 0; }
  uint32_t* dQ; CK(hipMalloc(&dQ, rows * ld * 4)); CK(hipMemcpy(dQ, h.data(), rows * ld * 4, hipMemcpyHostToDevice));
  run<32, 4, 1, 8, 8>(dQ, ld, N, B);    // the product kernel's shape
  run<32, 4, 1, 4, 12>(dQ, ld, N, B);   // its 12-wave variant (FF_REG12)
  run<16, 8, 2, 4, 8>(dQ, ld, N, B);    // 16 x 512, scalars in bursts of two rows
  run<16, 8, 1, 2, 12>(dQ, ld, N, B);   // 16 x 512, three waves per SIMD, scalars one row ahead
  run<16, 8, 2, 2, 12>(dQ, ld, N, B);   // ... and in bursts of two rows
  run<16, 4, 2, 4, 16>(dQ, ld, N, B);   // 16 x 256, four waves per SIMD
  return 0;
}

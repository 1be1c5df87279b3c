#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
// one wave: C[32][32] = A[32][K] * B[32][K]^T, A,B int8 row-major (K contiguous)
__global__ void k(const int8_t* A, const int8_t* B, int K, int* C) {
  int l = threadIdx.x;
  v16i acc = {0};
  for (int kk = 0; kk < K; kk += 32) {
    v4i a = *(const v4i*)(A + (size_t)(l & 31) * K + kk + 16 * (l >> 5));
    v4i b = *(const v4i*)(B + (size_t)(l & 31) * K + kk + 16 * (l >> 5));
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    C[row * 32 + col] = acc[r];
  }
}
int main() {
  const int K = 128;
  std::vector<int8_t> A(32 * K), B(32 * K);
  for (int i = 0; i < 32 * K; ++i) { A[i] = (int8_t)((i * 7 + i / 13) % 3 == 0); B[i] = (int8_t)((i * 31 + 5) % 127); }
  int8_t *dA, *dB; int* dC; CK(hipMalloc(&dA, 32 * K)); CK(hipMalloc(&dB, 32 * K)); CK(hipMalloc(&dC, 4096));
  CK(hipMemcpy(dA, A.data(), 32 * K, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 32 * K, hipMemcpyHostToDevice));
  k<<<1, 64>>>(dA, dB, K, dC);
  std::vector<int> C(1024); CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
  int bad = 0, badT = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    int ref = 0; for (int kk = 0; kk < K; ++kk) ref += (int)A[i * K + kk] * (int)B[j * K + kk];
    if (C[i * 32 + j] != ref) ++bad;
    if (C[j * 32 + i] != ref) ++badT;
  }
  printf("mismatches: C[i][j]=A_i.B_j : %d ; transposed: %d\n", bad, badT);
  return 0;
}

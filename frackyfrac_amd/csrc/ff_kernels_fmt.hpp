// ff_kernels_fmt.hpp -- the output formatter on the device: the distances of a pass become the text the
// reference's loop `for f := range dists { fmt.Fprintln(fout, f) }` writes (frcfrc/frcfrc.go:58-62) while they are
// still in HBM, so the host receives bytes to append to the file and formats nothing.
//
// The text of a value is 1 .. 25 bytes long (ff_fmt_core.hpp + the newline), so where a value's text starts
// depends on every value before it.  Three launches over blocks of FMT_BLOCK_VALUES consecutive values:
//   fmt_len_kernel    bytes of every block's text                              (reads 8 B per value)
//   fmt_scan_kernel   exclusive prefix sums of those: every block's byte offset (one workgroup)
//   fmt_write_kernel  formats the block again into LDS at the offsets of an in-block scan and copies the
//                     LDS image to the block's place in the text, coalesced    (8 B in, ~19 B out per value)
// HBM-bound by design: 16 B read + ~19 B written per value, no intermediate image of the digits.  (Computing the
// digits twice costs three 64 x 64 -> 128-bit products each time: nothing next to the memory traffic.)
#pragma once

#include "ff_fmt_core.hpp"

constexpr int FMT_THREADS = 256;
constexpr int FMT_PER_THREAD = 4;
constexpr int FMT_BLOCK_VALUES = FMT_THREADS * FMT_PER_THREAD;       // 1,024 values per workgroup
constexpr int FMT_LINE_MAX = ff::fmt::MAX_CHARS + 1;                 // with the newline
constexpr int FMT_BLOCK_BYTES_MAX = FMT_BLOCK_VALUES * FMT_LINE_MAX;  // 25,600 B of LDS
constexpr int FMT_SCAN_THREADS = 1024;

__device__ __forceinline__ uint32_t fmt_wave_inclusive_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(v, d, 64);
        if (lane >= d) v += up;
    }
    return v;
}

// Bytes of the text of block b (its values' lengths + one newline each) into block_bytes[b].
__global__ __launch_bounds__(FMT_THREADS) void fmt_len_kernel(const double *__restrict__ vals, int64_t n,
                                                              uint32_t *__restrict__ block_bytes)
{
    __shared__ uint32_t wave_sum[FMT_THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * FMT_BLOCK_VALUES + (int64_t)threadIdx.x * FMT_PER_THREAD;
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < FMT_PER_THREAD; ++j)
        if (base + j < n) mine += (uint32_t)ff::fmt::shape_of((uint64_t)__double_as_longlong(vals[base + j])).len + 1u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t s = 0;
        for (int w = 0; w < FMT_THREADS / 64; ++w) s += wave_sum[w];
        block_bytes[blockIdx.x] = s;
    }
}

// block_off[b] = sum of block_bytes[0 .. b), block_off[n_blocks] = the whole text's length.  One workgroup.
__global__ __launch_bounds__(FMT_SCAN_THREADS) void fmt_scan_kernel(const uint32_t *__restrict__ block_bytes, int64_t n_blocks,
                                                                    unsigned long long *__restrict__ block_off)
{
    __shared__ unsigned long long wave_tot[FMT_SCAN_THREADS / 64];
    __shared__ unsigned long long carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < n_blocks; b0 += FMT_SCAN_THREADS) {
        const int64_t b = b0 + threadIdx.x;
        const uint32_t v = b < n_blocks ? block_bytes[b] : 0u;
        const uint32_t inc = fmt_wave_inclusive_scan(v, lane);  // (a wave's 64 blocks hold < 2^32 bytes)
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        unsigned long long before = carry_s;
        for (int w = 0; w < wave; ++w) before += wave_tot[w];
        if (b < n_blocks) block_off[b] = before + (inc - v);
        __syncthreads();
        if (threadIdx.x == FMT_SCAN_THREADS - 1) carry_s = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) block_off[n_blocks] = carry_s;
}

// The text of block b at text[block_off[b] ..).
__global__ __launch_bounds__(FMT_THREADS) void fmt_write_kernel(const double *__restrict__ vals, int64_t n,
                                                                const unsigned long long *__restrict__ block_off,
                                                                char *__restrict__ text)
{
    __shared__ char image[FMT_BLOCK_BYTES_MAX];
    __shared__ uint32_t wave_tot[FMT_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * FMT_BLOCK_VALUES + (int64_t)threadIdx.x * FMT_PER_THREAD;
    ff::fmt::Shape sh[FMT_PER_THREAD];
    bool neg[FMT_PER_THREAD];
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < FMT_PER_THREAD; ++j) {
        sh[j].len = -1;
        if (base + j < n) {
            const uint64_t bits = (uint64_t)__double_as_longlong(vals[base + j]);
            sh[j] = ff::fmt::shape_of(bits);
            neg[j] = (bits >> 63) != 0;
            mine += (uint32_t)sh[j].len + 1u;
        }
    }
    const uint32_t inc = fmt_wave_inclusive_scan(mine, lane);
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t at = inc - mine, total = 0;
    for (int w = 0; w < FMT_THREADS / 64; ++w) {
        if (w < wave) at += wave_tot[w];
        total += wave_tot[w];
    }
#pragma unroll
    for (int j = 0; j < FMT_PER_THREAD; ++j)
        if (sh[j].len >= 0) {
            ff::fmt::write_shape(sh[j], neg[j], &image[at]);
            image[at + (uint32_t)sh[j].len] = '\n';
            at += (uint32_t)sh[j].len + 1u;
        }
    __syncthreads();
    // LDS image -> text: bytes up to the first 4-byte boundary of the destination, then whole dwords, then the tail
    char *dst = text + block_off[blockIdx.x];
    const uint32_t head = min(total, (uint32_t)((4u - ((uintptr_t)dst & 3u)) & 3u));
    if (threadIdx.x < head) dst[threadIdx.x] = image[threadIdx.x];
    const uint32_t n_dwords = (total - head) >> 2;
    uint32_t *dst4 = reinterpret_cast<uint32_t *>(dst + head);
    for (uint32_t w = threadIdx.x; w < n_dwords; w += FMT_THREADS) {
        const uint32_t p = head + 4u * w;
        dst4[w] = (uint32_t)(unsigned char)image[p] | ((uint32_t)(unsigned char)image[p + 1] << 8) |
                  ((uint32_t)(unsigned char)image[p + 2] << 16) | ((uint32_t)(unsigned char)image[p + 3] << 24);
    }
    const uint32_t done = head + 4u * n_dwords;
    if (threadIdx.x < total - done) dst[done + threadIdx.x] = image[done + threadIdx.x];
}

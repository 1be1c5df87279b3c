// ff_schedule.hpp -- tile geometry and host-side work schedules of the pair kernels
// (pure host C++: built with g++, unit-tested on CPU through ff_debug_schedule).
#pragma once

#include <algorithm>
#include <cstdint>
#include <utility>
#include <vector>

namespace ff {
namespace sched {

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// ---- v_sad_u32 kernel (pair_sad_kernel) and EXACT64 kernel ----
constexpr int TILE_I = 32;    // rows of a wave's pair tile: samples held in SGPRs
constexpr int TILE_J = 256;   // columns: 4 per lane (one 16-byte load per lane and row)
constexpr int KSTEP = 8;      // branch rows per vector buffer; the loop body covers 2*KSTEP rows
constexpr int SLACK_ROWS = 32;  // zero rows past the matrix, read by the prefetch
constexpr int WAVES_PER_WG = 8; // 512-thread workgroups: two waves per SIMD
constexpr int L_WAVES_PER_WG = 12;  // 768-thread workgroups of pair_sad_kernel12 (half the vector buffers): three per SIMD
constexpr int SYNC_TRIPS = 16;      // main-round items: a workgroup barrier every this many loop trips
// EXACT64 tile: H rows x 64 columns per wave, H one of these (picked per plan: ff_dev_run.hip schedule_exact64)
constexpr int X_TILE_HEIGHTS[] = {4, 8, 10, 12, 14, 16};
constexpr int X_TILE_H_DEFAULT = 12;
constexpr int X_TILE_J = 64;

// ---- How far past an item's end the kernels read, and the padding that covers it --------------------
// Three kernels over-read BY DESIGN (their prefetches run ahead of the loop's exit test; what they fetch
// past the end is zero padding or never used).  Every such distance is a constant here, next to the
// padding that has to cover it, and the two are tied at compile time; the kernels static_assert their own
// loop shape against the same constants, ff_dev_stage.hip sizes the allocations with the functions below, and
// tests/test_schedule_cpu.py replays each kernel's address stream per item against those sizes.
//   pair_sad_kernel / pair_sad_kernel12: the vector buffer refilled during an item's last trip holds the rows
//   k1 .. k1 + KS - 1 (KS = KSTEP or KSTEP / 2 rows per buffer); the scalar operands stop at row k1.
constexpr int SAD_ROWS_AHEAD = KSTEP;
//   pair_sad_sparse_kernel: absent list entries are replaced by the FIRST slack row (all zero); row numbers
//   are fetched in batches of four, two batches ahead: the trip at list position t < a1 reads entries up to t + 11.
constexpr int SPARSE_LIST_AHEAD = 11;
constexpr int SPARSE_LIST_PAD = 16;  // spare entries behind the active-row list
static_assert(SLACK_ROWS >= SAD_ROWS_AHEAD && SLACK_ROWS >= 1, "the staged matrix's slack rows must cover the pair kernels' prefetch");
static_assert(SPARSE_LIST_PAD > SPARSE_LIST_AHEAD, "the active-row list's spare entries must cover the batch prefetch");
inline int64_t sad_staged_rows(int64_t R) { return round_up(R, 2 * KSTEP); }  // rows the items cover
inline int64_t sad_alloc_rows(int64_t R) { return sad_staged_rows(R) + SLACK_ROWS; }
//   pair_exact64_kernel<.., H>: a wave reads H consecutive operands from column i0 of every row; in the last
//   row of the matrix that is up to H - 1 values past the end.
constexpr int X_VALUES_PAD = 16;
static_assert(X_VALUES_PAD >= X_TILE_HEIGHTS[sizeof(X_TILE_HEIGHTS) / sizeof(int) - 1] - 1, "EXACT64: padding behind the last row");

// ---- rare rows of a sparse table (pair_low_kernel) ----
// A staged row that few samples have a flat node on ("rare": at most N / LOW_SHARE_DIV of them, fewer where the cost
// estimate of ff_dev_stage.hip says so) is kept out of the dense
// matrix: its contribution sum |q_i - q_j| = q_i + q_j - 2 min(q_i, q_j) needs work only where BOTH samples have it.
// The rows' entries are grouped by blocks of T samples; a workgroup owns a T x T block of pairs and sums min(q_i, q_j)
// over the rows with entries in both of its sample blocks.  T is one of LOW_TILES, picked per plan: the blocks of pairs
// are dealt to two workgroups per CU, and what decides is how full their last round is (C3, 4,096 samples: 528 blocks
// of 128 x 128 on 512 slots are two rounds, the second all but empty; 946 of 96 x 96 fill 1.85).
constexpr int LOW_TILES[] = {128, 112, 96, 80, 64};
constexpr int LOW_TILE_MAX = 128;
constexpr int LOW_THREADS = 1024;  // 16 waves on one tile's accumulators: the kernel lives on waves in flight (latency)
constexpr int LOW_SHARE_DIV = 2;
// An entry of a rare row carries its sample's place in the block as the BYTE offset of the accumulator's row in LDS --
// li * LOW_STRIDE * 4 -- so that an update's address is one OR: (A entry's x) | (B entry's x >> LOW_COL_SHIFT = lj * 4).
constexpr int LOW_STRIDE = 128;     // words between accumulator rows, whatever the block side
constexpr int LOW_ROW_SHIFT = 9;    // li -> li * LOW_STRIDE * 4
constexpr int LOW_COL_SHIFT = 7;    // li * LOW_STRIDE * 4 -> li * 4
struct LowTile {
    int32_t bi, bj;  // sample blocks: pairs (i, j) with i in block bi, j in block bj <= bi, j < i
};

struct Item {        // one unit of work for a persistent wave: a pair tile over a
    int32_t i0, j0;  // branch range [k0, k1)
    int32_t k0, k1;  // multiples of 2*KSTEP
    uint32_t flags;  // bit 0: other items add to the same outputs -> atomic add
                     // bit 1: all waves of the workgroup run an item of this length now
                     // bit 2: half-width tile (32 x 128)
    int32_t pad[3];
};
static_assert(sizeof(Item) == 32, "Item must be 32 bytes");

struct XTile {
    int32_t i0, j0;
};

// ---- EXACT64 unweighted kernel (pair_exact_unw_kernel): presence bits, two chains of additions per pair ----
constexpr int XU_TILE_H = 8;       // rows of a wave's tile (scalar side)
constexpr int XU_SLAB = 32;        // staged rows per presence word
constexpr int XU_JMAX = 2;         // 64-sample column groups of a tile: 1 or 2
constexpr int XU_LEN_STEP = 8;     // lengths fetched at a time (one s_load_dwordx16)
//   the kernel requests slab s + 1's words while it works on slab s: one slab past the end is read (zeros); the
//   lengths are read in whole slabs
constexpr int XU_PAD_SLABS = 1;
constexpr int XU_LEN_PAD = 0;
inline int64_t xu_slabs(int64_t R) { return (std::max<int64_t>(R, 1) + XU_SLAB - 1) / XU_SLAB; }
inline int64_t xu_alloc_slabs(int64_t R) { return xu_slabs(R) + XU_PAD_SLABS; }
inline int64_t xu_alloc_lengths(int64_t R) { return xu_slabs(R) * XU_SLAB + XU_LEN_PAD; }
inline int64_t xu_ld(int64_t N) { return round_up(std::max<int64_t>(N, 1), 64 * XU_JMAX); }  // samples per slab row
struct XUTile {
    int32_t i0, j0;  // rows i0 .. i0 + XU_TILE_H - 1, columns j0 + 64 t + lane, t < jn
    int32_t jn;      // 64-sample column groups: 1 or 2 (the last tile of a row block may be the narrower one)
    int32_t pad;
};
static_assert(sizeof(XUTile) == 16, "XUTile must be 16 bytes");
// The tiles of rows [rb, re), widest first (the hardware hands them out in this order); jmax = 1: single groups only
// (a shard of few tiles is bound by one wave's chain of steps: narrower tiles, more waves).
void build_xu_tiles(int64_t N, int64_t rb, int64_t re, int jmax, std::vector<XUTile> *tiles);


// ---- int8 MFMA kernel (pair_common_mfma_kernel) ----
constexpr int M_TILE_I = 256;  // workgroup tile: 256 i-samples x 128 j-samples,
constexpr int M_TILE_J = 128;  //   4 waves (2 x 2) of 128 x 64, i.e. 4 x 2 MFMA tiles per wave and digit plane
constexpr int M_THREADS = 256; // one wave per SIMD (256 accumulator registers per lane)
constexpr int M_WGS_PER_CU = 1;
constexpr int M_KSLAB = 64;    // branches per LDS slab (two K = 32 MFMA steps) = bits of a presence word
constexpr int M_QUAD_SLABS = 4; // an item of pair_common_mfma_kernel is a whole number of these, and starts at one
//   pair_common_mfma_kernel keeps M_PAIRS_IN_FLIGHT pairs of slabs of presence words in flight: its prologue
//   requests that many pairs whatever the item's length, and the loop requests pair p + M_PAIRS_IN_FLIGHT when
//   it is done with pair p: up to 2 * M_PAIRS_IN_FLIGHT slabs past an item's end are read (zeros).
constexpr int M_PAIRS_IN_FLIGHT = 4;
constexpr int M_SLABS_AHEAD = 2 * M_PAIRS_IN_FLIGHT;
constexpr int M_PAD_SLABS = 12; // slabs of zero padding behind the presence words and the digit arrays
static_assert(M_PAD_SLABS >= M_SLABS_AHEAD && M_PAD_SLABS % 2 == 0, "the prefetch of pair_common_mfma_kernel must stay inside the padding");
inline int64_t mfma_staged_slabs(int64_t R) { return round_up(R > 0 ? R : 1, (int64_t)M_KSLAB * M_QUAD_SLABS) / M_KSLAB; }
inline int64_t mfma_alloc_slabs(int64_t R) { return mfma_staged_slabs(R) + M_PAD_SLABS; }
constexpr int M_ND = 2;        // digit planes multiplied per sweep of the presence operand
// pair_common_mfma_kernel<.., GRADED>: three SIGNED digit planes in one sweep, k = d0 + 128 d1 + 32768 d2 with
// d0 in [-64, 63], d1 in [-127, 128], d2 in [0, 127]; where every d2 is zero the sweep multiplies two planes.
// (An accumulator may wrap on the way: v_mfma_i32 adds in two's complement -- tools/microbench/mfma_i8_wrap.hip --
// and the sum it is part of, common(i, j) < 2^31, comes out modulo 2^32.)
constexpr int64_t TRI_KMAX = 63 + 128 * (128 + 256 * 127);      // 4,177,983: the largest such k
constexpr int64_t DUO_KMAX = 63 + 128 * 128;                    // 16,447: the largest with d2 = 0
// the three planes of k <= TRI_KMAX as the kernel multiplies them: {d0, -d1, d2}
inline void tri_digits(int64_t k, int8_t out[3])
{
    const int64_t d0 = ((k + 64) & 127) - 64, q1 = (k - d0) >> 7;  // k - d0: a multiple of 128, never negative
    const int64_t r1 = q1 & 255, d1 = r1 <= 128 ? r1 : r1 - 256;
    out[0] = (int8_t)d0;
    out[1] = (int8_t)(-d1);
    out[2] = (int8_t)((q1 - d1) >> 8);
}
constexpr int M_LDS_BYTES = M_TILE_I * M_TILE_J * 4;  // 128 KiB: the digit table of up to 512 slabs (64 KiB with two planes, 96
                                                      // with three: ff_kernels_mfma.hpp M_TABLE_SLABS), then the epilogue's
                                                      // 256 x 128 tile of 32-bit sums
// what one more item costs a workgroup of pair_common_mfma_kernel, in slabs of its loop (measured:
// about 10 us of prologue, accumulator write-out and copy-out against 0.63 us per two-digit slab)
constexpr int64_t M_ITEM_OVERHEAD_SLABS = 16;  // (a multiple of M_QUAD_SLABS)

// One unit of work: a 256 x 128 tile over the branch slabs [k0, k1) for the digit planes
// d0 .. d0+nd-1.  Every item adds its share of U = W_i + W_j - 2*common to num[]
// atomically (mod 2^32; U < 2^32); the item with `first` also brings W_i + W_j.
struct MItem {
    int32_t i0, j0;
    int32_t k0, k1;  // bytes (= branches), multiples of M_KSLAB
    int32_t d0, nd;  // nd in {1, 2}
    int32_t first;
    int32_t pad;     // 0: add into num[] atomically; -1: store into num[] (the tile's only item);
                     // p > 0: store into private partial tile p - 1
};
static_assert(sizeof(MItem) == 32, "MItem must be 32 bytes");


struct Tile {
    int32_t i0, j0;
    int32_t narrow;  // 1: 32 x 128 (the tile overhangs the diagonal by more than half)
};

int xcd_slices();
bool xcd_slices_forced();  // FF_XCD_SLICES is set: the XCD-sliced rounds wherever they apply
void build_tiles(int64_t N, int64_t rb, int64_t re, int ti, int tj, bool allow_narrow, std::vector<Tile> *tiles);
void build_schedule(const std::vector<Tile> &all_tiles, int64_t rows, int U, std::vector<Item> *items,
                    std::vector<int32_t> *item_ptr, double *elements, int xcds = 0, int wpw = WAVES_PER_WG,
                    int max_planes = 0, bool prefer_sliced = false);
int waves_per_wg();
// Returns the number of 256 x 128 tiles; items/item_ptr get one list per workgroup.
// partial_tiles / partial_ptr (may be null): filled when the ranges get private partial tiles
// (MItem.pad = ordinal + 1): (i0, j0) per tile and, per tile, its first ordinal (+ the total).
// max_private_tiles: when the whole schedule needs no more partial tiles than this, the items of the
// main rounds get one each as well (every item then stores aligned 512-byte rows and the reduce
// kernel writes the results); otherwise only the remainder's ranges do.
int64_t build_mfma_schedule(int64_t N, int64_t row_begin, int64_t row_end, int64_t slabs, int digits, int G,
                            std::vector<MItem> *items, std::vector<int32_t> *item_ptr,
                            std::vector<int32_t> *partial_tiles, std::vector<int32_t> *partial_ptr,
                            int64_t max_private_tiles = 0, int64_t duo_from_quad = -1);
//   duo_from_quad >= 0: a graded sweep (pair_common_mfma_kernel<.., GRADED>; `digits` = 2: one group per tile) whose
//   quads of slabs cost 12 up to that quad and 9 from there on (a quad of a base-128 two-plane sweep: 8); ranges are
//   cut by cost.

}  // namespace sched
}  // namespace ff

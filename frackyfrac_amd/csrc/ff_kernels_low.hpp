// ff_kernels_low.hpp -- the rare rows of a sparse table (FIXED32 weighted; ff_schedule.hpp LOW_*).
// A fragment of ff_dev_run.hip: included there, once, inside its anonymous namespace.
//
// unifracDistWeighted adds l * |x - y| per branch (frcfrc/unifrac.go:191) -- in the staged integers |q_i - q_j| =
// q_i + q_j - 2 min(q_i, q_j), and min(q_i, q_j) is zero unless BOTH samples have the branch.  For a row few samples
// reach, that is the reference's own economy -- its merge walk only ever touches branches one of the two samples has --
// made two-sided: U(i, j) = U_dense(i, j) + Wl_i + Wl_j - 2 M(i, j), with Wl_s the sample's sum over its rare rows and
// M(i, j) = sum over the rare rows both have of min(q_i, q_j).  Same integers as one v_sad_u32 per term gives, so the
// distances are the dense path's bit for bit (tests/test_gpu_parity.py::test_sparse_split_gives_the_same_integers).
//
// A workgroup owns a T x T block of pairs (T = LOW_TILE, one of ff_schedule.hpp's LOW_TILES, per plan) -- sample blocks
// bi (rows) and bj <= bi (columns) -- and keeps its M in LDS.  The rare rows' entries are grouped by sample block,
// block-major (ptr[k][r] .. ptr[k][r + 1]: row r's entries of block k, contiguous and ascending with r within the
// block); an entry is (x: the byte offset of its sample's accumulator row, li * LOW_STRIDE * 4; y: the staged value), so
// that an update's LDS address is A.x | B.x >> LOW_COL_SHIFT.  A bitmap per block says which rows have any entry.
// A WAVE takes the rows of a word of bits[bi] & bits[bj] at a time (or, where the words are sparse, 64 rows compacted
// from up to 64 of its words: below): lane l looks up its row -- the row's A entries of block bi, its B entries of block
// bj, A x B updates -- the B list lengths are scanned across the wave, and the wave works through the rows' B ENTRIES 64
// at a time: a lane finds its (row, B entry) by a binary search over the scanned lengths, loads the entry, and walks the
// row's A entries four loads ahead -- 29 instructions per four steps, no branch (past the row's end the add is of zero);
// the next 64's operands are fetched while these are worked.  Whatever the rows' weights every lane has an entry, and a
// round lasts as long as the longest A list among its rows.
// (History, measured at C3 / 1 % density: a thread per word with nested loops ran as long as the busiest lane of every
// step, 63 ms at 1 %; lanes dealt to the word's UPDATES (row, a, b) by search, 61 instructions per 64: 4.6 ms at C3;
// heavy rows one by one on scalar operands 4.30; by groups of G lanes, G a power of two, a row per group, one load per
// step 4.46 -- the load's latency in every step; four loads ahead 3.72; the address as one OR, the tail by adding zero
// instead of branching, the diagonal case compiled apart: 3.23; a round fetched ahead, block sides 112 and 80: 2.99;
// lanes dealt to the B entries without gaps -- this -- 2.65, and the search over updates, slower at every density, went.
// Sparse words compacted 64 rows at a time: nothing at C3, the kernel at 0.2 % / 0.05 % leaf density 2.14 -> 1.66 / 1.13 -> 0.74 ms.)
// Every slot of the tile is then written once (zeros included): no memset, no global atomics.
struct __attribute__((aligned(8))) LowQuad { uint2 e[4]; };
// One B entry (this lane's) against a row's A entries, four in flight at a time; past the row's end the loads bring other
// rows' entries (or the array's spare ones, zeros) and the add is of zero: no branch in the step.  DIAGONAL: the block
// of pairs lies on the diagonal, each pair once -- the other half adds zero (its cells are not written out anyway).
template <bool DIAGONAL>
__device__ __forceinline__ void low_quad(const LowQuad &e, uint32_t left, uint2 eb, char *col)
{
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        uint32_t v = min(e.e[u].y, eb.y);
        if (u > 0) v = left > (uint32_t)u ? v : 0u;
        if (DIAGONAL) v = eb.x < e.e[u].x ? v : 0u;
        atomicAdd((uint32_t *)(col + e.e[u].x), v);
    }
}
template <bool DIAGONAL>
__device__ __forceinline__ void low_walk_row(const uint2 *__restrict__ a, uint32_t rna, uint2 eb, char *col)
{
    for (uint32_t t = 0; t < rna; t += 4) low_quad<DIAGONAL>(*(const LowQuad *)(a + t), rna - t, eb, col);
}
struct LowRound { uint32_t ra0, rna; uint2 eb; bool on; };  // a lane's part of a round: its row's A list, its B entry, whether it has one

template <int LOW_TILE>
__global__ __launch_bounds__(LOW_THREADS)
void pair_low_kernel(const uint32_t *__restrict__ ptr, const uint2 *__restrict__ entries,
                     const unsigned long long *__restrict__ bits, int64_t words,
                     int64_t rows1, const LowTile *__restrict__ tiles, int64_t n_samples, int64_t row_begin, int64_t row_end,
                     int64_t slot_begin, uint32_t *__restrict__ mlow)
{
    __shared__ uint32_t acc[LOW_TILE * LOW_STRIDE];  // 64 / 48 / 32 KiB: row li at li * LOW_STRIDE words
    const LowTile tile = tiles[blockIdx.x];
    const int bi = tile.bi, bj = tile.bj;
    for (int q = threadIdx.x; q < LOW_TILE * LOW_STRIDE; q += LOW_THREADS) acc[q] = 0u;
    __syncthreads();
    const unsigned long long *wi = bits + (int64_t)bi * words, *wj = bits + (int64_t)bj * words;
    const uint32_t *pi = ptr + (int64_t)bi * rows1, *pj = ptr + (int64_t)bj * rows1;
    const uint32_t i_base = (uint32_t)bi * LOW_TILE, j_base = (uint32_t)bj * LOW_TILE;
    const bool diagonal = bi == bj;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // The wave's words are one of every 16 (the rows are numbered by weight: every wave gets its share of the heavy ones);
    // it looks at 64 of them at a time, a word per lane, and counts the rows each has in both sample blocks.
    //   Dense words (32 rows or more on average): word by word, lane l looks up row l of the word.
    //   Sparse ones (a table at 0.2 % density: 3 rows of 64): the 64 words' rows are compacted -- the k-th of them found by a
    //   search over the scanned counts, then the k-th set bit of that word -- and looked up 64 at a time, so that the look-
    //   ups, the scan and the search below are paid per 64 ROWS, not per word.
    constexpr int WAVES = LOW_THREADS / 64;
    // (boustrophedon: the g-th group of 16 words goes to the waves in ascending order for even g, descending for odd -- the
    // words' weights fall with their index, and wave 0 would get the heaviest word of every group)
    auto word_of = [&](int64_t g) { return WAVES * g + ((g & 1) ? WAVES - 1 - wave : wave); };
    for (int64_t chunk = 0;; ++chunk) {
        const int64_t my_word = word_of(64 * chunk + lane);
        const uint32_t n_words = (uint32_t)__builtin_popcountll(__ballot(my_word < words));  // (a prefix of the lanes)
        if (n_words == 0) break;
        const unsigned long long cw = my_word < words ? wi[my_word] & wj[my_word] : 0ull;
        const uint32_t pc = (uint32_t)__builtin_popcountll(cw);
        uint32_t incl_pc = pc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl_pc, d, 64);
            if (lane >= d) incl_pc += up;
        }
        const uint32_t rows_here = (uint32_t)__builtin_amdgcn_readlane((int)incl_pc, 63);  // (a scalar: what follows is uniform)
        if (rows_here == 0) continue;
        const bool dense = rows_here >= 32u * n_words;                                   // (uniform)
        const uint32_t trips = dense ? n_words : (rows_here + 63u) / 64u;
        for (uint32_t trip = 0; trip < trips; ++trip) {
        uint32_t a0 = 0, na = 0, b0 = 0, nbb = 0;
        {
            int64_t r = -1;
            if (dense) {
                const unsigned long long c = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)cw, (int)trip) |
                                             (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(cw >> 32), (int)trip) << 32;
                if (c == 0) continue;  // (uniform)
                if ((c >> lane) & 1ull) r = word_of(64 * chunk + (int64_t)trip) * 64 + lane;
            } else {
                const uint32_t k = trip * 64u + (uint32_t)lane;
                int L = 0;  // the lane whose word holds the chunk's k-th row: the number of lanes with incl_pc <= k
#pragma unroll
                for (int step = 32; step >= 1; step >>= 1) {
                    const uint32_t v = __shfl(incl_pc, L + step - 1, 64);
                    if (v <= k) L += step;
                }
                L = min(L, 63);
                uint32_t j = k - __shfl(incl_pc - pc, L, 64);  // which of that word's rows
                const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)cw, L, 64), hi = (uint32_t)__shfl((int)(uint32_t)(cw >> 32), L, 64);
                if (k < rows_here) {
                    // the j-th set bit of the word: halve the span six times
                    uint32_t x = lo, pos = 0;
                    const uint32_t in_lo = (uint32_t)__builtin_popcount(lo);
                    if (j >= in_lo) { j -= in_lo; x = hi; pos = 32; }
#pragma unroll
                    for (int width = 16; width >= 1; width >>= 1) {
                        const uint32_t below = (uint32_t)__builtin_popcount(x & ((1u << width) - 1u));
                        if (j >= below) { j -= below; x >>= width; pos += width; }
                    }
                    r = word_of(64 * chunk + (int64_t)L) * 64 + pos;
                }
            }
            if (r >= 0) {
                a0 = pi[r];
                na = pi[r + 1] - a0;
                b0 = pj[r];
                nbb = pj[r + 1] - b0;
            }
        }
        // The lanes are dealt to the word's B entries, row after row without gaps (lane -> (row, entry) by a binary
        // search over the scanned list lengths), 64 at a time, and every lane walks its row's A entries: no lane idles
        // for a short list, a round lasts as long as the longest A list among its rows (the rows are numbered by weight:
        // alike).  Lanes along B: the accumulator's row is A's, so a row's adds fall in one LDS row, bank by bank.
        uint32_t inclb = nbb;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(inclb, d, 64);
            if (lane >= d) inclb += up;
        }
        const uint32_t totalb = (uint32_t)__builtin_amdgcn_readlane((int)inclb, 63);  // (a scalar: the rounds' loop is uniform)
        const uint32_t b_at = b0 - (inclb - nbb);  // + k: the place of the word's k-th B entry, for k in this lane's row (mod 2^32)
        // (a round's operands -- the search, three shuffles, the lane's B entry -- are fetched while the round before is
        // worked)
        auto fetch = [&](uint32_t k0) {
            LowRound r;
            const uint32_t k = k0 + (uint32_t)lane;
            int L = 0;  // the lane whose row holds B entry k: the number of lanes with inclb <= k
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
                const uint32_t v = __shfl(inclb, L + step - 1, 64);
                if (v <= k) L += step;
            }
            L = min(L, 63);  // (lanes past the word's last entry)
            r.on = k < totalb;
            const uint32_t at = __shfl(b_at, L, 64) + k;
            r.ra0 = __shfl(a0, L, 64), r.rna = __shfl(na, L, 64);
            r.eb = entries[r.on ? at : 0u];
            return r;
        };
        LowRound cur = fetch(0);
        for (uint32_t k0 = 0; k0 < totalb; k0 += 64) {
            LowRound nxt = cur;
            if (k0 + 64 < totalb) nxt = fetch(k0 + 64);  // (uniform)
            if (cur.on) {
                char *col = (char *)acc + (cur.eb.x >> LOW_COL_SHIFT);
                if (diagonal) low_walk_row<true>(entries + cur.ra0, cur.rna, cur.eb, col);  // (uniform)
                else low_walk_row<false>(entries + cur.ra0, cur.rna, cur.eb, col);
            }
            cur = nxt;
        }
        }  // (trip)
    }
    __syncthreads();
    // the tile's slots: row i = i_base + li holds columns j_base .. of slot i (i - 1) / 2 + j -- contiguous in j
    for (int q = threadIdx.x; q < LOW_TILE * LOW_TILE; q += LOW_THREADS) {
        const int li = q / LOW_TILE, lj = q % LOW_TILE;
        const int64_t i = (int64_t)i_base + li, j = (int64_t)j_base + lj;
        if (i < row_begin || i >= row_end || i >= n_samples || j >= i) continue;
        mlow[i * (i - 1) / 2 - slot_begin + j] = acc[li * LOW_STRIDE + lj];
    }
}

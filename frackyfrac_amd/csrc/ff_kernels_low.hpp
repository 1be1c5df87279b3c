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
// A WAVE takes a word of bits[bi] & bits[bj] at a time: lane l looks up row 64 w + l -- its A entries of block bi, its
// B entries of block bj, A x B updates -- and the counts are scanned across the wave.  Then one of two ways:
//   light words (fewer than LOW_ROWWISE_MIN updates a row on average): the word's updates 64 at a time, every lane
//     finding its (row, a, b) by a binary search over the scanned counts -- whatever the rows' weights, an instruction
//     does 64 updates (61 instructions per 64);
//   heavy words: the wave splits into groups of G lanes (G = 8 .. 64, the power of two that holds the word's average
//     B list), a group takes a row, its lanes the row's B entries, and walks the row's A entries four loads ahead:
//     29 instructions per four steps of 64 / G rows each (low_walk_row).
// (History, measured at C3 / 1 % density: a thread per word with nested loops ran as long as the busiest lane of every
// step, 63 ms at 1 %; the search alone 4.6 ms at C3; rows one by one on scalar operands 4.30; groups with one load
// per step 4.46 -- the load's latency in every step; four loads ahead 3.72; the address as one OR, the tail by
// adding zero instead of branching, the diagonal case compiled apart: 3.23.)
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
// (first: the row's first four A entries, loaded a round ahead)
template <bool DIAGONAL>
__device__ __forceinline__ void low_walk_row(const uint2 *__restrict__ a, uint32_t rna, uint2 eb, char *col)
{
    for (uint32_t t = 0; t < rna; t += 4) low_quad<DIAGONAL>(*(const LowQuad *)(a + t), rna - t, eb, col);
}
struct LowRound { uint32_t ra0, rna, rb0, rnb; uint2 eb; };  // a group's row: its lists, this lane's B entry, the first A entries

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
    for (int64_t w = wave; w < words; w += LOW_THREADS / 64) {
        const unsigned long long c = wi[w] & wj[w];  // (uniform: one word per wave)
        if (c == 0) continue;
        uint32_t a0 = 0, na = 0, b0 = 0, nbb = 0;
        if ((c >> lane) & 1ull) {
            const int64_t r = w * 64 + lane;
            a0 = pi[r];
            na = pi[r + 1] - a0;
            b0 = pj[r];
            nbb = pj[r + 1] - b0;
        }
        const uint32_t cnt = na * nbb;
        uint32_t incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        const uint32_t total = __shfl(incl, 63, 64);
        const uint32_t excl = incl - cnt;
        if (total >= LOW_ROWWISE_MIN * (uint32_t)__builtin_popcountll(c)) {
            // A word of heavy rows (the rows are numbered by weight, so a word's rows are alike).  Lanes along B: the
            // accumulator's row is A's, so a group's adds fall in one LDS row, bank by bank.  A group's row with more than
            // G entries in B takes a second trip (G follows the average, not the longest: one long row in 64 would halve
            // the lanes in use for all of them).
            uint32_t nb_sum = nbb;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) nb_sum += (uint32_t)__shfl_xor((int)nb_sum, d, 64);
            const uint32_t nb_avg = (nb_sum + (uint32_t)__builtin_popcountll(c) - 1) / (uint32_t)__builtin_popcountll(c);
            const int lg = nb_avg <= 8 ? 3 : nb_avg <= 16 ? 4 : nb_avg <= 32 ? 5 : 6;
            const int G = 1 << lg, per = 64 >> lg;       // lanes per group, rows per round
            const int g = lane >> lg, idx = lane & (G - 1);
            // rows round * per + g of the word; a round's operands (four shuffles, the lane's B entry, the first four A
            // entries: unconditional loads from valid addresses, used or not) are fetched while the round before is worked
            auto fetch = [&](int round) {
                LowRound r;
                const int rl = round * per + g;
                r.ra0 = __shfl(a0, rl, 64), r.rna = __shfl(na, rl, 64), r.rb0 = __shfl(b0, rl, 64), r.rnb = __shfl(nbb, rl, 64);
                r.eb = entries[r.rb0 + min((uint32_t)idx, max(r.rnb, 1u) - 1u)];
                return r;
            };
            LowRound cur = fetch(0);
            for (int round = 0; round < G; ++round) {
                LowRound nxt = cur;
                if (round + 1 < G) nxt = fetch(round + 1);
                if ((uint32_t)idx < cur.rnb) {
                    char *col = (char *)acc + (cur.eb.x >> LOW_COL_SHIFT);
                    if (diagonal) low_walk_row<true>(entries + cur.ra0, cur.rna, cur.eb, col);  // (uniform)
                    else low_walk_row<false>(entries + cur.ra0, cur.rna, cur.eb, col);
                }
                for (uint32_t bk = (uint32_t)idx + G; bk < cur.rnb; bk += G) {  // (a second trip: the rows longer than G)
                    const uint2 eb = entries[cur.rb0 + bk];
                    char *col = (char *)acc + (eb.x >> LOW_COL_SHIFT);
                    if (diagonal) low_walk_row<true>(entries + cur.ra0, cur.rna, eb, col);
                    else low_walk_row<false>(entries + cur.ra0, cur.rna, eb, col);
                }
                cur = nxt;
            }
            continue;
        }
        for (uint32_t k0 = 0; k0 < total; k0 += 64) {
            const uint32_t k = k0 + (uint32_t)lane;
            int L = 0;  // the lane whose row holds update k: the number of lanes with incl <= k
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
                const uint32_t v = __shfl(incl, L + step - 1, 64);
                if (v <= k) L += step;
            }
            L = min(L, 63);  // (lanes past the word's last update)
            const uint32_t local = k - __shfl(excl, L, 64);
            const uint32_t ra0 = __shfl(a0, L, 64), rb0 = __shfl(b0, L, 64), rnb = __shfl(nbb, L, 64);
            if (k < total) {
                // local / rnb without the integer division's 25 instructions: the quotient is below 128 (a block holds at
                // most LOW_TILE <= 128 entries of a row), (local + 0.5) / rnb lies at least 0.5 / 128 inside (q, q + 1), and the
                // reciprocal and the product are off by less than 1e-4 of it
                const uint32_t qa_i = (uint32_t)(((float)local + 0.5f) * __frcp_rn((float)rnb));
                const uint32_t a = ra0 + qa_i, b = rb0 + (local - qa_i * rnb);
                const uint2 ea = entries[a], eb = entries[b];
                if (!diagonal || eb.x < ea.x)  // (a diagonal tile: each pair once)
                    atomicAdd((uint32_t *)((char *)acc + (ea.x | (eb.x >> LOW_COL_SHIFT))), min(ea.y, eb.y));
            }
        }
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

// ff_schedule.cpp -- host-side work schedules of the pair kernels (see ff_schedule.hpp).
#include "ff_host.hpp"
#include "ff_schedule.hpp"

#include <algorithm>
#include <functional>
#include <cstdlib>
#include <cstring>

#include "frackyfrac_amd.h"

namespace ff {
namespace sched {

// Pair tiles of the shard, row-block major (consecutive tiles share their rows).  tj is
// the full tile width; with allow_narrow the last tile of a row block is half as wide
// when at most tj/2 of its columns lie below the diagonal.
void build_tiles(int64_t N, int64_t rb, int64_t re, int ti, int tj, bool allow_narrow, std::vector<Tile> *tiles)
{
    tiles->clear();
    if (re <= rb) return;
    for (int64_t i0 = rb / ti * ti; i0 < re; i0 += ti) {
        const int64_t w = std::min<int64_t>(std::min<int64_t>(i0 + ti, re) - 1, N);  // valid columns: j < w
        for (int64_t j0 = 0; j0 < w; j0 += tj) {
            const bool narrow = allow_narrow && (w - j0) <= tj / 2;
            tiles->push_back({(int32_t)i0, (int32_t)j0, narrow ? 1 : 0});
        }
    }
}

void build_xu_tiles(int64_t N, int64_t rb, int64_t re, int jmax, std::vector<XUTile> *tiles)
{
    tiles->clear();
    if (re <= rb) return;
    for (int64_t i0 = rb / XU_TILE_H * XU_TILE_H; i0 < re; i0 += XU_TILE_H) {
        const int64_t w = std::min<int64_t>(std::min<int64_t>(i0 + XU_TILE_H, re) - 1, N);  // valid columns: j < w
        for (int64_t j0 = 0; j0 < w;) {
            int g = jmax;
            while (g > 1 && w - j0 <= 64 * g / 2) g /= 2;  // the narrowest tile that covers what is left of the row block
            tiles->push_back({(int32_t)i0, (int32_t)j0, g, 0});
            j0 += 64 * g;
        }
    }
    std::stable_sort(tiles->begin(), tiles->end(), [](const XUTile &a, const XUTile &b) { return a.jn > b.jn; });
}

// FF_XCD_SLICES: number of branch slices the main rounds pin to groups of XCDs (2, 4 or 8 of
// an MI355X's 8 XCDs; 0 = rounds that ignore the XCDs).  Default 2: measured at C3, fabric
// traffic 6.6 -> 2.2 GB per launch at the same kernel time (4 and 8 move no fewer bytes and
// cost 0.2 % and 1.5 %: more, shorter items).
bool xcd_slices_forced()
{
    const auto e = ff::tuning("FF_XCD_SLICES");
    return e && !e->empty();
}

int xcd_slices()
{
    const auto e = ff::tuning("FF_XCD_SLICES");
    const int v = e && !e->empty() ? atoi(e->c_str()) : 2;
    return v < 0 ? 0 : v;
}

// FF_WAVES_PER_WG: 12 = pair_sad_kernel12, the pair kernel with half the vector buffers and so
// three waves per SIMD; 8 = pair_sad_kernel with two.
int waves_per_wg()
{
    const auto e = ff::tuning("FF_WAVES_PER_WG");
    const int v = e && !e->empty() ? atoi(e->c_str()) : WAVES_PER_WG;
    return v == L_WAVES_PER_WG ? L_WAVES_PER_WG : WAVES_PER_WG;
}

// Balances tiles over U persistent waves.
//
//  * Rounds (full-width tiles only).  Each tile of a round is cut into S equal branch ranges and `pr` <= U/S tiles are
//    handed out per round, one range per wave.  All waves of a round then sweep the branches in step on S fronts, so
//    the rows they read are shared through L2 (each XCD's 256 waves read the same few rows; measured: without this
//    alignment 88 % of the loads miss L2).  S is chosen by estimated makespan (below).  What the first level's whole
//    rounds leave over -- fewer than pr tiles -- gets rounds of its own with its own, deeper split (second level),
//    as long as that is estimated to beat cutting it stream-K style.
//  * Remainder.  What is left then, and all half-width tiles, is cut stream-K style into ranges of cost -- not equal
//    ranges: a wave that the rounds left idle (a round rarely has work for every wave) takes remainder first, and
//    the shares are such that every wave ends at the same time (a remainder row costs 1.30 main-round rows).
//
// Ranges that share a tile store into planes of their own (rounds, when there are planes enough) or add their partial
// sums atomically; the sums are integers, so the result does not depend on the order.
//
// The makespan estimate (unit: one full tile on one wave), fitted to measured kernel times over sample counts and
// splits (round 3, profiles/r03_sched_split_sweep.txt: within 3 % of the measurement): a round takes 1/S plus about
// 120 rows' worth of start-up and epilogue per item; stream-K remainder runs 1.30x slower per term (its waves are not
// on common rows, its sums meet in atomics).  Round 2's estimate (48 rows, 1.15) and its rule "XCD-sliced rounds
// whenever they apply" left shapes between the round sizes up to 13 % behind (5,632 samples: 0.75 of the roofline
// where 4,096 and 8,192 reach 0.86; 0.82 now -- tools/shape_sweep.py, profiles/r03_shape_sweep_*.txt).
void build_schedule(const std::vector<Tile> &all_tiles, int64_t rows, int U,
                    std::vector<Item> *items, std::vector<int32_t> *item_ptr, double *elements, int xcds, int wpw,
                    int max_planes, bool prefer_sliced)
{
    constexpr double ITEM_ROWS = 120.0, REM_COST = 1.30;
    std::vector<Tile> wide, rest;
    for (const Tile &t : all_tiles) (t.narrow ? rest : wide).push_back(t);
    // Column-segment major: the 8 consecutive tiles a workgroup gets then share their 256 columns
    // (one vector row of 1 KiB per branch for all 8 waves, through L1) and differ in their 32 rows
    // (8 x 128 B of scalar operands): 2 KiB per workgroup and branch instead of 8 KiB + 128 B with
    // row-block major order.  Measured: fabric traffic 280 -> 101 GB per launch at C4, 2.30 -> 2.09
    // at C3, kernel times unchanged.
    std::stable_sort(wide.begin(), wide.end(), [](const Tile &a, const Tile &b) { return a.j0 < b.j0; });
    const int64_t T = (int64_t)wide.size();
    const std::vector<Tile> narrow_tiles = rest;  // (an attempt appends the full tiles its rounds leave over to its own copy)
    std::vector<std::vector<Item>> per;            // the schedule being built by the current attempt
    auto push = [&](int u, const Tile &t, int64_t k0, int64_t k1) {
        if (k1 <= k0) return;
        Item it{};
        it.i0 = t.i0;
        it.j0 = t.j0;
        it.k0 = (int32_t)k0;
        it.k1 = (int32_t)k1;
        it.flags = ((k0 == 0 && k1 == rows) ? 0u : 1u) | (t.narrow ? 4u : 0u);
        per[(size_t)u].push_back(it);
    };
    // One attempt: first-level rounds of the given kind (x > 0: XCD-sliced with x slices; else plain rounds with split
    // S0 and pr0 tiles per round), second-level rounds, levelled remainder.  Returns the estimated makespan in cost
    // units (the load of the busiest wave: 2 per main-round row, 2 * ITEM_ROWS per item, REM_COST per remainder unit).
    const int64_t n_wg = U / wpw;
    auto attempt = [&](int x, int64_t S0, int64_t pr0) -> double {
        per.assign((size_t)U, std::vector<Item>());
        rest = narrow_tiles;
        double makespan = 0;
        int64_t done = 0;  // full-width tiles scheduled in rounds
        const int64_t max_split = std::max<int64_t>(1, rows / (8 * KSTEP));  // ranges of >= 64 rows
        // best split for `left` tiles (the narrow tiles and what these rounds leave go to the remainder)
        auto choose = [&](int64_t left, int64_t *S_out, int64_t *pr_out) {
            double best = 1e300;
            *S_out = 1;
            *pr_out = 0;
            for (int64_t cand = 1; left > 0 && cand <= std::min<int64_t>(256, max_split); ++cand) {
                const int64_t pr = std::min<int64_t>(U / cand, left) / wpw * wpw;  // a multiple of the workgroup size: the waves
                if (pr <= 0) continue;                                            // of a workgroup hold equally long items
                const int64_t rounds = left / pr, rem = left - rounds * pr;
                const double est = (double)rounds / (double)cand + (double)rounds * ITEM_ROWS / (double)rows +
                                   REM_COST * ((double)rem + 0.5 * (double)rest.size()) / (double)U;
                if (est < best - 1e-9) {
                    best = est;
                    *S_out = cand;
                    *pr_out = pr;
                }
            }
            return best;
        };
        // rounds of `pr` tiles in S ranges each, from tile `done` on; `level`: waves barrier together only on items all
        // eight of a workgroup hold at the same position of their lists (pr is a multiple of the workgroup size)
        int64_t planes_used = 1;  // accumulator planes the first level's ranges own (the finish kernel reads every plane of every slot)
        auto plain_rounds = [&](int64_t S, int64_t pr, int64_t rounds, bool first_level) {
            // (the finish kernel reads every plane of every slot, about 8,192 slots a tile: a second level adds planes
            // of its own while reading them stays under a quarter of a gigabyte -- 2,560 samples with a twelve-fold
            // second level: 2.03 ms with planes, 2.11 adding atomically -- beyond that its ranges add atomically, like
            // the remainder's)
            const bool planes = S > 1 && S <= max_planes &&
                                (first_level || S <= planes_used || (double)all_tiles.size() * 8192.0 * 4.0 * (double)S <= 268435456.0);
            if (first_level && planes) planes_used = S;
            const int64_t part = round_up((rows + S - 1) / S, 2 * KSTEP);
            for (int64_t r = 0; r < rounds; ++r)
                for (int64_t q = 0; q < pr; ++q) {
                    const Tile &t = wide[(size_t)(done + r * pr + q)];
                    for (int64_t sidx = 0; sidx < S; ++sidx) {
                        // the S ranges of a tile go to waves pr apart: neighbouring waves keep
                        // neighbouring tiles (same columns -> one shared vector row per branch)
                        const int u = (int)(sidx * pr + q);
                        const size_t before = per[(size_t)u].size();
                        push(u, t, std::min(rows, sidx * part), std::min(rows, (sidx + 1) * part));
                        if (per[(size_t)u].size() > before) {
                            uint32_t &fl = per[(size_t)u].back().flags;
                            fl |= 2u;
                            if (planes) fl = (fl & ~1u) | ((uint32_t)sidx << 3);
                        }
                    }
                }
            done += rounds * pr;
        };
        if (x > 0) {
            const int xcds = x;
            // Branch slices pinned to XCDs.  Workgroup g runs on XCD g mod 8 (round-robin
            // dispatch); every tile is cut into `xcds` (2, 4 or 8) equal branch ranges and range x
            // always goes to a workgroup of XCD group x (8/xcds XCDs), so each XCD's L2 only ever
            // sees 1/xcds of the staged rows and all its waves sweep that slice together.  A round
            // = 8 consecutive tiles per workgroup of ONE group's share, i.e. n_wg/xcds * 8 tiles.
            const int64_t gsz = 8 / xcds;  // XCDs per group
            const int64_t pr = n_wg / xcds * wpw;
            const int64_t part = round_up((rows + xcds - 1) / xcds, 2 * KSTEP);
            const int64_t rounds = T / pr;
            for (int64_t r = 0; r < rounds; ++r)
                for (int64_t q = 0; q < pr; ++q) {
                    const Tile &t = wide[(size_t)(r * pr + q)];
                    const int64_t m = q / wpw, w = q % wpw;
                    for (int64_t x = 0; x < xcds; ++x) {
                        const int64_t wg = 8 * (m / gsz) + x * gsz + (m % gsz);  // m-th workgroup of group x
                        const int u = (int)(wg * wpw + w);
                        const size_t before = per[(size_t)u].size();
                        push(u, t, std::min(rows, x * part), std::min(rows, (x + 1) * part));
                        if (per[(size_t)u].size() > before) {
                            uint32_t &fl = per[(size_t)u].back().flags;
                            fl |= 2u;
                            // one plane of accumulators per range: each stores its sums plainly
                            // into its own plane instead of adding atomically
                            if (xcds <= max_planes && part < rows) {
                                fl = (fl & ~1u) | ((uint32_t)x << 3);
                                planes_used = xcds;
                            }
                        }
                    }
                }
            done = rounds * pr;
        } else if (T > 0 && pr0 > 0) {
            plain_rounds(S0, pr0, T / pr0, true);
        }
        // Second level: the tiles the whole rounds left over, in rounds with a split of their own, if that is
        // estimated to finish sooner than cutting them stream-K style with the half-width tiles.
        if (T - done >= wpw) {
            const int64_t left = T - done;
            int64_t S2 = 1, pr2 = 0;
            const double with_rounds = choose(left, &S2, &pr2);
            const double stream_only = REM_COST * ((double)left + 0.5 * (double)rest.size()) / (double)U;
            // (its ranges own planes only if the first level has as many: a deeper split adds atomically, like the remainder)
            if (pr2 > 0 && with_rounds < stream_only - 1e-9) plain_rounds(S2, pr2, left / pr2, false);
        }
        rest.insert(rest.begin(), wide.begin() + done, wide.end());  // leftover full tiles first
        if (!rest.empty()) {
            // cost units: a full-width row = 2, a half-width row = 1
            std::vector<int64_t> start(rest.size() + 1, 0);
            for (size_t t = 0; t < rest.size(); ++t) start[t + 1] = start[t] + rows * (rest[t].narrow ? 1 : 2);
            const int64_t total = start.back();
            // Shares: wave u holds load[u] units of round work; with the level L at which every wave ends, it takes
            // (L - load[u]) / REM_COST units of remainder (none if it is above the level).  L by bisection; shares
            // are whole multiples of 32 units (16 wide or 32 narrow rows), what rounding leaves goes round-robin.
            std::vector<double> load((size_t)U, 0.0);
            double lo = 0, hi = 0;
            for (int u = 0; u < U; ++u) {
                for (const Item &it : per[(size_t)u]) load[(size_t)u] += 2.0 * (double)(it.k1 - it.k0) + 2.0 * ITEM_ROWS;
                hi = std::max(hi, load[(size_t)u]);
            }
            hi += REM_COST * (double)total;
            for (int iter = 0; iter < 60; ++iter) {
                const double L = 0.5 * (lo + hi);
                double got = 0;
                for (int u = 0; u < U; ++u) got += std::max(0.0, L - load[(size_t)u]) / REM_COST;
                (got >= (double)total ? hi : lo) = L;
            }
            makespan = hi;
            const int64_t q32 = 4 * KSTEP;
            std::vector<int64_t> share((size_t)U, 0);
            int64_t sum = 0;
            for (int u = 0; u < U; ++u) {
                share[(size_t)u] = (int64_t)(std::max(0.0, hi - load[(size_t)u]) / REM_COST / (double)q32) * q32;
                sum += share[(size_t)u];
            }
            // (rounded down above: a few 32-unit steps are missing; they go to waves at or below the level)
            for (int u = 0, skipped = 0; sum < total; u = (u + 1) % U)
                if (share[(size_t)u] > 0 || load[(size_t)u] <= lo || skipped >= U) {
                    share[(size_t)u] += q32;
                    sum += q32;
                    skipped = 0;
                } else {
                    ++skipped;
                }
            size_t t = 0;
            int64_t a = 0;
            for (int u = 0; u < U && a < total; ++u) {
                const int64_t b = std::min(total, a + share[(size_t)u]);
                while (a < b) {
                    while (start[t + 1] <= a) ++t;
                    const int64_t unit = rest[t].narrow ? 1 : 2;
                    const int64_t k0 = (a - start[t]) / unit;
                    const int64_t k1 = std::min<int64_t>(rows, (std::min(b, start[t + 1]) - start[t]) / unit);
                    push(u, rest[t], k0, k1);
                    a = start[t] + k1 * unit;
                }
            }
        }
        for (int u = 0; u < U; ++u) {  // (no remainder: the busiest wave of the rounds)
            double l = 0;
            for (const Item &it : per[(size_t)u])
                l += ((it.flags & 2u) ? 2.0 : REM_COST * ((it.flags & 4u) ? 1.0 : 2.0)) * (double)(it.k1 - it.k0) + 2.0 * ITEM_ROWS;
            makespan = std::max(makespan, l);
        }
        return makespan;
    };
    if (rows > 0) {
        // Candidates for the first level: the plain rounds with the best estimated split, and the XCD-sliced rounds --
        // rounds with S = 2, 4 or 8 whose slices are pinned to groups of XCDs, so that an XCD's L2 sees one front of
        // the sweep instead of all of them.  Every candidate is built, its makespan estimated from the loads of its
        // waves, and the best is kept; an XCD-sliced candidate is preferred unless the plain rounds are estimated more
        // than 1 % faster -- 3 % for a shard that does not begin at row 0 (prefer_sliced): slicing costs about half a
        // percent where it is not needed (4,096 samples: 4.92 -> 4.94 ms) and is worth 7-12 % on some later row
        // shards of a multi-GPU run, whose plain rounds run far slower on some XCDs than on others (round 3: shards
        // 3 and 4 of 8 of 11,584 samples, 5.67 / 5.23 ms with plain thirds, 5.02 / 4.86 with four slices, the other
        // six shards within 1 % either way; tools/experiments/xcd_variants.py -- the estimate cannot tell those
        // shards from the others, so the preference covers every later shard; whole problems and first shards --
        // triangles -- showed no such outlier at any of fifteen sizes).
        // FF_XCD_SLICES = 0 forbids the sliced rounds, 2 / 4 / 8 asks for that slicing wherever it applies.
        struct Cand {
            int x;
            int64_t S, pr;
            double est;
        };
        std::vector<Cand> cands;
        {
            per.assign((size_t)U, std::vector<Item>());
            const int64_t max_split = std::max<int64_t>(1, rows / (8 * KSTEP));
            double best = 1e300;
            Cand g{0, 1, 0, 0};
            for (int64_t cand = 1; T > 0 && cand <= std::min<int64_t>(256, max_split); ++cand) {
                const int64_t pr = std::min<int64_t>(U / cand, T) / wpw * wpw;
                if (pr <= 0) continue;
                const int64_t rounds = T / pr, rem = T - rounds * pr;
                const double est = (double)rounds / (double)cand + (double)rounds * ITEM_ROWS / (double)rows +
                                   REM_COST * ((double)rem + 0.5 * (double)narrow_tiles.size()) / (double)U;
                if (est < best - 1e-9) {
                    best = est;
                    g.S = cand;
                    g.pr = pr;
                }
            }
            cands.push_back(g);
        }
        const bool forced = xcd_slices_forced();
        for (int x : {2, 4, 8}) {
            if (xcds <= 0 || (forced && x != xcds)) continue;
            if (n_wg % 8 == 0 && T >= n_wg / x * wpw && rows >= (int64_t)x * 8 * KSTEP) cands.push_back({x, 0, 0, 0});
        }
        if (forced && cands.size() > 1) cands.erase(cands.begin());  // (asked for by name: the sliced rounds, where they apply)
        size_t pick = 0;
        double pick_score = 1e300;
        for (size_t c = 0; c < cands.size(); ++c) {
            cands[c].est = attempt(cands[c].x, cands[c].S, cands[c].pr);
            const double score = cands[c].est * (cands[c].x > 0 ? (prefer_sliced ? 0.97 : 0.99) : 1.0);
            if (score < pick_score - 1e-9) {
                pick_score = score;
                pick = c;
            }
        }
        if (pick + 1 != cands.size() || cands.empty()) attempt(cands[pick].x, cands[pick].S, cands[pick].pr);  // (rebuild the one kept)
    } else {
        per.assign((size_t)U, std::vector<Item>());
    }
    items->clear();
    item_ptr->assign((size_t)U + 1, 0);
    double el = 0;
    for (int u = 0; u < U; ++u) {
        for (const Item &it : per[(size_t)u]) {
            items->push_back(it);
            el += (double)(it.k1 - it.k0) * TILE_I * ((it.flags & 4u) ? TILE_J / 2 : TILE_J);
        }
        (*item_ptr)[(size_t)u + 1] = (int32_t)items->size();
    }
    *elements = el;
}


int64_t build_mfma_schedule(int64_t N, int64_t row_begin, int64_t row_end, int64_t slabs, int digits, int G,
                            std::vector<MItem> *items, std::vector<int32_t> *item_ptr,
                            std::vector<int32_t> *partial_tiles, std::vector<int32_t> *partial_ptr,
                            int64_t max_private_tiles, int64_t duo_from_quad)
{
    // 256 x 128 tiles of the shard's part of the lower triangle, ordered so that 32
    // consecutive tiles form a compact block of 4 x 8 tiles (1024 x 1024 samples): the
    // 32 workgroups of one XCD then share their operand rows through that XCD's L2.
    struct MT {
        int32_t i0, j0;
    };
    std::vector<MT> tiles;
    for (int64_t i0 = row_begin / M_TILE_I * M_TILE_I; i0 < row_end; i0 += M_TILE_I) {
        const int64_t w = std::min<int64_t>(std::min<int64_t>(i0 + M_TILE_I, row_end) - 1, N);
        for (int64_t j0 = 0; j0 < w; j0 += M_TILE_J) tiles.push_back({(int32_t)i0, (int32_t)j0});
    }
    std::sort(tiles.begin(), tiles.end(), [](const MT &x, const MT &y) {
        const int64_t bx = ((int64_t)(x.i0 / (4 * M_TILE_I)) << 32) | (uint32_t)(x.j0 / (8 * M_TILE_J));
        const int64_t by = ((int64_t)(y.i0 / (4 * M_TILE_I)) << 32) | (uint32_t)(y.j0 / (8 * M_TILE_J));
        if (bx != by) return bx < by;
        if (x.i0 != y.i0) return x.i0 < y.i0;
        return x.j0 < y.j0;
    });
    // Units = (digit group, tile).  Main rounds: whole units, one per workgroup, workgroup
    // g of XCD g % 8 taking unit 32 * (g % 8) + g / 8 of the round, so all workgroups
    // sweep the branches in step.  Remainder (< G units): cut stream-K style into G equal
    // slab ranges so that every workgroup ends at the same time.
    const int groups = (digits + M_ND - 1) / M_ND;
    // The branch sweep is cut at multiples of M_QUAD_SLABS slabs (the kernel's loop works in quads; the
    // engine pads the staged rows to whole quads): from here on `slabs` counts quads, and make_item
    // turns quads back into branches.
    const int64_t true_slabs = slabs;
    slabs = (slabs + M_QUAD_SLABS - 1) / M_QUAD_SLABS;
    // A graded sweep's quads do not cost the same: the cuts below are made on an axis of COST -- G_TRI per quad
    // with three planes, G_DUO per quad with two, G_UNIT times the unit in which a quad of a base-128 two-plane
    // sweep costs 4 (measured at C3's shape, tools/mfma_digits.py: 0.346 and 0.2625 ms against 0.244) -- and
    // real_quad() turns a position on it back into the quad boundary at or after it.  Else the axis is quads.
    constexpr int64_t G_UNIT = 2, G_TRI = 12, G_DUO = 9;
    const bool graded = duo_from_quad >= 0;
    const int64_t real_quads = slabs, tri_quads = graded ? std::min(duo_from_quad, slabs) : 0;
    if (graded) slabs = G_TRI * tri_quads + G_DUO * (real_quads - tri_quads);
    auto real_quad = [&](int64_t s) {
        if (!graded) return s;
        const int64_t q = s <= G_TRI * tri_quads ? (s + G_TRI - 1) / G_TRI : tri_quads + (s - G_TRI * tri_quads + G_DUO - 1) / G_DUO;
        return std::min(q, real_quads);
    };
    auto make_item = [&](int64_t unit, int64_t s0, int64_t s1) {
        const int64_t grp = unit / (int64_t)tiles.size(), t = unit % (int64_t)tiles.size();
        MItem itm{};
        itm.i0 = tiles[(size_t)t].i0;
        itm.j0 = tiles[(size_t)t].j0;
        itm.k0 = (int32_t)(real_quad(s0) * M_QUAD_SLABS * M_KSLAB);
        itm.k1 = (int32_t)(std::min(true_slabs, real_quad(s1) * M_QUAD_SLABS) * M_KSLAB);
        itm.d0 = (int32_t)(M_ND * grp);
        itm.nd = std::min(M_ND, digits - M_ND * (int)grp);
        itm.first = (grp == 0 && s0 == 0) ? 1 : 0;
        return itm;
    };
    std::vector<std::vector<MItem>> per((size_t)G);
    std::vector<std::vector<int32_t>> per_tile((size_t)G);  // tile index of every item (for the ordinals below)
    const int64_t T = (int64_t)tiles.size();
    // Whole rounds are dealt out digit group by digit group: a sweep of a group with a single digit
    // plane costs about three quarters of a two-plane one (its loop is bound by the vector work,
    // not the MFMAs), and a round that mixed both kinds would leave some workgroups a quarter of a
    // unit behind.  What does not fill a round of its group goes to the remainder, which is cut by cost.
    const int64_t rounds_per_group = T / G;
    const int64_t rounds = rounds_per_group * groups;
    std::vector<int64_t> rem;  // units of the remainder, in (group, tile) order
    for (int grp = 0; grp < groups; ++grp)
        for (int64_t t = rounds_per_group * G; t < T; ++t) rem.push_back((int64_t)grp * T + t);
    const int64_t rem_units = (int64_t)rem.size();
    const int per_xcd = std::max(1, G / 8);
    // Device-scope atomics across XCDs are performed at the memory side and are slow (a problem
    // too small for even one round, all remainder, spent two thirds of its kernel in them).  Ways
    // out of the kernel, by MItem.pad:  p > 0 -- the item owns private partial tile p - 1, which a
    // reduce kernel sums with the tile's others;  -1 -- the tile's only item stores plainly into
    // num[];  0 -- adds into num[] atomically.
    //   all_private: every item owns a tile (any number of digit groups; the whole schedule needs
    //     no more than max_private_tiles of them -- at most G + rem_units ranges in the remainder);
    //   else, one digit group: a main-round item is its tile's only one (-1), the remainder's
    //     ranges own tiles;  else (several groups, over the budget): atomics throughout.
    const bool can_list = partial_tiles && partial_ptr;
    const bool all_private = can_list && rounds * G + (rem_units ? G + rem_units : 0) <= max_private_tiles;
    const bool private_remainder = can_list && (all_private || groups == 1);
    for (int grp = 0; grp < groups; ++grp)
        for (int64_t r = 0; r < rounds_per_group; ++r)
            for (int g = 0; g < G; ++g) {
                const int64_t local = (G % 8 == 0) ? (int64_t)(g % 8) * per_xcd + g / 8 : g;
                const int64_t t = r * G + local;
                MItem itm = make_item((int64_t)grp * T + t, 0, slabs);
                if (all_private) itm.pad = 1;  // (a private tile: the ordinal comes below)
                else if (can_list && groups == 1) itm.pad = -1;
                per[(size_t)g].push_back(itm);
                per_tile[(size_t)g].push_back((int32_t)t);
            }
    if (partial_tiles) partial_tiles->clear();
    if (partial_ptr) partial_ptr->clear();
    auto add_range = [&](int g, int64_t unit, int64_t s0, int64_t s1) {
        MItem itm = make_item(unit, s0, s1);
        if (itm.k0 >= itm.k1) return;  // (graded: a sliver of cost inside one quad, which its neighbour carries)
        if (private_remainder) itm.pad = 1;
        per[(size_t)g].push_back(itm);
        per_tile[(size_t)g].push_back((int32_t)(unit % T));
    };
    if (rem_units > 0) {
        // Cost of a quad of slabs of unit u: 4 with two digit planes, 3 with one.  Two ways to cut the
        // remainder over G workgroups.  Stream-K: equal shares of cost; a share that straddles a unit
        // boundary becomes two items, and an item costs its workgroup a prologue and an epilogue --
        // about M_ITEM_OVERHEAD_SLABS slabs' worth -- on top of its slabs.  Aligned: every unit gets
        // workgroups in proportion to its cost, which cut it evenly; nothing straddles, but ranges
        // come out unequal where the proportions do not divide.  The cut whose slowest workgroup
        // finishes first is taken.
        auto quad_cost = [&](int64_t unit) {
            if (graded) return (int64_t)1;  // (the axis is cost already)
            return (int64_t)(std::min(M_ND, digits - M_ND * (int)(unit / T)) > 1 ? 4 : 3);
        };
        std::vector<int64_t> start((size_t)rem_units + 1, 0);  // cost position at which unit k begins
        for (int64_t k = 0; k < rem_units; ++k) start[(size_t)k + 1] = start[(size_t)k] + slabs * quad_cost(rem[(size_t)k]);
        const int64_t total = start[(size_t)rem_units];
        const int64_t share = std::max<int64_t>(1, (total + G - 1) / G);
        // the quad of unit k at which cost position x falls (both neighbours of a boundary use this)
        auto quad_at = [&](int64_t k, int64_t x) {
            const int64_t c = quad_cost(rem[(size_t)k]);
            return std::min(slabs, std::max<int64_t>(0, (x - start[(size_t)k] + c - 1) / c));
        };
        auto stream_ranges = [&](int g, const std::function<void(int64_t, int64_t, int64_t)> &emit) {
            const int64_t a2 = (int64_t)g * share, b2 = std::min(total, a2 + share);
            for (int64_t k = 0; k < rem_units && a2 < b2; ++k) {
                if (start[(size_t)k + 1] <= a2 || start[(size_t)k] >= b2) continue;
                const int64_t s0 = quad_at(k, a2), s1 = quad_at(k, b2);
                if (s0 < s1) emit(k, s0, s1);
            }
        };
        const int64_t overhead = M_ITEM_OVERHEAD_SLABS / M_QUAD_SLABS * 4 * (graded ? G_UNIT : 1);
        int64_t cost_stream = 0;
        for (int g = 0; g < G; ++g) {
            int64_t c = 0, n_items = 0;
            stream_ranges(g, [&](int64_t k, int64_t s0, int64_t s1) {
                c += (s1 - s0) * quad_cost(rem[(size_t)k]);
                ++n_items;
            });
            cost_stream = std::max(cost_stream, c + std::max<int64_t>(0, n_items - 1) * overhead);
        }
        // aligned: w[k] workgroups for unit k, at least one, the spare ones to whoever carries most
        std::vector<int64_t> w((size_t)rem_units, 1);
        int64_t cost_aligned = INT64_MAX;
        if (rem_units <= G) {
            int64_t used = 0;
            for (int64_t k = 0; k < rem_units; ++k) {
                w[(size_t)k] = std::max<int64_t>(1, (int64_t)G * (start[(size_t)k + 1] - start[(size_t)k]) / total);
                used += w[(size_t)k];
            }
            auto load = [&](int64_t k) { return (slabs + w[(size_t)k] - 1) / w[(size_t)k] * quad_cost(rem[(size_t)k]); };
            for (; used > G; --used) {  // (the "at least one" may have overdrawn: take from the lightest)
                int64_t best = -1;
                for (int64_t k = 0; k < rem_units; ++k)
                    if (w[(size_t)k] > 1 && (best < 0 || load(k) < load(best))) best = k;
                if (best < 0) break;
                --w[(size_t)best];
            }
            for (; used < G; ++used) {
                int64_t best = 0;
                for (int64_t k = 1; k < rem_units; ++k)
                    if (load(k) > load(best)) best = k;
                ++w[(size_t)best];
            }
            if (used == G) {
                cost_aligned = 0;
                for (int64_t k = 0; k < rem_units; ++k) cost_aligned = std::max(cost_aligned, load(k));
            }
        }
        if (cost_aligned <= cost_stream) {
            int g = 0;
            for (int64_t k = 0; k < rem_units; ++k)
                for (int64_t q = 0; q < w[(size_t)k]; ++q, ++g) {
                    const int64_t s0 = slabs * q / w[(size_t)k], s1 = slabs * (q + 1) / w[(size_t)k];
                    if (s0 < s1) add_range(g, rem[(size_t)k], s0, s1);
                }
        } else {
            for (int g = 0; g < G; ++g)
                stream_ranges(g, [&](int64_t k, int64_t s0, int64_t s1) { add_range(g, rem[(size_t)k], s0, s1); });
        }
    }
    // Ordinals of the private tiles, tile by tile: partial_tiles lists the tiles that own any,
    // partial_ptr their first ordinal (+ the total); a tile's partials are contiguous.
    if (can_list) {
        std::vector<int32_t> count(tiles.size() + 1, 0);
        for (int g = 0; g < G; ++g)
            for (size_t k = 0; k < per[(size_t)g].size(); ++k)
                if (per[(size_t)g][k].pad > 0) ++count[(size_t)per_tile[(size_t)g][k] + 1];
        std::vector<int32_t> next(tiles.size(), 0);
        int32_t total_partials = 0;
        for (size_t t = 0; t < tiles.size(); ++t) {
            if (count[t + 1] == 0) continue;
            partial_ptr->push_back(total_partials);
            partial_tiles->push_back(tiles[t].i0);
            partial_tiles->push_back(tiles[t].j0);
            next[t] = total_partials;
            total_partials += count[t + 1];
        }
        if (!partial_ptr->empty()) partial_ptr->push_back(total_partials);
        for (int g = 0; g < G; ++g)
            for (size_t k = 0; k < per[(size_t)g].size(); ++k)
                if (per[(size_t)g][k].pad > 0) per[(size_t)g][k].pad = ++next[(size_t)per_tile[(size_t)g][k]];
    }
    std::vector<MItem> &mi = *items;
    std::vector<int32_t> &mptr = *item_ptr;
    mi.clear();
    mptr.assign((size_t)G + 1, 0);
    for (int g = 0; g < G; ++g) {
        mi.insert(mi.end(), per[(size_t)g].begin(), per[(size_t)g].end());
        mptr[(size_t)g + 1] = (int32_t)mi.size();
    }

    return (int64_t)tiles.size();
}

}  // namespace sched
}  // namespace ff

// Diagnostics / tests: the schedule the engine would build for a shard of rows [row_begin,
// row_end) of n_samples samples with `rows` staged branch rows on a device with n_cu compute
// units.  kernel = FF_KERNEL_SAD_U32 or FF_KERNEL_MFMA_I8 (then `rows` is the slab count and
// n_digits applies).  Items come back as 8 int32 each: SAD {i0, j0, k0, k1, flags, 0, 0, 0},
// MFMA {i0, j0, k0, k1, d0, nd, first, 0}, EXACT_F64_UNW (tiles) {i0, j0, jn, 0, ...}; item_ptr has one entry per wave slot (SAD: 8 per
// CU) or workgroup (MFMA: one per CU) plus one.  Returns the number of items, or -needed if
// max_items is too small.
extern "C" int64_t ff_debug_schedule(int kernel, int64_t n_samples, int64_t rows, int64_t row_begin, int64_t row_end,
                                     int n_cu, int n_digits, int allow_narrow, int32_t *items_out, int64_t max_items,
                                     int32_t *item_ptr_out, int64_t *n_tiles_out)
{
    using namespace ff::sched;
    std::vector<int32_t> ptr;
    int64_t n = 0;
    if (kernel == FF_KERNEL_MFMA_I8) {
        std::vector<MItem> items;
        const int64_t nt = build_mfma_schedule(n_samples, row_begin, row_end, rows, n_digits, n_cu * M_WGS_PER_CU, &items, &ptr, nullptr, nullptr);
        if (n_tiles_out) *n_tiles_out = nt;
        n = (int64_t)items.size();
        if (n > max_items) return -n;
        if (n) memcpy(items_out, items.data(), sizeof(MItem) * (size_t)n);
    } else if (kernel == FF_KERNEL_EXACT_F64_UNW) {  // (n_digits: the widest tile in column groups; no item_ptr)
        std::vector<XUTile> tiles;
        build_xu_tiles(n_samples, row_begin, row_end, n_digits, &tiles);
        if (n_tiles_out) *n_tiles_out = (int64_t)tiles.size();
        n = (int64_t)tiles.size();
        if (n > max_items) return -n;
        for (int64_t q = 0; q < n; ++q) {
            int32_t *o = items_out + 8 * q;
            memset(o, 0, 8 * sizeof(int32_t));
            o[0] = tiles[(size_t)q].i0;
            o[1] = tiles[(size_t)q].j0;
            o[2] = tiles[(size_t)q].jn;
        }
        return n;
    } else {
        std::vector<Tile> tiles;
        build_tiles(n_samples, row_begin, row_end, TILE_I, TILE_J, allow_narrow != 0, &tiles);
        if (n_tiles_out) *n_tiles_out = (int64_t)tiles.size();
        std::vector<Item> items;
        double elements = 0;
        build_schedule(tiles, rows, n_cu * waves_per_wg(), &items, &ptr, &elements, xcd_slices(), waves_per_wg(), 0,
                       row_begin > 0);
        n = (int64_t)items.size();
        if (n > max_items) return -n;
        if (n) memcpy(items_out, items.data(), sizeof(Item) * (size_t)n);
    }
    memcpy(item_ptr_out, ptr.data(), sizeof(int32_t) * ptr.size());
    return n;
}

// Diagnostics / tests: the constants that tie the kernels' prefetch depths to the padding of the staged
// arrays (ff_schedule.hpp), and the allocation sizes ff_dev_stage.hip derives from them for R staged rows.
// out[0..23] = {TILE_I, TILE_J, KSTEP, SLACK_ROWS, SAD_ROWS_AHEAD, SPARSE_LIST_AHEAD, SPARSE_LIST_PAD, M_KSLAB,
//               M_QUAD_SLABS, M_PAIRS_IN_FLIGHT, M_PAD_SLABS, X_VALUES_PAD, sad_staged_rows(R), sad_alloc_rows(R),
//               mfma_staged_slabs(R), mfma_alloc_slabs(R),
//               XU_TILE_H, XU_SLAB, XU_JMAX, XU_LEN_STEP, xu_slabs(R), xu_alloc_slabs(R), xu_alloc_lengths(R), 0}
extern "C" void ff_debug_layout(int64_t R, int64_t *out)
{
    using namespace ff::sched;
    const int64_t v[24] = {TILE_I, TILE_J, KSTEP, SLACK_ROWS, SAD_ROWS_AHEAD, SPARSE_LIST_AHEAD, SPARSE_LIST_PAD, M_KSLAB,
                           M_QUAD_SLABS, M_PAIRS_IN_FLIGHT, M_PAD_SLABS, X_VALUES_PAD, sad_staged_rows(R), sad_alloc_rows(R),
                           mfma_staged_slabs(R), mfma_alloc_slabs(R),
                           XU_TILE_H, XU_SLAB, XU_JMAX, XU_LEN_STEP, xu_slabs(R), xu_alloc_slabs(R), xu_alloc_lengths(R), 0};
    memcpy(out, v, sizeof v);
}

// Diagnostics / tests: the three signed digit planes of pair_common_mfma_kernel's TRI sweep for n lengths
// (out[3 * q .. 3 * q + 2] = {d0, -d1, d2} of k[q]); returns TRI_KMAX.
extern "C" int64_t ff_debug_tri_digits(const uint32_t *k, int64_t n, int8_t *out)
{
    for (int64_t q = 0; q < n; ++q) ff::sched::tri_digits((int64_t)k[q], out + 3 * q);
    return ff::sched::TRI_KMAX;
}

// Diagnostics / tests: the schedule of a GRADED matrix-core sweep (build_mfma_schedule with duo_from_quad): items
// as ff_debug_schedule returns them for FF_KERNEL_MFMA_I8.
extern "C" int64_t ff_debug_graded_schedule(int64_t n_samples, int64_t slabs, int64_t row_begin, int64_t row_end, int n_cu,
                                            int64_t duo_from_quad, int32_t *items_out, int64_t max_items,
                                            int32_t *item_ptr_out, int64_t *n_tiles_out)
{
    using namespace ff::sched;
    std::vector<int32_t> ptr;
    std::vector<MItem> items;
    const int64_t nt = build_mfma_schedule(n_samples, row_begin, row_end, slabs, 2, n_cu * M_WGS_PER_CU, &items, &ptr, nullptr,
                                           nullptr, 0, duo_from_quad);
    if (n_tiles_out) *n_tiles_out = nt;
    const int64_t n = (int64_t)items.size();
    if (n > max_items) return -n;
    if (n) memcpy(items_out, items.data(), sizeof(MItem) * (size_t)n);
    memcpy(item_ptr_out, ptr.data(), sizeof(int32_t) * ptr.size());
    return n;
}

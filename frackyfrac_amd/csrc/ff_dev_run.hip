// ff_dev_run.hip -- the pair kernels and one pass of a plan (one of the three translation units of the device
// path: ff_plan.hpp).
//
// Replaces the per-pair merge walks unifracDistWeighted / unifracDistUnweighted
// (frcfrc/unifrac.go:144-205) and their driver unifracDists (unifrac.go:209-228).
// See DESIGN.md for the derivation; in short, with q_s(b) the staged value of
// sample s on branch b (0 where the sample has no flat node):
//
//   FIXED32  q_s(b) = floor(l_b * abnd_s(b) * 2^e + u_b)   (weighted)
//            q_s(b) = k_b * [present], k_b = floor(l_b * 2^e + u_b)   (unweighted)
//            U(i,j) = sum_b |q_i(b) - q_j(b)|,  W_s = sum_b q_s(b)   -- exact integers
//            weighted   d = U / (W_i + W_j)          U by v_sad_u32, one per term
//            unweighted d = U / (U + C), C = (W_i + W_j - U) / 2
//                       C = sum_b k_b [i present][j present] is a contraction: int8 MFMA
//   EXACT64  binary64 running sums over b ascending, with the reference's own
//            operations (no contraction), so every rounding is the reference's.
//
// Here: the work schedule of every kernel for a shard (rebuilt by ff_plan_set_shard), the refinement queue and the
// run-time audit of FIXED32, and plan_run_impl, which launches a pass.
#include "ff_plan.hpp"

namespace {

using namespace ff::sched;

#include "ff_kernels_pair_sad.hpp"
#include "ff_kernels_finish_pair.hpp"
#include "ff_kernels_mfma.hpp"
#include "ff_kernels_mfma_small.hpp"
#include "ff_kernels_finish.hpp"
#include "ff_kernels_exact_unw.hpp"
#include "ff_kernels_exact_w.hpp"
#include "ff_kernels_low.hpp"

}  // namespace

namespace ff {
namespace dev {

using namespace ff::sched;

// ---- The shard-dependent part of a plan: work schedule and accumulators --------------------
// (rebuilt by ff_plan_set_shard; the staged matrix does not depend on the shard)


int schedule_sad(ff_plan *pl, char *err, size_t errlen)
{
    ff_plan_info &inf = pl->info;
    const int64_t N = inf.n_samples, rows = inf.rows_padded, n_slots = inf.slot_end - inf.slot_begin;
    free_and_null(pl->d_items);
    free_and_null(pl->d_item_ptr);
    free_and_null(pl->d_num);
    free_and_null(pl->d_stamps);
    free_and_null(pl->d_low_tiles);
    free_and_null(pl->d_mlow);
    pl->n_low_tiles = 0;
    if (pl->split && n_slots > 0) {
        // the blocks of pairs of pair_low_kernel that hold a row of the shard: sample blocks bi x bj <= bi
        std::vector<LowTile> lt;
        for (int64_t bi = inf.row_begin / pl->low_tile; bi * pl->low_tile < std::min(inf.row_end, N); ++bi)
            for (int64_t bj = 0; bj <= bi; ++bj) lt.push_back({(int32_t)bi, (int32_t)bj});
        pl->n_low_tiles = (int)lt.size();
        FF_HIP(hipMalloc(&pl->d_low_tiles, sizeof(LowTile) * std::max<size_t>(lt.size(), 1)));
        if (!lt.empty()) FF_HIP(hipMemcpy(pl->d_low_tiles, lt.data(), sizeof(LowTile) * lt.size(), hipMemcpyHostToDevice));
        FF_ALLOC(pl->d_mlow, sizeof(uint32_t) * (size_t)n_slots, "the rare rows' sums");
    }
    std::vector<Tile> tiles;
    build_tiles(N, inf.row_begin, inf.row_end, TILE_I, TILE_J, true, &tiles);
    inf.n_tiles = (int64_t)tiles.size();
    // 8 waves per workgroup (two per SIMD) for pair_sad_kernel and the sparse-aware kernel, 12 (three per SIMD,
    // paid for with half the vector prefetch) for pair_sad_kernel12.  FF_WAVES_PER_WG = 8 / 12 forces one; otherwise
    // the 12-wave variant takes
    //   * a shard that BEGINS AT ROW 0 -- a whole problem, the first rank's shard: a triangle -- and holds more than
    //     2.75 tiles per workgroup (3,300 samples up on 256 CUs): tools/shape_sweep.py, 4,096 samples 4.98 -> 4.91 ms,
    //     8,192 19.9 -> 19.2, 8,192 x 50k leaves 100.3 -> 95.9; 3,584: 3.94 -> 3.80; it ties at 3,328, loses 1.7 % at 3,072
    //     (2.44 tiles per workgroup) and 10 % at 2,816;
    //   * ANY shard with 200,000 or more (tile, branch row) units per workgroup -- about 11 ms of kernel: on the
    //     trapezoid of a later row shard the third wave pays once a shard is several rounds long, and not before
    //     (tools/shard_balance.py at HEAD, profiles/r04_shard_balance.txt, max over ranks in ms, 8 waves / 12 waves:
    //     C4 over 2 GPUs 40.2 / 38.5, over 4 19.9 / 19.3, over 8 10.12 / 10.42 (one rank 4 % behind the others);
    //     C5 over 2 50.3 / 48.4, over 4 25.0 / 24.5, over 8 13.06 / 12.73; the weak problem, C3's pairs per rank,
    //     5.00-5.13 / 5.04-5.26).  Round 3's rule gave the first rank alone the 12-wave kernel whatever the shard's
    //     size, and said otherwise in this comment.
    pl->waves_per_wg = pl->sparse ? WAVES_PER_WG : waves_per_wg();
    // The first rule holds from 8,000 matrix rows up only: with the rare rows out of the matrix (round 5) a tile has a
    // quarter of the rows to sweep between loading its columns and writing its sums, and the 8-wave kernel with its
    // full prefetch does that better (matrix rows' kernel, ms, 12 / 8 waves: C3 with 4,968 rows 1.306 / 1.260, 6,000
    // samples with 4,679 rows 2.76 / 2.69, C5's tree at 1 % / 0.2 % density with 2,438 / 528 rows 2.64 / 2.42 and 0.77 /
    // 0.58; C5 with 12,462 rows 12.17 / 12.51; C4 -- the second rule -- 16.5 / 17.0).
    if (!pl->sparse && !ff::tuning("FF_WAVES_PER_WG").has_value() &&
        ((inf.row_begin == 0 && inf.n_tiles * 4 >= (int64_t)pl->n_workgroups * 11 && rows >= 8000) ||
         (double)inf.n_tiles * (double)rows >= 200000.0 * (double)pl->n_workgroups))
        pl->waves_per_wg = L_WAVES_PER_WG;
    pl->lds_bytes = 96 * 1024;  // unused dynamic LDS sized so that exactly one workgroup fits a CU
    const int U = pl->n_workgroups * pl->waves_per_wg;
    inf.n_wave_slots = U;
    std::vector<Item> items;
    std::vector<int32_t> item_ptr;
    // up to 255 planes of accumulators, within 1 GiB
    int max_planes = 255;
    while (max_planes > 1 && (double)max_planes * 4.0 * (double)std::max<int64_t>(n_slots, 1) > 1073741824.0) --max_planes;
    build_schedule(tiles, rows, U, &items, &item_ptr, &inf.elements, xcd_slices(), pl->waves_per_wg,
                   max_planes > 1 ? max_planes : 0, inf.row_begin > 0);
    inf.n_items = (int64_t)items.size();
    pl->n_planes = 1;
    for (const Item &it : items) pl->n_planes = std::max(pl->n_planes, (int)((it.flags >> 3) & 255u) + 1);
    pl->plane_stride = round_up(std::max<int64_t>(n_slots, 1), FINISH_RUN);  // (planes start 16-byte aligned: finish_fixed32_kernel)
    FF_HIP(hipMalloc(&pl->d_items, sizeof(Item) * std::max<size_t>(items.size(), 1)));
    FF_HIP(hipMalloc(&pl->d_item_ptr, sizeof(int32_t) * item_ptr.size()));
    if (!items.empty())
        FF_HIP(hipMemcpy(pl->d_items, items.data(), sizeof(Item) * items.size(), hipMemcpyHostToDevice));
    FF_HIP(hipMemcpy(pl->d_item_ptr, item_ptr.data(), sizeof(int32_t) * item_ptr.size(), hipMemcpyHostToDevice));
    FF_ALLOC(pl->d_num, sizeof(uint32_t) * (size_t)pl->plane_stride * (size_t)pl->n_planes, "the pair accumulators");
    // slots of tiles that are not split that way are never written in planes 1..: zero once
    FF_HIP(hipMemset(pl->d_num, 0, sizeof(uint32_t) * (size_t)pl->plane_stride * (size_t)pl->n_planes));
#ifdef FF_MFMA_DIAG  // (diagnostic build: per-wave clock stamps, tools/wave_stamps.py)
    if (env_int("FF_STAMPS", 0)) {
        FF_HIP(hipMalloc(&pl->d_stamps, sizeof(unsigned long long) * 4 * (size_t)U));
        FF_HIP(hipMemset(pl->d_stamps, 0, sizeof(unsigned long long) * 4 * (size_t)U));
    }
#endif
    if (pl->waves_per_wg == L_WAVES_PER_WG) {
        FF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_sad_kernel12),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes));
    } else if (pl->sparse) {
        FF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_sad_sparse_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes));
    } else {
        FF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_sad_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes));
    }
    return FF_OK;
}

int schedule_mfma(ff_plan *pl, char *err, size_t errlen)
{
    ff_plan_info &inf = pl->info;
    const int64_t N = inf.n_samples, n_slots = inf.slot_end - inf.slot_begin;
    free_and_null(pl->d_mitems);
    free_and_null(pl->d_mitem_ptr);
    free_and_null(pl->d_num);
    free_and_null(pl->d_partial);
    free_and_null(pl->d_ptiles);
    free_and_null(pl->d_ptile_ptr);
    pl->n_ptiles = 0;
    pl->m_small = false;
    pl->n_stiles = 0;
    pl->n_mitems = 0;
    const int64_t slabs = pl->m_ldb / M_KSLAB;
    const int G = inf.n_compute_units * M_WGS_PER_CU;  // one 8-wave workgroup per CU
    pl->n_mgroups = G;
    {
        // A shard with fewer 256 x 128 tiles than workgroups is all "remainder" for the persistent kernel --
        // every tile cut into branch ranges that each pay its 10 us of prologue and write-out, plus a reduce
        // launch.  Below S_MAX_WORK (32 x 32 tiles x k-steps; calibrated with tools/mfma_small_sweep.py) such a
        // shard takes pair_common_small_kernel instead: one 32 x 32 tile per workgroup over all branches, the
        // sum over the waves' ranges and the division inside the same launch.  FF_MFMA_SMALL=1 / 0 forces.
        // its tiles: row blocks of 32 in ascending order, block I with the column blocks 0 .. I (the kernel maps a
        // tile's ordinal to (I, J) by itself: small_tile_of)
        int64_t n_st = 0;
        const int64_t ib0 = inf.row_begin / S_TILE;
        for (int64_t i0 = ib0 * S_TILE; i0 < inf.row_end; i0 += S_TILE) {
            const int64_t w = std::min<int64_t>(std::min<int64_t>(i0 + S_TILE, inf.row_end) - 1, N);  // valid columns: j < w
            n_st += (w + S_TILE - 1) / S_TILE;  // (= I + 1, or I for a last block of one row)
        }
        int64_t big_tiles = 0;
        for (int64_t i0 = inf.row_begin / M_TILE_I * M_TILE_I; i0 < inf.row_end; i0 += M_TILE_I)
            big_tiles += (std::min<int64_t>(std::min<int64_t>(i0 + M_TILE_I, inf.row_end) - 1, N) + M_TILE_J - 1) / M_TILE_J;
        const int force = env_int("FF_MFMA_SMALL", -1);
        const bool fits = n_st > 0 && pl->m_digits <= S_MAX_DIGITS && n_st < ((int64_t)1 << 30) &&
                          pl->m_ldb * pl->m_digits <= S_TABLE_BYTES;  // (its digit planes live in LDS)
        const bool small = fits && (force >= 0 ? force != 0
                                               : big_tiles < G && (double)n_st * 2.0 * (double)slabs <= S_MAX_WORK);
        if (small) {
            pl->m_small = true;
            pl->n_stiles = (int)n_st;
            pl->stile_c0 = ib0 * (ib0 + 1) / 2;
            inf.kernel = FF_KERNEL_MFMA_I8_SMALL;
            inf.n_sweeps = 1;  // (every digit plane in its one pass)
            inf.planes_per_sweep = pl->m_digits;
            inf.rows_three_planes = 0;
            inf.n_tiles = inf.n_items = n_st;
            inf.n_wave_slots = n_st * S_WAVES;
            inf.elements = (double)n_st * S_TILE * S_TILE * (double)pl->m_ldb * pl->m_digits;
            pl->m_all_private = false;
            pl->m_any_atomic = false;
            pl->m_fused = true;  // every slot has one writer: it writes the distance
#define FF_S_ATTR(ND)                                                                                                   \
    FF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_common_small_kernel<ND>),                            \
                               hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES));
            FF_S_ATTR(1) FF_S_ATTR(2) FF_S_ATTR(3) FF_S_ATTR(4) FF_S_ATTR(5)
#undef FF_S_ATTR
            FF_ALLOC(pl->d_num, sizeof(uint32_t) * (size_t)std::max<int64_t>(n_slots, 1), "the pair accumulators");
            return FF_OK;
        }
        inf.kernel = FF_KERNEL_MFMA_I8;
    }
    // graded rows: ONE sweep, three planes up to m_duo_from_slab and two from there on (one digit group per tile,
    // cut by cost); else base-128 digits, two planes per sweep
    const int sched_digits = pl->m_graded ? 2 : pl->m_digits;
    const int64_t duo_from_quad = pl->m_graded ? (pl->m_duo_from_slab + M_QUAD_SLABS - 1) / M_QUAD_SLABS : -1;
    inf.n_sweeps = (sched_digits + M_ND - 1) / M_ND;
    inf.planes_per_sweep = pl->m_graded ? 3 : std::min(pl->m_digits, M_ND);
    inf.rows_three_planes = pl->m_graded ? std::min<int64_t>((int64_t)pl->m_duo_from_slab * M_KSLAB, pl->m_ldb) : 0;
    std::vector<MItem> mi;
    std::vector<int32_t> mptr;
    std::vector<int32_t> ptiles, pptr;
    const bool want_partials = true;
    // Up to FF_MFMA_PRIVATE_MB of partial tiles (128 KiB each), every item gets its own: the kernel's
    // copy-out is then aligned 512-byte rows into a contiguous tile (2.8 us a tile at C3) instead of 4-byte
    // stores into rows of the triangle that start anywhere (12.8 us), and reduce_partials_kernel writes the
    // distances straight from the sums -- no num[] round trip, no finish launch.
    int64_t private_tiles = (int64_t)env_int("FF_MFMA_PRIVATE_MB", 2048) * (1 << 20) / (M_TILE_I * M_TILE_J * 4);
    int64_t n_mtiles = 0;
    for (;;) {
        n_mtiles = build_mfma_schedule(N, inf.row_begin, inf.row_end, slabs, sched_digits, G, &mi, &mptr,
                                       want_partials ? &ptiles : nullptr, want_partials ? &pptr : nullptr, private_tiles,
                                       duo_from_quad);
        if (pptr.empty()) break;
        if (hipMalloc(&pl->d_partial, sizeof(uint32_t) * (size_t)pptr.back() * M_TILE_I * M_TILE_J) == hipSuccess) break;
        (void)hipGetLastError();  // (the device is short of memory: only the remainder's ranges get private tiles)
        pl->d_partial = nullptr;
        if (private_tiles == 0)
            return ff::fail(FF_ERR_DEVICE, err, errlen, "out of device memory for %lld partial tiles of the matrix-core schedule",
                            (long long)pptr.back());
        private_tiles = 0;
    }
    if (!pptr.empty()) {
        pl->n_ptiles = (int)pptr.size() - 1;
        FF_HIP(hipMalloc(&pl->d_ptiles, sizeof(int32_t) * ptiles.size()));
        FF_HIP(hipMalloc(&pl->d_ptile_ptr, sizeof(int32_t) * pptr.size()));
        FF_HIP(hipMemcpy(pl->d_ptiles, ptiles.data(), sizeof(int32_t) * ptiles.size(), hipMemcpyHostToDevice));
        FF_HIP(hipMemcpy(pl->d_ptile_ptr, pptr.data(), sizeof(int32_t) * pptr.size(), hipMemcpyHostToDevice));
    }
    pl->n_mitems = (int)mi.size();
    inf.n_tiles = n_mtiles;
    inf.n_items = (int64_t)mi.size();
    inf.n_wave_slots = (int64_t)G * (M_THREADS / 64);
    inf.elements = (double)n_mtiles * M_TILE_I * M_TILE_J *
                   (pl->m_graded ? 3.0 * std::min<double>(pl->m_duo_from_slab * M_KSLAB, pl->m_ldb) +
                                       2.0 * std::max<double>(0.0, (double)pl->m_ldb - pl->m_duo_from_slab * M_KSLAB)
                                 : (double)pl->m_ldb * pl->m_digits);
    FF_HIP(hipMalloc(&pl->d_mitems, sizeof(MItem) * std::max<size_t>(mi.size(), 1)));
    FF_HIP(hipMalloc(&pl->d_mitem_ptr, sizeof(int32_t) * mptr.size()));
    if (!mi.empty()) FF_HIP(hipMemcpy(pl->d_mitems, mi.data(), sizeof(MItem) * mi.size(), hipMemcpyHostToDevice));
    FF_HIP(hipMemcpy(pl->d_mitem_ptr, mptr.data(), sizeof(int32_t) * mptr.size(), hipMemcpyHostToDevice));
    pl->lds_bytes = (size_t)M_LDS_BYTES;
    pl->m_all_private = !mi.empty();
    pl->m_any_atomic = false;
    for (const MItem &it : mi) {
        pl->m_all_private = pl->m_all_private && it.pad > 0;
        pl->m_any_atomic = pl->m_any_atomic || it.pad == 0;
    }
    // The matrix-core path can finish in place when every slot has exactly one writer (its tile's
    // only item, or a reduce kernel): the integer sums then never go through num[], and there is
    // neither a memset nor a finish launch: whenever every item owns a private partial tile.
    pl->m_fused = !pl->m_any_atomic && pl->m_all_private;
    FF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_common_mfma_kernel<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes));
    FF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_common_mfma_kernel<true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes));
    FF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_common_mfma_kernel<false, 0, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes));
    FF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pair_common_mfma_kernel<true, 0, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes));
    FF_ALLOC(pl->d_num, sizeof(uint32_t) * (size_t)std::max<int64_t>(n_slots, 1), "the pair accumulators");
    return FF_OK;
}

namespace {

// Workgroups of refine_exact_kernel a compute unit holds at once: its grid is one round of them (a workgroup walks
// its pairs one after the other; a second round of workgroups would wait for the first to finish all of theirs).
int refine_blocks_per_cu()
{
    static const int n = [] {
        int b = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, refine_exact_kernel, REFINE_THREADS, 0) != hipSuccess || b < 1) b = 2;
        return b;
    }();
    return n;
}

// One launch of the EXACT64 pair kernel with the plan's tile height.
int launch_exact64(ff_plan *pl, hipStream_t st, double *d_out, char *err, size_t errlen)
{
    const ff_plan_info &inf = pl->info;
    if (pl->xu) {
        if (pl->n_xutiles > 0)
            pair_exact_unw_kernel<<<dim3((unsigned)pl->n_xutiles), dim3(64), 0, st>>>(pl->d_Xbits, pl->xu_ldx, pl->d_len_rows, pl->xu_slabs,
                                                                                     pl->d_xutiles, inf.row_begin, inf.row_end,
                                                                                     inf.slot_begin, d_out);
        FF_HIP(hipGetLastError());
        return FF_OK;
    }
    if (pl->n_xtiles <= 0) return FF_OK;
    const unsigned nb = (unsigned)((pl->n_xtiles + 3) / 4);
    const double *len = pl->d_len_rows ? pl->d_len_rows : pl->d_len;
    // weighted: the kernel that takes the reference's shortcut for branches a row has not (heights 8, 12, 16;
    // FF_X_SKIP=0: pair_exact64_kernel, six operations for every term; same bits)
    if (pl->x_skip) {
        if (pl->x_tile_h == 8)
            pair_exact64_skip_kernel<8><<<dim3(nb), dim3(256), 0, st>>>(pl->d_DT, inf.ld, len, inf.n_rows, pl->d_xtiles, pl->n_xtiles,
                                                                      inf.row_begin, inf.row_end, inf.slot_begin, d_out);
        else if (pl->x_tile_h == 12)
            pair_exact64_skip_kernel<12><<<dim3(nb), dim3(256), 0, st>>>(pl->d_DT, inf.ld, len, inf.n_rows, pl->d_xtiles, pl->n_xtiles,
                                                                       inf.row_begin, inf.row_end, inf.slot_begin, d_out);
        else
            pair_exact64_skip_kernel<16><<<dim3(nb), dim3(256), 0, st>>>(pl->d_DT, inf.ld, len, inf.n_rows, pl->d_xtiles, pl->n_xtiles,
                                                                       inf.row_begin, inf.row_end, inf.slot_begin, d_out);
        FF_HIP(hipGetLastError());
        return FF_OK;
    }
#define FF_X_CASE(H)                                                                                              \
    case H:                                                                                                       \
        if (pl->weighted)                                                                                         \
            pair_exact64_kernel<true, H><<<dim3(nb), dim3(256), 0, st>>>(pl->d_DT, inf.ld, len, inf.n_rows, pl->d_xtiles, \
                                                                        pl->n_xtiles, inf.row_begin, inf.row_end,  \
                                                                        inf.slot_begin, d_out);                    \
        else                                                                                                      \
            pair_exact64_kernel<false, H><<<dim3(nb), dim3(256), 0, st>>>(pl->d_DT, inf.ld, len, inf.n_rows, pl->d_xtiles, \
                                                                         pl->n_xtiles, inf.row_begin, inf.row_end, \
                                                                         inf.slot_begin, d_out);                   \
        break;
    switch (pl->x_tile_h) {
        FF_X_CASE(4)
        FF_X_CASE(8)
        FF_X_CASE(10)
        FF_X_CASE(12)
        FF_X_CASE(14)
        FF_X_CASE(16)
    default: return ff::fail(FF_ERR_ARG, err, errlen, "EXACT64: no kernel for tile height %d", pl->x_tile_h);
    }
#undef FF_X_CASE
    FF_HIP(hipGetLastError());
    return FF_OK;
}

// Tiles of height h for the plan's shard.
int upload_exact64_tiles(ff_plan *pl, int h, char *err, size_t errlen)
{
    ff_plan_info &inf = pl->info;
    free_and_null(pl->d_xtiles);
    std::vector<Tile> tiles;
    build_tiles(inf.n_samples, inf.row_begin, inf.row_end, h, X_TILE_J, false, &tiles);
    inf.n_tiles = inf.n_items = (int64_t)tiles.size();
    inf.elements = (double)tiles.size() * h * X_TILE_J * (double)inf.n_rows;
    std::vector<XTile> xt(tiles.size());
    for (size_t k = 0; k < tiles.size(); ++k) xt[k] = {tiles[k].i0, tiles[k].j0};
    // one wave per tile, 4 per block: a launch carries fewer than 2^32 threads
    if (xt.size() >= ((size_t)1 << 26))
        return ff::fail(FF_ERR_ARG, err, errlen, "EXACT64: %zu pair tiles in one shard, at most %zu (use more shards)",
                        xt.size(), ((size_t)1 << 26) - 1);
    pl->n_xtiles = (int)xt.size();
    pl->x_tile_h = h;
    pl->x_skip = pl->weighted && inf.n_rows > 0 && env_int("FF_X_SKIP", 1) != 0 && (h == 8 || h == 12 || h == 16);
    inf.kernel = pl->x_skip ? FF_KERNEL_EXACT_F64_SKIP : FF_KERNEL_EXACT_F64;
    FF_HIP(hipMalloc(&pl->d_xtiles, sizeof(XTile) * std::max<size_t>(xt.size(), 1)));
    if (!xt.empty()) FF_HIP(hipMemcpy(pl->d_xtiles, xt.data(), sizeof(XTile) * xt.size(), hipMemcpyHostToDevice));
    inf.n_wave_slots = (int64_t)xt.size();
    return FF_OK;
}

}  // namespace

// The tile height is picked once per plan.  Every height gives every pair the same operations in the
// same order; what differs is the number of waves and how their count divides into rounds of resident
// waves (C3: 33.9 ms with 16 rows, 29.8 with 12; 2,500 samples x 20,000 leaves: 34.1 with 16, 27.2 with 10).
// The height comes from a model of the tiles' rounds fitted to sweeps over sample counts (below); FF_X_TILE_H forces one.
// The tiles of pair_exact_unw_kernel for the plan's shard.  Two column groups per tile (64 accumulator registers, six
// waves per SIMD) is what the vector ALU wants; a shard with fewer such tiles than SIMDs is bound by one wave's chain
// of steps and takes single groups (twice the waves, shorter steps).
int schedule_exact_unw(ff_plan *pl, char *err, size_t errlen)
{
    ff_plan_info &inf = pl->info;
    free_and_null(pl->d_xutiles);
    std::vector<XUTile> tiles;
    int jmax = XU_JMAX;
    build_xu_tiles(inf.n_samples, inf.row_begin, inf.row_end, jmax, &tiles);
    const int forced = env_int("FF_XU_JMAX", 0);
    if (forced == 1 || forced == 2) jmax = forced;
    // (single column groups while two would make fewer than six tiles per CU: 1,536 samples 3.62 -> 2.92 ms, 2,048 -- 8.5 per CU -- 3.80 against 3.55)
    else if ((int64_t)tiles.size() < (int64_t)inf.n_compute_units * 6) jmax = 1;
    if (jmax != XU_JMAX) build_xu_tiles(inf.n_samples, inf.row_begin, inf.row_end, jmax, &tiles);
    // one 64-thread workgroup per tile: a launch carries fewer than 2^31 of them
    if (tiles.size() >= ((size_t)1 << 31))
        return ff::fail(FF_ERR_ARG, err, errlen, "EXACT64: %zu pair tiles in one shard, at most %zu (use more shards)",
                        tiles.size(), ((size_t)1 << 31) - 1);
    inf.n_tiles = inf.n_items = inf.n_wave_slots = (int64_t)tiles.size();
    double cols = 0;
    for (const XUTile &t : tiles) cols += 64.0 * t.jn;
    inf.elements = cols * XU_TILE_H * (double)inf.n_rows;
    pl->n_xutiles = (int)tiles.size();
    FF_HIP(hipMalloc(&pl->d_xutiles, sizeof(XUTile) * std::max<size_t>(tiles.size(), 1)));
    if (!tiles.empty()) FF_HIP(hipMemcpy(pl->d_xutiles, tiles.data(), sizeof(XUTile) * tiles.size(), hipMemcpyHostToDevice));
    return FF_OK;
}

int schedule_exact64(ff_plan *pl, char *err, size_t errlen)
{
    ff_plan_info &inf = pl->info;
    if (pl->xu) return schedule_exact_unw(pl, err, errlen);
    if (pl->x_tile_h == 0) {
        const int forced = env_int("FF_X_TILE_H", 0);
        int h = X_TILE_H_DEFAULT;
        // A shard whose waves all fit the device at once is bound by one wave's chain of trips, not by the
        // vector ALU: the lowest tile that still leaves the device about half empty (C2: 1.31 ms with 16 rows, 0.84
        // with 8, 0.56 with 4; 2 rows and 16 two-value scalar loads per trip are slower again: 0.75).  Weighted, 10,000
        // leaves, ms with 4 / 8 / 12 rows: 1,280 samples 5.02 / 5.09 / 6.10, 1,536: 6.09 / 5.86 / 6.08, 1,792: 8.95 / 7.41 /
        // 7.68, 2,048: 12.08 / 8.15 / 7.41 -- 4 rows up to 0.55 of the wave slots, 8 rows up to 0.45 of them.
        const int64_t resident = (int64_t)inf.n_compute_units * 4 * 8;
        bool small = false;
        for (int cand : {4, 8}) {
            std::vector<Tile> count;
            build_tiles(inf.n_samples, inf.row_begin, inf.row_end, cand, X_TILE_J, false, &count);
            if ((int64_t)count.size() * 100 <= resident * (cand == 4 ? 55 : 45)) {
                h = cand;
                small = true;
                break;
            }
        }
        // Larger shards are bound by the vector ALU, every SIMD working through the tiles it is dealt one after the
        // other (interleaved): a SIMD gets floor or ceil of tiles / SIMDs of them, and the kernel ends with the SIMDs
        // that got the ceiling -- so the height decides how much of the last "tile per SIMD" is idle.  With
        // avg = tiles(h) / SIMDs the efficiency is avg / ceil(avg), times what the height itself is worth (scalar
        // operands per trip, waves per SIMD; from the sweep's largest sizes).  This ranks the five heights as
        // measured at every size of tools/exact64_sweep.py (round 3; the fixed 12 rows of round 2 lost 6 % at 3,072
        // samples, 4 % at 2,048 and 3,584); 4,096 samples keep their 12 rows.
        if (!small) {
            const double simds = (double)inf.n_compute_units * 4.0;
            double best_score = 0;
            // (weighted: pair_exact64_skip_kernel has the heights 8, 12 and 16; at C3's shape 25.5 / 21.2 / 22.4 ms with
            // 16.25 / 10.9 / 8.1 tiles per SIMD, i.e. per tile-round worth 0.83 / 0.96 / 1)
            const bool skip = pl->weighted && env_int("FF_X_SKIP", 1) != 0;
            for (int cand : {8, 10, 12, 14, 16}) {
                if (skip && cand != 8 && cand != 12 && cand != 16) continue;
                std::vector<Tile> count;
                build_tiles(inf.n_samples, inf.row_begin, inf.row_end, cand, X_TILE_J, false, &count);
                const double avg = (double)count.size() / simds;
                const double worth = skip ? (cand == 8 ? 0.83 : cand == 12 ? 0.96 : 1.0)
                                          : cand == 8 ? 0.95 : cand == 10 ? 0.97 : cand == 16 ? 0.985 : 1.0;
                const double score = worth * avg / std::ceil(avg);
                if (score > best_score + 1e-12) {
                    best_score = score;
                    h = cand;
                }
            }
        }
        for (int cand : X_TILE_HEIGHTS)
            if (cand == forced) h = forced;
        pl->x_tile_h = h;
    }
    return upload_exact64_tiles(pl, pl->x_tile_h, err, errlen);
}

// The queue of pairs to recompute exactly holds up to an eighth of the shard (at least 2^20);
// next to it the run-time audit's sample of the shard and its binary64 distances.
int alloc_refine_queue(ff_plan *pl, char *err, size_t errlen)
{
    const int64_t n_slots = pl->info.slot_end - pl->info.slot_begin;
    free_and_null(pl->d_refine_list);
    free_and_null(pl->d_audit_slots);
    free_and_null(pl->d_audit_exact);
    pl->n_audit = 0;
    pl->refine_cap = (unsigned long long)std::min<int64_t>(n_slots, std::max<int64_t>(1 << 20, n_slots / 8));
    FF_HIP(hipMalloc(&pl->d_refine_list, sizeof(unsigned long long) * (size_t)std::max<unsigned long long>(pl->refine_cap, 1)));
    if (!pl->d_refine_count) FF_HIP(hipMalloc(&pl->d_refine_count, sizeof(unsigned long long) * CNT_N + sizeof(uint32_t) * HEADROOM_SLOTS));
    if (!pl->d_risk_list && env_int("FF_AUDIT", 1) != 0) FF_HIP(hipMalloc(&pl->d_risk_list, sizeof(unsigned long long) * RISK_CAP));
    if (!pl->d_n_nodes) {
        const int64_t ns = pl->info.n_samples;
        FF_HIP(hipMalloc(&pl->d_n_nodes, sizeof(int32_t) * (size_t)std::max<int64_t>(ns, 1)));
        if (ns > 0) node_counts_kernel<<<dim3((unsigned)((ns + 255) / 256)), dim3(256)>>>(pl->d_indptr, ns, pl->d_n_nodes);
        FF_HIP(hipGetLastError());
        FF_HIP(hipDeviceSynchronize());  // (runs may come on any stream)
    }
    reset_counters_kernel<<<dim3(1), dim3(HEADROOM_SLOTS)>>>(pl->d_refine_count);
    FF_HIP(hipGetLastError());
    if (n_slots > 0 && env_int("FF_AUDIT", 1) != 0) {
        // the uniform sample grows with the shard: AUDIT_PAIRS per 2^23 pairs of it (C3 as a whole: 4,096; C4: 65,536)
        const int64_t want_n = std::min<int64_t>(AUDIT_PAIRS_MAX, AUDIT_PAIRS * ((n_slots + ((int64_t)1 << 23) - 1) >> 23));
        const int n = (int)std::min<int64_t>(want_n, n_slots);
        std::vector<int64_t> slots((size_t)n);
        uint64_t x = 0x5EEDF4ACull ^ (uint64_t)pl->info.slot_begin;
        for (int q = 0; q < n; ++q) {  // splitmix64
            uint64_t z = (x += 0x9E3779B97F4A7C15ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            slots[(size_t)q] = n_slots <= n ? q : (int64_t)(z % (uint64_t)n_slots);
        }
        FF_HIP(hipMalloc(&pl->d_audit_slots, sizeof(int64_t) * (size_t)n));
        FF_HIP(hipMalloc(&pl->d_audit_exact, sizeof(double) * (size_t)n));
        FF_HIP(hipMemcpy(pl->d_audit_slots, slots.data(), sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice));
        audit_exact_kernel<<<dim3((unsigned)n), dim3(64)>>>(pl->d_audit_slots, pl->d_indptr, pl->d_ids, pl->d_abnd,
                                                             pl->d_len, pl->weighted, pl->info.slot_begin,
                                                             pl->d_audit_exact);
        FF_HIP(hipGetLastError());
        FF_HIP(hipDeviceSynchronize());  // runs may come on any stream
        pl->n_audit = n;
    }
    return FF_OK;
}

int schedule_for_shard(ff_plan *pl, char *err, size_t errlen)
{
    if (pl->walk) return FF_OK;  // (a grid-stride loop over the shard's slots: nothing to build)
    int rc = pl->mfma ? schedule_mfma(pl, err, errlen)
             : pl->info.precision == FF_PRECISION_FIXED32 ? schedule_sad(pl, err, errlen)
                                                          : schedule_exact64(pl, err, errlen);
    if (rc == FF_OK && pl->refine) rc = alloc_refine_queue(pl, err, errlen);
    return rc;
}

int plan_run_impl(ff_plan *pl, hipStream_t st, double *d_out, bool timed, char *err, size_t errlen)
{
    const ff_plan_info &inf = pl->info;
    const int64_t n_slots = inf.slot_end - inf.slot_begin;
    if (n_slots <= 0) return FF_OK;
    if (!d_out) return ff::fail(FF_ERR_ARG, err, errlen, "null output pointer");
    DeviceScope scope;  // (the caller's current device is its own again when this returns: a host that drives several
    FF_HIP(scope.enter(pl->device));  // plans on several devices from one thread does not find it changed under it)
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_mid = nullptr;
    if (timed) {
        if (pl->events_used == pl->events.size()) {
            hipEvent_t a, b, m;
            FF_HIP(hipEventCreate(&a));
            FF_HIP(hipEventCreate(&b));
            FF_HIP(hipEventCreate(&m));
            pl->events.push_back({a, b, m});
        }
        ev0 = pl->events[pl->events_used].first;
        ev1 = pl->events[pl->events_used].second;
        ev_mid = pl->events[pl->events_used].mid;
        ++pl->events_used;
    }
    if (pl->walk) {
        if (timed) FF_HIP(hipEventRecord(ev0, st));
        pair_walk_kernel<<<dim3((unsigned)std::min<int64_t>((n_slots + 255) / 256, (int64_t)inf.n_compute_units * 8)), dim3(256), 0, st>>>(
            pl->d_indptr, pl->d_ids, pl->d_abnd, pl->d_len, pl->weighted, inf.slot_begin, n_slots, d_out);
        if (timed) FF_HIP(hipEventRecord(ev1, st));
        FF_HIP(hipGetLastError());
        return FF_OK;
    }
    if (inf.precision == FF_PRECISION_FIXED32) {
        FinishArgs fin;
        fin.W = pl->d_W;
        fin.wex = pl->d_wex;
        fin.out = d_out;
        fin.n_nodes = pl->refine ? pl->d_n_nodes : nullptr;
        fin.refine_list = pl->d_refine_list;
        fin.refine_count = pl->d_refine_count;
        fin.refine_cap = pl->refine_cap;
        fin.risk_list = pl->refine ? pl->d_risk_list : nullptr;
        fin.scale_log2 = inf.scale_log2;
        fin.weighted = pl->weighted;
        fin.mlow = pl->split ? pl->d_mlow : nullptr;
        fin.wl = pl->d_Wl;
        const bool fused = pl->mfma && pl->m_fused;  // (decided when the shard was scheduled: schedule_mfma)
        if (pl->refine) reset_counters_kernel<<<dim3(1), dim3(HEADROOM_SLOTS), 0, st>>>(pl->d_refine_count);
        if (!fused && (!pl->mfma || pl->m_any_atomic))
            FF_HIP(hipMemsetAsync(pl->d_num, 0, sizeof(uint32_t) * (size_t)n_slots, st));
        if (timed) FF_HIP(hipEventRecord(ev0, st));
        if (pl->mfma && pl->m_small) {
            FinishArgs none = fin;
            none.out = nullptr;  // null: integer sums into num[]
            const dim3 grid((unsigned)pl->n_stiles), block(S_THREADS);
            const uint4 *bits = reinterpret_cast<const uint4 *>(pl->d_Pbits);
            const int n_slab_pairs = (int)(pl->m_ldb / (2 * M_KSLAB));
#define FF_S_CASE(ND)                                                                                                  \
    case ND:                                                                                                           \
        pair_common_small_kernel<ND><<<grid, block, (size_t)(pl->m_ldb * ND + S_RED_BYTES), st>>>(bits, pl->m_n8, pl->d_Kd, pl->m_ldb, n_slab_pairs,        \
                                                             pl->stile_c0, pl->d_W, pl->d_num, inf.row_begin,          \
                                                             inf.row_end, inf.slot_begin, fused ? fin : none);         \
        break;
            switch (pl->m_digits) {
                FF_S_CASE(1)
                FF_S_CASE(2)
                FF_S_CASE(3)
                FF_S_CASE(4)
                FF_S_CASE(5)
            default: return ff::fail(FF_ERR_INTERNAL, err, errlen, "no small-shard kernel for %d digits", pl->m_digits);
            }
#undef FF_S_CASE
            if (timed) FF_HIP(hipEventRecord(ev1, st));
        } else if (pl->mfma) {
            auto kern = pl->m_graded ? (pl->m_all_private ? pair_common_mfma_kernel<true, 0, true> : pair_common_mfma_kernel<false, 0, true>)
                                     : (pl->m_all_private ? pair_common_mfma_kernel<true> : pair_common_mfma_kernel<false>);
#ifdef FF_MFMA_DIAG  // ablations for timing only (wrong results): see the kernel's DIAG parameter
            switch (env_int("FF_MFMA_DIAG", 0)) {
            case 2: kern = pair_common_mfma_kernel<false, 2>; break;
            case 4: kern = pair_common_mfma_kernel<false, 4>; break;
            case 8: kern = pair_common_mfma_kernel<false, 8>; break;
            case 6: kern = pair_common_mfma_kernel<false, 6>; break;
            case 14: kern = pair_common_mfma_kernel<false, 14>; break;
            default: break;
            }
            if (env_int("FF_MFMA_DIAG", 0))
                FF_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl->lds_bytes));
#endif
            FinishArgs none = fin;
            none.out = nullptr;  // null: the kernels leave integer sums in num[]
            if (pl->n_mitems > 0)
                kern<<<dim3((unsigned)pl->n_mgroups), dim3(M_THREADS), pl->lds_bytes, st>>>(
                    reinterpret_cast<const uint4 *>(pl->d_Pbits), pl->m_n8, pl->m_graded ? pl->d_Kt : pl->d_Kd, pl->m_ldb, pl->d_mitems, pl->d_mitem_ptr, pl->d_W, pl->d_num,
                    pl->d_partial, inf.row_begin, inf.row_end, inf.slot_begin, pl->m_duo_from_slab, fused ? fin : none);
            if (pl->n_ptiles > 0 && pl->m_all_private)
                reduce_private_kernel<<<dim3(M_REDUCE_BLOCKS, (unsigned)pl->n_ptiles), dim3(M_REDUCE_THREADS), 0, st>>>(
                    pl->d_partial, pl->d_ptiles, pl->d_ptile_ptr, pl->d_W, pl->d_num, inf.row_begin, inf.row_end,
                    inf.slot_begin, fused ? fin : none);
            else if (pl->n_ptiles > 0)
                reduce_partials_kernel<<<dim3(M_REDUCE_BLOCKS, (unsigned)pl->n_ptiles), dim3(M_REDUCE_THREADS), 0, st>>>(
                    pl->d_partial, pl->d_ptiles, pl->d_ptile_ptr, pl->d_num, inf.row_begin, inf.row_end, inf.slot_begin,
                    fused ? fin : none);
            // the timed region is the pair kernel AND the reduction of its partial tiles (sums, W_i + W_j,
            // divisions: work that round 1's pair kernel did itself)
            if (timed) FF_HIP(hipEventRecord(ev1, st));
        } else if (inf.n_items > 0 && pl->sparse)
            pair_sad_sparse_kernel<<<dim3((unsigned)pl->n_workgroups), dim3(WAVES_PER_WG * 64), pl->lds_bytes, st>>>(
                pl->d_QT, inf.ld, pl->d_items, pl->d_item_ptr, pl->d_arows, pl->d_aptr16, pl->aptr_stride, pl->d_cs16,
                pl->zero_row, pl->d_num, pl->plane_stride, inf.row_begin, inf.row_end, inf.slot_begin);
        else if (inf.n_items > 0)
            (pl->waves_per_wg == L_WAVES_PER_WG ? pair_sad_kernel12 : pair_sad_kernel)
                <<<dim3((unsigned)pl->n_workgroups), dim3((unsigned)pl->waves_per_wg * 64), pl->lds_bytes, st>>>(
                pl->d_QT, inf.ld, pl->d_items, pl->d_item_ptr, pl->d_num, pl->plane_stride, inf.row_begin, inf.row_end,
                inf.slot_begin, pl->d_stamps, SYNC_TRIPS);
        if (pl->split && pl->n_low_tiles > 0) {  // (inside the timed region: it is part of the pair reduction)
            if (timed) FF_HIP(hipEventRecord(ev_mid, st));
            static_assert(sizeof(LOW_TILES) / sizeof(int) == 5, "one instance of pair_low_kernel per block side");
            auto low = pl->low_tile == 128 ? pair_low_kernel<128> : pl->low_tile == 112 ? pair_low_kernel<112> :
                       pl->low_tile == 96 ? pair_low_kernel<96> : pl->low_tile == 80 ? pair_low_kernel<80> : pair_low_kernel<64>;
            low<<<dim3((unsigned)pl->n_low_tiles), dim3(LOW_THREADS), 0, st>>>(
                pl->d_low_ptr, pl->d_low_ent, pl->d_low_bits, pl->low_words, pl->low_rows + 1, pl->d_low_tiles,
                inf.n_samples, inf.row_begin, inf.row_end, inf.slot_begin, pl->d_mlow);
        }
        if (timed && !pl->mfma) FF_HIP(hipEventRecord(ev1, st));
        if (!fused) {
            const unsigned nb = (unsigned)std::min<int64_t>((n_slots + 256 * FINISH_RUN - 1) / (256 * FINISH_RUN), 1 << 22);
            finish_fixed32_kernel<<<dim3(nb), dim3(256), 0, st>>>(pl->d_num, pl->n_planes, pl->plane_stride, fin, inf.slot_begin, n_slots);
        }
        if (pl->refine)
            refine_exact_kernel<<<dim3((unsigned)(inf.n_compute_units * refine_blocks_per_cu())), dim3(REFINE_THREADS), 0, st>>>(
                pl->d_refine_list, pl->d_refine_count, pl->refine_cap, pl->d_indptr, pl->d_ids, pl->d_abnd,
                pl->d_len, pl->weighted, inf.slot_begin, d_out);
        if (pl->refine && pl->n_audit > 0)
            audit_compare_kernel<<<dim3((unsigned)((pl->n_audit + 255) / 256)), dim3(256), 0, st>>>(
                pl->d_audit_slots, pl->d_audit_exact, pl->n_audit, d_out, pl->d_refine_count);
        if (pl->refine && pl->d_risk_list)
            audit_risk_kernel<<<dim3((unsigned)RISK_CAP), dim3(64), 0, st>>>(pl->d_risk_list, pl->d_refine_count, pl->d_indptr, pl->d_ids,
                                                                             pl->d_abnd, pl->d_len, pl->weighted, inf.slot_begin, d_out);
    } else {
        if (timed) FF_HIP(hipEventRecord(ev0, st));
        {
            const int rc = launch_exact64(pl, st, d_out, err, errlen);
            if (rc != FF_OK) return rc;
        }
        if (timed) FF_HIP(hipEventRecord(ev1, st));
    }
    FF_HIP(hipGetLastError());
    return FF_OK;
}

// After a completed FIXED32 run: did it deliver what the tolerance promises?  Not when more pairs
// were queued for the binary64 walk than the queue holds, or when a pair of the audit sample is
// further than AUDIT_REL from its binary64 value.  `why` gets the sentence for the caller.
int plan_fixed32_verdict(ff_plan *pl, bool *ok, std::string *why)
{
    *ok = true;
    if (!pl->refine) return FF_OK;
    unsigned long long c[CNT_N] = {};
    if (hipMemcpy(c, pl->d_refine_count, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) return FF_ERR_DEVICE;
    char buf[256];
    if (c[0] > pl->refine_cap) {
        snprintf(buf, sizeof buf, "%llu nearly identical pairs, %llu can be re-computed exactly", c[0], pl->refine_cap);
        *ok = false;
    } else if (c[1] > 0) {
        double worst;
        memcpy(&worst, &c[2], sizeof worst);
        snprintf(buf, sizeof buf, "%llu of %llu audited pairs are further than %.1e from their binary64 value (worst %.2e)",
                 c[1], (unsigned long long)pl->n_audit + c[CNT_RISK_CHECKED], AUDIT_REL, worst);
        *ok = false;
    }
    if (!*ok && why) *why = buf;
    return FF_OK;
}

}  // namespace dev
}  // namespace ff

using namespace ff::dev;

extern "C" {

int ff_plan_refined_pairs(ff_plan *pl, int64_t *queued, int64_t *capacity)
{
    if (!pl || !queued || !capacity) return FF_ERR_ARG;
    *queued = 0;
    *capacity = (int64_t)pl->refine_cap;
    if (!pl->refine) return FF_OK;
    unsigned long long n = 0;  // (CNT_QUEUED is the first counter)
    if (hipMemcpy(&n, pl->d_refine_count, sizeof n, hipMemcpyDeviceToHost) != hipSuccess) return FF_ERR_DEVICE;
    *queued = (int64_t)n;
    return FF_OK;
}

int ff_plan_audit(ff_plan *pl, int64_t *checked, int64_t *failed, double *max_rel_err)
{
    if (!pl || !checked || !failed || !max_rel_err) return FF_ERR_ARG;
    *checked = 0;
    *failed = 0;
    *max_rel_err = 0.0;
    if (!pl->refine || (pl->n_audit <= 0 && !pl->d_risk_list)) return FF_OK;
    unsigned long long c[CNT_N] = {};
    if (hipMemcpy(c, pl->d_refine_count, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) return FF_ERR_DEVICE;
    *checked = pl->n_audit + (int64_t)c[CNT_RISK_CHECKED];
    *failed = (int64_t)c[CNT_AUDIT_FAILED];
    memcpy(max_rel_err, &c[CNT_AUDIT_WORST], sizeof(double));
    return FF_OK;
}

int ff_plan_audit_detail(ff_plan *pl, int64_t *uniform_checked, int64_t *risk_found, int64_t *risk_checked, double *min_headroom)
{
    if (!pl || !uniform_checked || !risk_found || !risk_checked || !min_headroom) return FF_ERR_ARG;
    *uniform_checked = *risk_found = *risk_checked = 0;
    *min_headroom = INFINITY;
    if (!pl->refine) return FF_OK;
    unsigned long long c[CNT_N] = {};
    if (hipMemcpy(c, pl->d_refine_count, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) return FF_ERR_DEVICE;
    *uniform_checked = pl->n_audit;
    *risk_found = (int64_t)c[CNT_RISK_FOUND];
    *risk_checked = (int64_t)c[CNT_RISK_CHECKED];
    std::vector<uint32_t> slots(HEADROOM_SLOTS);
    if (hipMemcpy(slots.data(), pl->d_refine_count + CNT_N, sizeof(uint32_t) * HEADROOM_SLOTS, hipMemcpyDeviceToHost) != hipSuccess)
        return FF_ERR_DEVICE;
    const uint32_t bits = *std::min_element(slots.begin(), slots.end());
    float h2;
    memcpy(&h2, &bits, sizeof h2);
    *min_headroom = pl->d_risk_list ? std::sqrt((double)h2) : INFINITY;
    return FF_OK;
}

#ifdef FF_MFMA_DIAG
// Diagnostic build only.  First call (host_out == null): allocates the stamp array for n_workgroups
// and arms the kernel.  Later calls copy the stamps out ([workgroup][4 items][8] 100 MHz ticks).
int ff_debug_mfma_stamps(unsigned long long *host_out, int64_t n_workgroups)
{
    static unsigned long long *d = nullptr;
    const size_t bytes = (size_t)n_workgroups * 4 * 8 * sizeof(unsigned long long);
    if (!host_out) {
        if (hipMalloc(&d, bytes) != hipSuccess || hipMemset(d, 0, bytes) != hipSuccess) return FF_ERR_DEVICE;
        return hipMemcpyToSymbol(HIP_SYMBOL(g_mfma_stamps), &d, sizeof(d)) == hipSuccess ? FF_OK : FF_ERR_DEVICE;
    }
    if (!d) return FF_ERR_ARG;
    return hipMemcpy(host_out, d, bytes, hipMemcpyDeviceToHost) == hipSuccess ? FF_OK : FF_ERR_DEVICE;
}
#endif

#ifdef FF_MFMA_DIAG
// Diagnostic build only: as ff_debug_mfma_stamps, for pair_common_small_kernel ([workgroup][8] 100 MHz ticks).
int ff_debug_small_stamps(unsigned long long *host_out, int64_t n_workgroups)
{
    static unsigned long long *d = nullptr;
    const size_t bytes = (size_t)n_workgroups * 8 * sizeof(unsigned long long);
    if (!host_out) {
        if (hipMalloc(&d, bytes) != hipSuccess || hipMemset(d, 0, bytes) != hipSuccess) return FF_ERR_DEVICE;
        return hipMemcpyToSymbol(HIP_SYMBOL(g_small_stamps), &d, sizeof(d)) == hipSuccess ? FF_OK : FF_ERR_DEVICE;
    }
    if (!d) return FF_ERR_ARG;
    return hipMemcpy(host_out, d, bytes, hipMemcpyDeviceToHost) == hipSuccess ? FF_OK : FF_ERR_DEVICE;
}
#endif

}  // extern "C"

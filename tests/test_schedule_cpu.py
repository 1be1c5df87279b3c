"""Host-side work schedules of the pair kernels (frackyfrac_amd/csrc/ff_schedule.cpp),
checked on CPU through the diagnostic entry point ff_debug_schedule: every pair of the
shard and every staged branch row must be covered exactly once, whatever the sizes."""
import ctypes

import numpy as np
import pytest

import frackyfrac_amd as ff
from frackyfrac_amd import _lib as L


def schedule(kernel, n, rows, rb, re, n_cu, digits=2, narrow=1, wpw=8):
    fn = L.lib().ff_debug_schedule
    fn.restype = ctypes.c_int64
    fn.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                   ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    slots = n_cu * wpw if kernel == 0 else n_cu  # SAD: 8 (or 12) waves per CU; MFMA: one workgroup per CU
    cap = 1 << 20
    items = np.zeros((cap, 8), dtype=np.int32)
    ptr = np.zeros(slots + 1, dtype=np.int32)
    nt = ctypes.c_int64()
    k = fn(kernel, n, rows, rb, re, n_cu, digits, narrow, items.ctypes.data, cap, ptr.ctypes.data, ctypes.byref(nt))
    assert k >= 0
    return items[:k], ptr, nt.value


def pair_cover(n, rb, re):
    want = np.zeros((n, n), dtype=np.int32)
    for i in range(rb, re):
        want[i, :i] = 1
    return want


@pytest.mark.parametrize("n,rows,n_cu", [(1, 16, 4), (2, 16, 4), (33, 48, 4), (300, 4000, 8), (1000, 20000, 256),
                                         (4096, 20000, 256), (700, 64, 256), (5000, 100000, 256)])
@pytest.mark.parametrize("world", [1, 3])
@pytest.mark.parametrize("narrow", [0, 1])
@pytest.mark.parametrize("wpw", [8, 12])
def test_sad_schedule_covers_pairs_and_rows_once(monkeypatch, n, rows, n_cu, world, narrow, wpw):
    """Whatever the first level turns out to be (plain rounds, XCD-sliced rounds), with or without a second level,
    for the 8- and the 12-wave kernel: every pair of the shard in exactly one tile, every tile's branch rows covered
    once, atomics exactly on ranges that share accumulators, barrier items matched within a workgroup, loads level."""
    monkeypatch.setenv("FF_WAVES_PER_WG", str(wpw))
    for rank in range(world):
        rb, re = ff.shard_rows(n, rank, world)
        items, ptr, n_tiles = schedule(0, n, rows, rb, re, n_cu, narrow=narrow, wpw=wpw)
        assert ptr[0] == 0 and ptr[-1] == len(items) and np.all(np.diff(ptr) >= 0)
        tiles = {}
        for i0, j0, k0, k1, flags, *_ in items:
            assert k0 % 16 == 0 and (k1 % 16 == 0 or k1 == rows) and 0 <= k0 < k1 <= rows
            assert i0 % 32 == 0 and j0 % 256 == 0
            tiles.setdefault((i0, j0, (flags >> 2) & 1), []).append((k0, k1, flags))
        assert len(tiles) == n_tiles
        cover = np.zeros((n, n), dtype=np.int32) if n <= 1000 else None
        for (i0, j0, nar), ranges in tiles.items():
            ranges.sort()
            assert ranges[0][0] == 0 and ranges[-1][1] == rows
            for a, b in zip(ranges, ranges[1:]):
                assert a[1] == b[0]                       # branch rows: no gap, no overlap
            shared = len(ranges) > 1
            assert all(bool(f & 1) == shared for _, _, f in ranges)   # atomic iff the tile is shared
            if cover is not None:
                w = 128 if nar else 256
                cover[i0:min(i0 + 32, n), j0:min(j0 + w, n)] += 1
        if cover is not None:
            want = pair_cover(n, rb, re)
            assert np.all(cover[want == 1] == 1)          # every pair of the shard in exactly one tile
        # items flagged for the in-workgroup barrier: same position, same length on all 8 waves
        for wg in range(len(ptr) // wpw):
            lists = [items[ptr[wpw * wg + w]:ptr[wpw * wg + w + 1]] for w in range(wpw)]
            depth = max(len(x) for x in lists)
            for pos in range(depth):
                flagged = [x[pos] for x in lists if len(x) > pos and (x[pos][4] & 2)]
                if flagged:
                    assert len(flagged) == wpw and len({int(f[3] - f[2]) for f in flagged}) == 1
        # balance: no wave carries more than its share plus one range
        if len(items) and rows >= 1024:   # (with a handful of rows a tile cannot be cut)
            cost = np.array([(it[3] - it[2]) * (1 if (it[4] & 4) else 2) for it in items], dtype=np.int64)
            per = np.array([cost[ptr[u]:ptr[u + 1]].sum() for u in range(len(ptr) - 1)])
            assert per.max() <= cost.sum() / len(per) * 1.6 + 2 * 64


def test_xcd_sliced_rounds_pin_branch_slices_to_xcds(monkeypatch):
    """FF_XCD_SLICES=8: in the main rounds workgroup g (XCD g mod 8) only ever sweeps branch
    slice g mod 8, and the coverage properties above still hold."""
    monkeypatch.setenv("FF_XCD_SLICES", "8")
    n, rows, n_cu = 4096, 20000, 256
    items, ptr, n_tiles = schedule(0, n, rows, 0, n, n_cu)
    part = ((rows + 7) // 8 + 15) // 16 * 16
    flagged = 0
    for wg in range(n_cu):
        for w in range(8):
            for i0, j0, k0, k1, flags, *_ in items[ptr[8 * wg + w]:ptr[8 * wg + w + 1]]:
                if flags & 2:                              # a main-round item
                    flagged += 1
                    x = wg % 8
                    assert k0 == x * part and k1 == min(rows, (x + 1) * part)
    assert flagged == 4 * 256 * 8                          # 4 rounds of 256 tiles in 8 slices
    cover = {}
    for i0, j0, k0, k1, flags, *_ in items:
        cover.setdefault((i0, j0), []).append((k0, k1))
    assert len(cover) == n_tiles
    for ranges in cover.values():
        ranges.sort()
        assert ranges[0][0] == 0 and ranges[-1][1] == rows and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    test_sad_schedule_covers_pairs_and_rows_once(monkeypatch, 1000, 20000, 256, 3, 1, 8)
    monkeypatch.setenv("FF_XCD_SLICES", "8")


@pytest.mark.parametrize("n,slabs,n_cu,digits", [(5, 3, 8, 1), (300, 60, 8, 2), (1000, 313, 256, 3), (4096, 313, 256, 2),
                                                  (777, 40, 256, 5)])
@pytest.mark.parametrize("world", [1, 2])
def test_mfma_schedule_covers_pairs_slabs_and_digits_once(n, slabs, n_cu, digits, world):
    for rank in range(world):
        rb, re = ff.shard_rows(n, rank, world)
        items, ptr, n_tiles = schedule(2, n, slabs, rb, re, n_cu, digits=digits)
        assert ptr[-1] == len(items)
        units = {}
        for i0, j0, k0, k1, d0, nd, first, _ in items:
            assert k0 % 64 == 0 and k1 % 64 == 0 and 0 <= k0 < k1 <= slabs * 64
            assert i0 % 256 == 0 and j0 % 128 == 0 and nd in (1, 2) and d0 % 2 == 0 and d0 + nd <= digits
            units.setdefault((i0, j0, d0, nd), []).append((k0, k1, first))
        groups = (digits + 1) // 2
        assert len(units) == n_tiles * groups
        firsts = {}
        for (i0, j0, d0, nd), ranges in units.items():
            ranges.sort()
            assert ranges[0][0] == 0 and ranges[-1][1] == slabs * 64
            for a, b in zip(ranges, ranges[1:]):
                assert a[1] == b[0]
            firsts[(i0, j0)] = firsts.get((i0, j0), 0) + sum(f for _, _, f in ranges)
            assert nd == min(2, digits - d0)
        assert all(v == 1 for v in firsts.values())        # W_i + W_j enters every tile exactly once
        if n <= 1000:
            cover = np.zeros((n, n), dtype=np.int32)
            for (i0, j0) in firsts:
                cover[i0:min(i0 + 256, n), j0:min(j0 + 128, n)] += 1
            want = pair_cover(n, rb, re)
            assert np.all(cover[want == 1] == 1)
        if len(items):
            per = np.array([sum(int(it[3] - it[2]) for it in items[ptr[g]:ptr[g + 1]]) for g in range(len(ptr) - 1)])
            assert per.max() - per[per > 0].min() <= 64 * max(1, slabs) if world == 1 else True


def test_mfma_remainder_is_cut_along_unit_boundaries_when_that_is_faster():
    """C3's shape: 272 tiles on 256 workgroups = one round and 16 remainder tiles of 316 slabs (79 quads;
    items are whole quads of slabs).  Equal stream-K shares of 5 quads would straddle tile boundaries in
    15 workgroups (three items = one more prologue and epilogue, about 16 slabs' worth); 16 workgroups
    per tile do not."""
    items, ptr, n_tiles = schedule(2, 4096, 316, 0, 4096, 256, digits=2)
    assert n_tiles == 272
    per = [items[ptr[g]:ptr[g + 1]] for g in range(256)]
    assert max(len(x) for x in per) == 2
    rem = [int(x[1][3] - x[1][2]) // 64 for x in per if len(x) == 2]
    assert len(rem) == 256 and set(rem) <= {16, 20}
    assert all(int(it[2]) % 256 == 0 for it in items)            # every item starts at a quad
    # few units on many workgroups, uneven division: stream-K shares stay (an aligned cut would leave
    # some units with half the workgroups of others)
    items, ptr, n_tiles = schedule(2, 4096, 316, 0, 4096, 200, digits=2)
    per = np.array([sum(int(it[3] - it[2]) for it in items[ptr[g]:ptr[g + 1]]) for g in range(200)])
    assert per.max() - np.sort(per)[4] <= 256                   # equal shares to within a quad of slabs, the last few shorter
    assert all(int(it[2]) % 256 == 0 for it in items)


def test_mfma_rounds_do_not_mix_digit_groups_and_the_remainder_is_cut_by_cost():
    """Three digits = a two-plane group and a one-plane group per tile; a one-plane sweep costs about three
    quarters of a two-plane one.  Every workgroup gets one whole unit of each group in the main rounds
    (a round that mixed them would leave 16 workgroups a quarter of a unit behind at C3's shape), and the
    remainder -- 16 units of each kind -- is shared out by cost."""
    items, ptr, n_tiles = schedule(2, 4096, 316, 0, 4096, 256, digits=3)
    assert n_tiles == 272
    cost = []
    for g in range(256):
        mine = items[ptr[g]:ptr[g + 1]]
        whole = [it for it in mine if int(it[3] - it[2]) == 316 * 64]
        assert sorted(int(it[5]) for it in whole) == [1, 2]      # nd of the whole units: one of each group
        cost.append(sum(int(it[3] - it[2]) // 64 * (4 if it[5] == 2 else 3) for it in mine))
    cost = np.array(cost)
    assert cost.max() - cost.min() <= 4 * 4 * 2                   # within two quads of slabs of each other



# ---- Bounds: what the kernels' loops can touch, item by item, against what the engine allocates ----------------
#
# Three kernels over-read by design (their prefetches run ahead of the loop's exit test).  Each replay below
# follows the kernel's own loop -- the order and the distance of its loads, nothing else -- and returns the
# highest index it touches; the allocation sizes come from the functions ff_dev_stage.hip allocates with
# (ff_debug_layout), so a change of prefetch depth, padding or item granularity on either side shows up here
# on the CPU instead of as a page fault on the GPU (round 2: a 32-slab problem read one slab past Pbits).

def layout(R):
    fn = L.lib().ff_debug_layout
    fn.restype = None
    fn.argtypes = [ctypes.c_int64, ctypes.c_void_p]
    v = np.zeros(24, dtype=np.int64)
    fn(R, v.ctypes.data)
    names = ("TILE_I TILE_J KSTEP SLACK_ROWS SAD_ROWS_AHEAD SPARSE_LIST_AHEAD SPARSE_LIST_PAD M_KSLAB M_QUAD_SLABS "
             "M_PAIRS_IN_FLIGHT M_PAD_SLABS X_VALUES_PAD sad_staged_rows sad_alloc_rows mfma_staged_slabs mfma_alloc_slabs "
             "XU_TILE_H XU_SLAB XU_JMAX XU_LEN_STEP xu_slabs xu_alloc_slabs xu_alloc_lengths _")
    return dict(zip(names.split(), (int(x) for x in v)))


def sad_item_reach(k0, k1, ks):
    """run_item<NC, KS> (ff_kernels_pair_sad.hpp): (highest vector row, highest scalar row) read."""
    hi_v = k0 + ks - 1                      # prologue: vA = rows k0 .. k0 + KS - 1
    pv = k0 + ks
    hi_s = k0                               # prologue: the scalars of row k0
    nk = k1 - k0
    assert nk % (2 * ks) == 0
    for _k in range(0, nk, 2 * ks):
        for _fill in ("vB", "vA"):          # FF_FILL in the first step of each half of the trip
            hi_v = max(hi_v, pv + ks - 1)
            pv += ks
        hi_s += 2 * ks                      # every step fetches the next row's scalars
    return hi_v, hi_s


@pytest.mark.parametrize("n,R,n_cu", [(1, 1, 4), (33, 40, 4), (300, 3999, 8), (1000, 19999, 256), (4096, 19999, 256),
                                      (700, 50, 256), (5000, 99999, 256), (16384, 19999, 256)])
@pytest.mark.parametrize("wpw", ["8", "12"])
def test_sad_items_stay_inside_the_staged_matrix(monkeypatch, n, R, n_cu, wpw):
    monkeypatch.setenv("FF_WAVES_PER_WG", wpw)
    lay = layout(R)
    rows, alloc_rows = lay["sad_staged_rows"], lay["sad_alloc_rows"]
    ld = (max(n, 1) + lay["TILE_J"] - 1) // lay["TILE_J"] * lay["TILE_J"]
    ks = lay["KSTEP"] if wpw == "8" else lay["KSTEP"] // 2
    fn = L.lib().ff_debug_schedule
    fn.restype = ctypes.c_int64
    fn.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                   ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    cap = 1 << 21
    items = np.zeros((cap, 8), dtype=np.int32)
    ptr = np.zeros(n_cu * int(wpw) + 1, dtype=np.int32)
    nt = ctypes.c_int64()
    k = fn(0, n, rows, 0, n, n_cu, 2, 1, items.ctypes.data, cap, ptr.ctypes.data, ctypes.byref(nt))
    assert k >= 0
    worst_v = worst_s = -1
    for i0, j0, k0, k1, flags, *_ in items[:k]:
        hi_v, hi_s = sad_item_reach(int(k0), int(k1), ks)
        worst_v, worst_s = max(worst_v, hi_v), max(worst_s, hi_s)
        nc = 2 if flags & 4 else 4
        assert j0 + nc * 63 + nc - 1 < ld and i0 + lay["TILE_I"] - 1 < ld     # columns: vector lanes, scalar operands
    if k:
        assert worst_v == rows + ks - 1 and worst_s == rows                    # (the model sees the over-read)
        assert worst_v < alloc_rows and worst_s < alloc_rows
    # the sparse-aware kernel replaces absent list entries by the first slack row
    assert rows < alloc_rows


def test_sparse_list_prefetch_stays_inside_the_spare_entries():
    lay = layout(100)
    pad = lay["SPARSE_LIST_PAD"]
    rng = np.random.default_rng(1)
    worst = 0
    for _ in range(2000):
        size = int(rng.integers(0, 200))
        a0 = int(rng.integers(0, size + 1))
        a1 = int(rng.integers(a0, size + 1))
        for a0_, a1_ in ((a0, a1), (a0, size), (size, size), (max(size - 1, 0), size)):
            # run_item_sparse: cur = batch(a0), nxt = batch(a0 + 4); every trip t < a1 (t += 4) reads batch(t + 8)
            hi = a0_ + 7
            t = a0_
            while t < a1_:
                hi = max(hi, t + 8 + 3)
                t += 4
            assert hi < size + pad, (size, a0_, a1_, hi)
            worst = max(worst, hi - size)
    assert worst == lay["SPARSE_LIST_AHEAD"] - 1  # reached: entries up to a1 + 10 when a1 = size


def mfma_item_reach(k0, k1, lay, table_slabs=512):
    """pair_common_mfma_kernel (ff_kernels_mfma.hpp): highest slab of presence words requested."""
    ks = lay["M_KSLAB"]
    assert k0 % (2 * ks) == 0 and (k1 - k0) % (lay["M_QUAD_SLABS"] * ks) == 0
    nslab = (k1 - k0) // ks
    hi_pair = -1
    for seg in range(0, nslab, table_slabs):
        nseg = min(table_slabs, nslab - seg)
        pair = k0 // (2 * ks) + seg // 2            # qa: pair pointer of the segment
        for _q in range(lay["M_PAIRS_IN_FLIGHT"]):  # prologue: four pairs whatever the length
            hi_pair = max(hi_pair, pair)
            pair += 1
        sl = 0
        while sl + 7 < nseg:                        # groups of eight slabs: one load_words per pair done
            for _p in range(4):
                hi_pair = max(hi_pair, pair)
                pair += 1
            sl += 8
    return 2 * hi_pair + 1


@pytest.mark.parametrize("n,R,n_cu,digits", [(5, 100, 8, 1), (300, 2047, 8, 2), (300, 2048, 8, 2), (300, 2049, 8, 2),
                                             (1000, 19999, 256, 3), (4096, 19999, 256, 2), (512, 3999, 256, 2),
                                             (777, 2500, 256, 5), (130, 39999, 256, 2), (2, 79999, 256, 2),
                                             (4096, 99999, 256, 2)])
def test_mfma_items_stay_inside_the_presence_words(n, R, n_cu, digits):
    lay = layout(R)
    slabs, alloc = lay["mfma_staged_slabs"], lay["mfma_alloc_slabs"]
    items, ptr, n_tiles = schedule(2, n, slabs, 0, n, n_cu, digits=digits)
    n8 = (n + 255) // 256 * 256
    worst = -1
    for i0, j0, k0, k1, d0, nd, first, _ in items:
        worst = max(worst, mfma_item_reach(int(k0), int(k1), lay))
        assert i0 + 255 < n8 and j0 + 127 < n8
    assert worst >= slabs                      # (the model sees the over-read: the prologue alone passes a short item's end)
    assert worst <= slabs + 2 * lay["M_PAIRS_IN_FLIGHT"] - 1
    assert worst < alloc


def test_exact64_operands_past_the_last_row_are_padded():
    lay = layout(10)
    for h in (4, 8, 10, 12, 14, 16):
        for n in (1, 63, 64, 65, 1000, 4096):
            ld = (n + 63) // 64 * 64
            i0_max = (n - 1) // h * h          # build_tiles: i0 = multiples of h below the shard's end
            assert i0_max + h - 1 < ld + lay["X_VALUES_PAD"]


def xu_tiles(n, rb, re, jmax):
    fn = L.lib().ff_debug_schedule
    fn.restype = ctypes.c_int64
    fn.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                   ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    cap = 1 << 20
    items = np.zeros((cap, 8), dtype=np.int32)
    nt = ctypes.c_int64()
    k = fn(5, n, 0, rb, re, 256, jmax, 0, items.ctypes.data, cap, None, ctypes.byref(nt))   # FF_KERNEL_EXACT_F64_UNW
    assert k >= 0 and k == nt.value
    return items[:k, :3]


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 63, 64, 65, 129, 200, 1000, 4096])
@pytest.mark.parametrize("world", [1, 3])
@pytest.mark.parametrize("jmax", [1, 2])
def test_exact_unw_tiles_cover_the_shard_once_and_stay_inside_the_staged_words(n, world, jmax):
    """pair_exact_unw_kernel: every pair (i, j < i) of the shard's rows lies in exactly one tile; a tile's column
    words (64 jn lanes from j0) and its eight row words (from i0) lie inside a slab row of xu_ld(N) samples; the widest
    tiles come first; and what the kernel requests ahead -- slab s + 1's words during slab s -- lies inside the
    padding (XU_PAD_SLABS: ff_schedule.hpp), the lengths it reads in whole slabs inside theirs."""
    lay = layout(1000)
    H, slab, step = lay["XU_TILE_H"], lay["XU_SLAB"], lay["XU_LEN_STEP"]
    ld = (max(n, 1) + 64 * lay["XU_JMAX"] - 1) // (64 * lay["XU_JMAX"]) * (64 * lay["XU_JMAX"])
    seen = np.zeros((n, n), dtype=np.int32)
    for rank in range(world):
        rb, re = ff.shard_rows(n, rank, world)
        tiles = xu_tiles(n, rb, re, jmax)
        assert all(a >= b for a, b in zip(tiles[:, 2], tiles[1:, 2]))           # widest first
        for i0, j0, jn in tiles:
            assert jn in (1, 2) and jn <= jmax and i0 % H == 0 and j0 % 64 == 0
            assert j0 + 64 * jn <= ld and i0 + H <= ld                           # one slab row holds what a wave reads of it
            for i in range(max(i0, rb), min(i0 + H, re)):
                hi = min(j0 + 64 * jn, i)
                if hi > j0:
                    seen[i, j0:hi] += 1
    want = np.tril(np.ones((n, n), dtype=np.int32), -1)
    assert np.array_equal(seen, want)
    for R in (1, 31, 32, 33, 1000, 19999):
        lay = layout(R)
        n_slabs = lay["xu_slabs"]
        assert n_slabs * slab >= R > (n_slabs - 1) * slab
        # the loop: for s < n_slabs request slab s + 1; per step the lengths [32 s + k0, 32 s + k0 + step)
        hi_slab = max(s + 1 for s in range(n_slabs))
        hi_len = max(s * slab + k0 + step - 1 for s in range(n_slabs) for k0 in range(0, slab, step))
        assert hi_slab == n_slabs                                                # (the model sees the over-read)
        assert hi_slab < lay["xu_alloc_slabs"] and hi_len < lay["xu_alloc_lengths"]


def test_row_shards_of_a_multi_gpu_run_take_xcd_sliced_rounds(monkeypatch):
    """Regression guard for round 3's finding: on shards 3 and 4 of 8 of an 11,584-sample problem the
    12-wave kernel's PLAIN thirds ran 15 and 7 % slower than sliced rounds (some XCDs far behind the others); the
    schedule therefore keeps an XCD-sliced first level unless plain rounds are estimated more than 3 % faster.  These
    shards hold 1,008 full-width tiles -- too few for a round of two sliced halves (1,024 with 8 waves per workgroup,
    1,536 with 12) -- so the first level is four sliced quarters either way, never plain halves or thirds."""
    n, rows = 11584, 20000
    for wpw, want_parts in ((8, 4), (12, 4)):
        monkeypatch.setenv("FF_WAVES_PER_WG", str(wpw))
        for rank in (3, 4):
            rb, re = ff.shard_rows(n, rank, 8)
            items, ptr, n_tiles = schedule(0, n, rows, rb, re, 256, wpw=wpw)
            first = [items[ptr[u]] for u in range(256 * wpw) if ptr[u + 1] > ptr[u] and (items[ptr[u]][4] & 2)]
            lengths = {int(it[3] - it[2]) for it in first}
            part = ((rows + want_parts - 1) // want_parts + 15) // 16 * 16
            assert lengths <= {part, rows - (want_parts - 1) * part}, (wpw, rank, lengths)
            # pinned: workgroup g (XCD g mod 8) only ever sweeps the slice of its XCD group
            gsz = 8 // want_parts
            for u in range(256 * wpw):
                if ptr[u + 1] > ptr[u] and (items[ptr[u]][4] & 2):
                    x = ((u // wpw) % 8) // gsz
                    assert int(items[ptr[u]][2]) == x * part


def test_tri_digits_cover_every_length_of_their_range():
    """pair_common_mfma_kernel's three-planes-in-one-sweep variant multiplies SIGNED digits: k = d0 + 128 d1 +
    32768 d2, d0 in [-64, 63], d1 in [-127, 128] (stored negated, the A operand carries -128), d2 in [0, 127]
    (ff_schedule.hpp tri_digits).  Every k of 0 .. TRI_KMAX must come back from its planes, every plane fit int8,
    and TRI_KMAX + 1 must not (the plan then keeps two sweeps of base-128 digits)."""
    fn = L.lib().ff_debug_tri_digits
    fn.restype = ctypes.c_int64
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    kmax = fn(None, 0, None)
    assert kmax == 63 + 128 * (128 + 256 * 127)
    k = np.arange(0, kmax + 2, dtype=np.uint32)
    out = np.zeros((k.size, 3), dtype=np.int8)
    fn(k.ctypes.data, k.size, out.ctypes.data)
    d0, nd1, d2 = (out[:, c].astype(np.int64) for c in range(3))
    back = d0 + (-128) * nd1 + 32768 * d2
    assert np.array_equal(back[:-1], k[:-1].astype(np.int64))
    assert back[-1] != kmax + 1                               # (d2 = 128 does not fit a signed byte)
    assert d0.min() == -64 and d0.max() == 63
    assert nd1[:-1].min() == -128 and nd1[:-1].max() == 127   # -d1, d1 in [-127, 128]
    assert d2[:-1].min() == 0 and d2[:-1].max() == 127
    # what an accumulator adds per branch stays within the bound TRI_MAX_ROWS is derived from
    assert np.abs(d0 - 128 * nd1)[:-1].max() <= 64 + 128 * 128


def graded_schedule(n, slabs, rb, re, n_cu, duo_from_quad):
    fn = L.lib().ff_debug_graded_schedule
    fn.restype = ctypes.c_int64
    fn.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int64,
                   ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    cap = 1 << 20
    items = np.zeros((cap, 8), dtype=np.int32)
    ptr = np.zeros(n_cu + 1, dtype=np.int32)
    nt = ctypes.c_int64()
    k = fn(n, slabs, rb, re, n_cu, duo_from_quad, items.ctypes.data, cap, ptr.ctypes.data, ctypes.byref(nt))
    assert k >= 0
    return items[:k], ptr, nt.value


@pytest.mark.parametrize("n,slabs,n_cu,duo_from", [(5, 3, 8, 0), (300, 60, 8, 4), (300, 60, 8, 15), (300, 60, 8, 99),
                                                    (1000, 313, 256, 30), (4096, 316, 256, 9), (4096, 316, 256, 70),
                                                    (4600, 20, 256, 2), (777, 1200, 256, 100)])
@pytest.mark.parametrize("world", [1, 2])
def test_graded_mfma_schedule_covers_every_slab_once_and_cuts_by_cost(n, slabs, n_cu, duo_from, world):
    """A graded sweep (rows staged by descending length: three digit planes per block up to a slab, two from there
    on) has ONE digit group per tile; its remainder is cut on an axis of cost, 12 per quad of slabs with three
    planes and 9 with two.  Coverage as for any matrix-core schedule; the workgroups' costs agree to within an item's
    granularity; and every item's reach stays inside the staged arrays."""
    lay = layout(slabs * 64)
    assert lay["mfma_staged_slabs"] >= slabs
    quads = (slabs + 3) // 4
    tri_quads = min(duo_from, quads)
    for rank in range(world):
        rb, re = ff.shard_rows(n, rank, world)
        items, ptr, n_tiles = graded_schedule(n, slabs, rb, re, n_cu, duo_from)
        assert ptr[-1] == len(items)
        units, cost = {}, np.zeros(n_cu)
        for g in range(n_cu):
            for i0, j0, k0, k1, d0, nd, first, _ in items[ptr[g]:ptr[g + 1]]:
                assert k0 % 256 == 0 and 0 <= k0 < k1 <= slabs * 64 and (k1 % 256 == 0 or k1 == slabs * 64)
                assert d0 == 0 and nd == 2                     # (the way out adds X + (Y << 15): two accumulator sets)
                units.setdefault((int(i0), int(j0)), []).append((int(k0), int(k1), int(first)))
                q0, q1 = k0 // 256, (k1 + 255) // 256
                cost[g] += 12 * max(0, min(q1, tri_quads) - q0) + 9 * max(0, q1 - max(q0, tri_quads))
                assert mfma_item_reach(int(k0), (int(k1) + 255) // 256 * 256, lay) < lay["mfma_alloc_slabs"]
        assert len(units) == n_tiles
        for ranges in units.values():
            ranges.sort()
            assert ranges[0][0] == 0 and ranges[-1][1] == slabs * 64
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            assert sum(f for _, _, f in ranges) == 1 and ranges[0][2] == 1
        if world == 1 and n_tiles >= n_cu:
            # whole rounds + a remainder cut by cost: nobody carries more than a round's share plus one quad and an
            # item's overhead more than anybody else
            busy = cost[cost > 0]
            assert busy.max() - busy.min() <= 12 * 2 + 32

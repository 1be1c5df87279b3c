"""Host-side work schedules of the pair kernels (frackyfrac_amd/csrc/ff_schedule.cpp),
checked on CPU through the diagnostic entry point ff_debug_schedule: every pair of the
shard and every staged branch row must be covered exactly once, whatever the sizes."""
import ctypes

import numpy as np
import pytest

import frackyfrac_amd as ff
from frackyfrac_amd import _lib as L


def schedule(kernel, n, rows, rb, re, n_cu, digits=2, narrow=1):
    fn = L.lib().ff_debug_schedule
    fn.restype = ctypes.c_int64
    fn.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                   ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    slots = n_cu * 8 if kernel == 0 else n_cu  # SAD: 8 waves per CU; MFMA: one workgroup per CU
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
def test_sad_schedule_covers_pairs_and_rows_once(n, rows, n_cu, world, narrow):
    for rank in range(world):
        rb, re = ff.shard_rows(n, rank, world)
        items, ptr, n_tiles = schedule(0, n, rows, rb, re, n_cu, narrow=narrow)
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
        for wg in range(len(ptr) // 8):
            lists = [items[ptr[8 * wg + w]:ptr[8 * wg + w + 1]] for w in range(8)]
            depth = max(len(x) for x in lists)
            for pos in range(depth):
                flagged = [x[pos] for x in lists if len(x) > pos and (x[pos][4] & 2)]
                if flagged:
                    assert len(flagged) == 8 and len({int(f[3] - f[2]) for f in flagged}) == 1
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
    test_sad_schedule_covers_pairs_and_rows_once(1000, 20000, 256, 3, 1)


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


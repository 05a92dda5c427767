"""The bridge-attention forward's work list (libra_amd/csrc/attention_bridge.hip, round 5) restated in numpy: a workgroup's key tiles
become UNITS (tile, operand variant), per wave of 32 query rows each skipped / plain / masked.  The test pins what the
kernel relies on: over all units of a wave, every (query, key) pair the reference attends to (causal, inside [start, len), real
query) is covered EXACTLY once and with the operand variant the closed form asks for (cross iff the two tokens' modalities differ:
modeling_libra.py:364-370 / :282-293), a plain unit has no pair to mask, and the staging protocol's tile sets cover every
variant some wave reads (every unit is staged, and only units somebody computes exist)."""
import numpy as np
import pytest

BQ, BKV = 256, 64


def bits_below(n):
    return (1 << 64) - 1 if n >= 64 else (0 if n <= 0 else (1 << n) - 1)


def plan_block(flag, S, length, start, qt):
    """-> per wave: list of units (kt, var, mode) in the workgroup's unit order (a unit = one key tile in ONE operand variant; it exists
    iff some wave has a pair of that kind in the tile), and the block-level tile sets."""
    n32 = (S + 31) // 32
    kmask = [0] * (2 * 64 + 2)
    for t in range(n32):
        for i in range(32):
            tok = t * 32 + i
            if tok < S and flag[tok]:
                kmask[t] |= 1 << i
    kend = min((qt + 1) * BQ, S)
    nkt = (kend + BKV - 1) // BKV
    per_wave = []
    same_blk = cross_blk = 0
    for wave in range(8):
        q0w = qt * BQ + wave * 32
        active = q0w < S
        rows = [r for r in range(q0w, q0w + 32) if r < S]
        wV = any(flag[r] for r in rows)
        wL = any(not flag[r] for r in rows)
        tiles = []
        for kt in range(64):
            kv0 = kt * BKV
            mm = kmask[2 * kt] | (kmask[2 * kt + 1] << 32)
            rng = bits_below(length - kv0) & ~bits_below(start - kv0)
            kV, kL = (mm & rng) != 0, (~mm & rng & ((1 << 64) - 1)) != 0
            inn = active and kt < nkt and kv0 <= q0w + 31
            wsame, wcross = inn and ((wL and kL) or (wV and kV)), inn and ((wL and kV) or (wV and kL))
            full = kv0 + BKV - 1 <= q0w and kv0 >= start and kv0 + BKV <= length
            plain = full and not (wsame and wcross)
            tiles.append(((1 if plain else 2) if wsame else 0, (1 if plain else 2) if wcross else 0))
            if wsame: same_blk |= 1 << kt
            if wcross: cross_blk |= 1 << kt
        per_wave.append(tiles)
    units = []
    for wave in range(8):
        lst = []
        for kt in range(nkt):
            ms, mc = per_wave[wave][kt]
            if (same_blk >> kt) & 1: lst.append((kt, 0, ms))
            if (cross_blk >> kt) & 1: lst.append((kt, 1, mc))
        units.append(lst)
    return units, nkt, same_blk, cross_blk, kmask


def element_valid(kmask, flag, q, key, kt, var, length, start):
    """apply_mask's predicate for one (query row, key) of a MASKED unit."""
    kv0 = kt * BKV
    lo = 0 if q < start else start
    hi = min(q, length - 1)
    rng = bits_below(hi - kv0 + 1) & ~bits_below(lo - kv0)
    j = key - kv0
    kbit = (kmask[2 * kt + (j >> 5)] >> (j & 31)) & 1
    cross = kbit != (1 if flag[q] else 0)
    return bool((rng >> j) & 1) and (cross == bool(var))


@pytest.mark.parametrize("case", ["bench", "two_spans", "random", "left_pad", "right_pad", "ragged", "text_only"])
def test_units_cover_every_attended_pair_once_with_the_right_variant(case):
    rs = np.random.RandomState(hash(case) % 1000)
    S, length, start = 2048, 2048, 0
    flag = np.zeros(S, dtype=bool)
    if case == "bench":
        flag[1:579] = True
    elif case == "two_spans":
        flag[5:583] = True; flag[900:1478] = True
    elif case == "random":
        S = length = 700; flag = rs.rand(S) < 0.4
    elif case == "left_pad":
        S = length = 1024; start = 133; flag = np.zeros(S, dtype=bool); flag[start + 1:start + 579] = True
    elif case == "right_pad":
        S = 1024; length = 801; flag = np.zeros(S, dtype=bool); flag[1:579] = True
    elif case == "ragged":
        S = length = 1000; flag = np.zeros(S, dtype=bool); flag[300:878] = True
    n_qt = (S + BQ - 1) // BQ
    for qt in range(n_qt):
        units, nkt, same_blk, cross_blk, kmask = plan_block(flag, S, length, start, qt)
        assert all([x[:2] for x in u] == [x[:2] for x in units[0]] for u in units)     # one unit list / barrier schedule per workgroup
        assert len(units[0]) == bin(same_blk).count("1") + bin(cross_blk).count("1") <= 128
        assert any(m for u in units for (_, _, m) in u) or not len(units[0])
        for i in range(len(units[0])):                                                 # no unit that nobody computes
            assert any(units[w][i][2] for w in range(8))
        for wave in range(8):
            q0w = qt * BQ + wave * 32
            cover = {}
            for (kt, var, mode) in units[wave]:
                if mode == 0:
                    continue
                for q in range(q0w, min(q0w + 32, S)):
                    for key in range(kt * BKV, min(kt * BKV + BKV, S)):
                        if mode == 1:
                            ok = True                                           # plain: no test of any kind in the kernel
                        else:
                            ok = element_valid(kmask, flag, q, key, kt, var, length, start)
                        if ok:
                            assert (q, key) not in cover, ("pair covered twice", q, key)
                            cover[(q, key)] = var
            for q in range(q0w, min(q0w + 32, S)):
                if not (start <= q < length):
                    continue                                                    # padding query rows: output unused
                for key in range(0, S):
                    want = key <= q and start <= key < length
                    if want:
                        assert cover.get((q, key)) == int(flag[q] != flag[key]), ("missing / wrong variant", q, key)
                    else:
                        assert (q, key) not in cover, ("pair should be masked", q, key)


# ---- the backward's dK / dV pass (attention_bridge_bwd.hip, "dkv6"): item = (128-key block, variant), unit = 64-query tile -------------
KV_KEYS, QT = 128, 64


def plan_dkv_item(flag, S, length, kb, var):
    """-> per wave (4 key sub-blocks; the dV and the dK wave of a pair share it): list of units (query tile, mode); the kernel's lane-parallel
    classification restated tile by tile."""
    n32 = (S + 31) // 32
    qmask = [0] * (2 * 64 + 2)
    for t in range(n32):
        for i in range(32):
            tok = t * 32 + i
            if tok < S and flag[tok]:
                qmask[t] |= 1 << i
    length = min(length, S)
    nqt = (S + QT - 1) // QT
    per_wave, uset = [], 0
    for ksub in range(4):
        kbase = kb * KV_KEYS + ksub * 32
        keys = [k for k in range(kbase, kbase + 32) if k < S and k < length]
        wkV, wkL = any(flag[k] for k in keys), any(not flag[k] for k in keys)
        tiles = []
        for t in range(64):
            q0 = t * QT
            qm = qmask[2 * t] | (qmask[2 * t + 1] << 32)
            rng = bits_below(S - q0)
            qV, qL = (qm & rng) != 0, (~qm & rng & ((1 << 64) - 1)) != 0
            wsame, wcross = (qL and wkL) or (qV and wkV), (qL and wkV) or (qV and wkL)
            inn = t < nqt and q0 + 63 >= kbase and kbase < length
            has = inn and (wcross if var else wsame)
            full = q0 >= kbase + 31 and q0 + 64 <= S and kbase + 32 <= length
            plain = full and not (wsame and wcross)
            tiles.append((1 if plain else 2) if has else 0)
            if has: uset |= 1 << t
        per_wave.append(tiles)
    return [[(t, per_wave[w][t]) for t in range(64) if (uset >> t) & 1] for w in range(4)], qmask


def dkv_element_valid(qmask, flag, q, key, t, var, S, length):
    q0 = t * QT
    rng = bits_below(S - q0) & ~bits_below(key - q0)
    if not (key < S and key < min(length, S)):
        rng = 0
    j = q - q0
    qbit = (qmask[2 * t + (j >> 5)] >> (j & 31)) & 1
    cross = qbit != (1 if flag[key] else 0)
    return bool((rng >> j) & 1) and (cross == bool(var))


@pytest.mark.parametrize("case", ["bench", "two_spans", "random", "right_pad", "ragged", "text_only"])
def test_dkv_items_cover_every_attended_pair_once(case):
    """Over the two variant items of a key block, every (query, key) pair the reference attends to (query >= key, key < len) is covered
    exactly once, by the item of the variant the closed form asks for; a plain unit has no pair to mask; every unit of an item is
    computed by some wave; an item without units is exactly a block with no pair of that variant (its gradients are zero)."""
    rs = np.random.RandomState(hash(case) % 1000)
    S, length = 1024, 1024
    flag = np.zeros(S, dtype=bool)
    if case == "bench":
        flag[1:579] = True
    elif case == "two_spans":
        flag[5:200] = True; flag[500:800] = True
    elif case == "random":
        S = length = 700; flag = rs.rand(S) < 0.4
    elif case == "right_pad":
        length = 801; flag[1:579] = True
    elif case == "ragged":
        S = length = 1000; flag = np.zeros(S, dtype=bool); flag[300:878] = True
    for kb in range((S + KV_KEYS - 1) // KV_KEYS):
        cover = {}
        for var in (0, 1):
            units, qmask = plan_dkv_item(flag, S, length, kb, var)
            assert all([t for t, _ in u] == [t for t, _ in units[0]] for u in units)      # one unit list / barrier schedule per workgroup
            for i in range(len(units[0])):
                assert any(units[w][i][1] for w in range(4))                               # no unit that nobody computes
            for ksub in range(4):
                kbase = kb * KV_KEYS + ksub * 32
                for (t, mode) in units[ksub]:
                    if mode == 0:
                        continue
                    for key in range(kbase, min(kbase + 32, S)):
                        for q in range(t * QT, min(t * QT + QT, S)):
                            ok = True if mode == 1 else dkv_element_valid(qmask, flag, q, key, t, var, S, length)
                            if ok:
                                assert (q, key) not in cover, ("pair covered twice", q, key)
                                cover[(q, key)] = var
        for key in range(kb * KV_KEYS, min(kb * KV_KEYS + KV_KEYS, S)):
            for q in range(S):
                if q >= key and key < length:
                    assert cover.get((q, key)) == int(flag[q] != flag[key]), ("missing / wrong variant", q, key)
                else:
                    assert (q, key) not in cover, ("pair should be masked", q, key)


# ---- the persistent kernels' static schedules (attention_bridge.hip, attention_bridge_bwd.hip, runtime.hip persistent_grid, gemm_bf16_256.hip) --------
def persistent_grid(nitems, period, cus=256, budget=0):
    if 0 < budget < cus:
        cus = budget
    nblk = cus // period * period
    if nblk < period:
        nblk = period
    return min(nblk, nitems)


def xcd_remap(bid, nblk):
    q, r = nblk >> 3, nblk & 7
    xcd, j = bid & 7, bid >> 3
    return (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + j


@pytest.mark.parametrize("B,H,S", [(8, 32, 2048), (2, 160, 512), (1, 1, 16), (3, 5, 700), (2, 32, 4096), (1, 7, 300)])
@pytest.mark.parametrize("budget", [0, 224, 24])
def test_persistent_schedules_visit_every_item_exactly_once(B, H, S, budget):
    """Forward / dQ pass: item i = w + k P -> (sequence-head i / n_qt, query block n_qt - 1 - (i + k) mod n_qt); dK/dV pass: item0 = w + k P ->
    (item0 / per_bh) per_bh + (item0 mod per_bh + k) mod per_bh with per_bh = 2 n_kb; persistent GEMM: tile w + k P.  For every grid the
    launchers can produce (CU budget or not) the map must be a bijection onto the items, and in the balanced case (P a multiple of the
    period, the training shape) every workgroup must meet every weight class equally often."""
    n_qt = (S + 255) // 256
    nitems = B * H * n_qt
    P = persistent_grid(nitems, n_qt, budget=budget)
    assert P % n_qt == 0 or P == nitems
    seen = []
    per_wg = {}
    for blk in range(P):
        w = xcd_remap(blk, P)
        k, item = 0, w
        while item < nitems:
            qt = n_qt - 1 - ((item % n_qt + k) % n_qt)
            seen.append((item // n_qt, qt))
            per_wg.setdefault(w, []).append(qt)
            k += 1; item += P
    assert sorted(seen) == [(bh, qt) for bh in range(B * H) for qt in range(n_qt)]
    if nitems % P == 0 and (nitems // P) % n_qt == 0:            # e.g. B=8 H=32 S=2048: 8 steps, every workgroup sees each block once
        for w, qts in per_wg.items():
            assert sorted(qts) == sorted(list(range(n_qt)) * (len(qts) // n_qt)), (w, qts)
    # dK/dV pass
    n_kb = (S + 127) // 128
    per_bh = 2 * n_kb
    nit = B * H * per_bh
    P = persistent_grid(nit, per_bh, budget=budget)
    assert P % per_bh == 0 or P == nit
    seen = []
    for blk in range(P):
        w = xcd_remap(blk, P)
        k, item0 = 0, w
        while item0 < nit:
            seen.append((item0 // per_bh) * per_bh + (item0 % per_bh + k) % per_bh)
            k += 1; item0 += P
    assert sorted(seen) == list(range(nit))
    # persistent GEMM: tiles in dispatch order
    ntiles = 560
    P = persistent_grid(ntiles, 1, budget=budget)
    assert sorted(t for w in range(P) for t in range(w, ntiles, P)) == list(range(ntiles))

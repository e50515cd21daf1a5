#!/usr/bin/env python3
"""Instruction proxy for the Lawson-flip stage of pattern_build_kernel (model first, then HIP).

Replays tools/delaunay_model.py's rounds on device-like patterns (P pilots on the 14 x 599 lattice) and counts, per round, the work
items each phase of the kernel executes; wave-level passes follow from the kernel's schedule (stage 0: four waves, triangle
t -> thread t mod 256, items of a wave packed 64 to a pass by wave_balanced; stage 1 from <= 192 active triangles on: one wave).
Vector instructions per pass are the kernel's own (PMC by round and phase, profiles/r03_pmc_pattern_rounds_P838.txt and the static
counts of the three lambdas): test pass 260 (3 in-circle slots), rewrite pass 140, relink pass 310, round overhead per wave 160.

Variants priced with the same constants:
  current      items = active triangles, every pass evaluates all three edge slots
  edge items   items = (triangle, edge) pairs that really need a test (one in-circle per item: 110 per pass); next round's items are
               emitted by the relink phase (outer edges of flipped triangles) + candidates that lost their claim
  + no diagonal  the same without re-testing the diagonal a flip just created (locally Delaunay by construction)
  ideal        every pass full (items / 64), no per-round overhead: the floor of ANY schedule of rounds of independent flips
The proxy reproduces the measured 53.7 k for `current` to a few per cent; the question it answers is whether a rewrite of the flip
loop could bring the kernel's 93.8 k vector instructions to <= 60 k (flips <= ~20 k)."""
import sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
import delaunay_model as M

TEST_PASS, REWRITE_PASS, RELINK_PASS, ROUND_OVERHEAD, EDGE_PASS = 260, 140, 310, 160, 110
SOLO_AT = 192


def passes_stage0(item_owner_threads):
    """wave-level passes of wave_balanced over 4 waves: per wave ceil(items / 64) (levels packed), at least the busiest lane's share"""
    n = 0
    for w in range(4):
        k = int(np.sum((item_owner_threads >> 6) == w))
        n += -(-k // 64) if k else 0
    return n


def run(P, seed):
    S, nsc = 14, 599
    rs = np.random.RandomState(seed)
    idx = np.sort(rs.permutation(S * nsc)[:P])
    px, py = (idx // nsc).astype(np.int64), (idx % nsc).astype(np.int64)
    tri, nbr = M.initial_triangulation(px, py)
    n = len(tri)
    active = np.ones(n, dtype=bool)
    fresh_diag = np.zeros((n, 3), dtype=bool)
    cost = {"current": 0.0, "edge": 0.0, "edge_nodiag": 0.0, "ideal": 0.0}
    stats = []
    rnd = 0
    while True:
        # phase 0 as the kernel does it: active triangles test the edges that are theirs to test
        cand, tests, tests_nodiag = [], 0, 0
        claim = np.full(n, 1 << 60, dtype=np.int64)
        for t in np.nonzero(active)[0]:
            for i in range(3):
                u = nbr[t, i]
                if u == M.NONE or (u < t and active[u]):
                    continue
                tests += 1
                if not fresh_diag[t, i]:
                    tests_nodiag += 1
                a, b, c = tri[t, i], tri[t, (i + 1) % 3], tri[t, (i + 2) % 3]
                j = [k for k in range(3) if tri[u, k] not in (b, c)][0]
                d = tri[u, j]
                if M.incircle(px[a], py[a], px[b], py[b], px[c], py[c], px[d], py[d]) > 0:
                    key = t * 4 + i
                    cand.append((key, t, i, u, j))
                    claim[t] = min(claim[t], key); claim[u] = min(claim[u], key)
        if not cand:
            break
        rnd += 1
        nact = int(active.sum())
        solo = nact <= SOLO_AT and rnd > 1
        owners = np.nonzero(active)[0] % 256
        prop = np.unique([t for _, t, _, _, _ in cand])
        flipped = np.zeros(n, dtype=bool)
        partner = np.full(n, M.NONE, dtype=np.int64)
        fresh_diag[:] = False
        winners = []
        for key, t, i, u, j in cand:
            if claim[t] != key or claim[u] != key:
                continue
            a, b, c, d = tri[t, i], tri[t, (i + 1) % 3], tri[t, (i + 2) % 3], tri[u, j]
            n_ab, n_ca = nbr[t, (i + 2) % 3], nbr[t, (i + 1) % 3]
            n_db, n_cd = nbr[u, (j + 1) % 3], nbr[u, (j + 2) % 3]
            tri[t] = (a, b, d); nbr[t] = (n_db, u, n_ab)
            tri[u] = (a, d, c); nbr[u] = (n_cd, n_ca, t)
            flipped[t] = flipped[u] = True
            partner[t], partner[u] = u, t
            fresh_diag[t, 1] = fresh_diag[u, 2] = True
            winners.append(t)
        for X in np.nonzero(flipped)[0]:
            for k in range(3):
                Y = nbr[X, k]
                if Y == M.NONE or Y == partner[X]:
                    continue
                x, y = tri[X, (k + 1) % 3], tri[X, (k + 2) % 3]
                if flipped[Y]:
                    if not (x in tri[Y] and y in tri[Y]):
                        Y = partner[Y]
                    nbr[X, k] = Y
                else:
                    k2 = [q for q in range(3) if tri[Y, q] not in (x, y)][0]
                    nbr[Y, k2] = X
        nflip = len(winners)
        waves = 1 if solo else 4
        if solo:
            p_test, p_rew, p_rel = -(-nact // 64), -(-len(prop) // 64), -(-nflip // 64)
            p_edge, p_edge_nd = -(-tests // 64), -(-max(tests_nodiag, 1) // 64)
        else:
            p_test = passes_stage0(owners)
            p_rew = passes_stage0(prop % 256)
            p_rel = passes_stage0(np.asarray(winners) % 256) if winners else 0
            p_edge = max(4, -(-tests // 64) + 2)          # items spread over four waves: ~half a pass of rounding per wave
            p_edge_nd = max(4, -(-tests_nodiag // 64) + 2)
        common = p_rew * REWRITE_PASS + p_rel * RELINK_PASS + waves * ROUND_OVERHEAD
        cost["current"] += p_test * TEST_PASS + common
        cost["edge"] += p_edge * EDGE_PASS + common
        cost["edge_nodiag"] += p_edge_nd * EDGE_PASS + common
        cost["ideal"] += tests_nodiag / 64 * EDGE_PASS + len(prop) / 64 * REWRITE_PASS + nflip / 64 * RELINK_PASS
        stats.append((nact, tests, tests_nodiag, len(prop), nflip))
        active = flipped | np.isin(np.arange(n), [t for _, t, _, _, _ in cand]) | np.isin(np.arange(n), [u for _, _, _, u, _ in cand])
        active &= (claim < (1 << 60))
    return cost, stats


if __name__ == "__main__":
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 838
    npat = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    tot = {}
    allstats = []
    for s in range(npat):
        c, st = run(P, 100 + s)
        allstats.append(st)
        for k, v in c.items():
            tot[k] = tot.get(k, 0.0) + v / npat
    print(f"P = {P}, {npat} patterns, mean per pattern")
    nr = max(len(s) for s in allstats)
    rows = np.zeros((nr, 5))
    for st in allstats:
        for i, r in enumerate(st):
            rows[i] += np.asarray(r) / npat
    print("round  active-triangles  edge-tests  (w/o new diagonals)  proposers  flips")
    for i, r in enumerate(rows):
        print(f"{i + 1:5d}  {r[0]:16.0f}  {r[1]:10.0f}  {r[2]:19.0f}  {r[3]:9.0f}  {r[4]:5.0f}")
    print(f"totals: edge tests {rows[:, 1].sum():.0f} ({rows[:, 2].sum():.0f} without new diagonals), flips {rows[:, 4].sum():.0f} = {rows[:, 4].sum() / P:.2f} per pilot")
    other = 93.8 - 53.7
    print("vector-instruction proxy of the flip stage (k instructions per pattern) and of the whole kernel (+ %.1f k for the other phases, PMC r03):" % other)
    for k, name in (("current", "current schedule"), ("edge", "edge items"), ("edge_nodiag", "edge items, no new-diagonal tests"), ("ideal", "ideal packing, no round overhead")):
        print(f"  {name:36s} flips {tot[k] / 1e3:6.1f} k   kernel {tot[k] / 1e3 + other:6.1f} k")

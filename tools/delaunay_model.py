#!/usr/bin/env python3
"""CPU model of the on-device Delaunay construction for lattice pilots (csrc/pattern_kernels.hip follows it step by step):

  1. initial triangulation: pilots lie on S parallel rows; every strip between two consecutive occupied rows is triangulated
     by merging the two sorted rows by edge midpoint = their two-row Delaunay triangulation (rank arithmetic only: every triangle and its three neighbours are computed
     independently), and the pockets between the left / right end chains and the convex hull are filled by a monotone-chain
     scan (a handful of triangles);
  2. Lawson flips with the EXACT integer in-circle test until every interior edge is locally Delaunay, in rounds of
     independent flips (a flip must win a min-claim on its two triangles and their four outer neighbours).

Used by tests (validity of the device result is checked with the same exact predicates) and to count rounds / flips.
x = symbol (row), y = subcarrier; all predicates in Python / NumPy int64 (|det| < 2^45).
"""
import numpy as np

NONE = -1


def orient(ax, ay, bx, by, cx, cy):
    return (bx - ax) * (cy - ay) - (by - ay) * (cx - ax)


def incircle(ax, ay, bx, by, cx, cy, dx, dy):
    """> 0 iff d lies strictly inside the circumcircle of the counter-clockwise triangle (a, b, c)."""
    adx, ady, bdx, bdy, cdx, cdy = ax - dx, ay - dy, bx - dx, by - dy, cx - dx, cy - dy
    return ((adx * adx + ady * ady) * (bdx * cdy - cdx * bdy) - (bdx * bdx + bdy * bdy) * (adx * cdy - cdx * ady)
            + (cdx * cdx + cdy * cdy) * (adx * bdy - bdx * ady))


def initial_triangulation(px, py, nsc=None):
    """px, py: int arrays sorted by (px, py).  Returns (tri [n,3] counter-clockwise vertex ids, nbr [n,3] with nbr[t][i] the
    triangle across the edge opposite vertex i, NONE on the hull); with ``nsc`` (the grid's column count) also region[n]: the
    column region 0..3 the device numbers the triangle into (a strip step by the doubled midpoint column of its row edge
    against the thresholds b * 2 nsc / 4; left pockets 0, right pockets 3)."""
    P = len(px)
    region = []
    rows = np.unique(px)
    start = {r: int(np.searchsorted(px, r, "left")) for r in rows}
    end = {r: int(np.searchsorted(px, r, "right")) for r in rows}
    tri, edge_owner = [], {}

    def add(a, b, c):
        # orientation: make counter-clockwise in (x, y)
        if orient(px[a], py[a], px[b], py[b], px[c], py[c]) < 0:
            b, c = c, b
        assert orient(px[a], py[a], px[b], py[b], px[c], py[c]) > 0
        tri.append((a, b, c))

    for ra, rb in zip(rows[:-1], rows[1:]):
        i, j = start[ra], start[rb]
        ia, jb = end[ra] - 1, end[rb] - 1
        while i < ia or j < jb:
            # the Delaunay triangulation of the two rows alone: advance the row whose next edge has its midpoint further left
            # (ties: the upper row); exhausted rows cannot advance
            if j >= jb or (i < ia and py[i] + py[i + 1] <= py[j] + py[j + 1]):
                esum = py[i] + py[i + 1]
                add(i, i + 1, j); i += 1
            else:
                esum = py[j] + py[j + 1]
                add(j, j + 1, i); j += 1
            if nsc is not None:
                region.append(sum(1 for b in (1, 2, 3) if esum >= (b * 2 * nsc) >> 2))
    # pockets: monotone chain over the first (left) and last (right) point of every occupied row
    for side in (0, 1):
        chain = [start[r] if side == 0 else end[r] - 1 for r in rows]
        stack = []
        for p in chain:
            while len(stack) >= 2:
                a, b = stack[-2], stack[-1]
                o = orient(px[a], py[a], px[b], py[b], px[p], py[p])
                # b lies inside the hull (in a pocket) if it is on the inner side of a -> p: larger column than the segment on
                # the left chain (o < 0), smaller on the right chain (o > 0); it is then cut off by triangle (a, b, p).
                # Collinear b (o == 0) stays a flat hull vertex.
                if (o < 0 if side == 0 else o > 0):
                    add(a, b, p)
                    region.append(0 if side == 0 else 3)
                    stack.pop()
                else:
                    break
            stack.append(p)
    tri = np.asarray(tri, dtype=np.int64).reshape(-1, 3)
    nbr = np.full(tri.shape, NONE, dtype=np.int64)
    for t, (a, b, c) in enumerate(tri):
        for i, (u, v) in enumerate(((b, c), (c, a), (a, b))):
            key = (min(u, v), max(u, v))
            if key in edge_owner:
                t2, i2 = edge_owner.pop(key)
                nbr[t, i] = t2; nbr[t2, i2] = t
            else:
                edge_owner[key] = (t, i)
    if nsc is not None:
        return tri, nbr, np.asarray(region, dtype=np.int64)
    return tri, nbr


def flip_rounds(px, py, tri, nbr, max_rounds=100000):
    """In-place Lawson flips in rounds of independent flips (no triangle takes part in two flips of a round: min-claim on
    the two triangles of the edge).  A round has three barrier-separated phases, as on the device:
      0  every interior edge (t < u) failing the exact in-circle test claims t and u with its key (atomic min);
      1  winners rewrite their two triangles and note the OLD ids of the four outer neighbours;
      2  every new triangle resolves each outer neighbour id -- the triangle across that edge is the old neighbour or, if that
         one flipped in this round too, its partner -- and writes the back-pointer of neighbours that did not flip.
    Returns (rounds, flips)."""
    rounds = flips = 0
    n = len(tri)
    while rounds < max_rounds:
        cand = []
        claim = np.full(n, 1 << 60, dtype=np.int64)
        for t in range(n):
            for i in range(3):
                u = nbr[t, i]
                if u == NONE or u < t:
                    continue
                a, b, c = tri[t, i], tri[t, (i + 1) % 3], tri[t, (i + 2) % 3]
                j = [k for k in range(3) if tri[u, k] not in (b, c)][0]
                d = tri[u, j]
                if incircle(px[a], py[a], px[b], py[b], px[c], py[c], px[d], py[d]) > 0:
                    key = t * 4 + i
                    cand.append((key, t, i, u, j))
                    claim[t] = min(claim[t], key); claim[u] = min(claim[u], key)
        if not cand:
            break
        rounds += 1
        flipped = np.zeros(n, dtype=bool)
        partner = np.full(n, NONE, dtype=np.int64)
        for key, t, i, u, j in cand:                                   # phase 1
            if claim[t] != key or claim[u] != key:
                continue
            a, b, c, d = tri[t, i], tri[t, (i + 1) % 3], tri[t, (i + 2) % 3], tri[u, j]
            n_ab, n_ca = nbr[t, (i + 2) % 3], nbr[t, (i + 1) % 3]
            # u = (d, c, b) rotated so that u[j] = d: the edge opposite c is (b, d), the edge opposite b is (d, c)
            n_db, n_cd = nbr[u, (j + 1) % 3], nbr[u, (j + 2) % 3]
            tri[t] = (a, b, d); nbr[t] = (n_db, u, n_ab)
            tri[u] = (a, d, c); nbr[u] = (n_cd, n_ca, t)
            flipped[t] = flipped[u] = True
            partner[t], partner[u] = u, t
            flips += 1
        for X in np.nonzero(flipped)[0]:                               # phase 2
            for k in range(3):
                Y = nbr[X, k]
                if Y == NONE or Y == partner[X]:
                    continue
                x, y = tri[X, (k + 1) % 3], tri[X, (k + 2) % 3]
                if flipped[Y]:
                    if not (x in tri[Y] and y in tri[Y]):
                        Y = partner[Y]
                    assert x in tri[Y] and y in tri[Y]
                    nbr[X, k] = Y
                else:
                    k2 = [q for q in range(3) if tri[Y, q] not in (x, y)][0]
                    nbr[Y, k2] = X
    return rounds, flips


def flip_rounds_staged(px, py, tri, nbr, region):
    """Model of a schedule that was built on the device in round 3 and DROPPED ("region-local flips"): stage 1 -- each
    of four waves runs flip_rounds' rounds on its own column region and flips an edge only if its two triangles and their four outer
    neighbours are the wave's, marking triangles with an edge it could not decide; stage 2 -- one wave owns everything and finishes
    from the marked triangles.  It reaches a valid Delaunay triangulation (tested), but the cascades blocked at the region boundaries
    restart in stage 2: about as many rounds again.  (What the device runs -- all waves while the active set is large, one wave for
    the tail, same rules throughout -- gives exactly flip_rounds' triangulation.)  Only ACTIVE triangles (changed or claimed in the
    previous round) test their edges, as on the device.
    Returns ([rounds of wave 0..3], rounds of stage 2, flips of stage 1, flips of stage 2)."""
    n = len(tri)
    marked = np.zeros(n, dtype=bool)

    def run(me, active):
        def mine(x):
            return x == NONE or me is None or region[x] == me
        rounds = flips = 0
        while True:
            cand = []
            claim = np.full(n, 1 << 60, dtype=np.int64)
            for t in np.nonzero(active)[0]:
                for i in range(3):
                    u = nbr[t, i]
                    if u == NONE:
                        continue
                    if not mine(u):
                        marked[t] = True
                        continue
                    if u < t and active[u]:
                        continue
                    a, b, c = tri[t, i], tri[t, (i + 1) % 3], tri[t, (i + 2) % 3]
                    j = [k for k in range(3) if tri[u, k] not in (b, c)][0]
                    d = tri[u, j]
                    if incircle(px[a], py[a], px[b], py[b], px[c], py[c], px[d], py[d]) > 0:
                        if not all(mine(x) for x in (nbr[t, (i + 1) % 3], nbr[t, (i + 2) % 3], nbr[u, (j + 1) % 3], nbr[u, (j + 2) % 3])):
                            marked[t] = True
                            continue
                        key = t * 4 + i
                        cand.append((key, t, i, u, j))
                        claim[t] = min(claim[t], key); claim[u] = min(claim[u], key)
            if not cand:
                return rounds, flips
            rounds += 1
            flipped = np.zeros(n, dtype=bool)
            partner = np.full(n, NONE, dtype=np.int64)
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
                flips += 1
            for X in np.nonzero(flipped)[0]:
                for k in range(3):
                    Y = nbr[X, k]
                    if Y == NONE or Y == partner[X]:
                        continue
                    x, y = tri[X, (k + 1) % 3], tri[X, (k + 2) % 3]
                    if flipped[Y]:
                        if not (x in tri[Y] and y in tri[Y]):
                            Y = partner[Y]
                        nbr[X, k] = Y
                    else:
                        k2 = [q for q in range(3) if tri[Y, q] not in (x, y)][0]
                        nbr[Y, k2] = X
            active = claim < (1 << 60)

    local = [run(w, region == w) for w in range(4)]
    r2, f2 = run(None, marked.copy())
    return [r for r, _ in local], r2, sum(f for _, f in local), f2


def is_delaunay(px, py, tri, nbr):
    """Exact checks: positive orientation everywhere, symmetric adjacency, every interior edge locally Delaunay (which
    implies the global empty-circle property), and the triangle areas sum to the hull's area."""
    for t in range(len(tri)):
        a, b, c = tri[t]
        if orient(px[a], py[a], px[b], py[b], px[c], py[c]) <= 0:
            return False
        for i in range(3):
            u = nbr[t, i]
            if u == NONE:
                continue
            bb, cc = tri[t, (i + 1) % 3], tri[t, (i + 2) % 3]
            js = [k for k in range(3) if tri[u, k] not in (bb, cc)]
            if len(js) != 1 or nbr[u, js[0]] != t:
                return False
            d = tri[u, js[0]]
            if incircle(px[a if i == 0 else tri[t, i]], py[tri[t, i]], px[bb], py[bb], px[cc], py[cc], px[d], py[d]) > 0:
                return False
    return True


if __name__ == "__main__":
    import sys
    import time
    S, nsc = 14, 599
    for dens in (0.01, 0.05, 0.1, 0.3):
        G = S * nsc
        P = int(G * dens)
        rs = np.random.RandomState(1)
        idx = np.sort(rs.permutation(G)[:P])
        px, py = (idx // nsc).astype(np.int64), (idx % nsc).astype(np.int64)
        t0 = time.time()
        tri, nbr = initial_triangulation(px, py)
        r, f = flip_rounds(px, py, tri, nbr)
        ok = is_delaunay(px, py, tri, nbr)
        from scipy.spatial import Delaunay
        ref = Delaunay(np.column_stack([px, py]))
        print(f"density {dens}: P {P} triangles {len(tri)} (scipy {len(ref.simplices)}) rounds {r} flips {f} ({f / P:.1f} per pilot) valid {ok}  {time.time() - t0:.1f}s", flush=True)
        if len(sys.argv) > 1:
            break

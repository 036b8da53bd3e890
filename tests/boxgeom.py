"""Independent geometry of two oriented boxes (numpy only) — a checker for the project's own box-box manifold (oracle
`orc_box_box`, device `c_box_box`: 15-axis SAT + incident face clipped by the reference face), which MuJoCo's `mjc_BoxBox`
(reached through `mj_step1`, /root/reference/src/mj_main.cpp:83) cannot be restated from without the library.

Nothing here shares a formula with the routines under test: the separating-axis intervals come from projecting the sixteen
vertices, the gap of a contact from casting the line `pos + t n` through both boxes (slab test), the manifold from a
Sutherland-Hodgman clip of the incident face polygon.  What is asserted of every contact set (VERDICT r03, next #6a):

  * contacts exist  iff  no axis separates the boxes by more than the margin
  * the normal is unit, points from box 1 to box 2 and is one of the 15 axis candidates; its overlap is the smallest one up to the
    documented edge handicap (an edge axis must win by 5 %)
  * every point lies within half its gap (or depth) of both boxes (it sits midway), and `dist` is the gap of the two surfaces along
    the normal THROUGH that point (line cast)
  * no point is deeper than the interval overlap along the normal
  * the point set is the clipped polygon's vertex set (face case), one point (edge case)
"""
import numpy as np

SIGNS = np.array([[a, b, c] for a in (-1, 1) for b in (-1, 1) for c in (-1, 1)], dtype=np.float64)
EDGE_HANDICAP = 0.05      # orc_box_box / c_box_box: an edge axis is chosen only if its separation beats the best face axis by 5 %


def vertices(p, R, s):
    return p + (SIGNS * s) @ R.T


def axes15(R1, R2):
    """unit candidate axes: 3 + 3 face normals, up to 9 edge x edge (parallel edges give none); second return: kind per axis
    (0..2 face of box 1, 3..5 face of box 2, 6 + 3 i + j edge pair)"""
    ax, kind = [], []
    for i in range(3):
        ax.append(R1[:, i]); kind.append(i)
    for j in range(3):
        ax.append(R2[:, j]); kind.append(3 + j)
    for i in range(3):
        for j in range(3):
            c = np.cross(R1[:, i], R2[:, j]); l = np.linalg.norm(c)
            if l * l >= 1e-6:
                ax.append(c / l); kind.append(6 + 3 * i + j)
    return np.array(ax), np.array(kind)


def separation(axis, V1, V2):
    """signed separation of the vertex sets' projection intervals along `axis` (negative: they overlap by that much)"""
    a, b = V1 @ axis, V2 @ axis
    return max(a.min() - b.max(), b.min() - a.max())


def line_cast(x, n, p, R, s):
    """parameter interval [t_in, t_out] of the line x + t n inside the box (slab test); empty: t_in > t_out"""
    o, d = R.T @ (x - p), R.T @ n
    tin, tout = -np.inf, np.inf
    for k in range(3):
        if abs(d[k]) < 1e-14:
            if abs(o[k]) > s[k]:
                return np.inf, -np.inf
            continue
        t1, t2 = (-s[k] - o[k]) / d[k], (s[k] - o[k]) / d[k]
        tin, tout = max(tin, min(t1, t2)), min(tout, max(t1, t2))
    return tin, tout


def inside(x, p, R, s, pad):
    return bool(np.all(np.abs(R.T @ (x - p)) <= s + pad))


def clip_polygon(poly, planes):
    """Sutherland-Hodgman: poly [k, 3] against half spaces a . x <= b"""
    for a, b in planes:
        if len(poly) == 0:
            break
        out = []
        d = poly @ a - b
        for i in range(len(poly)):
            j = (i + 1) % len(poly)
            if d[i] <= 0:
                out.append(poly[i])
            if (d[i] < 0 < d[j]) or (d[j] < 0 < d[i]):
                out.append(poly[i] + d[i] / (d[i] - d[j]) * (poly[j] - poly[i]))
        poly = np.array(out).reshape(-1, 3)
    return poly


def face_polygon(p, R, s, k, sign):
    """the four corners of face `sign * axis k` of a box, in cyclic order"""
    u, v = (k + 1) % 3, (k + 2) % 3
    c = p + sign * s[k] * R[:, k]
    return np.array([c + a * s[u] * R[:, u] + b * s[v] * R[:, v] for a, b in ((1, 1), (-1, 1), (-1, -1), (1, -1))])


def expected_manifold(n, ref, inc, margin):
    """Face case, independent construction: the face of `inc` most opposed to the reference normal, clipped by the four side
    planes of the reference face; a vertex whose gap to the reference face plane is below the margin gives the point midway.
    `n`: the reference face's outward unit normal (towards inc).  Returns (points [k, 3], gaps [k])."""
    pr, Rr, sr = ref; pi, Ri, si = inc
    kr = int(np.argmax(np.abs(Rr.T @ n))); sg = 1.0 if Rr[:, kr] @ n > 0 else -1.0
    ki = int(np.argmax(np.abs(Ri.T @ n))); si_sign = -1.0 if Ri[:, ki] @ n > 0 else 1.0
    poly = face_polygon(pi, Ri, si, ki, si_sign)
    planes = []
    for k in ((kr + 1) % 3, (kr + 2) % 3):
        for sgn in (1.0, -1.0):
            a = sgn * Rr[:, k]
            planes.append((a, a @ pr + sr[k]))
    poly = clip_polygon(poly, planes)
    nr = sg * Rr[:, kr]
    pts, gaps = [], []
    for x in poly:
        g = nr @ (x - pr) - sr[kr]
        if g < margin:
            pts.append(x - 0.5 * g * nr); gaps.append(g)
    return np.array(pts).reshape(-1, 3), np.array(gaps)


def dedupe(P, tol):
    keep = []
    for x in P:
        if not any(np.linalg.norm(x - y) <= tol for y in keep):
            keep.append(x)
    return np.array(keep).reshape(-1, 3)


def check_contacts(box1, box2, margin, dist, pos, n, tol=1e-9, count_tol=None):
    """Returns a list of violation strings (empty: the contact set passes).  box = (p[3], R[3,3] columns = axes, s[3]);
    dist[k], pos[k,3], n[3] = what the routine under test reported (k may be 0).  `tol`: absolute length tolerance (fp64 oracle
    1e-9; fp32 device ~1e-5 at unit scale); `count_tol`: a polygon vertex closer than this to a clip plane or to the margin
    threshold may be present or absent."""
    p1, R1, s1 = box1; p2, R2, s2 = box2
    count_tol = 10 * tol if count_tol is None else count_tol
    V1, V2 = vertices(p1, R1, s1), vertices(p2, R2, s2)
    A, kind = axes15(R1, R2)
    sep = np.array([separation(a, V1, V2) for a in A])
    bad = []
    k = len(dist)
    if sep.max() > margin + tol:
        if k:
            bad.append(f"{k} contacts although axis {kind[sep.argmax()]} separates by {sep.max():.3e} > margin {margin}")
        return bad
    if k == 0:
        if sep.max() < margin - tol:
            bad.append(f"no contact although no axis separates by more than {sep.max():.3e} (margin {margin})")
        return bad
    n = np.asarray(n, float)
    if abs(np.linalg.norm(n) - 1) > 10 * tol:
        bad.append(f"|n| = {np.linalg.norm(n)}")
    # orientation 1 -> 2: along the normal box 2's interval lies above box 1's
    if (V2 @ n).mean() < (V1 @ n).mean() - tol:
        bad.append("normal points from box 2 to box 1")
    cosang = A @ n
    c = int(np.argmax(np.abs(cosang)))
    # (an edge cross product can coincide with a face normal — boxes sharing an axis: among equally close candidates the face kind counts)
    dvec = np.linalg.norm(A * np.sign(cosang)[:, None] - n, axis=1)
    near = np.nonzero(dvec <= dvec.min() + max(1e-12, 0.01 * tol))[0]
    c = int(near[np.argmin(kind[near])])
    if 1 - abs(cosang[c]) > max(1e-10, 100 * tol * tol) and np.linalg.norm(A[c] * np.sign(cosang[c]) - n) > 10 * tol:
        bad.append(f"normal is none of the 15 axis candidates (closest: kind {kind[c]}, |cos| = {abs(cosang[c]):.9f})")
        return bad
    sn = separation(n, V1, V2)
    face = kind < 6
    best_face = sep[face].max()
    edge_mask = ~face
    if kind[c] < 6:
        # a face axis: the best face axis, unless an edge axis was allowed to take over
        if sn < best_face - tol:
            bad.append(f"face normal with separation {sn:.6e}, another face axis has {best_face:.6e}")
        if edge_mask.any() and sep[edge_mask].max() > best_face + EDGE_HANDICAP * abs(best_face) + 1e-9 + tol:
            bad.append(f"face normal although an edge axis leads by more than the handicap: {sep[edge_mask].max():.6e} vs {best_face:.6e}")
    else:
        if sn < sep[edge_mask].max() - tol:
            bad.append(f"edge normal with separation {sn:.6e}, another edge axis has {sep[edge_mask].max():.6e}")
        if sn < best_face + EDGE_HANDICAP * abs(best_face) + 1e-9 - tol:
            bad.append(f"edge normal without the handicap's lead: {sn:.6e} vs face {best_face:.6e}")
        if k != 1:
            bad.append(f"edge case with {k} points")
    for q in range(k):
        x, dq = np.asarray(pos[q], float), float(dist[q])
        if dq > margin + tol:
            bad.append(f"point {q}: dist {dq:.3e} beyond the margin")
        if dq < sn - tol:
            bad.append(f"point {q}: dist {dq:.6e} deeper than the interval overlap {sn:.6e}")
        pad = 0.5 * abs(dq) + tol      # (midway: at most half the gap, or half the depth, from either surface)
        if not (inside(x, p1, R1, s1, pad) and inside(x, p2, R2, s2, pad)):
            bad.append(f"point {q} outside a box inflated by half its gap")
        a_in, a_out = line_cast(x, n, p1, R1, s1 + tol)
        b_in, b_out = line_cast(x, n, p2, R2, s2 + tol)
        if a_in > a_out or b_in > b_out:
            bad.append(f"point {q}: the line along the normal misses a box")
            continue
        a_out -= tol; b_in += tol       # (undo the inflation along the line: both surfaces are crossed head-on or obliquely; the
        gap = b_in - a_out              #  inflation only widens what counts as a hit, the check below carries its own tolerance)
        lt = 4 * tol / max(1e-3, min(abs(n @ R1[:, int(np.argmax(np.abs(R1.T @ n)))]), abs(n @ R2[:, int(np.argmax(np.abs(R2.T @ n)))])))
        if abs(gap - dq) > lt:
            bad.append(f"point {q}: dist {dq:.6e}, surfaces along the normal through it are {gap:.6e} apart")
        if abs(b_in + a_out) > 2 * lt:
            bad.append(f"point {q} is not midway between the surfaces: {a_out:.3e} behind, {b_in:.3e} ahead")
    if kind[c] < 6:
        ref, inc, nref = ((box1, box2, n) if kind[c] < 3 else (box2, box1, -n))
        # (faces of both boxes parallel to n: either choice of the reference gives the same clipped polygon)
        E, G = expected_manifold(nref, ref, inc, margin)
        E = dedupe(E, count_tol)
        P = np.asarray(pos, float).reshape(-1, 3)
        for q in range(k):
            if len(E) == 0 or np.min(np.linalg.norm(E - P[q], axis=1)) > count_tol:
                bad.append(f"point {q} is not a vertex of the clipped incident face")
        Pd = dedupe(P, count_tol)
        # robust expectation: vertices that survive when the clip region and the margin shrink by count_tol
        Es, _ = expected_manifold(nref, (ref[0], ref[1], ref[2] - np.array([0 if i == int(np.argmax(np.abs(ref[1].T @ nref))) else count_tol for i in range(3)])), inc, margin - count_tol)
        # (a vertex that moves by more than 50 x the perturbation is ill-conditioned — an edge crossing a side at a shallow angle —
        #  and may be present or absent)
        need = [e for e in E if len(Es) and np.min(np.linalg.norm(Es - e, axis=1)) <= 50 * count_tol]
        for e in need:
            if len(Pd) == 0 or np.min(np.linalg.norm(Pd - e, axis=1)) > 3 * count_tol:
                bad.append(f"clipped-polygon vertex {np.round(e, 6)} missing from the manifold ({len(Pd)} reported, {len(E)} expected)")
        if len(Pd) > len(E):
            bad.append(f"{len(Pd)} distinct points, the clipped polygon has {len(E)} vertices within the margin")
    return bad


def random_rotation(rng, max_angle=np.pi):
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    a = rng.uniform(0, max_angle)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


def random_pairs(rng, n, smin=0.05, smax=0.125):
    """box pairs in the regimes the scenes visit: arbitrary poses with deep / shallow / no overlap, resting stacks (faces
    nearly parallel, penetration of a fraction of a millimetre), edge-on-edge crossings"""
    out = []
    for i in range(n):
        s1, s2 = rng.uniform(smin, smax, 3), rng.uniform(smin, smax, 3)
        mode = i % 4
        if mode == 0:      # arbitrary
            R1, R2 = random_rotation(rng), random_rotation(rng)
            d = rng.normal(size=3); d /= np.linalg.norm(d)
            p1 = rng.uniform(-1, 1, 3); p2 = p1 + d * rng.uniform(0.3, 1.15) * (np.linalg.norm(s1) + np.linalg.norm(s2)) * 0.6
        elif mode == 1:    # resting stack: box 2 on a face of box 1, small tilt, tiny penetration or gap
            R1 = random_rotation(rng); R2 = R1 @ random_rotation(rng, 10.0 ** rng.uniform(-4, -1))
            if rng.random() < 0.5:
                R2 = R2 @ random_rotation(rng)[:, [1, 2, 0]] if False else R2 @ np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]]) ** 1
            k = rng.integers(3); p1 = rng.uniform(-1, 1, 3)
            off = rng.uniform(-0.6, 0.6, 3) * (s1 + s2); off[k] = s1[k] + s2[k] - 10.0 ** rng.uniform(-5, -2.5) * rng.choice([1, 1, 1, -0.1])
            p2 = p1 + R1 @ off
        elif mode == 2:    # rotated about the stacking axis (octagon manifolds), moderate penetration
            R1 = random_rotation(rng); k = rng.integers(3)
            a = rng.uniform(0, np.pi); ax = np.zeros(3); ax[k] = 1
            K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
            R2 = R1 @ (np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K) @ random_rotation(rng, 10.0 ** rng.uniform(-3, -1))
            p1 = rng.uniform(-1, 1, 3); off = rng.uniform(-0.4, 0.4, 3) * (s1 + s2); off[k] = (s1[k] + s2[k]) * rng.uniform(0.9, 1.02)
            p2 = p1 + R1 @ off
        else:              # edge on edge
            R1 = random_rotation(rng, 0.3) @ np.array([[1, 0, 0], [0, np.sqrt(.5), -np.sqrt(.5)], [0, np.sqrt(.5), np.sqrt(.5)]])
            R2 = random_rotation(rng, 0.3) @ np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]]) @ np.array([[1, 0, 0], [0, np.sqrt(.5), -np.sqrt(.5)], [0, np.sqrt(.5), np.sqrt(.5)]])
            s1 = np.array([rng.uniform(0.2, 0.4), *rng.uniform(smin, smax, 2)]); s2 = np.array([rng.uniform(0.2, 0.4), *rng.uniform(smin, smax, 2)])
            p1 = rng.uniform(-1, 1, 3)
            h = (np.hypot(s1[1], s1[2]) + np.hypot(s2[1], s2[2])) * rng.uniform(0.8, 1.05)
            p2 = p1 + np.array([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), h])
        out.append(((p1, R1, s1), (p2, R2, s2)))
    return out

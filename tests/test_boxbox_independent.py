"""The project's own box-box manifold (oracle `orc_box_box`; the device's `c_box_box` is checked the same way in
tests/test_gpu_round4.py) against geometry that shares nothing with it: tests/boxgeom.py — vertex-projection separating axes, line
casts through both boxes, a Sutherland-Hodgman clip of the incident face.  MuJoCo's own routine (mjc_BoxBox under mj_step1,
/root/reference/src/mj_main.cpp:83) is not available to compare with; this pins what CAN be pinned: that the manifold
is geometrically right — contacts iff no separating axis, least-overlap normal, every point midway between the two surfaces with
`dist` the true gap through it, the point set = the clipped incident face."""
import numpy as np

import boxgeom as bg
from test_oracle_collision import box_box

TOL = 1e-8        # fp64 routine with its own 1e-9 guard on the axis cosines


def test_ten_thousand_random_pairs_pass_the_independent_check():
    rng = np.random.default_rng(20260929)
    pairs = bg.random_pairs(rng, 5000)
    ncon = nface = nedge = 0
    for margin in (0.0, 0.002):
        for i, (b1, b2) in enumerate(pairs):
            k, dist, pos, n = box_box(b1[0], b1[1], b1[2], b2[0], b2[1], b2[2], margin)
            bad = bg.check_contacts(b1, b2, margin, dist, pos, n, tol=TOL)
            assert not bad, f"pair {i} (family {i % 4}, margin {margin}, {k} points): {bad[:4]}"
            ncon += k; nface += k > 1; nedge += k == 1
    # the sample exercises every case: thousands of face manifolds, single points, separated pairs
    assert ncon > 25000 and nface > 5000 and nedge > 1500


def test_the_checker_is_not_vacuous():
    """every kind of defect the checker claims to see, planted into correct contact sets, is reported"""
    rng = np.random.default_rng(7)
    seen = dict(shift=0, depth=0, normal=0, drop=0, extra=0, flip=0)
    tried = 0
    for b1, b2 in bg.random_pairs(rng, 400):
        k, dist, pos, n = box_box(b1[0], b1[1], b1[2], b2[0], b2[1], b2[2], 0.0)
        if k < 3:
            continue
        tried += 1
        assert not bg.check_contacts(b1, b2, 0.0, dist, pos, n, tol=TOL)
        t = np.cross(n, [1.0, 0, 0]); t /= np.linalg.norm(t)
        p2 = pos.copy(); p2[0] += 1e-4 * t                       # a point slid along the surface
        seen["shift"] += bool(bg.check_contacts(b1, b2, 0.0, dist, p2, n, tol=TOL))
        d2 = dist.copy(); d2[1] -= 1e-5                          # a wrong depth
        seen["depth"] += bool(bg.check_contacts(b1, b2, 0.0, d2, pos, n, tol=TOL))
        n2 = n + 1e-3 * t; n2 /= np.linalg.norm(n2)              # a tilted normal
        seen["normal"] += bool(bg.check_contacts(b1, b2, 0.0, dist, pos, n2, tol=TOL))
        seen["drop"] += bool(bg.check_contacts(b1, b2, 0.0, dist[1:], pos[1:], n, tol=TOL))                     # a missing point
        seen["extra"] += bool(bg.check_contacts(b1, b2, 0.0, np.r_[dist, dist.mean()], np.vstack([pos, pos.mean(0)]), n, tol=TOL))   # an interior point
        seen["flip"] += bool(bg.check_contacts(b1, b2, 0.0, dist, pos, -n, tol=TOL))                          # normal from 2 to 1
    assert tried > 150
    for what, cnt in seen.items():
        assert cnt == tried, f"planted defect '{what}' went unnoticed in {tried - cnt} of {tried} contact sets"

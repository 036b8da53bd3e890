"""Narrow-phase known answers for the oracle's primitives (CPU).  The box-box routine is this
project's own definition (15-axis SAT + reference/incident-face manifold), shared by the HIP path;
MuJoCo's mjc_BoxBox point selection is not reproducible without the library."""
import ctypes as C

import numpy as np

import mujoco_sim_amd as ms
import orc
from helpers import D, free_body_model, set_opt


def box_box(p1, R1, s1, p2, R2, s2, margin=0.0):
    L = orc.lib()
    a = lambda x: np.ascontiguousarray(x, dtype=np.float64)
    p1, R1, s1, p2, R2, s2 = map(a, (p1, R1, s1, p2, R2, s2))
    dist = np.zeros(8); pos = np.zeros(24); n = np.zeros(3)
    P = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    k = L.orc_box_box(P(p1), P(R1.reshape(-1)), P(s1), P(p2), P(R2.reshape(-1)), P(s2), margin, P(dist), P(pos), P(n))
    return k, dist[:k], pos.reshape(8, 3)[:k], n


def rotz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


def rot(axis, a):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


def test_box_box_separated_and_touching():
    I = np.eye(3)
    k, *_ = box_box([0, 0, 0], I, [1, 1, 1], [2.5, 0, 0], I, [1, 1, 1])
    assert k == 0
    k, dist, pos, n = box_box([0, 0, 0], I, [1, 1, 1], [0, 0, 1.9], I, [0.5, 0.5, 1])
    assert k >= 4 and np.allclose(dist, -0.1) and np.allclose(n, [0, 0, 1])
    assert np.allclose(pos[:, 2], 0.95)                       # midway between the two faces
    assert np.abs(pos[:, :2]).max() <= 0.5 + 1e-12            # manifold = the smaller face
    corners = {(round(x, 6), round(y, 6)) for x, y, _ in pos}
    assert {(0.5, 0.5), (-0.5, 0.5), (-0.5, -0.5), (0.5, -0.5)} <= corners


def test_box_box_rotated_face_manifold_has_up_to_eight_points():
    I = np.eye(3)
    k, dist, pos, n = box_box([0, 0, 0], I, [1, 1, 1], [0, 0, 1.95], rotz(np.pi / 4), [1, 1, 1])
    assert k == 8 and np.allclose(n, [0, 0, 1]) and np.allclose(dist, -0.05)
    # octagon = intersection of the square with its 45-degree copy: every point on both boundaries
    r = np.abs(pos[:, :2])
    assert np.allclose(np.maximum(r[:, 0], r[:, 1]), 1.0, atol=1e-9)
    assert len({tuple(np.round(p, 6)) for p in pos}) == 8


def test_box_box_normal_points_from_first_to_second_whichever_is_reference():
    I = np.eye(3)
    big, small = [1, 1, 1], [0.2, 0.2, 0.2]
    for p1, s1, p2, s2 in (([0, 0, 0], big, [0, 0, 1.15], small), ([0, 0, 1.15], small, [0, 0, 0], big)):
        k, dist, pos, n = box_box(p1, I, s1, p2, I, s2)
        d = np.asarray(p2, float) - np.asarray(p1, float)
        assert k >= 4 and n @ d > 0 and np.allclose(dist, -0.05)


def test_box_box_edge_edge_single_point():
    # two long thin boxes crossing like a plus sign, one rotated about x so that an edge points down
    RA = rot([1, 0, 0], np.pi / 4)                       # long along x, rolled 45 deg: top/bottom are edges
    RB = rotz(np.pi / 2) @ rot([1, 0, 0], np.pi / 4)     # same, long axis turned to y
    s = [1.0, 0.1, 0.1]
    half_diag = 0.1 * np.sqrt(2)
    k, dist, pos, n = box_box([0, 0, 0], RA, s, [0, 0, 2 * half_diag - 0.01], RB, s)
    assert k == 1 and abs(dist[0] + 0.01) < 1e-9
    assert np.allclose(n, [0, 0, 1], atol=1e-9) and np.allclose(pos[0], [0, 0, half_diag - 0.005], atol=1e-9)


def test_box_box_symmetry_under_rigid_motion():
    rng = np.random.default_rng(5)
    for _ in range(50):
        R1, R2 = rot(rng.normal(size=3), rng.uniform(0, 3)), rot(rng.normal(size=3), rng.uniform(0, 3))
        s1, s2 = rng.uniform(0.05, 0.125, 3), rng.uniform(0.05, 0.125, 3)
        p2 = rng.normal(size=3) * 0.12
        k, dist, pos, n = box_box([0, 0, 0], R1, s1, p2, R2, s2)
        if k == 0:
            continue
        T, t = rot(rng.normal(size=3), rng.uniform(0, 3)), rng.normal(size=3)
        k2, dist2, pos2, n2 = box_box(t, T @ R1, s1, T @ p2 + t, T @ R2, s2)
        assert k2 == k
        np.testing.assert_allclose(dist2, dist, atol=1e-9)
        np.testing.assert_allclose(pos2, pos @ T.T + t, atol=1e-9)
        np.testing.assert_allclose(n2, T @ n, atol=1e-9)
        assert abs(np.linalg.norm(n) - 1) < 1e-12 and (dist <= 0).all()


def _contacts_of(lib, build):
    b = lib.mjh_builder_create()
    set_opt(lib, b, timestep=0.005)
    lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, 0.05), None, None, None, -1, -1, -1, -1)
    build(b)
    m = ms.Model(lib.mjh_builder_compile(b), lib)
    lib.mjh_builder_destroy(b)
    d = orc.OrcData(m.ptr)
    d.call("kinematics"); d.call("collision")
    return m, d, d.contacts()


def _free(lib, b, name, gtype, size, pos, quat=None):
    bd = lib.mjh_builder_add_body(b, name, 0, D(*pos), D(*quat) if quat else None, 0.0)
    lib.mjh_builder_add_joint(b, None, bd, 0, None, None, None, 0, 0, 0, 0, 0)
    lib.mjh_builder_add_geom(b, None, bd, gtype, D(*size), None, None, None, -1, -1, -1, -1)
    return bd


def test_plane_primitives(lib):
    m, d, c = _contacts_of(lib, lambda b: _free(lib, b, b"s", 2, (0.1, 0, 0), (0, 0, 0.08)))
    assert len(c) == 1 and abs(c[0]["dist"] + 0.02) < 1e-12 and np.allclose(c[0]["pos"], [0, 0, -0.01])
    assert np.allclose(c[0]["frame"][:3], [0, 0, 1]) and c[0]["geom"] == (0, 1)
    # capsule lying on its side: two end contacts
    q = [np.cos(np.pi / 4), 0, np.sin(np.pi / 4), 0]
    m, d, c = _contacts_of(lib, lambda b: _free(lib, b, b"c", 3, (0.05, 0.2, 0), (0, 0, 0.04), q))
    assert len(c) == 2 and np.allclose([x["dist"] for x in c], -0.01)
    assert sorted(round(x["pos"][0], 6) for x in c) == [-0.2, 0.2]
    # box tilted on an edge: two corner contacts; flat: four
    m, d, c = _contacts_of(lib, lambda b: _free(lib, b, b"b", 6, (0.1, 0.1, 0.1), (0, 0, 0.095)))
    assert len(c) == 4 and np.allclose([x["dist"] for x in c], -0.005)
    q = [np.cos(np.pi / 8), np.sin(np.pi / 8), 0, 0]
    m, d, c = _contacts_of(lib, lambda b: _free(lib, b, b"b", 6, (0.1, 0.1, 0.1), (0, 0, 0.1 * np.sqrt(2) - 0.003), q))
    assert len(c) == 2 and np.allclose([x["dist"] for x in c], -0.003, atol=1e-9)


def test_sphere_pairs_and_frames(lib):
    def build(b):
        _free(lib, b, b"a", 2, (0.1, 0, 0), (0, 0, 1.0))
        _free(lib, b, b"b", 2, (0.15, 0, 0), (0.2, 0, 1.0))
        _free(lib, b, b"c", 6, (0.1, 0.1, 0.1), (0, 0.18, 1.0))
    m, d, c = _contacts_of(lib, build)
    pairs = {x["geom"]: x for x in c}
    ss = pairs[(1, 2)]
    assert abs(ss["dist"] + 0.05) < 1e-12 and np.allclose(ss["frame"][:3], [1, 0, 0]) and np.allclose(ss["pos"], [0.075, 0, 1.0])
    F = ss["frame"].reshape(3, 3)
    assert np.allclose(F @ F.T, np.eye(3), atol=1e-12) and np.linalg.det(F) > 0
    sb = pairs[(1, 3)]          # sphere (type 2) is geom1, box geom2: normal from sphere towards the box
    assert abs(sb["dist"] + 0.02) < 1e-12 and np.allclose(sb["frame"][:3], [0, 1, 0])


def test_contact_parameter_mixing_and_rows(lib):
    """floor (condim 4, friction 2/.05/.01) x default geom -> condim 4, friction max, 6 pyramid rows"""
    m = ms.scene("s24")
    d = orc.OrcData(m.ptr)
    q = m.array("qpos0").copy(); q[2] = 0.085
    d.set_qpos(q); d.call("fwd_position")
    cons = d.contacts()
    floor = [c for c in cons if c["geom"][0] == 0]
    assert len(floor) == 4 and all(c["dim"] == 4 for c in floor)
    assert d.i("nefc") >= 24
    t = d.ifield("efc_type")
    assert (t[:24] == 6).all()
    # pyramid rows: J_n +- mu J_t ; row pairs differ by 2 mu J_t, sums give 2 J_n
    J = d.f("efc_J").reshape(d.i("nefc"), m.nv)
    Jn = 0.5 * (J[0] + J[1])
    np.testing.assert_allclose(0.5 * (J[2] + J[3]), Jn, atol=1e-12)
    assert abs(Jn[2] - 1.0) < 1e-12                    # normal row lifts the box along +z
    np.testing.assert_allclose(np.abs(0.5 * (J[0] - J[1]))[:3].max(), 2.0, atol=1e-12)   # mu1 = 2 on a unit tangent
    R = d.f("efc_R")
    assert np.allclose(R[:6], R[0]) and R[0] > 0


def test_plane_cylinder_contacts_and_rest(lib):
    """plane - cylinder (mjc_PlaneCylinder structure): standing -> a triangle of 3 points under the cap, lying -> the two
    ends of the line of touch; both rest with sum of normal forces = m g."""
    import orc
    from helpers import D, set_opt
    for quat, expect in (((1, 0, 0, 0), 3), ((np.cos(np.pi / 4), np.sin(np.pi / 4), 0, 0), 2)):
        b = lib.mjh_builder_create(); set_opt(lib, b, timestep=0.002)
        lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, 0.05), None, None, None, -1, -1, -1, -1)
        r, h = 0.06, 0.10
        z0 = (h if expect == 3 else r) - 1e-4
        bd = lib.mjh_builder_add_body(b, b"cyl", 0, D(0, 0, z0), D(*quat), 0.0)
        lib.mjh_builder_add_joint(b, None, bd, 0, None, None, None, 0, 0, 0, 0, 0)
        lib.mjh_builder_add_geom(b, None, bd, 5, D(r, h, 0), None, None, None, -1, -1, -1, -1)
        import mujoco_sim_amd as ms
        m = ms.Model(lib.mjh_builder_compile(b), lib); lib.mjh_builder_destroy(b)
        mass = m.array("body_mass")[1]
        np.testing.assert_allclose(mass, 1000 * np.pi * r * r * 2 * h, rtol=1e-12)
        d = orc.OrcData(m.ptr)
        d.call("forward")
        cons = d.contacts()
        assert len(cons) == expect, (expect, cons)
        for c in cons:
            np.testing.assert_allclose(c["frame"][:3], [0, 0, 1], atol=1e-12)
            assert abs(c["pos"][2]) < 1e-3 and c["dist"] < 0
            assert np.hypot(c["pos"][0], c["pos"][1]) <= r + (h if expect == 2 else 0) + 1e-9
        if expect == 3:      # the three points span the cap: centroid on the axis
            np.testing.assert_allclose(np.mean([c["pos"][:2] for c in cons], axis=0), [0, 0], atol=1e-9)
        d.step(1500)
        assert np.abs(d.f("qvel")).max() < 2e-3
        np.testing.assert_allclose(d.f("qpos")[2], z0 + 1e-4, atol=2e-3)


# ---- generic convex pairs (MPR over support mappings): cylinder-x, capsule-box, ellipsoid-x
SPH, CAP, ELL, CYL, BOX = 2, 3, 4, 5, 6


def convex_pair(t1, p1, R1, s1, t2, p2, R2, s2, margin=0.0):
    L = orc.lib()
    a = lambda x: np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))
    p1, R1, s1, p2, R2, s2 = map(a, (p1, R1, s1, p2, R2, s2))
    d = np.zeros(1); pos = np.zeros(3); n = np.zeros(3)
    P = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    k = L.orc_convex_pair(t1, P(p1), P(R1), P(s1), t2, P(p2), P(R2), P(s2), margin, P(d), P(pos), P(n))
    return k, d[0], pos, n


def test_convex_reproduces_analytic_pairs():
    I = np.eye(3)
    # sphere - sphere through the generic routine = the analytic pair (origin on the centre ray)
    c2 = np.array([1.5, 0.2, 0.1]); l = np.linalg.norm(c2)
    k, d, pos, n = convex_pair(SPH, [0, 0, 0], I, [1, 0, 0], SPH, c2, I, [1, 0, 0])
    assert k == 1
    np.testing.assert_allclose(d, l - 2, atol=1e-9)
    np.testing.assert_allclose(n, c2 / l, atol=1e-9)
    np.testing.assert_allclose(pos, c2 / l * (1 + 0.5 * (l - 2)), atol=1e-9)
    # axis-aligned boxes overlapping by 0.1 along x
    k, d, pos, n = convex_pair(BOX, [0, 0, 0], I, [1, 1, 1], BOX, [1.9, 0.3, 0.2], I, [1, 1, 1])
    assert k == 1
    np.testing.assert_allclose(d, -0.1, atol=1e-7); np.testing.assert_allclose(n, [1, 0, 0], atol=1e-6)
    assert abs(pos[0] - 0.95) < 1e-6
    # sphere on an ellipsoid along a principal axis
    k, d, pos, n = convex_pair(SPH, [0, 0, 0], I, [0.5, 0, 0], ELL, [0, 0, 0.95], I, [1, 0.7, 0.5])
    assert k == 1
    np.testing.assert_allclose(d, -0.05, atol=1e-7); np.testing.assert_allclose(n, [0, 0, 1], atol=1e-6)
    np.testing.assert_allclose(pos, [0, 0, 0.475], atol=1e-6)


def test_convex_cylinder_and_capsule_on_box():
    I = np.eye(3); box = ([0, 0, 0], I, [1, 1, 1])
    r, h = 0.3, 0.1
    # standing cylinder: the cap is 0.01 inside the top face; normal from the cylinder (geom 1) into the box
    k, d, pos, n = convex_pair(CYL, [0.1, 0.05, 1.09], I, [r, h, 0], BOX, *box)
    assert k == 1
    np.testing.assert_allclose(d, -0.01, atol=1e-6); np.testing.assert_allclose(n, [0, 0, -1], atol=1e-5)
    assert abs(pos[2] - 0.995) < 5e-3 and np.hypot(pos[0] - 0.1, pos[1] - 0.05) <= r + 1e-6
    # lying cylinder: line of touch, depth r - height
    k, d, pos, n = convex_pair(CYL, [0.1, 0.05, 1.29], rot([1, 0, 0], np.pi / 2), [r, h, 0], BOX, *box)
    assert k == 1
    np.testing.assert_allclose(d, -0.01, atol=2e-6); np.testing.assert_allclose(n, [0, 0, -1], atol=1e-5)
    # tilted cylinder: the lowest rim point
    th = 0.7
    k, d, pos, n = convex_pair(CYL, [0.1, 0.05, 1.2], rot([1, 0.3, 0], th), [r, h, 0], BOX, *box)
    lowest = 1.2 - (r * np.sin(th) + h * np.cos(th))
    assert k == 1
    np.testing.assert_allclose(d, lowest - 1.0, atol=2e-6); np.testing.assert_allclose(n, [0, 0, -1], atol=1e-5)
    # separated -> nothing; separated by less than the margin -> positive distance
    assert convex_pair(CYL, [0.1, 0.05, 1.5], rot([1, 0.3, 0], th), [r, h, 0], BOX, *box)[0] == 0
    k, d, pos, n = convex_pair(CYL, [0.1, 0.05, 1.105], I, [r, h, 0], BOX, *box, margin=0.01)
    assert k == 1
    np.testing.assert_allclose(d, 0.005, atol=2e-6)
    # capsule tilted over the face: depth = radius - height of the lower end point
    ang = 0.5
    k, d, pos, n = convex_pair(CAP, [0, 0, 1.25], rot([0, 1, 0], ang), [0.1, 0.3, 0], BOX, *box)
    assert k == 1
    np.testing.assert_allclose(d, (1.25 - 0.3 * np.cos(ang) - 0.1) - 1.0, atol=2e-6); np.testing.assert_allclose(n, [0, 0, -1], atol=1e-5)


def test_convex_is_invariant_under_rigid_motion_and_separates_the_pair():
    """depth and the frame-relative contact are the same in any world frame; translating geom 2 by depth along the
    normal brings the pair to touching (depth ~ 0)"""
    rng = np.random.default_rng(5)
    hits = soft = 0
    for _ in range(900):
        t1, t2 = [(CYL, CYL), (CYL, BOX), (CAP, BOX), (SPH, CYL), (ELL, BOX), (CAP, CYL)][rng.integers(6)]
        s1 = rng.uniform(0.1, 0.4, 3); s2 = rng.uniform(0.1, 0.4, 3)
        R1 = rot(rng.normal(size=3), rng.uniform(0, 3)); R2 = rot(rng.normal(size=3), rng.uniform(0, 3))
        p1 = np.zeros(3); p2 = rng.normal(size=3); p2 *= rng.uniform(0.2, 0.7) / np.linalg.norm(p2)
        k, d, pos, n = convex_pair(t1, p1, R1, s1, t2, p2, R2, s2)
        if not k or d < -0.04:     # deep overlaps (a centre inside the other geom) are ill-conditioned for portal refinement
            continue
        hits += 1
        assert d <= 1e-9 and abs(np.linalg.norm(n) - 1) < 1e-9
        Q = rot(rng.normal(size=3), rng.uniform(0, 3)); t = rng.normal(size=3)
        k2, d2, pos2, n2 = convex_pair(t1, Q @ p1 + t, Q @ R1, s1, t2, Q @ p2 + t, Q @ R2, s2)
        assert k2 == 1
        np.testing.assert_allclose(d2, d, atol=2e-4)
        soft += np.abs(n2 - Q @ n).max() > 0.02   # the ray leaves the Minkowski difference next to an edge: either face normal
        # moved apart by the depth (+ a little) the pair no longer overlaps by more than the tolerance
        k3, d3, _, _ = convex_pair(t1, p1, R1, s1, t2, p2 + n * (-d + 1e-4), R2, s2)
        assert k3 == 0 or d3 > -2e-4
    assert hits >= 40 and soft <= 0.15 * hits, (hits, soft)

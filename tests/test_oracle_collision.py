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


def convex_pair(t1, p1, R1, s1, t2, p2, R2, s2, margin=0.0, v1=None, v2=None):
    """v1 / v2: vertex clouds (n, 3) of mesh geoms (type 7), in the geom frame"""
    L = orc.lib()
    a = lambda x: np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))
    p1, R1, s1, p2, R2, s2 = map(a, (p1, R1, s1, p2, R2, s2))
    v1 = a(v1) if v1 is not None else np.zeros(3); v2 = a(v2) if v2 is not None else np.zeros(3)
    d = np.zeros(1); pos = np.zeros(3); n = np.zeros(3)
    P = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    k = L.orc_convex_pair(t1, P(p1), P(R1), P(s1), P(v1), len(v1) // 3 if t1 == 7 else 0,
                          t2, P(p2), P(R2), P(s2), P(v2), len(v2) // 3 if t2 == 7 else 0, margin, P(d), P(pos), P(n))
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


# ---- convex mesh assets
CUBE_V = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], dtype=np.float64)
CUBE_F = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]],
                  dtype=np.int32)


def mesh_body_model(lib, vert, face, scale=None, geom_pos=None, geom_quat=None, floor=True, body_pos=(0, 0, 1.0), density=1000.0):
    b = lib.mjh_builder_create(); set_opt(lib, b, timestep=0.002)
    if floor:
        lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, 0.05), None, None, None, -1, -1, -1, -1)
    v = np.ascontiguousarray(vert, dtype=np.float64); f = np.ascontiguousarray(face, dtype=np.int32)
    mid = lib.mjh_builder_add_mesh(b, v.ctypes.data_as(C.POINTER(C.c_double)), len(v), f.ctypes.data_as(C.POINTER(C.c_int)), len(f),
                                   D(*scale) if scale is not None else None)
    assert mid == 0, lib.mjh_last_error()
    bd = lib.mjh_builder_add_body(b, b"obj", 0, D(*body_pos), None, 0.0)
    lib.mjh_builder_add_joint(b, None, bd, 0, None, None, None, 0, 0, 0, 0, 0)
    g = lib.mjh_builder_add_mesh_geom(b, b"objgeom", bd, mid, D(*geom_pos) if geom_pos is not None else None,
                                      D(*geom_quat) if geom_quat is not None else None, None, -1, -1, -1, density)
    assert g >= 0
    m = ms.Model(lib.mjh_builder_compile(b), lib); lib.mjh_builder_destroy(b)
    return m


def test_cube_mesh_has_the_mass_properties_and_contacts_of_the_box(lib):
    h = np.array([0.1, 0.07, 0.04])
    m = mesh_body_model(lib, CUBE_V, CUBE_F, scale=h, body_pos=(0, 0, h[2] - 1e-3))
    assert m.c.nmesh == 1 and m.c.nmeshvert == 8 and m.array("geom_type")[-1] == 7
    mass = 1000 * 8 * h.prod()
    np.testing.assert_allclose(m.array("body_mass")[1], mass, rtol=1e-12)
    np.testing.assert_allclose(sorted(m.array("body_inertia")[3:6]), sorted(mass / 3 * np.array([h[1]**2 + h[2]**2, h[0]**2 + h[2]**2, h[0]**2 + h[1]**2])), rtol=1e-10)
    np.testing.assert_allclose(m.array("geom_rbound")[-1], np.linalg.norm(h), rtol=1e-12)
    d = orc.OrcData(m.ptr); d.call("forward")
    cons = d.contacts()
    assert len(cons) == 4                                      # the four bottom corners, like plane-box
    for c in cons:
        np.testing.assert_allclose(c["dist"], -1e-3, atol=1e-12); np.testing.assert_allclose(c["frame"][:3], [0, 0, 1], atol=1e-12)
    xy = sorted((round(c["pos"][0], 6), round(c["pos"][1], 6)) for c in cons)
    assert xy == sorted((sx * h[0], sy * h[1]) for sx in (-1, 1) for sy in (-1, 1))
    d.step(800)                                                # and it rests
    assert np.abs(d.f("qvel")).max() < 1e-3 and abs(d.f("qpos")[2] - h[2]) < 2e-3


def test_mesh_frame_is_centre_of_mass_and_principal_axes(lib):
    """the same box given in a shifted, rotated file frame: the compiler moves the geom frame to the centre of mass and
    the principal axes (mj_loadXML does the same), so the body's inertial properties do not depend on the file frame"""
    h = np.array([0.1, 0.07, 0.04]); Q = rot([1, 2, 3], 0.9); shift = np.array([0.3, -0.2, 0.5])
    V = (CUBE_V * h) @ Q.T + shift
    m = mesh_body_model(lib, V, CUBE_F, floor=False)
    mass = 1000 * 8 * h.prod()
    np.testing.assert_allclose(m.array("body_mass")[1], mass, rtol=1e-10)
    np.testing.assert_allclose(m.array("body_ipos")[3:6], shift, atol=1e-12)          # COM of the body = COM of the mesh
    np.testing.assert_allclose(m.array("geom_pos")[-3:], shift, atol=1e-12)
    I = mass / 3 * np.array([h[1]**2 + h[2]**2, h[0]**2 + h[2]**2, h[0]**2 + h[1]**2])
    np.testing.assert_allclose(sorted(m.array("body_inertia")[3:6]), sorted(I), rtol=1e-9)
    # the vertices in the geom frame are the axis-aligned box corners again (up to axis order and sign)
    mv = m.array("mesh_vert").reshape(-1, 3)
    np.testing.assert_allclose(sorted(np.abs(mv).max(axis=0)), sorted(h), rtol=1e-9)
    # a tetrahedron: volume 1/6 of the corner cube, centre of mass at the vertex mean
    T = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.0]]); F = np.array([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]], dtype=np.int32)
    m = mesh_body_model(lib, T, F, floor=False, density=600.0)
    np.testing.assert_allclose(m.array("body_mass")[1], 100.0, rtol=1e-12)
    np.testing.assert_allclose(m.array("body_ipos")[3:6], [0.25, 0.25, 0.25], atol=1e-12)


def test_mesh_support_mapping_collides_like_the_primitive(lib):
    """a cube mesh against a sphere / a cylinder through the generic convex routine = the box primitive"""
    h = np.array([0.2, 0.15, 0.1]); V = CUBE_V * h
    rng = np.random.default_rng(2)
    n = soft = 0
    for _ in range(400):
        R = rot(rng.normal(size=3), rng.uniform(0, 3)); p = rng.normal(size=3); p *= rng.uniform(0.15, 0.4) / np.linalg.norm(p)
        t2, s2 = ((SPH, [0.12, 0, 0]), (CYL, [0.08, 0.1, 0]))[rng.integers(2)]
        R2 = rot(rng.normal(size=3), rng.uniform(0, 3))
        a = convex_pair(t2, p, R2, s2, 7, [0, 0, 0], R, [0, 0, 0], v2=V)           # (sphere|cylinder, mesh)
        b = convex_pair(t2, p, R2, s2, BOX, [0, 0, 0], R, h)
        assert a[0] == b[0]
        if a[0] and b[1] > -0.05:      # (ties between equal support values are broken differently: compare shallow overlaps)
            n += 1
            np.testing.assert_allclose(a[1], b[1], atol=1e-4)
            soft += np.abs(a[3] - b[3]).max() > 1e-3 or np.abs(a[2] - b[2]).max() > 1e-3
    assert n >= 30 and soft <= 0.1 * n, (n, soft)


def test_binary_stl_through_the_mjcf_loader(tmp_path, lib):
    """<asset><mesh file scale> + <compiler meshdir> + <geom type="mesh">: the route the reference's robot files take"""
    import struct
    (tmp_path / "stl").mkdir()
    tris = (CUBE_V * 0.5)[CUBE_F]                                   # unit cube, 12 triangles
    with open(tmp_path / "stl" / "cube.stl", "wb") as f:
        f.write(b"\0" * 80 + struct.pack("<I", len(tris)))
        for t in tris:
            f.write(struct.pack("<12fH", 0, 0, 0, *t.reshape(-1).astype(np.float32), 0))
    xml = """<mujoco><compiler meshdir="stl"/><asset><mesh name="cube" file="cube.stl" scale="0.2 0.1 0.1"/></asset>
      <worldbody><geom type="plane" size="0 0 0.05"/>
        <body name="b" pos="0 0 0.3"><freejoint/><geom type="mesh" mesh="cube" pos="0.05 0 0"/></body></worldbody></mujoco>"""
    (tmp_path / "m.xml").write_text(xml)
    m = ms.load_mjcf(path=str(tmp_path / "m.xml"))
    assert m.c.nmesh == 1 and m.c.nmeshvert == 8 and m.c.npair == 1
    np.testing.assert_allclose(m.array("body_mass")[1], 1000 * 0.2 * 0.1 * 0.1, rtol=1e-6)      # float32 vertices in the file
    np.testing.assert_allclose(m.array("body_ipos")[3:6], [0.05, 0, 0], atol=1e-7)
    d = orc.OrcData(m.ptr); d.step(600)
    assert abs(d.f("qpos")[2] - 0.05) < 2e-3 and d.i("ncon") == 4      # lies on a 0.2 x 0.1 face... or stands: height of the COM
    # the same text without a directory: the mesh cannot be found, the geom is reported and skipped
    m2 = ms.load_mjcf(xml.replace("<freejoint/>", '<freejoint/><geom type="sphere" size="0.01"/>'))
    assert m2.c.nmesh == 0 and m2.c.ngeom == 2 and b"not loaded" in lib.mjh_load_note()


def test_ascii_stl_and_obj_meshes(tmp_path, lib):
    """text mesh formats next to the binary STL of the reference's assets: ASCII STL and Wavefront OBJ (quads, a/b/c indices)"""
    tris = (CUBE_V * np.array([0.1, 0.05, 0.05]))[CUBE_F]
    with open(tmp_path / "a.stl", "w") as f:
        f.write("solid cube\n")
        for t in tris:
            f.write(" facet normal 0 0 0\n  outer loop\n" + "".join("   vertex %.17g %.17g %.17g\n" % tuple(p) for p in t) + "  endloop\n endfacet\n")
        f.write("endsolid cube\n")
    V = CUBE_V * np.array([0.1, 0.05, 0.05])
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    with open(tmp_path / "o.obj", "w") as f:
        f.write("# cube\n" + "".join("v %.17g %.17g %.17g\n" % tuple(p) for p in V) + "vn 0 0 1\n" + "".join("f " + " ".join(f"{i+1}//1" for i in q) + "\n" for q in quads))
    for fn in ("a.stl", "o.obj"):
        xml = f"""<mujoco><asset><mesh name="m" file="{fn}"/></asset><worldbody><geom type="plane" size="0 0 0.05"/>
          <body pos="0 0 0.2"><freejoint/><geom type="mesh" mesh="m"/></body></worldbody></mujoco>"""
        (tmp_path / "m.xml").write_text(xml)
        m = ms.load_mjcf(path=str(tmp_path / "m.xml"))
        assert m.c.nmesh == 1 and m.c.nmeshvert == 8, (fn, lib.mjh_load_note())
        np.testing.assert_allclose(m.array("body_mass")[1], 1000 * 0.2 * 0.1 * 0.1, rtol=1e-9)
        np.testing.assert_allclose(sorted(np.abs(m.array("mesh_vert").reshape(-1, 3)).max(axis=0)), [0.05, 0.05, 0.1], rtol=1e-9)


def test_portal_refinement_against_the_exact_minkowski_difference():
    """An independent look at the restated portal refinement (`orc_convex_pair`; MuJoCo sends such pairs through libccd's MPR, which is
    not in this image): for POLYTOPE pairs (boxes, convex meshes) the Minkowski difference B - A is a convex hull scipy can build
    exactly.  Asserted on ~230 overlapping of 6000 random pairs: a contact is reported iff the hull contains the origin; the depth is
    never below the true minimal translation distance (min over facets of the origin's distance); and it IS exact portal refinement:
    the depth equals the plane distance of the facet through which the ray from the interior point (centre difference) through the
    origin leaves the hull — the property that defines the algorithm — in at least 90 % of the pairs (the rest: the ray leaves
    through an edge or a vertex), with the normal that facet's; the median pair gets the minimal depth itself."""
    from scipy.spatial import ConvexHull
    MESH = 7
    rng = np.random.default_rng(3)
    signs = np.array([[a, b, c] for a in (-1, 1) for b in (-1, 1) for c in (-1, 1)], dtype=float)

    def shape(kind):
        if kind == BOX:
            s = rng.uniform(0.05, 0.3, 3); return kind, s, signs * s, None
        pts = rng.normal(size=(rng.integers(6, 24), 3)) * rng.uniform(0.05, 0.3, 3)
        v = pts[ConvexHull(pts).vertices]; v = v - v.mean(0)
        return kind, np.ones(3), v, v
    overlapping = exact = 0; ratios = []
    for _ in range(8000):
        k1, s1, loc1, v1 = shape([BOX, MESH][rng.integers(2)]); k2, s2, loc2, v2 = shape([BOX, MESH][rng.integers(2)])
        if k1 == BOX and k2 == BOX:
            continue
        R1 = rot(rng.normal(size=3), rng.uniform(0, 3)); R2 = rot(rng.normal(size=3), rng.uniform(0, 3))
        p1 = rng.uniform(-1, 1, 3); d = rng.normal(size=3); d /= np.linalg.norm(d)
        A = p1 + loc1 @ R1.T
        p2 = p1 + d * (np.abs(loc1 @ R1.T @ d).max() + np.abs(loc2 @ R2.T @ d).max()) * rng.uniform(0.9, 1.02)
        B = p2 + loc2 @ R2.T
        k, dist, pos, n = convex_pair(k1, p1, R1, s1, k2, p2, R2, s2, v1=v1, v2=v2)
        H = ConvexHull((B[:, None, :] - A[None, :, :]).reshape(-1, 3))
        inside = bool((H.equations[:, 3] <= 0).all())
        assert bool(k) == inside or abs(H.equations[:, 3].max()) < 1e-7, (k, H.equations[:, 3].max())
        if not k:
            continue
        overlapping += 1
        true_depth = (-H.equations[:, 3]).min()
        assert -dist >= true_depth - 1e-9 and abs(np.linalg.norm(n) - 1) < 1e-9
        v0 = p2 - p1
        num = -(H.equations[:, :3] @ v0 + H.equations[:, 3]); den = H.equations[:, :3] @ (-v0)
        f = int(np.argmin(np.where(den > 1e-14, num / np.where(den > 1e-14, den, 1), np.inf)))
        if abs(-dist + H.equations[f, 3]) < 1e-6 and np.abs(n + H.equations[f, :3]).max() < 1e-5:      # (normal: from geom 1 into geom 2 = minus B - A's outward facet normal)
            exact += 1
        ratios.append(-dist / max(true_depth, 1e-12))
    assert overlapping >= 150 and exact >= 0.9 * overlapping, (overlapping, exact)
    assert np.median(ratios) < 1 + 1e-6 and np.quantile(ratios, 0.9) < 1.2, (np.median(ratios), np.quantile(ratios, 0.9))

"""What pins the oracle and the model compiler while libmujoco 2.3.7 cannot be had in this image (parity
unpinned).  CPU only.

1. The Gauss-Seidel visiting order.  The device and the oracle share an "independent pair" order; MuJoCo's mj_solPGS visits
   the rows in plain constraint order.  Both converge to the same dual solution, but settled S24 piles run into the 100-sweep
   cap first, so the iterates differ: the deviation is MEASURED here and stated in BASELINE.md §3 as part of the tolerance.
2. csrc/model_builder.cpp checked WITHOUT going through the quantities it derives: invweight0 / meaninertia against a dense
   numpy M^-1 built from an independent forward-kinematics + Jacobian restatement; inertia-from-geom against the closed
   forms; the decimated convex hulls against the full STL vertex sets of the reference's PR2 meshes.
3. Every SURVEY.md App. B item tagged (L)/(M) as a named known-answer test of the definition this repo uses (so that the day
   the library is reachable, a failing line names the convention that differs).  DESIGN.md §6 lists them as "restated, not
   verified against the library".
"""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import mujoco_sim_amd as ms
import orc
from helpers import D, oracle_s24, set_opt
from mujoco_sim_amd.tables import load_model_tables

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = "/root/reference"


# ----------------------------------------------------------------------------- 1. visiting order
def _clone(m, tab, i, src):
    d = oracle_s24(m, tab, i)
    for k in ("qpos", "qvel", "qacc_warmstart", "qacc"):
        d.f(k)[:] = src.f(k)
    return d


def test_pgs_row_order_deviation_at_the_sweep_cap_is_measured():
    """settled S24 (400 steps), then the SAME state stepped 150 more steps with (a) the device's independent-pair order and
    (b) plain row order (mj_solPGS): max |d qacc| after one step and max |d qpos|, |d qvel| after 150.  Measured over 12
    envs (tools/order_study.py prints the table): 1-step qacc differs by up to 0.7 m/s^2 where a pile is still moving
    (|qacc| ~ 10), qpos after 150 steps by 1e-2 in the worst env, median 7e-5 — the same size as the fp32-vs-fp64 forks
    the GPU tolerances already allow.  Asserted here on 4 envs with margins."""
    L = orc.lib()
    m = ms.scene("s24")
    N = 4
    tab = m.s24_randomize(0, N)
    worst_acc = worst_pos = 0.0; capped = 0
    try:
        for i in range(N):
            L.orc_set_pgs_row_order(0)
            s = oracle_s24(m, tab, i); s.step(400)
            a, b = _clone(m, tab, i, s), _clone(m, tab, i, s)
            for k in range(150):
                L.orc_set_pgs_row_order(0); a.step(1)
                L.orc_set_pgs_row_order(1); b.step(1)
                if k == 0:
                    worst_acc = max(worst_acc, float(np.abs(a.f("qacc") - b.f("qacc")).max()))
                    capped += a.i("solver_iter") >= 100
            worst_pos = max(worst_pos, float(np.abs(a.f("qpos") - b.f("qpos")).max()))
            assert a.i("ncon") > 8 and np.isfinite(a.f("qpos")).all() and np.isfinite(b.f("qpos")).all()
    finally:
        L.orc_set_pgs_row_order(1)
    print(f"row-order deviation over {N} envs: 1-step |d qacc| {worst_acc:.3e}, 150-step |d qpos| {worst_pos:.3e}, envs at the cap {capped}")
    assert capped >= 2                       # the premise: the default cap, not convergence, ends most solves
    assert 0 < worst_acc < 2.0 and worst_pos < 5e-2


def test_both_orders_converge_to_the_same_solution_when_allowed_to():
    """with the cap lifted (2000 sweeps, tolerance 1e-14) the two orders give the same qacc: the order changes the path,
    not the solution"""
    L = orc.lib()
    m = ms.scene("s24")
    tab = m.s24_randomize(0, 3)
    it0, tol0 = m.c.opt.iterations, m.c.opt.tolerance
    try:
        for i in range(3):
            L.orc_set_pgs_row_order(0)
            s = oracle_s24(m, tab, i); s.step(300)
            a, b = _clone(m, tab, i, s), _clone(m, tab, i, s)
            m.c.opt.iterations, m.c.opt.tolerance = 20000, 1e-16
            L.orc_set_pgs_row_order(0); a.call("forward")
            L.orc_set_pgs_row_order(1); b.call("forward")
            m.c.opt.iterations, m.c.opt.tolerance = it0, tol0
            scale = max(1.0, float(np.abs(a.f("qacc")).max()))
            assert np.abs(a.f("qacc") - b.f("qacc")).max() < 2e-5 * scale, (i, a.i("solver_iter"), b.i("solver_iter"))
    finally:
        m.c.opt.iterations, m.c.opt.tolerance = it0, tol0
        L.orc_set_pgs_row_order(1)


def test_list_scheduled_row_order_gives_the_sequential_iterates_bit_for_bit():
    """what the device's default relies on: blocks without a common kinematic tree commute EXACTLY under Gauss-Seidel, so a schedule
    that only swaps such blocks (orc_set_pgs_row_order(2): the constraint order, list-scheduled as patch_pgs.h / step_kernel.h do)
    reproduces the plain row-order sweep (1, mj_solPGS) bit for bit — in fp64 here, asserted on the device itself in
    tests/test_gpu_teacher_forced.py.  S24 settled piles at the sweep cap (where any REAL reordering shows: test above) and a 12-box pile."""
    L = orc.lib()
    cases = []
    m = ms.scene("s24"); tab = m.s24_randomize(0, 3)
    for i in range(3):
        cases.append((m, tab, i, 300, 40))
    mp = ms.scene("boxpile", 12); tp = ms.boxes_randomize(mp, 0, 2, jitter=0.01)
    for i in range(2):
        cases.append((mp, tp, i, 150, 25))
    try:
        for (mm, tt, i, settle, n) in cases:
            L.orc_set_pgs_row_order(1)
            s = oracle_s24(mm, tt, i); s.step(settle)
            a, b = _clone(mm, tt, i, s), _clone(mm, tt, i, s)
            its = []
            for k in range(n):
                L.orc_set_pgs_row_order(1); a.step(1)
                L.orc_set_pgs_row_order(2); b.step(1)
                its.append(a.i("solver_iter"))
                assert a.i("solver_iter") == b.i("solver_iter") and a.i("nefc") == b.i("nefc")
                assert np.array_equal(a.f("qacc"), b.f("qacc")) and np.array_equal(a.f("qpos"), b.f("qpos")), (mm.nv, i, k)
            assert a.i("ncon") >= 8 and max(its) > 5
    finally:
        L.orc_set_pgs_row_order(1)


def test_patch_and_pair_orders_converge_to_the_same_solution_and_deviate_little_at_the_cap():
    """the two device orders (contact patches: small free-body models; independent pairs: everything else) on the same settled S24
    states: identical qacc with the cap lifted, and at the default cap a 1-step deviation far below the one against MuJoCo's row
    order (tools/order_study.py: <= 0.09 m/s^2 over 12 envs)"""
    L = orc.lib()
    m = ms.scene("s24")
    tab = m.s24_randomize(0, 3)
    it0, tol0 = m.c.opt.iterations, m.c.opt.tolerance
    worst = 0.0
    try:
        L.orc_set_pgs_row_order(0)             # the two LEGACY orders (mjh_set_pgs_row_order(0))
        for i in range(3):
            L.orc_set_pgs_patch_order(-1)
            s = oracle_s24(m, tab, i); s.step(300)
            a, b = _clone(m, tab, i, s), _clone(m, tab, i, s)
            L.orc_set_pgs_patch_order(1); a.call("forward")
            L.orc_set_pgs_patch_order(0); b.call("forward")
            worst = max(worst, float(np.abs(a.f("qacc") - b.f("qacc")).max()))
            m.c.opt.iterations, m.c.opt.tolerance = 20000, 1e-16
            L.orc_set_pgs_patch_order(1); a.call("forward")
            L.orc_set_pgs_patch_order(0); b.call("forward")
            m.c.opt.iterations, m.c.opt.tolerance = it0, tol0
            scale = max(1.0, float(np.abs(a.f("qacc")).max()))
            assert np.abs(a.f("qacc") - b.f("qacc")).max() < 2e-5 * scale, (i, a.i("solver_iter"), b.i("solver_iter"))
    finally:
        m.c.opt.iterations, m.c.opt.tolerance = it0, tol0
        L.orc_set_pgs_patch_order(-1); L.orc_set_pgs_row_order(1)
    assert worst < 0.5, worst


def test_pgs_at_the_default_cap_against_the_converged_dual_solution_is_measured():
    """The reference's models name no solver, so MuJoCo runs Newton on them: to solver precision, the OPTIMUM of the convex problem
    PGS iterates on.  This engine (and the oracle) solve it with PGS at MuJoCo's default cap of 100 sweeps / tolerance 1e-8, which
    does not converge on settled S24 piles (250-870 sweeps are needed at tolerance 1e-13).  Measured over 10 envs from identical
    settled states (recorded in HISTORY.md §6): 1-step |d qacc| up to 0.5 m/s^2, 150-step |d qpos| up to 1.2e-2,
    median 1.9e-4.  Asserted on 3 envs with margins; part of the stated tolerance."""
    m = ms.scene("s24")
    N = 3
    tab = m.s24_randomize(0, N)
    it0, tol0 = m.c.opt.iterations, m.c.opt.tolerance
    worst_acc = worst_pos = 0.0; sweeps = []
    try:
        for i in range(N):
            s = oracle_s24(m, tab, i); s.step(400)
            a, b = _clone(m, tab, i, s), _clone(m, tab, i, s)
            for k in range(150):
                m.c.opt.iterations, m.c.opt.tolerance = it0, tol0; a.step(1)
                m.c.opt.iterations, m.c.opt.tolerance = 4000, 1e-13; b.step(1)
                if k == 0:
                    worst_acc = max(worst_acc, float(np.abs(a.f("qacc") - b.f("qacc")).max())); sweeps.append(b.i("solver_iter"))
            worst_pos = max(worst_pos, float(np.abs(a.f("qpos") - b.f("qpos")).max()))
    finally:
        m.c.opt.iterations, m.c.opt.tolerance = it0, tol0
    print(f"PGS at the cap vs converged: 1-step |d qacc| {worst_acc:.3e}, 150-step |d qpos| {worst_pos:.3e}, sweeps to converge {sweeps}")
    assert max(sweeps) > 100                      # the premise: the default cap ends the iteration early
    assert worst_acc < 2.0 and worst_pos < 5e-2


# ----------------------------------------------------------------------------- 2. the model compiler, independently
def _quat2mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _mulquat(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def dense_mass_matrix_at_qpos0(m, mass=None, inertia=None):
    """Independent restatement in numpy: forward kinematics at qpos0, world-frame Jacobians of every body's centre of mass,
    M = sum_b m Jp^T Jp + Jr^T (R I R^T) Jr + diag(armature).  Uses only the model's INPUT tables (frames, joints, masses)."""
    A = m.array
    nb, nv = m.nbody, m.nv
    par, bpos, bquat, ipos, iquat = A("body_parentid"), A("body_pos").reshape(-1, 3), A("body_quat").reshape(-1, 4), A("body_ipos").reshape(-1, 3), A("body_iquat").reshape(-1, 4)
    mass = A("body_mass") if mass is None else mass
    inertia = (A("body_inertia") if inertia is None else inertia).reshape(-1, 3)
    jadr, jnum, jtype, jpos, jaxis, jq, jd = A("body_jntadr"), A("body_jntnum"), A("jnt_type"), A("jnt_pos").reshape(-1, 3), A("jnt_axis").reshape(-1, 3), A("jnt_qposadr"), A("jnt_dofadr")
    q0 = A("qpos0")
    xpos = np.zeros((nb, 3)); xquat = np.zeros((nb, 4)); xquat[0, 0] = 1
    cols = [[] for _ in range(nb)]          # per body: list of (dof, kind, axis, anchor) acting on it directly
    for b in range(1, nb):
        p = par[b]
        if jnum[b] == 1 and jtype[jadr[b]] == 0:
            j = jadr[b]
            xpos[b] = q0[jq[j]:jq[j] + 3]; xquat[b] = q0[jq[j] + 3:jq[j] + 7] / np.linalg.norm(q0[jq[j] + 3:jq[j] + 7])
            R = _quat2mat(xquat[b])
            for k in range(3):
                cols[b].append((jd[j] + k, "lin", np.eye(3)[k], None))
            for k in range(3):
                cols[b].append((jd[j] + 3 + k, "rot", R[:, k], xpos[b].copy()))
            continue
        xpos[b] = xpos[p] + _quat2mat(xquat[p]) @ bpos[b]; xquat[b] = _mulquat(xquat[p], bquat[b])
        for j in range(jadr[b], jadr[b] + jnum[b]):
            R = _quat2mat(xquat[b])
            anchor = xpos[b] + R @ jpos[j]; axis = R @ jaxis[j]
            if jtype[j] == 1:      # ball: rotate about the anchor by the reference quaternion, dofs about the body axes
                xquat[b] = _mulquat(xquat[b], q0[jq[j]:jq[j] + 4]); R = _quat2mat(xquat[b]); xpos[b] = anchor - R @ jpos[j]
                for k in range(3):
                    cols[b].append((jd[j] + k, "rot", R[:, k], anchor))
            elif jtype[j] == 2:
                cols[b].append((jd[j], "lin", axis, None))
            else:
                cols[b].append((jd[j], "rot", axis, anchor))
    M = np.diag(A("dof_armature").astype(float)) if nv else np.zeros((0, 0))
    Jp_all, Jr_all = np.zeros((nb, 3, nv)), np.zeros((nb, 3, nv))
    for b in range(1, nb):
        R = _quat2mat(xquat[b]); c = xpos[b] + R @ ipos[b]
        a = b
        while a > 0:
            for (d, kind, axis, anchor) in cols[a]:
                if kind == "lin":
                    Jp_all[b, :, d] = axis
                else:
                    Jr_all[b, :, d] = axis; Jp_all[b, :, d] = np.cross(axis, c - anchor)
            a = par[a]
        Ri = _quat2mat(_mulquat(xquat[b], iquat[b]))
        Iw = Ri @ np.diag(inertia[b]) @ Ri.T
        M += mass[b] * Jp_all[b].T @ Jp_all[b] + Jr_all[b].T @ Iw @ Jr_all[b]
    return M, Jp_all, Jr_all


def _expected_invweights(m, M, Jp, Jr):
    Minv = np.linalg.inv(M)
    dinv = np.diag(Minv).copy()
    jt, jd = m.array("jnt_type"), m.array("jnt_dofadr")
    dof = np.zeros(m.nv)
    for j in range(m.njnt):
        a = jd[j]
        if jt[j] == 0:
            dof[a:a + 3] = dinv[a:a + 3].mean(); dof[a + 3:a + 6] = dinv[a + 3:a + 6].mean()
        elif jt[j] == 1:
            dof[a:a + 3] = dinv[a:a + 3].mean()
        else:
            dof[a] = dinv[a]
    body = np.zeros((m.nbody, 2))
    weld = m.array("body_weldid")
    for b in range(1, m.nbody):
        if weld[b] == 0:
            continue
        body[b, 0] = np.trace(Jp[b] @ Minv @ Jp[b].T) / 3; body[b, 1] = np.trace(Jr[b] @ Minv @ Jr[b].T) / 3
    return dof, body.reshape(-1)


@pytest.mark.parametrize("name", ["s24", "pendulum", "arm7", "pr2", "tiago", "hsrb4s", "ridgeback_panda"])
def test_invweight0_and_meaninertia_against_a_dense_numpy_inverse(name):
    """mj_setConst quantities derived by model_builder.cpp (they scale every regulariser R and the solver tolerance) against
    M^-1 from an independent numpy restatement of FK + Jacobians at qpos0"""
    if name in ("s24", "pendulum"):
        m = ms.scene(name)
    elif name == "arm7":
        m = ms.scene("arm7", 1)
    else:
        m, _ = load_model_tables(os.path.join(G, f"robot_{name}.npz"))
    M, Jp, Jr = dense_mass_matrix_at_qpos0(m)
    assert np.allclose(M, M.T) and np.linalg.eigvalsh(M).min() > 0
    np.testing.assert_allclose(m.meaninertia, np.trace(M) / m.nv, rtol=1e-9)
    dof, body = _expected_invweights(m, M, Jp, Jr)
    np.testing.assert_allclose(m.array("dof_invweight0"), dof, rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(m.array("body_invweight0"), body, rtol=1e-7, atol=1e-12)
    # and the oracle's CRBA mass matrix at qpos0 is that same M (sparse qM -> dense)
    d = orc.OrcData(m.ptr); d.call("reset"); d.call("fwd_position")
    qM, madr, dpar = d.f("qM"), m.array("dof_Madr"), m.array("dof_parentid")
    Mo = np.zeros_like(M)
    for i in range(m.nv):
        a = madr[i]; j = i
        while j >= 0:
            Mo[i, j] = Mo[j, i] = qM[a]; a += 1; j = dpar[j]
    np.testing.assert_allclose(Mo, M, rtol=1e-9, atol=1e-10)


def test_s24_randomized_tables_are_consistent_with_the_box_sizes():
    """the per-env S24 tables (mass, inertia, invweight0) follow from the drawn half-extents at density 1000 exactly as the
    compiler derives them for the shared model"""
    m = ms.scene("s24")
    tab = m.s24_randomize(0, 5)
    for i in range(5):
        size = tab["geom_size"][i].reshape(-1, 3)
        for k in range(4):
            g = m.array("body_geomadr")[1 + k]
            h = size[g]; mass = 1000 * 8 * h.prod()
            np.testing.assert_allclose(tab["body_mass"][i][1 + k], mass, rtol=1e-12)
            np.testing.assert_allclose(tab["body_inertia"][i].reshape(-1, 3)[1 + k], mass / 3 * np.array([h[1]**2 + h[2]**2, h[0]**2 + h[2]**2, h[0]**2 + h[1]**2]), rtol=1e-12)
        M, Jp, Jr = dense_mass_matrix_at_qpos0(m, mass=tab["body_mass"][i], inertia=tab["body_inertia"][i])
        dof, body = _expected_invweights(m, M, Jp, Jr)
        np.testing.assert_allclose(tab["dof_invweight0"][i], dof, rtol=1e-9)
        np.testing.assert_allclose(tab["body_invweight0"][i], body, rtol=1e-9)


def _one_geom_body(lib, gtype, size, density=1000.0):
    b = lib.mjh_builder_create()
    bd = lib.mjh_builder_add_body(b, b"b", 0, D(0, 0, 1), None, 0.0)
    lib.mjh_builder_add_joint(b, b"j", bd, 0, None, None, None, 0, 0, 0, 0, 0)
    lib.mjh_builder_add_geom(b, b"g", bd, gtype, D(*size), None, None, None, -1, -1, -1, density)
    m = ms.Model(lib.mjh_builder_compile(b), lib); lib.mjh_builder_destroy(b)
    return m.array("body_mass")[1], np.sort(m.array("body_inertia")[3:6])


def test_inertia_from_geoms_matches_the_closed_forms(lib):
    """<geom> without <inertial>: mass and principal inertias at density rho for every primitive (textbook solids)"""
    rho = 1000.0
    r, h = 0.07, 0.11
    # sphere
    mass, I = _one_geom_body(lib, 2, (r, 0, 0)); ms_ = rho * 4 / 3 * np.pi * r**3
    np.testing.assert_allclose(mass, ms_, rtol=1e-12); np.testing.assert_allclose(I, [0.4 * ms_ * r * r] * 3, rtol=1e-12)
    # box (half-extents)
    a, b, c = 0.05, 0.08, 0.12
    mass, I = _one_geom_body(lib, 6, (a, b, c)); mb = rho * 8 * a * b * c
    np.testing.assert_allclose(mass, mb, rtol=1e-12)
    np.testing.assert_allclose(I, np.sort([mb / 3 * (b*b + c*c), mb / 3 * (a*a + c*c), mb / 3 * (a*a + b*b)]), rtol=1e-12)
    # cylinder (radius, half-height), axis z
    mass, I = _one_geom_body(lib, 5, (r, h, 0)); mc = rho * np.pi * r * r * 2 * h
    np.testing.assert_allclose(mass, mc, rtol=1e-12)
    np.testing.assert_allclose(I, np.sort([mc * (3 * r * r + (2 * h)**2) / 12] * 2 + [0.5 * mc * r * r]), rtol=1e-12)
    # ellipsoid (semi-axes)
    mass, I = _one_geom_body(lib, 4, (a, b, c)); me = rho * 4 / 3 * np.pi * a * b * c
    np.testing.assert_allclose(mass, me, rtol=1e-12)
    np.testing.assert_allclose(I, np.sort([me / 5 * (b*b + c*c), me / 5 * (a*a + c*c), me / 5 * (a*a + b*b)]), rtol=1e-12)
    # capsule (radius, half-length of the cylinder part): cylinder + two hemispheres (parallel-axis for the caps)
    mass, I = _one_geom_body(lib, 3, (r, h, 0))
    mcyl = rho * np.pi * r * r * 2 * h; msph = rho * 4 / 3 * np.pi * r**3
    Izz = 0.5 * mcyl * r * r + 0.4 * msph * r * r
    Ixx = mcyl * (3 * r * r + (2 * h)**2) / 12 + msph * (0.4 * r * r + h * h + 0.75 * r * h)
    np.testing.assert_allclose(mass, mcyl + msph, rtol=1e-12)
    np.testing.assert_allclose(I, np.sort([Ixx, Ixx, Izz]), rtol=1e-10)
    # density scales everything linearly
    m2, I2 = _one_geom_body(lib, 6, (a, b, c), density=250.0)
    np.testing.assert_allclose([m2 / mb], [0.25], rtol=1e-12)


def _read_binary_stl(path):
    with open(path, "rb") as f:
        f.read(80); (n,) = struct.unpack("<I", f.read(4))
        raw = np.frombuffer(f.read(50 * n), dtype=np.uint8).reshape(n, 50)
    return raw[:, 12:48].copy().view("<f4").reshape(n * 3, 3).astype(np.float64)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "model/test/pr2")), reason="the reference's mesh assets are only present in the build container")
def test_decimated_hulls_support_the_full_pr2_meshes(lib):
    """The builder keeps only the vertices of a mesh that are extreme along 1526 directions.  For every STL asset of the
    reference's PR2 (pr2.xml:5-22) the support function of the kept set equals that of the FULL vertex set along those
    directions and stays within 2e-3 of the mesh size along 4000 random ones (measured worst: 1.2e-3, forearm.stl: 1020 -> 66
    vertices; an inner approximation, never outside)."""
    import glob
    import xml.etree.ElementTree as ET
    root = ET.parse(os.path.join(REF, "model/test/pr2/pr2.xml")).getroot()
    meshdir = root.find("compiler").get("meshdir") or ""
    files = sorted({(e.get("file"), e.get("scale") or "1 1 1") for e in root.find("asset").findall("mesh")})
    assert len(files) >= 10
    rng = np.random.default_rng(0)
    dirs = rng.normal(size=(4000, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    worst = 0.0
    for fn, scale in files:
        path = os.path.normpath(os.path.join(REF, "model/test/pr2", meshdir, fn))
        if not os.path.exists(path):
            cand = glob.glob(os.path.join(REF, "model", "**", os.path.basename(fn)), recursive=True)
            assert cand, fn
            path = cand[0]
        sc = np.array([float(x) for x in scale.split()])
        V = _read_binary_stl(path) * sc
        b = lib.mjh_builder_create()
        mid = lib.mjh_builder_add_mesh_stl(b, path.encode(), D(*sc))
        assert mid >= 0, lib.mjh_last_error()
        bd = lib.mjh_builder_add_body(b, b"b", 0, D(0, 0, 0), None, 0.0)
        lib.mjh_builder_add_joint(b, b"j", bd, 0, None, None, None, 0, 0, 0, 0, 0)
        lib.mjh_builder_add_mesh_geom(b, b"g", bd, mid, None, None, None, -1, -1, -1, -1)
        m = ms.Model(lib.mjh_builder_compile(b), lib); lib.mjh_builder_destroy(b)
        # hull vertices back in the FILE frame: geom frame (COM, principal axes) -> body frame = file frame (pos 0, quat 1)
        H = m.array("mesh_vert").reshape(-1, 3)
        g = m.ngeom - 1
        R = _quat2mat(m.array("geom_quat").reshape(-1, 4)[g]); p = m.array("geom_pos").reshape(-1, 3)[g]
        Hf = H @ R.T + p
        size = np.linalg.norm(V.max(0) - V.min(0))
        full, kept = (V @ dirs.T).max(0), (Hf @ dirs.T).max(0)
        assert (kept <= full + 1e-9 * size).all()                       # kept vertices are mesh vertices: never outside
        err = float((full - kept).max() / size)
        worst = max(worst, err)
        assert err < 2e-3, (fn, err, len(V), len(H))
    print(f"worst hull support error over {len(files)} PR2 meshes: {worst:.2e} of the mesh size")


# ----------------------------------------------------------------------------- 3. App. B (L)/(M) conventions as named KATs
def _box_on_floor(lib, mu_floor=(2, 0.05, 0.01), condim_floor=4, impratio=1.0, **opt):
    b = lib.mjh_builder_create()
    set_opt(lib, b, timestep=0.005, impratio=impratio, **opt)
    lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, 0.05), None, None, D(*mu_floor), condim_floor, -1, -1, -1)
    bd = lib.mjh_builder_add_body(b, b"box", 0, D(0, 0, 0.0995), None, 0.0)
    lib.mjh_builder_add_joint(b, b"free", bd, 0, None, None, None, 0, 0, 0, 0, 0)
    lib.mjh_builder_add_geom(b, b"g", bd, 6, D(0.1, 0.1, 0.1), None, None, None, -1, -1, -1, -1)
    m = ms.Model(lib.mjh_builder_compile(b), lib); lib.mjh_builder_destroy(b)
    d = orc.OrcData(m.ptr); d.call("reset"); d.call("fwd_position"); d.call("fwd_velocity")
    return m, d


def test_B5_contact_parameter_mixing_max_friction_max_condim(lib):
    """App. B.5 (M): condim = max, friction = element-wise max (equal priority), solref/solimp mixed by solmix, margin = max"""
    m, d = _box_on_floor(lib)
    c = d.contacts()
    assert len(c) == 4 and all(x["dim"] == 4 for x in c)
    assert d.i("nefc") == 4 * 6
    # friction max(2, 1) = 2 shows in the tangential rows: J_row = J_n +- mu J_t with mu = 2
    J = d.f("efc_J").reshape(d.i("nefc"), m.nv)
    jn = 0.5 * (J[0] + J[1]); jt = 0.5 * (J[0] - J[1])
    assert abs(np.linalg.norm(jt[:3]) / np.linalg.norm(jn[:3]) - 2.0) < 1e-9


def test_B5_contact_frame_tangents_are_the_deterministic_makeframe(lib):
    """App. B.5 (L): frame row 0 = normal; tangent 1 = the world axis least aligned with the normal, orthogonalised and
    normalised (y for a z normal, as mju_makeFrame picks it); tangent 2 = normal x tangent 1"""
    m, d = _box_on_floor(lib)
    fr = d.contacts()[0]["frame"].reshape(3, 3)
    np.testing.assert_allclose(fr[0], [0, 0, 1], atol=1e-12)
    np.testing.assert_allclose(np.cross(fr[0], fr[1]), fr[2], atol=1e-12)
    assert abs(abs(fr[1] @ np.array([0, 1, 0])) - 1) < 1e-12            # not x: the least aligned axis is searched from y


def test_B6_pyramid_rows_are_normal_plus_minus_mu_tangent(lib):
    """App. B.6 (M): condim 3 -> 4 rows J_n +- mu J_t1, J_n +- mu J_t2; condim 4 adds J_n +- mu_torsion J_r; all rows share pos = dist"""
    m, d = _box_on_floor(lib, mu_floor=(0.7, 0.02, 0.01), condim_floor=4)
    J = d.f("efc_J").reshape(d.i("nefc"), m.nv); pos = d.f("efc_pos")
    c0 = d.contacts()[0]; fr = c0["frame"].reshape(3, 3)
    assert np.allclose(pos[:6], c0["dist"])
    jn = 0.5 * (J[0] + J[1])
    np.testing.assert_allclose(jn[:3], fr[0], atol=1e-12)                 # geom2 (the box) minus geom1 (the world)
    np.testing.assert_allclose(0.5 * (J[0] - J[1])[:3], 1.0 * fr[1], atol=1e-12)    # mu = max(0.7, 1) = 1
    np.testing.assert_allclose(0.5 * (J[2] - J[3])[:3], 1.0 * fr[2], atol=1e-12)
    tors = 0.5 * (J[4] - J[5])
    assert np.allclose(tors[:3], 0) and abs(np.linalg.norm(tors[3:6]) - 0.02) < 1e-12   # torsional: max(0.02, 0.005), rotation about the normal


def test_B7_impedance_sigmoid_and_regulariser(lib):
    """App. B.7 (M/L): d(x) between solimp d0 and dwidth over `width` (power 2, midpoint 0.5), R = (1-d)/d * diagApprox with
    diagApprox = invweight0 of the two bodies (translational) x (1 + mu^2) for the first pyramid row, and every row of a
    pyramidal contact carries R_py = 2 mu^2 R (mu scaled by sqrt(1/impratio))"""
    for impratio in (1.0, 4.0):
        m, d = _box_on_floor(lib, impratio=impratio)
        c0 = d.contacts()[0]
        x = abs(c0["dist"]) / 0.001                                       # default solimp 0.9 0.95 0.001 0.5 2
        y = 2 * x * x if x <= 0.5 else 1 - 2 * (1 - x) ** 2
        imp = 0.9 + min(max(y, 0), 1) * 0.05 if x < 1 else 0.95
        tran = m.array("body_invweight0")[2]                              # box translational (world: 0)
        mu = 2.0
        diag = tran + mu * mu * tran
        R0 = (1 - imp) / imp * diag
        Rpy = 2 * (mu / np.sqrt(impratio)) ** 2 * R0
        np.testing.assert_allclose(d.f("efc_R")[:6], Rpy, rtol=1e-9)
        np.testing.assert_allclose(d.f("efc_diagApprox")[0], diag, rtol=1e-9)


def test_B7_reference_acceleration_gains_from_solref(lib):
    """App. B.7 (M): solref (0.02, 1), refsafe => timeconst >= 2 h; B = 2 / (dmax tc), K = 1 / (dmax^2 tc^2 dr^2);
    aref = -B (J qvel) - K imp (pos - margin)"""
    m, d = _box_on_floor(lib)
    d.f("qvel")[2] = -0.3
    d.call("fwd_velocity")
    tc, dr, dmax = 0.02, 1.0, 0.95
    Bc = 2 / (dmax * tc); K = 1 / (dmax * dmax * tc * tc * dr * dr)
    kbip = d.f("efc_KBIP").reshape(-1, 4)
    np.testing.assert_allclose(kbip[0, :2], [K, Bc], rtol=1e-12)
    imp = kbip[0, 2]
    J = d.f("efc_J").reshape(d.i("nefc"), m.nv)
    np.testing.assert_allclose(d.f("efc_aref")[0], -Bc * (J[0] @ d.f("qvel")) - K * imp * d.f("efc_pos")[0], rtol=1e-10)
    # refsafe: a time constant below 2 h is raised to 2 h (h = 0.02 -> tc = 0.04)
    m2, d2 = _box_on_floor(lib)
    m2.c.opt.timestep = 0.02
    d2.call("fwd_position"); d2.call("fwd_velocity")
    k2 = d2.f("efc_KBIP").reshape(-1, 4)[0]
    np.testing.assert_allclose(k2[:2], [1 / (dmax * dmax * 0.04 * 0.04), 2 / (dmax * 0.04)], rtol=1e-12)


def test_B8_gravcomp_cancels_gravity_in_qfrc_passive_while_bias_keeps_it(lib):
    """App. B.8 parity note: gravcomp = 1 puts +m g J^T into qfrc_passive, qfrc_bias still contains gravity, and the
    wrapper adds qfrc_bias on controlled dofs on top (mj_sim.cpp:1058-1063, 301-310) — reproduced literally"""
    m = ms.scene("arm7", 1)
    d = orc.OrcData(m.ptr); d.call("reset")
    d.f("qpos")[:] = [0.3, -0.5, 0.2, -1.2, 0.1, 0.9, 0.0]
    d.call("fwd_position"); d.call("fwd_velocity")
    np.testing.assert_allclose(d.f("qfrc_passive"), d.f("qfrc_bias"), rtol=1e-9, atol=1e-9)   # zero velocity: bias = gravity term only
    assert np.abs(d.f("qfrc_bias")).max() > 1.0


def test_B9_warm_start_is_kept_only_if_its_dual_cost_is_negative(lib):
    """App. B.9 (M): start from the forces implied by qacc_warmstart iff their dual cost < 0, else from zero forces: a wildly
    wrong warm start is DISCARDED (the solve is then bit-identical to one with the warm start disabled), a good one is kept
    (and the result agrees with the cold solve to solver accuracy)"""
    m, d = _box_on_floor(lib)
    d.step(200)
    good = d.f("qacc_warmstart").copy()
    a = d.f("qpos").copy(), d.f("qvel").copy()

    def solve(ws, cold=False):
        e = orc.OrcData(m.ptr); e.call("reset")
        e.f("qpos")[:] = a[0]; e.f("qvel")[:] = a[1]; e.f("qacc_warmstart")[:] = ws
        flags = m.c.opt.disableflags
        if cold:
            m.c.opt.disableflags = flags | (1 << 8)          # MJH_DSBL_WARMSTART
        try:
            e.call("forward")
        finally:
            m.c.opt.disableflags = flags
        return e.f("qacc").copy(), e.i("solver_iter")
    cold, warm, bad = solve(good, cold=True), solve(good), solve(good + 500.0)
    assert np.array_equal(bad[0], cold[0]) and bad[1] == cold[1]
    assert not np.array_equal(warm[0], cold[0])
    # (PGS at tolerance 1e-8 / 100 sweeps: warm and cold agree to solver accuracy, not to round-off)
    assert np.abs(warm[0] - cold[0]).max() < 2e-3


def test_B9_euler_with_damping_solves_M_plus_h_D(lib):
    """App. B.9 (M): (M + h diag(damping)) qacc' = qfrc_smooth + qfrc_constraint; qvel += h qacc'; qpos integrated with the NEW qvel"""
    from helpers import hinge_pendulum_model
    m = hinge_pendulum_model(lib, damping=0.5, mass=2.0, length=1.0, inertia=0.1)
    d = orc.OrcData(m.ptr); d.call("reset"); d.f("qpos")[0] = 0.4; d.f("qvel")[0] = 0.7
    h, I = m.c.opt.timestep, 0.1 + 2.0
    tau = -2.0 * 9.81 * 1.0 * np.sin(0.4) - 0.5 * 0.7
    v1 = 0.7 + h * tau / (I + h * 0.5)
    d.step(1)
    np.testing.assert_allclose(d.f("qvel")[0], v1, rtol=1e-12); np.testing.assert_allclose(d.f("qpos")[0], 0.4 + h * v1, rtol=1e-12)

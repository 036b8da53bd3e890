"""indep_model.py — TEST INFRASTRUCTURE: a second, independent model construction (numpy, fp64) for the scenes the headline numbers and the
plumbing config rest on — S24 / S24D (4 free boxes in a pen), C1 (model/test/pendulum.xml), C3 (the 7-hinge Panda chain) —, so that the
oracle does not have to be built on the PRODUCT's compiled model (VERDICT r05 weak #1b, next #6: "a model_builder.cpp defect is invisible to
every GPU parity test").  From a scene DESCRIPTION (bodies, joints, geoms at density 1000) it derives what mj_loadXML / mj_setConst derive:
mass, centre of mass and inertia tensor from the geoms (closed forms, composition by the parallel-axis theorem), principal axes, qpos0,
the joint / dof tables, dof_Madr, geom bounding radii, and — through tests/indep_dyn.py's Jacobian-sum mass matrix — dof_invweight0,
body_invweight0 and meaninertia.  No code shared with csrc/model_builder.cpp, csrc/scenes.cpp or oracle/.

tests/test_independent_model.py (a) asserts table equality with the product's compiler and (b) runs the oracle on a model whose physics
tables are THESE, against the oracle on the product's model."""
import numpy as np

from indep_dyn import FREE, BALL, SLIDE, HINGE, Tree, mulquat, quat2mat

PLANE, SPHERE, CAPSULE, ELLIPSOID, CYLINDER, BOX = 0, 2, 3, 4, 5, 6        # mjtGeom values (mujoco.h; include/mjhip.h MJH_GEOM_*)


def geom_mass_inertia(gtype, size, density=1000.0):
    """mass and diagonal inertia (own frame) of a primitive at uniform density — textbook closed forms"""
    a, b, c = (list(size) + [0, 0, 0])[:3]
    if gtype == SPHERE:
        m = density * 4.0 / 3.0 * np.pi * a ** 3
        return m, np.full(3, 0.4 * m * a * a)
    if gtype == BOX:
        m = density * 8 * a * b * c
        return m, m / 3.0 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    if gtype == CYLINDER:                      # radius a, half height b, axis z
        m = density * np.pi * a * a * 2 * b
        return m, np.array([m * (3 * a * a + 4 * b * b) / 12.0, m * (3 * a * a + 4 * b * b) / 12.0, 0.5 * m * a * a])
    if gtype == ELLIPSOID:
        m = density * 4.0 / 3.0 * np.pi * a * b * c
        return m, m / 5.0 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    if gtype == CAPSULE:                       # radius a, half height b: cylinder + two half spheres, each with its own centre of mass
        mc = density * np.pi * a * a * 2 * b; ms_ = density * 4.0 / 3.0 * np.pi * a ** 3
        izz = 0.5 * mc * a * a + 0.4 * ms_ * a * a
        # half sphere: I about its own com = (83 / 320) m_h a^2 (transverse); com at 3a/8 from the flat face
        mh = 0.5 * ms_; d = b + 3.0 * a / 8.0
        ixx = mc * (3 * a * a + 4 * b * b) / 12.0 + 2 * (83.0 / 320.0 * mh * a * a + mh * d * d)
        return mc + ms_, np.array([ixx, ixx, izz])
    return 0.0, np.zeros(3)


def geom_rbound(gtype, size):
    a, b, c = (list(size) + [0, 0, 0])[:3]
    return {SPHERE: a, BOX: float(np.sqrt(a * a + b * b + c * c)), CYLINDER: float(np.sqrt(a * a + b * b)), CAPSULE: a + b, ELLIPSOID: max(a, b, c)}.get(gtype, 0.0)


def _unit(q):
    q = np.asarray(q, float)
    return q / np.linalg.norm(q)


class Scene:
    def __init__(self, gravity=(0, 0, -9.81), timestep=0.005):
        self.gravity, self.timestep = np.array(gravity, float), timestep
        self.bodies = [dict(name="world", parent=-1, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]), joints=[], geoms=[], gravcomp=0.0)]

    def body(self, name, parent=0, pos=(0, 0, 0), quat=(1, 0, 0, 0), gravcomp=0.0):
        self.bodies.append(dict(name=name, parent=parent, pos=np.array(pos, float), quat=_unit(quat), joints=[], geoms=[], gravcomp=float(gravcomp)))
        return len(self.bodies) - 1

    def joint(self, body, jtype, pos=(0, 0, 0), axis=(0, 0, 1), range_=None, damping=0.0, armature=0.0):
        self.bodies[body]["joints"].append(dict(type=jtype, pos=np.array(pos, float), axis=_unit(axis), range=range_, damping=damping, armature=armature))

    def geom(self, body, gtype, size, pos=(0, 0, 0), quat=(1, 0, 0, 0), density=1000.0):
        self.bodies[body]["geoms"].append(dict(type=gtype, size=np.array((list(size) + [0, 0, 0])[:3], float), pos=np.array(pos, float), quat=_unit(quat), density=density))


class Tables(dict):
    """the arrays by name, with the little API tests/indep_dyn.Tree reads"""
    def array(self, name): return np.asarray(self[name])
    @property
    def nbody(self): return len(self["body_mass"])
    @property
    def nv(self): return len(self["dof_bodyid"])
    @property
    def nq(self): return len(self["qpos0"])
    @property
    def njnt(self): return len(self["jnt_type"])


def compile_scene(sc, boundmass=0.0, boundinertia=0.0):
    nb = len(sc.bodies)
    T = Tables()
    T["body_parentid"] = np.array([max(b["parent"], 0) for b in sc.bodies], np.int32)
    T["body_pos"] = np.array([b["pos"] for b in sc.bodies]).reshape(-1); T["body_quat"] = np.array([b["quat"] for b in sc.bodies]).reshape(-1)
    T["body_gravcomp"] = np.array([b["gravcomp"] for b in sc.bodies])
    mass = np.zeros(nb); ipos = np.zeros((nb, 3)); Ibody = np.zeros((nb, 3, 3))
    for i, b in enumerate(sc.bodies):
        if i == 0:
            continue
        parts = []
        for g in b["geoms"]:
            m, Id = geom_mass_inertia(g["type"], g["size"], g["density"])
            if m > 0:
                R = quat2mat(g["quat"]); parts.append((m, g["pos"], R @ np.diag(Id) @ R.T))
        mt = sum(p[0] for p in parts)
        if mt > 0:
            com = sum(p[0] * p[1] for p in parts) / mt
            I = np.zeros((3, 3))
            for m, p, Ig in parts:
                d = p - com
                I += Ig + m * (d @ d * np.eye(3) - np.outer(d, d))          # parallel-axis theorem
            mass[i], ipos[i], Ibody[i] = mt, com, I
        mass[i] = max(mass[i], boundmass)
    T["body_mass"] = mass; T["body_ipos"] = ipos.reshape(-1); T["body_Itensor"] = Ibody
    # principal axes (any proper rotation that diagonalises: the pair (inertia, iquat) is what matters)
    inertia = np.zeros((nb, 3)); iquat = np.zeros((nb, 4)); iquat[:, 0] = 1
    for i in range(1, nb):
        w, V = np.linalg.eigh(Ibody[i])
        if np.allclose(Ibody[i], np.diag(np.diag(Ibody[i])), atol=1e-14 * max(1.0, np.abs(Ibody[i]).max())):
            w, V = np.diag(Ibody[i]).copy(), np.eye(3)
        if np.linalg.det(V) < 0:
            V[:, 2] = -V[:, 2]
        inertia[i] = np.maximum(w, boundinertia); iquat[i] = _mat2quat(V)
    T["body_inertia"] = inertia.reshape(-1); T["body_iquat"] = iquat.reshape(-1)
    # joints, dofs, qpos0
    jt, jb, jp, ja, jr, jl, jq, jd = [], [], [], [], [], [], [], []
    q0 = []; dof_body, dof_jnt, dof_damp, dof_arm = [], [], [], []
    body_jntadr = np.full(nb, -1, np.int32); body_jntnum = np.zeros(nb, np.int32); body_dofadr = np.full(nb, -1, np.int32); body_dofnum = np.zeros(nb, np.int32)
    for i, b in enumerate(sc.bodies):
        for j in b["joints"]:
            if body_jntnum[i] == 0:
                body_jntadr[i] = len(jt); body_dofadr[i] = len(dof_body)
            body_jntnum[i] += 1
            jt.append(j["type"]); jb.append(i); jp.append(j["pos"]); ja.append(j["axis"])
            jr.append(j["range"] if j["range"] is not None else (0.0, 0.0)); jl.append(0 if j["range"] is None else 1)
            jq.append(len(q0)); jd.append(len(dof_body))
            nd = {FREE: 6, BALL: 3}.get(j["type"], 1)
            if j["type"] == FREE:
                q0 += list(b["pos"]) + list(b["quat"])            # a free body's reference pose is where the description puts it (top level)
            elif j["type"] == BALL:
                q0 += [1.0, 0, 0, 0]
            else:
                q0 += [0.0]
            for _ in range(nd):
                dof_body.append(i); dof_jnt.append(len(jt) - 1); dof_damp.append(j["damping"]); dof_arm.append(j["armature"])
            body_dofnum[i] += nd
    nv = len(dof_body)
    T["jnt_type"] = np.array(jt, np.int32); T["jnt_bodyid"] = np.array(jb, np.int32); T["jnt_pos"] = np.array(jp).reshape(-1); T["jnt_axis"] = np.array(ja).reshape(-1)
    T["jnt_range"] = np.array(jr, float).reshape(-1); T["jnt_limited"] = np.array(jl, np.int32)
    T["jnt_qposadr"] = np.array(jq, np.int32); T["jnt_dofadr"] = np.array(jd, np.int32)
    T["qpos0"] = np.array(q0, float)
    T["body_jntadr"], T["body_jntnum"], T["body_dofadr"], T["body_dofnum"] = body_jntadr, body_jntnum, body_dofadr, body_dofnum
    T["dof_bodyid"] = np.array(dof_body, np.int32); T["dof_jntid"] = np.array(dof_jnt, np.int32)
    T["dof_damping"] = np.array(dof_damp, float); T["dof_armature"] = np.array(dof_arm, float)
    # dof tree: the dof before it on the same body, else the last dof of the nearest ancestor that has one
    par = np.full(nv, -1, np.int32)
    for d in range(nv):
        b = dof_body[d]
        if d > 0 and dof_body[d - 1] == b:
            par[d] = d - 1
            continue
        a = sc.bodies[b]["parent"]
        while a > 0 and body_dofnum[a] == 0:
            a = sc.bodies[a]["parent"]
        if a > 0:
            par[d] = body_dofadr[a] + body_dofnum[a] - 1
    T["dof_parentid"] = par
    madr = np.zeros(nv, np.int32); n = 0
    for d in range(nv):
        madr[d] = n; k = d
        while k >= 0:
            n += 1; k = par[k]
    T["dof_Madr"] = madr; T["nM"] = n
    # geoms
    gt, gb, gs, gp, gq, gr = [], [], [], [], [], []
    for i, b in enumerate(sc.bodies):
        for g in b["geoms"]:
            gt.append(g["type"]); gb.append(i); gs.append(g["size"]); gp.append(g["pos"]); gq.append(g["quat"]); gr.append(geom_rbound(g["type"], g["size"]))
    T["geom_type"] = np.array(gt, np.int32); T["geom_bodyid"] = np.array(gb, np.int32); T["geom_size"] = np.array(gs).reshape(-1)
    T["geom_pos"] = np.array(gp).reshape(-1); T["geom_quat"] = np.array(gq).reshape(-1); T["geom_rbound"] = np.array(gr, float)
    # mj_setConst: M^-1 at qpos0 through the Jacobian-sum mass matrix
    tree = Tree(T)
    M = tree.mass_matrix(T["qpos0"]); Jp, Jr, _, _ = tree.jacobians(T["qpos0"])
    Minv = np.linalg.inv(M); dinv = np.diag(Minv)
    dofw = np.zeros(nv)
    for j in range(len(jt)):
        a = jd[j]
        if jt[j] == FREE:
            dofw[a:a + 3] = dinv[a:a + 3].mean(); dofw[a + 3:a + 6] = dinv[a + 3:a + 6].mean()
        elif jt[j] == BALL:
            dofw[a:a + 3] = dinv[a:a + 3].mean()
        else:
            dofw[a] = dinv[a]
    bodyw = np.zeros((nb, 2))
    moving = np.zeros(nb, bool)
    for i in range(1, nb):
        moving[i] = body_dofnum[i] > 0 or moving[sc.bodies[i]["parent"]]
        if moving[i]:
            bodyw[i, 0] = np.trace(Jp[i] @ Minv @ Jp[i].T) / 3; bodyw[i, 1] = np.trace(Jr[i] @ Minv @ Jr[i].T) / 3
    T["dof_invweight0"] = dofw; T["body_invweight0"] = bodyw.reshape(-1); T["meaninertia"] = float(np.trace(M) / nv) if nv else 1.0
    return T


def _mat2quat(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2; q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    else:
        i = int(np.argmax(np.diag(R))); j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = [0, 0, 0, 0]; q[0] = (R[k, j] - R[j, k]) / s; q[1 + i] = 0.25 * s; q[1 + j] = (R[j, i] + R[i, j]) / s; q[1 + k] = (R[k, i] + R[i, k]) / s
    return _unit(q)


# ------------------------------------------------------------------ the scenes, from their descriptions
def scene_s24(pen_half=0.175):
    """4 free boxes (shared model: the mean half-extent 0.0875 = 0.05 x 1.75; per-env sizes replace it at load time) in a
    square pen of 4 static wall boxes on the world/empty.xml floor; staggered column z = 0.15 + 0.30 k; gravity -9.81, dt 0.005"""
    s = Scene()
    s.geom(0, PLANE, (0, 0, 0.05))
    t, h = 0.025, 0.75; L = pen_half + 2 * t
    for sz, p in (((t, L, h), (pen_half + t, 0, h)), ((t, L, h), (-(pen_half + t), 0, h)), ((L, t, h), (0, pen_half + t, h)), ((L, t, h), (0, -(pen_half + t), h))):
        s.geom(0, BOX, sz, p)
    for k in range(4):
        b = s.body(f"box{k}", 0, (0, 0, 0.15 + 0.30 * k))
        s.joint(b, FREE); s.geom(b, BOX, (0.0875,) * 3)
    return s


def scene_pendulum():
    """/root/reference/model/test/pendulum.xml:2,19-30 (data): three bodies 2 m up on a circle of radius 1, each hung by a ball joint at the
    centre (damping 0.5), geoms sphere / box / cylinder of size 0.1; gravity -0.1"""
    s = Scene(gravity=(0, 0, -0.1))
    s.geom(0, PLANE, (0, 0, 0.05))
    for name, pos, gt in (("sphere", (1, 0, 2), SPHERE), ("cube", (-0.5, 0.866, 2), BOX), ("cylinder", (-0.5, -0.866, 2), CYLINDER)):
        b = s.body(name, 0, pos)
        s.joint(b, BALL, pos=(-pos[0], -pos[1], 0), damping=0.5)
        s.geom(b, gt, (0.1, 0.1, 0.1))
    return s


def scene_arm7(gravcomp=1):
    """the Panda chain of /root/reference/model/test/ridgeback_panda.xml:53-87 as SURVEY.md §8-d D3 restates it for C3: 7 hinges about the
    links' z axes with the file's frames and ranges, fixed base, link inertia from one cylinder per link at density 1000"""
    s = Scene()
    s.geom(0, PLANE, (0, 0, 0.05))
    r = 0.707107
    links = [((0.33, 0, 0.919499), (1, 0, 0, 0), (-2.8973, 2.8973), (0.06, 0.1415), (0, 0, -0.1915)),
             ((0, 0, 0), (r, -r, 0, 0), (-1.7628, 1.7628), (0.06, 0.06), (0, 0, 0)),
             ((0, -0.316, 0), (r, r, 0, 0), (-2.8973, 2.8973), (0.06, 0.075), (0, 0, -0.145)),
             ((0.0825, 0, 0), (r, r, 0, 0), (-3.0718, -0.0698), (0.06, 0.06), (0, 0, 0)),
             ((-0.0825, 0.384, 0), (r, -r, 0, 0), (-2.8973, 2.8973), (0.06, 0.05), (0, 0, -0.26)),
             ((0, 0, 0), (r, r, 0, 0), (-0.0175, 3.7525), (0.05, 0.04), (0, 0, -0.03)),
             ((0.088, 0, 0), (r, r, 0, 0), (-2.8973, 2.8973), (0.04, 0.07), (0, 0, 0.01))]
    parent = 0
    for k, (pos, quat, rng, gs, gp) in enumerate(links):
        b = s.body(f"panda_link{k + 1}", parent, pos, quat, gravcomp=float(gravcomp))
        s.joint(b, HINGE, axis=(0, 0, 1), range_=rng)
        s.geom(b, CYLINDER, gs, gp)
        parent = b
    return s

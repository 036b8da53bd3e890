"""indep_dyn.py — TEST INFRASTRUCTURE: rigid-body dynamics of a kinematic tree WITHOUT any of the recursions the oracle or the product use
(no composite-rigid-body pass, no recursive Newton-Euler, no spatial algebra, no subtree-COM frame): numpy, fp64, from the model's INPUT
tables only (frames, joints, masses, principal inertias, armature).

    M(q)      = sum_b  m_b Jp_b^T Jp_b + Jr_b^T (R_b I_b R_b^T) Jr_b  + diag(armature)         (kinetic energy, term by term)
    bias(q,v) = sum_b  Jp_b^T m_b (a_b - g) + Jr_b^T (Iw_b alpha_b + w_b x Iw_b w_b)           (projected Newton-Euler / Kane, qacc = 0)

with Jp / Jr the world-frame Jacobians of every body's centre of mass / orientation, and a_b, alpha_b the centre-of-mass and angular
accelerations the tree has when qvel is held constant — obtained by FINITE DIFFERENCES of c_b' = Jp_b v and w_b = Jr_b v along the motion
q(t +- eps) = integrate(q, +- eps v).  The Jacobians themselves are checked against finite differences of the forward kinematics
(jacobians_fd).  Conventions are MuJoCo's: free joint = world-frame linear velocity + BODY-frame angular velocity,
ball = body-frame angular velocity, hinge / slide about / along jnt_axis through jnt_pos, displacement from qpos0.

What this pins: mj_crb (A5), mj_rne (A9), mj_mulM / mj_solveM, mj_comPos, mj_comVel as restated in oracle/mjh_oracle.c —
by something that shares no algorithm with them.  Not MuJoCo itself: the conventions above are still this repo's reading of the docs."""
import numpy as np

FREE, BALL, SLIDE, HINGE = 0, 1, 2, 3


def quat2mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def mulquat(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def axisangle_quat(axis, angle):
    axis = np.asarray(axis, float)
    return np.concatenate([[np.cos(angle / 2)], np.sin(angle / 2) * axis])


def rotvec_quat(w):
    """unit quaternion of the rotation vector w (exp map)"""
    a = np.linalg.norm(w)
    if a < 1e-300:
        return np.array([1.0, 0, 0, 0])
    return axisangle_quat(w / a, a)


class Tree:
    def __init__(self, m):
        A = m.array
        self.nb, self.nv, self.nq, self.nj = m.nbody, m.nv, m.nq, m.njnt
        self.par = A("body_parentid").astype(int)
        self.bpos, self.bquat = A("body_pos").reshape(-1, 3).astype(float), A("body_quat").reshape(-1, 4).astype(float)
        self.ipos, self.iquat = A("body_ipos").reshape(-1, 3).astype(float), A("body_iquat").reshape(-1, 4).astype(float)
        self.mass, self.inertia = A("body_mass").astype(float), A("body_inertia").reshape(-1, 3).astype(float)
        self.jadr, self.jnum = A("body_jntadr").astype(int), A("body_jntnum").astype(int)
        self.jtype, self.jpos, self.jaxis = A("jnt_type").astype(int), A("jnt_pos").reshape(-1, 3).astype(float), A("jnt_axis").reshape(-1, 3).astype(float)
        self.jq, self.jd = A("jnt_qposadr").astype(int), A("jnt_dofadr").astype(int)
        self.q0 = A("qpos0").astype(float)
        self.armature = A("dof_armature").astype(float)
        self.jrange, self.jlimited = A("jnt_range").reshape(-1, 2).astype(float), A("jnt_limited").astype(int)

    # ---- forward kinematics at an arbitrary q: frames, and per body the joint columns that act on it directly
    def fk(self, q):
        nb = self.nb
        xpos = np.zeros((nb, 3)); xquat = np.zeros((nb, 4)); xquat[0, 0] = 1
        cols = [[] for _ in range(nb)]          # (dof, "lin" | "rot", world axis, world anchor)
        for b in range(1, nb):
            p = self.par[b]
            j0 = self.jadr[b]
            if self.jnum[b] == 1 and self.jtype[j0] == FREE:
                a = self.jq[j0]
                xpos[b] = q[a:a + 3]; xquat[b] = q[a + 3:a + 7] / np.linalg.norm(q[a + 3:a + 7])
                R = quat2mat(xquat[b])
                for k in range(3):
                    cols[b].append((self.jd[j0] + k, "lin", np.eye(3)[k], None))
                for k in range(3):
                    cols[b].append((self.jd[j0] + 3 + k, "rot", R[:, k], xpos[b].copy()))
                continue
            xpos[b] = xpos[p] + quat2mat(xquat[p]) @ self.bpos[b]; xquat[b] = mulquat(xquat[p], self.bquat[b])
            for j in range(j0, j0 + self.jnum[b]):
                R = quat2mat(xquat[b])
                anchor = xpos[b] + R @ self.jpos[j]; axis = R @ self.jaxis[j]
                t = self.jtype[j]
                if t == BALL:
                    qq = q[self.jq[j]:self.jq[j] + 4]
                    xquat[b] = mulquat(xquat[b], qq / np.linalg.norm(qq)); R = quat2mat(xquat[b]); xpos[b] = anchor - R @ self.jpos[j]
                    for k in range(3):
                        cols[b].append((self.jd[j] + k, "rot", R[:, k], anchor))
                elif t == SLIDE:
                    xpos[b] = xpos[b] + axis * (q[self.jq[j]] - self.q0[self.jq[j]])
                    cols[b].append((self.jd[j], "lin", axis, None))
                else:
                    xquat[b] = mulquat(xquat[b], axisangle_quat(self.jaxis[j], q[self.jq[j]] - self.q0[self.jq[j]]))
                    xpos[b] = anchor - quat2mat(xquat[b]) @ self.jpos[j]
                    cols[b].append((self.jd[j], "rot", axis, anchor))
        return xpos, xquat, cols

    def com_frames(self, q):
        """-> c [nb, 3] world centres of mass, Rw [nb, 3, 3] world orientation of the inertial frames, xquat"""
        xpos, xquat, cols = self.fk(q)
        c = np.zeros((self.nb, 3)); Rw = np.zeros((self.nb, 3, 3))
        for b in range(self.nb):
            c[b] = xpos[b] + quat2mat(xquat[b]) @ self.ipos[b]
            Rw[b] = quat2mat(mulquat(xquat[b], self.iquat[b]))
        return c, Rw, cols

    def jacobians(self, q):
        """analytic world-frame Jacobians of every body's centre of mass (Jp) and orientation (Jr): [nb, 3, nv]"""
        c, Rw, cols = self.com_frames(q)
        Jp, Jr = np.zeros((self.nb, 3, self.nv)), np.zeros((self.nb, 3, self.nv))
        for b in range(1, self.nb):
            a = b
            while a > 0:
                for (d, kind, axis, anchor) in cols[a]:
                    if kind == "lin":
                        Jp[b, :, d] = axis
                    else:
                        Jr[b, :, d] = axis; Jp[b, :, d] = np.cross(axis, c[b] - anchor)
                a = self.par[a]
        return Jp, Jr, c, Rw

    def integrate(self, q, v, h):
        """q (+) h v: the position integration of the conventions above (quaternions by the exponential map, body frame)"""
        q = np.array(q, float)
        for j in range(self.nj):
            a, d, t = self.jq[j], self.jd[j], self.jtype[j]
            if t == FREE:
                q[a:a + 3] += h * v[d:d + 3]
                q[a + 3:a + 7] = mulquat(q[a + 3:a + 7] / np.linalg.norm(q[a + 3:a + 7]), rotvec_quat(h * v[d + 3:d + 6]))
            elif t == BALL:
                q[a:a + 4] = mulquat(q[a:a + 4] / np.linalg.norm(q[a:a + 4]), rotvec_quat(h * v[d:d + 3]))
            else:
                q[a] += h * v[d]
        return q

    def jacobians_fd(self, q, eps=1e-6):
        """the same Jacobians by central differences of the forward kinematics alone"""
        Jp, Jr = np.zeros((self.nb, 3, self.nv)), np.zeros((self.nb, 3, self.nv))
        for d in range(self.nv):
            e = np.zeros(self.nv); e[d] = 1
            cp, Rp, _ = self.com_frames(self.integrate(q, e, eps)); cm, Rm, _ = self.com_frames(self.integrate(q, e, -eps))
            Jp[:, :, d] = (cp - cm) / (2 * eps)
            for b in range(self.nb):
                S = Rp[b] @ Rm[b].T                     # = exp([2 eps w]x)
                Jr[b, :, d] = np.array([S[2, 1] - S[1, 2], S[0, 2] - S[2, 0], S[1, 0] - S[0, 1]]) / (4 * eps)
        return Jp, Jr

    def mass_matrix(self, q):
        Jp, Jr, c, Rw = self.jacobians(q)
        M = np.diag(self.armature.copy()) if self.nv else np.zeros((0, 0))
        for b in range(1, self.nb):
            Iw = Rw[b] @ np.diag(self.inertia[b]) @ Rw[b].T
            M += self.mass[b] * Jp[b].T @ Jp[b] + Jr[b].T @ Iw @ Jr[b]
        return M

    def bias(self, q, v, gravity, eps=1e-6):
        """generalised bias force C(q, v) v + g(q) (MuJoCo's qfrc_bias): the force that keeps qacc = 0"""
        Jp, Jr, c, Rw = self.jacobians(q)
        Jpp, Jrp, _, _ = self.jacobians(self.integrate(q, v, eps)); Jpm, Jrm, _, _ = self.jacobians(self.integrate(q, v, -eps))
        out = np.zeros(self.nv)
        g = np.asarray(gravity, float)
        for b in range(1, self.nb):
            acc = (Jpp[b] @ v - Jpm[b] @ v) / (2 * eps)            # centre-of-mass acceleration at constant qvel
            alpha = (Jrp[b] @ v - Jrm[b] @ v) / (2 * eps)          # angular acceleration at constant qvel
            w = Jr[b] @ v
            Iw = Rw[b] @ np.diag(self.inertia[b]) @ Rw[b].T
            out += Jp[b].T @ (self.mass[b] * (acc - g)) + Jr[b].T @ (Iw @ alpha + np.cross(w, Iw @ w))
        return out

    def energy(self, q, v, gravity):
        c, _, _ = self.com_frames(q)
        return 0.5 * v @ self.mass_matrix(q) @ v, -float(sum(self.mass[b] * np.dot(gravity, c[b]) for b in range(1, self.nb)))

    def random_configuration(self, rng, spread=1.0):
        """hinges / slides inside their ranges (or +- spread about qpos0), random unit quaternions, free positions near qpos0"""
        q = self.q0.copy()
        for j in range(self.nj):
            a, t = self.jq[j], self.jtype[j]
            if t == FREE:
                q[a:a + 3] += rng.uniform(-0.5, 0.5, 3); u = rng.normal(size=4); q[a + 3:a + 7] = u / np.linalg.norm(u)
            elif t == BALL:
                u = rng.normal(size=4); q[a:a + 4] = u / np.linalg.norm(u)
            elif self.jlimited[j] and self.jrange[j, 1] > self.jrange[j, 0]:
                q[a] = rng.uniform(*self.jrange[j])
            else:
                q[a] += rng.uniform(-spread, spread) * (0.3 if t == SLIDE else 1.0)
        return q

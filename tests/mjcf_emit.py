"""Test infrastructure: writes a compiled mjh_model back out as MJCF text, so that the SAME model can be handed to the
reference's third-party engine (MuJoCo 2.3.7, absent from this image and from the GPU box) wherever it is installed:
tests/test_mujoco_reference.py steps both and reports the differences.  Everything is written
explicitly (inertials, frames, the candidate pair list as <contact><pair>, solver = PGS / pyramidal), so MuJoCo's
compiler has nothing left to derive differently."""
import numpy as np

_GEOM = {0: "plane", 2: "sphere", 3: "capsule", 4: "ellipsoid", 5: "cylinder", 6: "box", 7: "mesh"}
_JNT = {0: "free", 1: "ball", 2: "slide", 3: "hinge"}


def _v(a):
    return " ".join(repr(float(x)) for x in np.asarray(a).reshape(-1))


def emit_mjcf(m, geom_size=None, body_mass=None, body_inertia=None):
    """m: mujoco_sim_amd.Model.  geom_size / body_mass / body_inertia: per-env overrides (S24 draws them per env)."""
    A = m.array
    nb, ng, nj = m.c.nbody, m.c.ngeom, m.c.njnt
    gsize = np.asarray(geom_size if geom_size is not None else A("geom_size")).reshape(ng, 3)
    bmass = np.asarray(body_mass if body_mass is not None else A("body_mass")).reshape(nb)
    binert = np.asarray(body_inertia if body_inertia is not None else A("body_inertia")).reshape(nb, 3)
    o = m.c.opt
    dis = int(o.disableflags)
    flags = []
    for bit, name in ((1 << 0, "constraint"), (1 << 1, "equality"), (1 << 2, "frictionloss"), (1 << 3, "limit"), (1 << 4, "contact"),
                      (1 << 5, "passive"), (1 << 6, "gravity"), (1 << 8, "warmstart"), (1 << 9, "filterparent"), (1 << 11, "refsafe")):
        if dis & bit:
            flags.append(f'{name}="disable"')
    out = ['<mujoco>', '  <compiler angle="radian" autolimits="false" boundmass="0" boundinertia="0"/>',
           f'  <option timestep="{o.timestep!r}" gravity="{_v(o.gravity)}" iterations="{o.iterations}" tolerance="{o.tolerance!r}" '
           f'impratio="{o.impratio!r}" noslip_iterations="{o.noslip_iterations}" noslip_tolerance="{o.noslip_tolerance!r}" '
           f'solver="PGS" cone="pyramidal" jacobian="dense" integrator="Euler" collision="predefined">',
           f'    <flag energy="enable" {" ".join(flags)}/>', '  </option>']
    if m.c.nmesh:
        out.append('  <asset>')
        va, vn, vv = A("mesh_vertadr"), A("mesh_vertnum"), A("mesh_vert").reshape(-1, 3)
        for k in range(m.c.nmesh):
            out.append(f'    <mesh name="mesh{k}" vertex="{_v(vv[va[k]:va[k] + vn[k]])}"/>')
        out.append('  </asset>')
    parent = A("body_parentid"); children = {b: [] for b in range(nb)}
    for b in range(1, nb):
        children[int(parent[b])].append(b)
    gb, gt = A("geom_bodyid"), A("geom_type")
    jb, jt = A("jnt_bodyid"), A("jnt_type")
    gname = lambda g: f"g{g}"

    def geoms(b, ind):
        for g in range(ng):
            if gb[g] != b:
                continue
            t = int(gt[g])
            size = gsize[g] if t != 0 else [0, 0, 0.05]
            extra = f'mesh="mesh{int(A("geom_dataid")[g])}"' if t == 7 else f'size="{_v(size)}"'
            out.append(f'{ind}<geom name="{gname(g)}" type="{_GEOM[t]}" {extra} pos="{_v(A("geom_pos")[3*g:3*g+3])}" '
                       f'quat="{_v(A("geom_quat")[4*g:4*g+4])}" friction="{_v(A("geom_friction")[3*g:3*g+3])}" condim="{int(A("geom_condim")[g])}" '
                       f'contype="{int(A("geom_contype")[g])}" conaffinity="{int(A("geom_conaffinity")[g])}" solref="{_v(A("geom_solref")[2*g:2*g+2])}" '
                       f'solimp="{_v(A("geom_solimp")[5*g:5*g+5])}" margin="{float(A("geom_margin")[g])!r}" gap="{float(A("geom_gap")[g])!r}"/>')

    def body(b, ind):
        out.append(f'{ind}<body name="b{b}" pos="{_v(A("body_pos")[3*b:3*b+3])}" quat="{_v(A("body_quat")[4*b:4*b+4])}" '
                   f'gravcomp="{float(A("body_gravcomp")[b])!r}">')
        if bmass[b] > 0:
            out.append(f'{ind}  <inertial pos="{_v(A("body_ipos")[3*b:3*b+3])}" quat="{_v(A("body_iquat")[4*b:4*b+4])}" mass="{float(bmass[b])!r}" '
                       f'diaginertia="{_v(binert[b])}"/>')
        for j in range(nj):
            if jb[j] != b:
                continue
            t = int(jt[j])
            if t == 0:
                out.append(f'{ind}  <joint name="j{j}" type="free"/>')
                continue
            d = int(A("jnt_dofadr")[j])
            lim = int(A("jnt_limited")[j])
            rng = f' limited="true" range="{_v(A("jnt_range")[2*j:2*j+2])}"' if lim else ' limited="false"'
            ax = f' axis="{_v(A("jnt_axis")[3*j:3*j+3])}"' if t != 1 else ""
            out.append(f'{ind}  <joint name="j{j}" type="{_JNT[t]}" pos="{_v(A("jnt_pos")[3*j:3*j+3])}"{ax}{rng} damping="{float(A("dof_damping")[d])!r}" '
                       f'stiffness="{float(A("jnt_stiffness")[j])!r}" armature="{float(A("dof_armature")[d])!r}" frictionloss="{float(A("dof_frictionloss")[d])!r}" '
                       f'margin="{float(A("jnt_margin")[j])!r}" solreflimit="{_v(A("jnt_solref")[2*j:2*j+2])}" solimplimit="{_v(A("jnt_solimp")[5*j:5*j+5])}" '
                       f'solreffriction="{_v(A("dof_solref")[2*d:2*d+2])}" solimpfriction="{_v(A("dof_solimp")[5*d:5*d+5])}"/>')
        geoms(b, ind + "  ")
        for c in children[b]:
            body(c, ind + "  ")
        out.append(f'{ind}</body>')

    out.append('  <worldbody>')
    geoms(0, "    ")
    for c in children[0]:
        body(c, "    ")
    out.append('  </worldbody>')
    if m.c.npair:
        out.append('  <contact>')
        g1, g2 = A("pair_geom1"), A("pair_geom2")
        for i in range(m.c.npair):
            out.append(f'    <pair geom1="{gname(int(g1[i]))}" geom2="{gname(int(g2[i]))}"/>')
        out.append('  </contact>')
    if m.c.neq:
        out.append('  <equality>')
        for e in range(m.c.neq):
            j1, j2 = int(A("eq_obj1id")[e]), int(A("eq_obj2id")[e])
            second = f' joint2="j{j2}"' if j2 >= 0 else ""
            out.append(f'    <joint joint1="j{j1}"{second} polycoef="{_v(A("eq_data")[11*e:11*e+5])}" solref="{_v(A("eq_solref")[2*e:2*e+2])}" '
                       f'solimp="{_v(A("eq_solimp")[5*e:5*e+5])}" active="{"true" if int(A("eq_active")[e]) else "false"}"/>')
        out.append('  </equality>')
    out.append('</mujoco>')
    return "\n".join(out)

"""The product's MJCF reader (csrc/mjcf_loader.cpp: its own XML parser and compiler front end) against an INDEPENDENT reading of the
reference's robot files with the standard library's ElementTree (VERDICT r05 weak #1b: the oracle consumes the product's compiled model, so a
loader defect is invisible to the GPU parity tests).  For pr2, tiago, hsrb4s, ridgeback_panda and armar6 (model/test/<robot>/<robot>.xml — the
models C4 and the robot fixtures are compiled from) every kinematic and inertial table the file states explicitly is rebuilt here — body tree in
document order, frames, explicit <inertial> (mass, centre, inertia tensor), joints (type, axis, anchor, range under autolimits, damping,
armature, friction loss), joint equalities with their polycoef, primitive geoms — and compared (i) with the model the product's loader compiles
from the same file and (ii) with the committed fixture tables under tests/golden (what the GPU box steps).  CPU container only: the reference
tree does not exist on the GPU box."""
import os
import xml.etree.ElementTree as ET

import numpy as np
import pytest

import mujoco_sim_amd as ms
from indep_dyn import quat2mat
from mujoco_sim_amd.tables import load_model_tables

REF = "/root/reference/model/test"
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROBOTS = {"pr2": "pr2/pr2.xml", "tiago": "tiago/tiago.xml", "hsrb4s": "hsrb4s/hsrb4s.xml", "ridgeback_panda": "ridgeback_panda/ridgeback_panda.xml",
          "armar6": "armar/armar6.xml"}
GEOM = {"plane": 0, "sphere": 2, "capsule": 3, "ellipsoid": 4, "cylinder": 5, "box": 6, "mesh": 7}
JNT = {"free": 0, "ball": 1, "slide": 2, "hinge": 3}

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "pr2", "pr2.xml")), reason="reference tree absent (GPU box)")


def _f(el, key, default):
    v = el.get(key)
    return np.array([float(x) for x in v.split()]) if v is not None else np.array(default, float)


def read_robot(path):
    """-> dict of lists, bodies in document (depth-first) order as MuJoCo numbers them"""
    root = ET.parse(path).getroot()
    comp = root.find("compiler")
    assert comp is not None and comp.get("angle") == "radian", "these files are in radians (checked, not assumed)"
    autolimits = comp.get("autolimits") == "true"        # (MuJoCo 2.3.7's default is false: a range alone limits nothing)
    assert root.find("default") is None or len(root.find("default")) == 0, "no default classes in these files"
    B = dict(name=["world"], parent=[0], pos=[np.zeros(3)], quat=[np.array([1.0, 0, 0, 0])], inertial=[None], gravcomp=[0.0])
    J = dict(type=[], body=[], pos=[], axis=[], range=[], limited=[], damping=[], armature=[], frictionloss=[], name=[])
    Gm = dict(type=[], body=[], size=[], pos=[], quat=[])

    def walk(el, parent):
        for c in el:
            if c.tag == "geom":
                t = c.get("type", "sphere")
                Gm["type"].append(GEOM[t]); Gm["body"].append(parent); Gm["size"].append(_f(c, "size", [0, 0, 0])); Gm["pos"].append(_f(c, "pos", [0, 0, 0]))
                q = _f(c, "quat", [1, 0, 0, 0]); Gm["quat"].append(q / np.linalg.norm(q))
            elif c.tag == "body":
                i = len(B["name"])
                q = _f(c, "quat", [1, 0, 0, 0])
                B["name"].append(c.get("name")); B["parent"].append(parent); B["pos"].append(_f(c, "pos", [0, 0, 0])); B["quat"].append(q / np.linalg.norm(q))
                B["gravcomp"].append(float(c.get("gravcomp", 0)))
                ine = c.find("inertial")
                if ine is not None:
                    assert ine.get("fullinertia") is None
                    iq = _f(ine, "quat", [1, 0, 0, 0])
                    B["inertial"].append((float(ine.get("mass")), _f(ine, "pos", [0, 0, 0]), iq / np.linalg.norm(iq), _f(ine, "diaginertia", [0, 0, 0])))
                else:
                    B["inertial"].append(None)
                for j in c:                                  # joints of this body, in order
                    if j.tag not in ("joint", "freejoint"):
                        continue
                    t = "free" if j.tag == "freejoint" else j.get("type", "hinge")
                    ax = _f(j, "axis", [0, 0, 1])
                    J["type"].append(JNT[t]); J["body"].append(i); J["pos"].append(_f(j, "pos", [0, 0, 0])); J["axis"].append(ax / np.linalg.norm(ax))
                    J["range"].append(_f(j, "range", [0, 0])); J["limited"].append(int(j.get("limited") == "true" or (autolimits and j.get("limited") is None and j.get("range") is not None)))
                    J["damping"].append(float(j.get("damping", 0))); J["armature"].append(float(j.get("armature", 0))); J["frictionloss"].append(float(j.get("frictionloss", 0)))
                    J["name"].append(j.get("name"))
                walk(c, i)

    for wb in root.findall("worldbody"):
        walk(wb, 0)
    eq = []
    for e in root.findall("equality"):
        for c in e:
            assert c.tag == "joint", "only joint equalities in these files"
            eq.append((c.get("joint1"), c.get("joint2"), _f(c, "polycoef", [0, 1, 0, 0, 0])))
    return B, J, Gm, eq


def _compare(name, m, B, J, Gm, eq, geoms):
    A = m.array
    nb = len(B["name"])
    assert m.nbody == nb and m.njnt == len(J["type"])
    assert np.array_equal(A("body_parentid"), B["parent"])
    np.testing.assert_allclose(A("body_pos").reshape(-1, 3), np.array(B["pos"]), rtol=0, atol=1e-15)
    np.testing.assert_allclose(A("body_quat").reshape(-1, 4), np.array(B["quat"]), rtol=0, atol=1e-15)
    np.testing.assert_allclose(A("body_gravcomp"), B["gravcomp"], atol=0)
    mass, ipos = A("body_mass"), A("body_ipos").reshape(-1, 3)
    iq, inr = A("body_iquat").reshape(-1, 4), A("body_inertia").reshape(-1, 3)
    nexp = 0
    for b in range(1, nb):
        if B["inertial"][b] is None:
            continue
        mm, p, q, d = B["inertial"][b]; nexp += 1
        assert abs(mass[b] - mm) <= 1e-15 * mm, (name, b)
        np.testing.assert_allclose(ipos[b], p, atol=1e-15)
        R1, R2 = quat2mat(iq[b]), quat2mat(q)
        np.testing.assert_allclose(R1 @ np.diag(inr[b]) @ R1.T, R2 @ np.diag(d) @ R2.T, rtol=1e-12, atol=1e-15 * max(1.0, d.max()))
    assert nexp == {"pr2": 43, "tiago": 29, "hsrb4s": 26, "ridgeback_panda": 5, "armar6": 19}[name], "bodies that state their inertia (ridgeback_panda's arm links do not: theirs comes from the geoms)"
    assert np.array_equal(A("jnt_type"), J["type"]) and np.array_equal(A("jnt_bodyid"), J["body"])
    np.testing.assert_allclose(A("jnt_pos").reshape(-1, 3), np.array(J["pos"]), atol=1e-15)
    hs = np.isin(J["type"], (2, 3))
    np.testing.assert_allclose(A("jnt_axis").reshape(-1, 3)[hs], np.array(J["axis"])[hs], atol=1e-15)
    assert np.array_equal(A("jnt_limited"), J["limited"])
    lim = np.array(J["limited"], bool)
    np.testing.assert_allclose(A("jnt_range").reshape(-1, 2)[lim], np.array(J["range"])[lim], atol=0)
    # per-dof tables: one entry per hinge / slide, six for the free joint
    dd, da, df, k = A("dof_damping"), A("dof_armature"), A("dof_frictionloss"), 0
    for j, t in enumerate(J["type"]):
        n = {0: 6, 1: 3}.get(t, 1)
        assert np.all(dd[k:k + n] == J["damping"][j]) and np.all(da[k:k + n] == J["armature"][j]) and np.all(df[k:k + n] == J["frictionloss"][j]), (name, J["name"][j])
        assert A("jnt_dofadr")[j] == k
        k += n
    assert k == m.nv
    # joint equalities: obj ids by joint NAME, polycoef as the file gives it
    assert m.neq == len(eq)
    names = J["name"]
    for e, (j1, j2, pc) in enumerate(eq):
        assert A("eq_type")[e] == 2 and A("eq_obj1id")[e] == names.index(j1) and A("eq_obj2id")[e] == names.index(j2)       # mjEQ_JOINT
        np.testing.assert_allclose(A("eq_data").reshape(-1, 11)[e, :len(pc)], pc, atol=0)
    if geoms:      # primitive geoms in document order (mesh geoms are dropped or become hulls depending on the load mode)
        prim = [i for i, t in enumerate(Gm["type"]) if t != 7]
        gt, gb = A("geom_type"), A("geom_bodyid")
        mine = [i for i in range(m.ngeom) if gt[i] != 7]
        assert len(mine) == len(prim), (name, len(mine), len(prim))
        for a, b in zip(mine, prim):
            assert gt[a] == Gm["type"][b] and gb[a] == Gm["body"][b]
            n = {2: 1, 3: 2, 5: 2}.get(Gm["type"][b], 3)
            np.testing.assert_allclose(A("geom_size").reshape(-1, 3)[a][:n], Gm["size"][b][:n], atol=0)
            np.testing.assert_allclose(A("geom_pos").reshape(-1, 3)[a], Gm["pos"][b], atol=1e-15)
            np.testing.assert_allclose(A("geom_quat").reshape(-1, 4)[a], Gm["quat"][b], atol=1e-15)


@pytest.mark.parametrize("name", list(ROBOTS))
def test_product_loader_reads_the_reference_robot_files_as_an_independent_parser_does(name, lib):
    path = os.path.join(REF, ROBOTS[name])
    B, J, Gm, eq = read_robot(path)
    lib.mjh_load_set_mesh_mode(0); lib.mjh_load_set_bounds(1e-6, 1e-6)     # (the wrapper's own floor, mj_sim.cpp:584-590: bodies that only carry meshes stay well-posed)
    try:
        m = ms.load_mjcf(path=path)
    finally:
        lib.mjh_load_set_mesh_mode(1); lib.mjh_load_set_bounds(0.0, 0.0)
    _compare(name, m, B, J, Gm, eq, geoms=True)
    assert {"pr2": (45, 49, 6), "tiago": (31, 35, 0), "hsrb4s": (28, 32, 7), "ridgeback_panda": (16, 20, 0), "armar6": (21, 25, 0)}[name] == (m.nbody, m.nv, m.neq)


@pytest.mark.parametrize("name,fixture", [("pr2", "pr2"), ("pr2", "pr2_mesh"), ("tiago", "tiago"), ("hsrb4s", "hsrb4s"), ("ridgeback_panda", "ridgeback_panda"),
                                           ("ridgeback_panda", "ridgeback_panda_mesh"), ("armar6", "armar6_mesh")])
def test_committed_robot_fixtures_are_what_the_reference_files_say(name, fixture):
    """the tables the GPU box steps (tests/golden/robot_*.npz) against the independent reading of the file they were compiled from (boundmass /
    boundinertia of the fixture build only touch bodies without an explicit inertia)"""
    B, J, Gm, eq = read_robot(os.path.join(REF, ROBOTS[name]))
    m, _ = load_model_tables(os.path.join(G, f"robot_{fixture}.npz"))
    _compare(name, m, B, J, Gm, eq, geoms=not fixture.endswith("_mesh"))

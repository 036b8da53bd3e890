"""Randomised kinematic trees against the oracle, one step at a time from the oracle's state (teacher-forced), in both memory
layouts: chains deeper than the robots' (the wave factor's 16- and 32-ancestor forms and its generic fall-back), bushy trees, several
trees side by side (short trees factored four at a time or one per lane), more than 128 dofs (dof-by-dof solves instead of the level
solves), hinge / slide / ball joints with limits, damping, friction loss, armature, links landing on a floor.  What the reference's
robots (tests/golden/robot_*) do not reach: their trees stop at depth 15 and 49 dofs."""
import numpy as np
import pytest

import mujoco_sim_amd as ms
import orc
from helpers import D, set_opt

pytestmark = pytest.mark.gpu

# (kind, bodies of the main tree, extra trees)
# (a tree of the many-body layout may have at most 64 dofs: engine.hip maps one dof of a constraint row per lane)
CASES = {"chain24": ("chain", 24, 0), "chain44": ("chain", 44, 0), "bushy44": ("bushy", 44, 0), "mixed40+3": ("mixed", 40, 3),
         "forest3x40": ("forest", 40, 2), "two+6": ("mixed", 18, 6)}


def _unit(rng):
    a = rng.normal(size=3)
    return a / np.linalg.norm(a)


def _tree_model(lib, rng, kind, nb, nextra):
    b = lib.mjh_builder_create()
    set_opt(lib, b, timestep=0.003)
    lib.mjh_builder_add_geom(b, b"floor", 0, 0, D(0, 0, 0.05), None, None, None, 3, 0, 1, -1)       # collides with the links only
    bodies = []

    def link(parent, pos, first_joint):
        bd = lib.mjh_builder_add_body(b, None, parent, D(*pos), None, 0.0)
        r = rng.uniform()
        if first_joint == "free":
            lib.mjh_builder_add_joint(b, None, bd, 0, None, None, None, 0, 0, 0, 0, 0)
        elif r < 0.80 or first_joint == "hinge":
            rg = D(-rng.uniform(0.4, 1.2), rng.uniform(0.4, 1.2)) if rng.uniform() < 0.5 else None
            lib.mjh_builder_add_joint(b, None, bd, 3, None, D(*_unit(rng)), rg, rng.uniform(0, 0.5), 0.0, 0.01, rng.uniform(0, 0.3) if rng.uniform() < 0.3 else 0.0, 0.0)
        elif r < 0.92:
            lib.mjh_builder_add_joint(b, None, bd, 2, None, D(*_unit(rng)), D(-0.05, 0.05), rng.uniform(0.1, 1.0), 0.0, 0.01, 0.0, 0.0)
        else:
            lib.mjh_builder_add_joint(b, None, bd, 1, None, None, None, rng.uniform(0.05, 0.3), 0.0, 0.01, 0.0, 0.0)
        lib.mjh_builder_add_geom(b, None, bd, 3 if rng.uniform() < 0.7 else 2, D(0.025, rng.uniform(0.03, 0.06), 0), None, None, None, 3, 1, 0, -1)
        return bd

    ntrees = 1 + (nextra if kind == "forest" else 0)                       # forest: several big trees (more than 128 dofs in all)
    for t in range(ntrees):
        bodies = []
        root = link(0, (0.8 * t, 0, 0.45 if kind == "chain" else 0.25), "free" if kind != "chain" else "hinge")
        bodies.append(root)
        for i in range(1, nb):
            parent = bodies[-1] if kind == "chain" or (kind in ("mixed", "forest") and rng.uniform() < 0.6) else bodies[int(rng.integers(0, len(bodies)))]
            bodies.append(link(parent, 0.09 * _unit(rng) + np.array([0.04, 0, 0.01]), None))
    for k in range(0 if kind == "forest" else nextra):                     # further trees: short chains on free joints
        base = link(0, (0.5 + 0.3 * k, 0.4, 0.3 + 0.1 * k), "free")
        for j in range(int(rng.integers(0, 4))):
            base = link(base, 0.08 * _unit(rng), "hinge")
    m = ms.Model(lib.mjh_builder_compile(b), lib)
    lib.mjh_builder_destroy(b)
    m.c.maxcon = 48; m.c.maxefc = 48 * 4 + 3 * m.nv
    return m


@pytest.mark.parametrize("layout", [1, 2], ids=["lds-resident", "global-pools"])
@pytest.mark.parametrize("case", list(CASES))
def test_random_trees_match_the_oracle_step_by_step(lib, case, layout):
    kind, nb, nextra = CASES[case]
    rng = np.random.default_rng(sum(map(ord, case)))
    m = _tree_model(lib, rng, kind, nb, nextra)
    lib.mjh_set_layout_policy(layout)
    try:
        nenv = 3
        e = ms.Engine(m, nenv)
        e.set_controlled_dofs(np.zeros(m.nv, dtype=np.int32))
        d = orc.OrcData(m.ptr); d.call("reset")
        v0 = 0.5 * rng.normal(size=m.nv)
        d.f("qvel")[:] = v0
        worst = dict(q=0.0, v=0.0, a=0.0, f=0.0); ncon_seen = 0; agree = 0
        nsteps = 70
        for k in range(nsteps):
            e.set_state(qpos=np.tile(d.f("qpos"), (nenv, 1)), qvel=np.tile(d.f("qvel"), (nenv, 1)), warmstart=np.tile(d.f("qacc_warmstart"), (nenv, 1)))
            e.step(1, True); d.step(1, 1)
            _, q, v, w = e.get_state(); st = e.get_stats()
            assert np.array_equal(q[0], q[-1]) and np.array_equal(v[0], v[-1])
            assert st[0, 3] == 0 and d.i("warn") == 0, (k, st[0], d.i("warn"))
            ncon_seen = max(ncon_seen, d.i("ncon"))
            if st[0, 0] != d.i("ncon") or st[0, 1] != d.i("nefc"):
                continue                                   # a contact / limit within rounding of its margin: compared again next step
            agree += 1
            sc = lambda x: max(1.0, np.abs(x).max())
            worst["q"] = max(worst["q"], np.abs(q[0] - d.f("qpos")).max() / sc(d.f("qpos")))
            worst["v"] = max(worst["v"], np.abs(v[0] - d.f("qvel")).max() / sc(d.f("qvel")))
            worst["a"] = max(worst["a"], np.abs(w[0] - d.f("qacc")).max() / sc(d.f("qacc")))
            fi = e.get_field("qfrc_inverse")[0]
            worst["f"] = max(worst["f"], np.abs(fi - d.f("qfrc_inverse")).max() / sc(d.f("qfrc_inverse")))
        print(f"TREE-FUZZ {case} layout {layout}: nv {m.nv} nbody {m.nbody} ntree {m.c.ntree} solver order {e.solver_order()} dense {e.dense_solver()} "
              f"max ncon {ncon_seen} agree {agree}/{nsteps}  rel err qpos {worst['q']:.1e} qvel {worst['v']:.1e} qacc {worst['a']:.1e} qfrc_inverse {worst['f']:.1e}")
        assert agree >= 0.8 * nsteps
        # measured (MI355X, 12 cases x 70 steps): qpos <= 2.1e-7, qvel <= 2.8e-5, qacc <= 3.8e-4, qfrc_inverse <= 2.7e-3
        assert worst["q"] <= 2e-6 and worst["v"] <= 2e-4 and worst["a"] <= 3e-3 and worst["f"] <= 2e-2, worst
        e.close()
    finally:
        lib.mjh_set_layout_policy(0)

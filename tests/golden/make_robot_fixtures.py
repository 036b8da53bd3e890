"""Compiles the reference's robot MJCF files (data: model/test/<robot>/<robot>.xml) with THIS repo's MJCF-subset
loader and stores (i) every field of the compiled mjh_model and (ii) an oracle (fp64) trajectory as fixtures, so
that the GPU box — which has no /root/reference — can rebuild the model and check the HIP path against it.

    python tests/golden/make_robot_fixtures.py        (CPU container only; needs /root/reference)

The first six fixtures are compiled with the mesh assets switched off (mjh_load_set_mesh_mode(0)): they exercise the
articulated-body half of the path — 30-50 dof single trees, equality constraints, joint limits, friction loss, damping,
gravity compensation — with the primitive collision geometry only.  The *_mesh fixtures are PR2 with its 18 STL meshes
(37 mesh geoms colliding as convex hulls; pr2.xml carries the 105 <exclude> pairs that make this well-posed — the tiago,
hsrb4s and ridgeback_panda test files have no exclude list and their hulls overlap permanently at the rest pose, so they
stay primitive-only here); contacts through the generic convex narrow phase make long horizons chaotic, so those
trajectories are short."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import mujoco_sim_amd as ms  # noqa: E402
import orc  # noqa: E402
from mujoco_sim_amd.tables import save_model_tables  # noqa: E402

REF = "/root/reference/model/test"
# name -> (file under model/test, contact capacity: LDS budget; the meshes are skipped anyway)
ROBOTS = {"pr2": ("pr2/pr2.xml", 16), "tiago": ("tiago/tiago.xml", 40), "hsrb4s": ("hsrb4s/hsrb4s.xml", 8),
          "ridgeback_panda": ("ridgeback_panda/ridgeback_panda.xml", 32),
          # composed with the reference's world file (floor plane, condim 4) the way MjSim::init composes them: the robot
          # stands on its wheels / casters (plane-cylinder, plane-sphere, plane-box contacts)
          "pr2_world": ("../world/empty.xml+pr2/pr2.xml", 48), "hsrb4s_world": ("../world/empty.xml+hsrb4s/hsrb4s.xml", 24),
          "pr2_mesh": ("pr2/pr2.xml", 16), "pr2_world_mesh": ("../world/empty.xml+pr2/pr2.xml", 48),
          # tiago and hsrb4s with their meshes: no <exclude> list in these files, so some hulls overlap permanently (deep,
          # ill-conditioned portal-refinement contacts): a short horizon, compared segment by segment with a loose tolerance.
          # (ridgeback_panda's worst pair overlaps by 18 cm: fp32 and fp64 portal refinement leave through different faces)
          "tiago_mesh": ("tiago/tiago.xml", 40), "hsrb4s_mesh": ("hsrb4s/hsrb4s.xml", 32),
          # armar6 (model/test/armar/armar6.xml: a free-floating torso with 19 joints, every geom a mesh, no <exclude> list either: short
          # horizon, loose tolerance, like tiago_mesh / hsrb4s_mesh).  Only WITH its meshes: two of its bodies carry neither an <inertial>
          # nor a primitive geom, so with the mesh assets switched off they are left at boundmass = 1e-6 and the commanded accelerations
          # blow the mass matrix's conditioning (the oracle resets at step 3)
          "armar6_mesh": ("armar/armar6.xml", 48),
          # ridgeback_panda with its meshes: at the wrapper's default the arm's links overlap their grandparents by up to 18 cm for good
          # (fp32 and fp64 portal refinement leave such an overlap through different faces); with the launch argument
          # disable_parent_child_collision_level = 2 (mujoco_sim.launch:7, mujoco_compile.cpp:250-290: a body does not collide with its
          # first two ancestors) the deepest contact at rest is 0.15 mm and the model is well-posed
          "ridgeback_panda_mesh": ("ridgeback_panda/ridgeback_panda.xml", 32),
          # C5, literally: launch/multi_mujoco_sim.launch:3-4 = world pendulum.xml (three bodies on ball joints, damping 0.5,
          # gravity -0.1) + "robot" bowl.xml (37 static mesh geoms); started with a spin so that the bodies meet
          "c5_pendulum_bowl_mesh": ("pendulum.xml+bowl.xml", 16),
          "c4_pr2_world_objects_mesh": ("../world/empty.xml+pr2/pr2.xml+@objects", 72)}
# C4 as SURVEY.md §8-d D3 states it: PR2 on the world floor + a pool of spawnable objects of the types the reference's spawn
# test draws (test/test_spawn_and_destroy.py:13-14,32-41: cubes / spheres / cylinders of size 0.05 * [2, 5]); the pool is
# this repo's own MJCF text, parked beside the robot (slots are toggled at run time: tools/c4_bench.py)
OBJECT_POOL = """<mujoco><worldbody>
""" + "".join(
    f'<body name="object_{k}" pos="{4 + 0.6 * k} 4 0.3"><freejoint/><geom type="{t}" size="{sz}"/></body>\n'
    for k, (t, sz) in enumerate([("box", "0.10 0.10 0.10"), ("sphere", "0.15"), ("cylinder", "0.12 0.12"), ("box", "0.20 0.20 0.20"),
                                 ("sphere", "0.25"), ("cylinder", "0.20 0.20"), ("box", "0.15 0.15 0.15"), ("sphere", "0.10")])) + """</worldbody></mujoco>"""
PC_EXCLUDE = {"ridgeback_panda_mesh": 2}     # mjh_load_set_parent_child_exclude level per fixture (default 0)
QVEL0 = {"c5_pendulum_bowl_mesh": [0.0, 0.1, 2.0, 0.1, 0.0, -2.0, 0.2, -0.1, 0.3]}   # sphere and cube circle towards each other
STEPS = 300
KEEP = (1, 10, 50, 100, 200, 300)
MESH_STEPS = 60
MESH_KEEP = (1, 5, 20, 60)


def command(m, k):
    """a smooth joint-space acceleration command on every hinge/slide dof (what MjHWInterface::write feeds)"""
    nv = m.nv
    jt = m.array("jnt_type"); da = m.array("jnt_dofadr")
    ddq = np.zeros(nv)
    for j in range(m.njnt):
        if jt[j] in (2, 3):
            ddq[da[j]] = 0.8 * np.sin(0.05 * k + 0.37 * j)
    return ddq


def main():
    # the reference writes boundmass = boundinertia = 1e-6 into every file before mj_loadXML (mj_sim.cpp:584-590)
    ms.capi.load().mjh_load_set_bounds(1e-6, 1e-6)
    only = set(sys.argv[1:])          # python make_robot_fixtures.py [name ...]: only these fixtures
    for name, (rel, cap) in ROBOTS.items():
        if only and name not in only:
            continue
        mesh = name.endswith("_mesh")
        ms.capi.load().mjh_load_set_mesh_mode(1 if mesh else 0)
        paths = []
        for r in rel.split("+"):
            if r == "@objects":
                import tempfile
                tf = tempfile.NamedTemporaryFile("w", suffix=".xml", delete=False); tf.write(OBJECT_POOL); tf.close()
                paths.append(tf.name)
            else:
                paths.append(os.path.join(REF, r))
        ms.capi.load().mjh_load_set_parent_child_exclude(PC_EXCLUDE.get(name, 0))
        m = ms.load_mjcf(paths=paths)
        ms.capi.load().mjh_load_set_mesh_mode(1); ms.capi.load().mjh_load_set_parent_child_exclude(0)
        assert (m.c.nmesh > 0) == mesh
        steps, keep = (MESH_STEPS, MESH_KEEP) if mesh else (STEPS, KEEP)
        if name in ("tiago_mesh", "hsrb4s_mesh", "armar6_mesh"):
            steps, keep = 20, (1, 5, 10, 20)
        m.c.maxcon = cap; m.c.maxefc = 6 * cap + m.neq + 2 * m.njnt + m.nv
        d = orc.OrcData(m.ptr)
        ctrl = np.zeros(m.nv, dtype=np.int32)
        jt = m.array("jnt_type"); da = m.array("jnt_dofadr")
        for j in range(m.njnt):
            if jt[j] in (2, 3):
                ctrl[da[j]] = 1
        d.ifield("controlled")[:] = ctrl
        qvel0 = np.array(QVEL0.get(name, np.zeros(m.nv)), dtype=np.float64)
        d.f("qvel")[:] = qvel0
        if name.startswith("c5"):
            steps, keep = STEPS, KEEP
        out = {"qvel0": qvel0}
        maxcon_seen = 0
        for k in range(1, steps + 1):
            d.f("ddq")[:] = command(m, k)
            d.step(1, 1)
            assert d.i("warn") == 0 and d.i("ncon") < cap, (name, k, d.i("warn"), d.i("ncon"))   # no capacity overflow, no reset
            maxcon_seen = max(maxcon_seen, d.i("ncon"))
            if k in keep:
                out[f"qpos_{k}"] = d.f("qpos").copy(); out[f"qvel_{k}"] = d.f("qvel").copy()
                out[f"qfrc_inverse_{k}"] = d.f("qfrc_inverse").copy()
                out[f"nefc_{k}"] = np.int64(d.i("nefc"))
        save_model_tables(m, os.path.join(HERE, f"robot_{name}.npz"), controlled=ctrl, keep=np.array(keep), **out)
        print(name, "nq", m.nq, "nv", m.nv, "neq", m.neq, "nefc", d.i("nefc"), "|qvel|max %.3f" % np.abs(d.f("qvel")).max(),
              "max ncon", maxcon_seen, "npair", m.c.npair, "mesh vertices", m.c.nmeshvert, "file %.1f KB" % (os.path.getsize(os.path.join(HERE, f"robot_{name}.npz")) / 1e3))


if __name__ == "__main__":
    main()

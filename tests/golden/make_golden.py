"""Regenerates tests/golden/*.npz from the repo's own fp64 oracle (oracle/mjh_oracle.c).

The reference holds no golden vectors for this path and its arithmetic library is absent
(SURVEY.md §8-c C1/C3), so these fixtures pin the ORACLE's behaviour (regression anchor for the
restatement and a portable comparison target for the HIP path), not the reference's.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import mujoco_sim_amd as ms  # noqa: E402
import orc  # noqa: E402
from helpers import oracle_s24  # noqa: E402


def snap(d):
    return dict(qpos=d.f("qpos").copy(), qvel=d.f("qvel").copy(), qacc=d.f("qacc").copy(), time=d.f("time")[0],
                ncon=d.i("ncon"), nefc=d.i("nefc"))


def s24():
    m = ms.scene("s24")
    nenv = 6
    tab = m.s24_randomize(0, nenv)
    marks = [1, 10, 60, 150]
    out = {f"tab_{k}": v for k, v in tab.items()}
    out["marks"] = np.array(marks)
    hist_ncon = np.zeros((marks[-1], nenv), dtype=np.int32); hist_nefc = np.zeros((marks[-1], nenv), dtype=np.int32)
    for i in range(nenv):
        d = oracle_s24(m, tab, i)
        for step in range(1, marks[-1] + 1):
            d.step(1)
            hist_ncon[step - 1, i] = d.i("ncon"); hist_nefc[step - 1, i] = d.i("nefc")     # the contact-set history, step by step
            if step in marks:
                s = snap(d)
                for k, v in s.items():
                    out[f"env{i}_step{step}_{k}"] = np.asarray(v)
    out["hist_ncon"] = hist_ncon; out["hist_nefc"] = hist_nefc
    np.savez_compressed(os.path.join(HERE, "s24_golden.npz"), **out)


def pendulum():
    m = ms.scene("pendulum")
    d = orc.OrcData(m.ptr)
    d.f("qvel")[:] = [0.3, 0, 0, 0, 0.3, 0, 0, 0, 0.3]
    out = {}
    done = 0
    for mk in [1, 100, 400]:
        d.step(mk - done); done = mk
        for k, v in snap(d).items():
            out[f"step{mk}_{k}"] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "pendulum_golden.npz"), **out)


def arm7():
    """C3: computed-torque PD (Kp 200, Kd 50: model/ontology/box/box.yaml:8, integral term dropped), gravcomp on"""
    m = ms.scene("arm7", 1)
    d = orc.OrcData(m.ptr)
    q0 = np.array([0.0, 0.3, 0.0, -1.5, 0.0, 1.2, 0.0])
    target = np.array([0.8, -0.4, 0.5, -2.0, 0.6, 2.0, -0.7])
    d.set_qpos(q0); d.call("reset")
    d.ifield("controlled")[:] = 1
    out = dict(q0=q0, target=target)
    for s in range(1, 301):
        d.f("ddq")[:] = 200.0 * (target - d.f("qpos")) - 50.0 * d.f("qvel")   # MjHWInterface::write, effort interface
        d.step(1, 1)
        if s in (1, 50, 300):
            for k, v in snap(d).items():
                out[f"step{s}_{k}"] = np.asarray(v)
            out[f"step{s}_qfrc_inverse"] = d.f("qfrc_inverse").copy()
    np.savez_compressed(os.path.join(HERE, "arm7_golden.npz"), **out)


if __name__ == "__main__":
    s24(); pendulum(); arm7()
    print("golden fixtures written to", HERE)

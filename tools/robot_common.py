"""Shared by the robot bench tools: the commanded joint accelerations used by the robot fixtures (same generator as
tests/test_robot_fixtures.py::robot_command; duplicated so that nothing under tools/ imports the test harness)."""
import numpy as np


def robot_command(m, k):
    jt = m.array("jnt_type"); da = m.array("jnt_dofadr")
    ddq = np.zeros(m.nv)
    for j in range(m.njnt):
        if jt[j] in (2, 3):
            ddq[da[j]] = 0.8 * np.sin(0.05 * k + 0.37 * j)
    return ddq

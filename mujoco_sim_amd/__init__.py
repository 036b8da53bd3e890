"""mujoco_sim_amd — MI355X-native many-environment rigid-body stepper behind the
mujoco_sim step loop (C ABI in include/mjhip.h, HIP kernels in csrc/)."""
from .engine import Engine, Group, Model, boxes_randomize, load_mjcf, scene  # noqa: F401

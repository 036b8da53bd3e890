"""Build the C-ABI shared library (HIP kernels + host C++) in-tree for gfx950."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmjhip.so")
SOURCES = ["engine.hip", "model_builder.cpp", "scenes.cpp", "host_sim.cpp", "mjcf_loader.cpp"]
DEPS = SOURCES + ["step_kernel.h", "dev_math.h", "dev_collide.h", "dev_convex.h", "dev_types.h", "hmath.h", "host_sim.h",
                  os.path.join("..", "..", "include", "mjhip.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(os.path.join(CSRC, d)) and os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950: cross-compiles without a GPU."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics", "-fno-slp-vectorize",
           "-Wno-unused-result"] + [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


def build_oracle():
    """Test infrastructure: the fp64 C restatement under oracle/ (gcc)."""
    root = os.path.dirname(HERE)
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle")])
    return os.path.join(root, "oracle", "liboracle.so")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_oracle())

"""Build the C-ABI shared library (HIP kernels + host C++) in-tree for gfx950.

One object per source under mujoco_sim_amd/build/ (re-made only when the source or one of its headers is newer), then one link:
the kernel translation unit (engine.hip, every mjh_step_kernel instance) takes about two minutes, the host files seconds."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libmjhip.so")
API = os.path.join("..", "..", "include", "mjhip.h")
KERNEL_HEADERS = ["step_kernel.h", "patch_pgs.h", "window_pgs.h", "dev_math.h", "dev_collide.h", "dev_convex.h", "dev_types.h"]
DENSE_HEADERS = ["dense_pgs.h"] + KERNEL_HEADERS
WINDOW_HEADERS = ["window_kernel.h", "step_kernel.h", "patch_pgs.h", "window_pgs.h", "dev_math.h", "dev_collide.h", "dev_convex.h", "dev_types.h"]
# source -> headers it includes (besides itself)
SOURCES = {
    "engine.hip": KERNEL_HEADERS + [API],
    "window.hip": WINDOW_HEADERS + [API],
    "dense.hip": DENSE_HEADERS + [API],
    "group.hip": ["host_pool.h", API],
    "model_builder.cpp": ["hmath.h", API],
    "scenes.cpp": ["hmath.h", API],
    "host_sim.cpp": ["host_sim.h", API],
    "mjcf_loader.cpp": ["hmath.h", API],
}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-slp-vectorize", "-Wno-unused-result"]
# per-source flags.  window.hip: its window forms are written as fully unrolled loops over register-resident windows; past clang's default
# budget for `#pragma unroll` (16 k instructions) a window loop stays a loop and its operand arrays land in scratch memory
SOURCE_FLAGS = {"window.hip": ["-mllvm", "-pragma-unroll-threshold=200000"]}
FLAGS += os.environ.get("MJH_EXTRA_FLAGS", "").split()      # A/B builds (e.g. -DPP_NRC=0 -DMJH_STEP_WAVES=3), together with MJH_BUILD_DIR / MJHIP_LIB
if os.environ.get("MJH_BUILD_DIR"):
    OBJ = os.environ["MJH_BUILD_DIR"]; LIB = os.path.join(OBJ, "libmjhip.so")


def _digest(paths, extra=""):
    """content hash of a source and its headers: staleness does not depend on file times (a snapshot copied to the GPU box
    keeps contents, not necessarily mtimes)"""
    import hashlib
    h = hashlib.sha256(extra.encode())
    for p in paths:
        if os.path.exists(p):
            with open(p, "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950: cross-compiles without a GPU."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    objs, digests = [], []
    for src, deps in SOURCES.items():
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        op = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        sflags = FLAGS + SOURCE_FLAGS.get(src, [])
        dg = _digest([sp] + [os.path.join(CSRC, d) for d in deps], " ".join(sflags))
        stamp = op + ".sha256"
        have = open(stamp).read().strip() if os.path.exists(stamp) else ""
        if force or not os.path.exists(op) or have != dg:
            cmd = [hipcc] + sflags + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
            with open(stamp, "w") as f:
                f.write(dg)
        objs.append(op); digests.append(dg)
    link_dg = _digest([], " ".join(digests))
    link_stamp = os.path.join(OBJ, "libmjhip.sha256")
    have = open(link_stamp).read().strip() if os.path.exists(link_stamp) else ""
    if force or not os.path.exists(LIB) or have != link_dg:
        # RCCL (mjh_group_*: the all-gather of the published state slice across the GPUs of a node) is resolved at run time
        # with dlopen, so the library loads on boxes without it
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", LIB]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        with open(link_stamp, "w") as f:
            f.write(link_dg)
    return LIB


def build_oracle():
    """Test infrastructure: the fp64 C restatement under oracle/ (gcc)."""
    root = os.path.dirname(HERE)
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle")])
    return os.path.join(root, "oracle", "liboracle.so")


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_oracle())

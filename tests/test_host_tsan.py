"""Race detection for the host-side model code (the reference's loader runs under one mutex; this library keeps the
loader's state per thread instead): csrc/{model_builder,mjcf_loader,scenes}.cpp built with -fsanitize=thread and driven by six
threads that compile models concurrently (tests/tsan/host_threads.cpp).  No HIP, no GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_model_code_is_race_free_under_thread_sanitizer(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    probe = tmp_path / "probe.cpp"
    probe.write_text("int main() { return 0; }\n")
    if subprocess.run([gxx, "-fsanitize=thread", str(probe), "-o", str(tmp_path / "probe")], capture_output=True).returncode != 0:
        pytest.skip("ThreadSanitizer runtime not available")
    csrc = os.path.join(ROOT, "mujoco_sim_amd", "csrc")
    exe = tmp_path / "host_threads"
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I", os.path.join(ROOT, "include"),
           os.path.join(csrc, "model_builder.cpp"), os.path.join(csrc, "mjcf_loader.cpp"), os.path.join(csrc, "scenes.cpp"),
           os.path.join(ROOT, "tests", "tsan", "host_threads.cpp"), "-o", str(exe), "-lpthread"]
    subprocess.check_call(cmd)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66"))
    assert "ThreadSanitizer" not in r.stderr, r.stderr[:4000]
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[:2000])
    assert "0 failures" in r.stdout


def test_group_host_thread_pool_is_race_free_under_thread_sanitizer(tmp_path):
    """csrc/host_pool.h — the group host's persistent thread per device (mjh_group_*: every call posts one job per device and waits) —
    under ThreadSanitizer: 20 000 fork-join rounds on 8 workers, error propagation, pool life cycle (tests/tsan/pool_threads.cpp)"""
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    probe = tmp_path / "probe.cpp"
    probe.write_text("int main() { return 0; }\n")
    if subprocess.run([gxx, "-fsanitize=thread", str(probe), "-o", str(tmp_path / "probe")], capture_output=True).returncode != 0:
        pytest.skip("ThreadSanitizer runtime not available")
    exe = tmp_path / "pool_threads"
    subprocess.check_call([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=thread", os.path.join(ROOT, "tests", "tsan", "pool_threads.cpp"), "-o", str(exe), "-lpthread"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66"))
    assert "ThreadSanitizer" not in r.stderr, r.stderr[:4000]
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[:2000])
    assert "0 failures" in r.stdout

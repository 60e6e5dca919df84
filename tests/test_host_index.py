"""CPU: compile and run the C++ check of the engine's host-side node indices (no GPU needed)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_indices_match_kd_semantics(oa, tmp_path):
    exe = tmp_path / "host_index_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off",
                           os.path.join(ROOT, "tests", "cpp", "host_index_check.cpp"),
                           os.path.join(ROOT, "oracle", "_build", "okd.o"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok")

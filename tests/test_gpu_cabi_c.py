"""GPU: a plain C program against the C ABI (no Python, no torch in the data path)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_consumer(tmp_path):
    import trg_planner
    lib_dir = os.path.dirname(trg_planner.LIB_PATH)
    exe = tmp_path / "cabi_smoke"
    subprocess.check_call(["gcc", "-O2", os.path.join(ROOT, "tests", "cpp", "cabi_smoke.c"),
                           "-L" + lib_dir, "-ltrg_engine", "-lm", "-Wl,-rpath," + lib_dir,
                           "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    last = out.stdout.strip().splitlines()[-1]  # (RCCL prints a version banner when its communicator is made)
    assert last.startswith("ok arch=gfx950") and "device_bfs=1" in last, out.stdout

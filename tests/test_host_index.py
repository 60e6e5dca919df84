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


def test_map_order_replica_matches_libstdcxx(tmp_path):
    """cleanGraph's renumbering follows std::unordered_map iteration order; the engine's O(n)
    replica of that order (map_order_sim.h) must match the real container, bucket counts included."""
    exe = tmp_path / "map_order_check"
    subprocess.check_call(["g++", "-O2", "-std=c++17",
                           os.path.join(ROOT, "tests", "cpp", "map_order_check.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


def test_host_checks_under_sanitizers(oa, tmp_path):
    """The host-side structures of the engine that can run without a GPU (NodeKd / NodeGrid /
    tie-break, container-order replica) under AddressSanitizer + UBSan."""
    flags = ["-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined",
             "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
    okd_o = tmp_path / "okd_asan.o"
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-c", os.path.join(ROOT, "oracle", "okd.c"), "-o", str(okd_o)])
    for src, extra in (("host_index_check.cpp", [str(okd_o)]), ("map_order_check.cpp", [])):
        exe = tmp_path / (src + ".asan")
        subprocess.check_call(["g++"] + flags + ["-I", os.path.join(ROOT, "oracle"),
                                                 os.path.join(ROOT, "tests", "cpp", src)] + extra +
                              ["-o", str(exe)])
        out = subprocess.run([str(exe)], capture_output=True, text=True)
        assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr

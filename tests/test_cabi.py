"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol the header declares, and
fails loudly (no CPU fallback) when no GPU is present."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_and_exports():
    import trg_planner
    from trg_planner._engine import EXPORTS
    trg_planner.build_library()
    lib = trg_planner.load_library()
    header = open(os.path.join(ROOT, "include", "trg_engine.h")).read()
    declared = set(re.findall(r"\b(trg_engine_[a-z_0-9]+)\s*\(", header))
    assert declared == set(EXPORTS), declared ^ set(EXPORTS)
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_code_object_is_gfx950_only(tmp_path):
    import subprocess
    import trg_planner
    # llvm-objdump --offloading drops the extracted code objects into its working directory
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", trg_planner.LIB_PATH],
                         capture_output=True, text=True, cwd=str(tmp_path)).stdout
    archs = set(re.findall(r"gfx[0-9a-f]+", out))
    assert archs == {"gfx950"}, archs


def test_no_cpu_fallback_without_gpu():
    import torch
    import trg_planner
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(trg_planner.TrgError) as ei:
        trg_planner.Engine()
    assert "TRG_ERR_DEVICE" in str(ei.value)


def test_product_does_not_touch_the_oracle():
    """Nothing under trg-planner_amd/ may import, link or mention the oracle."""
    pkg = os.path.join(ROOT, "trg-planner_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".inc", ".ipp", ".h", ".hpp", ".sh")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_api" not in txt and "libtrg_oracle" not in txt, os.path.join(dp, f)
                assert "/root/reference" not in txt.replace("/root/reference/", "REF/") or f.endswith(
                    (".hip", ".inc", ".ipp", ".py", ".cpp", ".h", ".hpp")), f

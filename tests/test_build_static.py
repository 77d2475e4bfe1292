"""Static checks of the built GEMM code (CPU only; hipcc cross-compiles gfx950 without a GPU)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="needs hipcc")
def test_pp256_operand_loads_keep_their_registers():
    """The pp256 epilogue loads its fused operands with inline asm and hand-counted waits; a register copy between such a
    load and its wait would read stale data (it happened once: two sibling request sites, NaNs in the gated-residual
    epilogue).  scripts/check_pp_asm.py compiles gemm_pp.hip to assembly and asserts that every request site of a kernel's
    main loop writes the same physical registers and that no kernel spills VGPRs (scratch traffic would also break the
    counted vmcnt waits)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_pp_asm.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("-> ok") >= 3, r.stdout


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="needs hipcc")
def test_w4_compiler_stays_out_of_the_k_loops_registers():
    """The w4 GEMM's k-loop is generated inline asm on LITERAL registers (a[0:255], v[96:255]); v[144:255] hold data across an
    epilogue.  hipcc once overran a budget of 96 registers instead of spilling (v96 / v97 handed out as temporaries: the first staging
    register corrupted in every tile but a workgroup's first).  scripts/check_w4_asm.py compiles gemm_w4.hip to assembly and asserts
    that no compiler-generated instruction of any w4 kernel names v144+ or an accumulator register, and that nothing is spilled to
    scratch."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_w4_asm.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "10 w4 kernels, 0 problem lines" in r.stdout, r.stdout

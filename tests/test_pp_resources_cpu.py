"""The persistent GEMM (gemm_pp.hip) counts its vector-memory operations by hand (s_waitcnt vmcnt(N) against the DMA ring):
a register spill would add scratch loads / stores the counts do not know about (and a drain of the ring in front of each
reload).  Compile every instantiation for gfx950 and require: no VGPR spill, no scratch, 128 KiB of LDS, two waves per SIMD."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_pp256_kernels_do_not_spill(tmp_path):
    src = os.path.join(ROOT, "micro_diffusion_amd", "csrc", "gemm_pp.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-I", os.path.join(ROOT, "include"),
                        "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", str(tmp_path / "pp.o")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
    kernels = [b for b in blocks if "gemm_bf16_pp_kernel" in b.split("\n")[0]]
    assert len(kernels) == 10, ("expected the 10 pp256 instantiations (NT: bf16, residual, dact, dact from the cached derivative; NN: bf16, gelu, "
                                f"gelu + cached derivative, residual, f32; TN: f32), found {len(kernels)}")
    for b in kernels:
        name = b.split(" ")[0]

        def field(label):
            m = re.search(label + r": (\d+)", b)
            assert m, (label, name)
            return int(m.group(1))
        assert field(r"VGPRs Spill") == 0, f"{name} spills VGPRs"
        assert field(r"ScratchSize \[bytes/lane\]") == 0, f"{name} uses scratch"
        assert field(r"LDS Size \[bytes/block\]") == 131072, name
        assert field(r"Occupancy \[waves/SIMD\]") == 2, name

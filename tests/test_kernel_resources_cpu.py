"""Compiled-resource guard for the bandwidth / latency-bound kernels (norm, elementwise, routing, edm, optim), CPU only.

Two regressions this catches were real in this tree: (a) local arrays that the compiler leaves in private memory (the fused
attention backward staged its rows global -> VGPR -> scratch -> VGPR -> LDS until its `uint4` arrays became ext-vectors;
tests/test_attn_static_cpu.py guards that file); (b) a few registers too many for the next occupancy step (the LayerNorm
backward sat at 134 VGPRs = 3 waves / SIMD until it stopped keeping an fp32 copy of dL/dxhat: 128 = 4 waves / SIMD).
hipcc cross-compiles gfx950 without a GPU; -Rpass-analysis=kernel-resource-usage prints what the code object will ask for.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
FILES = ["norm.hip", "elementwise.hip", "routing.hip", "edm.hip", "optim.hip"]

# kernels that may use a few bytes of scratch: the GENERIC LayerNorm backward (activation-before-norm layers: 24 launches per
# step) indexes a small per-lane array dynamically
SCRATCH_OK = ("ln_bwd_kernelILi1ELb1", "ln_bwd_kernelILi2ELb1", "ln_bwd_kernelILi4ELb1")

# (substring of the mangled name, minimum waves / SIMD): the hot instantiations of the XL/2 step
OCCUPANCY = [
    ("ln_bwd_kernelILi2ELb0", 4),          # C = 1024 rows, templated hot path
    ("ln_fwd_kernelILi2ELb0", 6),
    ("qkln_fwd_kernelILi2E", 8),
    ("qkln_bwd_kernelILi2E", 7),
    ("moe_combine_kernelILi8E", 4),        # 8 experts: every selected expert row in flight at once
    ("moe_scatter_sum_kernelILi8E", 6),
    ("swiglu_bwd_kernel", 8),
    ("gate_bwd_kernelILi2E", 8),
    ("adamw_kernelILb0ELi0E", 8),
    ("sumsq_kernelILb1E", 8),
]


def _resources(src, tmp_path):
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result",
                        "-I", os.path.join(ROOT, "include"), "-Rpass-analysis=kernel-resource-usage", "-c",
                        os.path.join(ROOT, "micro_diffusion_amd", "csrc", src), "-o", str(tmp_path / (src + ".o"))],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = {}
    for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
        name = b.split()[0]

        def field(label):
            m = re.search(label + r": (\d+)", b)
            assert m, (label, name)
            return int(m.group(1))
        out[name] = dict(vgprs=field(r"VGPRs"), spill=field(r"VGPRs Spill"), scratch=field(r"ScratchSize \[bytes/lane\]"),
                         occ=field(r"Occupancy \[waves/SIMD\]"))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_no_spills_no_scratch_and_hot_kernel_occupancy(tmp_path):
    res = {}
    for f in FILES:
        res.update(_resources(f, tmp_path))
    assert len(res) >= 50, f"expected the kernels of {FILES}, parsed {len(res)}"
    for name, v in res.items():
        assert v["spill"] == 0, f"{name} spills {v['spill']} VGPRs"
        if not any(tag in name for tag in SCRATCH_OK):
            assert v["scratch"] == 0, f"{name} uses {v['scratch']} bytes of scratch per lane (a local array left in private memory?)"
        else:
            assert v["scratch"] <= 64, (name, v)
    for tag, occ in OCCUPANCY:
        hits = [k for k in res if tag in k]
        assert hits, f"no kernel matching {tag}"
        for k in hits:
            assert res[k]["occ"] >= occ, f"{k}: {res[k]['vgprs']} VGPRs -> {res[k]['occ']} waves / SIMD, expected >= {occ}"

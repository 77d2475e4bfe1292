"""Static check of the built pp256 kernels (run on the build machine, no GPU): the epilogue operand loads are inline asm with
hand-counted waits, so the compiler must not copy their destination registers between the load and the wait.  The
guarantee used (gemm_pp_common.h: asm_load16) is that every request site in a kernel's MAIN LOOP writes the same physical
registers; this script compiles gemm_pp.hip to assembly and asserts exactly that, and that no kernel spills VGPRs.
Usage: python scripts/check_pp_asm.py   (exit code 0 = ok)
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "micro_diffusion_amd", "csrc", "gemm_pp.hip")
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "gemm_pp.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
                    "-Wno-unused-command-line-argument", "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", src, "-o", out], check=True)
    lines = open(out).read().split("\n")

kern, inasm, inloop = None, False, False
sites = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> in-loop? -> [dest]
spills = {}
for l in lines:
    m = re.match(r"^(_ZN\S+):", l)
    if m:
        kern, inloop = m.group(1), False
    if kern and "s_endpgm" in l:
        kern = None
    if kern is None:
        m = re.match(r"\s+\.vgpr_spill_count:\s+(\d+)", l)
        if m:
            spills[len(spills)] = int(m.group(1))
        continue
    if "in Loop:" in l or "Loop Header" in l:
        inloop = True
    elif re.match(r"^\.LBB\d+_\d+:\s*$", l):      # a block label without a loop annotation: outside the main loop
        inloop = False
    if "ASMSTART" in l:
        inasm = True
    elif "ASMEND" in l:
        inasm = False
    elif inasm and "global_load_dwordx4" in l:
        sites[kern][inloop].append(l.split()[1].rstrip(","))
bad = 0
for k, v in sites.items():
    loop = collections.Counter(v[True])
    n_sites = max(loop.values()) if loop else 0
    ok = all(c == n_sites for c in loop.values()) and len(loop) in (4, 5)
    print(f"{k[-38:]}: main-loop operand registers {dict(loop)} -> {'ok' if ok else 'MISMATCH'}")
    bad += not ok
if any(spills.values()):
    print("VGPR spills:", spills)
    bad += 1
sys.exit(1 if bad else 0)

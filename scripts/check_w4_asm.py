"""Static audit of the built w4 GEMM (tests/test_build_static.py runs it): outside the generated inline asm hipcc must not touch
v[144:255] (the fragments that are live across an epilogue and the tail of the k-step-1 set; v[96:143] are dead there and may be used) or any
accumulator register (they belong to the k-loop: scripts/gen_w4_acc.py), and nothing may be spilled to scratch.
    python scripts/check_w4_asm.py [<gemm_w4 .s file>]     (without an argument: compiles gemm_w4.hip to assembly first; exit code 0 = ok)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def audit(text):
    problems = []
    names = re.findall(r'^(_ZN\S*gemm_bf16_w4_kernel\S*):', text, re.M)
    for name in names:
        body = text[text.index(name + ':'):]
        body = body[:body.index('s_endpgm')]
        inasm = False
        for l in body.split('\n'):
            if '#ASMSTART' in l:
                inasm = True
                continue
            if '#ASMEND' in l:
                inasm = False
                continue
            if inasm:
                continue
            code = l.split(';')[0]
            regs = [int(m) for m in re.findall(r'\bv(\d+)\b', code)]
            for m in re.finditer(r'v\[(\d+):(\d+)\]', code):
                regs += [int(m.group(1)), int(m.group(2))]
            if any(r >= 144 for r in regs) or re.search(r'\ba\d+\b|\ba\[', code) or 'scratch_' in code:
                problems.append((name, l.strip()))
    return names, problems


if __name__ == "__main__":
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "gemm_w4.s")
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-inline-asm",
                            "-Wno-unused-command-line-argument", "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                            os.path.join(ROOT, "micro_diffusion_amd", "csrc", "gemm_w4.hip"), "-o", out], check=True)
            text = open(out).read()
    names, problems = audit(text)
    print(f"{len(names)} w4 kernels, {len(problems)} problem lines")
    for n, l in problems[:20]:
        print(n[-30:], l)
    sys.exit(1 if problems or not names else 0)

"""Generates micro_diffusion_amd/csrc/gemm_w4_acc.inc: the k-loop of the 4-wave GEMM (gemm_w4.hip) as inline asm on LITERAL registers.

Operands are staged by LDS-DMA: `buffer_load_dwordx4 <offset>, <descriptor>, <k offset> offen lds` moves 64 lanes x 16 bytes from global
memory straight to the 1 KiB of LDS at M0 -- no staging registers, no ds_write.  (The round's first form staged through 64 registers:
buffer_load -> ds_write_b128.  Its ablations priced the LDS writes at 6 % and the loads at 10-13 % of the kernel; the library's own NT kernel,
disassembled from its code object, runs 16 such DMA loads, 32 ds_read_b128 and 128 v_mfma_f32_16x16x32_bf16 per 64-deep k-tile with all
16 fragments of a k-step resident and the LDS buffer released early in the k-tile.  This form: +4 % on hot operands, +0.7 % / +1.0 % on the
full step at microbatch 1024 / 256 in same-box A/Bs, profiles/r6_w4_lds_dma.txt.)

Register plan (one wave per SIMD, the whole 512-register file):
    a[0:255]    64 accumulator blocks of 16 x 16 fp32: block (i, j) = a[4 (8 i + j) : +3] (row fragment i, column fragment j of the wave's
                128 x 128 tile; v_mfma_f32_16x16x32_bf16 with swapped operands, D = B A^T: lane l holds row l % 16, columns 4 (l / 16) + r)
    v[96:127]   A fragments of k-step 1: FA[1][i] = v[96 + 4 i : +3]      } dead while an epilogue runs (re-read before their next use):
    v[128:159]  B fragments of k-step 1: FB[1][j] = v[128 + 4 j : +3]     } hipcc may spread into v[96:143] there (amdgpu_num_vgpr(144))
    v[160:191]  A fragments of k-step 0: FA[0][i]                         } live across an epilogue: the next tile's first k-step is read
    v[192:223]  B fragments of k-step 0: FB[0][j]                         } during the finished tile's last k-tile
    v[224:255]  unused
    v[0:95]     hipcc's across the k-loop (addresses, loop state): every k-loop statement clobbers v[96:255] and m0, so nothing of the
                compiler's lives there across one; scripts/check_w4_asm.py audits that no compiler instruction names v144+ or an accumulator.
Why asm: handed the same stream as C++, hipcc gave every re-load of a fragment register a fresh physical register, ran out of registers and
spilled behind a vmcnt(0) in the k-loop; compiler-owned accumulators that fill the accumulator file exactly were spilled at every loop header.

LDS image: rows of 128 bytes, 16-byte chunk index ^ ((row >> 1) & 7) (K-strided B: [64 k][32 chunks], chunk ^ swz(k)).  A DMA instruction
writes LANE-LINEARLY, so the swizzle is applied on the GLOBAL side: the lane that sits at physical chunk p of row r fetches logical chunk
p ^ swz(r) (W4Addr::aofs / bofs, gemm_w4.hip); the fragment reads see the image they always saw.  Piece x (4 KiB = 32 rows, or 8 k-rows of a
K-strided B) of an operand: wave w writes its 1 KiB at M0 = ad.ldsw + buffer + 4096 x.

One k-tile t (64 deep, buffer BUF = t % 2) = the 64 MFMAs of k-step 0, then the 64 of k-step 1 (for i: for j), cut at CUT1 = 16 and CUT2 = 112:
    [0, CUT1)      fillers: the 16 fragment reads of k-step 1 from BUF.                        s_waitcnt lgkmcnt(0); s_barrier
                   -> every wave holds all of k-tile t in registers: BUF is free
    [CUT1, CUT2)   fillers: the 16 DMA loads of k-tile t + 2 into BUF, the M0 write and its load in different MFMA gaps (the MFMA between
                   them is the wait state the M0 write needs).                                 s_waitcnt vmcnt(16); s_barrier
                   -> the 16 loads of k-tile t + 1, issued one k-tile ago into BUF ^ 1, have landed in every wave (loads return in order)
    [CUT2, 128)    fillers: the fragment reads of k-tile t + 1's k-step 0 from BUF ^ 1 (the k-step-0 registers are free since MFMA 64).
The next k-tile starts with s_waitcnt lgkmcnt(0).  FRESH (first k-tile of an output tile): k-step 0 takes C = 0 and the middle part waits
for no load -- the epilogue in front of it has waited for every load (its own stores never stand between a load and its wait).
gemm_w4.hip calls the three parts w4_h0 / w4_h1 / w4_h2 (w4_h3 is empty: the register-staged form had four).

python scripts/gen_w4_acc.py [out.inc] [c1=<n>] [c2=<n>] rewrites the file (scripts/build_w4_variant.sh builds A/B libraries)."""
import os
import sys

FLAGS = [a for a in sys.argv[2:]]
OUT = sys.argv[1] if len(sys.argv) > 1 else None
V0 = 96
FA_BASE = {1: 96, 0: 160}
FB_BASE = {1: 128, 0: 192}
FA = lambda s, i: f"v[{FA_BASE[s] + 4 * i}:{FA_BASE[s] + 4 * i + 3}]"
FB = lambda s, j: f"v[{FB_BASE[s] + 4 * j}:{FB_BASE[s] + 4 * j + 3}]"
ACC = lambda i, j: f"a[{4 * (8 * i + j)}:{4 * (8 * i + j) + 3}]"
BUF = 32768          # bytes of one operand of one k-tile buffer
BREG = 65536         # B buffers start here
FRAG = 2048          # bytes of 16 rows of a buffer
CLOB = ", ".join([f'"a{r}"' for r in range(256)] + [f'"v{r}"' for r in range(V0, 256)] + ['"m0"', '"memory"'])


def mfmas(ks, fresh):
    """the 64 MFMAs of k-step ks: for i: for j"""
    out = []
    for i in range(8):
        for j in range(8):
            acc = ACC(i, j)
            out.append(f"v_mfma_f32_16x16x32_bf16 {acc}, {FB(ks, j)}, {FA(ks, i)}, {'0' if fresh else acc}")
    return out


class Ops:
    def __init__(self):
        self.names, self.cons = [], []

    def __call__(self, expr, con="v"):
        if expr not in self.names:
            self.names.append(expr)
            self.cons.append(con)
        return f"%{self.names.index(expr)}"

    def text(self):
        assert len(self.names) <= 30, len(self.names)
        return ", ".join(f'"{c}"({n})' for c, n in zip(self.cons, self.names))


def reads(o, akc, bkc, ks, buf):
    """all fragment reads of k-step ks from buffer buf: A 0 first (the first MFMA row needs A 0 and every B), then B 0..7, then A 1..7.
    K-contiguous operand: one ds_read_b128 per 16-row fragment, the k-step in the address register (XOR swizzle).  K-strided operand (image
    [64 k][32 chunks], chunk ^ swz(k)): one address register per fragment, the k-step (32 k-rows = 16 KiB) in the offset; a fragment is two
    transposing 8-byte reads (k + 0..3, k + 4..7: 4 k-rows = 2 KiB apart)."""
    def frag(kc, base_reg, which, adname, n, region):
        if kc:
            return f"ds_read_b128 {which(ks, n)}, {o(f'ad.{adname}[{ks}]')} offset:{buf * BUF + n * FRAG}"
        lo = base_reg[ks] + 4 * n
        ad = o(f"ad.{adname}[{n}]")
        return [f"ds_read_b64_tr_b16 v[{lo}:{lo + 1}], {ad} offset:{buf * BUF + ks * 16384}",
                f"ds_read_b64_tr_b16 v[{lo + 2}:{lo + 3}], {ad} offset:{buf * BUF + ks * 16384 + 2048}"]
    ra = [frag(akc, FA_BASE, FA, "adA", i, 0) for i in range(8)]
    rb = [frag(bkc, FB_BASE, FB, "adB", j, 1) for j in range(8)]
    return [ra[0]] + rb + ra[1:]


def dma(o, x, buf, ka="koffA", kb="koffB"):
    """piece x (0..7: A, 8..15: B) of the cursor's k-tile -> buffer buf: 1 KiB per wave at M0 = this wave's 1 KiB of the piece's 4 KiB"""
    if x < 8:
        ofs, rsrc, koff, base = o(f"ad.aofs[{x}]"), o("rA", "s"), o(ka, "s"), buf * BUF + x * 4096
    else:
        ofs, rsrc, koff, base = o(f"ad.bofs[{x - 8}]"), o("rB", "s"), o(kb, "s"), BREG + buf * BUF + (x - 8) * 4096
    return [f"s_add_u32 m0, {o('ad.ldsw', 's')}, {base}", f"buffer_load_dwordx4 {ofs}, {rsrc}, {koff} offen lds"]


def part(mm, fillers, head, tail, span=None):
    """MFMAs mm with the fillers spread evenly over the gaps behind the first `span` of them"""
    lines = list(head)
    span = span or len(mm)
    n = len(fillers)
    at = {}
    for k, f in enumerate(fillers):
        at.setdefault((k * span) // n if n else 0, []).append(f)
    for g, m in enumerate(mm):
        lines.append(m)
        for f in at.get(g, []):
            lines.extend(f if isinstance(f, list) else [f])
    return lines + list(tail)


def emit(lines):
    flat = []
    for l in lines:
        flat.extend(l if isinstance(l, list) else [l])
    return '"' + "\\n\\t".join(flat) + '"'


def variants(out, conds_texts):
    for n, (cond, txt, ops) in enumerate(conds_texts):
        kw = ("if constexpr (" + cond + ")") if n == 0 else ("else if constexpr (" + cond + ")" if n + 1 < len(conds_texts) else "else")
        out.append(f"    {kw} asm volatile({txt} : : {ops} : {CLOB});")


LAYOUTS = ((1, 1), (1, 0), (0, 0))
SIG = "const W4Addr& ad, const u32x4& rA, const u32x4& rB, int koffA, int koffB"
out = ["// GENERATED by scripts/gen_w4_acc.py -- do not edit (register plan and schedule: the generator's docstring).",
       "// (AKC, BKC) = (1, 1): both operands K-contiguous (nn.Linear forward); (1, 0): B K-strided (dgrads, the [E, in, out] expert weights);",
       "// (0, 0): both K-strided (weight gradients: contraction over the tokens).",
       ""]

# ---- prologue: k-tile 0 -> buffer 0, k-tile 1 -> buffer 1; k-tile 0 landed
out.append("template <int AKC, int BKC>")
out.append(f"__device__ __forceinline__ void w4_prologue({SIG}, int koffA1, int koffB1) {{")
cs = []
for akc, bkc in LAYOUTS[:1]:             # (the DMA loads do not depend on the layout: the offsets carry it)
    o = Ops()
    pro = []
    for x in range(16):
        d = dma(o, x, 0)
        pro += [d[0], "s_nop 0", d[1]]           # (M0 write -> LDS-DMA: one wait state; in the loop an MFMA stands between them)
    for x in range(16):
        d = dma(o, x, 1, "koffA1", "koffB1")
        pro += [d[0], "s_nop 0", d[1]]
    pro += ["s_waitcnt vmcnt(16)"]
    cs.append(("true", emit(pro), o.text()))
variants(out, cs)
out.append("}")
out.append("// the barrier that publishes buffer 0, then the fragment reads of its k-step 0")
out.append("template <int AKC, int BKC>")
out.append("__device__ __forceinline__ void w4_first_reads(const W4Addr& ad) {")
cs = []
for akc, bkc in LAYOUTS:
    o = Ops()
    cs.append((f"AKC == {akc} && BKC == {bkc}", emit(["s_barrier"] + reads(o, akc, bkc, 0, 0)), o.text()))
variants(out, cs)
out.append("}")
out.append("")


def gen(name, flagname, build):
    out.append(f"template <int AKC, int BKC, int BUFI, bool {flagname}>")
    out.append(f"__device__ __forceinline__ void {name}({SIG}) {{")
    cs = []
    for akc, bkc in LAYOUTS:
        for b in range(2):
            for flag in (True, False):
                o = Ops()
                cs.append((f"AKC == {akc} && BKC == {bkc} && BUFI == {b} && {flagname if flag else '!' + flagname}",
                           emit(build(o, akc, bkc, b, flag)), o.text()))
    variants(out, cs)
    out.append("}")


# MFMA stream of one k-tile: 64 of k-step 0 then 64 of k-step 1, cut at CUT1 and CUT2 (flags c1=, c2=):
#   [0, CUT1)      fillers: the 16 fragment reads of k-step 1 from BUF;   then lgkmcnt(0) + barrier -> BUF is free
#   [CUT1, CUT2)   fillers: the 16 DMA loads of the cursor's k-tile into BUF; then vmcnt(16) + barrier -> k-tile t + 1 has landed
#   [CUT2, 128)    fillers: the fragment reads of k-tile t + 1's k-step 0 from BUF ^ 1 (CUT2 >= 64: the sets 0 are free)
CUT1, CUT2 = 16, 112
for f in FLAGS:
    if f.startswith("c1="):
        CUT1 = int(f[3:])
    if f.startswith("c2="):
        CUT2 = int(f[3:])
assert 0 < CUT1 <= 64 <= CUT2 < 128


def stream(fresh):
    return mfmas(0, fresh) + mfmas(1, False)


def p1(o, akc, bkc, b, fresh):
    return part(stream(fresh)[:CUT1], reads(o, akc, bkc, 1, b), ["s_waitcnt lgkmcnt(0)"], ["s_waitcnt lgkmcnt(0)", "s_barrier"])


def p2(o, akc, bkc, b, fresh):
    fill = []
    for x in range(16):
        fill += dma(o, x, b)             # M0 write and its load in different MFMA gaps (the MFMA between them is the wait state)
    return part(stream(fresh)[CUT1:CUT2], fill, [], ([] if fresh else ["s_waitcnt vmcnt(16)"]) + ["s_barrier"])


def p3(o, akc, bkc, b, unused):
    mm = stream(False)[CUT2:]
    return part(mm, reads(o, akc, bkc, 0, b ^ 1), [], [], span=max(len(mm) - 6, 1))


gen("w4_h0", "FRESH", p1)
gen("w4_h1", "FRESH", p2)
gen("w4_h2", "NOWAIT", p3)
out.append("template <int AKC, int BKC, int BUFI>")
out.append("__device__ __forceinline__ void w4_h3(const W4Addr&) {}      // (the LDS-DMA schedule has three parts)")
out.append("")
out.append("// Accumulator blocks (I, 4 JH + q), q = 0..3, as fp32: r[4 q + e] = row 16 I + lane % 16, column 16 (4 JH + q) + 4 (lane / 16) + e of the wave tile.")
out.append("// The caller has put the MFMA -> accumulator-read wait states in front.")
out.append("template <int I, int JH>")
out.append("__device__ __forceinline__ void w4_acc_read16(float (&r)[16]) {")
first = True
for i in range(8):
    for jh in range(2):
        base = 4 * (8 * i + 4 * jh)
        body = "\\n\\t".join(f"v_accvgpr_read_b32 %{k}, a{base + k}" for k in range(16))
        outs = ", ".join(f'"=v"(r[{k}])' for k in range(16))
        out.append(f"    {'if' if first else 'else if'} constexpr (I == {i} && JH == {jh}) asm volatile(\"{body}\" : {outs});")
        first = False
out.append("}")
path = OUT or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "micro_diffusion_amd", "csrc", "gemm_w4_acc.inc")
with open(path, "w") as f:
    f.write("\n".join(out) + "\n")
print("wrote", os.path.normpath(path))

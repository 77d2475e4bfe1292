"""Generates micro_diffusion_amd/csrc/gemm_w4_acc.inc: the hand-scheduled k-loop of the w4 GEMM (gemm_w4.hip) as inline asm.

Register plan of the kernel (one wave per SIMD, the whole 512-register file):
    a[0:255]    64 accumulator blocks of 16 x 16 fp32: block (i, j) = a[4 (8 i + j) : +3] (row fragment i, column fragment j of the wave's
                128 x 128 tile; v_mfma_f32_16x16x32_bf16 with swapped operands, D = B A^T: lane l holds row l % 16, columns 4 (l / 16) + r)
    v[96:127]   A fragments of k-step 1: FA[1][i] = v[96 + 4 i : +3], i = 0..7        } dead while an epilogue runs (re-read in H1 / H0
    v[128:143]  B fragments, slot 1:     FB[1][q] = v[128 + 4 q : +3]                  } before their next use)
    v[144:207]  staging registers ST[x] = v[144 + 4 x : +3]: pieces 0..7 of the A k-tile, 8..15 of the B k-tile (16 bytes per lane each)
    v[208:239]  A fragments of k-step 0: FA[0][i]
    v[240:255]  B fragments, slot 0:     FB[0][q].  A k-step is multiplied in two HALVES of 32 MFMAs (column fragments j = 4 h + q);
                half n of the stream uses slot n % 2 while the other slot is re-read
    v[0:95]     hipcc's across the k-loop (addresses, loop state): every k-loop statement clobbers v[96:255], so nothing of the compiler's
                can live there across one.  Inside an epilogue hipcc may spread into v[96:143] (amdgpu_num_vgpr(144): it was seen to overrun
                a lower budget instead of spilling, into what then were the staging registers); scripts/check_w4_asm.py audits that no
                compiler instruction names v144+ or an accumulator register.  The registers above are named literally and listed as
                clobbers (that also sizes the kernel descriptor).
Why asm: handed the same stream as plain loads / LDS accesses (or as asm with tied "+v" operands), hipcc gave every re-load of a staging
or fragment register a fresh physical register, ran out of registers and spilled staging registers behind a vmcnt(0) in the k-loop; and
compiler-owned accumulators that fill the accumulator file exactly were spilled at every loop header.
Why 16x16x32: the chip is POWER-bound on this kernel (1.42 GHz of 2.4 at 8192^3, profiles/r6_w4_v1_experiments.txt): the same instruction
stream on v_mfma_f32_16x16x32_bf16 ran 7 % faster than on 32x32x16 (half the accumulator-file traffic per flop).

One k-tile (64 deep) = 4 halves of 32 MFMAs, fillers spread evenly over the gaps between MFMAs (16 cycles each):
    H0 (k-step 0, j 0..3): 4 B reads (k-step 0, j 4..7);                pieces 0..5:  [s_waitcnt vmcnt(15)] ds_write_b128 | buffer_load
    H1 (k-step 0, j 4..7): 4 B reads (k-step 1, j 0..3), 8 A reads (k-step 1); pieces 6..9
    H2 (k-step 1, j 0..3): 4 B reads (k-step 1, j 4..7);                pieces 10..15
    s_waitcnt lgkmcnt(0); s_barrier
    H3 (k-step 1, j 4..7): 8 A reads + 4 B reads of the NEXT k-tile's k-step 0 (other buffer)
Waits (issue order = text order): vmcnt(15) in front of every LDS write -- the piece was loaded one k-tile ago and exactly 15 loads are
younger; the first k-tile of an output tile (FRESH) needs none: the epilogue in front of it has waited for every load (so that its own
stores never stand between a load and its wait).  lgkmcnt(W) in front of a half, W = LDS writes issued behind the reads it needs (LDS
operations retire in order).

python scripts/gen_w4_acc.py [out.inc] [experiment flags] rewrites the file."""
import os
import sys

# experiment switches (scripts/build_w4_variant.sh): python scripts/gen_w4_acc.py <out.inc> flag ...
FLAGS = set(sys.argv[2:])
OUT = sys.argv[1] if len(sys.argv) > 1 else None

V0 = 96
# A set 1 and B slot 1 are DEAD while the epilogue runs (re-read in H0 / H1 before their next use): they sit lowest, right above hipcc's
# registers, so that a compiler that overruns its budget in an epilogue (it does: amdgpu_num_vgpr is not a hard limit) lands in them
FA_BASE = {1: 96, 0: 208}
FB_BASE = {1: 128, 0: 240}
ST = lambda x: f"v[{144 + 4 * x}:{147 + 4 * x}]"
FA = lambda s, i: f"v[{FA_BASE[s] + 4 * i}:{FA_BASE[s] + 4 * i + 3}]"
FB = lambda s, q: f"v[{FB_BASE[s] + 4 * q}:{FB_BASE[s] + 4 * q + 3}]"
ACC = lambda i, j: f"a[{4 * (8 * i + j)}:{4 * (8 * i + j) + 3}]"
BUF = 32768          # bytes of one operand of one k-tile buffer
FRAG = 2048          # bytes of 16 rows of a buffer
CLOB = ", ".join([f'"a{r}"' for r in range(256)] + [f'"v{r}"' for r in range(V0, 256)] + ['"memory"'])


def mfmas(h, slot, aset, fresh):
    """the 32 MFMAs of half h (column fragments 4 h + q): for i: for q"""
    out = []
    for i in range(8):
        for q in range(4):
            acc = ACC(i, 4 * h + q)
            out.append(f"v_mfma_f32_16x16x32_bf16 {acc}, {FB(slot, q)}, {FA(aset, i)}, {'0' if fresh else acc}")
    return out


class Ops:
    """operand list of one asm statement: name -> %n, in first-use order"""

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


def reads_a(o, aset, ks, buf):
    """A is always K-contiguous: fragment i = 16 rows, one ds_read_b128; the k-step is in the address register (XOR swizzle)"""
    return [] if "noread" in FLAGS else [f"ds_read_b128 {FA(aset, i)}, {o(f'ad.adA[{ks}]')} offset:{buf * BUF + i * FRAG}" for i in range(8)]


def reads_b(o, bkc, slot, ks, h, buf):
    """B fragments j = 4 h + q of k-step ks.  K-contiguous: as A.  K-strided: image [64 k][32 chunks] (chunk ^ swz(k)), one address register per
    j, the k-step (32 k-rows = 16 KiB) in the offset; a fragment is two transposing 8-byte reads (k + 0..3, k + 4..7: 4 k-rows = 2 KiB apart)"""
    if "noread" in FLAGS:
        return []
    if bkc:
        return [f"ds_read_b128 {FB(slot, q)}, {o(f'ad.adB[{ks}]')} offset:{buf * BUF + (4 * h + q) * FRAG}" for q in range(4)]
    out = []
    for q in range(4):
        lo = FB_BASE[slot] + 4 * q
        ad = o(f"ad.adB[{4 * h + q}]")
        out.append(f"ds_read_b64_tr_b16 v[{lo}:{lo + 1}], {ad} offset:{buf * BUF + ks * 16384}")
        out.append(f"ds_read_b64_tr_b16 v[{lo + 2}:{lo + 3}], {ad} offset:{buf * BUF + ks * 16384 + 2048}")
    return out


def wl(o, bkc, x, buf, nowait):
    """piece x: [vmcnt] LDS write of the staged piece (k-tile t + 1) into buffer buf, then its re-load with k-tile t + 2"""
    out = []
    if x < 8:
        wr, ofs, rsrc, koff = o("ad.wrA"), o(f"ad.aofs[{x}]"), o("rA", "s"), o("koffA", "s")
    else:
        wr = o("ad.wrB[0]") if bkc else o(f"ad.wrB[{x & 1}]")     # K-strided B: the swizzle of a piece's k-rows depends on its parity
        ofs, rsrc, koff = o(f"ad.bofs[{x - 8}]"), o("rB", "s"), o("koffB", "s")
    if "nowrite" not in FLAGS:
        w = f"ds_write_b128 {wr}, {ST(x)} offset:{buf * BUF + (x & 7) * 4096}"
        out.append(w if (nowait or "nowait" in FLAGS) else ["s_waitcnt vmcnt(15)", w])
    if "noload" not in FLAGS:
        out.append(f"buffer_load_dwordx4 {ST(x)}, {ofs}, {rsrc}, {koff} offen")
    return out


def half(h, slot, aset, fresh, fillers, head):
    """32 MFMAs with the fillers spread evenly over the gaps behind them"""
    lines = [x for x in head if not ("nobarrier" in FLAGS and x == "s_barrier")]
    mm = mfmas(h, slot, aset, fresh)
    n = len(fillers)
    at = {}
    for k, f in enumerate(fillers):
        at.setdefault((k * 32) // n if n else 0, []).append(f)
    for g, m in enumerate(mm):
        lines.append(m)
        for f in at.get(g, []):
            lines.extend(f if isinstance(f, list) else [f])
    return lines


def emit(lines):
    flat = []
    for l in lines:
        flat.extend(l if isinstance(l, list) else [l])
    return '"' + "\\n\\t".join(flat) + '"'


def nwrites(fill):
    return sum(1 for f in fill for l in (f if isinstance(f, list) else [f]) if l.startswith("ds_write"))


def variants(out, conds_texts):
    for n, (cond, txt, ops) in enumerate(conds_texts):
        kw = ("if constexpr (" + cond + ")") if n == 0 else ("else if constexpr (" + cond + ")" if n + 1 < len(conds_texts) else "else")
        out.append(f"    {kw} asm volatile({txt} : : {ops} : {CLOB});")


SIG = "const W4Addr& ad, const u32x4& rA, const u32x4& rB, int koffA, int koffB"
out = ["// GENERATED by scripts/gen_w4_acc.py -- do not edit (register plan and schedule: the generator's docstring).",
       "// BKC = 1: B K-contiguous (nn.Linear forward); BKC = 0: B K-strided (dgrads, the [E, in, out] expert weights).", ""]

# ---- prologue: k-tile 0 -> buffer 0, k-tile 1 -> staging registers, all of it landed (the first k-tile is a FRESH one: no vmcnt waits)
out.append("template <int BKC>")
out.append(f"__device__ __forceinline__ void w4_prologue({SIG}, int koffA1, int koffB1) {{")
cs = []
for bkc in (1, 0):
    o = Ops()
    pro = []
    for x in range(16):
        pro.append(f"buffer_load_dwordx4 {ST(x)}, {o(f'ad.aofs[{x}]') if x < 8 else o(f'ad.bofs[{x - 8}]')}, {o('rA', 's') if x < 8 else o('rB', 's')}, {o('koffA', 's') if x < 8 else o('koffB', 's')} offen")
    for x in range(16):
        wr = o("ad.wrA") if x < 8 else (o("ad.wrB[0]") if bkc else o(f"ad.wrB[{x & 1}]"))
        pro.append(["s_waitcnt vmcnt(15)", f"ds_write_b128 {wr}, {ST(x)} offset:{(x & 7) * 4096}"])
        pro.append(f"buffer_load_dwordx4 {ST(x)}, {o(f'ad.aofs[{x}]') if x < 8 else o(f'ad.bofs[{x - 8}]')}, {o('rA', 's') if x < 8 else o('rB', 's')}, {o('koffA1', 's') if x < 8 else o('koffB1', 's')} offen")
    pro.append("s_waitcnt vmcnt(0)")
    cs.append((f"BKC == {bkc}", emit(pro), o.text()))
variants(out, cs)
out.append("}")
out.append("// the barrier that publishes buffer 0, then the fragment reads of its k-step 0 (A set 0, B slot 0)")
out.append("template <int BKC>")
out.append("__device__ __forceinline__ void w4_first_reads(const W4Addr& ad) {")
cs = []
for bkc in (1, 0):
    o = Ops()
    cs.append((f"BKC == {bkc}", emit(["s_waitcnt lgkmcnt(0)", "s_barrier"] + reads_a(o, 0, 0, 0) + reads_b(o, bkc, 0, 0, 0, 0)), o.text()))
variants(out, cs)
out.append("}")
out.append("")


def gen_half(name, second_flag, build):
    """build(o, bkc, b, flag) -> (h, slot, aset, fresh, fillers, head-without-lgkm, lgkm-count or None)"""
    out.append(f"template <int BKC, int BUFI, bool {second_flag}>")
    out.append(f"__device__ __forceinline__ void {name}({SIG}) {{")
    cs = []
    for bkc in (1, 0):
        for b in range(2):
            for flag in (True, False):
                o = Ops()
                h, slot, aset, fresh, fill, head = build(o, bkc, b, flag)
                cs.append((f"BKC == {bkc} && BUFI == {b} && {second_flag if flag else '!' + second_flag}", emit(half(h, slot, aset, fresh, fill, head)), o.text()))
    variants(out, cs)
    out.append("}")


# ---- H0: k-step 0, j 0..3 (slot 0, A set 0); reads B (k-step 0, j 4..7) -> slot 1; pieces 0..5 (A)
def h0(o, bkc, b, fresh):
    fill = reads_b(o, bkc, 1, 0, 1, b)
    for x in range(6):
        fill += wl(o, bkc, x, b ^ 1, fresh)
    return 0, 0, 0, fresh, fill, ["s_waitcnt lgkmcnt(0)"]


# ---- H1: k-step 0, j 4..7 (slot 1, A set 0); reads B (k-step 1, j 0..3) -> slot 0, A (k-step 1) -> set 1; pieces 6..9
def h1(o, bkc, b, fresh):
    fill = reads_b(o, bkc, 0, 1, 0, b) + reads_a(o, 1, 1, b)
    for x in range(6, 10):
        fill += wl(o, bkc, x, b ^ 1, fresh)
    return 1, 1, 0, fresh, fill, [f"s_waitcnt lgkmcnt({0 if 'nowrite' in FLAGS else 6})"]     # H0's 6 writes follow the reads H1 needs


# ---- H2: k-step 1, j 0..3 (slot 0, A set 1); reads B (k-step 1, j 4..7) -> slot 1; pieces 10..15
def h2(o, bkc, b, nowait):
    fill = reads_b(o, bkc, 1, 1, 1, b)
    for x in range(10, 16):
        fill += wl(o, bkc, x, b ^ 1, nowait)
    return 0, 0, 1, False, fill, [f"s_waitcnt lgkmcnt({0 if 'nowrite' in FLAGS else 4})"]     # H1's 4 writes follow its reads


gen_half("w4_h0", "FRESH", h0)
gen_half("w4_h1", "FRESH", h1)
gen_half("w4_h2", "NOWAIT", h2)
# ---- barrier + H3: k-step 1, j 4..7 (slot 1, A set 1); reads the NEXT k-tile's k-step 0 (other buffer): A -> set 0, B (j 0..3) -> slot 0
out.append("template <int BKC, int BUFI>")
out.append("__device__ __forceinline__ void w4_h3(const W4Addr& ad) {")
cs = []
for bkc in (1, 0):
    for b in range(2):
        o = Ops()
        fill = reads_a(o, 0, 0, b ^ 1) + reads_b(o, bkc, 0, 0, 0, b ^ 1)
        cs.append((f"BKC == {bkc} && BUFI == {b}", emit(half(1, 1, 1, False, fill, ["s_waitcnt lgkmcnt(0)", "s_barrier"])), o.text()))
variants(out, cs)
out.append("}")
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

"""CPU-side guards of the fused attention backward (csrc/attention.hip), no GPU needed:

(1) Resources of the compiled gfx950 code.  The training shapes are latency chains per workgroup, so what a CU can hold decides
    their throughput: the two-phase kernel must fit 3 waves / SIMD (4 in its SPLIT2 form for the 256-row buckets) with no spill
    and no scratch on HALF the single-phase LDS image, i.e. hold more workgroups per CU than the single-phase kernel for every
    bucket pair; no fused kernel may stage through scratch (an earlier build did: `uint4 r[..]` arrays filled under a ternary were
    left in private memory — global -> VGPR -> scratch -> VGPR -> LDS — or promoted to LDS, adding 12 KiB per workgroup).
(2) A host model of the two-phase LDS choreography: the rows a wave lifts into registers in phase 2 go back to exactly the
    bytes they came from in phase 5, the write-backs of all waves tile the query image without overlap, and K / V never
    reach past the image.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def _kernel_resources(tmp_path):
    src = os.path.join(ROOT, "micro_diffusion_amd", "csrc", "attention.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result",
                        "-I", os.path.join(ROOT, "include"), "-Rpass-analysis=kernel-resource-usage", "-c", src,
                        "-o", str(tmp_path / "attn.o")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = {}
    for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
        name = b.split()[0]

        def field(label):
            m = re.search(label + r": (\d+)", b)
            assert m, (label, name)
            return int(m.group(1))
        out[name] = dict(vgprs=field(r"VGPRs"), spill=field(r"VGPRs Spill"), scratch=field(r"ScratchSize \[bytes/lane\]"),
                         lds=field(r"LDS Size \[bytes/block\]"), occ=field(r"Occupancy \[waves/SIMD\]"))
    return out


def _lds_1phase(hd, sqp, skp):
    return (2 * sqp + 2 * skp) * (hd + 8) * 2 + 2 * sqp * 4


def _lds_2phase(hd, sqp, skp):
    return 2 * max(sqp, skp) * (hd + 8) * 2 + 2 * sqp * 4


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_attention_kernel_resources(tmp_path):
    res = _kernel_resources(tmp_path)

    def find(kernel, *targs):
        tag = f"{kernel}ILi" + "ELi".join(str(t) for t in targs) + "EE"
        hits = [k for k in res if tag in k]
        assert len(hits) == 1, (tag, hits)
        return res[hits[0]]

    for k, v in res.items():
        assert v["spill"] == 0 and v["scratch"] == 0, f"{k}: spill {v['spill']}, scratch {v['scratch']} bytes / lane"
    for hd in (64, 32):
        for sqp, skp in ((64, 64), (64, 96), (96, 64), (96, 96), (256, 96), (256, 256), (64, 256), (256, 64), (96, 256)):
            split2 = max(sqp, skp) > 96
            hits = [k for k in res if f"attn_bwd_fused2_kernelILi{hd}ELi{sqp}ELi{skp}ELb{int(split2)}EE" in k]
            assert len(hits) == 1, (hd, sqp, skp, hits)
            two, one = res[hits[0]], find("attn_bwd_fused_kernel", hd, sqp, skp)
            if not split2:       # the two-pass form is also built for the small buckets (forced variant 4): 4 waves / SIMD as well
                alt = [k for k in res if f"attn_bwd_fused2_kernelILi{hd}ELi{sqp}ELi{skp}ELb1EE" in k]
                assert len(alt) == 1 and res[alt[0]]["occ"] >= 4 and res[alt[0]]["lds"] == two["lds"], (hd, sqp, skp, alt)
            assert two["lds"] == _lds_2phase(hd, sqp, skp) and one["lds"] == _lds_1phase(hd, sqp, skp), (hd, sqp, skp, two, one)
            assert two["occ"] >= (4 if (split2 or hd == 32) else 3), (hd, sqp, skp, two)
            # resident workgroups per CU (LDS 160 KiB, 4 SIMDs): the reason the kernel exists
            waves = max(sqp, skp) // 32
            wg2 = min(160 * 1024 // two["lds"], 4 * two["occ"] // waves)
            wg1 = min(160 * 1024 // one["lds"], 4 * one["occ"] // waves)
            if hd == 64:
                assert wg2 > wg1, f"hd {hd} {sqp}x{skp}: two-phase holds {wg2} workgroups per CU, single-phase {wg1}"
            else:
                assert wg2 >= wg1


# ---------------------------------------------------------------------------------------- host model of the LDS image
def _row_frag_bytes(wave, lane, s, pk):
    """16 bytes row_frag(tile + wave * 32 * pk, pk, s * 16, lane) reads (attention.hip): row = lane & 31, k = s*16 + (lane>>5)*8"""
    a = (wave * 32 + (lane & 31)) * pk + (s * 16 + (lane >> 5) * 8) * 2
    return range(a, a + 16)


def _writeback_bytes(wave, lane, s, pk):
    """phase 5 of attn_bwd_fused2_kernel: sA + (wave * 32 + (lane & 31)) * PK + (s * 16 + hh * 8) * 2"""
    hh = lane >> 5
    a = (wave * 32 + (lane & 31)) * pk + (s * 16 + hh * 8) * 2
    return range(a, a + 16)


def _stage_bytes(task, cpr, pk):
    r, c = divmod(task, cpr)
    return r, range(r * pk + c * 16, r * pk + c * 16 + 16)


@pytest.mark.parametrize("hd", [64, 32])
@pytest.mark.parametrize("sqp,skp,sq,skv", [(64, 64, 64, 64), (64, 96, 64, 77), (96, 96, 77, 77), (96, 64, 96, 40), (64, 64, 16, 16),
                                            (64, 96, 40, 77), (256, 256, 256, 256), (256, 96, 256, 77), (96, 256, 96, 200),
                                            (64, 256, 64, 256), (256, 64, 200, 33)])
def test_two_phase_lds_choreography(hd, sqp, skp, sq, skv):
    pk, cpr = (hd + 8) * 2, hd // 8
    rmax = max(sqp, skp)
    nt, nwaves = rmax * 2, rmax // 32
    image = rmax * pk                      # bytes of ONE of the two tiles (sA or sB)
    nq32, nk32 = (sq + 31) // 32, (skv + 31) // 32
    assert nq32 <= nwaves and nk32 <= nwaves, "every 32-row tile of either side has a wave"
    itq, itk = -(-sqp * cpr // nt), -(-skp * cpr // nt)
    # phase 1 / phase 3 staging: every (row < padded count, 16-byte column group) exactly once, inside the image
    for rows, its in ((sqp, itq), (skp, itk)):
        seen = set()
        for tid in range(nt):
            for it in range(its):
                task = tid + it * nt
                if task >= rows * cpr:
                    continue
                r, by = _stage_bytes(task, cpr, pk)
                assert r < rows and by.stop <= image
                assert not (seen & set(by))
                seen |= set(by)
        assert len(seen) == rows * hd * 2
    # delta: the cpr lanes that hold one row are adjacent lanes of one wave (the shuffle reduction relies on it)
    assert nt % cpr == 0 and 64 % cpr == 0
    # phase 2 lift / phase 5 write-back: identical bytes, waves tile rows [0, nq32 * 32) x [0, hd) without overlap
    lifted = set()
    for wave in range(nq32):
        for lane in range(64):
            for s in range(hd // 16):
                rd, wr = _row_frag_bytes(wave, lane, s, pk), _writeback_bytes(wave, lane, s, pk)
                assert rd == wr
                assert not (lifted & set(wr))
                lifted |= set(wr)
    want = {r * pk + b for r in range(nq32 * 32) for b in range(hd * 2)}
    assert lifted == want
    # role 2 reads query tiles i < nq32 and lse / delta rows < nq32 * 32 <= SQP; role 1 reads key tiles j < nk32 <= SKP / 32
    assert nq32 * 32 <= sqp and nk32 * 32 <= skp

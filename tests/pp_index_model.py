"""Host-side model of the index arithmetic of the persistent ping-pong GEMM (micro_diffusion_amd/csrc/gemm_pp.hip):
DMA source offsets -> lane-linear LDS image of a half-tile -> fragment reads (ds_read_b128 / ds_read_b64_tr_b16) ->
v_mfma_f32_32x32x16_bf16 operand layout -> accumulator -> epilogue lane map (operand swap + v_permlane32_swap).

It mirrors the device formulas line by line (same names) and is used by tests/test_pp_index_cpu.py to prove, without a
GPU, that every (row, k) element lands where the MFMA expects it and every accumulator register is stored at the right
(row, column).  Hardware semantics assumed (each is pinned on the GPU by tests/test_gemm_gpu.py probes):
  * global_load_lds_dwordx4: lane l's 16 bytes land at M0 base + 16 l;
  * ds_read_b64_tr_b16: per 16-lane group, lane i supplies the address of 4 contiguous elements forming row (i / 4),
    columns 4 (i % 4).. of a [4][16] block and receives column i (4 rows);
  * MFMA 32x32x16: operand lane l holds index (l & 31), k = 8 (l >> 5) .. + 7; D[i][j]: lane holds j = l & 31,
    i = (r & 3) + 8 (r >> 2) + 4 (l >> 5) in register r;
  * v_permlane32_swap vdst, src: lanes 32-63 of vdst swap with lanes 0-31 of src.
"""
import numpy as np

HT = 16384
B_REGION = 65536


def stage_offsets(kc, r0, rmax, ld, wave, lane):
    """element offsets (not bytes) [h][j] relative to the tile base pointer."""
    ofs = [[0, 0], [0, 0]]
    for h in range(2):
        for j in range(2):
            if kc:
                row = (wave * 2 + j) * 8 + (lane >> 3)
                c = (lane & 7) ^ ((row >> 1) & 7)
                gr = r0 + h * 128 + row
                gr = (gr if gr < rmax else rmax - 1) - r0
                ofs[h][j] = gr * ld + c * 8
            else:
                kk = (wave * 2 + j) * 4 + (lane >> 4)
                c = (lane & 15) ^ ((kk & 3) << 2)
                gc = r0 + h * 128 + c * 8
                last = (rmax - 1) & ~7
                gc = (gc if gc < last else last) - r0
                ofs[h][j] = kk * ld + gc
    return ofs


def stage_half_image(kc, mat, r0, k0, h):
    """LDS image (8192 bf16 elements, as (row, k) tags) of half-tile h of operand `mat` (logical [rows, K] array of tags)
    for the tile starting at row r0, k-tile starting at k0.  mat is indexed mat[row, k] regardless of storage."""
    rmax, K = mat.shape[0], mat.shape[1]
    img = np.full((HT // 2, 2), -1, dtype=np.int64)
    ld = K if kc else (rmax + 7) // 8 * 8          # dense storage; K-strided rows padded to 16 bytes (md_gemm_bf16 requires ld % 8 == 0)
    for wave in range(8):
        for lane in range(64):
            ofs = stage_offsets(kc, r0, rmax, ld, wave, lane)
            for j in range(2):
                dst = ((wave * 2 + j) * 1024 + lane * 16) // 2
                o = ofs[h][j]
                for e in range(8):
                    if kc:      # element (row, k) at base[row * ld + k]; base = &A[r0, k0]
                        row, k = divmod(o + e, ld)
                        img[dst + e] = (r0 + row, k0 + k)
                    else:       # element (row, k) at base[k * ld + row]; base = &A[k0, r0]
                        k, row = divmod(o + e, ld)
                        img[dst + e] = (r0 + row, k0 + k)
    return img


def frag_addrs(kc, row0, lane):
    """byte addresses relative to the half-tile slot: KC -> ad0 (k-step via XOR), KS -> ad[i]."""
    if kc:
        r = row0 + (lane & 31)
        return [r * 128 + ((((lane >> 5) ^ (r >> 1)) & 7) << 4)]
    li = lane & 15
    out = []
    for i in range(2):
        col = row0 + i * 32 + ((lane >> 4) & 1) * 16 + (li & 3) * 4
        kk = (lane >> 5) * 8 + (li >> 2)
        pc = (col >> 3) ^ ((kk & 3) << 2)
        out.append(kk * 256 + pc * 16 + ((col >> 2) & 1) * 8)
    return out


def read_frag(kc, img, row0, i, ks):
    """What the 64 lanes receive for row-fragment i, k-step ks: array [64 lanes][8 elements] of (row, k) tags."""
    out = np.zeros((64, 8, 2), dtype=np.int64)
    if kc:
        for lane in range(64):
            ad0 = frag_addrs(1, row0, lane)[0]
            addr = (ad0 ^ (ks * 32)) + i * 4096
            out[lane] = img[addr // 2: addr // 2 + 8]
        return out
    for half, extra in ((0, 0), (1, 1024)):
        addrs = [frag_addrs(0, row0, lane)[i] + ks * 4096 + extra for lane in range(64)]
        for g in range(4):
            for li in range(16):
                lane = g * 16 + li
                for jr in range(4):          # element jr of lane li = block[row jr][column li]
                    src_lane = g * 16 + 4 * jr + li // 4
                    out[lane, half * 4 + jr] = img[addrs[src_lane] // 2 + (li % 4)]
    return out


def check_operand(kc, rows, K, r0, k0, wave_strip, strip_rows):
    """Every fragment of every wave strip holds (row = r0 + h*128 + strip + i*32 + (lane & 31), k = k0 + ks*16 + 8*(lane>>5) + e)."""
    mat = np.zeros((rows, K), dtype=np.int8)
    for h in range(2):
        img = stage_half_image(kc, mat, r0, k0, h)
        for strip in wave_strip:
            for i in range(strip_rows // 32):
                for ks in range(4):
                    got = read_frag(kc, img, strip, i, ks)
                    for lane in range(64):
                        want_row = r0 + h * 128 + strip + i * 32 + (lane & 31)
                        if want_row >= rows:        # clamped / padded rows only feed outputs that are never stored,
                            assert (got[lane, :, 0] >= 0).all()     # ... but must come from staged (mapped) memory
                            continue
                        for e in range(8):
                            want = (want_row, k0 + ks * 16 + 8 * (lane >> 5) + e)
                            if tuple(got[lane, e]) != want:
                                return f"kc={kc} h={h} strip={strip} i={i} ks={ks} lane={lane} e={e}: got {tuple(got[lane, e])} want {want}"
    return None


def epilogue_map(wr, wc, IH, JH):
    """(row, col) inside the 256x256 tile that each lane's v[0..7] of block (i, pp) holds, given acc register r of
    row-fragment i = D[n = (r&3) + 8(r>>2) + 4(lane>>5)][m = lane & 31] (operands swapped: first = B fragment)."""
    res = {}
    for i in range(2):
        for pp in range(2):
            a = np.zeros((64, 4, 2), dtype=np.int64)      # group 2pp
            b = np.zeros((64, 4, 2), dtype=np.int64)      # group 2pp + 1
            for lane in range(64):
                for e in range(4):
                    for arr, r in ((a, 8 * pp + e), (b, 8 * pp + 4 + e)):
                        n = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
                        m = lane & 31
                        arr[lane, e] = (IH * 128 + wr * 64 + i * 32 + m, JH * 128 + wc * 32 + n)
            # v_permlane32_swap vdst = a, src = b: a[32:64] <-> b[0:32]
            a2, b2 = a.copy(), b.copy()
            a2[32:], b2[:32] = b[:32], a[32:]
            v = np.concatenate([a2, b2], axis=1)            # v[0..3] = new a, v[4..7] = new b
            res[(i, pp)] = v
    return res


def quad_rows_model(wr, wc, IH, JH):
    """gemm_pp_common.h: quad_rows.  Returns T[t][lane] = list of the 8 (row, col) tags the lane's 16-byte piece holds after
    (1) packing to pairs, (2) v_permlane32_swap between the two row fragments, (3) the two DPP butterfly stages inside a quad."""
    def acc_tag(i, r, lane):            # accumulator register r of fragment i (operands swapped: lane <-> row)
        n = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
        return (IH * 128 + wr * 64 + i * 32 + (lane & 31), JH * 128 + wc * 32 + n)
    # packed pairs: P[i][g][d][lane] = (tag of low half, tag of high half)
    P = [[[[(acc_tag(i, 4 * g + 2 * d, lane), acc_tag(i, 4 * g + 2 * d + 1, lane)) for lane in range(64)] for d in range(2)]
          for g in range(4)] for i in range(2)]
    W = [[None] * 64 for _ in range(4)]
    for g in range(4):
        AB = []
        for d in range(2):
            a, b = list(P[0][g][d]), list(P[1][g][d])      # half_swap(vdst = a, src = b): a[32:] <-> b[:32]
            a[32:], b[:32] = P[1][g][d][:32], P[0][g][d][32:]
            AB.append((a, b))
        for lane in range(64):
            W[g][lane] = [AB[0][0][lane], AB[1][0][lane], AB[0][1][lane], AB[1][1][lane]]     # dwords x, y, z, w
    def nb(X, mask):                    # quad_perm: the lane that differs in `mask`
        return [X[lane ^ mask] for lane in range(64)]
    def sel(cond, own, other):
        return [other[lane] if cond(lane) else own[lane] for lane in range(64)]
    odd, hi2 = (lambda l: l & 1 != 0), (lambda l: l & 2 != 0)
    nodd, nhi2 = (lambda l: l & 1 == 0), (lambda l: l & 2 == 0)
    U0, U1 = sel(odd, W[0], nb(W[1], 1)), sel(nodd, W[1], nb(W[0], 1))
    V0, V1 = sel(odd, W[2], nb(W[3], 1)), sel(nodd, W[3], nb(W[2], 1))
    T = [sel(hi2, U0, nb(V0, 2)), sel(hi2, U1, nb(V1, 2)), sel(nhi2, V0, nb(U0, 2)), sel(nhi2, V1, nb(U1, 2))]
    return [[[tag for pair in T[t][lane] for tag in pair] for lane in range(64)] for t in range(4)]


# ----------------------------------------------------------------------------------------------------------------------
# Work partition of the persistent kernel with the "whole rounds + split-K tail" form (gemm_pp.hip: kernel prologue, stager_open,
# md_gemm_pp_plan_tail): which workgroup runs which (item, k-tile range).
# ----------------------------------------------------------------------------------------------------------------------
def plan_tail(total, nk, cus, tail_mode=0, ws_units=256):
    """md_gemm_pp_plan_tail: returns (G, tail_first, units, split, tail_nk); units = 0: no tail (G = min(total, cus))."""
    G = cus & ~7
    plain = (min(total, cus), total, 0, 0, 0)
    if tail_mode == 1 or G < 8 or total <= G:
        return plain
    r = total % G
    if r == 0 or r * 2 > G:
        return plain
    iters = nk // 2
    tile_us = 3.1 * iters
    s, best = 0, (-1e30 if tail_mode == 2 else 2.0)
    for c in range(iters, 1, -1):
        if iters % c or r * c > G or r * c > ws_units:
            continue
        if tail_mode == 2:
            s = c
            break
        gain = tile_us * (1.0 - 1.0 / c) - (0.13 * r * c + 6.0)
        if gain > best:
            best, s = gain, c
    if s < 2:
        return plain
    return (G, total - r, r * s, s, nk // s)


def workgroup_items(b, G, total, nk, tail_first, units, split, tail_nk):
    """What workgroup b of a grid of G walks: [(item, first k-tile, k-tiles)], in its stream order (kernel prologue + stager_open)."""
    x, j = b & 7, b >> 3
    body = tail_first if units else total
    q, r = body >> 3, body & 7
    lo = x * (q + 1) if x < r else r * (q + 1) + (x - r) * q
    cnt = q + (1 if x < r else 0)
    stride = (G - x + 7) >> 3
    count = (cnt - j + stride - 1) // stride if j < cnt else 0
    out = [(lo + j + n * stride, 0, nk) for n in range(count)]
    if b < units:
        t = b // split
        out.append((tail_first + t, (b - t * split) * tail_nk, tail_nk))
    return out

// Persistent ping-pong bf16 MFMA GEMM ("pp256") — the kernel the large MicroDiT linear layers run on.
//
//   C[m, n] (+)= alpha * sum_k A(m, k) * B(n, k)      same contract, operand layouts and epilogues as gemm.hip
//
// Why a second kernel family.  The 2-stage kernels of gemm.hip keep the matrix pipe busy 25-28 % of the time
// (profiles/r1_gemm_pmc_ablation.txt): a wave reads its fragments, waits, multiplies, and every workgroup pays an
// un-overlapped prologue and epilogue per output tile.  This kernel is built around three ideas instead:
//
//  1. Two wave groups per workgroup run HALF A PHASE APART (8 waves, 256 x 256 tile, one workgroup per CU, one
//     wave of each group on every SIMD).  A k-tile (64 deep) is 4 phases; in a phase one group multiplies one
//     64 x 32 quadrant of its 128 x 64 output (8 x v_mfma_f32_32x32x16_bf16, s_setprio 1) while the other group
//     reads the fragments of ITS next quadrant from LDS and issues its share of the LDS-DMA prefetch, so every
//     SIMD's matrix pipe always has one wave feeding it.  Two s_barrier per phase keep the groups in step.
//  2. The operand tiles are staged as HALF-TILES (128 rows x 64 k, 16 KiB, two global_load_lds_dwordx4 per wave),
//     one per phase, into a ring of 8 slots (2 k-tiles x {A0, A1, B0, B1}).  A slot is refilled two phases after its
//     last fragment read and is read five or six phases after its DMA was issued: 4 half-tiles (64 KiB per CU) are
//     always in flight and no s_waitcnt in the steady state ever waits for the most recent ones (vmcnt(8)).
//  3. The workgroup is PERSISTENT: it walks a list of output tiles (XCD-aware order, see work_decode) as ONE stream
//     of k-tiles, so the DMA prefetch of the next tile's first k-tiles runs under the last phases of the current tile
//     and there is no per-tile prologue.  The epilogue of a finished tile is spread over the first four phases of
//     the next tile (quadrant q is written in the load half of phase q, just before that phase's MFMAs start to
//     re-accumulate it) and goes straight from the accumulator registers to HBM: the MFMA operands are swapped
//     (D = B A^T) so a lane owns 4 consecutive columns of one row, v_permlane32_swap pairs them to 8, and every global
//     access of the epilogue is 16 bytes.  Operands of fused epilogues (residual, gate, activation input, fp32
//     accumulate) are prefetched one phase ahead with hand-counted vmcnt so they never drain the DMA ring.
//
// LDS image of a half-tile (the DMA writes lane-linear 1 KiB pieces, so swizzles are applied to the SOURCE address and
// again on the fragment reads):
//   K-contiguous operand: [128 rows][8 chunks of 16 B]   physical chunk = chunk ^ ((row >> 1) & 7)   (ds_read_b128)
//   K-strided operand   : [ 64 k   ][16 chunks of 16 B]  physical chunk = chunk ^ ((k & 3) << 2)     (ds_read_b64_tr_b16)
// Rows / columns beyond M / N are CLAMPED to the last valid one (they only feed outputs that are never stored).
//
// Requirements (md_gemm_bf16 falls back to the gemm.hip kernels otherwise): K per split a multiple of 128.
//
// Replaces the implicit cuBLAS calls behind every large nn.Linear / einsum of the reference
// (micro_diffusion/models/dit.py:84-89,131-142,224; utils.py:58-61,109-111,172-173,225-233) and their backward.
#include "gemm_pp_common.h"

namespace {

constexpr bool pp_is_dact(int epi) { return epi == PP_E_DACT_GELU || epi == PP_E_DACT_MUL; }
constexpr bool pp_is_gelu(int epi) { return epi == PP_E_BF16_GELU || epi == PP_E_BF16_GELU_D; }

// ---------------------------------------------------------------------------------------------------------------------
// Epilogue of one 64 x 32 quadrant (2 row-fragments of 32 x 32) of one wave.
// After the operand swap a lane holds C[row = lane & 31][8 g + 4 hi + e] in acc[4 g + e] (hi = lane >> 5); swapping the
// register groups (2 pp, 2 pp + 1) between the half-waves leaves 8 consecutive columns (16 pp + 8 hi ..) per lane.
// The epilogue kind is a template parameter (one kernel per kind): with the mode and the activation as run-time values
// every one of the eight inlined copies carried all the branches (65 k instructions per kernel).
// ---------------------------------------------------------------------------------------------------------------------
template <int EPI>
__device__ __forceinline__ int epi_prefetch_count(const md_gemm_args& p) {
    if (EPI == PP_E_RES) return p.gate ? 5 : 4;
    if (pp_is_dact(EPI)) return 4;
    return 0;
}

// Operands of quadrant (IH, JH), requested one phase ahead, in the STORE-side layout of the quadrant (EpiLane::quad_coords:
// lane 4 q + k takes piece k of rows rq .. rq + 3 -> 64 contiguous bytes per quad and load, like the stores).  ONE request
// site per call site of the kernel (no fast / general split here: see asm_load16); rows / columns are clamped into the
// matrix so every lane issues every load (the counted waits rely on exact instruction counts) -- on interior tiles the
// clamps are no-ops.
//   residual / activation input: pre[t] = row rq + t; gate: pre[4] (rows_per_sample is a multiple of 64, so the 64 rows of a
//   quadrant share one gate row: one 16-byte piece per lane).
template <int EPI, int IH, int JH>
__device__ __forceinline__ void epi_prefetch(const md_gemm_args& p, const PPPlan& w, const EpiTile& et, u32x4 (&pre)[5],
                                             const EpiLane& el) {
    const unsigned ld = EPI == PP_E_RES ? (unsigned)p.ldr : (unsigned)p.ldaux;
    const int mlast = w.M - 1 - et.m0;                               // last valid row / first column of the last chunk, tile-relative
    const int nlast = ((w.N - 1) & ~7) - et.n0;
    int rq, cq;
    el.quad_coords(rq, cq);
    const int c = cq + JH * 128;
    const unsigned coff = (unsigned)(c < nlast ? c : nlast);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int r = rq + IH * 128 + t;
        asm_load16(pre[t], et.opbase + ((unsigned)(r < mlast ? r : mlast) * ld + coff) * 2u);
    }
    if (EPI == PP_E_RES && p.gate) {
        int r0 = et.m0 + IH * 128 + el.wrow;                          // wave-uniform: the quadrant's first row
        r0 = r0 < w.M - 1 ? r0 : w.M - 1;
        const unsigned srow = w.rps_shift >= 0 ? (unsigned)r0 >> w.rps_shift : (unsigned)r0 / (unsigned)p.rows_per_sample;
        asm_load16(pre[4], et.gbase + ((size_t)(srow * (unsigned)p.ldg) + coff) * 2);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Epilogue of one quadrant.  One wave issues at most one instruction every four cycles and the other wave group's MFMA
// half of a phase is 256 cycles long, so the instruction count of this routine is what an epilogue phase costs.  The
// first version (per 8-element block: predicate + branch, 64-bit address arithmetic from the kernel arguments, four fp32
// lane swaps, software re-interleave after the conversion: ~250 executed instructions per quadrant) made the phase 4-5x
// longer than a plain one (tile time 25.6 -> 32.7 us at K = 1024).  Now: tile base pointers formed once per tile
// (epi_open), one 32-bit lane offset per quadrant, conversion to bf16 BEFORE the half-wave swap (2 swaps of packed pairs
// instead of 4 of fp32), fp32 outputs stored straight from the accumulator layout (a lane owns 4 consecutive columns =
// 16 bytes), operands of fused epilogues swapped INTO the accumulator layout where the arithmetic is per element,
// erf-GELU on pairs with one transcendental (md_common.h), accumulators cleared by the next MFMA (C = 0).
// ---------------------------------------------------------------------------------------------------------------------
// PLAIN = interior tile, no bias, alpha == 1 (every large launch of the MicroDiT step but ragged last tiles): no predicates,
// no scaling, no bias code -- ~55 executed instructions per quadrant against ~100 for the general form (+0.5-1 %).  The
// residual epilogue exists ONLY in this form (md_gemm_pp_eligible sends ragged / biased / scaled residual problems to the
// gemm.hip kernels): with two copies of that quadrant body the kernel leaves the register budget (spills, and the operand
// registers no longer agree between request sites: scripts/check_pp_asm.py).
// What an epilogue phase still costs after that (~4 us per tile at K = 1024) is its STORES, not its instructions: 16-byte
// pieces of 32 different rows per instruction; measurements and the LDS-transposed whole-line variant that did not pay:
// profiles/r2_gemm_pp256_store_pattern.txt.
template <int EPI, int IH, int JH, bool PLAIN>
__device__ __forceinline__ void epi_quadrant(const md_gemm_args& p, const PPPlan& w, f32x16 (&acc)[2], const EpiTile& et, u32x4 (&pre)[5],
                                             const EpiLane& el) {
    constexpr bool F32OUT = (EPI == PP_E_F32);
    constexpr int ES = F32OUT ? 4 : 2;
    constexpr int CW = F32OUT ? 4 : 8;                              // columns per store
    const unsigned ldc = (unsigned)et.ldc;
    int l = el.lane;
    asm volatile("" : "+v"(l));                                   // recomputed per quadrant, not kept across the main loop
    const int rl = el.wrow + (l & 31) + IH * 128;                 // row / first column inside the tile
    const int cl = el.wcol + (l >> 5) * CW + JH * 128;
    const int mlim = et.M - et.m0, nlim = et.N - et.n0;
    const float alpha = p.alpha;
    if constexpr (F32OUT) {
        // fp32 slices (weight gradients, long K): stored straight from the accumulator layout -- a lane's 4 consecutive columns are 16 bytes
        const unsigned off0 = ((unsigned)rl * ldc + (unsigned)cl) * 4;
        const unsigned rstep = 32u * ldc * 4;
        const bool rok[2] = {PLAIN || rl < mlim, PLAIN || rl + 32 < mlim};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned off = i ? off0 + rstep : off0;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = PLAIN ? acc[i][8 * pp + e] : acc[i][8 * pp + e] * alpha;
                if (!PLAIN && p.bias) {
                    const int cb = el.wcol + (l >> 5) * 4 + JH * 128 + pp * 16;   // accumulator layout: columns cb + e, cb + 8 + e
                    if (cb < nlim) {
                        const float* bp = reinterpret_cast<const float*>(p.bias) + (int64_t)et.batch * p.sBias + et.n0 + cb;
                        const float4 b0 = *reinterpret_cast<const float4*>(bp);
                        const float4 b1 = *reinterpret_cast<const float4*>(bp + 8);
                        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                        v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                    }
                }
                float* cp = reinterpret_cast<float*>(et.cbase + pp * 64 + off);
                if (rok[i] && (PLAIN || cl + pp * 16 < nlim)) *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                if (rok[i] && (PLAIN || cl + pp * 16 + 8 < nlim)) *reinterpret_cast<float4*>(cp + 8) = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
    } else {
        // bf16 outputs: bf16(alpha * acc + bias) = what nn.Linear returns under autocast, moved into the row-run layout (quad_rows);
        // everything after the matmul (activation, residual, gate, activation derivative) acts on that bf16 value, as it does
        // in the reference, in the layout the operands were requested in.
        uint4 T[4];
        quad_rows(l, [&](int i, int g, float (&v)[4]) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = PLAIN ? acc[i][4 * g + e] : acc[i][4 * g + e] * alpha;
            if (!PLAIN && !pp_is_dact(EPI) && p.bias) {            // rare (MicroDiT's large layers have no bias): plain loads
                const int cb = el.wcol + JH * 128 + 8 * g + 4 * (l >> 5);
                if (cb < nlim) {
                    const float4 b0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.bias) + (int64_t)et.batch * p.sBias + et.n0 + cb);
                    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                }
            }
        }, T);
        int rq, cq;
        el.quad_coords(rq, cq);
        rq += IH * 128;
        cq += JH * 128;
        const bool cok = PLAIN || cq < nlim;                        // N % 8 == 0: a piece is all inside or all outside
        const unsigned off0 = ((unsigned)rq * ldc + (unsigned)cq) * 2, off2 = ((unsigned)rq * (unsigned)p.ldc2 + (unsigned)cq) * 2;
        uint4 gq = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);   // bf16 1.0
        if (EPI == PP_E_RES && p.gate) gq = landed(pre[4]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool ok = PLAIN || (cok && rq + t < mlim);
            if (!pp_is_dact(EPI) && EPI != PP_E_BF16_GELU_D && et.c2base && ok) *reinterpret_cast<uint4*>(et.c2base + (size_t)t * ((unsigned)p.ldc2 * 2u) + off2) = T[t];
            uint4 out = T[t];
            if constexpr (EPI == PP_E_BF16_GELU_D) {        // C = gelu, C2 = gelu' (the backward multiplies by it: md_gemm_args.dact_cached)
                float v[8], dv[8];
                unpack8(T[t], v);
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    f32x2 g2, d2;
                    gelu_dgelu_erf_2(f32x2{v[e], v[e + 1]}, g2, d2);
                    v[e] = g2.x;
                    v[e + 1] = g2.y;
                    dv[e] = d2.x;
                    dv[e + 1] = d2.y;
                }
                out = make_uint4(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7]));
                if (et.c2base && ok)
                    *reinterpret_cast<uint4*>(et.c2base + (size_t)t * ((unsigned)p.ldc2 * 2u) + off2) =
                        make_uint4(cvt_pk_bf16(dv[0], dv[1]), cvt_pk_bf16(dv[2], dv[3]), cvt_pk_bf16(dv[4], dv[5]), cvt_pk_bf16(dv[6], dv[7]));
            } else if constexpr (EPI == PP_E_DACT_MUL) {
                float v[8], ax[8];
                unpack8(T[t], v);
                unpack8(landed(pre[t]), ax);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= ax[e];
                out = make_uint4(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7]));
            } else if constexpr (EPI == PP_E_BF16_GELU) {
                float v[8];
                unpack8(T[t], v);
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const f32x2 g2 = gelu_erf_2(f32x2{v[e], v[e + 1]});
                    v[e] = g2.x;
                    v[e + 1] = g2.y;
                }
                out = make_uint4(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7]));
            } else if constexpr (EPI == PP_E_RES) {
                const uint4 rsu = landed(pre[t]);
                const unsigned lw[4] = {T[t].x, T[t].y, T[t].z, T[t].w}, rw[4] = {rsu.x, rsu.y, rsu.z, rsu.w}, gw[4] = {gq.x, gq.y, gq.z, gq.w};
                unsigned ow[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) {                          // one packed pair at a time: few live registers
                    const float y0 = __uint_as_float(lw[h] << 16), y1 = __uint_as_float(lw[h] & 0xffff0000u);
                    const float r0 = __uint_as_float(rw[h] << 16), r1 = __uint_as_float(rw[h] & 0xffff0000u);
                    const float g0 = __uint_as_float(gw[h] << 16), g1 = __uint_as_float(gw[h] & 0xffff0000u);
                    ow[h] = cvt_pk_bf16(r0 + g0 * y0, r1 + g1 * y1);
                }
                out = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            } else if constexpr (EPI == PP_E_DACT_GELU) {
                float v[8], ax[8];
                unpack8(T[t], v);
                unpack8(landed(pre[t]), ax);
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const f32x2 d = dgelu_erf_2(f32x2{ax[e], ax[e + 1]});
                    v[e] *= d.x;
                    v[e + 1] *= d.y;
                }
                out = make_uint4(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7]));
            }
            if (ok) *reinterpret_cast<uint4*>(et.cbase + (size_t)t * (ldc * 2u) + off0) = out;
        }
    }
    // (the accumulators are not cleared here: the first MFMA of the quadrant's next k-loop takes C = 0, PP_MFMA)
}

// Called (by every lane, all values wave-uniform) when a tile's k-loop ends: where its outputs and epilogue operands live.
template <int EPI, bool GROUPABLE>
__device__ __forceinline__ void epi_open(const md_gemm_args& p, const PPPlan& w, EpiTile& et) {
    constexpr bool F32OUT = (EPI == PP_E_F32);
    et.M = w.M;
    et.N = w.N;
    et.ldc = (int)p.ldc;
    if (GROUPABLE && w.nprob) {                 // grouped launch: this tile belongs to problem et.q; dense rows of N inside the slice
        const PPProblem& pr = w.prob[et.q];
        et.M = pr.M;
        et.N = pr.N;
        et.ldc = pr.N;
        et.cbase = const_cast<char*>(tile_base(p.C, (int64_t)et.split * p.sSplit + pr.c_off, et, pr.N, 4));
        et.c2base = nullptr;
        et.opbase = nullptr;
        et.gbase = nullptr;
        et.plain = !p.bias && p.alpha == 1.f && et.m0 + PT <= et.M && et.n0 + PT <= et.N;
        return;
    }
    et.cbase = const_cast<char*>(tile_base(p.C, (int64_t)et.batch * p.sC + (F32OUT ? (int64_t)et.split * p.sSplit : 0), et, p.ldc, F32OUT ? 4 : 2));
    et.c2base = (!F32OUT && !pp_is_dact(EPI) && p.C2) ? const_cast<char*>(tile_base(p.C2, (int64_t)et.batch * p.sC2, et, p.ldc2, 2)) : nullptr;
    et.opbase = EPI == PP_E_RES   ? tile_base(p.res, 0, et, p.ldr, 2)
                : pp_is_dact(EPI) ? tile_base(p.aux, (int64_t)et.batch * p.sAux, et, p.ldaux, 2)
                                  : nullptr;
    et.plain = !(!pp_is_dact(EPI) && p.bias) && p.alpha == 1.f && et.m0 + PT <= w.M && et.n0 + PT <= w.N;
    et.gbase = (EPI == PP_E_RES && p.gate) ? reinterpret_cast<const char*>(p.gate) + (size_t)et.n0 * 2 : nullptr;
}


template <int AKC, int BKC, int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_pp_kernel(md_gemm_args p, PPPlan w) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * B_REGION];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;      // wave group = wr; rows wr * 64 .. of each A half, columns wc * 32 .. of each B half

    // ---- this workgroup's share of the work list
    // Tail form (PPPlan::tail_*; bf16-output epilogues only): the whole tiles are items [0, tail_first); this workgroup's split-K
    // unit of a left-over tile, if it has one, is its LAST item (ordinal tail_n) -- same k-tile stream, shorter k-loop, raw fp32
    // tile written in the drain.
    constexpr bool TAILABLE = EPI != PP_E_F32;
    int w_first, w_stride, w_count;
    {
        const int G = gridDim.x, x = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int body = (TAILABLE && w.tail_units) ? w.tail_first : w.total;
        const int q = body >> 3, r = body & 7;
        const int lo = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        const int cnt = q + (x < r ? 1 : 0);
        w_stride = (G - x + 7) >> 3;               // workgroups on this XCD
        w_first = lo + j;
        w_count = j < cnt ? (int)((unsigned)(cnt - j + w_stride - 1) / (unsigned)w_stride) : 0;
    }
    const int tail_n = (TAILABLE && (int)blockIdx.x < w.tail_units) ? w_count : -1;
    if (w_count == 0 && tail_n < 0) return;
    // Optional timeline (md_gemm_args.timeline, scripts/gemm_pp_timeline.py): 16 int64 per workgroup —
    // [0] clock at entry  [1] wall clock at entry  [2] first k-tile landed  [3 + n] tile n's k-loop done (n < 8)
    // [11] clock at exit  [12] wall clock at exit  [13] XCC_ID << 32 | HW_ID
    long long* const tl = p.timeline ? static_cast<long long*>(p.timeline) + (size_t)blockIdx.x * 16 : nullptr;
    if (tl && tid == 0) { tl[0] = clock64(); tl[1] = wall_clock64(); }
    const int half_iters = w_count * (w.nk >> 1) + (tail_n >= 0 ? (w.tail_nk >> 1) : 0);  // loop iterations: two k-tiles each

    // ---- per-lane constants
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;
    unsigned adA[4], adB[4];
    frag_addrs<AKC>(adA, lds0, wr * 64, lane);
    frag_addrs<BKC>(adB, lds0 + B_REGION, wc * 32, lane);
    // grouped launches (several problems, md_gemm_args.problems) exist for the weight-gradient kernel only (both operands
    // K-strided, fp32 slices); every other instantiation keeps its code unchanged
    constexpr bool GROUPABLE = !AKC && !BKC && EPI == PP_E_F32;
    int64_t a_kstep = AKC ? (int64_t)BKT * 2 : (int64_t)BKT * w.lda * 2;   // bytes per k-tile (per problem in a grouped launch)
    int64_t b_kstep = BKC ? (int64_t)BKT * 2 : (int64_t)BKT * w.ldb * 2;
    const EpiLane el = {lane, wr * 64, wc * 32};

    // ---- stager (DMA prefetch) cursor
    int s_n = 0, s_kt = 0;                         // item ordinal, k-tile inside the item
    bool s_live = true;
    const char *sA = nullptr, *sB = nullptr;       // uniform base pointers of the stager's current k-tile
    unsigned aofs[2][2], bofs[2][2];
    int s_m0 = 0, s_n0 = 0, s_li = 0, s_seg_kt = 0;   // operand lists: tile origin, first list index and k-tiles left in the segment
    auto stager_segment = [&](int li) {            // point the stager at operand pair li of the lists (uniform -> s_load)
        const bf16* Ab = reinterpret_cast<const bf16*>(p.A_list[li]);
        const bf16* Bb = reinterpret_cast<const bf16*>(p.B_list[li]);
        sA = reinterpret_cast<const char*>(AKC ? Ab + (int64_t)s_m0 * w.lda : Ab + s_m0);
        sB = reinterpret_cast<const char*>(BKC ? Bb + (int64_t)s_n0 * w.ldb : Bb + s_n0);
        s_seg_kt = w.nk_seg;
    };
    auto stager_open = [&](int n) {                // point the stager at item ordinal n
        int m0, n0, batch, split;
        if (GROUPABLE && w.nprob) {                // grouped launch: the item's problem supplies operands, extents, leading dimensions
            int q;
            work_decode_grouped(w, w_first + n * w_stride, q, m0, n0, split);
            const PPProblem& pr = w.prob[q];
            stage_offsets<AKC>(aofs, m0, pr.M, pr.lda, wave, lane);
            stage_offsets<BKC>(bofs, n0, pr.N, pr.ldb, wave, lane);
            a_kstep = (int64_t)BKT * pr.lda * 2;
            b_kstep = (int64_t)BKT * pr.ldb * 2;
            const int64_t kb = (int64_t)split * w.kspan;
            sA = reinterpret_cast<const char*>(reinterpret_cast<const bf16*>(pr.A) + kb * pr.lda + m0);
            sB = reinterpret_cast<const char*>(reinterpret_cast<const bf16*>(pr.B) + kb * pr.ldb + n0);
            return;
        }
        int item = w_first + n * w_stride, ksub = 0;
        if (TAILABLE && n == tail_n) {             // this workgroup's split-K unit of a left-over tile
            const unsigned u = blockIdx.x, t = u / (unsigned)w.tail_split;
            item = w.tail_first + (int)t;
            ksub = (int)(u - t * (unsigned)w.tail_split) * w.tail_nk * BKT;
        }
        work_decode(w, item, m0, n0, batch, split);
        stage_offsets<AKC>(aofs, m0, w.M, w.lda, wave, lane);
        stage_offsets<BKC>(bofs, n0, w.N, w.ldb, wave, lane);
        if (EPI == PP_E_F32 && p.A_list) {         // operand lists (fp32-slice kernels only: the others are at their register budget)
            s_m0 = m0;
            s_n0 = n0;
            s_li = __builtin_amdgcn_readfirstlane((batch * w.ksplit + split) * w.nseg);
            stager_segment(s_li);
            return;
        }
        const int64_t kbeg = (int64_t)split * w.kspan + ksub;
        const bf16* Ab = reinterpret_cast<const bf16*>(p.A) + (int64_t)batch * p.sA;
        const bf16* Bb = reinterpret_cast<const bf16*>(p.B) + (int64_t)batch * p.sB;
        sA = reinterpret_cast<const char*>(AKC ? Ab + (int64_t)m0 * w.lda + kbeg : Ab + kbeg * w.lda + m0);
        sB = reinterpret_cast<const char*>(BKC ? Bb + (int64_t)n0 * w.ldb + kbeg : Bb + kbeg * w.ldb + n0);
    };
    auto stager_advance = [&]() {                  // after the last half-tile (A1) of a k-tile
        sA += a_kstep;
        sB += b_kstep;
        if (++s_kt == ((TAILABLE && s_n == tail_n) ? w.tail_nk : w.nk)) {
            s_kt = 0;
            if (++s_n < w_count + (tail_n >= 0 ? 1 : 0)) stager_open(s_n);   // (the tail unit is one more item)
            else s_live = false;
        } else if (EPI == PP_E_F32 && p.A_list && --s_seg_kt == 0) {
            stager_segment(++s_li);                // the item goes on with the next operand pair of its list (same accumulators)
        }
    };
    // KIND: 0 = A0, 1 = B0, 2 = B1, 3 = A1 (the order they are consumed in); BUF = k-tile parity
#define PP_STAGE(KIND, BUF)                                                                                             \
    do {                                                                                                                \
        if (s_live) {                                                                                                   \
            if (KIND == 0) stage_half(sA, aofs[0], smem + (BUF) * 32768, wave);                                         \
            else if (KIND == 1) stage_half(sB, bofs[0], smem + B_REGION + (BUF) * 32768, wave);                         \
            else if (KIND == 2) stage_half(sB, bofs[1], smem + B_REGION + (BUF) * 32768 + HT, wave);                    \
            else { stage_half(sA, aofs[1], smem + (BUF) * 32768 + HT, wave); stager_advance(); }                        \
        }                                                                                                               \
    } while (0)

    // ---- compute cursor / epilogue state
    int c_n = 0, c_kt = 0;
    bool epi_pending = false;
    int pf_after = 0;                              // DMA instructions issued after the pending operand prefetch
    EpiTile et = {0, 0, 0, 0, 0, 0, 0, 0, false, nullptr, nullptr, nullptr, nullptr};
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // the epilogues with prefetched operands are built in the PLAIN form only (two copies of their quadrant body do not fit
    // the register budget); md_gemm_pp_eligible sends their ragged / biased / scaled problems to the gemm.hip kernels
    constexpr bool PLAIN_ONLY = EPI == PP_E_RES || pp_is_dact(EPI);
    const int npf = epi_prefetch_count<EPI>(p);
    const bool has_ops = npf > 0;
    u32x4 pre[5];
#pragma unroll
    for (int x = 0; x < 5; ++x) pre[x] = u32x4{0u, 0u, 0u, 0u};

    f32x16 acc[4][2];                              // quadrant q = ih * 2 + jh, row-fragment i
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][i][r] = 0.f;
    Frag<AKC> fa[2][4];                            // [row-fragment][k-step] of the current A half
    Frag<BKC> fb0[4], fb1[4];                      // [k-step] of B half 0 / 1

    // ---- prologue: k-tile 0 complete, k-tile 1's A0 / B0
    stager_open(0);
    PP_STAGE(0, 0); PP_STAGE(1, 0); PP_STAGE(2, 0); PP_STAGE(3, 0);
    PP_STAGE(0, 1); PP_STAGE(1, 1);
    PP_VMCNT(8);                                   // A0, B0 of k-tile 0 landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    if (tl && tid == 0) tl[2] = clock64();
    // Group 1 runs one barrier behind group 0: two barriers per phase (load | compute) enforce strict alternation of the
    // groups' MFMA halves.  (A one-barrier-per-phase variant — group 0 passing it before its load half, group 1 between its
    // load and compute halves, so that no wave idles when its partner's half is the longer one — was correct and 5-10 %
    // SLOWER on every shape: profiles/r2_gemm_pp256_one_vs_two_barriers.txt.)
    if (wr == 1) __builtin_amdgcn_s_barrier();

#define PP_READ_A(BUF, IH)                                                                                              \
    do {                                                                                                                \
        frag_read_at<AKC, (BUF) * 32768 + (IH) * HT, 0, 0>(fa[0][0], adA, kx32); frag_read_at<AKC, (BUF) * 32768 + (IH) * HT, 1, 0>(fa[1][0], adA, kx32); \
        frag_read_at<AKC, (BUF) * 32768 + (IH) * HT, 0, 1>(fa[0][1], adA, kx32); frag_read_at<AKC, (BUF) * 32768 + (IH) * HT, 1, 1>(fa[1][1], adA, kx32); \
        frag_read_at<AKC, (BUF) * 32768 + (IH) * HT, 0, 2>(fa[0][2], adA, kx32); frag_read_at<AKC, (BUF) * 32768 + (IH) * HT, 1, 2>(fa[1][2], adA, kx32); \
        frag_read_at<AKC, (BUF) * 32768 + (IH) * HT, 0, 3>(fa[0][3], adA, kx32); frag_read_at<AKC, (BUF) * 32768 + (IH) * HT, 1, 3>(fa[1][3], adA, kx32); \
    } while (0)
#define PP_READ_B(FB, BUF, JH)                                                                                          \
    do {                                                                                                                \
        frag_read_at<BKC, (BUF) * 32768 + (JH) * HT, 0, 0>(FB[0], adB, kx32); frag_read_at<BKC, (BUF) * 32768 + (JH) * HT, 0, 1>(FB[1], adB, kx32); \
        frag_read_at<BKC, (BUF) * 32768 + (JH) * HT, 0, 2>(FB[2], adB, kx32); frag_read_at<BKC, (BUF) * 32768 + (JH) * HT, 0, 3>(FB[3], adB, kx32); \
    } while (0)
#define PP_PIN_A()                                                                                                      \
    do {                                                                                                                \
        frag_pin<AKC>(fa[0][0]); frag_pin<AKC>(fa[1][0]); frag_pin<AKC>(fa[0][1]); frag_pin<AKC>(fa[1][1]);             \
        frag_pin<AKC>(fa[0][2]); frag_pin<AKC>(fa[1][2]); frag_pin<AKC>(fa[0][3]); frag_pin<AKC>(fa[1][3]);             \
    } while (0)
#define PP_PIN_B(FB)                                                                                                    \
    do { frag_pin<BKC>(FB[0]); frag_pin<BKC>(FB[1]); frag_pin<BKC>(FB[2]); frag_pin<BKC>(FB[3]); } while (0)
    // D = B A^T: first operand = the B fragment (its 32 "rows" are output columns), second = the A fragment
    // FRESH (wave-uniform): this quadrant was written out in the load half of this phase -- its first two MFMAs take C = 0
    // (an inline constant) instead of 32 v_mov to clear the accumulators.
#define PP_MFMA(Q, FB, FRESH)                                                                                           \
    do {                                                                                                                \
        {                                                                                                               \
            const bf16x8 bq = FB[0].get();                                                                              \
            if (FRESH) {                                                                                                \
                acc[Q][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq, fa[0][0].get(), zero16, 0, 0, 0);               \
                acc[Q][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq, fa[1][0].get(), zero16, 0, 0, 0);               \
            } else {                                                                                                    \
                acc[Q][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq, fa[0][0].get(), acc[Q][0], 0, 0, 0);            \
                acc[Q][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq, fa[1][0].get(), acc[Q][1], 0, 0, 0);            \
            }                                                                                                           \
        }                                                                                                               \
        _Pragma("unroll") for (int ks = 1; ks < 4; ++ks) {                                                              \
            const bf16x8 bq = FB[ks].get();                                                                             \
            acc[Q][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq, fa[0][ks].get(), acc[Q][0], 0, 0, 0);               \
            acc[Q][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq, fa[1][ks].get(), acc[Q][1], 0, 0, 0);               \
        }                                                                                                               \
    } while (0)
    // The load half of an epilogue phase: operands of this quadrant ready -> write it -> prefetch the next quadrant's.
#define PP_EPI(IH, JH, Q, NIH, NJH, HAS_NEXT)                                                                           \
    do {                                                                                                                \
        if (epi_pending) {                                                                                              \
            if (has_ops) { if (pf_after) PP_VMCNT(2); else PP_VMCNT(0); }                                               \
            if (PLAIN_ONLY || et.plain) epi_quadrant<EPI, IH, JH, true>(p, w, acc[Q], et, pre, el);                     \
            else epi_quadrant<EPI, IH, JH, PLAIN_ONLY>(p, w, acc[Q], et, pre, el);                                      \
            if (HAS_NEXT) {                                                                                             \
                if (has_ops) {                                                                                          \
                    epi_prefetch<EPI, NIH, NJH>(p, w, et, pre, el);                                                     \
                    pf_after = 0;                                                                                       \
                }                                                                                                       \
            }                                                                                                           \
            else epi_pending = false;                                                                                   \
        }                                                                                                               \
    } while (0)
    // RAW guard of the DMA ring: the half-tile issued 4 phases ago has landed (8 younger DMA instructions may be pending).
    // vmcnt also counts the epilogue stores, so in an epilogue phase this waits for more than it must; counts that allow for
    // the stores (8 + stores per quadrant x quadrants written in the last four phases) were measured and LOST 3-7 % on the
    // long-K shapes -- the extra uniform branches per phase cost more than the waits (profiles/r2_gemm_pp256_epilogue_v2.txt).
    // In an epilogue phase with prefetched operands the operand wait at its top has already proven the landing (loads
    // complete in order) and a count of 8 here would wait for the prefetch just issued.
#define PP_RAW_WAIT(EPI_PHASE)                                                                                          \
    do {                                                                                                                \
        if (!((EPI_PHASE) && has_ops && epi_was)) {                                                                     \
            if (!s_live) PP_VMCNT(0);                                                                                   \
            else PP_VMCNT(8);                                                                                           \
        }                                                                                                               \
    } while (0)
#define PP_PIN_NONE() do { } while (0)
#define PP_COMPUTE(Q, FB, PINS, FRESH)                                                                                       \
    do {                                                                                                                \
        __builtin_amdgcn_s_barrier();                                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
        PINS;                                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        __builtin_amdgcn_s_setprio(1);                                                                                  \
        PP_MFMA(Q, FB, FRESH);                                                                                          \
        __builtin_amdgcn_s_setprio(0);                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        __builtin_amdgcn_s_barrier();                                                                                   \
    } while (0)

    for (int it = 0; it < half_iters; ++it) {
        const bool tail_item = TAILABLE && c_n == tail_n;          // (its raw tile is written in the drain: no epilogue set-up)
        const bool last_pair = (c_kt + 2 == (tail_item ? w.tail_nk : w.nk));
        int kx32 = 32;
        asm volatile("" : "+s"(kx32));
        bool epi_was;
        // ================= k-tile in buffer 0 =================
        // phase 1: quadrant (0, 0) <- A0 B0
        epi_was = epi_pending;
        PP_EPI(0, 0, 0, 0, 1, true);
        PP_READ_B(fb0, 0, 0); PP_READ_A(0, 0);
        PP_STAGE(2, 1); if (s_live) pf_after = 2;
        PP_RAW_WAIT(true);
        PP_COMPUTE(0, fb0, PP_PIN_B(fb0); PP_PIN_A(), epi_was);
        // phase 2: quadrant (0, 1) <- A0 B1
        epi_was = epi_pending;
        PP_EPI(0, 1, 1, 1, 1, true);
        PP_READ_B(fb1, 0, 1);
        { const bool live = s_live; PP_STAGE(3, 1); if (live) pf_after = 2; }
        PP_RAW_WAIT(true);
        PP_COMPUTE(1, fb1, PP_PIN_B(fb1), epi_was);
        // phase 3: quadrant (1, 1) <- A1 B1
        epi_was = epi_pending;
        PP_EPI(1, 1, 3, 1, 0, true);
        PP_READ_A(0, 1);
        PP_STAGE(0, 0); if (s_live) pf_after = 2;
        PP_RAW_WAIT(true);
        PP_COMPUTE(3, fb1, PP_PIN_A(), epi_was);
        // phase 4: quadrant (1, 0) <- A1 B0
        epi_was = epi_pending;
        PP_EPI(1, 0, 2, 0, 0, false);
        PP_STAGE(1, 0);
        PP_RAW_WAIT(true);
        PP_COMPUTE(2, fb0, PP_PIN_NONE(), epi_was);
        // ================= k-tile in buffer 1 =================
        epi_was = false;
        PP_READ_B(fb0, 1, 0); PP_READ_A(1, 0);
        PP_STAGE(2, 0);
        PP_RAW_WAIT(false);
        PP_COMPUTE(0, fb0, PP_PIN_B(fb0); PP_PIN_A(), false);
        PP_READ_B(fb1, 1, 1);
        PP_STAGE(3, 0);
        PP_RAW_WAIT(false);
        PP_COMPUTE(1, fb1, PP_PIN_B(fb1), false);
        PP_READ_A(1, 1);
        PP_STAGE(0, 1);
        PP_RAW_WAIT(false);
        PP_COMPUTE(3, fb1, PP_PIN_A(), false);
        // phase 8: on the last k-tile pair of an output tile, request the operands of the first epilogue quadrant
        if (last_pair && !tail_item) {
            if (GROUPABLE && w.nprob) {
                work_decode_grouped(w, w_first + c_n * w_stride, et.q, et.m0, et.n0, et.split);
                et.batch = 0;
            } else {
                work_decode(w, w_first + c_n * w_stride, et.m0, et.n0, et.batch, et.split);
            }
            epi_open<EPI, GROUPABLE>(p, w, et);
            if (has_ops) epi_prefetch<EPI, 0, 0>(p, w, et, pre, el);
        }
        {
            const bool live = s_live;
            PP_STAGE(1, 1);
            pf_after = live ? 2 : 0;
            if (last_pair && has_ops && !tail_item) {
                // exact count: the 8 younger DMA instructions plus the operand loads just issued are all loads (in order)
                if (!live) PP_VMCNT(0);
                else if (npf == 5) PP_VMCNT(13);
                else PP_VMCNT(12);
            } else {
                if (live) PP_VMCNT(8); else PP_VMCNT(0);
            }
        }
        PP_COMPUTE(2, fb0, PP_PIN_NONE(), false);
        c_kt += 2;
        if (last_pair) {
            if (tl && tid == 0 && c_n < 8) tl[3 + c_n] = clock64();
            c_kt = 0; ++c_n; epi_pending = true;
        }
    }
    // ---- drain: the last tile's accumulators
    if (wr == 0) __builtin_amdgcn_s_barrier();     // pairs with group 1's extra barrier at the start
    PP_VMCNT(0);                                   // quadrant (0, 0)'s operands were requested in the last phase 8
   
#define PP_DRAIN_Q(IH, JH, Q)                                                                                           \
    do {                                                                                                                \
        if (PLAIN_ONLY || et.plain) epi_quadrant<EPI, IH, JH, true>(p, w, acc[Q], et, pre, el);                         \
        else epi_quadrant<EPI, IH, JH, PLAIN_ONLY>(p, w, acc[Q], et, pre, el);                                          \
    } while (0)
#define PP_DRAIN_PF(IH, JH)                                                                                             \
    do {                                                                                                                \
        if (has_ops) {                                                                                                  \
            epi_prefetch<EPI, IH, JH>(p, w, et, pre, el);                                                               \
            PP_VMCNT(0);                                                                                                \
        }                                                                                                               \
    } while (0)
    if (TAILABLE && tail_n >= 0) {
        // split-K unit: the raw accumulators as a dense 256 x 256 fp32 tile (rows beyond M / N hold clamped-operand products
        // the fix-up pass never reads)
        et.cbase = static_cast<char*>(p.tail_ws) + (size_t)blockIdx.x * (size_t)(PT * PT * 4);
        et.ldc = PT;
        epi_quadrant<PP_E_F32, 0, 0, true>(p, w, acc[0], et, pre, el);
        epi_quadrant<PP_E_F32, 0, 1, true>(p, w, acc[1], et, pre, el);
        epi_quadrant<PP_E_F32, 1, 1, true>(p, w, acc[3], et, pre, el);
        epi_quadrant<PP_E_F32, 1, 0, true>(p, w, acc[2], et, pre, el);
    } else {
        PP_DRAIN_Q(0, 0, 0);
        PP_DRAIN_PF(0, 1);
        PP_DRAIN_Q(0, 1, 1);
        PP_DRAIN_PF(1, 1);
        PP_DRAIN_Q(1, 1, 3);
        PP_DRAIN_PF(1, 0);
        PP_DRAIN_Q(1, 0, 2);
    }
    if (tl) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // include the store drain
        if (tid == 0) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            tl[11] = clock64();
            tl[12] = wall_clock64();
            tl[13] = ((long long)xcc << 32) | hw;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Second launch of the tail form: out tile = epilogue(sum of the tail_split raw fp32 partials of a left-over tile).  One thread
// per 16-byte piece of the output (8 columns); the arithmetic after the sum is the one epi_quadrant applies to a whole tile
// (bf16(alpha * acc + bias) first, then activation / gated residual / activation derivative on that bf16 value).
// r tiles x 64 Ki elements: a few microseconds, bound by the launch itself.
// ---------------------------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256) void pp_tail_fixup_kernel(md_gemm_args p, PPPlan w) {
    const int t = blockIdx.x >> 5;                                   // left-over tile
    const int e = (blockIdx.x & 31) * 256 + threadIdx.x;             // 16-byte piece inside the tile: row e / 32, columns 8 (e % 32) ..
    const int row = e >> 5, c8 = (e & 31) * 8;
    int m0, n0, batch, split;
    work_decode(w, w.tail_first + t, m0, n0, batch, split);
    const int gr = m0 + row, gc = n0 + c8;
    if (gr >= w.M || gc >= w.N) return;                              // N % 8 == 0: a piece is all inside or all outside
    const float* src = static_cast<const float*>(p.tail_ws) + ((size_t)t * w.tail_split) * (size_t)(PT * PT) + row * PT + c8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < w.tail_split; j0 += 8) {                   // fixed order: deterministic; 16 independent loads in flight per
        f32x4 a[8], b[8];                                            // thread (one at a time the pass was a chain of round trips)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + u < w.tail_split ? j0 + u : j0;       // (re-reads a valid partial; its value is dropped below)
            a[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)j * (PT * PT)));
            b[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)j * (PT * PT) + 4));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (j0 + u < w.tail_split) {
                v[0] += a[u][0]; v[1] += a[u][1]; v[2] += a[u][2]; v[3] += a[u][3];
                v[4] += b[u][0]; v[5] += b[u][1]; v[6] += b[u][2]; v[7] += b[u][3];
            }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= p.alpha;
    if (!pp_is_dact(EPI) && p.bias) {
        const float* bp = reinterpret_cast<const float*>(p.bias) + (int64_t)batch * p.sBias + gc;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += bp[i];
    }
    const uint4 T = make_uint4(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7]));
    if (!pp_is_dact(EPI) && EPI != PP_E_BF16_GELU_D && p.C2)
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(p.C2) + (int64_t)batch * p.sC2 + (int64_t)gr * p.ldc2 + gc) = T;
    uint4 out = T;
    if constexpr (EPI == PP_E_BF16_GELU_D) {
        float y[8], dy[8];
        unpack8(T, y);
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            f32x2 g2, d2;
            gelu_dgelu_erf_2(f32x2{y[i], y[i + 1]}, g2, d2);
            y[i] = g2.x;
            y[i + 1] = g2.y;
            dy[i] = d2.x;
            dy[i + 1] = d2.y;
        }
        out = make_uint4(cvt_pk_bf16(y[0], y[1]), cvt_pk_bf16(y[2], y[3]), cvt_pk_bf16(y[4], y[5]), cvt_pk_bf16(y[6], y[7]));
        if (p.C2)
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(p.C2) + (int64_t)batch * p.sC2 + (int64_t)gr * p.ldc2 + gc) =
                make_uint4(cvt_pk_bf16(dy[0], dy[1]), cvt_pk_bf16(dy[2], dy[3]), cvt_pk_bf16(dy[4], dy[5]), cvt_pk_bf16(dy[6], dy[7]));
    } else if constexpr (EPI == PP_E_DACT_MUL) {
        float y[8], ax[8];
        unpack8(T, y);
        unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(p.aux) + (int64_t)batch * p.sAux + (int64_t)gr * p.ldaux + gc), ax);
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] *= ax[i];
        out = make_uint4(cvt_pk_bf16(y[0], y[1]), cvt_pk_bf16(y[2], y[3]), cvt_pk_bf16(y[4], y[5]), cvt_pk_bf16(y[6], y[7]));
    } else if constexpr (EPI == PP_E_BF16_GELU) {
        float y[8];
        unpack8(T, y);
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            const f32x2 g2 = gelu_erf_2(f32x2{y[i], y[i + 1]});
            y[i] = g2.x;
            y[i + 1] = g2.y;
        }
        out = make_uint4(cvt_pk_bf16(y[0], y[1]), cvt_pk_bf16(y[2], y[3]), cvt_pk_bf16(y[4], y[5]), cvt_pk_bf16(y[6], y[7]));
    } else if constexpr (EPI == PP_E_RES) {
        float y[8], r[8], g[8];
        unpack8(T, y);
        unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(p.res) + (int64_t)gr * p.ldr + gc), r);
        if (p.gate) {
            unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(p.gate) + (int64_t)(gr / p.rows_per_sample) * p.ldg + gc), g);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] = 1.f;
        }
        out = make_uint4(cvt_pk_bf16(r[0] + g[0] * y[0], r[1] + g[1] * y[1]), cvt_pk_bf16(r[2] + g[2] * y[2], r[3] + g[3] * y[3]),
                         cvt_pk_bf16(r[4] + g[4] * y[4], r[5] + g[5] * y[5]), cvt_pk_bf16(r[6] + g[6] * y[6], r[7] + g[7] * y[7]));
    } else if constexpr (EPI == PP_E_DACT_GELU) {
        float y[8], ax[8];
        unpack8(T, y);
        unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(p.aux) + (int64_t)batch * p.sAux + (int64_t)gr * p.ldaux + gc), ax);
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            const f32x2 d = dgelu_erf_2(f32x2{ax[i], ax[i + 1]});
            y[i] *= d.x;
            y[i + 1] *= d.y;
        }
        out = make_uint4(cvt_pk_bf16(y[0], y[1]), cvt_pk_bf16(y[2], y[3]), cvt_pk_bf16(y[4], y[5]), cvt_pk_bf16(y[6], y[7]));
    }
    *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(p.C) + (int64_t)batch * p.sC + (int64_t)gr * p.ldc + gc) = out;
}

}  // namespace

// Layout x epilogue combinations that are instantiated = the ones the MicroDiT engine launches at sizes that fill the chip:
//   NT (activations x torch weights): bf16, residual, dact(gelu)   NN (dgrads, MoE experts): bf16, bf16+gelu, residual, f32
//   TN (weight gradients): f32 slices
static bool pp_instantiated(int akc, int bkc, int epi) {
    if (akc && bkc) return epi == PP_E_BF16 || epi == PP_E_RES || epi == PP_E_DACT_GELU || epi == PP_E_DACT_MUL;
    if (akc && !bkc) return epi == PP_E_BF16 || epi == PP_E_BF16_GELU || epi == PP_E_BF16_GELU_D || epi == PP_E_RES || epi == PP_E_F32;
    if (!akc && !bkc) return epi == PP_E_F32;
    return false;
}

bool md_gemm_pp_eligible(const md_gemm_args* a) {
    const int epi = md_gemm_pp_epi_kind(a);
    if (epi < 0 || !pp_instantiated(a->a_kcontig, a->b_kcontig, epi)) return false;
    return md_gemm_pp_shape_ok(a, epi);
}

// Everything md_gemm_pp_eligible asks of a problem except "this kernel family has the instantiation" (the w4 kernel shares the plan,
// the ranges and the plain-epilogue rule, and builds one epilogue kind pp256 does not have).
bool md_gemm_pp_shape_ok(const md_gemm_args* a, int epi) {
    if (a->K % a->ksplit) return false;
    const int64_t kspan = a->K / a->ksplit;
    if (kspan < 128 || kspan % 128) return false;
    if (a->N % 8) return false;                                  // 16-byte column chunks everywhere
    if ((a->A_list || a->B_list) && epi != PP_E_F32) return false;   // operand lists are built into the fp32-slice kernels only
    if (a->problems || a->n_problems) {                              // grouped launch: the weight-gradient kernel only
        if (!a->problems || a->n_problems < 1 || a->n_problems > MD_GEMM_MAX_PROBLEMS) return false;
        if (a->a_kcontig || a->b_kcontig || epi != PP_E_F32 || a->A_list || a->bias || a->alpha != 1.f || a->batch != 1) return false;
        for (int i = 0; i < a->n_problems; ++i) {
            const md_gemm_problem& s = a->problems[i];
            if (!s.A || !s.B || s.M <= 0 || s.N <= 0 || s.N % 8 || s.lda % 8 || s.ldb % 8 || s.c_off % 4 || s.c_off < 0) return false;
            if (s.lda > (1 << 22) || s.ldb > (1 << 22) || s.N > (1 << 20) || s.M >= (1 << 30)) return false;
        }
    }
    if (a->A_list && a->list_segments > 1 && (kspan % ((int64_t)a->list_segments * 128))) return false;   // whole k-tile pairs per segment
    if (epi == PP_E_RES && a->gate && a->rows_per_sample % 64) return false;   // one gate row per 64-row quadrant
    if ((epi == PP_E_RES || pp_is_dact(epi) || epi == PP_E_DACT_SWIGLU) && (a->bias || a->alpha != 1.f || a->M % PT || a->N % PT))
        return false;                                            // only the PLAIN form of these two epilogues is built
    if (a->M >= (1 << 30) || a->N >= (1 << 30) || a->K >= (1 << 30)) return false;
    if (a->lda > (1 << 22) || a->ldb > (1 << 22)) return false;  // 32-bit per-lane DMA offsets
    if (a->ldc > (1 << 20) || a->ldc2 > (1 << 20) || a->ldr > (1 << 20) || a->ldaux > (1 << 20) || a->ldg > (1 << 20))
        return false;                                            // 32-bit per-lane epilogue offsets (256 rows x ld x 4 bytes)
    return true;
}

// Whole rounds + split-K tail (md_gemm_args.tail_ws).  total = R * G + r items on G workgroups: without the tail the r left-over
// tiles cost a whole round (one tile time, ~3.1 us per pair of k-tiles at the sustained clock); with a split s they cost 1 / s of
// a round PLUS what the form itself costs: r * s raw 256 KiB tiles written by the main launch and read back by the fix-up
// (~0.13 us per unit at ~4 TB/s, both directions: measured -- 64 tiles x 4 units made the fix-up a 20 us launch) and the second
// launch (~6 us with its gap).  The split is the divisor of the k-loop's iteration count that maximises the predicted gain; the
// form is taken when that gain is > 2 us (tail_mode 0) or whenever the structure allows a split (tail_mode 2: the largest one):
// bf16-output epilogue, one operand pair, at least one whole round, r <= G / 2, s >= 2 dividing the k-loop into whole iterations
// (two k-tiles) with r * s <= G, and a workspace that holds r * s raw tiles.
static bool md_gemm_pp_plan_tail(const md_gemm_args* a, int epi, int cus, PPPlan* w) {
    if (!a->tail_ws || a->tail_mode == 1 || epi == PP_E_F32 || a->A_list || a->problems || a->timeline) return false;
    const int G = cus & ~7;                      // whole tiles are dealt XCD by XCD: every XCD needs the same number of workgroups
    if (G < 8 || w->total <= G) return false;
    const int r = w->total % G;
    if (r == 0 || r * 2 > G) return false;
    const int iters = w->nk >> 1;
    const double tile_us = 3.1 * iters;
    int s = 0;
    double best = a->tail_mode == 2 ? -1e30 : 2.0;
    for (int c = iters; c >= 2; --c) {
        if (iters % c || (int64_t)r * c > G || (int64_t)r * c * (PT * PT * 4) > a->tail_ws_bytes) continue;
        if (a->tail_mode == 2) { s = c; break; }                 // forced: the largest split (tests exercise the most units)
        const double gain = tile_us * (1.0 - 1.0 / c) - (0.13 * r * c + 6.0);
        if (gain > best) { best = gain; s = c; }
    }
    if (s < 2) return false;
    w->tail_first = w->total - r;
    w->tail_units = r * s;
    w->tail_split = s;
    w->tail_nk = w->nk / s;
    return true;
}

int md_gemm_pp_launch(const md_gemm_args* a, hipStream_t stream) {
    PPPlan w;
    if (!md_gemm_pp_plan(a, &w)) return MD_BAD_ARG;
    const int cus = (a->cu_limit > 0 && a->cu_limit < NUM_CU) ? a->cu_limit : NUM_CU;     // md_gemm_args.cu_limit: CUs left to a collective
    unsigned G = (unsigned)(w.total < cus ? w.total : cus);
    const int epi = md_gemm_pp_epi_kind(a);
    const bool tail = md_gemm_pp_plan_tail(a, epi, cus, &w);
    if (tail) G = (unsigned)(cus & ~7);
    if (a->tail_used) *a->tail_used = tail ? w.tail_split : 0;
    const dim3 grid(G, 1, 1), block(512);
#define PP_LAUNCH(AK, BK, E) hipLaunchKernelGGL((gemm_bf16_pp_kernel<AK, BK, E>), grid, block, 0, stream, *a, w)
#ifdef PP_EXPERIMENT_ONE   // compile-time experiments: a single instantiation
    hipLaunchKernelGGL((gemm_bf16_pp_kernel<PP_EXPERIMENT_ONE>), grid, block, 0, stream, *a, w);
    (void)epi;
#else
    if (a->a_kcontig && a->b_kcontig) {
        if (epi == PP_E_BF16) PP_LAUNCH(1, 1, PP_E_BF16);
        else if (epi == PP_E_RES) PP_LAUNCH(1, 1, PP_E_RES);
        else if (epi == PP_E_DACT_MUL) PP_LAUNCH(1, 1, PP_E_DACT_MUL);
        else PP_LAUNCH(1, 1, PP_E_DACT_GELU);
    } else if (a->a_kcontig) {
        if (epi == PP_E_BF16) PP_LAUNCH(1, 0, PP_E_BF16);
        else if (epi == PP_E_BF16_GELU) PP_LAUNCH(1, 0, PP_E_BF16_GELU);
        else if (epi == PP_E_BF16_GELU_D) PP_LAUNCH(1, 0, PP_E_BF16_GELU_D);
        else if (epi == PP_E_RES) PP_LAUNCH(1, 0, PP_E_RES);
        else PP_LAUNCH(1, 0, PP_E_F32);
    } else {
        PP_LAUNCH(0, 0, PP_E_F32);
    }
#endif
#undef PP_LAUNCH
    MD_LAUNCH_CHECK();
    if (tail) {
        const dim3 fgrid((unsigned)(w.tail_units / w.tail_split) * 32u), fblock(256);
#define PP_FIXUP(E) hipLaunchKernelGGL((pp_tail_fixup_kernel<E>), fgrid, fblock, 0, stream, *a, w)
        if (epi == PP_E_BF16) PP_FIXUP(PP_E_BF16);
        else if (epi == PP_E_BF16_GELU) PP_FIXUP(PP_E_BF16_GELU);
        else if (epi == PP_E_BF16_GELU_D) PP_FIXUP(PP_E_BF16_GELU_D);
        else if (epi == PP_E_RES) PP_FIXUP(PP_E_RES);
        else if (epi == PP_E_DACT_MUL) PP_FIXUP(PP_E_DACT_MUL);
        else PP_FIXUP(PP_E_DACT_GELU);
#undef PP_FIXUP
        MD_LAUNCH_CHECK();
    }
    return 0;
}

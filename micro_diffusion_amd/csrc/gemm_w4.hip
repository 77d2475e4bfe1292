// Four-wave persistent bf16 MFMA GEMM ("w4") for the K-contiguous x K-contiguous layout — every nn.Linear FORWARD of the reference
// (micro_diffusion/models/dit.py:84-89,136-139; utils.py:109-111,172-173): activations [tokens, in] x torch weights [out, in].
//
//   C[m, n] = sum_k A[m, k] * B[n, k]          same contract and epilogues as gemm_pp.hip (md_gemm_bf16 picks the kernel)
//
// Why a third kernel family (round 6).  The library yardstick (profiles/r5_gemm_vs_hipblaslt.txt) showed a 4-wave kernel with
// 128 x 128 wave tiles 5-15 % ahead of pp256 on this layout.  What the form buys over the 8-wave ping-pong kernel:
//   * one wave per SIMD owning the whole 512-register file: 16 accumulator blocks of 32 x 32 (256 registers, in the accumulator
//     half of the file) per wave -> 8 fragment reads (ds_read_b128) per 16 MFMAs instead of 6 per 8: a third less LDS traffic;
//   * ONE barrier per k-tile (64 deep) instead of eight: the two wave groups of pp256 hand the matrix pipe to each other 8 times
//     per k-tile, and every hand-over is a barrier pair with a wait in front of it;
//   * operands are staged through REGISTERS (global_load_dwordx4 -> ds_write_b128), not by LDS-DMA: a DMA piece costs its
//     issuing wave 60-185 cycles among MFMAs (MI355X_MICROARCH.md, per-instruction constants) -- hidden in pp256 by the partner
//     wave, fatal with one wave per SIMD -- while a plain load + a 16-byte LDS store are two ordinary fillers of an MFMA gap.
//     The loads of k-tile t + 2 are issued while k-tile t is multiplied (one k-tile of latency cover), written to LDS one
//     k-tile later; hipcc counts the waits (no LDS-DMA anywhere in this kernel, so its vmcnt bookkeeping is exact).
//
// LDS: two k-tile buffers of {A [256 rows][64 k], B [256 rows][64 k]} bf16 = 2 x 64 KiB, rows of 128 bytes with the 16-byte chunk
// index XOR-ed with (row >> 1) & 7 (the same image as pp256's K-contiguous half-tiles: ds_read_b128 conflict-free).
// Wave (wr, wc) owns output rows wr * 128 .., columns wc * 128 .. of the 256 x 256 tile.  MFMA operands are swapped (D = B A^T) so a
// lane owns consecutive columns of one row and the bf16 epilogue is pp256's quad_rows (64-byte runs per quad).
// The workgroup is persistent (PPPlan work list, XCD-blocked like pp256) and the load stream runs across tile boundaries: the
// next tile's first two k-tiles are in flight while the epilogue of the finished tile runs.
//
// Requirements (md_gemm_w4_eligible): both operands K-contiguous, K / ksplit a multiple of 128, N % 8 == 0, no bias, alpha == 1,
// bf16-output epilogues (plain / gated residual / activation derivative), no operand lists, no grouped launch.
#include "gemm_pp_common.h"

namespace {

#ifndef W4_ACC_INC          // (kernel experiments build with another generated schedule: scripts/build_w4_variant.sh)
#define W4_ACC_INC "gemm_w4_acc.inc"
#endif
#include W4_ACC_INC

constexpr int W4_BUF = 32768;      // bytes of one operand of one k-tile buffer: 256 rows x 128 B
constexpr int W4_BREG = 65536;     // B buffers start here

constexpr bool w4_is_dact(int epi) { return epi == PP_E_DACT_GELU || epi == PP_E_DACT_MUL; }



struct W4Tile {
    int m0, n0, batch;
    int mlim, nlim;            // rows / columns of the tile inside the matrix
    char* cbase;               // &C[batch, m0, n0]
    char* c2base;              // &C2[batch, m0, n0] or nullptr
    const char* opbase;        // &res[m0, n0] / &aux[batch, m0, n0]
    const char* gbase;         // &gate[0, n0] or nullptr
};

// Epilogue of one 64 x 32 block pair (two 32-row fragments a0 / a1, 32 columns) whose origin inside the tile is (wrow, wcol).
// bf16(acc) first (= what nn.Linear returns under autocast), then the fused arithmetic on that value, as in gemm_pp.hip.
template <int EPI>
__device__ __forceinline__ void w4_epi_block(const md_gemm_args& p, const PPPlan& w, const W4Tile& et, const float (&a0)[16], const float (&a1)[16],
                                             int wrow, int wcol, int lane_in) {
    int lane = lane_in;
    asm volatile("" : "+v"(lane));               // lane geometry is recomputed per block: nothing of it is hoisted out of the tile loop
    uint4 T[4];
    quad_rows(lane, [&](int i, int g, float (&v)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = i ? a1[4 * g + e] : a0[4 * g + e];
    }, T);
    const int q = lane >> 2;
    const int rq = wrow + (q >> 3) * 32 + (q & 7) * 4;
    const int cq = wcol + (lane & 3) * 8;
    const bool cok = cq < et.nlim;
    const unsigned ldc = (unsigned)p.ldc;
    const unsigned off0 = ((unsigned)rq * ldc + (unsigned)cq) * 2u;
    uint4 gq = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);   // bf16 1.0
    uint4 opv[4];
    if constexpr (EPI == PP_E_RES || w4_is_dact(EPI)) {
        const unsigned ldo = EPI == PP_E_RES ? (unsigned)p.ldr : (unsigned)p.ldaux;
        const int mlast = et.mlim - 1;
        const unsigned coff = (unsigned)(cok ? cq : 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r = rq + t < mlast ? rq + t : mlast;
            opv[t] = *reinterpret_cast<const uint4*>(et.opbase + ((unsigned)r * ldo + coff) * 2u);
        }
        if (EPI == PP_E_RES && et.gbase) {
            int r0 = et.m0 + wrow;                                   // wave-uniform: rows_per_sample % 64 == 0 -> one gate row per block
            r0 = r0 < w.M - 1 ? r0 : w.M - 1;
            const unsigned srow = w.rps_shift >= 0 ? (unsigned)r0 >> w.rps_shift : (unsigned)r0 / (unsigned)p.rows_per_sample;
            gq = *reinterpret_cast<const uint4*>(et.gbase + ((size_t)(srow * (unsigned)p.ldg) + coff) * 2);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const bool ok = cok && rq + t < et.mlim;
        uint4 out = T[t];
        if constexpr (EPI == PP_E_BF16 || EPI == PP_E_RES) {
            if (et.c2base && ok) *reinterpret_cast<uint4*>(et.c2base + ((unsigned)(rq + t) * (unsigned)p.ldc2 + (unsigned)cq) * 2u) = T[t];
        }
        if constexpr (EPI == PP_E_RES) {
            const unsigned lw[4] = {T[t].x, T[t].y, T[t].z, T[t].w}, rw[4] = {opv[t].x, opv[t].y, opv[t].z, opv[t].w}, gw[4] = {gq.x, gq.y, gq.z, gq.w};
            unsigned ow[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const float y0 = __uint_as_float(lw[h] << 16), y1 = __uint_as_float(lw[h] & 0xffff0000u);
                const float r0 = __uint_as_float(rw[h] << 16), r1 = __uint_as_float(rw[h] & 0xffff0000u);
                const float g0 = __uint_as_float(gw[h] << 16), g1 = __uint_as_float(gw[h] & 0xffff0000u);
                ow[h] = cvt_pk_bf16(r0 + g0 * y0, r1 + g1 * y1);
            }
            out = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        } else if constexpr (EPI == PP_E_DACT_MUL) {
            float v[8], ax[8];
            unpack8(T[t], v);
            unpack8(opv[t], ax);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= ax[e];
            out = make_uint4(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7]));
        } else if constexpr (EPI == PP_E_DACT_GELU) {
            float v[8], ax[8];
            unpack8(T[t], v);
            unpack8(opv[t], ax);
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x2 d = dgelu_erf_2(f32x2{ax[e], ax[e + 1]});
                v[e] *= d.x;
                v[e + 1] *= d.y;
            }
            out = make_uint4(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7]));
        }
#ifdef W4_X_NOSTORE      // experiment: everything but the stores (the result is kept alive)
        asm volatile("" : : "v"(out.x), "v"(out.y), "v"(out.z), "v"(out.w));
#else
        if (ok) *reinterpret_cast<uint4*>(et.cbase + (size_t)t * (ldc * 2u) + off0) = out;
#endif
    }
}

template <int EPI, int IP, int J>
__device__ __forceinline__ void w4_epi_pair(const md_gemm_args& p, const PPPlan& w, const W4Tile& et, int wr, int wc, int lane) {
    float a0[16], a1[16];
    w4_acc_read<(2 * IP) * 4 + J>(a0);
    w4_acc_read<(2 * IP + 1) * 4 + J>(a1);
    w4_epi_block<EPI>(p, w, et, a0, a1, wr * 128 + IP * 64, wc * 128 + J * 32, lane);
}

template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(128))) void gemm_bf16_w4_kernel(md_gemm_args p, PPPlan w) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * W4_BREG];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // ---- this workgroup's share of the work list (XCD-blocked, as gemm_pp.hip)
    int w_first, w_stride, w_count;
    {
        const int G = gridDim.x, x = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int q = w.total >> 3, r = w.total & 7;
        const int lo = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        const int cnt = q + (x < r ? 1 : 0);
        w_stride = (G - x + 7) >> 3;
        w_first = lo + j;
        w_count = j < cnt ? (int)((unsigned)(cnt - j + w_stride - 1) / (unsigned)w_stride) : 0;
    }
    if (w_count == 0) return;
    const int pairs = w_count * (w.nk >> 1);                    // loop iterations: two k-tiles each

    // ---- per-lane LDS addresses.  The k-loop touches LDS from inline asm only: lds0 is what keeps `smem` (and with it the
    // kernel's LDS allocation) alive for the compiler.
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;
    unsigned adA[4], adB[4];
    {
        const int ra = wr * 128 + (lane & 31), rb = wc * 128 + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            adA[ks] = lds0 + (unsigned)(ra * 128 + (((((lane >> 5) + 2 * ks) ^ (ra >> 1)) & 7) << 4));
            adB[ks] = lds0 + (unsigned)(W4_BREG + rb * 128 + (((((lane >> 5) + 2 * ks) ^ (rb >> 1)) & 7) << 4));
        }
    }
    // staging writes: piece x (0..7) of an operand = rows x * 32 + tid / 8, logical chunk tid % 8
    const unsigned wrA = lds0 + (unsigned)((tid >> 3) * 128 + ((((tid & 7) ^ ((tid >> 4) & 7))) << 4));
    const unsigned wrB = wrA + W4_BREG;

    // ---- load cursor: runs two k-tiles (one pair) ahead of the multiplications, across tile boundaries.  nk is even, so the
    // cursor changes tiles only at the bottom of the pair loop; inside a k-tile it only steps its scalar byte offset.
    int s_n = 0, s_kt = 0, s_koff = 0;
    u32x4 rA, rB;                                  // wave-uniform buffer descriptors of the cursor's A / B tile (first k element of its split)
    unsigned aofs[8], bofs[8];                     // per-lane byte offsets of this thread's 8 pieces of each operand
    auto stager_open = [&](int n) {
        int m0, n0, batch, split;
        work_decode(w, w_first + n * w_stride, m0, n0, batch, split);
        const int c = (tid & 7) * 8;
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const int row = x * 32 + (tid >> 3);
            int ga = m0 + row, gb = n0 + row;
            ga = (ga < w.M ? ga : w.M - 1) - m0;
            gb = (gb < w.N ? gb : w.N - 1) - n0;
            aofs[x] = (unsigned)(ga * w.lda + c) * 2u;
            bofs[x] = (unsigned)(gb * w.ldb + c) * 2u;
        }
        const int64_t kbeg = (int64_t)split * w.kspan;
        const bf16* pa = reinterpret_cast<const bf16*>(p.A) + (int64_t)batch * p.sA + (int64_t)m0 * w.lda + kbeg;
        const bf16* pb = reinterpret_cast<const bf16*>(p.B) + (int64_t)batch * p.sB + (int64_t)n0 * w.ldb + kbeg;
        // raw buffer: 48-bit base, stride 0, no bounds (rows are clamped above), DATA_FORMAT = 32 bits (0x00020000)
        const uint64_t ua = (uint64_t)(uintptr_t)pa, ub = (uint64_t)(uintptr_t)pb;
        rA = u32x4{(unsigned)ua, (unsigned)(ua >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
        rB = u32x4{(unsigned)ub, (unsigned)(ub >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
        s_koff = 0;
    };
    auto stager_pair_done = [&]() {
        s_kt += 2;
        if (s_kt == w.nk) {
            s_kt = 0;
            if (s_n + 1 < w_count) ++s_n;          // past the end of the list the cursor re-reads the last tile (never multiplied)
            stager_open(s_n);
        }
    };
    // ---- the k-loop is generated inline asm on literal registers (gemm_w4_acc.inc, scripts/gen_w4_acc.py: register plan, schedule
    // and the hand-counted waits are described there); hipcc owns v[0:127] only (amdgpu_num_vgpr on the kernel).
    // prologue: k-tile 0 into buffer 0, k-tile 1 into the staging registers, fragments of k-step 0
    stager_open(0);
    w4_prologue(wrA, wrB, aofs, bofs, rA, rB, 0, BKT * 2);
    s_koff = 2 * BKT * 2;
    stager_pair_done();
    w4_first_reads<0>(adA[0], adB[0]);

    int c_n = 0, c_kt = 0;
    // One k-tile: entering, fragment set 0 holds k-step 0 of this k-tile (buffer BUF; its reads possibly still in flight) and the
    // staging registers hold k-tile t + 1 (its loads possibly still in flight).  During k-steps 0..2 the staging registers are
    // written to the other buffer (free since the barrier of the previous k-tile) and re-loaded with k-tile t + 2; the barrier
    // before k-step 3 publishes them and frees buffer BUF (all of its fragment reads have been issued AND completed: lgkmcnt(0))
    // for the next k-tile's writes; k-step 3 reads the next k-tile's first fragments.
#define W4_KTILE(BUF, FRESH)                                                                                            \
    do {                                                                                                                \
        w4_ks0<BUF, FRESH>(adA[1], adB[1], wrA, aofs, rA, s_koff);                                                      \
        w4_ks1<BUF>(adA[2], adB[2], wrA, wrB, aofs, bofs, rA, rB, s_koff);                                              \
        w4_ks2<BUF>(adA[3], adB[3], wrB, bofs, rB, s_koff);                                                             \
        s_koff += BKT * 2;                                                                                              \
        w4_ks3<BUF>(adA[0], adB[0]);                                                                                    \
    } while (0)

    for (int it = 0; it < pairs; ++it) {
        if (c_kt == 0) W4_KTILE(0, true);            // first k-tile of an output tile: its first k-step takes C = 0
        else W4_KTILE(0, false);
        W4_KTILE(1, false);
        stager_pair_done();
        c_kt += 2;
        if (c_kt == w.nk) {
            // ---- epilogue of the finished tile (the loads of the next tile's first two k-tiles are in flight / in LDS)
            W4Tile et;
            int split;
            work_decode(w, w_first + c_n * w_stride, et.m0, et.n0, et.batch, split);
            et.mlim = w.M - et.m0;
            et.nlim = w.N - et.n0;
            et.cbase = reinterpret_cast<char*>(p.C) + ((int64_t)et.batch * p.sC + (int64_t)et.m0 * p.ldc + et.n0) * 2;
            et.c2base = (!w4_is_dact(EPI) && p.C2) ? reinterpret_cast<char*>(p.C2) + ((int64_t)et.batch * p.sC2 + (int64_t)et.m0 * p.ldc2 + et.n0) * 2 : nullptr;
            et.opbase = EPI == PP_E_RES   ? reinterpret_cast<const char*>(p.res) + ((int64_t)et.m0 * p.ldr + et.n0) * 2
                        : w4_is_dact(EPI) ? reinterpret_cast<const char*>(p.aux) + ((int64_t)et.batch * p.sAux + (int64_t)et.m0 * p.ldaux + et.n0) * 2
                                          : nullptr;
            et.gbase = (EPI == PP_E_RES && p.gate) ? reinterpret_cast<const char*>(p.gate) + (size_t)et.n0 * 2 : nullptr;
            asm volatile("s_nop 15\n\ts_nop 15");      // MFMA result -> v_accvgpr_read wait states (the last MFMA was just issued)
#ifndef W4_X_NOEPI
            w4_epi_pair<EPI, 0, 0>(p, w, et, wr, wc, lane); w4_epi_pair<EPI, 0, 1>(p, w, et, wr, wc, lane);
            w4_epi_pair<EPI, 0, 2>(p, w, et, wr, wc, lane); w4_epi_pair<EPI, 0, 3>(p, w, et, wr, wc, lane);
            w4_epi_pair<EPI, 1, 0>(p, w, et, wr, wc, lane); w4_epi_pair<EPI, 1, 1>(p, w, et, wr, wc, lane);
            w4_epi_pair<EPI, 1, 2>(p, w, et, wr, wc, lane); w4_epi_pair<EPI, 1, 3>(p, w, et, wr, wc, lane);
#endif
            c_kt = 0;
            ++c_n;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the cursor's last (unused) loads
}

}  // namespace

bool md_gemm_w4_eligible(const md_gemm_args* a) {
    if (!a->a_kcontig || !a->b_kcontig) return false;
    const int epi = md_gemm_pp_epi_kind(a);
    if (!(epi == PP_E_BF16 || epi == PP_E_RES || epi == PP_E_DACT_GELU || epi == PP_E_DACT_MUL)) return false;
    if (!md_gemm_pp_eligible(a)) return false;                   // K span, N % 8, leading-dimension ranges, gate rows, ...
    if (a->ksplit != 1 || a->A_list || a->B_list || a->problems || a->timeline) return false;
    if (a->bias || a->alpha != 1.f) return false;                // only the plain form of every epilogue is built
    return true;
}

int md_gemm_w4_launch(const md_gemm_args* a, hipStream_t stream) {
    PPPlan w;
    if (!md_gemm_pp_plan(a, &w)) return MD_BAD_ARG;
    const int cus = (a->cu_limit > 0 && a->cu_limit < NUM_CU) ? a->cu_limit : NUM_CU;
    const unsigned G = (unsigned)(w.total < cus ? w.total : cus);
    if (a->tail_used) *a->tail_used = 0;
    const int epi = md_gemm_pp_epi_kind(a);
    const dim3 grid(G, 1, 1), block(256);
#define W4_LAUNCH(E) hipLaunchKernelGGL((gemm_bf16_w4_kernel<E>), grid, block, 0, stream, *a, w)
    if (epi == PP_E_BF16) W4_LAUNCH(PP_E_BF16);
    else if (epi == PP_E_RES) W4_LAUNCH(PP_E_RES);
    else if (epi == PP_E_DACT_MUL) W4_LAUNCH(PP_E_DACT_MUL);
    else W4_LAUNCH(PP_E_DACT_GELU);
#undef W4_LAUNCH
    MD_LAUNCH_CHECK();
    return 0;
}

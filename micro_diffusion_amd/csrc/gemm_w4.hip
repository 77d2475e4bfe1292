// Four-wave persistent bf16 MFMA GEMM ("w4") for the K-contiguous x K-contiguous layout — every nn.Linear FORWARD of the reference
// (micro_diffusion/models/dit.py:84-89,136-139; utils.py:109-111,172-173): activations [tokens, in] x torch weights [out, in].
//
//   C[m, n] = sum_k A[m, k] * B[n, k]          same contract and epilogues as gemm_pp.hip (md_gemm_bf16 picks the kernel)
//
// Why a third kernel family (round 6).  The library yardstick (profiles/r5_gemm_vs_hipblaslt.txt) showed a 4-wave kernel with
// 128 x 128 wave tiles on 16 x 16 MFMAs 5-15 % ahead of pp256 on this layout.  What the form buys over the 8-wave ping-pong kernel:
//   * one wave per SIMD owning the whole 512-register file: 64 accumulator blocks of 16 x 16 (256 registers, in the accumulator
//     half of the file) per wave -> a third less LDS fragment traffic per MFMA flop than pp256's 128 x 64 wave tiles;
//   * v_mfma_f32_16x16x32_bf16 instead of 32x32x16: these kernels are POWER-bound (shader clock 1.42 GHz of 2.4 at 8192^3 with the
//     matrix pipes 80 % busy, profiles/r6_w4_v1_experiments.txt) and the 16 x 16 form moves half the accumulator bytes per flop:
//     +7 % on the whole instruction stream, +13 % MFMA-only;
//   * ONE barrier per k-tile (64 deep) instead of eight;
//   * operands are staged through REGISTERS (buffer_load_dwordx4 -> ds_write_b128), not by LDS-DMA: a DMA piece costs its
//     issuing wave 60-185 cycles among MFMAs (MI355X_MICROARCH.md, per-instruction constants) -- hidden in pp256 by the partner
//     wave, fatal with one wave per SIMD.  The loads of k-tile t + 2 are issued while k-tile t is multiplied;
//   * the epilogue goes through a private 8 KiB LDS slab per wave (the 32 KiB the two k-tile buffers leave of the 160): every
//     global store of the kernel is 16 lanes x 16 bytes = 256 contiguous bytes of one output row (pp256: 64-byte runs).
// The k-loop is generated inline asm on literal registers (gemm_w4_acc.inc <- scripts/gen_w4_acc.py: register plan, schedule and the
// hand-counted waits are documented there); hipcc owns v[0:95] (amdgpu_num_vgpr), the prologue / tile bookkeeping / epilogue.
//
// LDS: two k-tile buffers of {A [256 rows][64 k], B [256 rows][64 k]} bf16 = 2 x 64 KiB, rows of 128 bytes with the 16-byte chunk
// index XOR-ed with (row >> 1) & 7 (pp256's K-contiguous image: ds_read_b128 conflict-free for the 16-row fragments too), then
// the four epilogue slabs.  Wave (wr, wc) owns output rows wr * 128 .., columns wc * 128 .. of the 256 x 256 tile.
// The workgroup is persistent (PPPlan work list, XCD-blocked like pp256) and the load stream runs across tile boundaries: the
// next tile's first two k-tiles are in LDS / in flight while the epilogue of the finished tile runs.
//
// Requirements (md_gemm_w4_eligible): both operands K-contiguous, K / ksplit a multiple of 128, N % 8 == 0, no bias, alpha == 1,
// bf16-output epilogues (plain / gated residual / activation derivative), no operand lists, no grouped launch.
#include "gemm_pp_common.h"

namespace {

// Per-lane addresses the generated k-loop takes as asm operands (all in hipcc's registers).
struct W4Addr {
    unsigned adA[2];     // A fragment reads: [k-step]; row lane % 16 (+ 16 i: immediate), 16-byte chunk (4 ks + lane / 16) ^ swizzle
    unsigned adB[8];     // B K-contiguous: [k-step] as A ([2..7] unused).  B K-strided: [column fragment j], the k-step is an immediate
    unsigned wrA;        // staging writes of the A pieces (piece x: + x * 4096)
    unsigned wrB[2];     // ... of the B pieces; K-strided B: [piece parity] (the k-row swizzle differs)
    unsigned aofs[8], bofs[8];   // global byte offsets of this thread's 8 pieces of each operand, relative to the tile's descriptor
};

#ifndef W4_ACC_INC          // (kernel experiments build with another generated schedule: scripts/build_w4_variant.sh)
#define W4_ACC_INC "gemm_w4_acc.inc"
#endif
#include W4_ACC_INC

constexpr int W4_BUF = 32768;      // bytes of one operand of one k-tile buffer: 256 rows x 128 B
constexpr int W4_BREG = 65536;     // B buffers start here
constexpr int W4_SLAB = 8192;      // epilogue slab of one wave (behind the k-tile buffers)

constexpr bool w4_is_dact(int epi) { return epi == PP_E_DACT_GELU || epi == PP_E_DACT_MUL; }



// LDS slab accesses of the epilogue: 8-byte stores and 16-byte loads of the SAME bytes -- through may_alias types, or type-based
// alias analysis lets hipcc move a row group's loads across the stores of the group that reuses its slab half.
typedef unsigned w4_u32x2 __attribute__((ext_vector_type(2), may_alias));
typedef unsigned w4_u32x4 __attribute__((ext_vector_type(4), may_alias));

struct W4Tile {
    int m0, n0, batch;
    int mlim, nlim;            // rows / columns of the tile inside the matrix
    char* cbase;               // &C[batch, m0, n0]
    char* c2base;              // &C2[batch, m0, n0] or nullptr
    const char* opbase;        // &res[m0, n0] / &aux[batch, m0, n0]
    const char* gbase;         // &gate[0, n0] or nullptr
};

// Epilogue of rows 16 I .. 16 I + 15 of the wave's 128 x 128 tile (origin (wrow, wcol) inside the 256 x 256 tile).
// Out of the MFMAs a lane owns 4 consecutive columns of a row per 16 x 16 block; the row group goes through one 4 KiB half of the
// wave's private LDS slab (two halves alternate, so the writes of group I + 1 do not wait for the reads of group I):
//   write: bf16 pairs, 8 bytes at  row * 256 + ((32 j + 8 g) ^ (row << 4))     (row = lane % 16, g = lane / 16; 2-way store conflicts)
//   read : 16 bytes (8 columns) at rr * 256 + ((16 c) ^ (rr << 4)),  rr = 4 t + lane / 16, c = lane % 16   (conflict-free)
// so lane (rr, c) ends with columns 8 c .. 8 c + 7 of row rr and a store instruction writes 4 rows x 256 contiguous bytes.
// bf16(acc) first (= what nn.Linear returns under autocast), then the fused arithmetic on that value, as in gemm_pp.hip.
// WAIT_LOADS (the first row group): every VM load issued so far -- the k-loop's staging loads, this group's operand loads -- is
// waited for before the first store is issued; the next k-tile's LDS writes then need no vmcnt (gen_w4_acc.py, FRESH).
// Operands of the fused epilogues for row group I, in the store-side layout (lane = 8 columns of row 4 t + lane / 16).  ONE register
// set, refreshed in place: as soon as step t of group I has consumed its piece, the same registers request piece t of group I + 1
// (a whole row group of latency cover without a second set -- hipcc's 92 registers do not hold two).
struct W4Ops {
    uint4 v[4];
    uint4 gq;
};
template <int EPI, int I>
__device__ __forceinline__ void w4_epi_ops(const md_gemm_args& p, const PPPlan& w, const W4Tile& et, int wrow, int wcol, int lane_in, W4Ops& o) {
    if constexpr (EPI == PP_E_RES || w4_is_dact(EPI)) {
        int lane = lane_in;
        asm volatile("" : "+v"(lane));
        const int r = lane & 15, g = lane >> 4;
        const int c8 = wcol + r * 8;
        const unsigned ldo = EPI == PP_E_RES ? (unsigned)p.ldr : (unsigned)p.ldaux;
        const int mlast = et.mlim - 1;
        const unsigned coff = (unsigned)(c8 < et.nlim ? c8 : 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = wrow + 16 * I + 4 * t + g;
            o.v[t] = *reinterpret_cast<const uint4*>(et.opbase + ((unsigned)(row < mlast ? row : mlast) * ldo + coff) * 2u);
        }
        o.gq = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);   // bf16 1.0
        if (EPI == PP_E_RES && et.gbase) {
            int r0 = et.m0 + wrow + 16 * I;                           // wave-uniform: rows_per_sample % 64 == 0 -> one gate row per group
            r0 = r0 < w.M - 1 ? r0 : w.M - 1;
            const unsigned srow = w.rps_shift >= 0 ? (unsigned)r0 >> w.rps_shift : (unsigned)r0 / (unsigned)p.rows_per_sample;
            o.gq = *reinterpret_cast<const uint4*>(et.gbase + ((size_t)(srow * (unsigned)p.ldg) + coff) * 2);
        }
    }
}

template <int EPI, int I, bool WAIT_LOADS>
__device__ __forceinline__ void w4_epi_rows(const md_gemm_args& p, const PPPlan& w, const W4Tile& et, unsigned char* slab, int wrow, int wcol, int lane_in,
                                            W4Ops& ops) {
    int lane = lane_in;
    asm volatile("" : "+v"(lane));               // lane geometry is recomputed per row group: nothing of it is hoisted out of the tile loop
    const int r = lane & 15, g = lane >> 4;
    unsigned char* half = slab + (I & 1) * 4096;
    const int c8 = wcol + r * 8;                                       // first column of this lane's 8 (tile-relative)
    const bool cok = c8 < et.nlim;
    uint4 (&opv)[4] = ops.v;
    const uint4 gq = ops.gq;
#pragma unroll
    for (int jh = 0; jh < 2; ++jh) {
        float a[16];
        if (jh == 0) w4_acc_read16<I, 0>(a);
        else w4_acc_read16<I, 1>(a);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const w4_u32x2 v = {cvt_pk_bf16(a[4 * q], a[4 * q + 1]), cvt_pk_bf16(a[4 * q + 2], a[4 * q + 3])};
            *reinterpret_cast<w4_u32x2*>(half + r * 256 + (((32 * (4 * jh + q)) + 8 * g) ^ (r << 4))) = v;
        }
    }
    if constexpr (WAIT_LOADS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned ldc = (unsigned)p.ldc;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int rr = 4 * t + g;
        const w4_u32x4 Tv = *reinterpret_cast<const w4_u32x4*>(half + rr * 256 + ((16 * r) ^ (rr << 4)));
        const uint4 T = make_uint4(Tv.x, Tv.y, Tv.z, Tv.w);
        const int row = wrow + 16 * I + rr;
        const bool ok = cok && row < et.mlim;
        uint4 out = T;
        if constexpr (EPI == PP_E_BF16 || EPI == PP_E_RES || EPI == PP_E_BF16_GELU) {
            if (et.c2base && ok) *reinterpret_cast<uint4*>(et.c2base + ((unsigned)row * (unsigned)p.ldc2 + (unsigned)c8) * 2u) = T;
        }
        if constexpr (EPI == PP_E_BF16_GELU) {
            float v[8];
            unpack8(T, v);
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x2 g2 = gelu_erf_2(f32x2{v[e], v[e + 1]});
                v[e] = g2.x;
                v[e + 1] = g2.y;
            }
            out = make_uint4(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7]));
        } else if constexpr (EPI == PP_E_BF16_GELU_D) {      // C = gelu, C2 = gelu' (the backward multiplies by it: md_gemm_args.dact_cached)
            float v[8], dv[8];
            unpack8(T, v);
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
                *reinterpret_cast<uint4*>(et.c2base + ((unsigned)row * (unsigned)p.ldc2 + (unsigned)c8) * 2u) =
                    make_uint4(cvt_pk_bf16(dv[0], dv[1]), cvt_pk_bf16(dv[2], dv[3]), cvt_pk_bf16(dv[4], dv[5]), cvt_pk_bf16(dv[6], dv[7]));
        } else if constexpr (EPI == PP_E_RES) {
            const unsigned lw[4] = {T.x, T.y, T.z, T.w}, rw[4] = {opv[t].x, opv[t].y, opv[t].z, opv[t].w}, gw[4] = {gq.x, gq.y, gq.z, gq.w};
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
            unpack8(T, v);
            unpack8(opv[t], ax);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= ax[e];
            out = make_uint4(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7]));
        } else if constexpr (EPI == PP_E_DACT_GELU) {
            float v[8], ax[8];
            unpack8(T, v);
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
        if (ok) *reinterpret_cast<uint4*>(et.cbase + ((unsigned)row * ldc + (unsigned)c8) * 2u) = out;
#endif
        if constexpr ((EPI == PP_E_RES || w4_is_dact(EPI)) && I < 7) {     // this piece is consumed: request the next row group's
            const unsigned ldo = EPI == PP_E_RES ? (unsigned)p.ldr : (unsigned)p.ldaux;
            const int mlast = et.mlim - 1, nrow = row + 16;
            opv[t] = *reinterpret_cast<const uint4*>(et.opbase + ((unsigned)(nrow < mlast ? nrow : mlast) * ldo + (unsigned)(cok ? c8 : 0)) * 2u);
        }
    }
    if constexpr (EPI == PP_E_RES && I < 7) {
        if (et.gbase) {
            int r0 = et.m0 + wrow + 16 * (I + 1);
            r0 = r0 < w.M - 1 ? r0 : w.M - 1;
            const unsigned srow = w.rps_shift >= 0 ? (unsigned)r0 >> w.rps_shift : (unsigned)r0 / (unsigned)p.rows_per_sample;
            ops.gq = *reinterpret_cast<const uint4*>(et.gbase + ((size_t)(srow * (unsigned)p.ldg) + (unsigned)(cok ? c8 : 0)) * 2);
        }
    }
}

// amdgpu_num_vgpr(92), not 96: when hipcc spills SGPRs (the gated-residual instantiations do) it takes the last allowed VGPR for the
// spill lanes and was seen to hand out the two registers ABOVE the limit as ordinary temporaries -- v96 / v97, the first staging
// register.  v[92:95] are the guard band; scripts/check_w4_asm.py (tests/test_build_static.py) audits every build.
template <int BKC, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(92))) void gemm_bf16_w4_kernel(md_gemm_args p, PPPlan w) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * W4_BREG + 4 * W4_SLAB];   // 160 KiB: the whole LDS of a CU
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
    W4Addr ad;
    {
        const int ra = wr * 128 + (lane & 15), rb = wc * 128 + (lane & 15);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            ad.adA[ks] = lds0 + (unsigned)(ra * 128 + ((((4 * ks + (lane >> 4)) ^ (ra >> 1)) & 7) << 4));
            if (BKC) ad.adB[ks] = lds0 + (unsigned)(W4_BREG + rb * 128 + ((((4 * ks + (lane >> 4)) ^ (rb >> 1)) & 7) << 4));
        }
        if (!BKC) {
            // K-strided B image: [64 k-rows][32 chunks of 16 B], physical chunk = chunk ^ swz(k), swz(k) = (k & 3) << 1 | ((k >> 3) & 1) << 3
            // (the 8 k-rows x 32 bytes a half-wave's transposing reads touch land on 16 different 16-byte bank positions).
            // Lane (g = lane / 16, li = lane % 16) of fragment j reads 4 columns (8 bytes) at k-row 8 g + li / 4 (+ 4: second read, + 32 ks).
            const int li = lane & 15, g = lane >> 4;
            const int kk = 8 * g + (li >> 2);
            const int swz = ((kk & 3) << 1) | (((kk >> 3) & 1) << 3);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int col = wc * 128 + 16 * j + 4 * (li & 3);
                ad.adB[j] = lds0 + (unsigned)(W4_BREG + kk * 512 + ((((col >> 3) ^ swz) & 31) << 4) + ((col >> 2) & 1) * 8);
            }
        } else {
#pragma unroll
            for (int j = 2; j < 8; ++j) ad.adB[j] = 0;
        }
    }
    // staging writes.  K-contiguous operand: piece x (0..7) = rows x * 32 + tid / 8, logical chunk tid % 8.
    // K-strided operand: piece x = k-rows x * 8 + tid / 32, logical chunk tid % 32 (a wave loads two whole 512-byte k-rows).
    ad.wrA = lds0 + (unsigned)((tid >> 3) * 128 + ((((tid & 7) ^ ((tid >> 4) & 7))) << 4));
    if (BKC) {
        ad.wrB[0] = ad.wrA + W4_BREG;
        ad.wrB[1] = ad.wrB[0];
    } else {
        const int t5 = tid >> 5, c = tid & 31;
#pragma unroll
        for (int par = 0; par < 2; ++par)          // k-row = 8 x + t5: k & 3 = t5 & 3, (k >> 3) & 1 = x & 1
            ad.wrB[par] = lds0 + (unsigned)(W4_BREG + t5 * 512 + ((c ^ (((t5 & 3) << 1) | (par << 3))) << 4));
    }

    // ---- load cursor: runs two k-tiles (one pair) ahead of the multiplications, across tile boundaries.  nk is even, so the
    // cursor changes tiles only at the bottom of the pair loop; inside a k-tile it only steps its scalar byte offset.
    int s_n = 0, s_kt = 0, s_koffA = 0, s_koffB = 0;
    const int kstepB = BKC ? BKT * 2 : BKT * w.ldb * 2;       // bytes per k-tile of the B tile (K-strided: 64 rows of ldb elements)
    u32x4 rA, rB;                                  // wave-uniform buffer descriptors of the cursor's A / B tile (first k element of its split)
    auto stager_open = [&](int n) {
        int m0, n0, batch, split;
        work_decode(w, w_first + n * w_stride, m0, n0, batch, split);
        const int c = (tid & 7) * 8;
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const int row = x * 32 + (tid >> 3);
            int ga = m0 + row;
            ga = (ga < w.M ? ga : w.M - 1) - m0;
            ad.aofs[x] = (unsigned)(ga * w.lda + c) * 2u;
            if (BKC) {
                int gb = n0 + row;
                gb = (gb < w.N ? gb : w.N - 1) - n0;
                ad.bofs[x] = (unsigned)(gb * w.ldb + c) * 2u;
            } else {                                       // k-row 8 x + tid / 32, columns 8 (tid % 32) .. (clamped to the last whole chunk)
                int gc = n0 + (tid & 31) * 8;
                const int last = (w.N - 1) & ~7;
                gc = (gc < last ? gc : last) - n0;
                ad.bofs[x] = (unsigned)((x * 8 + (tid >> 5)) * w.ldb + gc) * 2u;
            }
        }
        const int64_t kbeg = (int64_t)split * w.kspan;
        const bf16* pa = reinterpret_cast<const bf16*>(p.A) + (int64_t)batch * p.sA + (int64_t)m0 * w.lda + kbeg;
        const bf16* pb = reinterpret_cast<const bf16*>(p.B) + (int64_t)batch * p.sB + (BKC ? (int64_t)n0 * w.ldb + kbeg : kbeg * w.ldb + n0);
        // raw buffer: 48-bit base, stride 0, no bounds (rows are clamped above), DATA_FORMAT = 32 bits (0x00020000)
        const uint64_t ua = (uint64_t)(uintptr_t)pa, ub = (uint64_t)(uintptr_t)pb;
        rA = u32x4{(unsigned)ua, (unsigned)(ua >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
        rB = u32x4{(unsigned)ub, (unsigned)(ub >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
        s_koffA = 0;
        s_koffB = 0;
    };
    auto stager_pair_done = [&]() {
        s_kt += 2;
        if (s_kt == w.nk) {
            s_kt = 0;
            if (s_n + 1 < w_count) ++s_n;          // past the end of the list the cursor re-reads the last tile (never multiplied)
            stager_open(s_n);
        }
    };
    // ---- the k-loop is generated inline asm on literal registers (gemm_w4_acc.inc, scripts/gen_w4_acc.py).
    // prologue: k-tile 0 into buffer 0, k-tile 1 into the staging registers (landed), fragments of k-step 0
    stager_open(0);
    w4_prologue<BKC>(ad, rA, rB, 0, 0, BKT * 2, kstepB);
    s_koffA = 2 * BKT * 2;
    s_koffB = 2 * kstepB;
    stager_pair_done();
    w4_first_reads<BKC>(ad);

    int c_n = 0, c_kt = 0;
    // One k-tile: entering, A set 0 / B slot 0 hold k-step 0 of this k-tile (buffer BUF; the reads possibly still in flight) and the
    // staging registers hold k-tile t + 1.  During H0..H2 the staging registers are written to the other buffer (free since the
    // barrier of the previous k-tile) and re-loaded with k-tile t + 2; the barrier in front of H3 publishes them and frees buffer
    // BUF for the next k-tile's writes; H3 reads the next k-tile's first fragments.  FRESH = first k-tile of an output tile.
#define W4_KTILE(BUF, FRESH)                                                                                            \
    do {                                                                                                                \
        w4_h0<BKC, BUF, FRESH>(ad, rA, rB, s_koffA, s_koffB);                                                           \
        w4_h1<BKC, BUF, FRESH>(ad, rA, rB, s_koffA, s_koffB);                                                           \
        w4_h2<BKC, BUF, FRESH>(ad, rA, rB, s_koffA, s_koffB);                                                           \
        s_koffA += BKT * 2;                                                                                             \
        s_koffB += kstepB;                                                                                              \
        w4_h3<BKC, BUF>(ad);                                                                                            \
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
            unsigned char* const slab = smem + 2 * W4_BREG + wave * W4_SLAB;
            const int wrow = wr * 128, wcol = wc * 128;
            W4Ops ops;
            w4_epi_ops<EPI, 0>(p, w, et, wrow, wcol, lane, ops);
            w4_epi_rows<EPI, 0, true>(p, w, et, slab, wrow, wcol, lane, ops);  w4_epi_rows<EPI, 1, false>(p, w, et, slab, wrow, wcol, lane, ops);
            w4_epi_rows<EPI, 2, false>(p, w, et, slab, wrow, wcol, lane, ops); w4_epi_rows<EPI, 3, false>(p, w, et, slab, wrow, wcol, lane, ops);
            w4_epi_rows<EPI, 4, false>(p, w, et, slab, wrow, wcol, lane, ops); w4_epi_rows<EPI, 5, false>(p, w, et, slab, wrow, wcol, lane, ops);
            w4_epi_rows<EPI, 6, false>(p, w, et, slab, wrow, wcol, lane, ops); w4_epi_rows<EPI, 7, false>(p, w, et, slab, wrow, wcol, lane, ops);
#else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            c_kt = 0;
            ++c_n;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the cursor's last (unused) loads
}

}  // namespace

// Instantiated: NT (nn.Linear forward) bf16 / gated residual / activation derivative; NN (dgrads, the MoE's [E, in, out] experts) bf16 /
// GELU(erf) with the raw copy or the cached derivative / residual.
static bool w4_instantiated(int bkc, int epi) {
    if (bkc) return epi == PP_E_BF16 || epi == PP_E_RES || epi == PP_E_DACT_GELU || epi == PP_E_DACT_MUL;
    return epi == PP_E_BF16 || epi == PP_E_BF16_GELU || epi == PP_E_BF16_GELU_D || epi == PP_E_RES;
}

bool md_gemm_w4_eligible(const md_gemm_args* a) {
    if (!a->a_kcontig) return false;
    const int epi = md_gemm_pp_epi_kind(a);
    if (epi < 0 || !w4_instantiated(a->b_kcontig, epi)) return false;
    if (!md_gemm_pp_eligible(a)) return false;                   // K span, N % 8, leading-dimension ranges, gate rows, ...
    if (a->ksplit != 1 || a->A_list || a->B_list || a->problems || a->timeline) return false;
    if (a->bias || a->alpha != 1.f) return false;                // only the plain form of every epilogue is built
    if (!a->b_kcontig && (a->K * a->ldb >= (int64_t)1 << 30)) return false;   // 32-bit scalar byte offset of the K-strided cursor
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
#define W4_LAUNCH(BK, E) hipLaunchKernelGGL((gemm_bf16_w4_kernel<BK, E>), grid, block, 0, stream, *a, w)
    if (a->b_kcontig) {
        if (epi == PP_E_BF16) W4_LAUNCH(1, PP_E_BF16);
        else if (epi == PP_E_RES) W4_LAUNCH(1, PP_E_RES);
        else if (epi == PP_E_DACT_MUL) W4_LAUNCH(1, PP_E_DACT_MUL);
        else W4_LAUNCH(1, PP_E_DACT_GELU);
    } else {
        if (epi == PP_E_BF16) W4_LAUNCH(0, PP_E_BF16);
        else if (epi == PP_E_BF16_GELU) W4_LAUNCH(0, PP_E_BF16_GELU);
        else if (epi == PP_E_BF16_GELU_D) W4_LAUNCH(0, PP_E_BF16_GELU_D);
        else W4_LAUNCH(0, PP_E_RES);
    }
#undef W4_LAUNCH
    MD_LAUNCH_CHECK();
    return 0;
}

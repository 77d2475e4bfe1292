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
//   * TWO barriers per k-tile (64 deep) instead of eight;
//   * operands are staged by LDS-DMA (buffer_load_dwordx4 ... offen lds: 16 per wave and k-tile, spread one per ~6 MFMAs, the loads
//     of k-tile t + 2 issued while k-tile t is multiplied) and ALL 16 fragments of a k-step are resident (128 registers), so the
//     k-tile's LDS buffer is free for the next DMA an eighth into the k-tile.  (The round's first form staged through 64 registers,
//     buffer_load -> ds_write_b128, on the guide's warning that a DMA piece costs its issuing wave 60-185 cycles; at this density it
//     does not -- the DMA form is +4 % on hot operands, +0.7-1.0 % on the step: profiles/r6_w4_lds_dma.txt.  It is also what the
//     library's own NT kernel does.);
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
    unsigned adA[8];     // A K-contiguous: fragment reads [k-step] ([2..7] unused): row lane % 16 (+ 16 i: immediate), 16-byte chunk (4 ks + lane / 16) ^ swizzle.
                         // A K-strided (weight gradients): [row fragment i], as B K-strided
    unsigned adB[8];     // B K-contiguous: [k-step] as A ([2..7] unused).  B K-strided: [column fragment j], the k-step is an immediate
    unsigned aofs[8], bofs[8];   // global byte offsets of this thread's 8 pieces of each operand, relative to the tile's descriptor
    unsigned ldsw;       // LDS address of this wave's 1 KiB inside a 4 KiB piece (M0 of its DMA loads = ldsw + buffer + 4096 x: immediates)
};

#ifndef W4_ACC_INC          // (kernel experiments build with another generated schedule: scripts/build_w4_variant.sh)
#define W4_ACC_INC "gemm_w4_acc.inc"
#endif
#include W4_ACC_INC

constexpr int W4_BUF = 32768;      // bytes of one operand of one k-tile buffer: 256 rows x 128 B
constexpr int W4_BREG = 65536;     // B buffers start here
constexpr int W4_SLAB = 8192;      // epilogue slab of one wave (behind the k-tile buffers)

constexpr bool w4_is_dact(int epi) { return epi == PP_E_DACT_GELU || epi == PP_E_DACT_MUL || epi == PP_E_DACT_SWIGLU; }



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

// ---------------------------------------------------------------------------------------------------------------------
// Epilogue.  Out of the MFMAs a lane owns 4 consecutive columns of a row per 16 x 16 block; a row group (16 rows x the wave's 128
// columns) goes through one 4 KiB half of the wave's private LDS slab:
//   write: bf16 pairs, 8 bytes at  row * 256 + ((32 j + 8 g) ^ (row << 4))     (row = lane % 16, g = lane / 16; 2-way store conflicts)
//   read : 16 bytes (8 columns) at rr * 256 + ((16 c) ^ (rr << 4)),  rr = 4 t + lane / 16, c = lane % 16   (conflict-free)
// so lane (rr, c) ends with columns 8 c .. 8 c + 7 of row rr and a store instruction writes 4 rows x 256 contiguous bytes.
// The groups are software-pipelined over the two slab halves: group I's four reads are issued, group I + 1 is converted and written
// to the other half while they fly, then group I is finished and stored (the first version waited for every read separately: 32
// serialized LDS round trips per tile).  Per-lane offsets are formed once per tile (W4EpiLane); rows advance by scalar arithmetic.
// bf16(acc) first (= what nn.Linear returns under autocast), then the fused arithmetic on that value, as in gemm_pp.hip.
// Before the FIRST store of a tile every VM load issued so far (the k-loop's DMA loads, the first operand requests) is waited
// for: the next k-tile's LDS writes then need no vmcnt (gen_w4_acc.py, FRESH) and no store ever stands between a load and its wait.
// ---------------------------------------------------------------------------------------------------------------------
struct W4EpiLane {
    unsigned wr[8];      // slab write offsets of the lane's 8-byte piece of column fragment j (without the half)
    unsigned rd[4];      // slab read offsets of step t
    unsigned oc, oc2, oo, og;   // global byte offsets of (row lane / 16, column 8 (lane % 16)) under ldc / ldc2 / the operand's ld; gate: column only
    int rg, c8;          // lane / 16 and the first of the lane's 8 columns, both wave-tile-relative (predicates of ragged tiles)
};
// Operands of the fused epilogues for one row group, in the store-side layout.  ONE register set, refreshed in place: as soon as step
// t of group I has consumed its piece, the same registers request piece t of group I + 1.
struct W4Ops {
    uint4 v[4];
    uint4 gq;
    uint4 v2[4];          // PP_E_DACT_SWIGLU: h2 (aux columns N + n) beside h1 in v (unused members cost the other kinds nothing)
};

template <int EPI>
__device__ __forceinline__ void w4_epi_lane(const md_gemm_args& p, const W4Tile& et, int wcol, int lane_in, W4EpiLane& L) {
    int lane = lane_in;
    asm volatile("" : "+v"(lane));               // formed per tile: nothing of it lives across the k-loop
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) L.wr[j] = (unsigned)(r * 256 + ((32 * j + 8 * g) ^ (r << 4)));
#pragma unroll
    for (int t = 0; t < 4; ++t) L.rd[t] = (unsigned)((4 * t + g) * 256 + ((16 * r) ^ ((4 * t + g) << 4)));
    L.rg = g;
    L.c8 = r * 8;
    const int cc = wcol + r * 8 < et.nlim ? r * 8 : 0;           // (ragged tiles: a column inside the matrix for the operand requests)
    L.oc = (unsigned)(g * (int)p.ldc + r * 8) * 2u;
    L.oc2 = (unsigned)(g * (int)p.ldc2 + r * 8) * 2u;
    L.oo = (unsigned)(g * (int)(EPI == PP_E_RES ? p.ldr : p.ldaux) + cc) * 2u;
    L.og = (unsigned)cc * 2u;
}

// Wave-uniform row cursors of the epilogue (SGPR pairs, stepped by 4 rows): first byte of the wave tile's current 4-row step in C, C2
// and the operand matrix.  (Formed per step from the tile base, the 32 x 3 pointers were all kept live: 130-220 SGPR spills.)
struct W4Rows {
    char* c;
    char* c2;
    const char* op;       // the NEXT row group's step t (the operands are requested one group ahead)
    int64_t sc, sc2, so;  // bytes per 4 rows
};

// request the fused-epilogue operands of row group I (rows clamped into the matrix)
template <int EPI, bool INTERIOR>
__device__ __forceinline__ void w4_epi_req(const md_gemm_args& p, const W4Tile& et, const W4EpiLane& L, const char* oprow, int row0, int wcol, uint4& dst) {
    if (INTERIOR) {
        dst = *reinterpret_cast<const uint4*>(oprow + L.oo);       // oprow = &op[wave tile row row0, wave tile column 0]
    } else if constexpr (EPI == PP_E_DACT_SWIGLU) {
        dst = make_uint4(0u, 0u, 0u, 0u);                             // (interior tiles only: md_gemm_pp_shape_ok)
    } else {                                                          // ragged tile: every lane's row clamped into the matrix
        const int64_t ldo = EPI == PP_E_RES ? p.ldr : p.ldaux;
        const int row = row0 + L.rg < et.mlim - 1 ? row0 + L.rg : et.mlim - 1;
        dst = *reinterpret_cast<const uint4*>(et.opbase + ((int64_t)row * ldo + wcol) * 2 + L.og);
    }
}
template <int EPI, int I>
__device__ __forceinline__ void w4_epi_gate(const md_gemm_args& p, const PPPlan& w, const W4Tile& et, const W4EpiLane& L, int wrow, int wcol, uint4& gq) {
    gq = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);   // bf16 1.0
    if (et.gbase) {
        int r0 = et.m0 + wrow + 16 * I;                               // wave-uniform: rows_per_sample % 64 == 0 -> one gate row per group
        r0 = r0 < w.M - 1 ? r0 : w.M - 1;
        const unsigned srow = w.rps_shift >= 0 ? (unsigned)r0 >> w.rps_shift : (unsigned)r0 / (unsigned)p.rows_per_sample;
        gq = *reinterpret_cast<const uint4*>(et.gbase + ((int64_t)srow * p.ldg + wcol) * 2 + L.og);
    }
}

// convert row group I and write it to its slab half
template <int I>
__device__ __forceinline__ void w4_epi_write(unsigned char* slab, const W4EpiLane& L) {
    unsigned char* half = slab + (I & 1) * 4096;
#pragma unroll
    for (int jh = 0; jh < 2; ++jh) {
        float a[16];
        if (jh == 0) w4_acc_read16<I, 0>(a);
        else w4_acc_read16<I, 1>(a);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const w4_u32x2 v = {cvt_pk_bf16(a[4 * q], a[4 * q + 1]), cvt_pk_bf16(a[4 * q + 2], a[4 * q + 3])};
            *reinterpret_cast<w4_u32x2*>(half + L.wr[4 * jh + q]) = v;
        }
    }
}

// finish row group I: T[t] = its four slab reads (issued before group I + 1 was written); fused arithmetic; stores
template <int EPI, int I, bool INTERIOR>
__device__ __forceinline__ void w4_epi_store(const md_gemm_args& p, const PPPlan& w, const W4Tile& et, const W4EpiLane& L, int wrow, int wcol,
                                             const w4_u32x4 (&Tv)[4], W4Ops& ops, W4Rows& R) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint4 T = make_uint4(Tv[t].x, Tv[t].y, Tv[t].z, Tv[t].w);
        const int row0 = wrow + 16 * I + 4 * t;                      // wave-uniform: first row of the step's four
        const bool ok = INTERIOR || (wcol + L.c8 < et.nlim && row0 + L.rg < et.mlim);
        char* const crow = R.c;
        uint4 out = T;
        if constexpr (EPI == PP_E_BF16 || EPI == PP_E_RES || EPI == PP_E_BF16_GELU) {
            if (et.c2base && ok) *reinterpret_cast<uint4*>(R.c2 + L.oc2) = T;
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
                *reinterpret_cast<uint4*>(R.c2 + L.oc2) =
                    make_uint4(cvt_pk_bf16(dv[0], dv[1]), cvt_pk_bf16(dv[2], dv[3]), cvt_pk_bf16(dv[4], dv[5]), cvt_pk_bf16(dv[6], dv[7]));
        } else if constexpr (EPI == PP_E_RES) {
            const uint4 gq = ops.gq;
            const unsigned lw[4] = {T.x, T.y, T.z, T.w}, rw[4] = {ops.v[t].x, ops.v[t].y, ops.v[t].z, ops.v[t].w}, gw[4] = {gq.x, gq.y, gq.z, gq.w};
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
            unpack8(ops.v[t], ax);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= ax[e];
            out = make_uint4(cvt_pk_bf16(v[0], v[1]), cvt_pk_bf16(v[2], v[3]), cvt_pk_bf16(v[4], v[5]), cvt_pk_bf16(v[6], v[7]));
        } else if constexpr (EPI == PP_E_DACT_SWIGLU) {
            // da = bf16(acc); dh1 = da * h2 * silu'(h1) -> C[:, n], dh2 = da * silu(h1) -> C[:, N + n]   (elementwise.hip: swiglu_bwd8)
            float g[8], x1[8], x2[8], o2[8];
            unpack8(T, g);
            unpack8(ops.v[t], x1);
            unpack8(ops.v2[t], x2);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x1[e] * -1.4426950408889634f));
                const float sl = x1[e] * sg;
                o2[e] = g[e] * sl;
                g[e] = g[e] * x2[e] * (sg + sl * (1.f - sg));
            }
            out = make_uint4(cvt_pk_bf16(g[0], g[1]), cvt_pk_bf16(g[2], g[3]), cvt_pk_bf16(g[4], g[5]), cvt_pk_bf16(g[6], g[7]));
            if (ok) *reinterpret_cast<uint4*>(crow + L.oc + 2 * (int64_t)p.N) =
                make_uint4(cvt_pk_bf16(o2[0], o2[1]), cvt_pk_bf16(o2[2], o2[3]), cvt_pk_bf16(o2[4], o2[5]), cvt_pk_bf16(o2[6], o2[7]));
        } else if constexpr (EPI == PP_E_DACT_GELU) {
            float v[8], ax[8];
            unpack8(T, v);
            unpack8(ops.v[t], ax);
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
        if (ok) *reinterpret_cast<uint4*>(crow + L.oc) = out;
#endif
        if constexpr ((EPI == PP_E_RES || w4_is_dact(EPI)) && I < 7) {    // this piece is consumed: request the next row group's
            w4_epi_req<EPI, INTERIOR>(p, et, L, R.op, row0 + 16, wcol, ops.v[t]);
            if constexpr (EPI == PP_E_DACT_SWIGLU) w4_epi_req<EPI, INTERIOR>(p, et, L, R.op + 2 * (int64_t)p.N, row0 + 16, wcol, ops.v2[t]);
        }
        R.c += R.sc;
        R.c2 += R.sc2;
        R.op += R.so;
    }
    if constexpr (EPI == PP_E_RES && I < 7) w4_epi_gate<EPI, I + 1>(p, w, et, L, wrow, wcol, ops.gq);
}

template <int I>
__device__ __forceinline__ void w4_epi_read(unsigned char* slab, const W4EpiLane& L, w4_u32x4 (&Tv)[4]) {
    unsigned char* half = slab + (I & 1) * 4096;
#pragma unroll
    for (int t = 0; t < 4; ++t) Tv[t] = *reinterpret_cast<const w4_u32x4*>(half + L.rd[t]);
}

template <int EPI, bool INTERIOR>
__device__ __forceinline__ void w4_epilogue(const md_gemm_args& p, const PPPlan& w, const W4Tile& et, unsigned char* slab, int wrow, int wcol, int lane) {
    W4EpiLane L;
    w4_epi_lane<EPI>(p, et, wcol, lane, L);
    W4Ops ops;
    W4Rows R;
    const int64_t ldo = EPI == PP_E_RES ? p.ldr : p.ldaux;
    R.sc = 8 * p.ldc;
    R.sc2 = 8 * p.ldc2;
    R.so = 8 * ldo;
    R.c = et.cbase + ((int64_t)wrow * p.ldc + wcol) * 2;
    R.c2 = et.c2base ? et.c2base + ((int64_t)wrow * p.ldc2 + wcol) * 2 : nullptr;
    R.op = nullptr;
    if constexpr (EPI == PP_E_RES || w4_is_dact(EPI)) {
        R.op = et.opbase + ((int64_t)wrow * ldo + wcol) * 2;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            w4_epi_req<EPI, INTERIOR>(p, et, L, R.op, wrow + 4 * t, wcol, ops.v[t]);
            if constexpr (EPI == PP_E_DACT_SWIGLU) w4_epi_req<EPI, INTERIOR>(p, et, L, R.op + 2 * (int64_t)p.N, wrow + 4 * t, wcol, ops.v2[t]);
            R.op += R.so;
        }
        if constexpr (EPI == PP_E_RES) w4_epi_gate<EPI, 0>(p, w, et, L, wrow, wcol, ops.gq);
    }
    w4_u32x4 Tv[4];
    w4_epi_write<0>(slab, L);
#define W4_EPI_STEP(I)                                                                                                  \
    w4_epi_read<I>(slab, L, Tv);                                                                                        \
    if constexpr ((I) < 7) w4_epi_write<((I) < 7 ? (I) + 1 : 7)>(slab, L);                                               \
    if constexpr ((I) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
    w4_epi_store<EPI, I, INTERIOR>(p, w, et, L, wrow, wcol, Tv, ops, R);
    W4_EPI_STEP(0) W4_EPI_STEP(1) W4_EPI_STEP(2) W4_EPI_STEP(3) W4_EPI_STEP(4) W4_EPI_STEP(5) W4_EPI_STEP(6) W4_EPI_STEP(7)
#undef W4_EPI_STEP
}

// fp32 slice epilogue (weight gradients: split-K slices for md_splitk_reduce; interior tiles only).  A lane owns 4 consecutive columns of a
// row per 16 x 16 block: stored as they are, 16 bytes per lane -- a store instruction writes 16 rows x 64 contiguous bytes.
__device__ __forceinline__ void w4_epilogue_f32(float* tile, int64_t ldc, int wrow, int wcol, int lane_in) {
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    float* const p0 = tile + (int64_t)(wrow + (lane & 15)) * ldc + wcol + 4 * (lane >> 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // no store between a DMA load and its wait (gen_w4_acc.py, FRESH)
#define W4_F32_STEP(I, JH)                                                                                               \
    {                                                                                                                    \
        float a[16];                                                                                                     \
        w4_acc_read16<I, JH>(a);                                                                                         \
        float* const pr = p0 + (int64_t)(16 * (I)) * ldc + 64 * (JH);                                                    \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                                    \
            *reinterpret_cast<float4*>(pr + 16 * q) = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);   \
    }
    W4_F32_STEP(0, 0) W4_F32_STEP(0, 1) W4_F32_STEP(1, 0) W4_F32_STEP(1, 1) W4_F32_STEP(2, 0) W4_F32_STEP(2, 1) W4_F32_STEP(3, 0) W4_F32_STEP(3, 1)
    W4_F32_STEP(4, 0) W4_F32_STEP(4, 1) W4_F32_STEP(5, 0) W4_F32_STEP(5, 1) W4_F32_STEP(6, 0) W4_F32_STEP(6, 1) W4_F32_STEP(7, 0) W4_F32_STEP(7, 1)
#undef W4_F32_STEP
}

// amdgpu_num_vgpr(144): hipcc's values that live across the k-loop are below v96 by construction (every k-loop statement clobbers
// v[96:255]); inside an epilogue it may use v[96:143], which hold nothing then (gen_w4_acc.py register plan).  A budget of 96 was
// overrun -- not spilled -- by the instantiations with the heaviest epilogues (v96 / v97 handed out as temporaries: registers the k-loop
// owns).  scripts/check_w4_asm.py (tests/test_build_static.py) audits every build for v144+ / accumulator use.
template <int AKC, int BKC, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(144))) void gemm_bf16_w4_kernel(md_gemm_args p, PPPlan w) {
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
            if (AKC) ad.adA[ks] = lds0 + (unsigned)(ra * 128 + ((((4 * ks + (lane >> 4)) ^ (ra >> 1)) & 7) << 4));
            if (BKC) ad.adB[ks] = lds0 + (unsigned)(W4_BREG + rb * 128 + ((((4 * ks + (lane >> 4)) ^ (rb >> 1)) & 7) << 4));
        }
        if (AKC) {
#pragma unroll
            for (int i = 2; i < 8; ++i) ad.adA[i] = 0;
        } else {                                   // K-strided A image: as the K-strided B image below, rows <-> this wave's 128 output rows
            const int li = lane & 15, g = lane >> 4;
            const int kk = 8 * g + (li >> 2);
            const int swz = ((kk & 3) << 1) | (((kk >> 3) & 1) << 3);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int col = wr * 128 + 16 * i + 4 * (li & 3);
                ad.adA[i] = lds0 + (unsigned)(kk * 512 + ((((col >> 3) ^ swz) & 31) << 4) + ((col >> 2) & 1) * 8);
            }
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
    ad.ldsw = lds0 + (unsigned)wave * 1024u;
    // DMA pieces.  K-contiguous operand: piece x (0..7) = rows x * 32 + tid / 8, PHYSICAL chunk tid % 8 (= lane-linear: wave w writes
    // rows x * 32 + 8 w .. + 7).  K-strided operand: piece x = k-rows x * 8 + tid / 32, physical chunk tid % 32 (a wave loads two whole
    // 512-byte k-rows).  The logical chunk each lane fetches is formed in stager_open.

    // ---- load cursor: runs two k-tiles (one pair) ahead of the multiplications, across tile boundaries.  nk is even, so the
    // cursor changes tiles only at the bottom of the pair loop; inside a k-tile it only steps its scalar byte offset.
    int s_n = 0, s_kt = 0, s_koffA = 0, s_koffB = 0;
    int kstepA = AKC ? BKT * 2 : BKT * w.lda * 2;             // bytes per k-tile of the cursor's A / B tile (K-strided: 64 rows of ld elements;
    int kstepB = BKC ? BKT * 2 : BKT * w.ldb * 2;             // a grouped launch takes the leading dimensions of the tile's problem)
    constexpr bool GROUPABLE = !AKC && !BKC && EPI == PP_E_F32;
    u32x4 rA, rB;                                  // wave-uniform buffer descriptors of the cursor's A / B tile (first k element of its split)
    auto stager_open = [&](int n) {
        int m0, n0, batch, split;
        if (GROUPABLE && w.nprob) {                // grouped launch (the weight gradients of one DiT block): the item's problem supplies operands,
            int q;                                 // extents and leading dimensions (interior tiles only: md_gemm_w4_eligible)
            work_decode_grouped(w, w_first + n * w_stride, q, m0, n0, split);
            const PPProblem& pr = w.prob[q];
            const int sw = (tid >> 5) & 3;
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int c8 = ((tid & 31) ^ ((sw << 1) | ((x & 1) << 3))) * 8;
                const int kr = x * 8 + (tid >> 5);
                ad.aofs[x] = (unsigned)(kr * pr.lda + c8) * 2u;
                ad.bofs[x] = (unsigned)(kr * pr.ldb + c8) * 2u;
            }
            kstepA = BKT * pr.lda * 2;
            kstepB = BKT * pr.ldb * 2;
            const int64_t kb = (int64_t)split * w.kspan;
            const uint64_t ua = (uint64_t)(uintptr_t)(reinterpret_cast<const bf16*>(pr.A) + kb * pr.lda + m0);
            const uint64_t ub = (uint64_t)(uintptr_t)(reinterpret_cast<const bf16*>(pr.B) + kb * pr.ldb + n0);
            rA = u32x4{(unsigned)ua, (unsigned)(ua >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
            rB = u32x4{(unsigned)ub, (unsigned)(ub >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
            s_koffA = 0;
            s_koffB = 0;
            return;
        }
        work_decode(w, w_first + n * w_stride, m0, n0, batch, split);
        const int c = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;   // a DMA lane sits at PHYSICAL chunk tid % 8 of its row: it fetches the logical chunk the swizzle puts there
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const int row = x * 32 + (tid >> 3);
            if (AKC) {
                int ga = m0 + row;
                ga = (ga < w.M ? ga : w.M - 1) - m0;
                ad.aofs[x] = (unsigned)(ga * w.lda + c) * 2u;
            } else {                                       // K-strided A: as the K-strided B below (rows <-> the tile's 256 output rows)
                int gc = m0 + ((tid & 31) ^ ((((tid >> 5) & 3) << 1) | ((x & 1) << 3))) * 8;
                const int last = (w.M - 1) & ~7;
                gc = (gc < last ? gc : last) - m0;
                ad.aofs[x] = (unsigned)((x * 8 + (tid >> 5)) * w.lda + gc) * 2u;
            }
            if (BKC) {
                int gb = n0 + row;
                gb = (gb < w.N ? gb : w.N - 1) - n0;
                ad.bofs[x] = (unsigned)(gb * w.ldb + c) * 2u;
            } else {                                       // k-row 8 x + tid / 32, columns 8 (tid % 32) .. (clamped to the last whole chunk)
                int gc = n0 + ((tid & 31) ^ ((((tid >> 5) & 3) << 1) | ((x & 1) << 3))) * 8;     // physical chunk tid % 32 <- logical chunk ^ swz(k-row)
                const int last = (w.N - 1) & ~7;
                gc = (gc < last ? gc : last) - n0;
                ad.bofs[x] = (unsigned)((x * 8 + (tid >> 5)) * w.ldb + gc) * 2u;
            }
        }
        const int64_t kbeg = (int64_t)split * w.kspan;
        const bf16* pa = reinterpret_cast<const bf16*>(p.A) + (int64_t)batch * p.sA + (AKC ? (int64_t)m0 * w.lda + kbeg : kbeg * w.lda + m0);
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
    // prologue: k-tile 0 into buffer 0 (landed), k-tile 1 into buffer 1 (in flight), fragments of k-tile 0's k-step 0
    stager_open(0);
    w4_prologue<AKC, BKC>(ad, rA, rB, 0, 0, kstepA, kstepB);
    s_koffA = 2 * kstepA;
    s_koffB = 2 * kstepB;
    stager_pair_done();
    w4_first_reads<AKC, BKC>(ad);

    int c_n = 0, c_kt = 0;
    // One k-tile (scripts/gen_w4_acc.py): entering, the k-step-0 registers hold this k-tile's first fragments (buffer BUF; the reads possibly
    // still in flight) and the DMA loads of k-tile t + 1 are in flight into BUF ^ 1.  h0: first MFMAs + the k-step-1 fragment reads, barrier
    // (BUF is free); h1: the DMA loads of k-tile t + 2 (the cursor's) into BUF, vmcnt + barrier (k-tile t + 1 has landed); h2: the last MFMAs
    // + the next k-tile's first fragments.  FRESH = first k-tile of an output tile.
#define W4_KTILE(BUF, FRESH)                                                                                            \
    do {                                                                                                                \
        w4_h0<AKC, BKC, BUF, FRESH>(ad, rA, rB, s_koffA, s_koffB);                                                      \
        w4_h1<AKC, BKC, BUF, FRESH>(ad, rA, rB, s_koffA, s_koffB);                                                      \
        w4_h2<AKC, BKC, BUF, FRESH>(ad, rA, rB, s_koffA, s_koffB);                                                      \
        s_koffA += kstepA;                                                                                              \
        s_koffB += kstepB;                                                                                              \
        w4_h3<AKC, BKC, BUF>(ad);                                                                                       \
    } while (0)

    for (int it = 0; it < pairs; ++it) {
        if (c_kt == 0) W4_KTILE(0, true);            // first k-tile of an output tile: its first k-step takes C = 0
        else W4_KTILE(0, false);
        W4_KTILE(1, false);
        stager_pair_done();
        c_kt += 2;
        if (c_kt == w.nk) {
            // ---- epilogue of the finished tile (the loads of the next tile's first two k-tiles are in flight / in LDS)
            if constexpr (EPI == PP_E_F32) {
                int m0, n0, batch = 0, split;
                float* tile;
                int64_t ldc;
                if (GROUPABLE && w.nprob) {                // dense rows of the problem's N inside the slice, at its c_off
                    int q;
                    work_decode_grouped(w, w_first + c_n * w_stride, q, m0, n0, split);
                    const PPProblem& pr = w.prob[q];
                    ldc = pr.N;
                    tile = reinterpret_cast<float*>(p.C) + (int64_t)split * p.sSplit + pr.c_off + (int64_t)m0 * ldc + n0;
                } else {
                    work_decode(w, w_first + c_n * w_stride, m0, n0, batch, split);
                    ldc = p.ldc;
                    tile = reinterpret_cast<float*>(p.C) + (int64_t)batch * p.sC + (int64_t)split * p.sSplit + (int64_t)m0 * ldc + n0;
                }
                asm volatile("s_nop 15\n\ts_nop 15");      // MFMA result -> v_accvgpr_read wait states
                w4_epilogue_f32(tile, ldc, wr * 128, wc * 128, lane);
                c_kt = 0;
                ++c_n;
                continue;
            }
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
            if (et.mlim >= PT && et.nlim >= PT) w4_epilogue<EPI, true>(p, w, et, slab, wr * 128, wc * 128, lane);     // interior tile: no predicates
            else w4_epilogue<EPI, false>(p, w, et, slab, wr * 128, wc * 128, lane);
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

// Instantiated: TN (weight gradients) fp32 slices; NT (nn.Linear forward) bf16 / gated residual / activation derivative; NN (dgrads, the MoE's [E, in, out] experts) bf16 /
// GELU(erf) with the raw copy or the cached derivative / residual / the SwiGLU backward (the w3 data gradient).
static bool w4_instantiated(int akc, int bkc, int epi) {
    if (!akc) return !bkc && epi == PP_E_F32;                    // weight gradients: both operands K-strided, fp32 slices
    if (bkc) return epi == PP_E_BF16 || epi == PP_E_RES || epi == PP_E_DACT_GELU || epi == PP_E_DACT_MUL;
    return epi == PP_E_BF16 || epi == PP_E_BF16_GELU || epi == PP_E_BF16_GELU_D || epi == PP_E_RES || epi == PP_E_DACT_SWIGLU;
}

bool md_gemm_w4_eligible(const md_gemm_args* a) {
    const int epi = md_gemm_pp_epi_kind(a);
    if (epi < 0 || !w4_instantiated(a->a_kcontig, a->b_kcontig, epi)) return false;
    if (!md_gemm_pp_shape_ok(a, epi)) return false;              // K span, N % 8, leading-dimension ranges, gate rows, interior tiles, problem table, ...
    if (epi == PP_E_DACT_SWIGLU && (!a->aux || a->batch != 1 || a->ldaux < 2 * a->N || a->ldc < 2 * a->N)) return false;
    if (a->A_list || a->B_list || a->timeline) return false;
    if (a->bias || a->alpha != 1.f) return false;                // only the plain form of every epilogue is built
    const int64_t kspan = a->K / a->ksplit;
    if (epi == PP_E_F32) {                                       // fp32 slices: whole interior tiles, scalar byte offsets of both K-strided cursors in 31 bits
        if (a->ksplit > 1 && a->sSplit <= 0) return false;
        if (a->problems) {
            for (int i = 0; i < a->n_problems; ++i) {
                const md_gemm_problem& s = a->problems[i];
                if (s.M % PT || s.N % PT || kspan * (s.lda > s.ldb ? s.lda : s.ldb) >= ((int64_t)1 << 30)) return false;
            }
            return true;
        }
        if (a->M % PT || a->N % PT || a->ldc % 4) return false;
        return kspan * (a->lda > a->ldb ? a->lda : a->ldb) < ((int64_t)1 << 30);
    }
    if (a->ksplit != 1 || a->problems) return false;
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
#define W4_LAUNCH(AK, BK, E) hipLaunchKernelGGL((gemm_bf16_w4_kernel<AK, BK, E>), grid, block, 0, stream, *a, w)
    if (!a->a_kcontig) {
        W4_LAUNCH(0, 0, PP_E_F32);
    } else if (a->b_kcontig) {
        if (epi == PP_E_BF16) W4_LAUNCH(1, 1, PP_E_BF16);
        else if (epi == PP_E_RES) W4_LAUNCH(1, 1, PP_E_RES);
        else if (epi == PP_E_DACT_MUL) W4_LAUNCH(1, 1, PP_E_DACT_MUL);
        else W4_LAUNCH(1, 1, PP_E_DACT_GELU);
    } else {
        if (epi == PP_E_BF16) W4_LAUNCH(1, 0, PP_E_BF16);
        else if (epi == PP_E_BF16_GELU) W4_LAUNCH(1, 0, PP_E_BF16_GELU);
        else if (epi == PP_E_BF16_GELU_D) W4_LAUNCH(1, 0, PP_E_BF16_GELU_D);
        else if (epi == PP_E_DACT_SWIGLU) W4_LAUNCH(1, 0, PP_E_DACT_SWIGLU);
        else W4_LAUNCH(1, 0, PP_E_RES);
    }
#undef W4_LAUNCH
    MD_LAUNCH_CHECK();
    return 0;
}

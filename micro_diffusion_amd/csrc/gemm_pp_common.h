// Device helpers shared by the two persistent GEMM kernels (gemm_pp.hip: whole-tile schedule; the half-by-half experiment under scripts/experiments:
// schedule): fragment types and LDS reads, DMA staging offsets, the host-computed work plan, epilogue address helpers.
#pragma once
#include "md_common.h"
#include "gemm_common.h"

namespace {

constexpr int PT = 256;            // output tile rows / columns
constexpr int HT = 16384;          // bytes of one half-tile
constexpr int B_REGION = 65536;    // A slots live in [0, 64 KiB), B slots in [64 KiB, 128 KiB): slot(b, h) = b * 32768 + h * 16384
constexpr int NUM_CU = 256;        // MI355X

template <int KC>
struct Frag;
template <>
struct Frag<1> {
    bf16x8 v;
    __device__ __forceinline__ bf16x8 get() const { return v; }
};
template <>
struct Frag<0> {
    bf16x4 lo, hi;
    __device__ __forceinline__ bf16x8 get() const {
        bf16x8 f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return f;
    }
};

// Fragment reads are inline asm: hipcc puts a full vmcnt(0) in front of every LDS read it can see while an LDS-DMA is
// pending.  Their destinations are "pinned" after the caller's lgkmcnt(0) (frag_pin: an empty asm that re-defines the
// registers), so no copy of a destination can be scheduled before the data has landed.
template <int KC, int OFF>
__device__ __forceinline__ void frag_read(Frag<KC>& f, unsigned addr) {
    static_assert(OFF >= 0 && OFF + 1024 < 65536, "ds offset field is 16 bits");
    if constexpr (KC) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.v) : "v"(addr), "i"(OFF) : "memory");
    } else {
        asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                     : "=&v"(f.lo), "=&v"(f.hi)
                     : "v"(addr), "i"(OFF), "i"(OFF + 1024)
                     : "memory");
    }
}
template <int KC>
__device__ __forceinline__ void frag_pin(Frag<KC>& f) {
    if constexpr (KC) asm volatile("" : "+v"(f.v));
    else asm volatile("" : "+v"(f.lo), "+v"(f.hi));
}

// Per-lane LDS byte addresses of the fragment reads (computed once).
//  K-contiguous: ad[ks] (k-step 0..3); the row-fragment index i is an immediate (i * 4096).
//  K-strided   : ad[i]  (row-fragment 0..1); the k-step is an immediate (ks * 4096).
// row0 = first row (column) of this wave's 64- (A) or 32- (B) wide strip inside the 128-wide half-tile.
template <int KC>
__device__ __forceinline__ void frag_addrs(unsigned (&ad)[4], unsigned base, int row0, int lane) {
    if constexpr (KC) {
        const int r = row0 + (lane & 31);
        ad[0] = base + r * 128 + ((((lane >> 5) ^ (r >> 1)) & 7) << 4);   // k-step ks: chunk (2 ks + hi) ^ s == (hi ^ s) ^ 2 ks,
        ad[1] = ad[2] = ad[3] = 0;                                        // i.e. ad[0] ^ (ks << 5)
    } else {
        const int li = lane & 15;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int col = row0 + i * 32 + ((lane >> 4) & 1) * 16 + (li & 3) * 4;
            const int kk = (lane >> 5) * 8 + (li >> 2);
            const int pc = (col >> 3) ^ ((kk & 3) << 2);
            ad[i] = base + kk * 256 + pc * 16 + ((col >> 2) & 1) * 8;
        }
        ad[2] = ad[3] = 0;
    }
}
template <int KC, int SLOT, int I, int KS>
__device__ __forceinline__ void frag_read_at(Frag<KC>& f, const unsigned (&ad)[4], int kx32) {
    // kx32 = 32 hidden behind an asm so that ad[0] ^ (KS * 32) is recomputed at the read (one v_xor) instead of living in
    // three more registers per operand for the whole kernel
    if constexpr (KC) frag_read<KC, SLOT + I * 4096>(f, KS == 0 ? ad[0] : (ad[0] ^ (unsigned)(KS * kx32)));
    else frag_read<KC, SLOT + KS * 4096>(f, ad[I]);
}

// Per-lane byte offsets (relative to the tile's base pointer) of the two 1 KiB DMA pieces this wave contributes to each
// half-tile h of an operand whose tile starts at row r0 (rmax rows in total).
template <int KC>
__device__ __forceinline__ void stage_offsets(unsigned (&ofs)[2][2], int r0, int rmax, int ld, int wave, int lane) {
    asm volatile("" : "+v"(lane));     // evaluated once per output tile: keep its lane-derived sub-terms out of the loop-invariant registers
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (KC) {
                const int row = (wave * 2 + j) * 8 + (lane >> 3);            // row inside the half-tile
                const int c = (lane & 7) ^ ((row >> 1) & 7);                  // logical 16-byte chunk this lane fetches
                int gr = r0 + h * 128 + row;
                gr = (gr < rmax ? gr : rmax - 1) - r0;
                ofs[h][j] = (unsigned)(gr * ld + c * 8) * 2u;
            } else {
                const int kk = (wave * 2 + j) * 4 + (lane >> 4);
                const int c = (lane & 15) ^ ((kk & 3) << 2);
                int gc = r0 + h * 128 + c * 8;
                const int last = (rmax - 1) & ~7;
                gc = (gc < last ? gc : last) - r0;
                ofs[h][j] = (unsigned)(kk * ld + gc) * 2u;
            }
        }
}

__device__ __forceinline__ void stage_half(const char* base, const unsigned (&ofs)[2], unsigned char* slot, int wave) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(base + ofs[j]), (lds_void_t*)(slot + (wave * 2 + j) * 1024), 16, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------------
// Work list.  W = tiles x batch x ksplit items; XCD x (workgroups with blockIdx % 8 == x share an L2) owns a contiguous
// range of item ids, and its workgroups take consecutive ids in every round, so the ~32 tiles an XCD works on at any
// time form a (32 / group_n) x group_n block of the output that shares A row-panels and B column-panels in L2.
// The plan is computed on the host (md_gemm_pp_launch) and passed by value.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PP_MAX_PROB = MD_GEMM_MAX_PROBLEMS;
struct PPProblem {                 // one problem of a grouped launch (md_gemm_args.problems), everything the kernel needs of it
    const void* A;
    const void* B;
    int64_t c_off;                 // elements into every fp32 slice
    int lda, ldb, M, N;
    int ntm, ntn, tile0, group_n;  // tiles, first global tile id, raster group
};

struct PPPlan {
    int ntm, ntn, ntiles, total;   // tiles per problem (grouped: ntiles = all problems' tiles); items in total
    int nk;                        // k-tiles (64 deep) per item, even
    int kspan;                     // elements of K per split
    int group_n;                   // column-tiles per raster group
    int M, N, ksplit;
    int lda, ldb;
    int rps_shift;                 // log2(rows_per_sample) when that is a power of two, else -1 (gated residual only)
    int nseg, nk_seg;              // operand lists: segments per item and k-tiles per segment (nseg <= 1: one operand pair per item)
    int nprob;                     // grouped launch: number of problems (0 = the single problem of md_gemm_args)
    // "whole rounds + split-K tail" (md_gemm_args.tail_ws): items [0, tail_first) are dealt to the workgroups as whole tiles;
    // item tail_first + u / tail_split is cut along K into tail_split units of tail_nk k-tiles, unit u = workgroup u's LAST
    // item, written as a raw 256 x 256 fp32 tile to tail_ws + u * 256 KiB (pp_tail_fixup_kernel finishes those tiles).
    int tail_first, tail_units, tail_split, tail_nk;   // tail_units = 0: no tail
    PPProblem prob[PP_MAX_PROB];
};

__device__ __forceinline__ void work_decode(const PPPlan& w, int item, int& m0, int& n0, int& batch, int& split) {
    const unsigned y = (unsigned)item / (unsigned)w.ntiles;
    const unsigned t = (unsigned)item - y * (unsigned)w.ntiles;
    batch = (int)(y / (unsigned)w.ksplit);
    split = (int)(y - (unsigned)batch * (unsigned)w.ksplit);
    const unsigned per_group = (unsigned)(w.group_n * w.ntm);
    const unsigned g = t / per_group, rem = t - g * per_group;
    const int first_n = (int)g * w.group_n;
    const int gn = (w.ntn - first_n) < w.group_n ? (w.ntn - first_n) : w.group_n;
    const unsigned tm = rem / (unsigned)gn;
    m0 = (int)tm * PT;
    n0 = (first_n + (int)(rem - tm * (unsigned)gn)) * PT;
}

// Grouped launch: item -> (problem q, tile origin, split).  Items are split-major (y = item / all tiles), tiles of a problem
// rastered like a single problem's.  All values wave-uniform (scalar loop over <= 8 problems).
__device__ __forceinline__ void work_decode_grouped(const PPPlan& w, int item, int& q, int& m0, int& n0, int& split) {
    const unsigned y = (unsigned)item / (unsigned)w.ntiles;
    const int t = (int)((unsigned)item - y * (unsigned)w.ntiles);
    split = (int)y;
    q = 0;
#pragma unroll 1
    for (int i = 1; i < w.nprob; ++i)
        if (t >= w.prob[i].tile0) q = i;
    const PPProblem& pr = w.prob[q];
    const unsigned tl = (unsigned)(t - pr.tile0);
    const unsigned per_group = (unsigned)(pr.group_n * pr.ntm);
    const unsigned g = tl / per_group, rem = tl - g * per_group;
    const int first_n = (int)g * pr.group_n;
    const int gn = (pr.ntn - first_n) < pr.group_n ? (pr.ntn - first_n) : pr.group_n;
    const unsigned tm = rem / (unsigned)gn;
    m0 = (int)tm * PT;
    n0 = (first_n + (int)(rem - tm * (unsigned)gn)) * PT;
}

enum {
    PP_E_BF16 = 0,       // MD_EPI_STORE_BF16, no activation            (+ bias, + C2)
    PP_E_BF16_GELU = 1,  // MD_EPI_STORE_BF16, GELU(erf)                (+ bias, + C2: the MoE fc1)
    PP_E_RES = 2,        // MD_EPI_RESIDUAL                             (+ bias, + gate, + C2)
    PP_E_DACT_GELU = 3,  // MD_EPI_DACT through GELU(erf)               (the MoE fc1 dgrad)
    PP_E_F32 = 4,        // MD_EPI_STORE_F32                            (+ bias; split-K slices)
    PP_E_BF16_GELU_D = 5,  // MD_EPI_STORE_BF16, GELU(erf), dact_cached: C2 = gelu'(pre-activation) instead of the pre-activation
    PP_E_DACT_MUL = 6,     // MD_EPI_DACT, dact_cached: C = acc * aux  (aux = the cached derivative)
    PP_E_DACT_SWIGLU = 7   // MD_EPI_SWIGLU_BWD: the SwiGLU backward fused into the w3 data gradient (w4 kernel only):
                           //   aux = h12 [M, 2N], C = dh12 [M, 2N]:  C[:, n] = da * h2 * silu'(h1),  C[:, N + n] = da * silu(h1),  da = bf16(acc)
    // MD_EPI_ACCUM_F32 stays on the gemm.hip kernels: its operand (32 fp32 per lane and quadrant) does not fit beside the
    // fragments, and without the one-phase-ahead prefetch each quadrant would drain the DMA ring.
};

struct EpiTile {
    int m0, n0, batch, split;
    int q, M, N, ldc;      // grouped launch: problem index; extent and leading dimension of the output this tile belongs to
    // filled by epi_open() when the tile's k-loop finishes (all wave-uniform):
    bool plain;            // interior tile, no bias, alpha == 1: the short form of the epilogue
    char* cbase;           // &C[batch, (split,) m0, n0]
    char* c2base;          // &C2[batch, m0, n0] or nullptr
    const char* opbase;    // &res[m0, n0] (RESIDUAL) / &aux[batch, m0, n0] (DACT)
    const char* gbase;     // &gate[0, n0] or nullptr
};

// Epilogue operand loads are inline asm with hand-counted waits (hipcc would put a vmcnt(0) in front of their first use and
// drain the DMA ring).  The compiler therefore does not know that the destination is not valid yet, and any register COPY
// it places between the load and the wait would copy stale data.  What keeps copies out of that window:
//  * ONE request routine per call site of the kernel (epi_prefetch has no fast / general split): with two sibling request
//    sites merged by a phi, the two asm results were allocated to different physical registers and a v_mov followed the load
//    directly -> NaNs.  scripts/check_pp_asm.py verifies on the built code that every request site of a kernel's main loop
//    writes the same physical registers (tests/test_build_static.py runs it);
//  * the consumer re-defines the registers in place (landed(): an empty asm volatile, which cannot move above the asm
//    volatile wait that precedes it in program order) before their first use.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void asm_load16(u32x4& dst, const void* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
}
__device__ __forceinline__ uint4 landed(u32x4& v) {
    asm volatile("" : "+v"(v));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Lane geometry of the epilogue: rl / cl = row / column inside the 256 x 256 tile of element block (i = 0, pp = 0) of
// quadrant (0, 0); block (IH, JH, i, pp) adds (IH * 128 + i * 32) rows and (JH * 128 + pp * 16) columns.
struct EpiLane {
    int lane, wrow, wcol;   // wrow = wr * 64, wcol = wc * 32: this wave's strip inside a 128-wide half (wave-uniform)
    // rl / cl are recomputed from the lane id at every use (behind an asm, so they are not hoisted into loop-invariant
    // registers): every VGPR that lives across the main loop is one the fragment / prefetch registers cannot have.
    __device__ __forceinline__ void coords(int& rl, int& cl) const {          // accumulator-side geometry (fp32 slices, bias)
        int l = lane;
        asm volatile("" : "+v"(l));
        rl = wrow + (l & 31);
        cl = wcol + (l >> 5) * 8;
    }
    // Store-side geometry of the bf16 epilogues (quad_rows below): the four lanes of quad q = lane / 4 hold the four
    // 16-byte pieces (k = lane % 4) of rows rq .. rq + 3 of the quadrant, rq = (q / 8) * 32 + (q % 8) * 4.
    __device__ __forceinline__ void quad_coords(int& rq, int& cq) const {
        int l = lane;
        asm volatile("" : "+v"(l));
        const int q = l >> 2;
        rq = wrow + (q >> 3) * 32 + (q & 7) * 4;
        cq = wcol + (l & 3) * 8;
    }
};

// All epilogue addresses are  uniform 64-bit base of the tile (SGPRs)  +  32-bit per-lane byte offset  (one VGPR each,
// global_* saddr form): ptr + batch stride + m0 * ld + n0, and (row_in_tile * ld + col_in_tile) * element size.
__device__ __forceinline__ const char* tile_base(const void* ptr, int64_t batch_off, const EpiTile& et, int64_t ld, int esize) {
    return reinterpret_cast<const char*>(ptr) + (batch_off + (int64_t)et.m0 * ld + et.n0) * esize;
}
__device__ __forceinline__ unsigned lane_off(int r, int c, int ld, int esize) { return (unsigned)(r * ld + c) * (unsigned)esize; }

__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[2 * e] = __uint_as_float(w[e] << 16);
        f[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
    }
}
__device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
    U128 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t.e[e] = f2bf(v[e]);
    return t.u;
}
__device__ __forceinline__ float bf_round(float v) { return bf2f(f2bf(v)); }

// {lo, hi} -> packed bf16 pair (round to nearest even), one instruction; spelled out so that the pairing is the one the
// epilogue wants (hipcc's own vectorisation of eight scalar conversions paired (0,2)(1,3) and re-shuffled afterwards).
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// v_permlane32_swap: the upper half-wave of a and the lower half-wave of b change places.
__device__ __forceinline__ void half_swap(unsigned& a, unsigned& b) {
    const auto sw = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = sw[0];
    b = sw[1];
}
// Eight fp32 in the accumulator ("pre-swap") layout -- v[0..3] = columns 4 hi + e of the block's first 8-column group,
// v[4..7] = the same of its second group -- to one 16-byte row piece of 8 consecutive columns (16 pp + 8 hi ..).
__device__ __forceinline__ uint4 pack_swap8(const float (&v)[8]) {
    unsigned a0 = cvt_pk_bf16(v[0], v[1]), a1 = cvt_pk_bf16(v[2], v[3]);
    unsigned b0 = cvt_pk_bf16(v[4], v[5]), b1 = cvt_pk_bf16(v[6], v[7]);
    half_swap(a0, b0);
    half_swap(a1, b1);
    return make_uint4(a0, a1, b0, b1);
}
// The inverse for an operand that was loaded as a 16-byte row piece: back to the accumulator layout, as fp32.
__device__ __forceinline__ void swap_unpack8(uint4 u, float (&f)[8]) {
    half_swap(u.x, u.z);
    half_swap(u.y, u.w);
    unpack8(u, f);
}

// ---------------------------------------------------------------------------------------------------------------------
// quad_rows: one 64 x 32 quadrant (acc[2] = its two 32-row fragments) as bf16 in a layout whose STORES are runs of 64
// contiguous bytes.  Straight out of the MFMA a lane owns 16-byte pieces of one row and a store instruction covers 32 rows x
// 32 bytes -- 64 separate write requests per instruction; with the four lanes of a quad writing one 64-byte run the same
// bytes cost half (profiles/r2_gemm_pp256_store_pattern.txt: 4.4 -> 2.2 us per tile), and unlike a transposition through
// LDS this one uses only the VALU, which has slack in an epilogue phase:
//   1. pack to bf16 pairs; 8 x v_permlane32_swap between the two row fragments: lane L then owns ALL 32 columns (four pieces
//      W[0..3]) of row 32 (L / 32) + L % 32;
//   2. a 4 x 4 transpose of 16-byte elements inside each quad, two butterfly stages of 16 DPP moves + selects: lane 4 q + k
//      ends with piece k of the quad's four rows, T[t] = row rq + t (EpiLane::quad_coords).
// ---------------------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ unsigned quad_mov(unsigned v) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ uint4 quad_sel(bool take_nb, const uint4& own, const uint4& nb) { return take_nb ? nb : own; }
template <int CTRL>
__device__ __forceinline__ uint4 quad_mov4(const uint4& v) {
    return make_uint4(quad_mov<CTRL>(v.x), quad_mov<CTRL>(v.y), quad_mov<CTRL>(v.z), quad_mov<CTRL>(v.w));
}
template <typename F>
__device__ __forceinline__ void quad_rows(int l, F&& block, uint4 (&T)[4]) {
    // block(i, g, v[4]): the lane's fp32 values of fragment i, column group g (columns 8 g + 4 hi + e), already scaled / biased
    unsigned A[4][2], B[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float v0[4], v1[4];
        block(0, g, v0);
        block(1, g, v1);
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            A[g][d] = cvt_pk_bf16(v0[2 * d], v0[2 * d + 1]);
            B[g][d] = cvt_pk_bf16(v1[2 * d], v1[2 * d + 1]);
            half_swap(A[g][d], B[g][d]);        // A: columns 8 g + 2 d (+1), B: columns 8 g + 4 + 2 d (+1) of this lane's row
        }
    }
    uint4 W[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) W[g] = make_uint4(A[g][0], A[g][1], B[g][0], B[g][1]);
    constexpr int X1 = 0xB1, X2 = 0x4E;         // quad_perm [1,0,3,2] / [2,3,0,1]: the lane that differs in bit 0 / bit 1
    const bool odd = (l & 1) != 0, hi2 = (l & 2) != 0;
    const uint4 U0 = quad_sel(odd, W[0], quad_mov4<X1>(W[1])), U1 = quad_sel(!odd, W[1], quad_mov4<X1>(W[0]));
    const uint4 V0 = quad_sel(odd, W[2], quad_mov4<X1>(W[3])), V1 = quad_sel(!odd, W[3], quad_mov4<X1>(W[2]));
    T[0] = quad_sel(hi2, U0, quad_mov4<X2>(V0));
    T[1] = quad_sel(hi2, U1, quad_mov4<X2>(V1));
    T[2] = quad_sel(!hi2, V0, quad_mov4<X2>(U0));
    T[3] = quad_sel(!hi2, V1, quad_mov4<X2>(U1));
}

#define PP_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")


// ---- host side, shared by the launchers
inline int md_gemm_pp_epi_kind(const md_gemm_args* a) {
    switch (a->mode) {
        case MD_EPI_STORE_BF16:
            return a->act == MD_ACT_NONE ? PP_E_BF16 : a->act == MD_ACT_GELU_ERF ? ((a->dact_cached && a->C2) ? PP_E_BF16_GELU_D : PP_E_BF16_GELU) : -1;
        case MD_EPI_RESIDUAL: return PP_E_RES;
        case MD_EPI_DACT: return a->act == MD_ACT_GELU_ERF ? (a->dact_cached ? PP_E_DACT_MUL : PP_E_DACT_GELU) : -1;
        case MD_EPI_STORE_F32: return PP_E_F32;
        case MD_EPI_SWIGLU_BWD: return PP_E_DACT_SWIGLU;
        default: return -1;
    }
}


inline bool md_gemm_pp_plan(const md_gemm_args* a, PPPlan* w) {
    w->nprob = 0;
    w->ntm = (int)((a->M + PT - 1) / PT);
    w->ntn = (int)((a->N + PT - 1) / PT);
    w->ntiles = w->ntm * w->ntn;
    if (a->problems && a->n_problems > 0) {
        w->nprob = a->n_problems;
        int t0 = 0;
        const int64_t kspan_ = a->K / a->ksplit;
        for (int i = 0; i < a->n_problems; ++i) {
            const md_gemm_problem& s = a->problems[i];
            PPProblem& d = w->prob[i];
            d.A = s.A; d.B = s.B; d.c_off = s.c_off;
            d.lda = (int)s.lda; d.ldb = (int)s.ldb; d.M = (int)s.M; d.N = (int)s.N;
            d.ntm = (int)((s.M + PT - 1) / PT);
            d.ntn = (int)((s.N + PT - 1) / PT);
            d.tile0 = t0;
            int64_t g = (2 << 20) / (PT * kspan_ * 2);      // same L2 raster rule as a single problem (md_gemm_bf16)
            if (g < 4) g = 4;
            if (g > d.ntn) g = d.ntn;
            d.group_n = (int)g;
            t0 += d.ntm * d.ntn;
        }
        w->ntiles = t0;
    }
    const int64_t total = (int64_t)w->ntiles * (w->nprob ? 1 : a->batch) * a->ksplit;
    if (total > (1 << 30)) return false;
    w->total = (int)total;
    w->kspan = (int)(a->K / a->ksplit);
    w->nk = w->kspan / BKT;
    w->nseg = (a->A_list && a->list_segments > 1) ? a->list_segments : 1;
    w->nk_seg = w->nk / w->nseg;
    w->group_n = a->raster_group_n > 0 ? a->raster_group_n : 1;
    w->M = (int)a->M;
    w->N = (int)a->N;
    w->ksplit = a->ksplit;
    w->lda = (int)a->lda;
    w->ldb = (int)a->ldb;
    w->tail_first = w->total;
    w->tail_units = w->tail_split = w->tail_nk = 0;
    w->rps_shift = -1;
    if (a->rows_per_sample > 0 && (a->rows_per_sample & (a->rows_per_sample - 1)) == 0) {
        w->rps_shift = 0;
        while (((int64_t)1 << w->rps_shift) < a->rows_per_sample) ++w->rps_shift;
    }
    return true;
}

}  // namespace

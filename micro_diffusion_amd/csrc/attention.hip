// Fused multi-head attention (forward + backward) for the four MicroDiT shapes: patch-mixer self-attention
// (S = T), backbone self-attention over the un-masked tokens (S = Tk), cross-attention to the 77 caption tokens,
// and the caption block's 77 x 77 self-attention.  Non-causal, no mask, head_dim 64 (XL/2) or 32 (Tiny).
//
// One wave owns 32 query rows (forward, dQ) or 32 key rows (dK/dV); K/V (resp. Q/dO) tiles of 32 rows are shared
// by the workgroup's waves through LDS.  All products run on v_mfma_f32_32x32x16_bf16:
//   S^T = K Q^T is computed "swapped" so that every lane holds one query column of the 32x32 score tile: the
//   softmax max / sum are 16 in-register values + one cross-half shuffle (no LDS round trip), and the fp32
//   probabilities are exactly the B operand of the following P.V product (O^T = V^T P^T) after a bf16 pack.
//   Operands that are needed transposed (V^T, K^T, Q^T, dO^T) are read from the row-major LDS tile with
//   ds_read_b64_tr_b16, so nothing is transposed through memory.
// Replaces F.scaled_dot_product_attention (utils.py:127-132, 188-193) and its autograd backward.
#include <stdlib.h>
#include "md_common.h"
#include "../../include/microdit_hip.h"

namespace {

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x4 tr4(const unsigned char* p) {
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    U64 t;
    t.s = v;
    return t.h;
}

// A-operand fragment (32 rows x 16 k) of the TRANSPOSE of a row-major LDS tile T[k][col]:
// result row = col0 + (lane & 31), k-slots 0..3 = T rows k_lo..k_lo+3, slots 4..7 = rows k_hi..k_hi+3.
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, int pitch, int k_lo, int k_hi, int col0, int lane) {
    const int li = lane & 15;
    const int col = col0 + ((lane >> 4) & 1) * 16 + (li & 3) * 4;
    const bf16x4 lo = tr4(tile + (k_lo + (li >> 2)) * pitch + col * 2);
    const bf16x4 hi = tr4(tile + (k_hi + (li >> 2)) * pitch + col * 2);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// Row-major fragment (row = row0 + (lane & 31), k = kbase + (lane >> 5) * 8 .. +7) from an LDS tile.
__device__ __forceinline__ bf16x8 row_frag(const unsigned char* tile, int pitch, int kbase, int lane) {
    U128 t;
    t.u = *reinterpret_cast<const uint4*>(tile + (lane & 31) * pitch + (kbase + (lane >> 5) * 8) * 2);
    return t.h;
}

// Stage `nstage` rows x HD of a [rows, ld] bf16 matrix (rows >= nrows are zero-filled) into an LDS tile.
template <int HD>
__device__ __forceinline__ void stage_tile(unsigned char* tile, int pitch, const bf16* src, int64_t ld, int64_t row0,
                                           int64_t nrows, int nstage, int tid, int nthreads) {
    constexpr int CPR = HD / 8;
    for (int task = tid; task < nstage * CPR; task += nthreads) {
        const int r = task / CPR, c = task % CPR;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row0 + r < nrows) v = *reinterpret_cast<const uint4*>(src + (row0 + r) * ld + c * 8);
        *reinterpret_cast<uint4*>(tile + r * pitch + c * 16) = v;
    }
}

constexpr int STG = 32;    // rows staged per barrier phase.  Measured (scripts/bench_attn.py): 32 beats 64 and 128 -- at S = 64..256 these
                           // kernels stream q/k/v/o at 3.4-4.1 TB/s (HBM-bound); bigger phases only cost occupancy (LDS, VGPRs)

// Stage the same `nstage` (<= STG) rows of TWO matrices (K and V, or Q and dO) into their LDS tiles with every global load issued
// before the first LDS write.  The staging registers cost occupancy (attn_fwd at 4 chunks: 124 -> 176 VGPRs, -20 % on the
// 256-row shapes), so the kernels are instantiated per chunk count and launched with the smallest that covers their block size.
// (Two stage_tile calls are two run-time loops of load -> wait -> write: four dependent HBM round trips per phase.)
template <int HD, int MAXIT>
__device__ __forceinline__ void stage_pair(unsigned char* tileA, unsigned char* tileB, int pitch, const bf16* A, int64_t lda, const bf16* B,
                                           int64_t ldb, int64_t row0, int64_t nrows, int nstage, int tid, int nthreads) {
    constexpr int CPR = HD / 8;                 // MAXIT >= nstage * CPR / nthreads chunks per thread and matrix (the callers pick it)
    uint4 ra[MAXIT], rb[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int task = tid + it * nthreads, r = task / CPR, c = task % CPR;
        ra[it] = rb[it] = make_uint4(0, 0, 0, 0);
        if (task < nstage * CPR && row0 + r < nrows) {
            ra[it] = *reinterpret_cast<const uint4*>(A + (row0 + r) * lda + c * 8);
            rb[it] = *reinterpret_cast<const uint4*>(B + (row0 + r) * ldb + c * 8);
        }
    }
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int task = tid + it * nthreads, r = task / CPR, c = task % CPR;
        if (task < nstage * CPR) {
            *reinterpret_cast<uint4*>(tileA + r * pitch + c * 16) = ra[it];
            *reinterpret_cast<uint4*>(tileB + r * pitch + c * 16) = rb[it];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Softmax arithmetic of the backward kernels.  At S = 256 these kernels are VALU-bound, not MFMA- or HBM-bound (round 4: with
// the tile loops emptied the 256 x 256 backward runs 2.2x faster, the 64-row shapes 1.04x): per score element the first version
// spent ~12 instructions -- a 64-bit index compare + select for the sequence mask on EVERY tile, scale, subtract, exp through
// a multiply, two more multiplies for dS -- next to ~1.5 MFMA cycles.  Now: ONE fma + v_exp_f32 (scores go to the log2 domain
// with c1 = scale * log2(e) and the saved log-sum-exp pre-multiplied by log2(e)), dS without the softmax scale (it is applied
// once, to the dQ / dK accumulators at the store), and the mask only on a ragged last tile (`rem` = rows / keys left, wave-uniform).
// ---------------------------------------------------------------------------------------------------------------------
constexpr float LOG2E = 1.4426950408889634f;
__device__ __forceinline__ int tile_slot(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }   // row / key of accumulator register r

// (The mask is a per-element compare + select against the wave-uniform `rem`, and the exponent is formed as s * c1 - l2, not as a
// fused multiply-add inside a branch on `rem < 32`: the branchy / fma forms cost 8-12 more VGPRs in every kernel here and pushed
// the capped two-phase kernels into scratch.)
// lane <-> query (l2, dlt belong to this lane), registers <-> the tile's 32 keys:  s <- dS^T / scale = P^T (dP^T - delta)
__device__ __forceinline__ void ds_cols(f32x16& s, const f32x16& dp, float c1, float l2, float dlt, int rem, int hh) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float pr = tile_slot(r, hh) < rem ? __builtin_amdgcn_exp2f(s[r] * c1 - l2) : 0.f;
        s[r] = pr * (dp[r] - dlt);
    }
}
// lane <-> key, registers <-> the tile's 32 query rows (tL2 / tDlt: their log2-domain log-sum-exp / delta in LDS):
// s <- P,  dp <- dS / scale = P (dP - delta)      (WANT_P / WANT_DS select the outputs a pass needs)
template <bool WANT_P, bool WANT_DS>
__device__ __forceinline__ void p_ds_rows(f32x16& s, f32x16& dp, float c1, const float* tL2, const float* tDlt, int rem, int hh) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ql = tile_slot(r, hh);
        const float pr = ql < rem ? __builtin_amdgcn_exp2f(s[r] * c1 - tL2[ql]) : 0.f;
        if (WANT_DS) dp[r] = pr * (dp[r] - tDlt[ql]);
        if (WANT_P) s[r] = pr;
    }
}
// The same without the row mask and with the per-row values fetched as four 16-byte LDS reads each (the 16 query rows of a lane are
// four runs of four: rows 8 j + 4 hh + e): for callers that stage log2-domain log-sum-exp values of ROWS BEYOND THE SEQUENCE as
// LSE_MASKED (exp2(s c1 - 1e30) = 0: P = dS = 0 there with no compare / select).  Per 32 x 32 tile the masked form issues 32
// ds_read_b32 + 16 compares + 16 selects that this one does not (round 5: the long-sequence kernels are VALU / LDS-issue bound).
constexpr float LSE_MASKED = 1e30f;
template <bool WANT_P, bool WANT_DS, bool VEC = true>
__device__ __forceinline__ void p_ds_rows_nomask(f32x16& s, f32x16& dp, float c1, const float* tL2, const float* tDlt, int hh) {
    if (!VEC) {      // one value per read (tried on the capped two-phase fused kernels, which have no registers for the 16-byte form:
                     // they spill 8-14 registers with it and 10 even with this form, so they keep the masked p_ds_rows)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ql = tile_slot(r, hh);
            const float pr = __builtin_amdgcn_exp2f(s[r] * c1 - tL2[ql]);
            if (WANT_DS) dp[r] = pr * (dp[r] - tDlt[ql]);
            if (WANT_P) s[r] = pr;
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(tL2 + 8 * j + 4 * hh);
        f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
        if (WANT_DS) d4 = *reinterpret_cast<const f32x4*>(tDlt + 8 * j + 4 * hh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * j + e;
            const float pr = __builtin_amdgcn_exp2f(s[r] * c1 - l4[e]);
            if (WANT_DS) dp[r] = pr * (dp[r] - d4[e]);
            if (WANT_P) s[r] = pr;
        }
    }
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int base) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(a[base + e]);
    return o;
}

// Store a transposed accumulator tile: acc[di] holds X^T[d][row] with lane <-> row, regs <-> d.
// Workgroup -> (batch, head).  The dispatcher deals consecutive workgroups to the 8 XCDs in turn, so with (b, h) = the plain grid index
// an XCD gets every 8th head: a 128-byte slice of every row of every sample, its neighbours' slices going through seven other L2s.
// Here the XCDs share a window of 8 consecutive samples and each takes ONE WHOLE sample of it, its heads in consecutive slots
// (workgroup 8 s + x -> sample 8 (s / H) + x, head s mod H; the B mod 8 last samples keep the plain order).  Same bytes and the same
// L2 hit rate as before; 40 % fewer read-credit and tag stalls at the L2 channels (profiles/r6_attention_xcd_pmc.txt).  Measured on
// the 64-row shapes (1024 x 16 heads, A B A B on one box, us plain -> this): 64 x 64 forward 138 -> 120, backward 315 -> 298; 64 x 77
// forward 147 -> 130, backward 372 -> 343; 77 x 77 forward 176 -> 168, backward 418 -> 394; 256-row shapes +-1 %.  The group size was
// swept (1 ... B H / 8 pairs per XCD turn: flat optimum at 8-32 pairs, "contiguous eighths" 3 % behind); the same idea LOSES 9-14 % on the
// row-contiguous LayerNorm kernels (profiles/r6_notes.txt section 8).
__device__ __forceinline__ void bh_of(int64_t lin, int64_t H, int64_t B, int64_t& b, int64_t& h) {
    const int64_t n8 = (B >> 3) * 8 * H;                // whole windows of 8 samples
    if (lin < n8) {
        const int64_t s = lin >> 3, x = lin & 7;        // slot s of XCD x
        lin = ((s / H) * 8 + x) * H + s % H;
    }
    b = lin / H;
    h = lin - b * H;
}

template <int HD>
__device__ __forceinline__ void store_rows(bf16* dst_row, const f32x16 (&acc)[HD / 32], float mul, int lane) {
    const int hh = lane >> 5;
#pragma unroll
    for (int di = 0; di < HD / 32; ++di)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[di][4 * g + e] * mul);
            st_bf16x4(dst_row + di * 32 + 8 * g + 4 * hh, o);
        }
}

// (Measured and dropped: capping this kernel at 128 VGPRs -- 117 without a spill at head_dim 64, 4 waves / SIMD instead of 3 --
// made the 2-wave instantiation 15 % SLOWER in the step, 142.7 -> 164.9 us; the allocation the compiler picks on its own keeps
// more of a phase's loads in flight.  profiles/r2_attention_two_phase_bwd.txt.)
template <int HD, int MAXIT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(md_attn_args p) {
    constexpr int PK = (HD + 8) * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STG * PK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x, nw = nthreads >> 6;
    const int hh = lane >> 5;
    const float c1 = p.scale * LOG2E;
    int64_t b = blockIdx.z, h = blockIdx.y;
    if (gridDim.x == 1) bh_of(blockIdx.y + (int64_t)gridDim.y * blockIdx.z, gridDim.y, gridDim.z, b, h);
    const int64_t q = ((int64_t)blockIdx.x * nw + wave) * 32 + (lane & 31);
    const bool qvalid = q < p.Sq;
    const bf16* Q = reinterpret_cast<const bf16*>(p.q) + b * p.sq + h * p.hsq;
    const bf16* K = reinterpret_cast<const bf16*>(p.k) + b * p.sk + h * p.hsk;
    const bf16* V = reinterpret_cast<const bf16*>(p.v) + b * p.sv + h * p.hsv;

    bf16x8 qf[HD / 16];
#pragma unroll
    for (int s = 0; s < HD / 16; ++s) {
        if (qvalid)
            qf[s] = ld_bf16x8(Q + q * p.ldq + s * 16 + hh * 8);
        else
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[s][e] = f2bf(0.f);
    }
    f32x16 oacc[HD / 32];
#pragma unroll
    for (int di = 0; di < HD / 32; ++di)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[di][r] = 0.f;
    float m = -1e30f, l = 0.f;

    for (int64_t kbase = 0; kbase < p.Skv; kbase += STG) {
        __syncthreads();
        const int nst = (int)((p.Skv - kbase >= STG) ? STG : ((p.Skv - kbase + 31) / 32) * 32);
        stage_pair<HD, MAXIT>(smem, smem + STG * PK, PK, K, p.ldk, V, p.ldv, kbase, p.Skv, nst, tid, nthreads);
        __syncthreads();
      for (int sub = 0; sub < STG / 32 && kbase + sub * 32 < p.Skv; ++sub) {
        const int64_t key0 = kbase + sub * 32;
        const unsigned char* sK = smem + sub * 32 * PK;
        const unsigned char* sV = smem + STG * PK + sub * 32 * PK;
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < HD / 16; ++s)
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(sK, PK, s * 16, lane), qf[s], sacc, 0, 0, 0);
        // online softmax in the log2 domain (m, m_new, the scores: all x log2 e): one multiply + v_exp_f32 per element, a 32-bit
        // compare + select for the key mask (the first version: a 64-bit index compare per element and exp through __expf)
        const int rem = (int)(p.Skv - key0);
        float tmax = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sacc[r] = tile_slot(r, hh) < rem ? sacc[r] * c1 : -1e30f;
            tmax = fmaxf(tmax, sacc[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pr = __builtin_amdgcn_exp2f(sacc[r] - m_new);
            sacc[r] = pr;
            psum += pr;
        }
        psum += __shfl_xor(psum, 32, 64);
        l = l * alpha + psum;
        m = m_new;
#pragma unroll
        for (int di = 0; di < HD / 32; ++di)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[di][r] *= alpha;
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            const bf16x8 pf = pack8(sacc, 8 * sp);
#pragma unroll
            for (int di = 0; di < HD / 32; ++di)
                oacc[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    tr_frag(sV, PK, 16 * sp + 4 * hh, 16 * sp + 8 + 4 * hh, di * 32, lane), pf, oacc[di], 0, 0, 0);
        }
      }
    }
    if (qvalid) {
        bf16* O = reinterpret_cast<bf16*>(p.o) + b * p.so + h * p.hso + q * p.ldo;
        store_rows<HD>(O, oacc, 1.f / l, lane);
        if (lane < 32 && p.lse) reinterpret_cast<float*>(p.lse)[(b * p.H + h) * p.Sq + q] = (m + __log2f(l)) * 0.6931471805599453f;   // natural log
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Streaming pair for LONG sequences (Sq or Skv > 256: the res-512 patch mixer, 1024 tokens, and its 1024 x 77 cross-attention;
// /root/reference/configs/res_512_pretrain.yaml:24, utils.py:188-193).  Same arithmetic and the same two launches as the pair above
// (dQ: every wave owns 32 query rows and walks the keys; dK / dV: every wave owns 32 key rows and walks the queries), re-organised
// around what the 32-row phases cost there -- a global -> LDS round trip with nothing in flight under each phase's 12-16 MFMAs:
//   * the walked side arrives in CHUNKS of SCH rows (128 for workgroups of 5-8 waves: 4 tiles per barrier pair instead of 1; 64 / 32
//     for smaller workgroups, so that a thread never stages more than 4 x 16 bytes per matrix -- the staging registers are what
//     the dK / dV kernel, with its four accumulators, has least of);
//   * chunk c + 1 is requested into registers right after chunk c has been written to LDS and lands while chunk c is being
//     multiplied (the compiler's waits for those registers sit at the next LDS write, not in the tile loop: MFMA and ds_read do
//     not depend on them);
//   * workgroups of up to 8 waves (256 owned rows), so a chunk is fetched once per 256 rows instead of once per 128.
// The dK / dV kernel also prefetches the chunk's log-sum-exp / delta values (one float each per row).
// ---------------------------------------------------------------------------------------------------------------------

template <int HD, int MAXIT>
struct ChunkRegs {
    u32x4 a[MAXIT], b[MAXIT];
};
// global -> registers: rows row0 .. row0 + SCH of A and B (rows >= nrows: zeros); every load of the chunk in flight at once
template <int HD, int SCH, int MAXIT>
__device__ __forceinline__ void chunk_load(ChunkRegs<HD, MAXIT>& R, const bf16* A, int64_t lda, const bf16* B, int64_t ldb, int64_t row0,
                                           int64_t nrows, int tid, int nthreads) {
    constexpr int CPR = HD / 8;
    const u32x4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int task = tid + it * nthreads, r = task / CPR, c = task % CPR;
        const bool ok = task < SCH * CPR && row0 + r < nrows;
        R.a[it] = ok ? *reinterpret_cast<const u32x4*>(A + (row0 + r) * lda + c * 8) : z4;
        R.b[it] = ok ? *reinterpret_cast<const u32x4*>(B + (row0 + r) * ldb + c * 8) : z4;
    }
}
template <int HD, int SCH, int MAXIT>
__device__ __forceinline__ void chunk_store(const ChunkRegs<HD, MAXIT>& R, unsigned char* tileA, unsigned char* tileB, int pitch, int tid,
                                            int nthreads) {
    constexpr int CPR = HD / 8;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int task = tid + it * nthreads, r = task / CPR, c = task % CPR;
        if (task < SCH * CPR) {
            *reinterpret_cast<u32x4*>(tileA + r * pitch + c * 16) = R.a[it];
            *reinterpret_cast<u32x4*>(tileB + r * pitch + c * 16) = R.b[it];
        }
    }
}

// Forward for long key sequences on the same plan (chunks of SCH keys, the next chunk's K / V in flight under the current one's
// tiles, up to 8 waves of 32 query rows): the online-softmax tile loop is attn_fwd_kernel's.
template <int HD, int SCH, int MAXIT>
__global__ __launch_bounds__(512) void attn_fwd_stream_kernel(md_attn_args p) {      // (capped at 128 VGPRs it spills 10-50 registers)
    constexpr int PK = (HD + 8) * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * SCH * PK];
    unsigned char* sKc = smem;
    unsigned char* sVc = smem + SCH * PK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x, nw = nthreads >> 6;
    const int hh = lane >> 5;
    const float c1 = p.scale * LOG2E;
    const int64_t b = blockIdx.z, h = blockIdx.y;
    const int64_t q = ((int64_t)blockIdx.x * nw + wave) * 32 + (lane & 31);
    const bool qvalid = q < p.Sq;
    const bf16* Q = reinterpret_cast<const bf16*>(p.q) + b * p.sq + h * p.hsq;
    const bf16* K = reinterpret_cast<const bf16*>(p.k) + b * p.sk + h * p.hsk;
    const bf16* V = reinterpret_cast<const bf16*>(p.v) + b * p.sv + h * p.hsv;

    ChunkRegs<HD, MAXIT> R;
    chunk_load<HD, SCH, MAXIT>(R, K, p.ldk, V, p.ldv, 0, p.Skv, tid, nthreads);
    bf16x8 qf[HD / 16];
#pragma unroll
    for (int s = 0; s < HD / 16; ++s) {
        if (qvalid)
            qf[s] = ld_bf16x8(Q + q * p.ldq + s * 16 + hh * 8);
        else
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[s][e] = f2bf(0.f);
    }
    f32x16 oacc[HD / 32];
#pragma unroll
    for (int di = 0; di < HD / 32; ++di)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[di][r] = 0.f;
    float m = -1e30f, l = 0.f;

    for (int64_t kbase = 0; kbase < p.Skv; kbase += SCH) {
        __syncthreads();
        chunk_store<HD, SCH, MAXIT>(R, sKc, sVc, PK, tid, nthreads);
        __syncthreads();
        if (kbase + SCH < p.Skv) chunk_load<HD, SCH, MAXIT>(R, K, p.ldk, V, p.ldv, kbase + SCH, p.Skv, tid, nthreads);
        auto tile = [&](int sub, int rem) {
            const unsigned char* sK = sKc + sub * 32 * PK;
            const unsigned char* sV = sVc + sub * 32 * PK;
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < HD / 16; ++s)
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(sK, PK, s * 16, lane), qf[s], sacc, 0, 0, 0);
            float tmax = -1e30f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] = tile_slot(r, hh) < rem ? sacc[r] * c1 : -1e30f;
                tmax = fmaxf(tmax, sacc[r]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m, tmax);
            const float alpha = __builtin_amdgcn_exp2f(m - m_new);
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pr = __builtin_amdgcn_exp2f(sacc[r] - m_new);
                sacc[r] = pr;
                psum += pr;
            }
            psum += __shfl_xor(psum, 32, 64);
            l = l * alpha + psum;
            m = m_new;
#pragma unroll
            for (int di = 0; di < HD / 32; ++di)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[di][r] *= alpha;
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const bf16x8 pf = pack8(sacc, 8 * sp);
#pragma unroll
                for (int di = 0; di < HD / 32; ++di)
                    oacc[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        tr_frag(sV, PK, 16 * sp + 4 * hh, 16 * sp + 8 + 4 * hh, di * 32, lane), pf, oacc[di], 0, 0, 0);
            }
        };
        if (kbase + SCH <= p.Skv) {            // a whole chunk: the key mask folds away (rem = 32 is a constant)
#pragma unroll 1
            for (int sub = 0; sub < SCH / 32; ++sub) tile(sub, 32);
        } else {
            for (int sub = 0; sub < SCH / 32 && kbase + sub * 32 < p.Skv; ++sub) tile(sub, (int)(p.Skv - kbase - sub * 32));
        }
    }
    if (qvalid) {
        bf16* O = reinterpret_cast<bf16*>(p.o) + b * p.so + h * p.hso + q * p.ldo;
        store_rows<HD>(O, oacc, 1.f / l, lane);
        if (lane < 32 && p.lse) reinterpret_cast<float*>(p.lse)[(b * p.H + h) * p.Sq + q] = (m + __log2f(l)) * 0.6931471805599453f;   // natural log
    }
}

template <int HD, int SCH, int MAXIT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu((SCH == 64 && MAXIT == 2) ? 3 : 2)))   // 4-wave form: 168 VGPRs, 3 workgroups / CU
void attn_bwd_dq_stream_kernel(md_attn_args p) {
    constexpr int PK = (HD + 8) * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * SCH * PK];
    unsigned char* sK = smem;
    unsigned char* sV = smem + SCH * PK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x, nw = nthreads >> 6;
    const int hh = lane >> 5;
    const int64_t b = blockIdx.z, h = blockIdx.y;
    const int64_t q = ((int64_t)blockIdx.x * nw + wave) * 32 + (lane & 31);
    const bool qvalid = q < p.Sq;
    const bf16* Q = reinterpret_cast<const bf16*>(p.q) + b * p.sq + h * p.hsq;
    const bf16* K = reinterpret_cast<const bf16*>(p.k) + b * p.sk + h * p.hsk;
    const bf16* V = reinterpret_cast<const bf16*>(p.v) + b * p.sv + h * p.hsv;
    const bf16* dO = reinterpret_cast<const bf16*>(p.d_o) + b * p.sdo + h * p.hsdo;

    ChunkRegs<HD, MAXIT> R;
    chunk_load<HD, SCH, MAXIT>(R, K, p.ldk, V, p.ldv, 0, p.Skv, tid, nthreads);       // chunk 0 on its way before the row set-up
    bf16x8 qf[HD / 16], dof[HD / 16];
#pragma unroll
    for (int s = 0; s < HD / 16; ++s) {
        if (qvalid) {
            qf[s] = ld_bf16x8(Q + q * p.ldq + s * 16 + hh * 8);
            dof[s] = ld_bf16x8(dO + q * p.lddo + s * 16 + hh * 8);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                qf[s][e] = f2bf(0.f);
                dof[s][e] = f2bf(0.f);
            }
        }
    }
    const float lse = qvalid ? reinterpret_cast<const float*>(p.lse)[(b * p.H + h) * p.Sq + q] * LOG2E : 0.f;   // log2 domain (ds_cols)
    const float c1 = p.scale * LOG2E;
    float dlt = 0.f;      // delta[q] = sum_d dO[q, d] * O[q, d]: this lane holds half of row q, the partner lane the rest
    if (qvalid) {
        const bf16* O = reinterpret_cast<const bf16*>(p.o) + b * p.so + h * p.hso + q * p.ldo;
#pragma unroll
        for (int s = 0; s < HD / 16; ++s) {
            const bf16x8 ov = ld_bf16x8(O + s * 16 + hh * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) dlt += bf2f(ov[e]) * bf2f(dof[s][e]);
        }
    }
    dlt += __shfl_xor(dlt, 32, 64);
    if (qvalid && lane < 32) reinterpret_cast<float*>(p.delta)[(b * p.H + h) * p.Sq + q] = dlt;   // for the dK / dV kernel
    f32x16 dqacc[HD / 32];
#pragma unroll
    for (int di = 0; di < HD / 32; ++di)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[di][r] = 0.f;

    for (int64_t kbase = 0; kbase < p.Skv; kbase += SCH) {
        __syncthreads();                                   // every wave is done with the previous chunk's tiles
        chunk_store<HD, SCH, MAXIT>(R, sK, sV, PK, tid, nthreads);
        __syncthreads();
        if (kbase + SCH < p.Skv) chunk_load<HD, SCH, MAXIT>(R, K, p.ldk, V, p.ldv, kbase + SCH, p.Skv, tid, nthreads);   // lands under the tile loop
        auto tile = [&](int sub, int rem) {
            const unsigned char* tK = sK + sub * 32 * PK;
            const unsigned char* tV = sV + sub * 32 * PK;
            f32x16 sacc, dpacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] = 0.f;
                dpacc[r] = 0.f;
            }
#pragma unroll
            for (int s = 0; s < HD / 16; ++s) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(tK, PK, s * 16, lane), qf[s], sacc, 0, 0, 0);
                dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(tV, PK, s * 16, lane), dof[s], dpacc, 0, 0, 0);
            }
            ds_cols(sacc, dpacc, c1, lse, dlt, rem, hh);   // dS^T / scale
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const bf16x8 dsf = pack8(sacc, 8 * sp);
#pragma unroll
                for (int di = 0; di < HD / 32; ++di)
                    dqacc[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        tr_frag(tK, PK, 16 * sp + 4 * hh, 16 * sp + 8 + 4 * hh, di * 32, lane), dsf, dqacc[di], 0, 0, 0);
            }
        };
        if (SCH == 128 && kbase + SCH <= p.Skv) {
            // a whole chunk (every chunk but a ragged last one): no masks, and ONE basic block of SCH / 32 tiles -- the scheduler
            // can place the next tile's fragment reads and S / dP products under this tile's exponentials (S = 1024: 4376 -> 4218 us;
            // costs 64 VGPRs, so the small-workgroup forms -- few keys, latency-bound, 3 waves / SIMD -- keep the rolled loop)
#pragma unroll
            for (int sub = 0; sub < SCH / 32; ++sub) tile(sub, 32);
        } else {
            for (int sub = 0; sub < SCH / 32 && kbase + sub * 32 < p.Skv; ++sub) tile(sub, (int)(p.Skv - kbase - sub * 32));
        }
    }
    if (qvalid) {
        bf16* dQ = reinterpret_cast<bf16*>(p.dq) + b * p.sdq + h * p.hsdq + q * p.lddq;
        store_rows<HD>(dQ, dqacc, p.scale, lane);
    }
}

template <int HD, int SCH, int MAXIT>
__global__ __launch_bounds__(512) void attn_bwd_dkv_stream_kernel(md_attn_args p) {
    constexpr int PK = (HD + 8) * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * SCH * PK + 2 * SCH * 4];
    unsigned char* sQ = smem;
    unsigned char* sdO = smem + SCH * PK;
    float* sLseAll = reinterpret_cast<float*>(smem + 2 * SCH * PK);
    float* sDltAll = sLseAll + SCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x, nw = nthreads >> 6;
    const int hh = lane >> 5;
    const float c1 = p.scale * LOG2E;        // scores -> log2 domain (p_ds_rows)
    const int64_t b = blockIdx.z, h = blockIdx.y;
    const int64_t key = ((int64_t)blockIdx.x * nw + wave) * 32 + (lane & 31);
    const bool kvalid = key < p.Skv;
    const bf16* Q = reinterpret_cast<const bf16*>(p.q) + b * p.sq + h * p.hsq;
    const bf16* K = reinterpret_cast<const bf16*>(p.k) + b * p.sk + h * p.hsk;
    const bf16* V = reinterpret_cast<const bf16*>(p.v) + b * p.sv + h * p.hsv;
    const bf16* dO = reinterpret_cast<const bf16*>(p.d_o) + b * p.sdo + h * p.hsdo;
    const float* LSE = reinterpret_cast<const float*>(p.lse) + (b * p.H + h) * p.Sq;
    const float* DLT = reinterpret_cast<const float*>(p.delta) + (b * p.H + h) * p.Sq;

    ChunkRegs<HD, MAXIT> R;
    float rl = 0.f, rd = 0.f;                 // this thread's row of the chunk's log-sum-exp / delta (threads < SCH)
    chunk_load<HD, SCH, MAXIT>(R, Q, p.ldq, dO, p.lddo, 0, p.Sq, tid, nthreads);
    if (tid < SCH) {
        rl = tid < p.Sq ? LSE[tid] * LOG2E : LSE_MASKED;
        rd = tid < p.Sq ? DLT[tid] : 0.f;
    }
    bf16x8 kf[HD / 16], vf[HD / 16];
#pragma unroll
    for (int s = 0; s < HD / 16; ++s) {
        if (kvalid) {
            kf[s] = ld_bf16x8(K + key * p.ldk + s * 16 + hh * 8);
            vf[s] = ld_bf16x8(V + key * p.ldv + s * 16 + hh * 8);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                kf[s][e] = f2bf(0.f);
                vf[s][e] = f2bf(0.f);
            }
        }
    }
    f32x16 dkacc[HD / 32], dvacc[HD / 32];
#pragma unroll
    for (int di = 0; di < HD / 32; ++di)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dkacc[di][r] = 0.f;
            dvacc[di][r] = 0.f;
        }

    for (int64_t qbase = 0; qbase < p.Sq; qbase += SCH) {
        __syncthreads();
        chunk_store<HD, SCH, MAXIT>(R, sQ, sdO, PK, tid, nthreads);
        if (tid < SCH) {
            sLseAll[tid] = rl;                  // log2 domain; rows beyond Sq: LSE_MASKED (p_ds_rows_nomask: P = dS = 0 there)
            sDltAll[tid] = rd;
        }
        __syncthreads();
        if (qbase + SCH < p.Sq) {
            chunk_load<HD, SCH, MAXIT>(R, Q, p.ldq, dO, p.lddo, qbase + SCH, p.Sq, tid, nthreads);
            const int64_t rn = qbase + SCH + tid;
            const bool ok = tid < SCH && rn < p.Sq;
            rl = ok ? LSE[rn] * LOG2E : LSE_MASKED;
            rd = ok ? DLT[rn] : 0.f;
        }
        auto tile = [&](int sub, int rem) {
            const unsigned char* tQ = sQ + sub * 32 * PK;
            const unsigned char* tdO = sdO + sub * 32 * PK;
            const float* tLse = sLseAll + sub * 32;
            const float* tDlt = sDltAll + sub * 32;
            f32x16 sacc, dpacc;  // S[q][key], dP[q][key]: lane <-> key, regs <-> q
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] = 0.f;
                dpacc[r] = 0.f;
            }
#pragma unroll
            for (int s = 0; s < HD / 16; ++s) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(tQ, PK, s * 16, lane), kf[s], sacc, 0, 0, 0);
                dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(tdO, PK, s * 16, lane), vf[s], dpacc, 0, 0, 0);
            }
            (void)rem;
            p_ds_rows_nomask<true, true>(sacc, dpacc, c1, tLse, tDlt, hh);   // P, dS / scale (rows beyond Sq: staged as LSE_MASKED)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const bf16x8 pf = pack8(sacc, 8 * sp);
                const bf16x8 dsf = pack8(dpacc, 8 * sp);
#pragma unroll
                for (int di = 0; di < HD / 32; ++di) {
                    dvacc[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        tr_frag(tdO, PK, 16 * sp + 4 * hh, 16 * sp + 8 + 4 * hh, di * 32, lane), pf, dvacc[di], 0, 0, 0);
                    dkacc[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        tr_frag(tQ, PK, 16 * sp + 4 * hh, 16 * sp + 8 + 4 * hh, di * 32, lane), dsf, dkacc[di], 0, 0, 0);
                }
            }
        };
        // (One copy of the tile body, rolled: a mask-free copy for whole chunks, or two / four tiles per basic block as in the dQ kernel,
        // pushes this kernel -- four accumulators -- past the 256-register budget: 8-126 spilled registers.)
        for (int sub = 0; sub < SCH / 32 && qbase + sub * 32 < p.Sq; ++sub) tile(sub, (int)(p.Sq - qbase - sub * 32));
    }
    if (kvalid) {
        bf16* dK = reinterpret_cast<bf16*>(p.dk) + b * p.sdk + h * p.hsdk + key * p.lddk;
        bf16* dV = reinterpret_cast<bf16*>(p.dv) + b * p.sdv + h * p.hsdv + key * p.lddv;
        store_rows<HD>(dK, dkacc, p.scale, lane);
        store_rows<HD>(dV, dvacc, 1.f, lane);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused backward for the training shapes (Sq, Skv <= 256): ONE workgroup per (batch, head) stages Q, dO, K, V into LDS once,
// computes delta = rowsum(dO * O) in place, and every wave then plays both roles of the two kernels above on its 32 rows —
// dQ for query rows w * 32.. (looping over the key tiles in LDS) and dK / dV for key rows w * 32.. (looping over the query
// tiles) — so each of Q, K, V, dO, O is read from HBM once and delta / lse never round-trip through memory between two
// launches (the split kernels read Q, K, V, dO twice: 12 tensor passes against 8; these shapes are HBM-bound).
// SQP / SKP = padded row counts (multiples of 32) the LDS image is sized for.
// ---------------------------------------------------------------------------------------------------------------------
template <int HD, int SQP, int SKP>
__global__ __launch_bounds__((SQP > SKP ? SQP : SKP) * 2) void attn_bwd_fused_kernel(md_attn_args p) {
    constexpr int PK = (HD + 8) * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[(2 * SQP + 2 * SKP) * PK + 2 * SQP * 4];
    unsigned char* sQ = smem;
    unsigned char* sdO = smem + SQP * PK;
    unsigned char* sK = smem + 2 * SQP * PK;
    unsigned char* sV = sK + SKP * PK;
    float* sLse = reinterpret_cast<float*>(sV + SKP * PK);
    float* sDlt = sLse + SQP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
    const int hh = lane >> 5;
    const float c1 = p.scale * LOG2E;        // scores -> log2 domain (ds_cols / p_ds_rows)
    int64_t b, h;
    bh_of(blockIdx.x + (int64_t)gridDim.x * blockIdx.y, gridDim.x, gridDim.y, b, h);
    const bf16* Q = reinterpret_cast<const bf16*>(p.q) + b * p.sq + h * p.hsq;
    const bf16* K = reinterpret_cast<const bf16*>(p.k) + b * p.sk + h * p.hsk;
    const bf16* V = reinterpret_cast<const bf16*>(p.v) + b * p.sv + h * p.hsv;
    const bf16* dO = reinterpret_cast<const bf16*>(p.d_o) + b * p.sdo + h * p.hsdo;
    const bf16* O = reinterpret_cast<const bf16*>(p.o) + b * p.so + h * p.hso;
    const int nq32 = (int)((p.Sq + 31) / 32), nk32 = (int)((p.Skv + 31) / 32);

    // Staging: EVERY global load of the workgroup is issued before the first LDS write -- Q, dO, O (3 x ITQ) and K, V (2 x ITK)
    // 16-byte chunks per thread, all in flight together.  (The first version staged tile by tile in loops with run-time trip
    // counts: load -> wait -> ds_write per iteration, i.e. ~20 dependent HBM round trips per workgroup, and a workgroup lived
    // 25-30 us for ~3 us of arithmetic.)  delta = rowsum(dO * O) is formed from the staged registers (dO is read once).
    {
        constexpr int CPR = HD / 8;
        constexpr int NT = (SQP > SKP ? SQP : SKP) * 2;                 // = blockDim.x
        constexpr int ITQ = (SQP * CPR + NT - 1) / NT, ITK = (SKP * CPR + NT - 1) / NT;
        const float* LSE = reinterpret_cast<const float*>(p.lse) + (b * p.H + h) * p.Sq;
        u32x4 rq[ITQ], rdo[ITQ], ro[ITQ], rk[ITK], rv[ITK];
        float rl[ITQ];
        const u32x4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int it = 0; it < ITQ; ++it) {
            const int task = tid + it * NT, r = task / CPR, c = task % CPR;
            const bool ok = task < SQP * CPR && r < p.Sq;
            rq[it] = ok ? *reinterpret_cast<const u32x4*>(Q + (int64_t)r * p.ldq + c * 8) : z4;
            rdo[it] = ok ? *reinterpret_cast<const u32x4*>(dO + (int64_t)r * p.lddo + c * 8) : z4;
            ro[it] = ok ? *reinterpret_cast<const u32x4*>(O + (int64_t)r * p.ldo + c * 8) : z4;
            rl[it] = (ok && c == 0) ? LSE[r] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < ITK; ++it) {
            const int task = tid + it * NT, r = task / CPR, c = task % CPR;
            const bool ok = task < SKP * CPR && r < p.Skv;
            rk[it] = ok ? *reinterpret_cast<const u32x4*>(K + (int64_t)r * p.ldk + c * 8) : z4;
            rv[it] = ok ? *reinterpret_cast<const u32x4*>(V + (int64_t)r * p.ldv + c * 8) : z4;
        }
#pragma unroll
        for (int it = 0; it < ITQ; ++it) {
            const int task = tid + it * NT, r = task / CPR, c = task % CPR;
            if (task < SQP * CPR) {
                *reinterpret_cast<u32x4*>(sQ + r * PK + c * 16) = rq[it];
                *reinterpret_cast<u32x4*>(sdO + r * PK + c * 16) = rdo[it];
            }
            const bf16x8 g = __builtin_bit_cast(bf16x8, rdo[it]), o = __builtin_bit_cast(bf16x8, ro[it]);
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) d += bf2f(o[e]) * bf2f(g[e]);
#pragma unroll
            for (int s2 = CPR / 2; s2 > 0; s2 >>= 1) d += __shfl_xor(d, s2, 64);      // the CPR lanes of a row are adjacent (NT % CPR == 0)
            if (c == 0 && task < SQP * CPR) {
                sDlt[r] = d;
                sLse[r] = rl[it] * LOG2E;          // log2 domain (ds_cols / p_ds_rows)
            }
        }
#pragma unroll
        for (int it = 0; it < ITK; ++it) {
            const int task = tid + it * NT, r = task / CPR, c = task % CPR;
            if (task < SKP * CPR) {
                *reinterpret_cast<u32x4*>(sK + r * PK + c * 16) = rk[it];
                *reinterpret_cast<u32x4*>(sV + r * PK + c * 16) = rv[it];
            }
        }
    }
    __syncthreads();

    // ---------------- role 1: dQ for query rows wave * 32 ..
    if (wave < nq32) {
        const unsigned char* myQ = sQ + wave * 32 * PK;
        const unsigned char* mydO = sdO + wave * 32 * PK;
        const int64_t q = (int64_t)wave * 32 + (lane & 31);
        bf16x8 qf[HD / 16], dof[HD / 16];
#pragma unroll
        for (int s = 0; s < HD / 16; ++s) {
            qf[s] = row_frag(myQ, PK, s * 16, lane);
            dof[s] = row_frag(mydO, PK, s * 16, lane);
        }
        const float lse = sLse[wave * 32 + (lane & 31)], dlt = sDlt[wave * 32 + (lane & 31)];
        f32x16 dqacc[HD / 32];
#pragma unroll
        for (int di = 0; di < HD / 32; ++di)
#pragma unroll
            for (int r = 0; r < 16; ++r) dqacc[di][r] = 0.f;
        for (int j = 0; j < nk32; ++j) {
            const unsigned char* tK = sK + j * 32 * PK;
            const unsigned char* tV = sV + j * 32 * PK;
            f32x16 sacc, dpacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] = 0.f;
                dpacc[r] = 0.f;
            }
#pragma unroll
            for (int s = 0; s < HD / 16; ++s) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(tK, PK, s * 16, lane), qf[s], sacc, 0, 0, 0);
                dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(tV, PK, s * 16, lane), dof[s], dpacc, 0, 0, 0);
            }
            ds_cols(sacc, dpacc, c1, lse, dlt, (int)p.Skv - j * 32, hh);   // dS^T / scale
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const bf16x8 dsf = pack8(sacc, 8 * sp);
#pragma unroll
                for (int di = 0; di < HD / 32; ++di)
                    dqacc[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        tr_frag(tK, PK, 16 * sp + 4 * hh, 16 * sp + 8 + 4 * hh, di * 32, lane), dsf, dqacc[di], 0, 0, 0);
            }
        }
        if (q < p.Sq) {
            bf16* dQ = reinterpret_cast<bf16*>(p.dq) + b * p.sdq + h * p.hsdq + q * p.lddq;
            store_rows<HD>(dQ, dqacc, p.scale, lane);
        }
    }
    // ---------------- role 2: dK, dV for key rows wave * 32 ..
    if (wave < nk32) {
        const unsigned char* myK = sK + wave * 32 * PK;
        const unsigned char* myV = sV + wave * 32 * PK;
        const int64_t key = (int64_t)wave * 32 + (lane & 31);
        bf16x8 kf[HD / 16], vf[HD / 16];
#pragma unroll
        for (int s = 0; s < HD / 16; ++s) {
            kf[s] = row_frag(myK, PK, s * 16, lane);
            vf[s] = row_frag(myV, PK, s * 16, lane);
        }
        f32x16 dkacc[HD / 32], dvacc[HD / 32];
#pragma unroll
        for (int di = 0; di < HD / 32; ++di)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dkacc[di][r] = 0.f;
                dvacc[di][r] = 0.f;
            }
        for (int i = 0; i < nq32; ++i) {
            const unsigned char* tQ = sQ + i * 32 * PK;
            const unsigned char* tdO = sdO + i * 32 * PK;
            const float* tLse = sLse + i * 32;
            const float* tDlt = sDlt + i * 32;
            f32x16 sacc, dpacc;  // S[q][key], dP[q][key]: lane <-> key, regs <-> q
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] = 0.f;
                dpacc[r] = 0.f;
            }
#pragma unroll
            for (int s = 0; s < HD / 16; ++s) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(tQ, PK, s * 16, lane), kf[s], sacc, 0, 0, 0);
                dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(tdO, PK, s * 16, lane), vf[s], dpacc, 0, 0, 0);
            }
            p_ds_rows<true, true>(sacc, dpacc, c1, tLse, tDlt, (int)p.Sq - i * 32, hh);   // P, dS / scale
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const bf16x8 pf = pack8(sacc, 8 * sp);
                const bf16x8 dsf = pack8(dpacc, 8 * sp);
#pragma unroll
                for (int di = 0; di < HD / 32; ++di) {
                    dvacc[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        tr_frag(tdO, PK, 16 * sp + 4 * hh, 16 * sp + 8 + 4 * hh, di * 32, lane), pf, dvacc[di], 0, 0, 0);
                    dkacc[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        tr_frag(tQ, PK, 16 * sp + 4 * hh, 16 * sp + 8 + 4 * hh, di * 32, lane), dsf, dkacc[di], 0, 0, 0);
                }
            }
        }
        if (key < p.Skv) {
            bf16* dK = reinterpret_cast<bf16*>(p.dk) + b * p.sdk + h * p.hsdk + key * p.lddk;
            bf16* dV = reinterpret_cast<bf16*>(p.dv) + b * p.sdv + h * p.hsdv + key * p.lddv;
            store_rows<HD>(dK, dkacc, p.scale, lane);
            store_rows<HD>(dV, dvacc, 1.f, lane);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Two-phase fused backward: the same arithmetic and the same one-launch-per-(batch, head) traffic as attn_bwd_fused_kernel,
// with HALF its LDS image and a register budget of 3 waves / SIMD.  The single-phase kernel keeps Q, dO, K and V in LDS
// together (37-47 KiB at S <= 96) and needs 220 VGPRs, which leaves a CU with 6-8 resident waves; at these sizes a workgroup is
// a serial chain (load -> LDS -> ~60 dependent MFMAs -> store) whose only overlap is with OTHER workgroups, so throughput
// follows the number of workgroups a CU can hold.  Here ONE pair of tiles is alive at a time:
//   P1  Q, dO -> LDS (and delta = rowsum(dO * O), lse);  P2  every wave lifts its own 32 query rows into registers;
//   P3  K, V overwrite the image;  P4  dQ (role 1), then every wave lifts its own 32 key rows;
//   P5  the waves write their query rows back (the registers are the only copy: nothing is re-read from HBM);
//   P6  dK, dV (role 2).
// LDS: 2 * max(SQP, SKP) * (HD + 8) * 2 bytes + lse / delta (18.9 KiB at 64 x 64, 28.2 KiB with 77 keys).
// ---------------------------------------------------------------------------------------------------------------------
// SPLIT2 (the 256-row buckets, 8-wave workgroups): role 2 runs as two passes over the query tiles — dV from P alone, then dK
// from dS — so that only ONE 32 x HD accumulator pair is alive at a time and the kernel fits 128 VGPRs: 16 wave slots per CU
// hold two of those workgroups (75.8 KiB of LDS each), where the single-phase kernel (190 VGPRs, 149.5 KiB) holds one and has
// nothing to overlap its load and store phases with.  Price: S = Q K^T and the exponentials of role 2 are formed twice
// (20 instead of 16 MFMAs per tile pair).
// ---------------------------------------------------------------------------------------------------------------------
// Sequences longer than 256 rows run on the streaming pair (the block-pair form of round 5 -- one launch of this kernel per
// <= 256 x 256 block pair, accumulating in place -- measured 1.6x slower and was removed in round 6).
// ---------------------------------------------------------------------------------------------------------------------
template <int HD, int SQP, int SKP, bool SPLIT2>
__global__ __launch_bounds__((SQP > SKP ? SQP : SKP) * 2) __attribute__((amdgpu_waves_per_eu(SPLIT2 ? 4 : 3)))
void attn_bwd_fused2_kernel(md_attn_args p) {
    constexpr int PK = (HD + 8) * 2;
    constexpr int RMAX = SQP > SKP ? SQP : SKP;
    constexpr int NT = RMAX * 2;                                       // = blockDim.x: one wave per 32 rows of the taller side
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * RMAX * PK + 2 * SQP * 4];
    unsigned char* sA = smem;                     // Q, later K
    unsigned char* sB = smem + RMAX * PK;         // dO, later V
    float* sLse = reinterpret_cast<float*>(smem + 2 * RMAX * PK);
    float* sDlt = sLse + SQP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5;
    const float c1 = p.scale * LOG2E;        // scores -> log2 domain (ds_cols / p_ds_rows)
    int64_t b, h;
    bh_of(blockIdx.x + (int64_t)gridDim.x * blockIdx.y, gridDim.x, gridDim.y, b, h);
    const bf16* Q = reinterpret_cast<const bf16*>(p.q) + b * p.sq + h * p.hsq;
    const bf16* K = reinterpret_cast<const bf16*>(p.k) + b * p.sk + h * p.hsk;
    const bf16* V = reinterpret_cast<const bf16*>(p.v) + b * p.sv + h * p.hsv;
    const bf16* dO = reinterpret_cast<const bf16*>(p.d_o) + b * p.sdo + h * p.hsdo;
    const bf16* O = reinterpret_cast<const bf16*>(p.o) + b * p.so + h * p.hso;
    const int nq32 = (int)((p.Sq + 31) / 32), nk32 = (int)((p.Skv + 31) / 32);

    constexpr int CPR = HD / 8;
    constexpr int ITQ = (SQP * CPR + NT - 1) / NT, ITK = (SKP * CPR + NT - 1) / NT;
    u32x4 rk[ITK], rv[ITK];
    {
        // every global load of the workgroup in flight before the first LDS write (as in the single-phase kernel)
        const float* LSE = reinterpret_cast<const float*>(p.lse) + (b * p.H + h) * p.Sq;
        u32x4 rq[ITQ], rdo[ITQ], ro[ITQ];
        float rl[ITQ];
        const u32x4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int it = 0; it < ITQ; ++it) {
            const int task = tid + it * NT, r = task / CPR, c = task % CPR;
            const bool ok = task < SQP * CPR && r < p.Sq;
            rq[it] = ok ? *reinterpret_cast<const u32x4*>(Q + (int64_t)r * p.ldq + c * 8) : z4;
            rdo[it] = ok ? *reinterpret_cast<const u32x4*>(dO + (int64_t)r * p.lddo + c * 8) : z4;
            ro[it] = ok ? *reinterpret_cast<const u32x4*>(O + (int64_t)r * p.ldo + c * 8) : z4;
            rl[it] = (ok && c == 0) ? LSE[r] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < ITK; ++it) {
            const int task = tid + it * NT, r = task / CPR, c = task % CPR;
            const bool ok = task < SKP * CPR && r < p.Skv;
            rk[it] = ok ? *reinterpret_cast<const u32x4*>(K + (int64_t)r * p.ldk + c * 8) : z4;
            rv[it] = ok ? *reinterpret_cast<const u32x4*>(V + (int64_t)r * p.ldv + c * 8) : z4;
        }
        // ---- P1: Q, dO -> LDS; delta from the staged registers
#pragma unroll
        for (int it = 0; it < ITQ; ++it) {
            const int task = tid + it * NT, r = task / CPR, c = task % CPR;
            if (task < SQP * CPR) {
                *reinterpret_cast<u32x4*>(sA + r * PK + c * 16) = rq[it];
                *reinterpret_cast<u32x4*>(sB + r * PK + c * 16) = rdo[it];
            }
            const bf16x8 g = __builtin_bit_cast(bf16x8, rdo[it]), o = __builtin_bit_cast(bf16x8, ro[it]);
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) d += bf2f(o[e]) * bf2f(g[e]);
#pragma unroll
            for (int s2 = CPR / 2; s2 > 0; s2 >>= 1) d += __shfl_xor(d, s2, 64);      // the CPR lanes of a row are adjacent (NT % CPR == 0)
            if (c == 0 && task < SQP * CPR) {
                sDlt[r] = d;
                sLse[r] = rl[it] * LOG2E;          // log2 domain (ds_cols / p_ds_rows)
            }
        }
    }
    __syncthreads();
    // ---- P2: this wave's 32 query rows as MFMA operand fragments
    const bool qrole = wave < nq32, krole = wave < nk32;
    bf16x8 qf[HD / 16], dof[HD / 16];
    float lse = 0.f, dlt = 0.f;
    if (qrole) {
#pragma unroll
        for (int s = 0; s < HD / 16; ++s) {
            qf[s] = row_frag(sA + wave * 32 * PK, PK, s * 16, lane);
            dof[s] = row_frag(sB + wave * 32 * PK, PK, s * 16, lane);
        }
        lse = sLse[wave * 32 + (lane & 31)];
        dlt = sDlt[wave * 32 + (lane & 31)];
    }
    __syncthreads();
    // ---- P3: K, V take the place of Q, dO
#pragma unroll
    for (int it = 0; it < ITK; ++it) {
        const int task = tid + it * NT, r = task / CPR, c = task % CPR;
        if (task < SKP * CPR) {
            *reinterpret_cast<u32x4*>(sA + r * PK + c * 16) = rk[it];
            *reinterpret_cast<u32x4*>(sB + r * PK + c * 16) = rv[it];
        }
    }
    __syncthreads();
    // ---- P4 role 1: dQ for query rows wave * 32 ..
    if (qrole) {
        const int64_t q = (int64_t)wave * 32 + (lane & 31);
        f32x16 dqacc[HD / 32];
#pragma unroll
        for (int di = 0; di < HD / 32; ++di)
#pragma unroll
            for (int r = 0; r < 16; ++r) dqacc[di][r] = 0.f;
        for (int j = 0; j < nk32; ++j) {
            const unsigned char* tK = sA + j * 32 * PK;
            const unsigned char* tV = sB + j * 32 * PK;
            f32x16 sacc, dpacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] = 0.f;
                dpacc[r] = 0.f;
            }
#pragma unroll
            for (int s = 0; s < HD / 16; ++s) {
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(tK, PK, s * 16, lane), qf[s], sacc, 0, 0, 0);
                dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(tV, PK, s * 16, lane), dof[s], dpacc, 0, 0, 0);
            }
            ds_cols(sacc, dpacc, c1, lse, dlt, (int)p.Skv - j * 32, hh);   // dS^T / scale
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const bf16x8 dsf = pack8(sacc, 8 * sp);
#pragma unroll
                for (int di = 0; di < HD / 32; ++di)
                    dqacc[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        tr_frag(tK, PK, 16 * sp + 4 * hh, 16 * sp + 8 + 4 * hh, di * 32, lane), dsf, dqacc[di], 0, 0, 0);
            }
        }
        if (q < p.Sq) {
            bf16* dQ = reinterpret_cast<bf16*>(p.dq) + b * p.sdq + h * p.hsdq + q * p.lddq;
            store_rows<HD>(dQ, dqacc, p.scale, lane);
        }
    }
    // this wave's 32 key rows, while K and V are still in LDS
    bf16x8 kf[HD / 16], vf[HD / 16];
    if (krole) {
#pragma unroll
        for (int s = 0; s < HD / 16; ++s) {
            kf[s] = row_frag(sA + wave * 32 * PK, PK, s * 16, lane);
            vf[s] = row_frag(sB + wave * 32 * PK, PK, s * 16, lane);
        }
    }
    __syncthreads();
    // ---- P5: the query rows go back (row_frag's inverse: lane -> row lane & 31, 16-byte column group s * 2 + hh)
    if (qrole) {
#pragma unroll
        for (int s = 0; s < HD / 16; ++s) {
            *reinterpret_cast<u32x4*>(sA + (wave * 32 + (lane & 31)) * PK + (s * 16 + hh * 8) * 2) = __builtin_bit_cast(u32x4, qf[s]);
            *reinterpret_cast<u32x4*>(sB + (wave * 32 + (lane & 31)) * PK + (s * 16 + hh * 8) * 2) = __builtin_bit_cast(u32x4, dof[s]);
        }
    }
    __syncthreads();
    // ---- P6 role 2: dK, dV for key rows wave * 32 ..
    if (krole) {
        const int64_t key = (int64_t)wave * 32 + (lane & 31);
        bf16* dK = reinterpret_cast<bf16*>(p.dk) + b * p.sdk + h * p.hsdk + key * p.lddk;
        bf16* dV = reinterpret_cast<bf16*>(p.dv) + b * p.sdv + h * p.hsdv + key * p.lddv;
        if (SPLIT2) {
            f32x16 acc[HD / 32];
            // pass 1: dV = P^T dO
#pragma unroll
            for (int di = 0; di < HD / 32; ++di)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[di][r] = 0.f;
            for (int i = 0; i < nq32; ++i) {
                const unsigned char* tQ = sA + i * 32 * PK;
                const unsigned char* tdO = sB + i * 32 * PK;
                const float* tLse = sLse + i * 32;
                f32x16 sacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
                for (int s = 0; s < HD / 16; ++s)
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(tQ, PK, s * 16, lane), kf[s], sacc, 0, 0, 0);
                p_ds_rows<true, false>(sacc, sacc, c1, tLse, nullptr, (int)p.Sq - i * 32, hh);   // P
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    const bf16x8 pf = pack8(sacc, 8 * sp);
#pragma unroll
                    for (int di = 0; di < HD / 32; ++di)
                        acc[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            tr_frag(tdO, PK, 16 * sp + 4 * hh, 16 * sp + 8 + 4 * hh, di * 32, lane), pf, acc[di], 0, 0, 0);
                }
            }
            if (key < p.Skv) {
                store_rows<HD>(dV, acc, 1.f, lane);
            }
            // pass 2: dK = dS^T Q
#pragma unroll
            for (int di = 0; di < HD / 32; ++di)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[di][r] = 0.f;
            for (int i = 0; i < nq32; ++i) {
                const unsigned char* tQ = sA + i * 32 * PK;
                const unsigned char* tdO = sB + i * 32 * PK;
                const float* tLse = sLse + i * 32;
                const float* tDlt = sDlt + i * 32;
                f32x16 sacc, dpacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sacc[r] = 0.f;
                    dpacc[r] = 0.f;
                }
#pragma unroll
                for (int s = 0; s < HD / 16; ++s) {
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(tQ, PK, s * 16, lane), kf[s], sacc, 0, 0, 0);
                    dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(tdO, PK, s * 16, lane), vf[s], dpacc, 0, 0, 0);
                }
                p_ds_rows<false, true>(sacc, dpacc, c1, tLse, tDlt, (int)p.Sq - i * 32, hh);   // dS / scale
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    const bf16x8 dsf = pack8(dpacc, 8 * sp);
#pragma unroll
                    for (int di = 0; di < HD / 32; ++di)
                        acc[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            tr_frag(tQ, PK, 16 * sp + 4 * hh, 16 * sp + 8 + 4 * hh, di * 32, lane), dsf, acc[di], 0, 0, 0);
                }
            }
            if (key < p.Skv) {
                store_rows<HD>(dK, acc, p.scale, lane);
            }
        } else {
            f32x16 dkacc[HD / 32], dvacc[HD / 32];
#pragma unroll
            for (int di = 0; di < HD / 32; ++di)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    dkacc[di][r] = 0.f;
                    dvacc[di][r] = 0.f;
                }
            for (int i = 0; i < nq32; ++i) {
                const unsigned char* tQ = sA + i * 32 * PK;
                const unsigned char* tdO = sB + i * 32 * PK;
                const float* tLse = sLse + i * 32;
                const float* tDlt = sDlt + i * 32;
                f32x16 sacc, dpacc;  // S[q][key], dP[q][key]: lane <-> key, regs <-> q
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sacc[r] = 0.f;
                    dpacc[r] = 0.f;
                }
#pragma unroll
                for (int s = 0; s < HD / 16; ++s) {
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(tQ, PK, s * 16, lane), kf[s], sacc, 0, 0, 0);
                    dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(tdO, PK, s * 16, lane), vf[s], dpacc, 0, 0, 0);
                }
                p_ds_rows<true, true>(sacc, dpacc, c1, tLse, tDlt, (int)p.Sq - i * 32, hh);   // P, dS / scale
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    const bf16x8 pf = pack8(sacc, 8 * sp);
                    const bf16x8 dsf = pack8(dpacc, 8 * sp);
#pragma unroll
                    for (int di = 0; di < HD / 32; ++di) {
                        dvacc[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            tr_frag(tdO, PK, 16 * sp + 4 * hh, 16 * sp + 8 + 4 * hh, di * 32, lane), pf, dvacc[di], 0, 0, 0);
                        dkacc[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            tr_frag(tQ, PK, 16 * sp + 4 * hh, 16 * sp + 8 + 4 * hh, di * 32, lane), dsf, dkacc[di], 0, 0, 0);
                    }
                }
            }
            if (key < p.Skv) {
                store_rows<HD>(dK, dkacc, p.scale, lane);
                store_rows<HD>(dV, dvacc, 1.f, lane);
            }
        }
    }
}

// Padded row-count bucket of the fused backward: 64, 96 or 256 (0 = not covered)
inline int fused_bucket(int64_t S) { return S <= 64 ? 64 : S <= 96 ? 96 : S <= 256 ? 256 : 0; }

// variant: 2 = single-phase image (Q, dO, K, V in LDS together), 3 = two-phase image (SPLIT2 for the 256-row buckets only),
// 4 = two-phase with SPLIT2 everywhere (128 VGPRs for every bucket), 0 = the measured rule (MI355X, batch 1024,
// profiles/r2_attention_two_phase_bwd.txt): two-phase everywhere (-9 % with 77 keys, -10 % / -18 % on the 256-row buckets) except
// the 64 x 64 bucket pair, where the single-phase kernel is 10 % ahead (two-wave workgroups: the five extra barriers cost more
// than 6 instead of 4 resident workgroups return).
template <int HD>
void launch_bwd_fused_one(const md_attn_args* a, int variant, int bq, int bk, hipStream_t stream) {
    const dim3 grid((unsigned)a->H, (unsigned)a->B);
#define FUSED(SQP, SKP) hipLaunchKernelGGL((attn_bwd_fused_kernel<HD, SQP, SKP>), grid, dim3((SQP > SKP ? SQP : SKP) * 2), 0, stream, *a)
#define FUSED2(SQP, SKP, SPL) \
    hipLaunchKernelGGL((attn_bwd_fused2_kernel<HD, SQP, SKP, SPL>), grid, dim3((SQP > SKP ? SQP : SKP) * 2), 0, stream, *a)
#define SMALL(SQP, SKP)                          \
    do {                                         \
        if (variant == 2 || (variant == 0 && SQP == 64 && SKP == 64)) FUSED(SQP, SKP); \
        else if (variant == 4) FUSED2(SQP, SKP, true); \
        else FUSED2(SQP, SKP, false);            \
    } while (0)
#define BIG(SQP, SKP)                            \
    do {                                         \
        if (variant == 2) FUSED(SQP, SKP);       \
        else FUSED2(SQP, SKP, true);             \
    } while (0)
    if (bq == 64 && bk == 64) SMALL(64, 64);
    else if (bq == 64 && bk == 96) SMALL(64, 96);
    else if (bq == 96 && bk == 64) SMALL(96, 64);
    else if (bq == 96 && bk == 96) SMALL(96, 96);
    else if (bq == 256 && bk == 96) BIG(256, 96);
    else if (bq == 256 && bk == 256) BIG(256, 256);
    else if (bq == 64 && bk == 256) BIG(64, 256);
    else if (bq == 256 && bk == 64) BIG(256, 64);
    else BIG(96, 256);
#undef SMALL
#undef BIG
#undef FUSED
#undef FUSED2
}

template <int HD>
bool launch_bwd_fused(const md_attn_args* a, int variant, hipStream_t stream) {
    const int bq = fused_bucket(a->Sq), bk = fused_bucket(a->Skv);
    if (!bq || !bk) return false;                                 // longer than one image of the kernel: the streaming pair
    launch_bwd_fused_one<HD>(a, variant, bq, bk, stream);
    return true;
}

// The streaming pair: workgroups of up to 8 waves (MD_ATTN_STREAM_WAVES lowers the cap: A/B runs).
inline int stream_waves(int64_t S) {
    static const int cap = [] {
        const char* e = getenv("MD_ATTN_STREAM_WAVES");
        const int v = e ? atoi(e) : 8;
        return v < 3 ? 3 : (v > 8 ? 8 : v);
    }();
    int w = (int)((S + 31) / 32 > cap ? cap : (S + 31) / 32);
    if (w == 5) w = 6;          // workgroup sizes whose staging loop is 2 or 3 pieces per thread and matrix: 8, 7, 6, 4, 3 waves
    if (w < 3) w = 3;           // (5 or <= 2 waves would need 4: the dK / dV kernel has no registers left for them; the extra waves own
    return w;                   // rows beyond the sequence and multiply zeros)
}
template <int HD>
void launch_bwd_stream(const md_attn_args* a, hipStream_t stream) {
    int nwq = stream_waves(a->Sq), nwk = stream_waves(a->Skv);
    // a walked side of one chunk (the res-512 cross-attention: 77 keys) has nothing to prefetch: smaller workgroups, more of them
    // per CU, hide its single round trip better (1024 x 77, 256 samples x 12 heads: 815 us with 4 waves, 912 with 8)
    if (a->Skv <= 128 && nwq > 4) nwq = 4;
    if (a->Sq <= 128 && nwk > 4) nwk = 4;
    const dim3 gq((unsigned)((a->Sq + 32 * nwq - 1) / (32 * nwq)), (unsigned)a->H, (unsigned)a->B);
    const dim3 gk((unsigned)((a->Skv + 32 * nwk - 1) / (32 * nwk)), (unsigned)a->H, (unsigned)a->B);
    // (chunk rows, 16-byte pieces per thread and matrix) by workgroup size: 8 waves (128, 2); 6-7 (128, 3); 4 (64, 2); 3 (64, 3)
#define STREAM(KERN, GRID, NW)                                                                                       \
    do {                                                                                                             \
        const dim3 blk(64 * NW);                                                                                     \
        if (NW == 8) hipLaunchKernelGGL((KERN<HD, 128, 2>), GRID, blk, 0, stream, *a);                               \
        else if (NW >= 6) hipLaunchKernelGGL((KERN<HD, 128, 3>), GRID, blk, 0, stream, *a);                          \
        else if (NW == 4) hipLaunchKernelGGL((KERN<HD, 64, 2>), GRID, blk, 0, stream, *a);                           \
        else hipLaunchKernelGGL((KERN<HD, 64, 3>), GRID, blk, 0, stream, *a);                                        \
    } while (0)
    STREAM(attn_bwd_dq_stream_kernel, gq, nwq);      // also writes delta
    STREAM(attn_bwd_dkv_stream_kernel, gk, nwk);
#undef STREAM
}

inline int waves_for(int64_t S) {
    int64_t w = (S + 31) / 32;
    return (int)(w > 4 ? 4 : (w < 1 ? 1 : w));
}

inline bool attn_ok(const md_attn_args* a) {
    return a && a->q && a->k && a->v && a->o && a->B > 0 && a->H > 0 && a->Sq > 0 && a->Skv > 0 &&
           (a->hd == 32 || a->hd == 64) && a->ldq % 8 == 0 && a->ldk % 8 == 0 && a->ldv % 8 == 0 && a->ldo % 4 == 0 &&
           a->sq % 8 == 0 && a->sk % 8 == 0 && a->sv % 8 == 0 && a->so % 4 == 0 && a->hsq % 8 == 0 && a->hsk % 8 == 0 &&
           a->hsv % 8 == 0 && a->hso % 4 == 0 && a->hsdo % 8 == 0 && a->hsdq % 4 == 0 && a->hsdk % 4 == 0 && a->hsdv % 4 == 0;
}

// head strides of 0 = the packed default: head h of a row starts h * hd elements into it
inline md_attn_args with_head_strides(const md_attn_args* a) {
    md_attn_args c = *a;
    int64_t* hs[8] = {&c.hsq, &c.hsk, &c.hsv, &c.hso, &c.hsdq, &c.hsdk, &c.hsdv, &c.hsdo};
    for (int64_t* p : hs)
        if (*p == 0) *p = c.hd;
    return c;
}

}  // namespace

extern "C" int md_attn_fwd(const md_attn_args* a_in, hipStream_t stream) {
    if (!attn_ok(a_in)) return MD_BAD_ARG;
    const md_attn_args args = with_head_strides(a_in);
    const md_attn_args* a = &args;
    static const bool fwd_stream = [] { const char* e = getenv("MD_ATTN_FWD_STREAM"); return !e || atoi(e) != 0; }();   // A/B: 0 = the phased kernel everywhere
    static const int fwd_stream_min = [] { const char* e = getenv("MD_ATTN_FWD_STREAM_MIN_SKV"); return e ? atoi(e) : 257; }();   // A/B
    if (fwd_stream && a->Skv >= fwd_stream_min && a->Sq >= 192) {
        // long key sequences (the res-512 mixer: 1024): chunks of 128 keys with the next chunk in flight, workgroups of 6-8 waves
        int nw = stream_waves(a->Sq);
        if (nw < 6) nw = 6;
        const dim3 grid((unsigned)((a->Sq + 32 * nw - 1) / (32 * nw)), (unsigned)a->H, (unsigned)a->B), blk(64 * nw);
        if (a->hd == 64) {
            if (nw == 8) hipLaunchKernelGGL((attn_fwd_stream_kernel<64, 128, 2>), grid, blk, 0, stream, *a);
            else hipLaunchKernelGGL((attn_fwd_stream_kernel<64, 128, 3>), grid, blk, 0, stream, *a);
        } else {
            if (nw == 8) hipLaunchKernelGGL((attn_fwd_stream_kernel<32, 128, 2>), grid, blk, 0, stream, *a);
            else hipLaunchKernelGGL((attn_fwd_stream_kernel<32, 128, 3>), grid, blk, 0, stream, *a);
        }
        MD_LAUNCH_CHECK();
        return 0;
    }
    const int nw = waves_for(a->Sq);
    dim3 grid((unsigned)((a->Sq + 32 * nw - 1) / (32 * nw)), (unsigned)a->H, (unsigned)a->B);
    // chunks of a 32-row phase per thread and matrix: 32 * hd / 8 / (64 * nw)
#define FWD(HD_)                                                                                                  \
    do {                                                                                                           \
        const int chunks = (32 * (HD_ / 8) + 64 * nw - 1) / (64 * nw);                                             \
        if (chunks <= 1) hipLaunchKernelGGL((attn_fwd_kernel<HD_, 1>), grid, dim3(64 * nw), 0, stream, *a);        \
        else if (chunks == 2) hipLaunchKernelGGL((attn_fwd_kernel<HD_, 2>), grid, dim3(64 * nw), 0, stream, *a);   \
        else hipLaunchKernelGGL((attn_fwd_kernel<HD_, 4>), grid, dim3(64 * nw), 0, stream, *a);                    \
    } while (0)
    if (a->hd == 64) FWD(64);
    else FWD(32);
#undef FWD
    MD_LAUNCH_CHECK();
    return 0;
}

// bwd_split: 0 = the library's rule (one fused launch per (batch, head) for Sq, Skv <= 256, the streaming pair beyond);
// 2 / 3 / 4 force the fused kernel's single-phase / two-phase / two-phase SPLIT2 form (Sq, Skv <= 256 only: -1 otherwise, nothing
// launched); 5 forces the streaming pair (any size).  (1 was the round-4 kernel pair, removed in round 6.)
extern "C" int md_attn_bwd(const md_attn_args* a_in, hipStream_t stream) {
    if (!attn_ok(a_in) || !a_in->d_o || !a_in->dq || !a_in->dk || !a_in->dv || !a_in->lse || !a_in->delta) return MD_BAD_ARG;
    const md_attn_args args = with_head_strides(a_in);
    const md_attn_args* a = &args;
    if (a->lddq % 4 || a->lddk % 4 || a->lddv % 4 || a->lddo % 8 || a->sdo % 8) return MD_BAD_ARG;
    if (a->bwd_split < 0 || a->bwd_split > 5 || a->bwd_split == 1) return MD_BAD_ARG;
    const bool long_seq = a->Sq > 256 || a->Skv > 256;
    if (a->bwd_split == 5 || (a->bwd_split == 0 && long_seq)) {
        if (a->hd == 64) launch_bwd_stream<64>(a, stream);
        else launch_bwd_stream<32>(a, stream);
        MD_LAUNCH_CHECK();
        return 0;
    }
    if (a->hd == 64 ? launch_bwd_fused<64>(a, a->bwd_split, stream) : launch_bwd_fused<32>(a, a->bwd_split, stream)) {
        MD_LAUNCH_CHECK();
        return 0;
    }
    return -1;      // a forced fused variant that does not cover this problem: nothing was launched
}

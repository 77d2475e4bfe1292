// bf16 MFMA GEMM family for the MicroDiT linear layers (forward, dgrad, wgrad, grouped MoE experts).
//
//   C[m, n] (+)= alpha * sum_k A(m, k) * B(n, k)          (fp32 accumulate on v_mfma_f32_32x32x16_bf16)
//
// Either operand may be stored "K-contiguous" (element (r, k) at base[r * ld + k]; torch Linear weights
// [out, in] and activations [tokens, channels] feeding a forward GEMM) or "K-strided" (element (r, k) at
// base[k * ld + r]; both operands of a weight-gradient GEMM, the [in, out] expert weights of the MoE, and
// the weight operand of a dgrad).  K-strided tiles are staged in LDS exactly as they lie in HBM (coalesced
// 16-byte rows) and the MFMA fragments are formed with ds_read_b64_tr_b16, CDNA4's transposing LDS read, so
// no operand is ever transposed through HBM.
//
// Tile: 128 x 128 x 64 per 256-thread workgroup (4 waves, 64 x 64 per wave = 2 x 2 MFMA 32x32x16 tiles),
// register-staged global->LDS prefetch (loads of tile t+1 are in flight while tile t is multiplied),
// XCD-aware tile order (neighbouring tiles that share an A row-panel run on the same XCD / L2).
// The epilogue is staged through LDS so that every global access is a 16-byte row-contiguous chunk, and
// fuses: bias, GELU(tanh|erf), gated residual (adaLN-Zero), activation-derivative multiply (dgrad through
// GELU), fp32 store / accumulate / split-K atomic accumulate (wgrad).
//
// Replaces the implicit cuBLAS calls behind every nn.Linear / einsum of the reference
// (micro_diffusion/models/dit.py:84-89,131-142,224; utils.py:58-61,109-111,172-173,225-233).
#include "md_common.h"
#include "gemm_common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128;
constexpr int PITCH_KC = (BKT + 8) * 2;   // bytes per row of a K-contiguous tile  [128][64+8] bf16
constexpr int PITCH_KS = (BM + 32) * 2;   // bytes per row of a K-strided tile     [64][128+32] bf16
constexpr int TILE_BYTES = (BM * PITCH_KC > BKT * PITCH_KS) ? BM * PITCH_KC : BKT * PITCH_KS;  // 20480
constexpr int SLAB_PITCH = 68;            // floats per row of the epilogue slab [32][64+4]
constexpr int SLAB_FLOATS = 32 * SLAB_PITCH;
static_assert(4 * SLAB_FLOATS * 4 <= 2 * TILE_BYTES, "epilogue slab must fit in the staging LDS");

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x4 lds_tr_read(const unsigned char* p) {
    // 16 lanes read one [4 k][16 col] block; lane i of the group receives column i (4 consecutive k).
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    U64 t;
    t.s = v;
    return t.h;
}

// Tile rasterisation.  Workgroup b runs on XCD b % 8 (private 4 MiB L2 each): first give every XCD a contiguous range of
// logical tile ids, then walk the tiles in GROUPS of `group_n` column-tiles x all row-tiles (row-major inside a group), so
// that the B (weight) sub-panel of a group stays L2-resident while the A row-panels stream through once per group.
// (With plain N-fastest order and N = 3072, K = 1024 the 6 MiB weight panel cycled through L2 for every row of tiles.)
__device__ __forceinline__ void tile_coords(int bid, int nwg, int ntn, int group_n, int& tm, int& tn) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    const int ntm = nwg / ntn;
    const int per_group = group_n * ntm;
    const int g = logical / per_group, rem = logical % per_group;
    const int first_n = g * group_n;
    const int gn = (ntn - first_n) < group_n ? (ntn - first_n) : group_n;
    tm = rem / gn;
    tn = first_n + rem % gn;
}

// Optional per-workgroup timeline (md_gemm_args.timeline): shader-clock stamps at entry, after the prologue (first tile
// landed), after the k-loop and after the epilogue's stores have drained, plus HW_ID (XCD / CU / SIMD placement).  Off
// (null pointer) in normal operation; used by scripts/gemm_timeline.py to attribute a launch's time to its phases.
__device__ __forceinline__ void timeline_stamp(long long* tl, int slot) {
    if (tl && threadIdx.x == 0) tl[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + slot] = clock64();
}
__device__ __forceinline__ void timeline_finish(long long* tl) {
    if (!tl) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // include the store drain in the epilogue stamp
    if (threadIdx.x == 0) {
        long long* r = tl + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8;
        r[3] = clock64();
        r[4] = wall_clock64();
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        r[5] = ((long long)xcc << 32) | hw;
    }
}

template <int KC>
__device__ __forceinline__ void load_tile(uint4 (&r)[4], const bf16* __restrict__ base, int64_t ld, int64_t r0,
                                          int64_t rmax, int64_t k0, int64_t kend, int tid) {
    if (KC) {
        const int c = tid & 7;
        const int64_t gk = k0 + c * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t gr = r0 + (tid >> 3) + 32 * i;
            if (gr < rmax && gk < kend)
                r[i] = *reinterpret_cast<const uint4*>(base + gr * ld + gk);
            else
                r[i] = make_uint4(0, 0, 0, 0);
        }
    } else {
        const int c = tid & 15;
        const int64_t gr = r0 + c * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t gk = k0 + (tid >> 4) + 16 * i;
            if (gr < rmax && gk < kend)
                r[i] = *reinterpret_cast<const uint4*>(base + gk * ld + gr);
            else
                r[i] = make_uint4(0, 0, 0, 0);
        }
    }
}

template <int KC>
__device__ __forceinline__ void store_tile(const uint4 (&r)[4], unsigned char* s, int tid) {
    if (KC) {
        const int c = tid & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 3) + 32 * i;
            *reinterpret_cast<uint4*>(s + row * PITCH_KC + c * 16) = r[i];
        }
    } else {
        const int c = tid & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = (tid >> 4) + 16 * i;
            *reinterpret_cast<uint4*>(s + kk * PITCH_KS + c * 16) = r[i];
        }
    }
}

// Fragment of 32 rows x 16 k for v_mfma_f32_32x32x16_bf16: lane l holds row (l & 31), k = (l >> 5) * 8 .. +7.
template <int KC>
__device__ __forceinline__ bf16x8 load_frag(const unsigned char* s, int row0, int ks, int lane) {
    if (KC) {
        const int row = row0 + (lane & 31);
        const int kb = ks * 16 + (lane >> 5) * 8;
        U128 t;
        t.u = *reinterpret_cast<const uint4*>(s + row * PITCH_KC + kb * 2);
        return t.h;
    } else {
        const int li = lane & 15;
        const int col = row0 + ((lane >> 4) & 1) * 16 + (li & 3) * 4;
        const int kk = ks * 16 + (lane >> 5) * 8 + (li >> 2);
        const unsigned char* p = s + kk * PITCH_KS + col * 2;
        bf16x4 lo = lds_tr_read(p);
        bf16x4 hi = lds_tr_read(p + 4 * PITCH_KS);
        bf16x8 f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return f;
    }
}

// Shared epilogue: accumulators -> per-wave fp32 LDS slab -> row-contiguous 16-byte global accesses with the fused ops.
// (mw0, nw0) = global row / column of this wave's (MI * 32) x 64 output block.
template <int MI>
__device__ __forceinline__ void gemm_epilogue(const md_gemm_args& p, f32x16 (&acc)[MI][2], unsigned char* smem, int64_t mw0,
                                              int64_t nw0, int batch, int split, int wave, int lane) {
    float* slab = reinterpret_cast<float*>(smem) + wave * SLAB_FLOATS;
    const int mode = p.mode;
    const float alpha = p.alpha;
    const int erow = lane >> 3;        // 0..7
    const int ecol = (lane & 7) * 8;   // 0..56
    const int64_t gc = nw0 + ecol;
    const bool col_ok = gc < p.N;

    // Every global operand of the fused ops (residual, gate, activation input, bias) is requested up front, before the
    // barrier and the LDS transpose, so their HBM round trips overlap each other and the slab traffic.  Issued one by one
    // at their point of use they serialised 8-16 dependent ~1 us loads per workgroup: the gated-residual epilogue took
    // 18.9 k cycles against 7.5 k for the plain store, 12.8 k now (profiles/r1_gemm_timeline.txt).  Requesting them even
    // earlier, under the last k-tile, cost more in registers than it saved (A/B: -3.5 % on the forward GEMMs).
    bf16x8 pre[2][4], gpre[2][4];      // double-buffered over mi: the set of mi + 1 is requested before mi is processed
    float bv[8];
    const bool has_pre = mode == MD_EPI_RESIDUAL || mode == MD_EPI_DACT;
    const bool has_gate = mode == MD_EPI_RESIDUAL && p.gate != nullptr;
    const bf16* pb = mode == MD_EPI_RESIDUAL ? reinterpret_cast<const bf16*>(p.res)
                                             : reinterpret_cast<const bf16*>(p.aux) + (int64_t)batch * p.sAux;
    const int64_t ldp = mode == MD_EPI_RESIDUAL ? p.ldr : p.ldaux;
    const unsigned rps = has_gate ? (unsigned)p.rows_per_sample : 1u;
    auto prefetch = [&](int mi) {
        if (!has_pre) return;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int64_t gr = mw0 + mi * 32 + it * 8 + erow;
            if (gr < p.M && col_ok) {
                pre[mi & 1][it] = ld_bf16x8(pb + gr * ldp + gc);
                if (has_gate)
                    gpre[mi & 1][it] = ld_bf16x8(reinterpret_cast<const bf16*>(p.gate) + (int64_t)((unsigned)gr / rps) * p.ldg + gc);
            }
        }
    };
    prefetch(0);
    if (p.bias && col_ok) {
        const float* bp = reinterpret_cast<const float*>(p.bias) + (int64_t)batch * p.sBias + gc;
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = bp[e];
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = 0.f;
    }
    __syncthreads();   // every wave is done reading the staging buffers the slabs alias
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        if (mi + 1 < MI) prefetch(mi + 1);
        // C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                slab[row * SLAB_PITCH + ni * 32 + (lane & 31)] = acc[mi][ni][r];
            }
        // The slab is private to this wave: LDS operations of one wave execute in issue order, so no barrier is needed
        // between its writes and its own read-back (a block-wide barrier here cost ~25 % of the K = 1024 kernels).
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int lr = it * 8 + erow;
            const int64_t gr = mw0 + mi * 32 + lr;
            if (gr >= p.M || !col_ok) continue;
            float v[8];
            {
                const float4 v0 = *reinterpret_cast<const float4*>(slab + lr * SLAB_PITCH + ecol);
                const float4 v1 = *reinterpret_cast<const float4*>(slab + lr * SLAB_PITCH + ecol + 4);
                v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w;
                v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * alpha + bv[e];
            if (mode == MD_EPI_STORE_BF16 || mode == MD_EPI_RESIDUAL) {
                if (p.C2) {  // raw (pre-activation / pre-gate) copy for the backward pass -- or, with dact_cached, the activation's derivative
                    bf16x8 o;
                    const bool cache = mode == MD_EPI_STORE_BF16 && p.dact_cached && p.act == MD_ACT_GELU_ERF;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = cache ? f2bf(apply_dact(bf2f(f2bf(v[e])), p.act)) : f2bf(v[e]);
                    st_bf16x8(reinterpret_cast<bf16*>(p.C2) + (int64_t)batch * p.sC2 + gr * p.ldc2 + gc, o);
                }
                if (mode == MD_EPI_STORE_BF16) {
                    if (p.act != MD_ACT_NONE) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], p.act);
                    }
                } else {
                    const bf16x8 rs = pre[mi & 1][it];
                    if (has_gate) {
                        const bf16x8 g = gpre[mi & 1][it];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = bf2f(rs[e]) + bf2f(g[e]) * bf2f(f2bf(v[e]));
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = bf2f(rs[e]) + bf2f(f2bf(v[e]));
                    }
                }
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
                st_bf16x8(reinterpret_cast<bf16*>(p.C) + (int64_t)batch * p.sC + gr * p.ldc + gc, o);
            } else if (mode == MD_EPI_DACT) {
                const bf16x8 ax = pre[mi & 1][it];
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e] * (p.dact_cached ? bf2f(ax[e]) : apply_dact(bf2f(ax[e]), p.act)));
                st_bf16x8(reinterpret_cast<bf16*>(p.C) + (int64_t)batch * p.sC + gr * p.ldc + gc, o);
            } else {
                float* cp = reinterpret_cast<float*>(p.C) + (int64_t)batch * p.sC + (int64_t)split * p.sSplit + gr * p.ldc + gc;
                if (mode == MD_EPI_STORE_F32) {
                    *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else if (mode == MD_EPI_ACCUM_F32) {
                    float4 c0 = *reinterpret_cast<const float4*>(cp);
                    float4 c1 = *reinterpret_cast<const float4*>(cp + 4);
                    c0.x += v[0]; c0.y += v[1]; c0.z += v[2]; c0.w += v[3];
                    c1.x += v[4]; c1.y += v[5]; c1.z += v[6]; c1.w += v[7];
                    *reinterpret_cast<float4*>(cp) = c0;
                    *reinterpret_cast<float4*>(cp + 4) = c1;
                } else {  // MD_EPI_ATOMIC_F32 (split-K weight gradients)
#pragma unroll
                    for (int e = 0; e < 8; ++e) unsafeAtomicAdd(cp + e, v[e]);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
}

template <int AKC, int BKC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void gemm_bf16_kernel(md_gemm_args p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES];
    unsigned char* sA = smem;
    unsigned char* sB = smem + TILE_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- XCD-aware tile order (block b runs on XCD b % 8; give each XCD a contiguous tile range) ----
    const int ntn = (int)((p.N + BN - 1) / BN);
    const int nwg = gridDim.x;
    int tile_m, tile_n;
    tile_coords(blockIdx.x, nwg, ntn, p.raster_group_n, tile_m, tile_n);
    long long* const tl = static_cast<long long*>(p.timeline);
    timeline_stamp(tl, 0);
    const int64_t m0 = (int64_t)tile_m * BM;
    const int64_t n0 = (int64_t)tile_n * BN;

    const int batch = blockIdx.y / p.ksplit;
    const int split = blockIdx.y % p.ksplit;

    const bf16* A = reinterpret_cast<const bf16*>(p.A) + (int64_t)batch * p.sA;
    const bf16* B = reinterpret_cast<const bf16*>(p.B) + (int64_t)batch * p.sB;

    // split-K range, in whole 64-wide k-tiles
    const int64_t ntk = (p.K + BKT - 1) / BKT;
    const int64_t tps = (ntk + p.ksplit - 1) / p.ksplit;
    const int64_t kbeg = (int64_t)split * tps * BKT;
    int64_t kend = kbeg + tps * BKT;
    if (kend > p.K) kend = p.K;
    const int nt = kbeg < kend ? (int)((kend - kbeg + BKT - 1) / BKT) : 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint4 ra[4], rb[4];
    if (nt > 0) {
        load_tile<AKC>(ra, A, p.lda, m0, p.M, kbeg, kend, tid);
        load_tile<BKC>(rb, B, p.ldb, n0, p.N, kbeg, kend, tid);
    }
    for (int t = 0; t < nt; ++t) {
        __syncthreads();  // everyone finished reading the previous tile
        store_tile<AKC>(ra, sA, tid);
        store_tile<BKC>(rb, sB, tid);
        __syncthreads();
        if (t == 0) timeline_stamp(tl, 1);
        if (t + 1 < nt) {
            const int64_t k0 = kbeg + (int64_t)(t + 1) * BKT;
            load_tile<AKC>(ra, A, p.lda, m0, p.M, k0, kend, tid);
            load_tile<BKC>(rb, B, p.ldb, n0, p.N, k0, kend, tid);
        }
#pragma unroll
        for (int ks = 0; ks < BKT / 16; ++ks) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = load_frag<AKC>(sA, wm * 64 + i * 32, ks, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = load_frag<BKC>(sB, wn * 64 + j * 32, ks, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }

    timeline_stamp(tl, 2);
    gemm_epilogue<2>(p, acc, smem, m0 + wm * 64, n0 + wn * 64, batch, split, wave, lane);
    timeline_finish(tl);
}

// =====================================================================================================================
// LDS-DMA variant (default): tiles are streamed HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip, no
// ds_write: the register-staged kernel above is LDS-WRITE bound at 12 waves / CU), two LDS buffers, the loads of
// tile t+1 stay in flight across the barrier while tile t is multiplied (counted s_waitcnt vmcnt).
// The LDS image of one wave-instruction is lane-linear (M0 base + lane * 16 B), so bank-conflict avoidance is an XOR
// swizzle of the 16-byte chunk index applied to the per-lane SOURCE address and again on the fragment reads:
//   K-contiguous tile  [128 rows][8 chunks]   : physical chunk = chunk ^ ((row >> 1) & 7)   (ds_read_b128, 128-B rows)
//   K-strided   tile   [ 64 k   ][16 chunks]  : physical chunk = chunk ^ ((k & 3) << 2)     (ds_read_b64_tr_b16, 256-B rows)
// Out-of-range lanes fetch from a 16-byte zero word instead of being predicated off.
// =====================================================================================================================
__device__ uint4 g_zero16 = {0u, 0u, 0u, 0u};


// Tile geometry is a template parameter: ROWS = rows (or columns) of the operand tile (128 or 256), always 64 deep in K.
// One wave-instruction moves 1 KiB: 8 rows of a K-contiguous tile, or 512/ROWS rows of a K-strided tile; every wave
// issues 4 of them per operand per tile for both supported geometries (128^2 x 4 waves, 256^2 x 8 waves).
template <int KC, int ROWS>
__device__ __forceinline__ void dma_tile(unsigned char* s, const bf16* __restrict__ base, int64_t ld, int64_t r0, int64_t rmax,
                                         int64_t k0, int64_t kend, int wave, int lane, int j0 = 0, int j1 = 4) {
    constexpr int RPC = 512 / ROWS;       // K-strided: k-rows per 1 KiB chunk (4 or 2)
    constexpr int LPR = 64 / RPC;         // lanes (16-byte chunks) per k-row (16 or 32)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < j0 || j >= j1) continue;
        const bf16* src;
        if (KC) {
            const int row = wave * 32 + j * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            const int64_t gr = r0 + row, gk = k0 + c * 8;
            src = (gr < rmax && gk < kend) ? base + gr * ld + gk : reinterpret_cast<const bf16*>(&g_zero16);
        } else {
            const int kk = (wave * 4 + j) * RPC + lane / LPR;
            const int c = (lane % LPR) ^ ((kk & 3) << 2);
            const int64_t gk = k0 + kk, gr = r0 + c * 8;
            src = (gr < rmax && gk < kend) ? base + gk * ld + gr : reinterpret_cast<const bf16*>(&g_zero16);
        }
        __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(s + (wave * 4 + j) * 1024), 16, 0, 0);
    }
}

// Fragment reads as inline asm: hipcc would otherwise put a full s_waitcnt vmcnt(0) in front of every LDS read
// while an LDS-DMA is pending and so drain the prefetch of the next tile.  The caller waits lgkmcnt(0) itself.
__device__ __forceinline__ bf16x8 asm_read_b128(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ void asm_read_tr2(unsigned addr, bf16x4& lo, bf16x4& hi) {
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3" : "=&v"(lo), "=&v"(hi) : "v"(addr), "i"(OFF) : "memory");
}

template <int KC, int ROWS>
__device__ __forceinline__ bf16x8 dma_frag(unsigned sbase, int row0, int ks, int lane) {
    if (KC) {
        const int row = row0 + (lane & 31);
        const int c = (ks * 2 + (lane >> 5)) ^ ((row >> 1) & 7);
        return asm_read_b128(sbase + row * 128 + c * 16);
    } else {
        const int li = lane & 15;
        const int col = row0 + ((lane >> 4) & 1) * 16 + (li & 3) * 4;
        const int kk = ks * 16 + (lane >> 5) * 8 + (li >> 2);
        const int pc = (col >> 3) ^ ((kk & 3) << 2);
        bf16x4 lo, hi;
        asm_read_tr2<4 * ROWS * 2>(sbase + kk * (ROWS * 2) + pc * 16 + ((col >> 2) & 1) * 8, lo, hi);   // hi: k-rows + 4
        bf16x8 f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return f;
    }
}

// WM x WN waves; every wave owns (MI * 32) x 64 outputs.  TM = WM * MI * 32, TN = WN * 64.
template <int AKC, int BKC, int WM, int WN, int MI, bool PACED = false>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_dma_kernel(md_gemm_args p) {
    constexpr int TM = WM * MI * 32, TN = WN * 64;
    constexpr int ATILE = TM * 128, BTILE = TN * 128;          // bytes (64 k x 2 B per row / column)
    constexpr int BUF = ATILE + BTILE;
    constexpr int SLAB = WM * WN * SLAB_FLOATS * 4;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[(2 * BUF > SLAB) ? 2 * BUF : SLAB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int ntn = (int)((p.N + TN - 1) / TN);
    const int nwg = gridDim.x;
    int tile_m, tile_n;
    tile_coords(blockIdx.x, nwg, ntn, p.raster_group_n, tile_m, tile_n);
    long long* const tl = static_cast<long long*>(p.timeline);
    timeline_stamp(tl, 0);
    const int64_t m0 = (int64_t)tile_m * TM;
    const int64_t n0 = (int64_t)tile_n * TN;
    const int batch = blockIdx.y / p.ksplit;
    const int split = blockIdx.y % p.ksplit;
    const bf16* A = reinterpret_cast<const bf16*>(p.A) + (int64_t)batch * p.sA;
    const bf16* B = reinterpret_cast<const bf16*>(p.B) + (int64_t)batch * p.sB;
    const int64_t ntk = (p.K + BKT - 1) / BKT;
    const int64_t tps = (ntk + p.ksplit - 1) / p.ksplit;
    const int64_t kbeg = (int64_t)split * tps * BKT;
    int64_t kend = kbeg + tps * BKT;
    if (kend > p.K) kend = p.K;
    const int nt = kbeg < kend ? (int)((kend - kbeg + BKT - 1) / BKT) : 0;

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;   // LDS byte address of the staging area
    if (nt > 0) {
        dma_tile<AKC, TM>(smem, A, p.lda, m0, p.M, kbeg, kend, wave, lane);
        dma_tile<BKC, TN>(smem + ATILE, B, p.ldb, n0, p.N, kbeg, kend, wave, lane);
    }
    // Fragments are double-buffered in registers: the LDS reads of k-step ks+1 are issued in front of the MFMAs of k-step
    // ks and waited for behind them, so a wave's own MFMAs cover its LDS latency (with one or two waves per SIMD nothing
    // else does).  The reads are inline asm (see asm_read_b128); sched_barrier keeps hipcc from moving the MFMAs across
    // the waits that guard their operands.
    bf16x8 fa[2][MI], fb[2][2];
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        const bool more = t + 1 < nt;
        const int64_t k0 = kbeg + (int64_t)(t + 1) * BKT;
        unsigned char* nb = smem + (cur ^ 1) * BUF;
        if (PACED) {
            // One barrier per tile: the DMAs of tile t+1 are issued AFTER the barrier that opens tile t (all waves are done
            // with tile t-1, whose buffer they overwrite) and paced one chunk of each operand per k-step, instead of an
            // 8-deep burst at the top of the loop that backs up the vector-memory queue in front of the wave's MFMAs.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of tile t has landed
            __builtin_amdgcn_s_barrier();
        } else {
            if (more) {
                dma_tile<AKC, TM>(nb, A, p.lda, m0, p.M, k0, kend, wave, lane);
                dma_tile<BKC, TN>(nb + ATILE, B, p.ldb, n0, p.N, k0, kend, wave, lane);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tile t landed (this wave's 8 DMAs); tile t+1 in flight
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();                          // ... and every other wave's part of tile t
        }
        if (t == 0) timeline_stamp(tl, 1);
        const unsigned sA = lds0 + cur * BUF, sB = sA + ATILE;
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[0][i] = dma_frag<AKC, TM>(sA, wm * (MI * 32) + i * 32, 0, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[0][j] = dma_frag<BKC, TN>(sB, wn * 64 + j * 32, 0, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < BKT / 16; ++ks) {
            const int cb = ks & 1, nx = cb ^ 1;
            if (PACED && more) {
                dma_tile<AKC, TM>(nb, A, p.lda, m0, p.M, k0, kend, wave, lane, ks, ks + 1);
                dma_tile<BKC, TN>(nb + ATILE, B, p.ldb, n0, p.N, k0, kend, wave, lane, ks, ks + 1);
            }
            if (ks + 1 < BKT / 16) {
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[nx][i] = dma_frag<AKC, TM>(sA, wm * (MI * 32) + i * 32, ks + 1, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[nx][j] = dma_frag<BKC, TN>(sB, wn * 64 + j * 32, ks + 1, lane);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cb][i], fb[cb][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < BKT / 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!PACED) __builtin_amdgcn_s_barrier();                  // buffer `cur` is free for the DMA of tile t+2
    }
    if (PACED) __builtin_amdgcn_s_barrier();
    timeline_stamp(tl, 2);
    gemm_epilogue<MI>(p, acc, smem, m0 + wm * (MI * 32), n0 + wn * 64, batch, split, wave, lane);
    timeline_finish(tl);
}

// out[b][m][n] (+)= sum_s ws[b][s][m][n]   (ws slices are dense [M, N]; 16-byte accesses).  The slices are read once:
// non-temporal loads, four independent slice loads in flight per thread (the one-load-per-iteration loop ran at 2.8 TB/s);
// the summation order is fixed (deterministic weight gradients).
__device__ __forceinline__ float4 nt_ld4(const float* p) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* ws, float* out, int64_t M, int64_t N, int64_t ldo,
                                                            int64_t sOut, int ksplit, int batch, int accumulate, int64_t slice) {
    const int64_t n4 = N / 4, per = M * n4, total = per * batch;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / per, r = i % per, m = r / n4, c = (r % n4) * 4;
        const float* src = ws + (b * ksplit) * slice + m * N + c;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int s = 0;
        for (; s + 4 <= ksplit; s += 4) {
            const float4 v0 = nt_ld4(src + (int64_t)s * slice), v1 = nt_ld4(src + (int64_t)(s + 1) * slice);
            const float4 v2 = nt_ld4(src + (int64_t)(s + 2) * slice), v3 = nt_ld4(src + (int64_t)(s + 3) * slice);
            acc.x += (v0.x + v1.x) + (v2.x + v3.x); acc.y += (v0.y + v1.y) + (v2.y + v3.y);
            acc.z += (v0.z + v1.z) + (v2.z + v3.z); acc.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; s < ksplit; ++s) {
            const float4 v = nt_ld4(src + (int64_t)s * slice);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        if (accumulate == 2) {           // bf16 store (out is a bf16 matrix: the gradient-exchange buffer of a one-microbatch step)
            bf16x4 o;
            o[0] = f2bf(acc.x); o[1] = f2bf(acc.y); o[2] = f2bf(acc.z); o[3] = f2bf(acc.w);
            st_bf16x4(reinterpret_cast<bf16*>(out) + b * sOut + m * ldo + c, o);
            continue;
        }
        float* dst = out + b * sOut + m * ldo + c;
        if (accumulate) {
            const float4 o = *reinterpret_cast<const float4*>(dst);
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        *reinterpret_cast<float4*>(dst) = acc;
    }
}

}  // namespace

extern "C" int md_splitk_reduce(const float* ws, float* out, int64_t M, int64_t N, int64_t ldo, int64_t sOut, int32_t ksplit,
                                int32_t batch, int32_t accumulate, hipStream_t stream) {
    if (!ws || !out || M <= 0 || N <= 0 || N % 4 || ldo % 4 || ksplit <= 0 || batch <= 0 || accumulate < 0 || accumulate > 2) return MD_BAD_ARG;
    int64_t grid = (M * (N / 4) * batch + 255) / 256;
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)grid), dim3(256), 0, stream, ws, out, M, N, ldo, sOut, ksplit, batch,
                       accumulate, M * N);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_splitk_reduce_flat(const float* ws, float* out, int64_t n, int64_t slice_stride, int32_t ksplit, int32_t accumulate,
                                     hipStream_t stream) {
    if (!ws || !out || n <= 0 || n % 4 || slice_stride < n || slice_stride % 4 || ksplit <= 0 || accumulate < 0 || accumulate > 2) return MD_BAD_ARG;
    int64_t grid = (n / 4 + 255) / 256;
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)grid), dim3(256), 0, stream, ws, out, (int64_t)1, n, n, (int64_t)0, ksplit, 1,
                       accumulate, slice_stride);
    MD_LAUNCH_CHECK();
    return 0;
}

// Under a CU hold (md_gemm_args.cu_limit: the data-parallel step while a collective is on the wire) w4 -- which has no split-K tail -- is
// taken only when its whole tiles make nearly whole rounds of the cu_limit workgroups: filled share of all rounds >= 92 % (256 tiles on 248
// workgroups would be two rounds, 52 %: those launches keep PP256 and its tail form).  On a free chip the >= 192-tile rule stands alone.
static bool md_gemm_w4_fills(int64_t tiles, int cu_limit) {
    if (!(cu_limit > 0 && cu_limit < 256)) return true;
    if (tiles <= cu_limit) return false;
    const int64_t rounds = (tiles + cu_limit - 1) / cu_limit;
    return tiles * 100 >= rounds * cu_limit * 92;
}

extern "C" int md_gemm_bf16(const md_gemm_args* a_in, hipStream_t stream) {
    if (!a_in) return MD_BAD_ARG;
    md_gemm_args a_copy = *a_in;                     // raster_group_n is filled in below
    const md_gemm_args* a = &a_copy;
    if (a->problems) {            // grouped launch: pp256 or nothing; A / B / M / N / leading dimensions come from the problem table
        if (!a->C || a->K <= 0 || a->ksplit <= 0 || a->mode != MD_EPI_STORE_F32 || a->sSplit <= 0 || !md_gemm_pp_eligible(a)) return MD_BAD_ARG;
        if (a->variant != MD_GEMM_AUTO && a->variant != MD_GEMM_PP256 && a->variant != MD_GEMM_W4) return MD_NOT_ELIGIBLE;
        if (a->variant == MD_GEMM_W4 && !md_gemm_w4_eligible(a)) return MD_NOT_ELIGIBLE;
        // the 4-wave kernel takes the grouped weight gradients it covers (whole interior tiles) on a free chip (profiles/r6_w4_wgrad.txt)
        bool w4 = a->variant == MD_GEMM_W4;
        if (a->variant == MD_GEMM_AUTO && md_gemm_w4_eligible(a) && !getenv("MD_GEMM_NO_W4") && !getenv("MD_GEMM_NO_W4_TN")) {
            int64_t t = 0;
            for (int i = 0; i < a->n_problems; ++i) t += ((a->problems[i].M + 255) / 256) * ((a->problems[i].N + 255) / 256);
            w4 = md_gemm_w4_fills(t * a->ksplit, a->cu_limit);
        }
        if (a->chosen_variant) *a->chosen_variant = w4 ? MD_GEMM_W4 : MD_GEMM_PP256;
        return w4 ? md_gemm_w4_launch(a, stream) : md_gemm_pp_launch(a, stream);
    }
    if (!a->A || !a->B || !a->C) return MD_BAD_ARG;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->batch <= 0 || a->ksplit <= 0) return MD_BAD_ARG;
    // All global accesses are 16-byte chunks of 8 bf16 along the contiguous dimension: leading dimensions must be
    // multiples of 8 (4 for fp32 outputs).  A contiguous extent that is not a multiple of 8 (the MoE gate: N = K =
    // num_experts) is allowed as long as the caller pads the rows to the next multiple of 8 with finite values
    // (zeros): whole chunks are read / written whenever their first element is in range.
    if (a->lda % 8 || a->ldb % 8) return MD_BAD_ARG;
    if ((a->mode == MD_EPI_STORE_BF16 || a->mode == MD_EPI_RESIDUAL || a->mode == MD_EPI_DACT || a->mode == MD_EPI_SWIGLU_BWD) ? (a->ldc % 8) : (a->ldc % 4))
        return MD_BAD_ARG;
    if (a->C2 && a->ldc2 % 8) return MD_BAD_ARG;
    // split-K: either atomics into C, or every split stores its own fp32 slice (C + split * sSplit) for md_splitk_reduce
    if (a->ksplit > 1 && !(a->mode == MD_EPI_ATOMIC_F32 || (a->mode == MD_EPI_STORE_F32 && a->sSplit > 0))) return MD_BAD_ARG;
    if (a->mode == MD_EPI_RESIDUAL && (!a->res || (a->gate && a->rows_per_sample <= 0))) return MD_BAD_ARG;
    if (a->mode == MD_EPI_DACT && !a->aux) return MD_BAD_ARG;
    if (a->mode < MD_EPI_STORE_BF16 || a->mode > MD_EPI_SWIGLU_BWD) return MD_BAD_ARG;
    if (a->mode == MD_EPI_SWIGLU_BWD) {
        // the SwiGLU backward fused into the w3 data gradient: built into the 4-wave kernel only (interior tiles, free chip); any
        // other problem is NOT_ELIGIBLE -- nothing is launched and the caller runs the plain data gradient + md_swiglu_bwd
        if (!a->aux || a->ldaux % 8) return MD_BAD_ARG;
        if (a->variant != MD_GEMM_AUTO && a->variant != MD_GEMM_W4) return MD_NOT_ELIGIBLE;
        if (!md_gemm_w4_eligible(a) || (a->cu_limit > 0 && a->cu_limit < 256) || getenv("MD_GEMM_NO_W4")) return MD_NOT_ELIGIBLE;
        a_copy.variant = MD_GEMM_W4;
    }
    // dact_cached: C2 of the forward (STORE_BF16) holds gelu'(h) in place of h and the DACT launch multiplies by aux as it is -- both
    // sides only with the erf-GELU, the forward only with a C2 to write (every kernel family checks the same thing here, once)
    if (a->dact_cached && (a->act != MD_ACT_GELU_ERF || !(a->mode == MD_EPI_DACT || (a->mode == MD_EPI_STORE_BF16 && a->C2)))) return MD_BAD_ARG;
    // Kernel choice.  a->variant forces one (parity tests drive every kernel on the real shapes; A/B runs); AUTO applies the
    // rules measured on MI355X in the XL/2 shape mix (profiles/r2_gemm_variants.txt):
    //  * PP256 (persistent ping-pong) whenever it is eligible and the launch has enough tiles to fill the chip;
    //  * else PACED256 for weight gradients / long K on grids that make whole rounds of 256 workgroups;
    //  * else 128 x 128: register-staged (3 workgroups / CU) for short K and TN shapes, LDS-DMA otherwise.
    const int64_t tiles256 = ((a->M + 255) / 256) * ((a->N + 255) / 256) * (int64_t)a->batch * a->ksplit;
    const int64_t kspan = (a->K + a->ksplit - 1) / a->ksplit;   // contraction length one workgroup walks
    int variant = a->variant;
    if (variant < MD_GEMM_AUTO || variant > MD_GEMM_W4) return MD_BAD_ARG;
    if (variant == MD_GEMM_PP256 && !md_gemm_pp_eligible(a)) return MD_NOT_ELIGIBLE;
    if (variant == MD_GEMM_W4 && !md_gemm_w4_eligible(a)) return MD_NOT_ELIGIBLE;
    if ((a->A_list || a->B_list) && !(a->A_list && a->B_list && md_gemm_pp_eligible(a))) return MD_BAD_ARG;   // operand lists: PP256 only
    if (a->A_list) variant = MD_GEMM_PP256;
    if (variant == MD_GEMM_AUTO) {
        // W4 (round 6): the 4-wave 16x16x32 kernel wherever it is built for the problem and the launch has at least 192 tiles that make
        // (nearly) whole rounds of the workgroups it may use (md_gemm_w4_fills: w4 has no split-K tail -- under a CU hold, 256 tiles on 248
        // workgroups would be two rounds; those launches keep PP256 and its tail)
        const char* w4min = getenv("MD_GEMM_W4_MIN_TILES");            // (A/B runs; default: the launches PP256 used to take)
        if (md_gemm_w4_eligible(a) && tiles256 >= (w4min ? atoi(w4min) : 192) && md_gemm_w4_fills(tiles256, a->cu_limit) && !getenv("MD_GEMM_NO_W4") &&
            (a->a_kcontig || !getenv("MD_GEMM_NO_W4_TN")))
            variant = MD_GEMM_W4;
        else if (md_gemm_pp_eligible(a) && tiles256 >= 192)
            variant = MD_GEMM_PP256;
        else if (!a->a_kcontig && !a->b_kcontig && kspan >= 2048 && tiles256 >= 128)
            variant = MD_GEMM_PACED256;   // weight gradients (TN) when the caller's split-K makes ~one full round of 256 workgroups
        else if (a->K >= 1024 && tiles256 >= 224 && tiles256 <= 256)
            variant = MD_GEMM_PACED256;   // exactly one round of 256^2 workgroups
        else if (kspan > 2048 && tiles256 >= 240 && (tiles256 % 256 == 0 || tiles256 % 256 >= 160 || tiles256 >= 2048))
            variant = MD_GEMM_PACED256;   // long K amortises the un-overlapped prologue/epilogue; avoid ragged rounds
        else if (a->a_kcontig && a->K <= 2048)
            variant = MD_GEMM_REG128;     // short contraction: 3 workgroups / CU hide each other's prologue / epilogue
        else
            variant = ((!a->a_kcontig && !a->b_kcontig) || a->K < 1024) ? MD_GEMM_REG128 : MD_GEMM_DMA128;
    }
    const int TMv = variant >= MD_GEMM_PACED256 ? 256 : 128;
    if (a->raster_group_n <= 0) {   // column-tiles per raster group: keep the group's B sub-panel (TN x K bf16) within ~2 MiB of the 4 MiB L2
        const int64_t ntn_ = (a->N + TMv - 1) / TMv;
        int64_t g = (2 << 20) / (TMv * kspan * 2);
        if (variant >= MD_GEMM_PP256 && g < 4) g = 4;   // an XCD's 32 concurrent tiles form an (32 / g) x g block
        if (g < 1) g = 1;
        if (g > ntn_) g = ntn_;
        a_copy.raster_group_n = (int)g;
    }
    if (a->chosen_variant) *a->chosen_variant = variant;
    if (variant == MD_GEMM_W4) return md_gemm_w4_launch(a, stream);
    if (variant >= MD_GEMM_PP256) return md_gemm_pp_launch(a, stream);
    const int64_t tiles = ((a->M + TMv - 1) / TMv) * ((a->N + TMv - 1) / TMv);
    dim3 grid((unsigned)tiles, (unsigned)(a->batch * a->ksplit), 1);
#define LAUNCH(KERN, THREADS, ...)                                                                                          \
    do {                                                                                                                   \
        if (a->a_kcontig && a->b_kcontig) hipLaunchKernelGGL((KERN<1, 1 __VA_ARGS__>), grid, dim3(THREADS), 0, stream, *a);       \
        else if (a->a_kcontig && !a->b_kcontig) hipLaunchKernelGGL((KERN<1, 0 __VA_ARGS__>), grid, dim3(THREADS), 0, stream, *a); \
        else if (!a->a_kcontig && a->b_kcontig) hipLaunchKernelGGL((KERN<0, 1 __VA_ARGS__>), grid, dim3(THREADS), 0, stream, *a); \
        else hipLaunchKernelGGL((KERN<0, 0 __VA_ARGS__>), grid, dim3(THREADS), 0, stream, *a);                                     \
    } while (0)
#define COMMA ,
    if (variant == MD_GEMM_REG128) LAUNCH(gemm_bf16_kernel, 256, );
    else if (variant == MD_GEMM_DMA128) LAUNCH(gemm_bf16_dma_kernel, 256, COMMA 2 COMMA 2 COMMA 2);
    else LAUNCH(gemm_bf16_dma_kernel, 512, COMMA 2 COMMA 4 COMMA 4 COMMA true);
#undef COMMA
#undef LAUNCH
    MD_LAUNCH_CHECK();
    return 0;
}

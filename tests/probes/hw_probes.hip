// Hardware-semantics probes: TEST INFRASTRUCTURE, built into tests/probes/libmd_probes.so (tests/probes/__init__.py), never
// into the product library and not part of its C ABI (include/microdit_hip.h).  They pin down the three hardware behaviours
// the hand-scheduled GEMM relies on: the ds_read_b64_tr_b16 lane map, the v_mfma_f32_32x32x16_bf16 operand / accumulator
// layout, and in-order retirement of a wave's loads and stores in vmcnt.
#include "../../micro_diffusion_amd/csrc/md_common.h"

namespace {
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// LDS holds sm[i] = i (as int16).  Lane l passes the address of element addr_elems[l]; the 4 int16 each lane
// receives from ds_read_b64_tr_b16 are written to out[l*4 .. l*4+3].
__global__ void tr_probe_kernel(const int* addr_elems, short* out) {
    __shared__ __attribute__((aligned(16))) short sm[8192];
    const int l = threadIdx.x;
    for (int i = l; i < 8192; i += 64) sm[i] = (short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sm + addr_elems[l]));
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = v[i];
}

// One v_mfma_f32_32x32x16_bf16 with A(i,k), B(k,j) supplied per the layout the GEMM assumes; dumps D.
__global__ void mfma_probe_kernel(const bf16* A /*[32][16]*/, const bf16* B /*[16][32]*/, float* D /*[32][32]*/) {
    const int l = threadIdx.x;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = A[(l & 31) * 16 + (l >> 5) * 8 + e];
        b[e] = B[((l >> 5) * 8 + e) * 32 + (l & 31)];
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        D[row * 32 + (l & 31)] = acc[r];
    }
}
// Does vmcnt retire LOADS and STORES of one wave in issue order?  Every lane issues a cold 16-byte load (one distinct
// cache line per lane of a large buffer: an HBM miss), then NST hot 16-byte stores (the same few L2-resident lines), then
// waits s_waitcnt vmcnt(4): with in-order retirement the load, the oldest operation, must have landed.  The destination
// register is pre-set to a sentinel inside the same asm; a lane that still reads the sentinel proves that younger stores
// retired (dropped the counter) ahead of the older load.  out[0] += number of such lanes.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__global__ __launch_bounds__(256) void vmcnt_order_probe_kernel(const u32x4* cold, int64_t cold_stride16, u32x4* hot, unsigned* out) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const u32x4* src = cold + gid * cold_stride16;
    u32x4* dst = hot + (threadIdx.x & 63);
    u32x4 v = {0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu};
    const u32x4 st = {1u, 2u, 3u, 4u};
    asm volatile("global_load_dwordx4 %0, %1, off\n\t"
                 "global_store_dwordx4 %2, %3, off\n\tglobal_store_dwordx4 %2, %3, off offset:1024\n\t"
                 "global_store_dwordx4 %2, %3, off offset:2048\n\tglobal_store_dwordx4 %2, %3, off offset:3072\n\t"
                 "s_waitcnt vmcnt(4)\n\ts_nop 1"
                 : "+v"(v) : "v"(src), "v"(dst), "v"(st) : "memory");
    const bool stale = (v[0] == 0xdeadbeefu);          // sampled right after the counted wait ...
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ... before everything is drained
    const unsigned long long m = __ballot(stale);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, (unsigned)__popcll(m));
}
// A stand-in for an RCCL kernel on a single GPU: `blocks` workgroups that each hold a CU's LDS (96 KiB: a workgroup of the
// persistent GEMM, 128 KiB, cannot share the CU) and spin for `ticks` of the 100 MHz wall clock.  tests/test_gemm_variants_gpu.py
// uses it to show what md_gemm_args.cu_limit is for.
__global__ __launch_bounds__(256) void cu_hog_kernel(long long ticks, unsigned* out) {
    __shared__ unsigned char big[96 * 1024];
    big[threadIdx.x] = (unsigned char)threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0 && big[blockIdx.x & 255] == 77 && ticks < 0) atomicAdd(out, 1u);   // keeps `big` alive
}
}  // namespace

extern "C" int mdp_cu_hog(int32_t blocks, int64_t microseconds, uint32_t* scratch, hipStream_t stream) {
    if (blocks <= 0 || blocks > 256 || microseconds < 0 || !scratch) return MD_BAD_ARG;
    hipLaunchKernelGGL(cu_hog_kernel, dim3(blocks), dim3(256), 0, stream, (long long)microseconds * 100, scratch);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int mdp_tr_probe(const int32_t* addr_elems, int16_t* out, hipStream_t stream) {
    hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, stream, addr_elems, out);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int mdp_mfma_probe(const void* A, const void* B, float* D, hipStream_t stream) {
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, stream, (const bf16*)A, (const bf16*)B, D);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int mdp_vmcnt_order_probe(const void* cold, int64_t cold_bytes, void* hot, uint32_t* stale_lanes, int32_t blocks,
                                          hipStream_t stream) {
    if (!cold || !hot || !stale_lanes || blocks <= 0) return MD_BAD_ARG;
    const int64_t lanes = (int64_t)blocks * 256;
    const int64_t stride16 = cold_bytes / 16 / lanes;                 // one private, far-apart 16-byte word per lane
    if (stride16 < 8) return MD_BAD_ARG;
    hipLaunchKernelGGL(vmcnt_order_probe_kernel, dim3(blocks), dim3(256), 0, stream, (const u32x4*)cold, stride16, (u32x4*)hot,
                       stale_lanes);
    MD_LAUNCH_CHECK();
    return 0;
}

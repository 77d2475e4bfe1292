// Hardware-semantics probes (test infrastructure only; not on the product path).
#include "md_common.h"
#include "../../include/microdit_hip.h"

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
}  // namespace

extern "C" int md_abi_version(void) { return MD_ABI_VERSION; }

extern "C" int md_debug_tr_probe(const int32_t* addr_elems, int16_t* out, hipStream_t stream) {
    hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, stream, addr_elems, out);
    MD_LAUNCH_CHECK();
    return 0;
}

extern "C" int md_debug_mfma_probe(const void* A, const void* B, float* D, hipStream_t stream) {
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, stream, (const bf16*)A, (const bf16*)B, D);
    MD_LAUNCH_CHECK();
    return 0;
}

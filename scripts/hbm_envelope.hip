// What HBM3E delivers to the ACCESS PATTERNS of the MicroDiT bandwidth-bound kernels (standalone: hipcc scripts/hbm_envelope.hip).
// The step's LayerNorm / SwiGLU / gate kernels sit at 3.3-4.0 TB/s of algorithmic traffic while the in-place QK-LayerNorm
// reaches 5.5; the guide's "~6.3 TB/s achievable" is a single number for an unspecified pattern.  This measures, on buffers far
// larger than the 256 MiB Infinity Cache, with 16-byte accesses and several loads in flight per lane:
//   read-only, write-only, copy A -> B, in-place A -> A, 2 reads + 1 write (SwiGLU forward), 3 reads + 2 writes out of place and
//   with the outputs overwriting two of the inputs (SwiGLU backward), the same with non-temporal accesses, and a wave-per-row
//   walk (the LayerNorm kernels' shape).  Output: GB/s of algorithmic bytes (reads + writes) per pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// grid-stride over 16-byte words, U words per lane in flight
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const u32x4* a, size_t n, unsigned* sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NT>(a + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_write(u32x4* a, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    const u32x4 v = {1u, 2u, 3u, (unsigned)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride)
#pragma unroll
        for (int u = 0; u < U; ++u) st<NT>(a + i + u * stride, v);
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const u32x4* a, u32x4* b, size_t n) {   // b may alias a (in place)
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NT>(a + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) st<NT>(b + i + u * stride, v[u] + 1u);
    }
}
// SwiGLU forward shape: rows of [2f] in, [f] out; lane keeps its column chunk, 4 waves take interleaved rows (as md_swiglu_fwd)
template <int RPT, bool NT>
__global__ __launch_bounds__(256) void k_2in1out(const u32x4* h12, u32x4* a, long M, long f16 /* 16-byte words per half row */) {
    const long c = (long)blockIdx.x * 64 + (threadIdx.x & 63);
    if (c >= f16) return;
    const long rpb = (M + gridDim.y - 1) / gridDim.y;
    long r = (long)blockIdx.y * rpb + (threadIdx.x >> 6), r1 = (long)(blockIdx.y + 1) * rpb;
    if (r1 > M) r1 = M;
    for (; r + 4 * (RPT - 1) < r1; r += 4 * RPT) {
        u32x4 x[RPT], y[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) { x[i] = ld<NT>(h12 + (r + 4 * i) * 2 * f16 + c); y[i] = ld<NT>(h12 + (r + 4 * i) * 2 * f16 + f16 + c); }
#pragma unroll
        for (int i = 0; i < RPT; ++i) st<NT>(a + (r + 4 * i) * f16 + c, x[i] ^ y[i]);
    }
}
// SwiGLU backward shape: da [f], h12 [2f] in; dh12 [2f] out (out == h12: the outputs overwrite the dead inputs)
template <int RPT, bool NT>
__global__ __launch_bounds__(256) void k_3in2out(const u32x4* da, const u32x4* h12, u32x4* dh12, long M, long f16) {
    const long c = (long)blockIdx.x * 64 + (threadIdx.x & 63);
    if (c >= f16) return;
    const long rpb = (M + gridDim.y - 1) / gridDim.y;
    long r = (long)blockIdx.y * rpb + (threadIdx.x >> 6), r1 = (long)(blockIdx.y + 1) * rpb;
    if (r1 > M) r1 = M;
    for (; r + 4 * (RPT - 1) < r1; r += 4 * RPT) {
        u32x4 x[RPT], y[RPT], d[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            x[i] = ld<NT>(h12 + (r + 4 * i) * 2 * f16 + c); y[i] = ld<NT>(h12 + (r + 4 * i) * 2 * f16 + f16 + c); d[i] = ld<NT>(da + (r + 4 * i) * f16 + c);
        }
#pragma unroll
        for (int i = 0; i < RPT; ++i) { st<NT>(dh12 + (r + 4 * i) * 2 * f16 + c, x[i] ^ d[i]); st<NT>(dh12 + (r + 4 * i) * 2 * f16 + f16 + c, y[i] + d[i]); }
    }
}
// LayerNorm shape: one wave per 2 KiB row (64 lanes x 2 x 16 bytes), RW consecutive rows per wave, next row's loads issued before this row's store
template <int RW>
__global__ __launch_bounds__(256) void k_rows(const u32x4* x, u32x4* y, long rows) {   // y may alias x
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    long r = wave * RW, r1 = r + RW;
    if (r1 > rows) r1 = rows;
    if (r >= r1) return;
    u32x4 n0 = x[r * 128 + lane], n1 = x[r * 128 + 64 + lane];
    for (; r < r1; ++r) {
        const u32x4 a = n0, b = n1;
        if (r + 1 < r1) { n0 = x[(r + 1) * 128 + lane]; n1 = x[(r + 1) * 128 + 64 + lane]; }
        y[r * 128 + lane] = a + 1u;
        y[r * 128 + 64 + lane] = b + 1u;
    }
}

template <typename F>
double time_ms(F&& f, int iters = 6) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main() {
    const size_t BYTES = (size_t)1 << 30;              // 1 GiB per stream: 4x the Infinity Cache
    const size_t n = BYTES / 16;
    u32x4 *A, *B, *C, *D;
    unsigned* sink;
    CK(hipMalloc(&A, 2 * BYTES)); CK(hipMalloc(&B, 2 * BYTES)); CK(hipMalloc(&C, BYTES)); CK(hipMalloc(&D, 2 * BYTES)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(A, 1, 2 * BYTES)); CK(hipMemset(B, 2, 2 * BYTES)); CK(hipMemset(C, 3, BYTES)); CK(hipMemset(D, 4, 2 * BYTES));
    auto rep = [&](const char* name, double bytes, double ms) { printf("%-64s %8.1f us  %7.2f TB/s\n", name, ms * 1e3, bytes / ms / 1e9); fflush(stdout); };
    for (int grid : {2048, 8192}) {
        printf("-- flat grid-stride kernels, %d workgroups of 256\n", grid);
        rep("read 1 GiB, 4 x 16 B per lane in flight", BYTES, time_ms([&] { k_read<4, false><<<grid, 256>>>(A, n, sink); }));
        rep("read 1 GiB, 8 x 16 B per lane in flight", BYTES, time_ms([&] { k_read<8, false><<<grid, 256>>>(A, n, sink); }));
        rep("read 1 GiB, non-temporal", BYTES, time_ms([&] { k_read<4, true><<<grid, 256>>>(A, n, sink); }));
        rep("write 1 GiB", BYTES, time_ms([&] { k_write<4, false><<<grid, 256>>>(A, n); }));
        rep("write 1 GiB, non-temporal", BYTES, time_ms([&] { k_write<4, true><<<grid, 256>>>(A, n); }));
        rep("copy A -> B (1 + 1 GiB)", 2.0 * BYTES, time_ms([&] { k_copy<4, false><<<grid, 256>>>(A, B, n); }));
        rep("copy A -> B, non-temporal", 2.0 * BYTES, time_ms([&] { k_copy<4, true><<<grid, 256>>>(A, B, n); }));
        rep("in place A -> A (1 + 1 GiB)", 2.0 * BYTES, time_ms([&] { k_copy<4, false><<<grid, 256>>>(A, A, n); }));
        rep("in place A -> A, non-temporal", 2.0 * BYTES, time_ms([&] { k_copy<4, true><<<grid, 256>>>(A, A, n); }));
    }
    // SwiGLU shapes: M = 65536 rows, f = 2816 bf16 (352 words) -> h12 704 MB, a 352 MB
    {
        const long M = 131072, f16 = 352;
        const dim3 g((f16 + 63) / 64, 2048);
        const double b_fwd = (double)M * f16 * 16 * 3, b_bwd = (double)M * f16 * 16 * 5;
        printf("-- SwiGLU shapes, %ld rows x %ld bf16 per half row\n", M, f16 * 8);
        rep("2 in 1 out (swiglu fwd), 4 rows in flight", b_fwd, time_ms([&] { k_2in1out<4, false><<<g, 256>>>(A, C, M, f16); }));
        rep("2 in 1 out, non-temporal", b_fwd, time_ms([&] { k_2in1out<4, true><<<g, 256>>>(A, C, M, f16); }));
        rep("3 in 2 out, out of place (swiglu bwd), 2 rows in flight", b_bwd, time_ms([&] { k_3in2out<2, false><<<g, 256>>>(C, A, B, M, f16); }));
        rep("3 in 2 out, outputs overwrite h12 in place", b_bwd, time_ms([&] { k_3in2out<2, false><<<g, 256>>>(C, A, A, M, f16); }));
        rep("3 in 2 out, out of place, non-temporal", b_bwd, time_ms([&] { k_3in2out<2, true><<<g, 256>>>(C, A, B, M, f16); }));
        rep("3 in 2 out, in place, non-temporal", b_bwd, time_ms([&] { k_3in2out<2, true><<<g, 256>>>(C, A, A, M, f16); }));
        rep("3 in 2 out, in place, 4 rows in flight", b_bwd, time_ms([&] { k_3in2out<4, false><<<g, 256>>>(C, A, A, M, f16); }));
    }
    // LayerNorm shape: 2 KiB rows (1024 bf16), 262144 rows = 512 MiB per stream
    {
        const long rows = 262144 * 2;
        const double by = (double)rows * 2048 * 2;
        printf("-- wave-per-row walk, %ld rows of 2 KiB\n", rows);
        rep("rows x -> y, 4 rows per wave", by, time_ms([&] { k_rows<4><<<(unsigned)((rows / 4 + 3) / 4), 256>>>(A, B, rows); }));
        rep("rows x -> y, 8 rows per wave", by, time_ms([&] { k_rows<8><<<(unsigned)((rows / 8 + 3) / 4), 256>>>(A, B, rows); }));
        rep("rows x -> y, 16 rows per wave", by, time_ms([&] { k_rows<16><<<(unsigned)((rows / 16 + 3) / 4), 256>>>(A, B, rows); }));
        rep("rows x -> x in place, 4 rows per wave", by, time_ms([&] { k_rows<4><<<(unsigned)((rows / 4 + 3) / 4), 256>>>(A, A, rows); }));
        rep("rows x -> x in place, 16 rows per wave", by, time_ms([&] { k_rows<16><<<(unsigned)((rows / 16 + 3) / 4), 256>>>(A, A, rows); }));
    }
    return 0;
}

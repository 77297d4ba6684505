// Does ordinary VALU work steal matrix-pipe time?  Each wave issues 16 MFMAs (32x32x2 f32) plus NV independent VALU ops per
// iteration; W waves per SIMD.  If MFMA throughput falls as NV grows, the K-loop's address / activation VALU work is not free.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a0 * i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV / 8; ++q) v[(r + q) & 7] = __builtin_fmaf(v[(r + q) & 7], 1.0001f, 0.5f);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV>
static void run(int w) {
    const int blocks = 256 * w, iters = 1200 / w;
    float* out;
    (void)hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NV>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NV>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 16.0 * 4096.0;
    printf("VALU per 16 MFMA = %3d, waves/SIMD = %d: %.3f ms  %.1f TFLOP/s\n", NV, w, ms, flops / ms / 1e9);
    (void)hipFree(out);
}

int main() {
    for (int w : {1, 2, 4, 5}) {
        run<0>(w); run<8>(w); run<16>(w); run<32>(w); run<64>(w); run<128>(w);
    }
    return 0;
}

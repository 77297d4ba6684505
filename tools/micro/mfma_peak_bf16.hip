// Register-only bf16 MFMA throughput on gfx950 (v_mfma_f32_32x32x16_bf16), NACC independent accumulators per wave, W waves per SIMD,
// no memory traffic: what the matrix pipe sustains at the board's power limit (the shader clock gives way under this instruction).
// Reported in TFLOP/s of bf16 work and as the fp32-equivalent rate of the bf16x3 scheme (6 MFMAs per fp32 block: divide by 6).
//   hipcc --offload-arch=gfx950 -O3 mfma_peak_bf16.hip -o mfma_peak_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, unsigned a0, unsigned b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const u32x4 au = {a0 + threadIdx.x, a0 ^ 0x3f803f80u, a0 + 7u * threadIdx.x, a0};
    const u32x4 bu = {b0, b0 + threadIdx.x, b0 ^ 0x3f803f80u, b0 + 3u};
    const bf16x8 a = __builtin_bit_cast(bf16x8, au), b = __builtin_bit_cast(bf16x8, bu);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
static void run(int wgs_per_cu, int iters) {
    const int blocks = 256 * wgs_per_cu;
    float* out;
    (void)hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 0x3f803f80u, 0x3f003f00u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 0x3f803f80u, 0x3f003f00u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double flops = (double)blocks * 4 /*waves*/ * iters * 8.0 * NACC * 32768.0;
    printf("NACC=%d waves/SIMD=%d: %.3f ms  %.0f TFLOP/s bf16  (= %.0f TFLOP/s fp32-equivalent at 6 MFMAs per block)\n", NACC, wgs_per_cu, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 6);
    (void)hipFree(out);
}

int main() {
    for (int w = 1; w <= 4; ++w) {
        run<1>(w, 8000 / w);
        run<4>(w, 2000 / w);
    }
    return 0;
}

// Shared device/host helpers for libgim_hip (gfx950 / CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gim_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

extern "C" void gim_set_error(const char* msg);

#define GIM_CHECK_ARG(cond, msg)      \
    do {                              \
        if (!(cond)) {                \
            gim_set_error(msg);       \
            return GIM_E_BADARG;      \
        }                             \
    } while (0)

static inline int gim_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        char buf[256];
        snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
        gim_set_error(buf);
        return GIM_E_LAUNCH;
    }
    return GIM_OK;
}

static inline int ilog2_exact(int v) {  // returns -1 if v is not a power of two
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum over a 256-thread block; result valid in every thread. `red` = 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ float lrelu_f(float v, float slope) { return v > 0.f ? v : v * slope; }

// conv_tiny.hip: direct convolution of the 3 -> 3 / 1 -> 1 image layers (3x3, 9x9; plain geometry).  Each returns false when the
// shape is not one of them (the implicit-GEMM kernels run); plan_out != nullptr: record the launch plan, launch nothing.
bool gim_tiny_shape(const gim_conv_shape* s);
bool gim_tiny_fwd(const float* x, const float* w, const float* bias, const float* sigma, const float* res, float* y,
                  const gim_conv_shape* s, hipStream_t st, int32_t* plan_out);
bool gim_tiny_dgrad(const float* dy, const float* w, const float* sigma, const float* mask_x, float* dx, const gim_conv_shape* s,
                    hipStream_t st, int32_t* plan_out);
bool gim_tiny_wgrad_acc(const float* dy, const float* x, float* acc, float* bias_acc, const gim_conv_shape* s, hipStream_t st,
                        int32_t* plan_out);

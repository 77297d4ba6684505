// Fused multi-tensor Adam over flat parameter / gradient / moment buffers (HBM-bound: 16 B read +
// 12 B written per parameter) plus the library-wide error string.
//
// torch.optim.Adam form (no weight decay, no amsgrad):
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
//   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// The step counter lives in device memory and is advanced by a 1-thread kernel enqueued after the
// update, so a captured hipGraph replays correctly.
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";
extern "C" void gim_set_error(const char* msg) {
    strncpy(g_err, msg, sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* gim_last_error(void) { return g_err; }
extern "C" int gim_version(void) { return 1; }

#define ADAM_MAX_SEG 16

struct AdamSeg {
    long long end[ADAM_MAX_SEG];
};

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, const long long* __restrict__ seg_end,
                                                   const float* __restrict__ lr, int n_seg, float b1, float b2, float eps,
                                                   float gscale, const int32_t* __restrict__ step) {
    __shared__ float s_bc1, s_bc2s;
    __shared__ long long s_end[ADAM_MAX_SEG];
    __shared__ float s_lr[ADAM_MAX_SEG];
    if (threadIdx.x == 0) {
        const double t = (double)(step[0] + 1);
        s_bc1 = (float)(1.0 - pow((double)b1, t));
        s_bc2s = (float)sqrt(1.0 - pow((double)b2, t));
    }
    if (threadIdx.x < n_seg) {
        s_end[threadIdx.x] = seg_end[threadIdx.x];
        s_lr[threadIdx.x] = lr[threadIdx.x];
    }
    __syncthreads();
    const float bc1 = s_bc1, bc2s = s_bc2s;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        int sg = 0;
        while (sg < n_seg - 1 && i >= s_end[sg]) ++sg;
        const float gr = g[i] * gscale;
        const float mi = b1 * m[i] + (1.0f - b1) * gr;
        const float vi = b2 * v[i] + (1.0f - b2) * gr * gr;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2s + eps;
        p[i] -= (s_lr[sg] / bc1) * (mi / denom);
    }
}

__global__ void adam_advance_kernel(int32_t* step) { step[0] += 1; }

extern "C" int gim_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* seg_end, const float* lr,
                             int n_seg, float beta1, float beta2, float eps, float grad_scale, int32_t* step, void* stream) {
    GIM_CHECK_ARG(p && g && m && v && seg_end && lr && step && n > 0, "adam_step: bad args");
    GIM_CHECK_ARG(n_seg >= 1 && n_seg <= ADAM_MAX_SEG, "adam_step: 1..16 segments");
    long long blocks = (n + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_kernel, dim3((int)blocks), dim3(256), 0, st, p, g, m, v, (long long)n,
                       reinterpret_cast<const long long*>(seg_end), lr, n_seg, beta1, beta2, eps, grad_scale, step);
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, st, step);
    return gim_check_launch("gim_adam_step");
}

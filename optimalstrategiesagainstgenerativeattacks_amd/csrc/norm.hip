// InstanceNorm2d(affine) and AdaIN on NHWC data: per-(n, c) statistics over H*W.  HBM-bound.
// One workgroup = one image x 64 channels; lanes run along the channel axis (coalesced NHWC rows),
// thread groups split the H*W axis, LDS combines the groups.  Two-pass variance (mean first), so a
// spatially constant map yields variance exactly 0 and the output is exactly `shift`
// (defined behaviour for zero-variance input).
#include "common.h"

// CB = channels per workgroup (64, or 4 for the 1/3-channel image-sized maps: then 64 thread groups split H*W)
template <int VEC, int CB>
struct NormCfg {
    static constexpr int CL = CB / VEC;   // lanes along channels
    static constexpr int HG = 256 / CL;   // groups along H*W
};

template <int VEC>
__device__ __forceinline__ void ld_vec(const float* p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {
        v[0] = p[0];
    }
}
template <int VEC>
__device__ __forceinline__ void st_vec(float* p, const float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        f32x4 t = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p) = t;
    } else {
        p[0] = v[0];
    }
}

// reduce VEC values per thread across the HG groups; result broadcast to every thread of a channel lane
template <int VEC, int CB>
__device__ __forceinline__ void group_reduce(float (&v)[VEC], float* red, int cl, int hg) {
    constexpr int HG = NormCfg<VEC, CB>::HG;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[hg * CB + cl * VEC + e] = v[e];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        float s = 0.f;
        for (int g = 0; g < HG; ++g) s += red[g * CB + cl * VEC + e];
        v[e] = s;
    }
}

template <int VEC, int CB>
__global__ __launch_bounds__(256) void norm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ res,
                                                       float* __restrict__ y, float* __restrict__ stats, int HW, int C, int mode,
                                                       float eps, float post_slope) {
    constexpr int CL = NormCfg<VEC, CB>::CL, HG = NormCfg<VEC, CB>::HG;
    __shared__ float red[HG * CB];
    const int n = blockIdx.y;
    const int cl = threadIdx.x % CL, hg = threadIdx.x / CL;
    const int c = blockIdx.x * CB + cl * VEC;
    const bool ok = c < C;
    const float* xb = x + (long long)n * HW * C + c;
    float mean[VEC], ssq[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) mean[e] = 0.f;
    if (ok)
        for (int i = hg; i < HW; i += HG) {
            float v[VEC];
            ld_vec<VEC>(xb + (long long)i * C, v);
#pragma unroll
            for (int e = 0; e < VEC; ++e) mean[e] += v[e];
        }
    group_reduce<VEC, CB>(mean, red, cl, hg);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        mean[e] /= (float)HW;
        ssq[e] = 0.f;
    }
    if (ok)
        for (int i = hg; i < HW; i += HG) {
            float v[VEC];
            ld_vec<VEC>(xb + (long long)i * C, v);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float d = v[e] - mean[e];
                ssq[e] += d * d;
            }
        }
    group_reduce<VEC, CB>(ssq, red, cl, hg);
    if (!ok) return;
    float invd[VEC], sc[VEC], sh[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        float d, c2;
        if (mode == 0) {
            d = sqrtf(ssq[e] / (float)HW + eps);
            c2 = 1.0f / ((float)HW * d);
        } else {
            const float sd = sqrtf(ssq[e] / (float)(HW - 1));
            d = sd + eps;
            c2 = sd > 0.f ? 1.0f / ((float)(HW - 1) * sd) : 0.f;  // zero-variance map: no gradient through the std
        }
        invd[e] = 1.0f / d;
        const long long pi = (mode == 0) ? (long long)(c + e) : (long long)n * C + c + e;
        sc[e] = scale[pi];
        sh[e] = shift[pi];
        if (hg == 0) {
            float* st = stats + ((long long)n * C + c + e) * 3;
            st[0] = mean[e];
            st[1] = invd[e];
            st[2] = c2;
        }
    }
    float* yb = y + (long long)n * HW * C + c;
    const float* rb = res ? res + (long long)n * HW * C + c : nullptr;
    for (int i = hg; i < HW; i += HG) {
        float v[VEC], o[VEC];
        ld_vec<VEC>(xb + (long long)i * C, v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = sc[e] * ((v[e] - mean[e]) * invd[e]) + sh[e];
        if (rb) {
            float rr[VEC];
            ld_vec<VEC>(rb + (long long)i * C, rr);
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] += rr[e];
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = fmaxf(o[e], o[e] * post_slope);   // 1 = identity (gim_norm_fwd_act)
        st_vec<VEC>(yb + (long long)i * C, o);
    }
}

template <int VEC, int CB>
__global__ __launch_bounds__(256) void norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                       const float* __restrict__ scale, const float* __restrict__ stats,
                                                       float* __restrict__ dx, float* __restrict__ dscale_nc,
                                                       float* __restrict__ dshift_nc, int HW, int C, int mode) {
    constexpr int CL = NormCfg<VEC, CB>::CL, HG = NormCfg<VEC, CB>::HG;
    __shared__ float red[HG * CB];
    const int n = blockIdx.y;
    const int cl = threadIdx.x % CL, hg = threadIdx.x / CL;
    const int c = blockIdx.x * CB + cl * VEC;
    const bool ok = c < C;
    const long long off = (long long)n * HW * C + c;
    float mean[VEC], invd[VEC], c2[VEC], sc[VEC], s1[VEC], s2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        s1[e] = 0.f;
        s2[e] = 0.f;
        mean[e] = invd[e] = c2[e] = sc[e] = 0.f;
        if (ok) {
            const float* st = stats + ((long long)n * C + c + e) * 3;
            mean[e] = st[0];
            invd[e] = st[1];
            c2[e] = st[2];
            sc[e] = scale[(mode == 0) ? (long long)(c + e) : (long long)n * C + c + e];
        }
    }
    if (ok)
        for (int i = hg; i < HW; i += HG) {
            float g[VEC], v[VEC];
            ld_vec<VEC>(dy + off + (long long)i * C, g);
            ld_vec<VEC>(x + off + (long long)i * C, v);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                s1[e] += g[e];
                s2[e] += g[e] * ((v[e] - mean[e]) * invd[e]);
            }
        }
    group_reduce<VEC, CB>(s1, red, cl, hg);
    group_reduce<VEC, CB>(s2, red, cl, hg);
    if (!ok) return;
    if (hg == 0) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            dscale_nc[(long long)n * C + c + e] = s2[e];
            dshift_nc[(long long)n * C + c + e] = s1[e];
        }
    }
    const float inv_hw = 1.0f / (float)HW;
    for (int i = hg; i < HW; i += HG) {
        float g[VEC], v[VEC], o[VEC];
        ld_vec<VEC>(dy + off + (long long)i * C, g);
        ld_vec<VEC>(x + off + (long long)i * C, v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float xh = (v[e] - mean[e]) * invd[e];
            o[e] = sc[e] * ((g[e] - s1[e] * inv_hw) * invd[e] - xh * s2[e] * c2[e]);
        }
        st_vec<VEC>(dx + off + (long long)i * C, o);
    }
}

static int norm_fwd_impl(const float* x, const float* scale, const float* shift, const float* residual, float* y,
                         float* stats, int N, int HW, int C, int mode, float eps, float post_slope, void* stream) {
    GIM_CHECK_ARG(x && scale && shift && y && stats, "norm_fwd: null pointer");
    GIM_CHECK_ARG(N > 0 && HW > 0 && C > 0 && (mode == 0 || mode == 1), "norm_fwd: bad dims");
    const bool vec = (C % 4 == 0) && !(((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual) & 15);
    hipStream_t st = (hipStream_t)stream;
    if (vec) hipLaunchKernelGGL((norm_fwd_kernel<4, 64>), dim3((C + 63) / 64, N), dim3(256), 0, st, x, scale, shift, residual, y, stats, HW, C, mode, eps, post_slope);
    else if (C <= 4) hipLaunchKernelGGL((norm_fwd_kernel<1, 4>), dim3(1, N), dim3(256), 0, st, x, scale, shift, residual, y, stats, HW, C, mode, eps, post_slope);
    else hipLaunchKernelGGL((norm_fwd_kernel<1, 64>), dim3((C + 63) / 64, N), dim3(256), 0, st, x, scale, shift, residual, y, stats, HW, C, mode, eps, post_slope);
    return gim_check_launch("gim_norm_fwd");
}

extern "C" int gim_norm_fwd(const float* x, const float* scale, const float* shift, const float* residual, float* y,
                            float* stats, int N, int HW, int C, int mode, float eps, void* stream) {
    return norm_fwd_impl(x, scale, shift, residual, y, stats, N, HW, C, mode, eps, 1.0f, stream);
}

extern "C" int gim_norm_fwd_act(const float* x, const float* scale, const float* shift, const float* residual, float* y,
                                float* stats, int N, int HW, int C, int mode, float eps, float post_slope, void* stream) {
    GIM_CHECK_ARG(post_slope > 0.f && post_slope <= 1.f, "norm_fwd_act: post_slope must be in (0, 1]");
    return norm_fwd_impl(x, scale, shift, residual, y, stats, N, HW, C, mode, eps, post_slope, stream);
}

extern "C" int gim_norm_bwd(const float* dy, const float* x, const float* scale, const float* stats, float* dx,
                            float* dscale_nc, float* dshift_nc, int N, int HW, int C, int mode, void* stream) {
    GIM_CHECK_ARG(dy && x && scale && stats && dx && dscale_nc && dshift_nc, "norm_bwd: null pointer");
    GIM_CHECK_ARG(N > 0 && HW > 0 && C > 0 && (mode == 0 || mode == 1), "norm_bwd: bad dims");
    const bool vec = (C % 4 == 0) && !(((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15);
    hipStream_t st = (hipStream_t)stream;
    if (vec) hipLaunchKernelGGL((norm_bwd_kernel<4, 64>), dim3((C + 63) / 64, N), dim3(256), 0, st, dy, x, scale, stats, dx, dscale_nc, dshift_nc, HW, C, mode);
    else if (C <= 4) hipLaunchKernelGGL((norm_bwd_kernel<1, 4>), dim3(1, N), dim3(256), 0, st, dy, x, scale, stats, dx, dscale_nc, dshift_nc, HW, C, mode);
    else hipLaunchKernelGGL((norm_bwd_kernel<1, 64>), dim3((C + 63) / 64, N), dim3(256), 0, st, dy, x, scale, stats, dx, dscale_nc, dshift_nc, HW, C, mode);
    return gim_check_launch("gim_norm_bwd");
}

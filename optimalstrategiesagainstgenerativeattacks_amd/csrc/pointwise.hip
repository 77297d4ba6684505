// HBM-bound pointwise / pooling / small-reduction kernels of the GIM hot path (NHWC, fp32).
#include "common.h"

#define PW_MAX_BLOCKS 2048

static inline int pw_blocks(long long n) {
    long long b = (n + 255) / 256;
    if (b > PW_MAX_BLOCKS) b = PW_MAX_BLOCKS;
    if (b < 1) b = 1;
    return (int)b;
}
#define GRID_STRIDE(i, n) for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (n); i += (long long)gridDim.x * 256)

// ---------------------------------------------------------------- avg pool 2x2
__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long long n_out = (long long)N * Ho * Wo * C;
    GRID_STRIDE(i, n_out) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int n = (int)(r / Ho);
        const float* p = x + (((long long)n * H + 2 * ho) * W + 2 * wo) * C + c;
        y[i] = 0.25f * (p[0] + p[C] + p[(long long)W * C] + p[(long long)W * C + C]);
    }
}
// x stored ACTIVATED (lrelu(x, slope) written by its producer, see gim_conv_shape.post_slope): the pool needs the raw values,
// and LeakyReLU is invertible - x = a for a > 0, a / slope otherwise (inv = 1 / slope; one rounding more than the raw tensor)
__global__ __launch_bounds__(256) void avgpool2_fwd_act_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C, float inv) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long long n_out = (long long)N * Ho * Wo * C;
    GRID_STRIDE(i, n_out) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int n = (int)(r / Ho);
        const float* p = x + (((long long)n * H + 2 * ho) * W + 2 * wo) * C + c;
        const float a0 = p[0], a1 = p[C], a2 = p[(long long)W * C], a3 = p[(long long)W * C + C];
        y[i] = 0.25f * ((a0 > 0.f ? a0 : a0 * inv) + (a1 > 0.f ? a1 : a1 * inv) + (a2 > 0.f ? a2 : a2 * inv) + (a3 > 0.f ? a3 : a3 * inv));
    }
}
__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W, int C) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long long n_in = (long long)N * H * W * C;
    GRID_STRIDE(i, n_in) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H);
        const int n = (int)(r / H);
        dx[i] = 0.25f * dy[(((long long)n * Ho + (h >> 1)) * Wo + (w >> 1)) * C + c];
    }
}
extern "C" int gim_avgpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    GIM_CHECK_ARG(x && y && N > 0 && H >= 2 && W >= 2 && !(H & 1) && !(W & 1) && C > 0, "avgpool2_fwd: bad args");
    const long long n = (long long)N * (H / 2) * (W / 2) * C;
    hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(pw_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C);
    return gim_check_launch("gim_avgpool2_fwd");
}
extern "C" int gim_avgpool2_fwd_act(const float* x, float* y, int N, int H, int W, int C, float in_slope, void* stream) {
    GIM_CHECK_ARG(x && y && N > 0 && H >= 2 && W >= 2 && !(H & 1) && !(W & 1) && C > 0 && in_slope > 0.f && in_slope <= 1.f, "avgpool2_fwd_act: bad args");
    const long long n = (long long)N * (H / 2) * (W / 2) * C;
    hipLaunchKernelGGL(avgpool2_fwd_act_kernel, dim3(pw_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C, 1.0f / in_slope);
    return gim_check_launch("gim_avgpool2_fwd_act");
}
// Gradient fan-in of a tensor with several consumers, in one pass (autograd would add the incoming gradients pairwise, one
// launch and one full read-modify-write each): out = a + b (+ c) (+ d);  and the ResBlockDown form, where the second consumer
// is the 2x2 average pool of the skip path: out = g + avgpool2_bwd(dy_pooled) without materialising the un-pooled gradient.
__global__ __launch_bounds__(256) void add_n_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                    const float* __restrict__ d, float* __restrict__ out, long long n) {
    GRID_STRIDE(i, n) {
        float v = a[i] + b[i];
        if (c) v += c[i];
        if (d) v += d[i];
        out[i] = v;
    }
}
__global__ __launch_bounds__(256) void add_avgpool2_bwd_kernel(const float* __restrict__ g, const float* __restrict__ dy, float* __restrict__ out,
                                                               int N, int H, int W, int C) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long long n_in = (long long)N * H * W * C;
    GRID_STRIDE(i, n_in) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H);
        const int n = (int)(r / H);
        out[i] = g[i] + 0.25f * dy[(((long long)n * Ho + (h >> 1)) * Wo + (w >> 1)) * C + c];
    }
}
extern "C" int gim_add_n(const float* a, const float* b, const float* c, const float* d, float* out, int64_t n, void* stream) {
    GIM_CHECK_ARG(a && b && out && n > 0 && (c || !d), "add_n: bad args");
    hipLaunchKernelGGL(add_n_kernel, dim3(pw_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, c, d, out, (long long)n);
    return gim_check_launch("gim_add_n");
}
extern "C" int gim_add_avgpool2_bwd(const float* g, const float* dy, float* out, int N, int H, int W, int C, void* stream) {
    GIM_CHECK_ARG(g && dy && out && N > 0 && H >= 2 && W >= 2 && !(H & 1) && !(W & 1) && C > 0, "add_avgpool2_bwd: bad args");
    const long long n = (long long)N * H * W * C;
    hipLaunchKernelGGL(add_avgpool2_bwd_kernel, dim3(pw_blocks(n)), dim3(256), 0, (hipStream_t)stream, g, dy, out, N, H, W, C);
    return gim_check_launch("gim_add_avgpool2_bwd");
}
extern "C" int gim_avgpool2_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
    GIM_CHECK_ARG(dy && dx && N > 0 && H >= 2 && W >= 2 && !(H & 1) && !(W & 1) && C > 0, "avgpool2_bwd: bad args");
    const long long n = (long long)N * H * W * C;
    hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(pw_blocks(n)), dim3(256), 0, (hipStream_t)stream, dy, dx, N, H, W, C);
    return gim_check_launch("gim_avgpool2_bwd");
}

// ---------------------------------------------------------------- nearest x2 upsample backward
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ dyu, const float* __restrict__ mx, float slope,
                                                             float* __restrict__ dx, int N, int H, int W, int C) {
    const long long n_lo = (long long)N * H * W * C;
    const int W2 = 2 * W;
    GRID_STRIDE(i, n_lo) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H);
        const int n = (int)(r / H);
        const float* p = dyu + (((long long)n * 2 * H + 2 * h) * W2 + 2 * w) * C + c;
        float g = p[0] + p[C] + p[(long long)W2 * C] + p[(long long)W2 * C + C];
        if (mx) g *= (mx[i] > 0.f ? 1.0f : slope);
        dx[i] = g;
    }
}
extern "C" int gim_upsample2x_bwd(const float* dy_up, const float* mask_x, float slope, float* dx, int N, int H, int W, int C, void* stream) {
    GIM_CHECK_ARG(dy_up && dx && N > 0 && H > 0 && W > 0 && C > 0, "upsample2x_bwd: bad args");
    const long long n = (long long)N * H * W * C;
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(pw_blocks(n)), dim3(256), 0, (hipStream_t)stream, dy_up, mask_x, slope, dx, N, H, W, C);
    return gim_check_launch("gim_upsample2x_bwd");
}

// ---------------------------------------------------------------- global max pool + leaky relu
__global__ __launch_bounds__(256) void maxpool_lrelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int32_t* __restrict__ idx,
                                                                int HW, int C, float slope) {
    __shared__ float sv[4][64];
    __shared__ int si[4][64];
    const int n = blockIdx.y, cl = threadIdx.x & 63, hg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if (c < C) {
        const float* xb = x + (long long)n * HW * C + c;
        for (int i = hg; i < HW; i += 4) {
            const float v = xb[(long long)i * C];
            if (v > best || bi == 0x7fffffff) { best = v; bi = i; }
        }
    }
    sv[hg][cl] = best;
    si[hg][cl] = bi;
    __syncthreads();
    if (hg == 0 && c < C) {
        for (int g = 1; g < 4; ++g) {
            const float v = sv[g][cl];
            const int i2 = si[g][cl];
            if (i2 != 0x7fffffff && (v > best || (v == best && i2 < bi))) { best = v; bi = i2; }
        }
        y[(long long)n * C + c] = lrelu_f(best, slope);
        idx[(long long)n * C + c] = bi;
    }
}
__global__ __launch_bounds__(256) void maxpool_lrelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const int32_t* __restrict__ idx,
                                                                float* __restrict__ dx, int N, int HW, int C, float slope) {
    const long long n_nc = (long long)N * C;
    GRID_STRIDE(i, n_nc) {
        const int c = (int)(i % C);
        const long long n = i / C;
        dx[(n * HW + idx[i]) * C + c] = dy[i] * (y[i] > 0.f ? 1.0f : slope);
    }
}
extern "C" int gim_maxpool_lrelu_fwd(const float* x, float* y, int32_t* idx, int N, int HW, int C, float slope, void* stream) {
    GIM_CHECK_ARG(x && y && idx && N > 0 && HW > 0 && C > 0, "maxpool_lrelu_fwd: bad args");
    hipLaunchKernelGGL(maxpool_lrelu_fwd_kernel, dim3((C + 63) / 64, N), dim3(256), 0, (hipStream_t)stream, x, y, idx, HW, C, slope);
    return gim_check_launch("gim_maxpool_lrelu_fwd");
}
extern "C" int gim_maxpool_lrelu_bwd(const float* dy, const float* y, const int32_t* idx, float* dx, int N, int HW, int C, float slope, void* stream) {
    GIM_CHECK_ARG(dy && y && idx && dx && N > 0 && HW > 0 && C > 0, "maxpool_lrelu_bwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(dx, 0, (size_t)N * HW * C * sizeof(float), st) != hipSuccess) return gim_check_launch("maxpool_lrelu_bwd memset");
    hipLaunchKernelGGL(maxpool_lrelu_bwd_kernel, dim3(pw_blocks((long long)N * C)), dim3(256), 0, st, dy, y, idx, dx, N, HW, C, slope);
    return gim_check_launch("gim_maxpool_lrelu_bwd");
}

// ---------------------------------------------------------------- softmax over dim -2 of [B][R][Cc]
__global__ __launch_bounds__(256) void softmax_dim1_fwd_kernel(const float* __restrict__ s, float* __restrict__ p, int R, int Cc) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    const bool ok = col < Cc;
    const float* sb = s + (long long)b * R * Cc + col;
    float mx = -INFINITY;
    if (ok) for (int r = rg; r < R; r += 4) mx = fmaxf(mx, sb[(long long)r * Cc]);
    red[rg][cl] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][cl], red[1][cl]), fmaxf(red[2][cl], red[3][cl]));
    __syncthreads();
    float sum = 0.f;
    if (ok) for (int r = rg; r < R; r += 4) sum += __expf(sb[(long long)r * Cc] - mx);
    red[rg][cl] = sum;
    __syncthreads();
    sum = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    if (!ok) return;
    const float inv = 1.0f / sum;
    float* pb = p + (long long)b * R * Cc + col;
    for (int r = rg; r < R; r += 4) pb[(long long)r * Cc] = __expf(sb[(long long)r * Cc] - mx) * inv;
}
__global__ __launch_bounds__(256) void softmax_dim1_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ p, float* __restrict__ ds, int R, int Cc) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    const bool ok = col < Cc;
    const long long base = (long long)b * R * Cc + col;
    float dot = 0.f;
    if (ok) for (int r = rg; r < R; r += 4) dot += dp[base + (long long)r * Cc] * p[base + (long long)r * Cc];
    red[rg][cl] = dot;
    __syncthreads();
    dot = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    if (!ok) return;
    for (int r = rg; r < R; r += 4) {
        const long long o = base + (long long)r * Cc;
        ds[o] = p[o] * (dp[o] - dot);
    }
}
// The same two operators with the column held in registers (R <= 4 * RPT rows: the 16x16 attention map has R = 256): ONE pass over
// memory with all of a thread's loads in flight at once instead of three dependent strided sweeps; same arithmetic in the same
// order, so bit-identical to the kernels above.
template <int RPT>
__global__ __launch_bounds__(256) void softmax_dim1_fwd_reg_kernel(const float* __restrict__ s, float* __restrict__ p, int R, int Cc) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    const bool ok = col < Cc;
    const float* sb = s + (long long)b * R * Cc + col;
    float v[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = rg + 4 * i;
        v[i] = (ok && r < R) ? sb[(long long)r * Cc] : -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < RPT; ++i) mx = fmaxf(mx, v[i]);
    red[rg][cl] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][cl], red[1][cl]), fmaxf(red[2][cl], red[3][cl]));
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = rg + 4 * i;
        if (ok && r < R) {
            v[i] = __expf(v[i] - mx);
            sum += v[i];
        }
    }
    red[rg][cl] = sum;
    __syncthreads();
    sum = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    if (!ok) return;
    const float inv = 1.0f / sum;
    float* pb = p + (long long)b * R * Cc + col;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = rg + 4 * i;
        if (r < R) pb[(long long)r * Cc] = v[i] * inv;
    }
}
template <int RPT>
__global__ __launch_bounds__(256) void softmax_dim1_bwd_reg_kernel(const float* __restrict__ dp, const float* __restrict__ p, float* __restrict__ ds, int R, int Cc) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    const bool ok = col < Cc;
    const long long base = (long long)b * R * Cc + col;
    float vp[RPT], vd[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = rg + 4 * i;
        const bool in = ok && r < R;
        vp[i] = in ? p[base + (long long)r * Cc] : 0.f;
        vd[i] = in ? dp[base + (long long)r * Cc] : 0.f;
    }
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = rg + 4 * i;
        if (ok && r < R) dot += vd[i] * vp[i];
    }
    red[rg][cl] = dot;
    __syncthreads();
    dot = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    if (!ok) return;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = rg + 4 * i;
        if (r < R) ds[base + (long long)r * Cc] = vp[i] * (vd[i] - dot);
    }
}
extern "C" int gim_softmax_dim1_fwd(const float* s, float* p, int B, int R, int Ccols, void* stream) {
    GIM_CHECK_ARG(s && p && B > 0 && R > 0 && Ccols > 0, "softmax_dim1_fwd: bad args");
    const dim3 grid((Ccols + 63) / 64, B);
    hipStream_t st = (hipStream_t)stream;
    if (R <= 64) hipLaunchKernelGGL(softmax_dim1_fwd_reg_kernel<16>, grid, dim3(256), 0, st, s, p, R, Ccols);
    else if (R <= 256) hipLaunchKernelGGL(softmax_dim1_fwd_reg_kernel<64>, grid, dim3(256), 0, st, s, p, R, Ccols);
    else hipLaunchKernelGGL(softmax_dim1_fwd_kernel, grid, dim3(256), 0, st, s, p, R, Ccols);
    return gim_check_launch("gim_softmax_dim1_fwd");
}
extern "C" int gim_softmax_dim1_bwd(const float* dp, const float* p, float* ds, int B, int R, int Ccols, void* stream) {
    GIM_CHECK_ARG(dp && p && ds && B > 0 && R > 0 && Ccols > 0, "softmax_dim1_bwd: bad args");
    const dim3 grid((Ccols + 63) / 64, B);
    hipStream_t st = (hipStream_t)stream;
    if (R <= 64) hipLaunchKernelGGL(softmax_dim1_bwd_reg_kernel<16>, grid, dim3(256), 0, st, dp, p, ds, R, Ccols);
    else if (R <= 256) hipLaunchKernelGGL(softmax_dim1_bwd_reg_kernel<64>, grid, dim3(256), 0, st, dp, p, ds, R, Ccols);
    else hipLaunchKernelGGL(softmax_dim1_bwd_kernel, grid, dim3(256), 0, st, dp, p, ds, R, Ccols);
    return gim_check_launch("gim_softmax_dim1_bwd");
}

// ---------------------------------------------------------------- y = gamma * a + x
__global__ __launch_bounds__(256) void scale_add_fwd_kernel(const float* __restrict__ a, const float* __restrict__ x, const float* __restrict__ gamma,
                                                            float* __restrict__ y, long long n) {
    const float g = gamma[0];
    GRID_STRIDE(i, n) y[i] = g * a[i] + x[i];
}
__global__ __launch_bounds__(256) void scale_add_fwd_act_kernel(const float* __restrict__ a, const float* __restrict__ x, const float* __restrict__ gamma,
                                                                float* __restrict__ y, long long n, float post_slope) {
    const float g = gamma[0];
    GRID_STRIDE(i, n) { const float v = g * a[i] + x[i]; y[i] = fmaxf(v, v * post_slope); }
}
__global__ __launch_bounds__(256) void scale_add_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ a, const float* __restrict__ gamma,
                                                            float* __restrict__ da, float* __restrict__ partial, long long n) {
    __shared__ float red[4];
    const float g = gamma[0];
    float dot = 0.f;
    GRID_STRIDE(i, n) {
        const float d = dy[i];
        da[i] = g * d;
        dot += d * a[i];
    }
    dot = block_sum_256(dot, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = dot;
}
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) out[0] = s;
}
extern "C" int gim_scale_add_fwd(const float* a, const float* x, const float* gamma, float* y, int64_t n, void* stream) {
    GIM_CHECK_ARG(a && x && gamma && y && n > 0, "scale_add_fwd: bad args");
    hipLaunchKernelGGL(scale_add_fwd_kernel, dim3(pw_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, x, gamma, y, (long long)n);
    return gim_check_launch("gim_scale_add_fwd");
}
extern "C" int gim_scale_add_fwd_act(const float* a, const float* x, const float* gamma, float* y, int64_t n, float post_slope, void* stream) {
    GIM_CHECK_ARG(a && x && gamma && y && n > 0 && post_slope > 0.f && post_slope <= 1.f, "scale_add_fwd_act: bad args");
    hipLaunchKernelGGL(scale_add_fwd_act_kernel, dim3(pw_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, x, gamma, y, (long long)n, post_slope);
    return gim_check_launch("gim_scale_add_fwd_act");
}
extern "C" int gim_scale_add_bwd(const float* dy, const float* a, const float* gamma, float* da, float* dgamma, float* scratch,
                                 int64_t n, void* stream) {
    GIM_CHECK_ARG(dy && a && gamma && da && dgamma && scratch && n > 0, "scale_add_bwd: bad args");
    const int blocks = pw_blocks(n);
    hipLaunchKernelGGL(scale_add_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, a, gamma, da, scratch, (long long)n);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, blocks, dgamma);
    return gim_check_launch("gim_scale_add_bwd");
}

// ---------------------------------------------------------------- tanh
__global__ __launch_bounds__(256) void tanh_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
    GRID_STRIDE(i, n) y[i] = tanhf(x[i]);
}
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, long long n) {
    GRID_STRIDE(i, n) { const float t = y[i]; dx[i] = dy[i] * (1.0f - t * t); }
}
extern "C" int gim_tanh_fwd(const float* x, float* y, int64_t n, void* stream) {
    GIM_CHECK_ARG(x && y && n > 0, "tanh_fwd: bad args");
    hipLaunchKernelGGL(tanh_fwd_kernel, dim3(pw_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, y, (long long)n);
    return gim_check_launch("gim_tanh_fwd");
}
extern "C" int gim_tanh_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
    GIM_CHECK_ARG(dy && y && dx && n > 0, "tanh_bwd: bad args");
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(pw_blocks(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, (long long)n);
    return gim_check_launch("gim_tanh_bwd");
}

// ---------------------------------------------------------------- NCHW <-> NHWC
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int HW) {
    const long long n_el = (long long)N * C * HW;
    GRID_STRIDE(i, n_el) {  // i indexes the NHWC output
        const int c = (int)(i % C);
        const long long r = i / C;
        const int hw = (int)(r % HW);
        const long long n = r / HW;
        y[i] = x[(n * C + c) * HW + hw];
    }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int HW) {
    const long long n_el = (long long)N * C * HW;
    GRID_STRIDE(i, n_el) {  // i indexes the NCHW output
        const int hw = (int)(i % HW);
        const long long r = i / HW;
        const int c = (int)(r % C);
        const long long n = r / C;
        y[i] = x[(n * HW + hw) * C + c];
    }
}
extern "C" int gim_nchw_to_nhwc(const float* x, float* y, int N, int C, int HW, void* stream) {
    GIM_CHECK_ARG(x && y && N > 0 && C > 0 && HW > 0, "nchw_to_nhwc: bad args");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(pw_blocks((long long)N * C * HW)), dim3(256), 0, (hipStream_t)stream, x, y, N, C, HW);
    return gim_check_launch("gim_nchw_to_nhwc");
}
extern "C" int gim_nhwc_to_nchw(const float* x, float* y, int N, int C, int HW, void* stream) {
    GIM_CHECK_ARG(x && y && N > 0 && C > 0 && HW > 0, "nhwc_to_nchw: bad args");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(pw_blocks((long long)N * C * HW)), dim3(256), 0, (hipStream_t)stream, x, y, N, C, HW);
    return gim_check_launch("gim_nhwc_to_nchw");
}

// ---------------------------------------------------------------- set statistics over the sample dim
__global__ __launch_bounds__(256) void set_stats_fwd_kernel(const float* __restrict__ x, float* __restrict__ mean, float* __restrict__ sd,
                                                            int B, int t, int D, long long ldm, long long lds) {
    const long long n = (long long)B * D;
    GRID_STRIDE(i, n) {
        const int d = (int)(i % D);
        const long long b = i / D;
        const float* xb = x + b * t * D + d;
        float m = 0.f;
        for (int j = 0; j < t; ++j) m += xb[(long long)j * D];
        m /= (float)t;
        mean[b * ldm + d] = m;
        if (sd) {
            float v = 0.f;
            if (t > 1) {
                float ss = 0.f;
                for (int j = 0; j < t; ++j) { const float dd = xb[(long long)j * D] - m; ss += dd * dd; }
                v = sqrtf(ss / (float)(t - 1) + 1e-8f);
            }
            sd[b * lds + d] = v;
        }
    }
}
__global__ __launch_bounds__(256) void set_stats_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dmean, const float* __restrict__ dsd,
                                                            float* __restrict__ dx, int B, int t, int D, long long ldm, long long lds) {
    const long long n = (long long)B * D;
    GRID_STRIDE(i, n) {
        const int d = (int)(i % D);
        const long long b = i / D;
        const float* xb = x + b * t * D + d;
        float* db = dx + b * t * D + d;
        const float gm = dmean ? dmean[b * ldm + d] / (float)t : 0.f;
        float m = 0.f, coef = 0.f;
        if (dsd && t > 1) {
            for (int j = 0; j < t; ++j) m += xb[(long long)j * D];
            m /= (float)t;
            float ss = 0.f;
            for (int j = 0; j < t; ++j) { const float dd = xb[(long long)j * D] - m; ss += dd * dd; }
            const float sdv = sqrtf(ss / (float)(t - 1) + 1e-8f);
            coef = dsd[b * lds + d] / ((float)(t - 1) * sdv);
        }
        for (int j = 0; j < t; ++j) db[(long long)j * D] = gm + coef * (xb[(long long)j * D] - m);
    }
}
extern "C" int gim_set_stats_fwd(const float* x, float* mean, float* std, int B, int t, int D, int64_t ld_mean, int64_t ld_std, void* stream) {
    GIM_CHECK_ARG(x && mean && B > 0 && t > 0 && D > 0, "set_stats_fwd: bad args");
    hipLaunchKernelGGL(set_stats_fwd_kernel, dim3(pw_blocks((long long)B * D)), dim3(256), 0, (hipStream_t)stream, x, mean, std, B, t, D,
                       (long long)ld_mean, (long long)ld_std);
    return gim_check_launch("gim_set_stats_fwd");
}
extern "C" int gim_set_stats_bwd(const float* x, const float* dmean, const float* dstd, float* dx, int B, int t, int D,
                                 int64_t ld_dmean, int64_t ld_dstd, void* stream) {
    GIM_CHECK_ARG(x && dx && B > 0 && t > 0 && D > 0, "set_stats_bwd: bad args");
    hipLaunchKernelGGL(set_stats_bwd_kernel, dim3(pw_blocks((long long)B * D)), dim3(256), 0, (hipStream_t)stream, x, dmean, dstd, dx, B, t, D,
                       (long long)ld_dmean, (long long)ld_dstd);
    return gim_check_launch("gim_set_stats_bwd");
}

// ---------------------------------------------------------------- BCE with logits, constant target
__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ x, float* __restrict__ loss, float tgt, int n) {
    GRID_STRIDE(i, n) {
        const float v = x[i];
        loss[i] = fmaxf(v, 0.f) - v * tgt + log1pf(expf(-fabsf(v)));
    }
}
__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ dl, const float* __restrict__ x, float* __restrict__ dx, float tgt, int n) {
    GRID_STRIDE(i, n) {
        const float v = x[i];
        const float sg = 1.0f / (1.0f + expf(-v));
        dx[i] = dl[i] * (sg - tgt);
    }
}
extern "C" int gim_bce_logits_fwd(const float* x, float* loss, float target, int n, void* stream) {
    GIM_CHECK_ARG(x && loss && n > 0, "bce_logits_fwd: bad args");
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(pw_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, loss, target, n);
    return gim_check_launch("gim_bce_logits_fwd");
}
extern "C" int gim_bce_logits_bwd(const float* dloss, const float* x, float* dx, float target, int n, void* stream) {
    GIM_CHECK_ARG(dloss && x && dx && n > 0, "bce_logits_bwd: bad args");
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(pw_blocks(n)), dim3(256), 0, (hipStream_t)stream, dloss, x, dx, target, n);
    return gim_check_launch("gim_bce_logits_bwd");
}

// ---------------------------------------------------------------- column sums (bias gradients)
__global__ __launch_bounds__(256) void colsum_part_kernel(const float* __restrict__ x, float* __restrict__ part, long long rows, int C, long long rows_per) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const long long r0 = (long long)blockIdx.y * rows_per;
    const long long r1 = r0 + rows_per < rows ? r0 + rows_per : rows;
    float s = 0.f;
    if (c < C) for (long long r = r0 + rg; r < r1; r += 4) s += x[r * C + c];
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && c < C) part[(long long)blockIdx.y * C + c] = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
}
// Sum of `rows` short rows per column, 64 columns x 4 row groups per block (a single thread walking 80-256 rows of one
// column is a chain of dependent-latency loads: 20-30 us for a few KB).
__device__ __forceinline__ float colsum_rows_64x4(const float* __restrict__ x, int rows, int C, int c, int rg, float (*red)[64], int cl) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < C) {
        int r = rg;
        for (; r + 12 < rows; r += 16) {
            s0 += x[(long long)r * C + c];
            s1 += x[(long long)(r + 4) * C + c];
            s2 += x[(long long)(r + 8) * C + c];
            s3 += x[(long long)(r + 12) * C + c];
        }
        for (; r < rows; r += 4) s0 += x[(long long)r * C + c];
    }
    red[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    const float s = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    __syncthreads();
    return s;
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int P, int C) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    const float s = colsum_rows_64x4(part, P, C, c, rg, red, cl);
    if (rg == 0 && c < C) out[c] = s;
}
extern "C" int gim_colsum(const float* x, float* out, float* scratch, int64_t rows, int C, void* stream) {
    GIM_CHECK_ARG(x && out && scratch && rows > 0 && C > 0, "colsum: bad args");
    long long P = (rows + 63) / 64;
    if (P > 256) P = 256;
    const long long rows_per = (rows + P - 1) / P;
    P = (rows + rows_per - 1) / rows_per;
    hipLaunchKernelGGL(colsum_part_kernel, dim3((C + 63) / 64, (int)P), dim3(256), 0, (hipStream_t)stream, x, scratch, (long long)rows, C, rows_per);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, scratch, out, (int)P, C);
    return gim_check_launch("gim_colsum");
}

// ---------------------------------------------------------------- sums / repeats over the sample dim
// y[b][d] = scale * sum_j x[b][j][d]
__global__ __launch_bounds__(256) void sum_dim1_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int t, int D, float scale) {
    const long long n = (long long)B * D;
    GRID_STRIDE(i, n) {
        const int d = (int)(i % D);
        const long long b = i / D;
        const float* xb = x + b * t * D + d;
        float s = 0.f;
        for (int j = 0; j < t; ++j) s += xb[(long long)j * D];
        y[i] = s * scale;
    }
}
// y[b][j][d] = scale * x[b][d]
__global__ __launch_bounds__(256) void repeat_dim1_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int t, int D, float scale) {
    const long long n = (long long)B * t * D;
    GRID_STRIDE(i, n) {
        const int d = (int)(i % D);
        const long long b = i / ((long long)t * D);
        y[i] = scale * x[b * D + d];
    }
}
extern "C" int gim_sum_dim1(const float* x, float* y, int B, int t, int D, float scale, void* stream) {
    GIM_CHECK_ARG(x && y && B > 0 && t > 0 && D > 0, "sum_dim1: bad args");
    hipLaunchKernelGGL(sum_dim1_kernel, dim3(pw_blocks((long long)B * D)), dim3(256), 0, (hipStream_t)stream, x, y, B, t, D, scale);
    return gim_check_launch("gim_sum_dim1");
}
extern "C" int gim_repeat_dim1(const float* x, float* y, int B, int t, int D, float scale, void* stream) {
    GIM_CHECK_ARG(x && y && B > 0 && t > 0 && D > 0, "repeat_dim1: bad args");
    hipLaunchKernelGGL(repeat_dim1_kernel, dim3(pw_blocks((long long)B * t * D)), dim3(256), 0, (hipStream_t)stream, x, y, B, t, D, scale);
    return gim_check_launch("gim_repeat_dim1");
}

// ---------------------------------------------------------------- generator noise combine
// fwd: y[b][j] = env[b] + w[b][j] - (remove_mean ? mean_j w[b][j] : 0)
// bwd (same kernel, env == NULL): dw[b][j] = dy[b][j] - (remove_mean ? mean_j dy[b][j] : 0)
__global__ __launch_bounds__(256) void noise_combine_kernel(const float* __restrict__ env, const float* __restrict__ w, float* __restrict__ y,
                                                            int B, int t, int D, int remove_mean) {
    const long long n = (long long)B * D;
    GRID_STRIDE(i, n) {
        const int d = (int)(i % D);
        const long long b = i / D;
        const float* wb = w + b * t * D + d;
        float* yb = y + b * t * D + d;
        float m = 0.f;
        if (remove_mean) {
            for (int j = 0; j < t; ++j) m += wb[(long long)j * D];
            m /= (float)t;
        }
        const float e = env ? env[i] : 0.f;
        for (int j = 0; j < t; ++j) yb[(long long)j * D] = e + wb[(long long)j * D] - m;
    }
}
extern "C" int gim_noise_combine(const float* env, const float* w, float* y, int B, int t, int D, int remove_mean, void* stream) {
    GIM_CHECK_ARG(w && y && B > 0 && t > 0 && D > 0, "noise_combine: bad args");
    hipLaunchKernelGGL(noise_combine_kernel, dim3(pw_blocks((long long)B * D)), dim3(256), 0, (hipStream_t)stream, env, w, y, B, t, D, remove_mean);
    return gim_check_launch("gim_noise_combine");
}

// ---------------------------------------------------------------- channel concat with broadcast
// y[r][0:Ca] = a[r][:],  y[r][Ca:Ca+Cb] = b[(r / rows_per_a_img / rep) ...]: a has R rows of Ca channels, b has
// R/rep "image blocks": row r of y takes b's row  (r / (P*rep)) * P + (r % P)  with P = pixels per image.
__global__ __launch_bounds__(256) void concat2_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                                                      long long R, int Ca, int Cb, int P, int rep) {
    const int Cy = Ca + Cb;
    const long long n = R * Cy;
    GRID_STRIDE(i, n) {
        const int c = (int)(i % Cy);
        const long long r = i / Cy;
        if (c < Ca) {
            y[i] = a[r * Ca + c];
        } else {
            const long long img = r / P;
            const int px = (int)(r - img * P);
            y[i] = b[((img / rep) * P + px) * Cb + (c - Ca)];
        }
    }
}
// da[r][c] = dy[r][c] for c < Ca
__global__ __launch_bounds__(256) void slice_channels_kernel(const float* __restrict__ dy, float* __restrict__ da, long long R, int Ca, int Cy) {
    const long long n = R * Ca;
    GRID_STRIDE(i, n) {
        const int c = (int)(i % Ca);
        const long long r = i / Ca;
        da[i] = dy[r * Cy + c];
    }
}
extern "C" int gim_concat2(const float* a, const float* b, float* y, int64_t R, int Ca, int Cb, int P, int rep, void* stream) {
    GIM_CHECK_ARG(a && b && y && R > 0 && Ca > 0 && Cb > 0 && P > 0 && rep > 0, "concat2: bad args");
    hipLaunchKernelGGL(concat2_kernel, dim3(pw_blocks(R * (Ca + Cb))), dim3(256), 0, (hipStream_t)stream, a, b, y, (long long)R, Ca, Cb, P, rep);
    return gim_check_launch("gim_concat2");
}
extern "C" int gim_slice_channels(const float* dy, float* da, int64_t R, int Ca, int Cy, void* stream) {
    GIM_CHECK_ARG(dy && da && R > 0 && Ca > 0 && Cy >= Ca, "slice_channels: bad args");
    hipLaunchKernelGGL(slice_channels_kernel, dim3(pw_blocks(R * Ca)), dim3(256), 0, (hipStream_t)stream, dy, da, (long long)R, Ca, Cy);
    return gim_check_launch("gim_slice_channels");
}

// ---------------------------------------------------------------- ImgAttention mix (models/model_blocks.py:598-608)
// s1 = sum_c q1*k1, s2 = sum_c q2*k2, (a1, a2) = softmax(s1, s2); out = x1*a1 + v2*a2.   One thread per pixel.
__global__ __launch_bounds__(256) void img_att_mix_fwd_kernel(const float* __restrict__ q1, const float* __restrict__ k1,
                                                              const float* __restrict__ q2, const float* __restrict__ k2,
                                                              const float* __restrict__ x1, const float* __restrict__ v2,
                                                              float* __restrict__ out, float* __restrict__ att, long long P, int C) {
    GRID_STRIDE(i, P) {
        const long long o = i * C;
        float s1 = 0.f, s2 = 0.f;
        for (int c = 0; c < C; ++c) {
            s1 += q1[o + c] * k1[o + c];
            s2 += q2[o + c] * k2[o + c];
        }
        const float a1 = 1.0f / (1.0f + expf(s2 - s1));
        att[i] = a1;
        for (int c = 0; c < C; ++c) out[o + c] = x1[o + c] * a1 + v2[o + c] * (1.0f - a1);
    }
}
__global__ __launch_bounds__(256) void img_att_mix_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ q1,
                                                              const float* __restrict__ k1, const float* __restrict__ q2,
                                                              const float* __restrict__ k2, const float* __restrict__ x1,
                                                              const float* __restrict__ v2, const float* __restrict__ att,
                                                              float* __restrict__ dq1, float* __restrict__ dk1, float* __restrict__ dq2,
                                                              float* __restrict__ dk2, float* __restrict__ dx1, float* __restrict__ dv2,
                                                              long long P, int C) {
    GRID_STRIDE(i, P) {
        const long long o = i * C;
        const float a1 = att[i], a2 = 1.0f - a1;
        float da1 = 0.f;
        for (int c = 0; c < C; ++c) {
            const float d = dout[o + c];
            dx1[o + c] = d * a1;
            dv2[o + c] = d * a2;
            da1 += d * (x1[o + c] - v2[o + c]);
        }
        const float ds1 = da1 * a1 * a2;
        for (int c = 0; c < C; ++c) {
            dq1[o + c] = ds1 * k1[o + c];
            dk1[o + c] = ds1 * q1[o + c];
            dq2[o + c] = -ds1 * k2[o + c];
            dk2[o + c] = -ds1 * q2[o + c];
        }
    }
}
extern "C" int gim_img_att_mix_fwd(const float* q1, const float* k1, const float* q2, const float* k2, const float* x1,
                                   const float* v2, float* out, float* att, int64_t P, int C, void* stream) {
    GIM_CHECK_ARG(q1 && k1 && q2 && k2 && x1 && v2 && out && att && P > 0 && C > 0, "img_att_mix_fwd: bad args");
    hipLaunchKernelGGL(img_att_mix_fwd_kernel, dim3(pw_blocks(P)), dim3(256), 0, (hipStream_t)stream, q1, k1, q2, k2, x1, v2, out, att,
                       (long long)P, C);
    return gim_check_launch("gim_img_att_mix_fwd");
}
extern "C" int gim_img_att_mix_bwd(const float* dout, const float* q1, const float* k1, const float* q2, const float* k2,
                                   const float* x1, const float* v2, const float* att, float* dq1, float* dk1, float* dq2, float* dk2,
                                   float* dx1, float* dv2, int64_t P, int C, void* stream) {
    GIM_CHECK_ARG(dout && q1 && k1 && q2 && k2 && x1 && v2 && att && dq1 && dk1 && dq2 && dk2 && dx1 && dv2 && P > 0 && C > 0,
                  "img_att_mix_bwd: bad args");
    hipLaunchKernelGGL(img_att_mix_bwd_kernel, dim3(pw_blocks(P)), dim3(256), 0, (hipStream_t)stream, dout, q1, k1, q2, k2, x1, v2, att,
                       dq1, dk1, dq2, dk2, dx1, dv2, (long long)P, C);
    return gim_check_launch("gim_img_att_mix_bwd");
}

// ---------------------------------------------------------------- second-order helpers (R1 regulariser)
// adjoint of maxpool_lrelu_bwd: g_dy[n][c] = g_dx[n][idx[n][c]][c] * lrelu'(y[n][c])
__global__ __launch_bounds__(256) void maxpool_gather_kernel(const float* __restrict__ gdx, const float* __restrict__ y,
                                                             const int32_t* __restrict__ idx, float* __restrict__ gdy, int N, int HW, int C,
                                                             float slope) {
    const long long n_nc = (long long)N * C;
    GRID_STRIDE(i, n_nc) {
        const int c = (int)(i % C);
        const long long n = i / C;
        gdy[i] = gdx[(n * HW + idx[i]) * C + c] * (y[i] > 0.f ? 1.0f : slope);
    }
}
extern "C" int gim_maxpool_gather(const float* gdx, const float* y, const int32_t* idx, float* gdy, int N, int HW, int C, float slope,
                                  void* stream) {
    GIM_CHECK_ARG(gdx && y && idx && gdy && N > 0 && HW > 0 && C > 0, "maxpool_gather: bad args");
    hipLaunchKernelGGL(maxpool_gather_kernel, dim3(pw_blocks((long long)N * C)), dim3(256), 0, (hipStream_t)stream, gdx, y, idx, gdy, N, HW, C,
                       slope);
    return gim_check_launch("gim_maxpool_gather");
}

// derivative of the softmax backward  dS = P * (dP - c),  c_j = sum_i P_ij dP_ij  w.r.t. P, contracted with gS:
//   gP_ij = gS_ij * (dP_ij - c_j) - dP_ij * d_j,   d_j = sum_i gS_ij P_ij        (columns j of [B][R][Cc])
__global__ __launch_bounds__(256) void softmax_dim1_bwd_dp_kernel(const float* __restrict__ gs, const float* __restrict__ dp,
                                                                  const float* __restrict__ p, float* __restrict__ gp, int R, int Cc) {
    __shared__ float red[2][4][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    const bool ok = col < Cc;
    const long long base = (long long)b * R * Cc + col;
    float c = 0.f, d = 0.f;
    if (ok) for (int r = rg; r < R; r += 4) {
        const long long o = base + (long long)r * Cc;
        c += p[o] * dp[o];
        d += gs[o] * p[o];
    }
    red[0][rg][cl] = c;
    red[1][rg][cl] = d;
    __syncthreads();
    c = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
    d = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
    if (!ok) return;
    for (int r = rg; r < R; r += 4) {
        const long long o = base + (long long)r * Cc;
        gp[o] = gs[o] * (dp[o] - c) - dp[o] * d;
    }
}
extern "C" int gim_softmax_dim1_bwd_dp(const float* gs, const float* dp, const float* p, float* gp, int B, int R, int Ccols, void* stream) {
    GIM_CHECK_ARG(gs && dp && p && gp && B > 0 && R > 0 && Ccols > 0, "softmax_dim1_bwd_dp: bad args");
    hipLaunchKernelGGL(softmax_dim1_bwd_dp_kernel, dim3((Ccols + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, gs, dp, p, gp, R, Ccols);
    return gim_check_launch("gim_softmax_dim1_bwd_dp");
}

// second-order set statistics.  First-order backward: dx_j = dm/t + ds * (x_j - m) / ((t-1) sd).
// Given g (same shape as dx): adjoint w.r.t. (dm, ds):  g_dm = sum_j g_j / t,  g_ds = sum_j g_j (x_j - m) / ((t-1) sd);
// w.r.t. x: gx_k = ds/((t-1) sd) * [ g_k - mean(g) - (x_k - m) * sum_j g_j (x_j - m) / ((t-1) sd^2) ]
__global__ __launch_bounds__(256) void set_stats_bwd_bwd_kernel(const float* __restrict__ x, const float* __restrict__ ds, const float* __restrict__ g,
                                                                float* __restrict__ g_dm, float* __restrict__ g_ds, float* __restrict__ gx,
                                                                int B, int t, int D, long long ld_in, long long ld_out) {
    const long long n = (long long)B * D;
    GRID_STRIDE(i, n) {
        const int d = (int)(i % D);
        const long long b = i / D;
        const float* xb = x + b * t * D + d;
        const float* gb = g + b * t * D + d;
        float m = 0.f, gsum = 0.f;
        for (int j = 0; j < t; ++j) { m += xb[(long long)j * D]; gsum += gb[(long long)j * D]; }
        m /= (float)t;
        if (g_dm) g_dm[b * ld_out + d] = gsum / (float)t;
        float sdv = 0.f, gdot = 0.f, coef = 0.f;
        if (t > 1) {
            float ss = 0.f;
            for (int j = 0; j < t; ++j) { const float dd = xb[(long long)j * D] - m; ss += dd * dd; gdot += gb[(long long)j * D] * dd; }
            sdv = sqrtf(ss / (float)(t - 1) + 1e-8f);
            coef = 1.0f / ((float)(t - 1) * sdv);
        }
        if (g_ds) g_ds[b * ld_out + d] = gdot * coef;
        if (gx) {
            float* ob = gx + b * t * D + d;
            const float dsv = (ds && t > 1) ? ds[b * ld_in + d] : 0.f;
            const float k2 = (t > 1) ? gdot * coef / sdv : 0.f;  // sum_j g_j (x_j - m) / ((t-1) sd^2)
            for (int j = 0; j < t; ++j)
                ob[(long long)j * D] = dsv * coef * (gb[(long long)j * D] - gsum / (float)t - (xb[(long long)j * D] - m) * k2);
        }
    }
}
extern "C" int gim_set_stats_bwd_bwd(const float* x, const float* dstd, const float* g, float* g_dmean, float* g_dstd, float* gx, int B,
                                     int t, int D, int64_t ld_dstd, int64_t ld_out, void* stream) {
    GIM_CHECK_ARG(x && g && B > 0 && t > 0 && D > 0, "set_stats_bwd_bwd: bad args");
    hipLaunchKernelGGL(set_stats_bwd_bwd_kernel, dim3(pw_blocks((long long)B * D)), dim3(256), 0, (hipStream_t)stream, x, dstd, g, g_dmean,
                       g_dstd, gx, B, t, D, (long long)ld_dstd, (long long)ld_out);
    return gim_check_launch("gim_set_stats_bwd_bwd");
}

// per-episode squared norm: out[b] = sum_i x[b][i]^2 (training/utils.py:122-123), and its gradient 2 * x * dout[b]
__global__ __launch_bounds__(256) void sqsum_rows_kernel(const float* __restrict__ x, float* __restrict__ out, long long L) {
    __shared__ float red[4];
    const float* xb = x + (long long)blockIdx.x * L;
    float s = 0.f;
    for (long long i = threadIdx.x; i < L; i += 256) s += xb[i] * xb[i];
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sqsum_rows_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dout, float* __restrict__ dx,
                                                             long long B, long long L) {
    const long long n = B * L;
    GRID_STRIDE(i, n) dx[i] = 2.0f * x[i] * dout[i / L];
}
extern "C" int gim_sqsum_rows_fwd(const float* x, float* out, int B, int64_t L, void* stream) {
    GIM_CHECK_ARG(x && out && B > 0 && L > 0, "sqsum_rows_fwd: bad args");
    hipLaunchKernelGGL(sqsum_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, out, (long long)L);
    return gim_check_launch("gim_sqsum_rows_fwd");
}
extern "C" int gim_sqsum_rows_bwd(const float* x, const float* dout, float* dx, int B, int64_t L, void* stream) {
    GIM_CHECK_ARG(x && dout && dx && B > 0 && L > 0, "sqsum_rows_bwd: bad args");
    hipLaunchKernelGGL(sqsum_rows_bwd_kernel, dim3(pw_blocks((long long)B * L)), dim3(256), 0, (hipStream_t)stream, x, dout, dx, (long long)B,
                       (long long)L);
    return gim_check_launch("gim_sqsum_rows_bwd");
}

// out = g * lrelu'(x)  (the activation mask of a fused-prologue conv, applied to a second-order cotangent)
__global__ __launch_bounds__(256) void lrelu_mask_mul_kernel(const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ out,
                                                             long long n, float slope) {
    GRID_STRIDE(i, n) out[i] = g[i] * (x[i] > 0.f ? 1.0f : slope);
}
extern "C" int gim_lrelu_mask_mul(const float* g, const float* x, float slope, float* out, int64_t n, void* stream) {
    GIM_CHECK_ARG(g && x && out && n > 0, "lrelu_mask_mul: bad args");
    hipLaunchKernelGGL(lrelu_mask_mul_kernel, dim3(pw_blocks((long long)n)), dim3(256), 0, (hipStream_t)stream, g, x, out, (long long)n, slope);
    return gim_check_launch("gim_lrelu_mask_mul");
}

// ---------------------------------------------------------------- input pipeline
// Episode batching on the GPU (data_handling/img_datasets.py:68-103,296-303 restated for a resident image bank):
// out[i][c][y][x] = (bank[idx[i]][y][flip[i] ? W-1-x : x][c] / 255) * 2 - 1     (ToTensor, dynamic range (0,1) -> (-1,1),
// RandomHorizontalFlip decided by the caller).  bank is uint8 NHWC [n_img][H][W][C]; out is float NCHW [n_out][C][H][W].
__global__ __launch_bounds__(256) void episode_gather_kernel(const uint8_t* __restrict__ bank, const int32_t* __restrict__ idx,
                                                             const uint8_t* __restrict__ flip, float* __restrict__ out, int H, int W,
                                                             int C, long long n_total) {
    const long long hw = (long long)H * W, chw = hw * C;
    GRID_STRIDE(o, n_total) {
        const long long i = o / chw;
        long long r = o - i * chw;
        const int c = (int)(r / hw);
        r -= (long long)c * hw;
        const int y = (int)(r / W), x = (int)(r - (long long)y * W);
        const int xs = flip[i] ? W - 1 - x : x;
        const float u = (float)bank[(((long long)idx[i] * H + y) * W + xs) * C + c];
        out[o] = (u / 255.0f) * 2.0f + (-1.0f);
    }
}
extern "C" int gim_episode_gather(const uint8_t* bank, const int32_t* idx, const uint8_t* flip, float* out, int n_out, int H, int W,
                                  int C, void* stream) {
    GIM_CHECK_ARG(bank && idx && flip && out && n_out > 0 && H > 0 && W > 0 && C > 0, "episode_gather: bad args");
    const long long n = (long long)n_out * H * W * C;
    hipLaunchKernelGGL(episode_gather_kernel, dim3(pw_blocks(n)), dim3(256), 0, (hipStream_t)stream, bank, idx, flip, out, H, W, C, n);
    return gim_check_launch("gim_episode_gather");
}

// ---------------------------------------------------------------- small-gradient reductions that ADD into the flat gradient bucket
// out_a[c] (+)= sum_r a[r][c],  out_b[c] (+)= sum_r b[r][c]   for short row counts (InstanceNorm affine gradients: rows = images):
// one launch instead of two two-stage column sums plus two AccumulateGrad adds.
__global__ __launch_bounds__(256) void colsum2_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out_a,
                                                      float* __restrict__ out_b, int rows, int C, int accumulate) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    const float sa = colsum_rows_64x4(a, rows, C, c, rg, red, cl);
    const float sb = colsum_rows_64x4(b, rows, C, c, rg, red, cl);
    if (rg != 0 || c >= C) return;
    out_a[c] = accumulate ? out_a[c] + sa : sa;
    out_b[c] = accumulate ? out_b[c] + sb : sb;
}
extern "C" int gim_colsum2(const float* a, const float* b, float* out_a, float* out_b, int rows, int C, int accumulate, void* stream) {
    GIM_CHECK_ARG(a && b && out_a && out_b && rows > 0 && C > 0, "colsum2: bad args");
    hipLaunchKernelGGL(colsum2_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, a, b, out_a, out_b, rows, C, accumulate);
    return gim_check_launch("gim_colsum2");
}

__global__ __launch_bounds__(256) void colsum_final_acc_kernel(const float* __restrict__ part, float* __restrict__ out, int P, int C) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    const float s = colsum_rows_64x4(part, P, C, c, rg, red, cl);
    if (rg == 0 && c < C) out[c] += s;
}
// out[c] += sum_r x[r][c]  (gim_colsum that ADDS: bias gradients straight into the optimizer's gradient bucket)
extern "C" int gim_colsum_acc(const float* x, float* out, float* scratch, int64_t rows, int C, void* stream) {
    GIM_CHECK_ARG(x && out && scratch && rows > 0 && C > 0, "colsum_acc: bad args");
    long long P = (rows + 63) / 64;
    if (P > 256) P = 256;
    const long long rows_per = (rows + P - 1) / P;
    P = (rows + rows_per - 1) / rows_per;
    hipLaunchKernelGGL(colsum_part_kernel, dim3((C + 63) / 64, (int)P), dim3(256), 0, (hipStream_t)stream, x, scratch, (long long)rows, C, rows_per);
    hipLaunchKernelGGL(colsum_final_acc_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, scratch, out, (int)P, C);
    return gim_check_launch("gim_colsum_acc");
}

// ---------------------------------------------------------------- depth to space
// y[n][2 h + py][2 w + px][c] = lrelu(y4[n][h][w][(2 py + px) * C + c] + bias[c], post_slope): un-stacks the four output-parity classes of the merged
// sub-pixel convolution (gim_conv2d_pack_subpixel_weights in conv_igemm.hip)
__global__ __launch_bounds__(256) void depth_to_space2_kernel(const float* __restrict__ y4, const float* __restrict__ bias, float* __restrict__ y,
                                                              int N, int Hs, int Ws, int C, float post_slope) {
    const long long total = (long long)N * Hs * Ws * 4 * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int cls = (int)(r & 3); r >>= 2;
        const int w = (int)(r % Ws); r /= Ws;
        const int h = (int)(r % Hs);
        const long long n = r / Hs;
        const float v = y4[i] + (bias ? bias[c] : 0.f);
        y[((n * 2 * Hs + 2 * h + (cls >> 1)) * 2 * Ws + 2 * w + (cls & 1)) * C + c] = fmaxf(v, v * post_slope);
    }
}
extern "C" int gim_depth_to_space2(const float* y4, const float* bias, float* y, int N, int Hs, int Ws, int C, float post_slope, void* stream) {
    GIM_CHECK_ARG(y4 && y && N > 0 && Hs > 0 && Ws > 0 && C > 0 && post_slope > 0.f && post_slope <= 1.f, "depth_to_space2: bad args");
    const long long total = (long long)N * Hs * Ws * 4 * C;
    hipLaunchKernelGGL(depth_to_space2_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)stream,
                       y4, bias, y, N, Hs, Ws, C, post_slope);
    return gim_check_launch("gim_depth_to_space2");
}

// ---------------------------------------------------------------- stream self-check
// One wave that keeps a compute unit busy for `usec` microseconds of the constant 100 MHz wall clock (s_memrealtime), capped at
// 5 ms: every wave reaches the exit.  ops.stream_concurrency_check launches one per engine stream at the same time: streams that
// HIP dealt onto ONE hardware queue run their kernels one after the other (GPU_MAX_HW_QUEUES, profiles/r03_q_hw_queue_sweep.txt),
// which the event-timed total shows.
__global__ __launch_bounds__(64) void spin_kernel(unsigned ticks, unsigned* sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned n = 0;
    while (wall_clock64() - t0 < ticks) ++n;
    if (sink && threadIdx.x == 0) *sink = n;
}
extern "C" int gim_spin(int usec, void* stream) {
    GIM_CHECK_ARG(usec > 0 && usec <= 5000, "spin: 1 .. 5000 microseconds");
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned)usec * 100u, (unsigned*)nullptr);
    return gim_check_launch("gim_spin");
}

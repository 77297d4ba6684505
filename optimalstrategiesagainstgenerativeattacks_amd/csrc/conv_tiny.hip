// Direct convolution for image-to-image layers with <= 3 channels on BOTH sides: the last conv of the generator's two decoders
// (models/model_blocks.py:749-750,839-840 instantiated 3 -> 3 by models/gim_img_models.py:128 and :187-193: a 3x3 and a 9x9 'same'
// convolution on the full-resolution image; 1 -> 1 for one-channel data).  729 multiply-adds per pixel at most - 0.5 GFLOP per
// launch - for which the implicit-GEMM kernels pay a 16-column MFMA tile with 3 useful columns and a scalar gather per K element:
// 0.10 ms per launch (5 TFLOP/s), forward, dgrad and wgrad each.  Here: one 16 x 16 output tile of one image per workgroup, the input
// tile with its halo in LDS, weights through the scalar cache (wave-uniform addresses), plain vector FMAs.
//   forward : y = conv(lrelu(x), w) / sigma + bias + residual (full or half resolution), optional LeakyReLU on the stored output
//   dgrad   : dx = lrelu'(x) * conv(dy, flipped / transposed w) / sigma
//   wgrad   : dW[co][ta][tb][ci] += sum_pixels dy[p][co] * lrelu(x)[p + tap][ci], db[co] += sum_pixels dy[p][co]  (float atomics
//             into the caller's pre-zeroed slot, as every other pixel slice of a weight gradient)
#include "common.h"

struct TinyP {
    const float* in;      // gathered tensor [N, H, W, CG]
    const float* w;       // [CO][K][K][CI] (the conv's weights, whatever the direction)
    const float* bias;
    const float* sigma;
    const float* res;     // forward: residual; dgrad: mask_x (the conv's input: sign decides the LeakyReLU derivative)
    float* out;
    float* out2;          // wgrad: bias gradient slot (may be null)
    int N, H, W;
    float pre_slope;      // forward / wgrad: LeakyReLU on the gathered input (1 = none); dgrad: slope of the mask
    float post_slope;
    int res_ups;
};

constexpr int TT = 16;   // output tile edge

// FLIP = false: forward (gathered channels = CI of w, produced = CO); true: dgrad (gathered = CO, produced = CI, taps flipped)
template <int K, int CI, int CO, bool FLIP>
__global__ __launch_bounds__(256) void conv_tiny_kernel(const TinyP p) {
    constexpr int PAD = (K - 1) / 2, TW = TT + K - 1;
    constexpr int CG = FLIP ? CO : CI, CP = FLIP ? CI : CO;   // gathered / produced channels
    __shared__ float xs[TW * TW * CG];
    const int t = threadIdx.x;
    const int tiles_x = (p.W + TT - 1) / TT;
    const int oy0 = (blockIdx.x / tiles_x) * TT, ox0 = (blockIdx.x % tiles_x) * TT, n = blockIdx.y;
    const float* img = p.in + (long long)n * p.H * p.W * CG;
    for (int idx = t; idx < TW * TW * CG; idx += 256) {
        const int c = idx % CG, rc = idx / CG, col = rc % TW, row = rc / TW;
        const int iy = oy0 + row - PAD, ix = ox0 + col - PAD;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) v = img[((long long)iy * p.W + ix) * CG + c];
        if (!FLIP) v = fmaxf(v, v * p.pre_slope);   // slope 1 = identity (0 < slope <= 1)
        xs[idx] = v;
    }
    __syncthreads();
    const int ty = t >> 4, tx = t & 15;
    float acc[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) acc[c] = 0.f;
    for (int ta = 0; ta < K; ++ta) {      // (wave-uniform: the weight addresses below are scalar loads)
#pragma unroll
        for (int tb = 0; tb < K; ++tb) {
            const int wa = FLIP ? K - 1 - ta : ta, wb = FLIP ? K - 1 - tb : tb;
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
                const float xv = xs[((ty + ta) * TW + tx + tb) * CG + cg];
#pragma unroll
                for (int cp = 0; cp < CP; ++cp) {
                    const int co = FLIP ? cg : cp, ci = FLIP ? cp : cg;
                    acc[cp] += xv * p.w[((co * K + wa) * K + wb) * CI + ci];
                }
            }
        }
    }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= p.H || ox >= p.W) return;
    const float scale = p.sigma ? 1.0f / p.sigma[0] : 1.0f;
    const long long o = (((long long)n * p.H + oy) * p.W + ox) * CP;
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        float v = acc[c] * scale;
        if (FLIP) {
            if (p.res) v *= (p.res[o + c] > 0.f ? 1.0f : p.pre_slope);
        } else {
            if (p.bias) v += p.bias[c];
            if (p.res) v += p.res_ups ? p.res[(((long long)n * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1)) * CP + c] : p.res[o + c];
            v = fmaxf(v, v * p.post_slope);
        }
        p.out[o + c] = v;
    }
}

// one thread per (tap, input channel) [x pixel group]: it walks the tile's pixels, keeps CO sums, adds them to the slot at the end
template <int K, int CI, int CO>
__global__ __launch_bounds__(256) void conv_tiny_wgrad_kernel(const TinyP p) {   // in = x, res = dy, out = dW slot, out2 = db slot
    constexpr int PAD = (K - 1) / 2, TW = TT + K - 1, NT = K * K * CI, G = 256 / NT;
    static_assert(G >= 1, "one thread per (tap, channel)");
    __shared__ float xs[TW * TW * CI];
    __shared__ float dys[TT * TT * CO];
    const int t = threadIdx.x;
    const int tiles_x = (p.W + TT - 1) / TT;
    const int oy0 = (blockIdx.x / tiles_x) * TT, ox0 = (blockIdx.x % tiles_x) * TT, n = blockIdx.y;
    const float* img = p.in + (long long)n * p.H * p.W * CI;
    const float* dyi = p.res + (long long)n * p.H * p.W * CO;
    for (int idx = t; idx < TW * TW * CI; idx += 256) {
        const int c = idx % CI, rc = idx / CI, col = rc % TW, row = rc / TW;
        const int iy = oy0 + row - PAD, ix = ox0 + col - PAD;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) v = img[((long long)iy * p.W + ix) * CI + c];
        xs[idx] = fmaxf(v, v * p.pre_slope);
    }
    for (int idx = t; idx < TT * TT * CO; idx += 256) {
        const int c = idx % CO, pp = idx / CO, oy = oy0 + (pp >> 4), ox = ox0 + (pp & 15);
        dys[idx] = (oy < p.H && ox < p.W) ? dyi[((long long)oy * p.W + ox) * CO + c] : 0.f;
    }
    __syncthreads();
    // the G pixel groups of one (tap, channel) are summed through LDS first: ONE atomic per weight-gradient element and workgroup
    // (with G = 9 groups a 3x3 layer sent 9 x 1280 adds to each of its 81 addresses: same-address atomics serialize - 0.70 ms)
    __shared__ float red[256 * CO];
    if (t < G * NT) {
        const int grp = t / NT, id = t - grp * NT;
        const int ci = id % CI, tap = id / CI, tb = tap % K, ta = tap / K;
        float acc[CO];
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[c] = 0.f;
        for (int pp = grp; pp < TT * TT; pp += G) {
            const float xv = xs[(((pp >> 4) + ta) * TW + (pp & 15) + tb) * CI + ci];
#pragma unroll
            for (int c = 0; c < CO; ++c) acc[c] += dys[pp * CO + c] * xv;
        }
#pragma unroll
        for (int c = 0; c < CO; ++c) red[(grp * NT + id) * CO + c] = acc[c];
    }
    __syncthreads();
    for (int o = t; o < NT * CO; o += 256) {     // o = id * CO + c
        const int id = o / CO, c = o - id * CO;
        float v = 0.f;
        for (int gq = 0; gq < G; ++gq) v += red[(gq * NT + id) * CO + c];
        const int ci = id % CI, tap = id / CI, tb = tap % K, ta = tap / K;
        atomicAdd(&p.out[((c * K + ta) * K + tb) * CI + ci], v);
    }
    if (p.out2 && t >= 256 - CO) {     // the last CO threads (idle above whenever G * NT < 256 - always true for these shapes' bias lanes or cheap otherwise)
        const int c = t - (256 - CO);
        float s = 0.f;
        for (int pp = 0; pp < TT * TT; ++pp) s += dys[pp * CO + c];
        atomicAdd(&p.out2[c], s);
    }
}

// ---------------------------------------------------------------------------------------------------------------- host
bool gim_tiny_shape(const gim_conv_shape* s) {
    if (s->N > 65535) return false;
    return !s->ups && !s->pool && !s->wfold && s->Cin == s->Cout && (s->Cin == 3 || s->Cin == 1) && (s->KH == 3 || s->KH == 9);
}

template <bool FLIP>
static void tiny_launch(const TinyP& p, const gim_conv_shape* s, hipStream_t st) {
    const dim3 g(((s->H + TT - 1) / TT) * ((s->W + TT - 1) / TT), s->N);
    if (s->KH == 3 && s->Cin == 3) hipLaunchKernelGGL((conv_tiny_kernel<3, 3, 3, FLIP>), g, dim3(256), 0, st, p);
    else if (s->KH == 9 && s->Cin == 3) hipLaunchKernelGGL((conv_tiny_kernel<9, 3, 3, FLIP>), g, dim3(256), 0, st, p);
    else if (s->KH == 3) hipLaunchKernelGGL((conv_tiny_kernel<3, 1, 1, FLIP>), g, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_tiny_kernel<9, 1, 1, FLIP>), g, dim3(256), 0, st, p);
}

bool gim_tiny_fwd(const float* x, const float* w, const float* bias, const float* sigma, const float* res, float* y,
                  const gim_conv_shape* s, hipStream_t st, int32_t* plan_out) {
    if (!gim_tiny_shape(s) || s->N > 65535) return false;
    if (plan_out) {
        const int32_t v[8] = {0, TT * TT, s->Cout, 1, ((s->H + TT - 1) / TT) * ((s->W + TT - 1) / TT), s->N, 1, 3};
        for (int i = 0; i < 8; ++i) plan_out[i] = v[i];
        return true;
    }
    TinyP p{};
    p.in = x; p.w = w; p.bias = bias; p.sigma = sigma; p.res = res; p.out = y;
    p.N = s->N; p.H = s->H; p.W = s->W; p.pre_slope = s->pre_slope; p.post_slope = s->post_slope > 0.f ? s->post_slope : 1.f; p.res_ups = s->res_ups;
    tiny_launch<false>(p, s, st);
    return true;
}

bool gim_tiny_dgrad(const float* dy, const float* w, const float* sigma, const float* mask_x, float* dx, const gim_conv_shape* s,
                    hipStream_t st, int32_t* plan_out) {
    if (!gim_tiny_shape(s) || s->N > 65535) return false;
    if (plan_out) {
        const int32_t v[8] = {0, TT * TT, s->Cin, 1, ((s->H + TT - 1) / TT) * ((s->W + TT - 1) / TT), s->N, 1, 3};
        for (int i = 0; i < 8; ++i) plan_out[i] = v[i];
        return true;
    }
    TinyP p{};
    p.in = dy; p.w = w; p.sigma = sigma; p.res = mask_x; p.out = dx;
    p.N = s->N; p.H = s->H; p.W = s->W; p.pre_slope = s->pre_slope; p.post_slope = 1.f;
    tiny_launch<true>(p, s, st);
    return true;
}

bool gim_tiny_wgrad_acc(const float* dy, const float* x, float* acc, float* bias_acc, const gim_conv_shape* s, hipStream_t st,
                        int32_t* plan_out) {
    if (!gim_tiny_shape(s) || s->N > 65535) return false;
    const dim3 g(((s->H + TT - 1) / TT) * ((s->W + TT - 1) / TT), s->N);
    if (plan_out) {
        const int32_t v[8] = {0, 0, 0, (int32_t)(g.x * g.y), (int32_t)g.x, (int32_t)g.y, 1, 3};
        for (int i = 0; i < 8; ++i) plan_out[i] = v[i];
        return true;
    }
    TinyP p{};
    p.in = x; p.res = dy; p.out = acc; p.out2 = bias_acc;
    p.N = s->N; p.H = s->H; p.W = s->W; p.pre_slope = s->pre_slope;
    if (s->KH == 3 && s->Cin == 3) hipLaunchKernelGGL((conv_tiny_wgrad_kernel<3, 3, 3>), g, dim3(256), 0, st, p);
    else if (s->KH == 9 && s->Cin == 3) hipLaunchKernelGGL((conv_tiny_wgrad_kernel<9, 3, 3>), g, dim3(256), 0, st, p);
    else if (s->KH == 3) hipLaunchKernelGGL((conv_tiny_wgrad_kernel<3, 1, 1>), g, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_tiny_wgrad_kernel<9, 1, 1>), g, dim3(256), 0, st, p);
    return true;
}

// Spectral-norm power iteration and the weight-gradient finish (HBM-bound GEMV-shaped work).
//
// Weight W is [Cout][T][Cin] in memory (T = KH*KH taps); torch's weight_mat is [Cout][Cin*T] with
// column q = ci*T + tap.  u is [Cout]; v is [Cin*T] in torch's order, so the kernels translate
// physical column p = tap*Cin + ci  <->  q = ci*T + tap.
#include "common.h"

#define SN_R 8  // row chunks of the W^T u partial sums

// part[r][p] = sum over the r-th chunk of rows co of W[co][p] * u[co]
__global__ __launch_bounds__(256) void sn_colpart_kernel(const float* __restrict__ w, const float* __restrict__ u,
                                                         float* __restrict__ part, int Cout, int K, int rows_per) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    const int c0 = r * rows_per, c1 = min(Cout, c0 + rows_per);
    if (p >= K) return;
    float acc = 0.f;
    for (int co = c0; co < c1; ++co) acc += w[(long long)co * K + p] * u[co];
    part[(long long)r * K + p] = acc;
}

// single block.  training: a = sum_r part[r]; v = a / max(|a|, eps) ; else v given.
// writes v_phys (physical order, scratch), v (torch order, in place when training) and v_out (copy).
__global__ __launch_bounds__(256) void sn_vnorm_kernel(const float* __restrict__ part, int R, float* __restrict__ v,
                                                       float* __restrict__ v_phys, float* __restrict__ v_out, int K, int Cin,
                                                       int T, int training) {
    __shared__ float red[4];
    if (training) {
        float ss = 0.f;
        for (int p = threadIdx.x; p < K; p += 256) {
            float a = 0.f;
            for (int r = 0; r < R; ++r) a += part[(long long)r * K + p];
            v_phys[p] = a;
            ss += a * a;
        }
        const float nrm = sqrtf(block_sum_256(ss, red));
        const float inv = 1.0f / fmaxf(nrm, 1e-12f);
        for (int p = threadIdx.x; p < K; p += 256) {
            const float val = v_phys[p] * inv;
            const int tap = p / Cin, ci = p - tap * Cin;
            const int q = ci * T + tap;
            v_phys[p] = val;
            v[q] = val;
            v_out[q] = val;
        }
    } else {
        for (int q = threadIdx.x; q < K; q += 256) {
            const int ci = q / T, tap = q - ci * T;
            const float val = v[q];
            v_phys[tap * Cin + ci] = val;
            v_out[q] = val;
        }
    }
}

// t[co] = sum_p W[co][p] * v_phys[p]; one wave per row
__global__ __launch_bounds__(256) void sn_rowdot_kernel(const float* __restrict__ w, const float* __restrict__ v_phys,
                                                        float* __restrict__ tvec, int Cout, int K) {
    const int co = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (co >= Cout) return;
    float acc = 0.f;
    const float* row = w + (long long)co * K;
    for (int p = lane; p < K; p += 64) acc += row[p] * v_phys[p];
    acc = wave_sum(acc);
    if (lane == 0) tvec[co] = acc;
}

// single block. training: u = t / max(|t|, eps); sigma = u . t ; else sigma = u . t with the stored u.
__global__ __launch_bounds__(256) void sn_final_kernel(const float* __restrict__ tvec, float* __restrict__ u,
                                                       float* __restrict__ u_out, float* __restrict__ sigma, int Cout,
                                                       int training) {
    __shared__ float red[4];
    float inv = 0.f;
    if (training) {
        float ss = 0.f;
        for (int c = threadIdx.x; c < Cout; c += 256) ss += tvec[c] * tvec[c];
        inv = 1.0f / fmaxf(sqrtf(block_sum_256(ss, red)), 1e-12f);
    }
    float dot = 0.f;
    for (int c = threadIdx.x; c < Cout; c += 256) {
        float uv;
        if (training) {
            uv = tvec[c] * inv;
            u[c] = uv;
        } else {
            uv = u[c];
        }
        u_out[c] = uv;
        dot += uv * tvec[c];
    }
    dot = block_sum_256(dot, red);
    if (threadIdx.x == 0) sigma[0] = dot;
}

extern "C" int gim_spectral_sigma(const float* w, float* u, float* v, float* sigma, float* u_out, float* v_out, float* scratch,
                                  int Cout, int Cin, int KH, int training, void* stream) {
    GIM_CHECK_ARG(w && u && v && sigma && u_out && v_out && scratch, "spectral_sigma: null pointer");
    GIM_CHECK_ARG(Cout > 0 && Cin > 0 && KH > 0, "spectral_sigma: bad dims");
    hipStream_t st = (hipStream_t)stream;
    const int T = KH * KH, K = Cin * T;
    float* part = scratch;                 // [SN_R][K]
    float* v_phys = scratch + SN_R * K;    // [K]
    float* tvec = v_phys + K;              // [Cout]
    int R = 1;
    if (training) {
        R = min(SN_R, (Cout + 63) / 64);
        const int rows_per = (Cout + R - 1) / R;
        hipLaunchKernelGGL(sn_colpart_kernel, dim3((K + 255) / 256, R), dim3(256), 0, st, w, u, part, Cout, K, rows_per);
    }
    hipLaunchKernelGGL(sn_vnorm_kernel, dim3(1), dim3(256), 0, st, part, R, v, v_phys, v_out, K, Cin, T, training);
    hipLaunchKernelGGL(sn_rowdot_kernel, dim3((Cout + 3) / 4), dim3(256), 0, st, w, v_phys, tvec, Cout, K);
    hipLaunchKernelGGL(sn_final_kernel, dim3(1), dim3(256), 0, st, tvec, u, u_out, sigma, Cout, training);
    return gim_check_launch("gim_spectral_sigma");
}

// -------------------------------------------------------------------------------------------------
// weight-gradient finish
// -------------------------------------------------------------------------------------------------
#define WF_BLOCKS 512

// dw[i] = sum_s slabs[s][i] (+ <dw, w> partials, + bias sums).  64 elements x 4 slab groups per block pass: the
// slab loop of one element is spread over 4 threads and unrolled, so a 100-slab reduction keeps ~32 independent
// loads in flight per element instead of one dependent chain.
__global__ __launch_bounds__(256) void wgrad_sum_kernel(const float* __restrict__ slabs, int n_slabs, long long n,
                                                        const float* __restrict__ w, float* __restrict__ dw,
                                                        float* __restrict__ partial, const float* __restrict__ bias_slabs,
                                                        float* __restrict__ db, int Cout, int acc_dw, int acc_db) {
    __shared__ float red[4];
    __shared__ float part[4][64];
    float dot = 0.f;
    if (db) {
        for (int c = blockIdx.x * 256 + threadIdx.x; c < Cout; c += gridDim.x * 256) {
            float g = acc_db ? db[c] : 0.f;
            for (int s = 0; s < n_slabs; ++s) g += bias_slabs[(long long)s * Cout + c];
            db[c] = g;
        }
    }
    const int el = threadIdx.x & 63, sg = threadIdx.x >> 6;
    for (long long i0 = (long long)blockIdx.x * 64; i0 < n; i0 += (long long)gridDim.x * 64) {
        const long long i = i0 + el;
        float g = 0.f;
        if (i < n) {
            int s = sg;
#pragma unroll 1
            for (; s + 12 < n_slabs; s += 16) {
                const float g0 = slabs[(long long)s * n + i], g1 = slabs[(long long)(s + 4) * n + i];
                const float g2 = slabs[(long long)(s + 8) * n + i], g3 = slabs[(long long)(s + 12) * n + i];
                g += (g0 + g1) + (g2 + g3);
            }
            for (; s < n_slabs; s += 4) g += slabs[(long long)s * n + i];
        }
        part[sg][el] = g;
        __syncthreads();
        if (sg == 0 && i < n) {
            g = (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
            dw[i] = acc_dw ? dw[i] + g : g;
            if (w) dot += g * w[i];
        }
        __syncthreads();
    }
    if (partial) {
        dot = block_sum_256(dot, red);
        if (threadIdx.x == 0) partial[blockIdx.x] = dot;
    }
}

// fold 1: src = dF [Cout][KF][KF][Cin]:  dW[co][kh][kw][ci] = 0.25 * sum_{dh,dw} dF[co][kh+dh][kw+dw][ci]
// fold 2: src = G  [Cin][KF][KF][Cout] with dF[co][a][b][ci] = G[ci][K-a][K-b][co]:  dW = sum_{dh,dw} dF[..][kh+dh][kw+dw][..]
__global__ __launch_bounds__(256) void wgrad_unfold_kernel(const float* __restrict__ src, const float* __restrict__ w,
                                                           float* __restrict__ dw, float* __restrict__ partial, int Cout, int Cin,
                                                           int K, int fold) {
    __shared__ float red[4];
    const int KF = K + 1;
    const long long n = (long long)Cout * K * K * Cin;
    float dot = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int ci = (int)(i % Cin);
        long long rr = i / Cin;
        const int kw = (int)(rr % K); rr /= K;
        const int kh = (int)(rr % K);
        const int co = (int)(rr / K);
        float g = 0.f;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int dwd = 0; dwd < 2; ++dwd) {
                const int a = kh + dh, b = kw + dwd;
                if (fold == 1) g += src[(((long long)co * KF + a) * KF + b) * Cin + ci];
                else g += src[(((long long)ci * KF + (K - a)) * KF + (K - b)) * Cout + co];
            }
        if (fold == 1) g *= 0.25f;
        dw[i] = g;
        if (w) dot += g * w[i];
    }
    if (partial) {
        dot = block_sum_256(dot, red);
        if (threadIdx.x == 0) partial[blockIdx.x] = dot;
    }
}

__global__ __launch_bounds__(256) void wgrad_sn_apply_kernel(const float* __restrict__ g, float* __restrict__ dw, int accumulate,
                                                             const float* __restrict__ partial, int n_part,
                                                             const float* __restrict__ sigma, const float* __restrict__ u,
                                                             const float* __restrict__ v, long long n, int Cin, int T) {
    __shared__ float red[4];
    float d = 0.f;
    for (int i = threadIdx.x; i < n_part; i += 256) d += partial[i];
    d = block_sum_256(d, red);
    const float s = sigma[0];
    const float inv = 1.0f / s;
    const float coef = d * inv * inv;
    const int K = Cin * T;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int co = (int)(i / K);
        const int p = (int)(i - (long long)co * K);
        const int tap = p / Cin, ci = p - tap * Cin;
        const float val = g[i] * inv - coef * u[co] * v[ci * T + tap];
        dw[i] = accumulate ? dw[i] + val : val;
    }
}

extern "C" int gim_wgrad_finish(const float* slabs, const float* bias_slabs, int n_slabs, const float* w, const float* sigma,
                                const float* u, const float* v, float* dw, float* db, float* scratch, int Cout, int Cin, int KH,
                                int fold, float* acc_dw, float* acc_db, void* stream) {
    GIM_CHECK_ARG(slabs && dw && n_slabs > 0, "wgrad_finish: bad args");
    GIM_CHECK_ARG((!db && !acc_db) || bias_slabs, "wgrad_finish: db needs bias_slabs");
    GIM_CHECK_ARG(!sigma || (w && u && v && scratch), "wgrad_finish: spectral form needs w, u, v, scratch");
    GIM_CHECK_ARG(fold >= 0 && fold <= 2 && (!fold || scratch), "wgrad_finish: bad fold / missing scratch");
    hipStream_t st = (hipStream_t)stream;
    const int T = KH * KH;
    const long long n = (long long)Cout * Cin * T;
    int blocks = (int)((n + 1023) / 1024);
    if (blocks > WF_BLOCKS) blocks = WF_BLOCKS;
    if (blocks < 1) blocks = 1;
    int sblocks = (int)((n + 63) / 64);  // wgrad_sum_kernel: 64 elements per block pass
    if (sblocks > WF_BLOCKS) sblocks = WF_BLOCKS;
    float* bias_out = acc_db ? acc_db : db;
    const int bias_acc = acc_db ? 1 : 0;
    // the plain (no sigma, no fold) form can sum straight into the accumulation target
    const bool direct = !sigma && !fold && acc_dw;
    if (fold == 0) {
        hipLaunchKernelGGL(wgrad_sum_kernel, dim3(sblocks), dim3(256), 0, st, slabs, n_slabs, n, sigma ? w : nullptr,
                           direct ? acc_dw : dw, sigma ? scratch : nullptr, bias_slabs, bias_out, Cout, direct ? 1 : 0, bias_acc);
        blocks = sblocks;  // number of <g, w> partials for the spectral apply
    } else {
        // 1) sum the slabs (folded layout) into scratch[512 ...], bias on the way; 2) un-fold into dw (+ <g, w>)
        const long long nf = (long long)Cout * Cin * (KH + 1) * (KH + 1);
        float* fsum = scratch + WF_BLOCKS;
        int fb = (int)((nf + 63) / 64);
        if (fb > WF_BLOCKS) fb = WF_BLOCKS;
        hipLaunchKernelGGL(wgrad_sum_kernel, dim3(fb), dim3(256), 0, st, slabs, n_slabs, nf, (const float*)nullptr, fsum,
                           (float*)nullptr, bias_slabs, bias_out, Cout, 0, bias_acc);
        hipLaunchKernelGGL(wgrad_unfold_kernel, dim3(blocks), dim3(256), 0, st, fsum, sigma ? w : nullptr, dw,
                           sigma ? scratch : nullptr, Cout, Cin, KH, fold);
    }
    if (sigma) {
        hipLaunchKernelGGL(wgrad_sn_apply_kernel, dim3(blocks), dim3(256), 0, st, dw, acc_dw ? acc_dw : dw, acc_dw ? 1 : 0, scratch,
                           blocks, sigma, u, v, n, Cin, T);
    } else if (acc_dw && !direct) {
        GIM_CHECK_ARG(false, "wgrad_finish: acc_dw without sigma is only supported for fold == 0");
    }
    return gim_check_launch("gim_wgrad_finish");
}

// -------------------------------------------------------------------------------------------------
// batched power iteration: ONE round (one iteration of every conv of a model) in 4 launches.
// The iterations are data independent, so a model runs all its rounds up front instead of 4 tiny launches
// per conv call.  Job table and block->job maps are static per model; per-round outputs live at static
// offsets from `out_base`.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void snb_colpart_kernel(const gim_sn_job* __restrict__ jobs, const int4* __restrict__ tab,
                                                          float* __restrict__ out_base) {
    const int4 e = tab[blockIdx.x];  // {job, column block, row chunk, rows per chunk}
    const gim_sn_job jb = jobs[e.x];
    const int K = jb.Cin * jb.KH * jb.KH;
    const int p = e.y * 256 + threadIdx.x;
    if (p >= K) return;
    const int c0 = e.z * e.w, c1 = min(jb.Cout, c0 + e.w);
    float acc = 0.f;
    for (int co = c0; co < c1; ++co) acc += jb.w[(long long)co * K + p] * jb.u[co];
    (out_base + jb.off_scratch)[(long long)e.z * K + p] = acc;
}

// One block of 1024 threads per job (the jobs are few - one per conv - and a 256-thread block walking the 4608-5120 columns of
// the largest ones was a 46 us chain of dependent-latency loads at the head of every forward pass).
__global__ __launch_bounds__(1024) void snb_vnorm_kernel(const gim_sn_job* __restrict__ jobs, float* __restrict__ out_base, int training) {
    __shared__ float red[16];
    const gim_sn_job jb = jobs[blockIdx.x];
    const int T = jb.KH * jb.KH, K = jb.Cin * T;
    const int R = min(SN_R, (jb.Cout + 63) / 64);
    float* part = out_base + jb.off_scratch;
    float* v_phys = part + (long long)SN_R * K;
    float* v_out = out_base + jb.off_v;
    if (training) {
        float ss = 0.f;
        for (int p = threadIdx.x; p < K; p += 1024) {
            float a = 0.f;
            for (int r = 0; r < R; ++r) a += part[(long long)r * K + p];
            v_phys[p] = a;
            ss += a * a;
        }
        ss = wave_sum(ss);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) tot += red[i];
        const float inv = 1.0f / fmaxf(sqrtf(tot), 1e-12f);
        for (int p = threadIdx.x; p < K; p += 1024) {   // each thread re-reads only what it wrote itself
            const float val = v_phys[p] * inv;
            const int tap = p / jb.Cin, ci = p - tap * jb.Cin;
            const int q = ci * T + tap;
            v_phys[p] = val;
            jb.v[q] = val;
            v_out[q] = val;
        }
    } else {
        for (int q = threadIdx.x; q < K; q += 1024) {
            const int ci = q / T, tap = q - ci * T;
            const float val = jb.v[q];
            v_phys[tap * jb.Cin + ci] = val;
            v_out[q] = val;
        }
    }
}

__global__ __launch_bounds__(256) void snb_rowdot_kernel(const gim_sn_job* __restrict__ jobs, const int2* __restrict__ tab,
                                                         float* __restrict__ out_base) {
    const int2 e = tab[blockIdx.x];  // {job, row block}
    const gim_sn_job jb = jobs[e.x];
    const int K = jb.Cin * jb.KH * jb.KH;
    const int co = e.y * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (co >= jb.Cout) return;
    const float* v_phys = out_base + jb.off_scratch + (long long)SN_R * K;
    float* tvec = out_base + jb.off_scratch + (long long)(SN_R + 1) * K;
    const float* row = jb.w + (long long)co * K;
    float acc = 0.f;
    for (int p = lane; p < K; p += 64) acc += row[p] * v_phys[p];
    acc = wave_sum(acc);
    if (lane == 0) tvec[co] = acc;
}

__global__ __launch_bounds__(256) void snb_final_kernel(const gim_sn_job* __restrict__ jobs, float* __restrict__ out_base, int training) {
    __shared__ float red[4];
    const gim_sn_job jb = jobs[blockIdx.x];
    const int K = jb.Cin * jb.KH * jb.KH;
    const float* tvec = out_base + jb.off_scratch + (long long)(SN_R + 1) * K;
    float* u_out = out_base + jb.off_u;
    float inv = 0.f;
    if (training) {
        float ss = 0.f;
        for (int c = threadIdx.x; c < jb.Cout; c += 256) ss += tvec[c] * tvec[c];
        inv = 1.0f / fmaxf(sqrtf(block_sum_256(ss, red)), 1e-12f);
    }
    float dot = 0.f;
    for (int c = threadIdx.x; c < jb.Cout; c += 256) {
        float uv;
        if (training) {
            uv = tvec[c] * inv;
            jb.u[c] = uv;
        } else {
            uv = jb.u[c];
        }
        u_out[c] = uv;
        dot += uv * tvec[c];
    }
    dot = block_sum_256(dot, red);
    if (threadIdx.x == 0) (out_base + jb.off_sigma)[0] = dot;
}

extern "C" int gim_spectral_sigma_batched(const gim_sn_job* jobs, int n_jobs, const int32_t* tab_cols, int n_col_blocks,
                                          const int32_t* tab_rows, int n_row_blocks, float* out_base, int training, void* stream) {
    GIM_CHECK_ARG(jobs && n_jobs > 0 && tab_cols && tab_rows && out_base && n_col_blocks > 0 && n_row_blocks > 0,
                  "spectral_sigma_batched: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (training)
        hipLaunchKernelGGL(snb_colpart_kernel, dim3(n_col_blocks), dim3(256), 0, st, jobs, reinterpret_cast<const int4*>(tab_cols), out_base);
    hipLaunchKernelGGL(snb_vnorm_kernel, dim3(n_jobs), dim3(1024), 0, st, jobs, out_base, training);
    hipLaunchKernelGGL(snb_rowdot_kernel, dim3(n_row_blocks), dim3(256), 0, st, jobs, reinterpret_cast<const int2*>(tab_rows), out_base);
    hipLaunchKernelGGL(snb_final_kernel, dim3(n_jobs), dim3(256), 0, st, jobs, out_base, training);
    return gim_check_launch("gim_spectral_sigma_batched");
}

// -------------------------------------------------------------------------------------------------
// batched weight-gradient finish: every conv / linear of one backward pass in TWO launches.
// Each job's raw gradient was accumulated (float atomics) into a pre-zeroed arena slot by gim_conv2d_wgrad_acc; here it is
// un-folded (pool / sub-pixel forms), pushed through the spectral-norm chain rule and ADDED into the optimizer's flat
// gradient bucket.  One block handles WGQ_CHUNK elements of one job (table built on the host, static per model).
// -------------------------------------------------------------------------------------------------
#define WGQ_CHUNK 4096

// (All index arithmetic is 32-bit - the caller guarantees Cout*(K+1)^2*Cin < 2^31 per job, ops.WgradQueue checks it: the first
//  version's 64-bit divisions per element made these two HBM-bound passes 3x slower than the bytes they move.)
__device__ __forceinline__ float wgq_load(const gim_wgrad_job& jb, unsigned i) {
    if (jb.fold == 0) return jb.src[i];
    const unsigned K = jb.K, KF = K + 1, Cin = jb.Cin, Cout = jb.Cout;
    const unsigned r1 = i / Cin, ci = i - r1 * Cin;
    const unsigned r2 = r1 / K, kw = r1 - r2 * K;
    if (jb.fold == 3)   // row-padded slot of gim_conv2d_wgrad_rows_acc: [Cout][K][K * Cin rounded up to 16]
        return jb.src[r2 * ((K * Cin + 15u) & ~15u) + kw * Cin + ci];
    const unsigned co = r2 / K, kh = r2 - co * K;
    float g = 0.f;
#pragma unroll
    for (unsigned dh = 0; dh < 2; ++dh)
#pragma unroll
        for (unsigned dwd = 0; dwd < 2; ++dwd) {
            const unsigned a = kh + dh, b = kw + dwd;
            if (jb.fold == 1) g += jb.src[((co * KF + a) * KF + b) * Cin + ci];
            else g += jb.src[((ci * KF + (K - a)) * KF + (K - b)) * Cout + co];
        }
    return jb.fold == 1 ? 0.25f * g : g;
}

__device__ __forceinline__ bool wgq_aligned16(const void* a, const void* b, const void* c) {
    return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0;
}

__global__ __launch_bounds__(256) void wgq_reduce_kernel(const gim_wgrad_job* __restrict__ jobs, const int2* __restrict__ tab) {
    __shared__ float red[4];
    const int2 e = tab[blockIdx.x];  // {job, chunk}
    const gim_wgrad_job jb = jobs[e.x];
    const unsigned n = (unsigned)jb.Cout * jb.K * jb.K * jb.Cin;
    const unsigned i0 = (unsigned)e.y * WGQ_CHUNK;
    const unsigned cnt = min((unsigned)WGQ_CHUNK, n - i0);
    const bool sn = jb.sigma != nullptr;
    float dot = 0.f;
    if (sn && jb.fold == 2 && (jb.Cout & 63) == 0 && (jb.Cin & 63) == 0) {
        // Sub-pixel (role-swapped) jobs hold G[ci][KF][KF][co]; the gradient wants [co][K][K][ci].  Read by flat output index,
        // consecutive lanes (ci) are KF*KF*Cout floats apart in G - one 64-byte sector per lane and load.  Instead this block takes
        // a 64 co x 64 ci tile of ONE tap (the chunk id is re-read as (co tile, ci tile, tap): the same 4096 elements per block,
        // every element covered once, so `partial` and `tmp` keep their meaning), reads G along co, turns the tile in LDS and
        // writes along ci.
        __shared__ float tile[64][65];
        const unsigned K = jb.K, KF = K + 1, T = K * K, Cin = jb.Cin, Cout = jb.Cout, tiles_ci = Cin >> 6;
        const unsigned tap = (unsigned)e.y % T, r = (unsigned)e.y / T;
        const unsigned ci0 = (r % tiles_ci) << 6, co0 = (r / tiles_ci) << 6;
        const unsigned kh = tap / K, kw = tap - kh * K;
        const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll 4
        for (unsigned j = 0; j < 16; ++j) {
            const unsigned ci = ci0 + wv + 4 * j;
            const float* base = jb.src + (((size_t)ci * KF + (K - kh)) * KF + (K - kw)) * Cout + co0 + lane;
            // taps (K-kh-dh, K-kw-dw), dh, dw in {0, 1}
            tile[wv + 4 * j][lane] = (base[0] + base[-(long)Cout]) + (base[-(long)(KF * Cout)] + base[-(long)((KF + 1) * Cout)]);
        }
        __syncthreads();
#pragma unroll 4
        for (unsigned j = 0; j < 16; ++j) {
            const unsigned co = co0 + wv + 4 * j;
            const float g = tile[lane][wv + 4 * j];
            const unsigned i = ((co * K + kh) * K + kw) * Cin + ci0 + lane;
            jb.tmp[i] = g;
            dot += g * jb.w[i];
        }
    } else if (sn && jb.fold == 0 && (n & 3) == 0 && wgq_aligned16(jb.src, jb.w, jb.w)) {
        // <G, W> only (the apply kernel reads G again): 16-byte loads
        for (unsigned t = threadIdx.x * 4; t < cnt; t += 1024) {
            const float4 g = *reinterpret_cast<const float4*>(jb.src + i0 + t);
            const float4 w = *reinterpret_cast<const float4*>(jb.w + i0 + t);
            dot += g.x * w.x + g.y * w.y + g.z * w.z + g.w * w.w;
        }
    } else if (!sn && jb.fold == 0 && jb.exclusive && (n & 3) == 0 && wgq_aligned16(jb.src, jb.grad_w, jb.grad_w)) {
        // the only job of this gradient in the pass: plain 16-byte read-modify-write instead of float atomics
        for (unsigned t = threadIdx.x * 4; t < cnt; t += 1024) {
            const float4 g = *reinterpret_cast<const float4*>(jb.src + i0 + t);
            float4* dst = reinterpret_cast<float4*>(jb.grad_w + i0 + t);
            float4 o = *dst;
            o.x += g.x; o.y += g.y; o.z += g.z; o.w += g.w;
            *dst = o;
        }
    } else {
        // lane-consecutive elements: the float atomics of a wave then hit 256 consecutive bytes (4 elements per lane made every
        // atomic instruction span 1 KB and the pass 1.4x slower than the scalar form)
#pragma unroll 4
        for (unsigned t = threadIdx.x; t < cnt; t += 256) {
            const unsigned i = i0 + t;
            const float g = wgq_load(jb, i);
            if (sn) {
                if (jb.fold) jb.tmp[i] = g;
                dot += g * jb.w[i];
            } else {   // several jobs (calls of the same conv) may target one gradient
                atomicAdd(&jb.grad_w[i], g);
            }
        }
    }
    if (sn) {
        dot = block_sum_256(dot, red);
        if (threadIdx.x == 0) jb.partial[e.y] = dot;
    }
    if (e.y == 0 && jb.bias_src)
        for (int c = threadIdx.x; c < jb.Cout; c += 256) atomicAdd(&jb.grad_b[c], jb.bias_src[c]);
}

__global__ __launch_bounds__(256) void wgq_apply_kernel(const gim_wgrad_job* __restrict__ jobs, const int2* __restrict__ tab) {
    __shared__ float red[4];
    const int2 e = tab[blockIdx.x];
    const gim_wgrad_job jb = jobs[e.x];
    float d = 0.f;
    for (int i = threadIdx.x; i < jb.n_chunks; i += 256) d += jb.partial[i];
    d = block_sum_256(d, red);
    const float inv = 1.0f / jb.sigma[0];
    const float coef = d * inv * inv;
    const unsigned Cin = jb.Cin, T = (unsigned)jb.K * jb.K, Kc = Cin * T;
    const unsigned n = (unsigned)jb.Cout * Kc;
    const unsigned i0 = (unsigned)e.y * WGQ_CHUNK;
    const unsigned cnt = min((unsigned)WGQ_CHUNK, n - i0);
    const float* __restrict__ g = jb.fold ? jb.tmp : jb.src;
    // (co, p) of the chunk's first element once per block; per element a carry (only layers with fewer than 4096 weights per
    // output channel wrap more than once) and ONE 32-bit division by Cin
    const unsigned co0 = i0 / Kc, p0 = i0 - co0 * Kc;
    if (jb.exclusive && (Cin & 3) == 0 && wgq_aligned16(g, jb.grad_w, jb.grad_w)) {
        // the only job of this gradient in the pass: 16-byte read-modify-write (4 consecutive input channels share the output
        // channel and the tap)
        for (unsigned t = threadIdx.x * 4; t < cnt; t += 1024) {
            unsigned p = p0 + t, co = co0;
            if (p >= Kc) {
                const unsigned q = p / Kc;
                co += q;
                p -= q * Kc;
            }
            const unsigned tap = p / Cin, ci = p - tap * Cin;
            const float4 gv = *reinterpret_cast<const float4*>(g + i0 + t);
            const float cu = coef * jb.u[co];
            const float* v = jb.v + ci * T + tap;
            float4* dst = reinterpret_cast<float4*>(jb.grad_w + i0 + t);
            float4 o = *dst;
            o.x += gv.x * inv - cu * v[0];
            o.y += gv.y * inv - cu * v[T];
            o.z += gv.z * inv - cu * v[2 * T];
            o.w += gv.w * inv - cu * v[3 * T];
            *dst = o;
        }
        return;
    }
#pragma unroll 4
    for (unsigned t = threadIdx.x; t < cnt; t += 256) {
        unsigned p = p0 + t, co = co0;
        if (p >= Kc) {
            const unsigned q = p / Kc;
            co += q;
            p -= q * Kc;
        }
        const unsigned tap = p / Cin, ci = p - tap * Cin;
        const unsigned i = i0 + t;
        atomicAdd(&jb.grad_w[i], g[i] * inv - coef * jb.u[co] * jb.v[ci * T + tap]);
    }
}

extern "C" int gim_wgrad_finish_batched(const gim_wgrad_job* jobs, int n_jobs, const int32_t* tab, int n_blocks,
                                        const int32_t* tab_sn, int n_blocks_sn, void* stream) {
    GIM_CHECK_ARG(jobs && n_jobs > 0 && tab && n_blocks > 0 && n_blocks_sn >= 0 && (tab_sn || n_blocks_sn == 0),
                  "wgrad_finish_batched: bad args");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wgq_reduce_kernel, dim3(n_blocks), dim3(256), 0, st, jobs, reinterpret_cast<const int2*>(tab));
    if (n_blocks_sn > 0)
        hipLaunchKernelGGL(wgq_apply_kernel, dim3(n_blocks_sn), dim3(256), 0, st, jobs, reinterpret_cast<const int2*>(tab_sn));
    return gim_check_launch("gim_wgrad_finish_batched");
}

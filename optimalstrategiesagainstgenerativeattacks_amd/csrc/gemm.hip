// Batched strided fp32 GEMM on v_mfma_f32_32x32x2_f32 for the SelfAttention bmm's and their
// gradients (N = H*W tokens <= 1024, C' = C/8): C[b](i,j) = sum_k A[b](i,k) * B[b](k,j).
// 64x64 tile per workgroup (4 waves, one 32x32 accumulator each), K step 16.  Arbitrary element
// strides; the thread->element mapping of each operand follows its unit-stride axis so that global
// loads stay coalesced for all four transpose combinations.  LDS images are k-major, rows padded to
// 65 floats so that both fill patterns write conflict-free.
#include "common.h"

#define GB 64
#define GK 16
#define GLD 65

struct BgP {
    const float* A;
    const float* B;
    float* C;
    int M, N, K;
    long long sAb, sAi, sAk, sBb, sBk, sBj;
};

__global__ __launch_bounds__(256) void bgemm_kernel(const BgP p) {
    __shared__ float As[2][GK * GLD];
    __shared__ float Bs[2][GK * GLD];
    const int t = threadIdx.x;
    const int i0 = blockIdx.y * GB, j0 = blockIdx.x * GB;
    const float* A = p.A + (long long)blockIdx.z * p.sAb;
    const float* B = p.B + (long long)blockIdx.z * p.sBb;
    const bool a_kfast = (p.sAk == 1);
    const bool b_jfast = (p.sBj == 1);

    float ra[4], rb[4];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = t + 256 * e;
            int i, k;
            if (a_kfast) { k = idx & 15; i = idx >> 4; } else { i = idx & 63; k = idx >> 6; }
            const bool v = (i0 + i) < p.M && (k0 + k) < p.K;
            ra[e] = v ? A[(long long)(i0 + i) * p.sAi + (long long)(k0 + k) * p.sAk] : 0.f;
            int j, kb;
            if (b_jfast) { j = idx & 63; kb = idx >> 6; } else { kb = idx & 15; j = idx >> 4; }
            const bool vb = (j0 + j) < p.N && (k0 + kb) < p.K;
            rb[e] = vb ? B[(long long)(k0 + kb) * p.sBk + (long long)(j0 + j) * p.sBj] : 0.f;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = t + 256 * e;
            int i, k;
            if (a_kfast) { k = idx & 15; i = idx >> 4; } else { i = idx & 63; k = idx >> 6; }
            As[buf][k * GLD + i] = ra[e];
            int j, kb;
            if (b_jfast) { j = idx & 63; kb = idx >> 6; } else { kb = idx & 15; j = idx >> 4; }
            Bs[buf][kb * GLD + j] = rb[e];
        }
    };

    const int lane = t & 63, r = lane & 31, h = lane >> 5, wv = t >> 6;
    const int wm0 = (wv >> 1) * 32, wn0 = (wv & 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    const int nk = (p.K + GK - 1) / GK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) load_tiles((ks + 1) * GK);
#pragma unroll
        for (int kp = 0; kp < GK / 2; ++kp) {
            const float a = As[buf][(2 * kp + h) * GLD + wm0 + r];
            const float b = Bs[buf][(2 * kp + h) * GLD + wn0 + r];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (ks + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }
    float* C = p.C + (long long)blockIdx.z * p.M * p.N;
    const int j = j0 + wn0 + r;
    if (j < p.N) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = i0 + wm0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (i < p.M) C[(long long)i * p.N + j] = acc[e];
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Grouped form: many small independent products in ONE launch, each with its own operands, strides and output (a job table on
// the device, one workgroup per (job, 64 x 64 tile)).  The generator's 36 style projections (models/model_blocks.py:776-865:
// lin1_mean / lin1_std / lin2_mean / lin2_std of every AdaResBlock2 / AdaResBlockUp2, nn.Linear(512, C) on ONE shared style
// matrix [B * n, 512]) are 36 forward, 36 input-gradient and 36 weight-gradient products of 5-7 us each, launch-bound; as
// three grouped launches they are what they compute: ~1 GFLOP.  flags bit 0: C += product (single writer per element: the
// weight gradients add into the optimizer's bucket); bit 1: combine with float atomics (all jobs add into one C: the gradient
// w.r.t. the shared input, pre-zeroed by the caller).  bias: added per output column (forward).
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bgemm_grouped_kernel(const gim_gemm_job* __restrict__ jobs, const int* __restrict__ tiles) {
    __shared__ float As[2][GK * GLD];
    __shared__ float Bs[2][GK * GLD];
    const int t = threadIdx.x;
    const int job = tiles[3 * blockIdx.x], i0 = tiles[3 * blockIdx.x + 1] * GB, j0 = tiles[3 * blockIdx.x + 2] * GB;
    const gim_gemm_job p = jobs[job];
    const float* A = p.A;
    const float* B = p.B;
    const bool a_kfast = (p.sAk == 1);
    const bool b_jfast = (p.sBj == 1);
    float ra[4], rb[4];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = t + 256 * e;
            int i, k;
            if (a_kfast) { k = idx & 15; i = idx >> 4; } else { i = idx & 63; k = idx >> 6; }
            const bool v = (i0 + i) < p.M && (k0 + k) < p.K;
            ra[e] = v ? A[(long long)(i0 + i) * p.sAi + (long long)(k0 + k) * p.sAk] : 0.f;
            int j, kb;
            if (b_jfast) { j = idx & 63; kb = idx >> 6; } else { kb = idx & 15; j = idx >> 4; }
            const bool vb = (j0 + j) < p.N && (k0 + kb) < p.K;
            rb[e] = vb ? B[(long long)(k0 + kb) * p.sBk + (long long)(j0 + j) * p.sBj] : 0.f;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = t + 256 * e;
            int i, k;
            if (a_kfast) { k = idx & 15; i = idx >> 4; } else { i = idx & 63; k = idx >> 6; }
            As[buf][k * GLD + i] = ra[e];
            int j, kb;
            if (b_jfast) { j = idx & 63; kb = idx >> 6; } else { kb = idx & 15; j = idx >> 4; }
            Bs[buf][kb * GLD + j] = rb[e];
        }
    };
    const int lane = t & 63, r = lane & 31, h = lane >> 5, wv = t >> 6;
    const int wm0 = (wv >> 1) * 32, wn0 = (wv & 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int nk = (p.K + GK - 1) / GK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) load_tiles((ks + 1) * GK);
#pragma unroll
        for (int kp = 0; kp < GK / 2; ++kp) {
            const float a = As[buf][(2 * kp + h) * GLD + wm0 + r];
            const float b = Bs[buf][(2 * kp + h) * GLD + wn0 + r];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (ks + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }
    const int j = j0 + wn0 + r;
    if (j < p.N) {
        const float bv = p.bias ? p.bias[j] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = i0 + wm0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (i < p.M) {
                float* c = p.C + (long long)i * p.ldc + j;
                const float v = acc[e] + bv;
                if (p.flags & 2) atomicAdd(c, v);
                else if (p.flags & 1) *c += v;
                else *c = v;
            }
        }
    }
}

// column sums of many small matrices in one launch: out[j] (+)= sum_i A[i][j]  (the bias gradients of the grouped linears);
// one workgroup per (job, block of 64 columns); the job's B / C / bias fields are unused, out = C, flags bit 0 = accumulate
__global__ __launch_bounds__(256) void colsum_grouped_kernel(const gim_gemm_job* __restrict__ jobs, const int* __restrict__ tiles) {
    __shared__ float red[4][64];
    const int job = tiles[3 * blockIdx.x], j0 = tiles[3 * blockIdx.x + 2] * 64;
    const gim_gemm_job p = jobs[job];
    const int j = j0 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    float s = 0.f;
    if (j < p.N)
        for (int i = part; i < p.M; i += 4) s += p.A[(long long)i * p.sAi + (long long)j * p.sAk];
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && j < p.N) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (p.flags & 1) p.C[j] += v;
        else p.C[j] = v;
    }
}

extern "C" int gim_bgemm_grouped(const gim_gemm_job* jobs, const int32_t* tiles, int n_tiles, void* stream) {
    GIM_CHECK_ARG(jobs && tiles && n_tiles > 0, "bgemm_grouped: bad args");
    hipLaunchKernelGGL(bgemm_grouped_kernel, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, jobs, (const int*)tiles);
    return gim_check_launch("gim_bgemm_grouped");
}

extern "C" int gim_colsum_grouped(const gim_gemm_job* jobs, const int32_t* tiles, int n_tiles, void* stream) {
    GIM_CHECK_ARG(jobs && tiles && n_tiles > 0, "colsum_grouped: bad args");
    hipLaunchKernelGGL(colsum_grouped_kernel, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, jobs, (const int*)tiles);
    return gim_check_launch("gim_colsum_grouped");
}

// Vector form of the same product for operands whose unit-stride extent is a multiple of 4 and 16-byte aligned (every
// SelfAttention product of the benchmark shapes): one 16-byte global load per thread, operand and 16 K values instead of four
// scalar loads with 64-bit address arithmetic each, K step 32 (16 MFMAs per wave between barriers instead of 8), LDS rows of 68
// floats (16-byte aligned rows for the b128 stores of an m- / n-contiguous operand; the four scalar stores of a k-contiguous
// one and the scalar operand reads stay conflict-free per half-wave).  AK / BK: the operand's unit stride is along K.
#define GLV 68
#define GKV 32
template <bool AK, bool BK_>
__global__ __launch_bounds__(256) void bgemm_vec_kernel(const BgP p) {
    __shared__ __attribute__((aligned(16))) float As[2][GKV * GLV];
    __shared__ __attribute__((aligned(16))) float Bs[2][GKV * GLV];
    const int t = threadIdx.x;
    const int i0 = blockIdx.y * GB, j0 = blockIdx.x * GB;
    const float* A = p.A + (long long)blockIdx.z * p.sAb;
    const float* B = p.B + (long long)blockIdx.z * p.sBb;
    // per-thread element of each 16-byte load.  k-contiguous operand: row = t >> 3 (+32), k quad = t & 7;
    // m- / n-contiguous operand: k row = t >> 4 (+16), column quad = t & 15
    f32x4 ra[2], rb[2];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if constexpr (AK) {
                const int i = i0 + (t >> 3) + 32 * e, k = k0 + 4 * (t & 7);
                ra[e] = (i < p.M && k < p.K) ? *reinterpret_cast<const f32x4*>(A + (long long)i * p.sAi + k) : zero4;
            } else {
                const int k = k0 + (t >> 4) + 16 * e, i = i0 + 4 * (t & 15);
                ra[e] = (i < p.M && k < p.K) ? *reinterpret_cast<const f32x4*>(A + (long long)k * p.sAk + i) : zero4;
            }
            if constexpr (BK_) {
                const int j = j0 + (t >> 3) + 32 * e, k = k0 + 4 * (t & 7);
                rb[e] = (j < p.N && k < p.K) ? *reinterpret_cast<const f32x4*>(B + (long long)j * p.sBj + k) : zero4;
            } else {
                const int k = k0 + (t >> 4) + 16 * e, j = j0 + 4 * (t & 15);
                rb[e] = (j < p.N && k < p.K) ? *reinterpret_cast<const f32x4*>(B + (long long)k * p.sBk + j) : zero4;
            }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if constexpr (AK) {
#pragma unroll
                for (int q = 0; q < 4; ++q) As[buf][(4 * (t & 7) + q) * GLV + (t >> 3) + 32 * e] = ra[e][q];
            } else {
                *reinterpret_cast<f32x4*>(&As[buf][((t >> 4) + 16 * e) * GLV + 4 * (t & 15)]) = ra[e];
            }
            if constexpr (BK_) {
#pragma unroll
                for (int q = 0; q < 4; ++q) Bs[buf][(4 * (t & 7) + q) * GLV + (t >> 3) + 32 * e] = rb[e][q];
            } else {
                *reinterpret_cast<f32x4*>(&Bs[buf][((t >> 4) + 16 * e) * GLV + 4 * (t & 15)]) = rb[e];
            }
        }
    };
    const int lane = t & 63, r = lane & 31, h = lane >> 5, wv = t >> 6;
    const int wm0 = (wv >> 1) * 32, wn0 = (wv & 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int nk = (p.K + GKV - 1) / GKV;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) load_tiles((ks + 1) * GKV);
#pragma unroll
        for (int kp = 0; kp < GKV / 2; ++kp) {
            const float a = As[buf][(2 * kp + h) * GLV + wm0 + r];
            const float b = Bs[buf][(2 * kp + h) * GLV + wn0 + r];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (ks + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }
    float* C = p.C + (long long)blockIdx.z * p.M * p.N;
    const int j = j0 + wn0 + r;
    if (j < p.N) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = i0 + wm0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (i < p.M) C[(long long)i * p.N + j] = acc[e];
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Attention probabilities in one kernel:  P[b][i][j] = softmax_i( f[b][i][:] . g[b][j][:] )   (models/model_blocks.py:540-543:
// energy = bmm(f^T, g), Softmax(dim=-2)).  The unfused form writes the energy (T x T per image), reads it back for the softmax
// and writes P - 3 passes over 21 MB at the benchmark shape for a product with K = C/8 = 16.  Here one workgroup owns ALL T
// rows of a block of 64 columns: the energy tile lives in the MFMA accumulators, the column maxima and sums are reduced through
// the wave (lanes r / r + 32 hold different rows of a column) and 2 KB of LDS (the four waves), and only P is written.
// T = 256 tokens (the 16 x 16 map the attention sits on at 64 x 64 and 128 x 128 input), K = 16.
// -------------------------------------------------------------------------------------------------
#define AT_T 256
#define AT_K 16
#define AT_LD 260   // k-major LDS rows of the f tile (256 + pad, 16-byte aligned)
__global__ __launch_bounds__(256) void attn_prob_kernel(const float* __restrict__ f, const float* __restrict__ g, float* __restrict__ P) {
    __shared__ __attribute__((aligned(16))) float Fs[AT_K * AT_LD];   // [k][i]
    __shared__ __attribute__((aligned(16))) float Gs[AT_K * 68];      // [k][j]
    __shared__ float red[4][64];
    const int t = threadIdx.x, b = blockIdx.y, j0 = blockIdx.x * 64;
    const float* fb = f + (long long)b * AT_T * AT_K;
    const float* gb = g + ((long long)b * AT_T + j0) * AT_K;
    // f: 256 rows x 16 k = 1024 float4, 4 per thread; g: 64 rows x 16 k = 256 float4, 1 per thread; stored k-major
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int idx = t + 256 * e, i = idx >> 2, kq = idx & 3;
        const f32x4 v = *reinterpret_cast<const f32x4*>(fb + (long long)i * AT_K + 4 * kq);
#pragma unroll
        for (int q = 0; q < 4; ++q) Fs[(4 * kq + q) * AT_LD + i] = v[q];
    }
    {
        const int j = t >> 2, kq = t & 3;
        const f32x4 v = *reinterpret_cast<const f32x4*>(gb + (long long)j * AT_K + 4 * kq);
#pragma unroll
        for (int q = 0; q < 4; ++q) Gs[(4 * kq + q) * 68 + j] = v[q];
    }
    __syncthreads();
    const int lane = t & 63, r = lane & 31, h = lane >> 5, wv = t >> 6;
    f32x16 acc[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[rb][cb][e] = 0.f;
#pragma unroll
    for (int kp = 0; kp < AT_K / 2; ++kp) {
        float a[2], bb[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) a[rb] = Fs[(2 * kp + h) * AT_LD + 64 * wv + 32 * rb + r];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) bb[cb] = Gs[(2 * kp + h) * 68 + 32 * cb + r];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rb], bb[cb], acc[rb][cb], 0, 0, 0);
    }
    // column (j) statistics over all 256 rows: in-lane over 32 rows, across the two lane halves, across the four waves
    float mx[2], sm[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        float m = -INFINITY;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int e = 0; e < 16; ++e) m = fmaxf(m, acc[rb][cb][e]);
        m = fmaxf(m, __shfl_xor(m, 32));
        if (h == 0) red[wv][32 * cb + r] = m;
        mx[cb] = m;
    }
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int c = 32 * cb + r;
        mx[cb] = fmaxf(fmaxf(red[0][c], red[1][c]), fmaxf(red[2][c], red[3][c]));
    }
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        float s = 0.f;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[rb][cb][e] = __expf(acc[rb][cb][e] - mx[cb]);
                s += acc[rb][cb][e];
            }
        s += __shfl_xor(s, 32);
        if (h == 0) red[wv][32 * cb + r] = s;
    }
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int c = 32 * cb + r;
        sm[cb] = 1.0f / (red[0][c] + red[1][c] + red[2][c] + red[3][c]);
    }
    float* Pb = P + (long long)b * AT_T * AT_T + j0;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = 64 * wv + 32 * rb + (e & 3) + 8 * (e >> 2) + 4 * h;
                Pb[(long long)i * AT_T + 32 * cb + r] = acc[rb][cb][e] * sm[cb];
            }
}

extern "C" int gim_attn_prob_fwd(const float* f, const float* g, float* P, int batch, int T, int K, void* stream) {
    GIM_CHECK_ARG(f && g && P && batch > 0, "attn_prob_fwd: bad args");
    GIM_CHECK_ARG(T == AT_T && K == AT_K, "attn_prob_fwd: built for T = 256 tokens and K = 16 channels (use gim_bgemm + gim_softmax_dim1_fwd otherwise)");
    GIM_CHECK_ARG(batch <= 65535 && !(((uintptr_t)f | (uintptr_t)g) & 15), "attn_prob_fwd: batch too large for grid.y or operands not 16-byte aligned");
    hipLaunchKernelGGL(attn_prob_kernel, dim3(AT_T / 64, batch), dim3(256), 0, (hipStream_t)stream, f, g, P);
    return gim_check_launch("gim_attn_prob_fwd");
}

extern "C" int gim_bgemm(const float* A, const float* B, float* C, int batch, int M, int N, int K, int64_t sAb, int64_t sAi,
                         int64_t sAk, int64_t sBb, int64_t sBk, int64_t sBj, void* stream) {
    GIM_CHECK_ARG(A && B && C && batch > 0 && M > 0 && N > 0 && K > 0, "bgemm: bad args");
    GIM_CHECK_ARG(batch <= 65535, "bgemm: batch too large for grid.z");
    BgP p;
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K;
    p.sAb = sAb; p.sAi = sAi; p.sAk = sAk; p.sBb = sBb; p.sBk = sBk; p.sBj = sBj;
    dim3 g((N + GB - 1) / GB, (M + GB - 1) / GB, batch);
    // vector path: each operand's unit-stride extent (K for a k-contiguous operand, M / N otherwise) is a multiple of 4, its
    // other strides too, and the base pointers are 16-byte aligned - then every 16-byte load is aligned and never straddles a tile edge
    const bool ak = sAk == 1, bk = sBk == 1;
    const bool va = ak ? (K % 4 == 0 && sAi % 4 == 0) : (sAi == 1 && M % 4 == 0 && sAk % 4 == 0);
    const bool vb = bk ? (K % 4 == 0 && sBj % 4 == 0) : (sBj == 1 && N % 4 == 0 && sBk % 4 == 0);
    const bool al = (((uintptr_t)A | (uintptr_t)B) & 15) == 0 && sAb % 4 == 0 && sBb % 4 == 0;
    hipStream_t st = (hipStream_t)stream;
    if (va && vb && al) {
        if (ak && bk) hipLaunchKernelGGL((bgemm_vec_kernel<true, true>), g, dim3(256), 0, st, p);
        else if (ak) hipLaunchKernelGGL((bgemm_vec_kernel<true, false>), g, dim3(256), 0, st, p);
        else if (bk) hipLaunchKernelGGL((bgemm_vec_kernel<false, true>), g, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((bgemm_vec_kernel<false, false>), g, dim3(256), 0, st, p);
    } else {
        hipLaunchKernelGGL(bgemm_kernel, g, dim3(256), 0, st, p);
    }
    return gim_check_launch("gim_bgemm");
}

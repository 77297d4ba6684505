// Feasibility study: fp32 GEMM on the bf16 matrix pipe by exact 3-way operand splitting ("bf16x3").
//
//   x = hi + mid + lo   (three bf16 numbers, truncation split: 24-bit significand = 8 + 8 + 8 bits, EXACT)
//   x*y ~= hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi        (6 of the 9 partial products)
//
// The dropped terms (mid*lo, lo*mid, lo*lo) are <= 2^-23 |x*y|: the result is as accurate as the fp32 MFMA's own
// rounding, while the six v_mfma_f32_32x32x16_bf16 cost 6*32 = 192 cycles per 32x32x16 block against 8*64 = 512 for
// the eight v_mfma_f32_32x32x2_f32 (2.67x fewer matrix-pipe cycles).  Operands stay fp32 in HBM; they are split when
// the tile is written to LDS (5.5 VALU per element), three bf16 planes per k-row.
//
// C[M][N] = A[M][K] * B[N][K]^T, both operands k-contiguous (the layout of the conv kernels' fast path at one tap).
// Two kernels with the same tiling (128x128x16 per workgroup, 64x64 per wave, register-staged double buffer):
//   gemm_bf16x3  and  gemm_f32 (v_mfma_f32_32x32x2_f32)  for a like-for-like rate and accuracy comparison.
//   hipcc --offload-arch=gfx950 -O3 bf16x3_gemm.hip -o bf16x3_gemm && ./bf16x3_gemm
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, KB = 16;

__device__ __forceinline__ unsigned fbits(float f) { return __builtin_bit_cast(unsigned, f); }
__device__ __forceinline__ float bfloat(unsigned u) { return __builtin_bit_cast(float, u); }
// (upper 16 bits of b) << 16 | (upper 16 bits of a): two truncated bf16 in one dword, element order a, b
__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// 8 consecutive-k floats -> three planes of 8 bf16
template <int TERMS>
__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, u32x4& hi, u32x4& mid, u32x4& lo) {
    float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    unsigned r1[8], r2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float d1 = x[e] - bfloat(fbits(x[e]) & 0xFFFF0000u);      // exact
        r1[e] = fbits(d1);
        const float d2 = d1 - bfloat(r1[e] & 0xFFFF0000u);               // exact, <= 8 significant bits
        r2[e] = fbits(d2);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hi[j] = pack_hi16(fbits(x[2 * j]), fbits(x[2 * j + 1]));
        mid[j] = pack_hi16(r1[2 * j], r1[2 * j + 1]);
        lo[j] = pack_hi16(r2[2 * j], r2[2 * j + 1]);
    }
}

// LDS row: [hi: 8 dwords][mid: 8 dwords][lo: 8 dwords][pad 4]: 28-dword stride makes the 16-lane groups of a
// ds_read_b128 (rows {0-3,12-15,20-27} ...) start on 16 different multiples of 4 banks: conflict-free.
constexpr int LDR = 28;

template <int TERMS>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                      int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) unsigned lds[2 * (BM + BN) * LDR];
    unsigned* As = lds;
    unsigned* Bs = lds + 2 * BM * LDR;
    const int t = threadIdx.x;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int lrow = t >> 1, lhalf = t & 1;
    const float* ap = A + (size_t)(m0 + lrow) * K + 8 * lhalf;
    const float* bp = B + (size_t)(n0 + lrow) * K + 8 * lhalf;
    f32x4 ra0, ra1, rb0, rb1;
    auto load_g = [&](int k0) {
        ra0 = *reinterpret_cast<const f32x4*>(ap + k0);
        ra1 = *reinterpret_cast<const f32x4*>(ap + k0 + 4);
        rb0 = *reinterpret_cast<const f32x4*>(bp + k0);
        rb1 = *reinterpret_cast<const f32x4*>(bp + k0 + 4);
    };
    auto store_l = [&](int buf) {
        u32x4 h, m, l;
        split8<TERMS>(ra0, ra1, h, m, l);
        unsigned* d = As + buf * BM * LDR + lrow * LDR + 4 * lhalf;
        *reinterpret_cast<u32x4*>(d) = h;
        *reinterpret_cast<u32x4*>(d + 8) = m;
        if (TERMS > 3) *reinterpret_cast<u32x4*>(d + 16) = l;
        split8<TERMS>(rb0, rb1, h, m, l);
        d = Bs + buf * BN * LDR + lrow * LDR + 4 * lhalf;
        *reinterpret_cast<u32x4*>(d) = h;
        *reinterpret_cast<u32x4*>(d + 8) = m;
        if (TERMS > 3) *reinterpret_cast<u32x4*>(d + 16) = l;
    };
    const int lane = t & 63, r = lane & 31, h = lane >> 5, wv = t >> 6;
    const int wm0 = (wv >> 1) * 64, wn0 = (wv & 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int nk = K / KB;
    load_g(0);
    store_l(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) load_g((ks + 1) * KB);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned* Ab = As + buf * BM * LDR;
        const unsigned* Bb = Bs + buf * BN * LDR;
        constexpr int NP = TERMS > 3 ? 3 : 2;
        bf16x8 a[2][NP], b[2][NP];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                a[i][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Ab + (wm0 + 32 * i + r) * LDR + 8 * p + 4 * h));
                b[i][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Bb + (wn0 + 32 * i + r) * LDR + 8 * p + 4 * h));
            }
        // small terms first
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (TERMS > 3) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][NP - 1], b[j][0], acc[i][j], 0, 0, 0);   // lo*hi
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][NP - 1], acc[i][j], 0, 0, 0);   // hi*lo
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);        // mid*mid
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);            // mid*hi
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);            // hi*mid
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);            // hi*hi
            }
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < nk) store_l(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm0 + 32 * i + 8 * (e >> 2) + 4 * h + (e & 3);
                const int col = n0 + wn0 + 32 * j + r;
                C[(size_t)row * N + col] = acc[i][j][e];
            }
}

// the same tiling on the fp32 matrix pipe (k-rows of 16 floats padded to 20, one ds_read_b128 feeds 4 MFMAs with the k
// permutation 8kk+4h+t on both operands, as in csrc/conv_igemm.hip)
constexpr int LDK = 20;
__global__ __launch_bounds__(256, 2) void gemm_f32(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                   int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDK];
    float* As = lds;
    float* Bs = lds + 2 * BM * LDK;
    const int t = threadIdx.x;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int lrow = t >> 1, lhalf = t & 1;
    const float* ap = A + (size_t)(m0 + lrow) * K + 8 * lhalf;
    const float* bp = B + (size_t)(n0 + lrow) * K + 8 * lhalf;
    f32x4 ra0, ra1, rb0, rb1;
    auto load_g = [&](int k0) {
        ra0 = *reinterpret_cast<const f32x4*>(ap + k0);
        ra1 = *reinterpret_cast<const f32x4*>(ap + k0 + 4);
        rb0 = *reinterpret_cast<const f32x4*>(bp + k0);
        rb1 = *reinterpret_cast<const f32x4*>(bp + k0 + 4);
    };
    auto store_l = [&](int buf) {
        float* d = As + buf * BM * LDK + lrow * LDK + 8 * lhalf;
        *reinterpret_cast<f32x4*>(d) = ra0;
        *reinterpret_cast<f32x4*>(d + 4) = ra1;
        d = Bs + buf * BN * LDK + lrow * LDK + 8 * lhalf;
        *reinterpret_cast<f32x4*>(d) = rb0;
        *reinterpret_cast<f32x4*>(d + 4) = rb1;
    };
    const int lane = t & 63, r = lane & 31, h = lane >> 5, wv = t >> 6;
    const int wm0 = (wv >> 1) * 64, wn0 = (wv & 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int nk = K / KB;
    load_g(0);
    store_l(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) load_g((ks + 1) * KB);
        __builtin_amdgcn_sched_barrier(0);
        const float* Ab = As + buf * BM * LDK;
        const float* Bb = Bs + buf * BN * LDK;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const f32x4*>(Ab + (wm0 + 32 * i + r) * LDK + 8 * kk + 4 * h);
                b[i] = *reinterpret_cast<const f32x4*>(Bb + (wn0 + 32 * i + r) * LDK + 8 * kk + 4 * h);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < nk) store_l(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm0 + 32 * i + 8 * (e >> 2) + 4 * h + (e & 3);
                const int col = n0 + wn0 + 32 * j + r;
                C[(size_t)row * N + col] = acc[i][j][e];
            }
}

static double time_ms(void (*launch)(const float*, const float*, float*, int, int, int), const float* A, const float* B, float* C,
                      int M, int N, int K, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(A, B, C, M, N, K);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch(A, B, C, M, N, K);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

static void l_x6(const float* A, const float* B, float* C, int M, int N, int K) {
    hipLaunchKernelGGL(gemm_bf16x3<6>, dim3(M / BM, N / BN), dim3(256), 0, 0, A, B, C, M, N, K);
}
static void l_x3(const float* A, const float* B, float* C, int M, int N, int K) {
    hipLaunchKernelGGL(gemm_bf16x3<3>, dim3(M / BM, N / BN), dim3(256), 0, 0, A, B, C, M, N, K);
}
static void l_f32(const float* A, const float* B, float* C, int M, int N, int K) {
    hipLaunchKernelGGL(gemm_f32, dim3(M / BM, N / BN), dim3(256), 0, 0, A, B, C, M, N, K);
}

static double check(const std::vector<float>& hA, const std::vector<float>& hB, const float* dC, int M, int N, int K) {
    std::vector<float> hC((size_t)M * N);
    hipMemcpy(hC.data(), dC, hC.size() * sizeof(float), hipMemcpyDeviceToHost);
    double worst = 0;
    unsigned s = 12345;
    for (int q = 0; q < 4096; ++q) {
        s = s * 1664525u + 1013904223u;
        const int m = (s >> 8) % M;
        s = s * 1664525u + 1013904223u;
        const int n = (s >> 8) % N;
        double ref = 0, mag = 0;
        for (int k = 0; k < K; ++k) {
            const double p = (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k];
            ref += p;
            mag += fabs(p);
        }
        worst = fmax(worst, fabs(hC[(size_t)m * N + n] - ref) / mag);
    }
    return worst;
}

int main(int argc, char** argv) {
    const int shapes[][3] = {{65536, 128, 576}, {655360, 128, 576}, {163840, 256, 1152}, {40960, 512, 2304}, {16384, 512, 4608}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
        unsigned s = 777;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f + ((s >> 24) & 0xFF) * 1e-6f; };
        for (auto& v : hA) v = rnd();
        for (auto& v : hB) v = rnd() * 0.05f;
        float *A, *B, *C;
        hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, (size_t)M * N * 4);
        hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
        const double fl = 2.0 * M * N * K;
        const int reps = 10;
        const double t32 = time_ms(l_f32, A, B, C, M, N, K, reps);
        const double e32 = check(hA, hB, C, M, N, K);
        const double t6 = time_ms(l_x6, A, B, C, M, N, K, reps);
        const double e6 = check(hA, hB, C, M, N, K);
        const double t3 = time_ms(l_x3, A, B, C, M, N, K, reps);
        const double e3 = check(hA, hB, C, M, N, K);
        printf("M=%7d N=%4d K=%5d | fp32 MFMA %.3f ms %6.1f TF err %.2e | bf16x3 (6 terms) %.3f ms %6.1f TF err %.2e | (3 terms) %.3f ms %6.1f TF err %.2e\n",
               M, N, K, t32, fl / t32 / 1e9, e32, t6, fl / t6 / 1e9, e6, t3, fl / t3 / 1e9, e3);
        fflush(stdout);
        hipFree(A); hipFree(B); hipFree(C);
    }
    return 0;
}

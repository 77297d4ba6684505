// Feasibility study for round 2: bf16x3 3x3 convolution with the INPUT PATCH resident in LDS.
//
// The engine's bf16x3 kernel (csrc/conv_igemm.hip, PREC 1) re-loads and re-splits every activation element once per filter tap
// (and N tile): PMC shows VALU = 55 % of its instruction stream, MfmaUtil 38-46 %.  Here a workgroup loads the 4 x 66 pixel patch
// its 2 x 64 output pixels need ONCE per 32-channel half, splits it once into three bf16 planes in LDS, and then runs all 9 taps
// x 2 channel chunks (18 K steps of 16) out of that patch: the A operand of tap (ta, tb) is the same LDS image read at a constant
// offset.  Only the weight tile (64 output channels x 16 k) is streamed and split per K step.
//   VALU per 32-channel half and thread: 9 float4 of patch (198) + 18 weight float4 (396)  vs  216 MFMAs per wave.
// Shape family of the dominant layer: NHWC x [N][64][64][64], w [64][3][3][64] (k-contiguous), y [N][64][64][64], pad 1.
//   hipcc --offload-arch=gfx950 -O3 bf16x3_patch_conv.hip -o bf16x3_patch_conv && ./bf16x3_patch_conv
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int H = 64, W = 64, C = 64, CO = 64;
constexpr int PW = W + 2;                 // patch width (halo columns)
constexpr int PIX = 208;                  // bytes per patch pixel: 2 chunks x [hi 32 B][mid][lo] + 16 B pad (conflict-free ds_read_b128)
constexpr int PATCH = 4 * PW * PIX;       // 54912 B
constexpr int BROW = 28 * 4;              // bytes per weight row of a K step: [hi][mid][lo] + pad
constexpr int BTILE = CO * BROW;          // 7168 B

__device__ __forceinline__ void split4(const f32x4& x, u32x2& hi, u32x2& mid, u32x2& lo) {
    unsigned xb[4], r1[4], r2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float xe = x[e];
        xb[e] = __builtin_bit_cast(unsigned, xe);
        const float d1 = xe - __builtin_bit_cast(float, xb[e] & 0xFFFF0000u);
        r1[e] = __builtin_bit_cast(unsigned, d1);
        const float d2 = d1 - __builtin_bit_cast(float, r1[e] & 0xFFFF0000u);
        r2[e] = __builtin_bit_cast(unsigned, d2);
    }
    hi = u32x2{__builtin_amdgcn_perm(xb[1], xb[0], 0x07060302u), __builtin_amdgcn_perm(xb[3], xb[2], 0x07060302u)};
    mid = u32x2{__builtin_amdgcn_perm(r1[1], r1[0], 0x07060302u), __builtin_amdgcn_perm(r1[3], r1[2], 0x07060302u)};
    lo = u32x2{__builtin_amdgcn_perm(r2[1], r2[0], 0x07060302u), __builtin_amdgcn_perm(r2[3], r2[2], 0x07060302u)};
}

__global__ __launch_bounds__(256, 2) void conv3x3_patch_bf16x3(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) char lds[PATCH + 2 * BTILE];
    char* patch = lds;
    char* Bs = lds + PATCH;
    const int t = threadIdx.x, lane = t & 63, r = lane & 31, h = lane >> 5, wv = t >> 6;
    const int tile = blockIdx.x;                    // 32 tiles of 2 rows per image
    const int n = tile >> 5, y0 = (tile & 31) * 2;
    const float* xin = x + (size_t)n * H * W * C;

    f32x16 acc[2], acc2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[j][e] = 0.f; acc2[j][e] = 0.f; }

    // this wave's 32 output pixels: row yy of the tile, columns x0 .. x0 + 31; patch pixel of tap (0,0) = (yy, x0 + r)
    const int yy = wv >> 1, x0 = (wv & 1) * 32;
    const int a_lane = ((yy * PW) + (x0 + r)) * PIX + h * 16;
    const int b_lane = r * BROW + h * 16;             // weight row r (+32 for the second block)

    // weight tile of one K step: 64 rows x 16 k = 256 float4, one per thread
    const int brow = t >> 2, bq = t & 3;

    for (int half = 0; half < 2; ++half) {
        __syncthreads();                              // the previous half's MFMAs are done with the patch
        // ---- patch: 4 rows x 66 pixels x 32 channels; 8 float4 per pixel -> 2112 float4 over 256 threads ----
        for (int idx = t; idx < 4 * PW * 8; idx += 256) {
            const int q8 = idx & 7, pp = idx >> 3;
            const int py = pp / PW, px = pp - py * PW;
            const int iy = y0 - 1 + py, ix = px - 1;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                v = *reinterpret_cast<const f32x4*>(xin + ((size_t)iy * W + ix) * C + half * 32 + q8 * 4);
            u32x2 hi, mid, lo;
            split4(v, hi, mid, lo);
            char* d = patch + pp * PIX + (q8 >> 2) * 96 + (q8 & 3) * 8;   // chunk (q8 >> 2), k-quad (q8 & 3)
            *reinterpret_cast<u32x2*>(d) = hi;
            *reinterpret_cast<u32x2*>(d + 32) = mid;
            *reinterpret_cast<u32x2*>(d + 64) = lo;
        }
        // ---- 18 K steps: tap-major, 2 chunks per tap; weights register-staged one step ahead ----
        auto wload = [&](int ks) {
            const int tap = ks >> 1, ch = ks & 1;
            return *reinterpret_cast<const f32x4*>(w + ((size_t)brow * 9 + tap) * C + half * 32 + ch * 16 + bq * 4);
        };
        auto wstore = [&](int buf, const f32x4& v) {
            u32x2 hi, mid, lo;
            split4(v, hi, mid, lo);
            char* d = Bs + buf * BTILE + brow * BROW + bq * 8;
            *reinterpret_cast<u32x2*>(d) = hi;
            *reinterpret_cast<u32x2*>(d + 32) = mid;
            *reinterpret_cast<u32x2*>(d + 64) = lo;
        };
        f32x4 wr = wload(0);
        wstore(0, wr);
        __syncthreads();
#pragma unroll 2
        for (int ks = 0; ks < 18; ++ks) {
            const int buf = ks & 1;
            if (ks + 1 < 18) wr = wload(ks + 1);
            const int tap = ks >> 1, ch = ks & 1;
            const int ta = tap / 3, tb = tap - ta * 3;
            const char* ap = patch + a_lane + (ta * PW + tb) * PIX + ch * 96;
            const char* bp = Bs + buf * BTILE + b_lane;
            bf16x8 a[3], b[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                a[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ap + pl * 32));
                b[0][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bp + pl * 32));
                b[1][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bp + 32 * BROW + pl * 32));
            }
            constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (q < 5) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[q]], b[j][TB[q]], acc2[j], 0, 0, 0);
                    else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[q]], b[j][TB[q]], acc[j], 0, 0, 0);
                }
            if (ks + 1 < 18) wstore(buf ^ 1, wr);
            __syncthreads();
        }
    }
    // accumulator (pixel row e-mapped, column = lane r = output channel within the block)
    float* yo = y + ((size_t)n * H * W + (size_t)(y0 + yy) * W + x0) * CO;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int px = 8 * (e >> 2) + 4 * h + (e & 3);
            yo[(size_t)px * CO + 32 * j + r] = acc[j][e] + acc2[j][e];
        }
}


// Variant 2: weights PRE-SPLIT once (host here; a per-update kernel in the engine) into the MFMA operand layout
//   ws[kstep = (half, tap, chunk)][plane][32-channel block][lane (row, k half)][8 bf16]
// and loaded by every lane STRAIGHT INTO REGISTERS (6 x 16 B per K step, one step ahead): no weight tile in LDS, no split
// arithmetic and NO BARRIER inside the 18 K steps of a patch - the waves run free between the two patch barriers.
__global__ __launch_bounds__(256, 2) void conv3x3_patch_bf16x3_wreg(const float* __restrict__ x, const unsigned short* __restrict__ ws,
                                                                    float* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) char patch[PATCH];
    const int t = threadIdx.x, lane = t & 63, r = lane & 31, h = lane >> 5, wv = t >> 6;
    const int tile = blockIdx.x;
    const int n = tile >> 5, y0 = (tile & 31) * 2;
    const float* xin = x + (size_t)n * H * W * C;
    f32x16 acc[2], acc2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[j][e] = 0.f; acc2[j][e] = 0.f; }
    const int yy = wv >> 1, x0 = (wv & 1) * 32;
    const int a_lane = ((yy * PW) + (x0 + r)) * PIX + h * 16;
    // ws[kstep][plane][block j][lane (r, h)][16 B]: one load instruction of a wave reads 1 KB of contiguous memory
    const char* wb = reinterpret_cast<const char*>(ws) + (r * 2 + h) * 16;
    u32x4 bn[2][3], bc[2][3];
    auto bload = [&](int kidx, u32x4 (&b)[2][3]) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            b[0][pl] = *reinterpret_cast<const u32x4*>(wb + ((kidx * 3 + pl) * 2 + 0) * 1024);
            b[1][pl] = *reinterpret_cast<const u32x4*>(wb + ((kidx * 3 + pl) * 2 + 1) * 1024);
        }
    };
    bload(0, bn);
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        for (int idx = t; idx < 4 * PW * 8; idx += 256) {
            const int q8 = idx & 7, pp = idx >> 3;
            const int py = pp / PW, px = pp - py * PW;
            const int iy = y0 - 1 + py, ix = px - 1;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                v = *reinterpret_cast<const f32x4*>(xin + ((size_t)iy * W + ix) * C + half * 32 + q8 * 4);
            u32x2 hi, mid, lo;
            split4(v, hi, mid, lo);
            char* d = patch + pp * PIX + (q8 >> 2) * 96 + (q8 & 3) * 8;
            *reinterpret_cast<u32x2*>(d) = hi;
            *reinterpret_cast<u32x2*>(d + 32) = mid;
            *reinterpret_cast<u32x2*>(d + 64) = lo;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bc[j][pl] = bn[j][pl];
            const int knext = half * 18 + ks + 1;
            if (knext < 36) bload(knext, bn);
            const int tap = ks >> 1, ch = ks & 1;
            const int ta = tap / 3, tb = tap - ta * 3;
            const char* ap = patch + a_lane + (ta * PW + tb) * PIX + ch * 96;
            bf16x8 a[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) a[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ap + pl * 32));
            constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bf16x8 bv = __builtin_bit_cast(bf16x8, bc[j][TB[q]]);
                    if (q < 5) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[q]], bv, acc2[j], 0, 0, 0);
                    else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[q]], bv, acc[j], 0, 0, 0);
                }
        }
    }
    float* yo = y + ((size_t)n * H * W + (size_t)(y0 + yy) * W + x0) * CO;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int px = 8 * (e >> 2) + 4 * h + (e & 3);
            yo[(size_t)px * CO + 32 * j + r] = acc[j][e] + acc2[j][e];
        }
}

// Variant 3: variant 2 + the patch load SOFTWARE-PIPELINED under the MFMAs.  Persistent workgroups (one per CU), two patch buffers:
// while the 18 K steps of stage s (= tile, 32-channel half) run out of one buffer, the 9 float4 per thread of stage s + 1 are in
// flight from global memory and are split and written into the other buffer one piece every second K step; one barrier per stage.
template <int D>
__global__ __launch_bounds__(256, 1) void conv3x3_patch_bf16x3_pipe(const float* __restrict__ x, const unsigned short* __restrict__ ws,
                                                                    float* __restrict__ y, int ntiles) {
    __shared__ __attribute__((aligned(16))) char patch2[2 * PATCH];
    const int t = threadIdx.x, lane = t & 63, r = lane & 31, h = lane >> 5, wv = t >> 6;
    f32x16 acc[2], acc2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[j][e] = 0.f; acc2[j][e] = 0.f; }
    const int yy = wv >> 1, x0 = (wv & 1) * 32;
    const int a_lane = ((yy * PW) + (x0 + r)) * PIX + h * 16;
    const char* wb = reinterpret_cast<const char*>(ws) + (r * 2 + h) * 16;
    // per-thread constants of the 9 patch pieces
    int p_lds[9], p_py[9], p_goff[9];
    bool p_ok[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int idx = t + 256 * i;
        const int q8 = idx & 7, pp = idx >> 3;
        const int py = pp / PW, px = pp - py * PW;
        p_ok[i] = idx < 4 * PW * 8 && (unsigned)(px - 1) < (unsigned)W;
        p_py[i] = py - 1;
        p_lds[i] = idx < 4 * PW * 8 ? pp * PIX + (q8 >> 2) * 96 + (q8 & 3) * 8 : -1;
        p_goff[i] = ((py - 1) * W + (px - 1)) * C + q8 * 4;
    }
    f32x4 pv[9];
    auto issue_loads = [&](int stage) {
        const int tile = blockIdx.x + (stage >> 1) * gridDim.x, half = stage & 1;
        const int n = tile >> 5, y0 = (tile & 31) * 2;
        const float* base = x + ((size_t)n * H * W + (size_t)y0 * W) * C + half * 32;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const bool ok = p_ok[i] && (unsigned)(y0 + p_py[i]) < (unsigned)H;
            pv[i] = ok ? *reinterpret_cast<const f32x4*>(base + p_goff[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_piece = [&](int i, int bufsel) {
        if (p_lds[i] < 0) return;
        u32x2 hi, mid, lo;
        split4(pv[i], hi, mid, lo);
        char* d = patch2 + bufsel * PATCH + p_lds[i];
        *reinterpret_cast<u32x2*>(d) = hi;
        *reinterpret_cast<u32x2*>(d + 32) = mid;
        *reinterpret_cast<u32x2*>(d + 64) = lo;
    };
    u32x4 ring[D][2][3];   // weight operands of the next D K steps (D = 1: one step ahead); 18 % D == 0
    auto bload = [&](int kidx, u32x4 (&b)[2][3]) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            b[0][pl] = *reinterpret_cast<const u32x4*>(wb + ((kidx * 3 + pl) * 2 + 0) * 1024);
            b[1][pl] = *reinterpret_cast<const u32x4*>(wb + ((kidx * 3 + pl) * 2 + 1) * 1024);
        }
    };
    const int my_tiles = (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const int nstages = 2 * my_tiles;
    if (nstages == 0) return;
    issue_loads(0);
#pragma unroll
    for (int i = 0; i < 9; ++i) store_piece(i, 0);
#pragma unroll
    for (int d = 0; d < D; ++d) bload(d, ring[d]);
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
        const int cur = s & 1, half = s & 1;
        const bool more = s + 1 < nstages;
        if (more) issue_loads(s + 1);
        const char* pbase = patch2 + cur * PATCH + a_lane;
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) {
            u32x4 bc[2][3];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bc[j][pl] = ring[ks % D][j][pl];
            int knext = half * 18 + ks + D;
            if (knext >= 36) knext -= 36;         // the next stage starts over at half 0
            bload(knext, ring[ks % D]);
            const int tap = ks >> 1, ch = ks & 1;
            const int ta = tap / 3, tb = tap - ta * 3;
            const char* ap = pbase + (ta * PW + tb) * PIX + ch * 96;
            bf16x8 a[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) a[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ap + pl * 32));
            constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bf16x8 bv = __builtin_bit_cast(bf16x8, bc[j][TB[q]]);
                    if (q < 5) acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[q]], bv, acc2[j], 0, 0, 0);
                    else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[q]], bv, acc[j], 0, 0, 0);
                }
            if (more && (ks & 1) == 0) store_piece(ks >> 1, cur ^ 1);   // one piece of the next patch every second K step
        }
        if (half == 1) {
            const int tile = blockIdx.x + (s >> 1) * gridDim.x;
            const int n = tile >> 5, y0 = (tile & 31) * 2;
            float* yo = y + ((size_t)n * H * W + (size_t)(y0 + yy) * W + x0) * CO;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int px = 8 * (e >> 2) + 4 * h + (e & 3);
                    yo[(size_t)px * CO + 32 * j + r] = acc[j][e] + acc2[j][e];
                    acc[j][e] = 0.f;
                    acc2[j][e] = 0.f;
                }
        }
        __syncthreads();
    }
}

static void host_split(float v, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
    unsigned b; memcpy(&b, &v, 4);
    hi = b >> 16;
    unsigned hb = b & 0xFFFF0000u; float hf; memcpy(&hf, &hb, 4);
    float r1 = v - hf; unsigned b1; memcpy(&b1, &r1, 4);
    mid = b1 >> 16;
    unsigned mb = b1 & 0xFFFF0000u; float mf; memcpy(&mf, &mb, 4);
    float r2 = r1 - mf; unsigned b2; memcpy(&b2, &r2, 4);
    lo = b2 >> 16;
}

int main() {
    const int N = 320;
    const size_t nx = (size_t)N * H * W * C, nw = (size_t)CO * 9 * C, ny = (size_t)N * H * W * CO;
    std::vector<float> hx(nx), hw(nw);
    unsigned s = 99;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 9) & 0x7FFFFF) / 4194304.0f - 1.0f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd() * 0.05f;
    float *x, *w, *y;
    hipMalloc(&x, nx * 4); hipMalloc(&w, nw * 4); hipMalloc(&y, ny * 4);
    hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
    const dim3 grid(N * 32);
    hipLaunchKernelGGL(conv3x3_patch_bf16x3, grid, dim3(256), 0, 0, x, w, y);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(conv3x3_patch_bf16x3, grid, dim3(256), 0, 0, x, w, y);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    std::vector<float> hy(ny);
    hipMemcpy(hy.data(), y, ny * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int q = 0; q < 2000; ++q) {
        s = s * 1664525u + 1013904223u; const int n = (s >> 8) % N;
        s = s * 1664525u + 1013904223u; const int oy = (s >> 8) % H;
        s = s * 1664525u + 1013904223u; const int ox = (s >> 8) % W;
        s = s * 1664525u + 1013904223u; const int co = (s >> 8) % CO;
        double ref = 0, mag = 0;
        for (int ta = 0; ta < 3; ++ta)
            for (int tb = 0; tb < 3; ++tb) {
                const int iy = oy + ta - 1, ix = ox + tb - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                for (int c = 0; c < C; ++c) {
                    const double p = (double)hx[(((size_t)n * H + iy) * W + ix) * C + c] * hw[((size_t)co * 9 + ta * 3 + tb) * C + c];
                    ref += p; mag += fabs(p);
                }
            }
        worst = fmax(worst, fabs(hy[(((size_t)n * H + oy) * W + ox) * CO + co] - ref) / mag);
    }
    const double fl = 2.0 * N * H * W * CO * C * 9;
    printf("conv 64->64 3x3 @64x64 x%d images, input patch in LDS, bf16x3: %.3f ms  %.1f TFLOP/s  max err / sum|terms| %.2e\n", N, ms, fl / ms / 1e9, worst);
    {   // variant 2: pre-split weights straight to registers
        std::vector<unsigned short> hws((size_t)CO * 36 * 48);
        for (int co = 0; co < CO; ++co)
            for (int half = 0; half < 2; ++half)
                for (int tap = 0; tap < 9; ++tap)
                    for (int ch = 0; ch < 2; ++ch)
                        for (int kk = 0; kk < 16; ++kk) {
                            const int kidx = half * 18 + tap * 2 + ch;
                            unsigned short a, b, c;
                            host_split(hw[((size_t)co * 9 + tap) * C + half * 32 + ch * 16 + kk], a, b, c);
                            const int j = co >> 5, rr = co & 31, hh = kk >> 3, i8 = kk & 7;
                            unsigned short pls[3] = {a, b, c};
                            for (int pl = 0; pl < 3; ++pl)
                                hws[((((size_t)kidx * 3 + pl) * 2 + j) * 1024 + (rr * 2 + hh) * 16) / 2 + i8] = pls[pl];
                        }
        unsigned short* ws;
        hipMalloc(&ws, hws.size() * 2);
        hipMemcpy(ws, hws.data(), hws.size() * 2, hipMemcpyHostToDevice);
        hipMemset(y, 0, ny * 4);
        hipLaunchKernelGGL(conv3x3_patch_bf16x3_wreg, grid, dim3(256), 0, 0, x, ws, y);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(conv3x3_patch_bf16x3_wreg, grid, dim3(256), 0, 0, x, ws, y);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms2;
        hipEventElapsedTime(&ms2, e0, e1);
        ms2 /= reps;
        std::vector<float> hy2(ny);
        hipMemcpy(hy2.data(), y, ny * 4, hipMemcpyDeviceToHost);
        double diff = 0;
        for (size_t i = 0; i < ny; i += 997) diff = fmax(diff, fabs((double)hy2[i] - hy[i]));
        printf("  + pre-split weights straight to registers, no barrier in the K loop: %.3f ms  %.1f TFLOP/s  max |diff to variant 1| %.2e\n",
               ms2, fl / ms2 / 1e9, diff);
        for (int depth = 1; depth <= 6; depth = depth == 3 ? 6 : depth + 1) {
            const int wgs = 256;
            auto launch = [&]() {
                if (depth == 1) hipLaunchKernelGGL(conv3x3_patch_bf16x3_pipe<1>, dim3(wgs), dim3(256), 0, 0, x, ws, y, N * 32);
                else if (depth == 2) hipLaunchKernelGGL(conv3x3_patch_bf16x3_pipe<2>, dim3(wgs), dim3(256), 0, 0, x, ws, y, N * 32);
                else if (depth == 3) hipLaunchKernelGGL(conv3x3_patch_bf16x3_pipe<3>, dim3(wgs), dim3(256), 0, 0, x, ws, y, N * 32);
                else hipLaunchKernelGGL(conv3x3_patch_bf16x3_pipe<6>, dim3(wgs), dim3(256), 0, 0, x, ws, y, N * 32);
            };
            hipMemset(y, 0, ny * 4);
            launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < reps; ++i) launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms3;
            hipEventElapsedTime(&ms3, e0, e1);
            ms3 /= reps;
            hipMemcpy(hy2.data(), y, ny * 4, hipMemcpyDeviceToHost);
            double diff3 = 0;
            for (size_t i = 0; i < ny; i += 997) diff3 = fmax(diff3, fabs((double)hy2[i] - hy[i]));
            printf("  + patch load pipelined under the MFMAs, persistent workgroups, weights %d K step(s) ahead: %.3f ms  %.1f TFLOP/s  max |diff to variant 1| %.2e\n",
                   depth, ms3, fl / ms3 / 1e9, diff3);
        }
    }
    printf("(engine on this layer: fp32 MFMA 1.01 ms 96 TFLOP/s; bf16x3 with per-tap loads 0.76 ms 127 TFLOP/s)\n");
    return 0;
}

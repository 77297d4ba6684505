// fp32 implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), NHWC.
//
//   forward / dgrad :  Y[m][n] = sum_k A[m][k] * B[k][n]
//        m = output pixel (n_img, oh, ow), k = (tap, channel of the gathered tensor), n = output channel
//        A is gathered on the fly from the NHWC activation (im2col never materialised); the gather
//        applies the fused prologue  nearest-up2( leaky_relu(x) )  and zero padding.
//        B comes straight from the weight tensor [Cout][KH][KW][Cin]:
//          forward: k-contiguous rows  (BMODE 0, LDS image [n][k], operands by ds_read_b128)
//          dgrad  : the same tensor read "k-major" with the taps flipped (BMODE 1, LDS image [k][n])
//   wgrad : C[co][j] = sum_m dY[m][co] * A[m][j],  j = (tap, ci), split over pixel slices into slabs.
//
// Tiles: 256 threads = 4 waves, each wave owns a (32*TM) x (32*TN) block of 32x32 MFMA accumulators;
// K step 16; LDS double-buffered, register-staged (global -> VGPR -> LDS) so that the loads of step
// k+1 are in flight under the MFMAs of step k; one barrier per K step.
// k-contiguous LDS rows are padded 16 -> 20 floats: every 16-lane group of a ds_read_b128 then
// touches 16 distinct 16-byte bank slots (conflict-free, MI355X_MICROARCH LDS table).
#include <stdlib.h>

#include "common.h"

#define BK 16  // wgrad pixel step; also the K granularity the fast paths require (channels % 16 == 0)

struct ConvP {
    const float* x;
    const float* w;
    const float* bias;
    const float* sigma;
    const float* res;
    const float* mask_x;
    float* y;
    int N, H, W, logH, logW;
    int Ca, Cb;
    int KH, pad, ups, T;
    int Cin_w;
    int M;
    int Ktot;
    float pre_slope, mask_slope;
    int ksplit;   // > 1: grid.z K-slices, partial results combined with float atomics into a pre-zeroed y
    int kper;     // K-steps per slice
};

// GENF bit 0: generic K (channel count of the gathered tensor not a multiple of 16, or unaligned base)
// GENF bit 1: (BMODE 1 only) scalar loads of the k-major weight tile (output channels not a multiple of 4)
template <int BM, int BN, int TM, int TN, int BMODE, int GENF, int KB>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvP p) {
    constexpr bool GEN = (GENF & 1) != 0;
    constexpr bool BSCALAR = (GENF & 2) != 0;
    constexpr int WAVES_N = BN / (32 * TN);
    constexpr int WAVES_M = BM / (32 * TM);
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    constexpr int LDK = KB + 4;                    // padded k-row (conflict-free ds_read_b128 for KB = 16 and 32)
    constexpr int QPR = KB / 4;                    // float4 per k-row
    constexpr int RP = 256 / QPR;                  // tile rows filled per pass
    constexpr int A_ROWS = BM / RP;
    constexpr int B_ROWS = (BN + RP - 1) / RP;     // BMODE 0
    constexpr int B_PER = KB * BN / 256;           // BMODE 1, scalar
    constexpr int B_U = BN / 4;                    // BMODE 1, vector: float4 units per k-row
    constexpr int B_RSTEP = 256 / B_U;
    constexpr int B_PER4 = (KB + B_RSTEP - 1) / B_RSTEP;
    constexpr int A_SZ = BM * LDK;
    constexpr int B_SZ = (BMODE == 0) ? BN * LDK : KB * BN;
    __shared__ __attribute__((aligned(16))) float lds[2 * A_SZ + 2 * B_SZ];
    float* As = lds;
    float* Bs = lds + 2 * A_SZ;

    const int t = threadIdx.x;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int arow = t / QPR, aq = (t % QPR) * 4;
    const int Hs = p.H >> p.ups, Ws = p.W >> p.ups;

    int a_oh[A_ROWS], a_ow[A_ROWS];
    long long a_base[A_ROWS];
    bool a_ok[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        const int m = m0 + arow + RP * i;
        a_ok[i] = m < p.M;
        const int n = m >> (p.logH + p.logW);
        a_oh[i] = (m >> p.logW) & (p.H - 1);
        a_ow[i] = m & (p.W - 1);
        a_base[i] = (long long)n * Hs * Ws * p.Ca;
    }

    f32x4 ra[A_ROWS];
    f32x4 rb0[B_ROWS];
    float rb1[B_PER];
    f32x4 rb4[B_PER4];

    auto load_tiles = [&](int k0) {
        // ---- A: gathered activations ----
        if constexpr (!GEN) {
            const int tap = k0 / p.Ca;
            const int c0 = k0 - tap * p.Ca;
            const int kh = tap / p.KH, kw = tap - kh * p.KH;
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) {
                const int ih = a_oh[i] + kh - p.pad, iw = a_ow[i] + kw - p.pad;
                const bool v = a_ok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                f32x4 val = {0.f, 0.f, 0.f, 0.f};
                if (v) {
                    const float* src = p.x + a_base[i] + (long long)((ih >> p.ups) * Ws + (iw >> p.ups)) * p.Ca + c0 + aq;
                    val = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[e] = lrelu_f(val[e], p.pre_slope);
                }
                ra[i] = val;
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kf = k0 + aq + e;
                    const int tap = kf / p.Ca;
                    const int c = kf - tap * p.Ca;
                    const int kh = tap / p.KH, kw = tap - kh * p.KH;
                    const int ih = a_oh[i] + kh - p.pad, iw = a_ow[i] + kw - p.pad;
                    const bool v = a_ok[i] && kf < p.Ktot && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                    float val = 0.f;
                    if (v) val = lrelu_f(p.x[a_base[i] + (long long)((ih >> p.ups) * Ws + (iw >> p.ups)) * p.Ca + c], p.pre_slope);
                    ra[i][e] = val;
                }
            }
        }
        // ---- B: weights ----
        if constexpr (BMODE == 0) {
#pragma unroll
            for (int i = 0; i < B_ROWS; ++i) {
                const int row = arow + RP * i;
                const int co = n0 + row;
                f32x4 val = {0.f, 0.f, 0.f, 0.f};
                if (row < BN && co < p.Cb) {
                    if constexpr (!GEN) {
                        val = *reinterpret_cast<const f32x4*>(p.w + (long long)co * p.Ktot + k0 + aq);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int kf = k0 + aq + e;
                            if (kf < p.Ktot) val[e] = p.w[(long long)co * p.Ktot + kf];
                        }
                    }
                }
                rb0[i] = val;
            }
        } else {
            int tap_u = 0, c0_u = 0;
            if constexpr (!GEN) {
                tap_u = k0 / p.Ca;
                c0_u = k0 - tap_u * p.Ca;
            }
            if constexpr (BSCALAR) {
#pragma unroll
                for (int i = 0; i < B_PER; ++i) {
                    const int idx = t + 256 * i;
                    const int krow = idx / BN, col = idx % BN;
                    int tap, ca;
                    bool v = true;
                    if constexpr (!GEN) {
                        tap = tap_u;
                        ca = c0_u + krow;
                    } else {
                        const int kf = k0 + krow;
                        tap = kf / p.Ca;
                        ca = kf - tap * p.Ca;
                        v = kf < p.Ktot;
                    }
                    const int ci = n0 + col;
                    v = v && ci < p.Cb;
                    rb1[i] = v ? p.w[((long long)ca * p.T + (p.T - 1 - tap)) * p.Cin_w + ci] : 0.f;
                }
            } else {
#pragma unroll
                for (int i = 0; i < B_PER4; ++i) {
                    const int krow = t / B_U + i * B_RSTEP, col = (t % B_U) * 4;
                    int tap, ca;
                    bool v = krow < KB;
                    if constexpr (!GEN) {
                        tap = tap_u;
                        ca = c0_u + krow;
                    } else {
                        const int kf = k0 + krow;
                        tap = kf / p.Ca;
                        ca = kf - tap * p.Ca;
                        v = v && kf < p.Ktot;
                    }
                    const int ci = n0 + col;
                    v = v && ci < p.Cb;  // Cb % 4 == 0 on this path: the whole quad is in range
                    f32x4 val = {0.f, 0.f, 0.f, 0.f};
                    if (v) val = *reinterpret_cast<const f32x4*>(p.w + ((long long)ca * p.T + (p.T - 1 - tap)) * p.Cin_w + ci);
                    rb4[i] = val;
                }
            }
        }
    };

    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i)
            *reinterpret_cast<f32x4*>(&As[buf * A_SZ + (arow + RP * i) * LDK + aq]) = ra[i];
        if constexpr (BMODE == 0) {
#pragma unroll
            for (int i = 0; i < B_ROWS; ++i) {
                const int row = arow + RP * i;
                if (row < BN) *reinterpret_cast<f32x4*>(&Bs[buf * B_SZ + row * LDK + aq]) = rb0[i];
            }
        } else {
            if constexpr (BSCALAR) {
#pragma unroll
                for (int i = 0; i < B_PER; ++i) Bs[buf * B_SZ + t + 256 * i] = rb1[i];  // [krow][col] row-major == idx
            } else {
#pragma unroll
                for (int i = 0; i < B_PER4; ++i) {
                    const int krow = t / B_U + i * B_RSTEP, col = (t % B_U) * 4;
                    if (krow < KB) *reinterpret_cast<f32x4*>(&Bs[buf * B_SZ + krow * BN + col]) = rb4[i];
                }
            }
        }
    };

    const int lane = t & 63, r = lane & 31, h = lane >> 5, wv = t >> 6;
    const int wm0 = (wv / WAVES_N) * 32 * TM, wn0 = (wv % WAVES_N) * 32 * TN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk_all = (p.Ktot + KB - 1) / KB;
    const int ks0 = blockIdx.z * p.kper;
    const int nk = min(nk_all, ks0 + p.kper);
    load_tiles(ks0 * KB);
    store_tiles(ks0 & 1);
    __syncthreads();
    for (int ks = ks0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) load_tiles((ks + 1) * KB);
        const float* Ab = As + buf * A_SZ;
        const float* Bb = Bs + buf * B_SZ;
#pragma unroll
        for (int kk = 0; kk < KB / 8; ++kk) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(&Ab[(wm0 + 32 * i + r) * LDK + 8 * kk + 4 * h]);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (BMODE == 0) {
                    b[j] = *reinterpret_cast<const f32x4*>(&Bb[(wn0 + 32 * j + r) * LDK + 8 * kk + 4 * h]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[j][e] = Bb[(8 * kk + 4 * h + e) * BN + wn0 + 32 * j + r];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
        if (ks + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: scale by 1/sigma, bias, residual, activation mask ----
    const float inv_sigma = p.sigma ? 1.0f / p.sigma[0] : 1.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int co = n0 + wn0 + 32 * j + r;
            if (co >= p.Cb) continue;
            const bool first = blockIdx.z == 0;
            const float bv = (p.bias && first) ? p.bias[co] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m >= p.M) continue;
                const long long o = (long long)m * p.Cb + co;
                float v = acc[i][j][e] * inv_sigma + bv;
                if (p.res && first) v += p.res[o];
                if (p.mask_x) v *= (p.mask_x[o] > 0.f ? 1.0f : p.mask_slope);
                if (p.ksplit > 1) atomicAdd(&p.y[o], v);
                else p.y[o] = v;
            }
        }
}

// -------------------------------------------------------------------------------------------------
// wgrad
// -------------------------------------------------------------------------------------------------
struct WgP {
    const float* dy;
    const float* x;
    float* slabs;
    float* bias_slabs;
    int N, H, W, logH, logW;
    int Cin, Cout;
    int KH, pad, ups, T;
    int M;
    int Kcols;
    int mper;
    float pre_slope;
};

// VEC = 4: dY rows and gathered x rows are fetched as float4 (Cout % 4 == 0, Cin % 4 == 0, 16-byte aligned
// bases); VEC = 1: scalar fallback (3- and 6-channel image layers, 1-channel Omniglot).
// Workgroups of the first column tile also produce the bias gradient sum_m dY[m][co] of their pixel slice from
// the dY values they stream anyway (bias_slabs[slice][Cout]).
template <int BM, int BN, int TM, int TN, int VEC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgP p) {
    constexpr int WAVES_N = BN / (32 * TN);
    constexpr int WAVES_M = BM / (32 * TM);
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    constexpr int AU = BM / VEC, BU = BN / VEC;                 // load units per tile row
    constexpr int A_RSTEP = 256 / AU, B_RSTEP = 256 / BU;       // tile rows covered per pass
    constexpr int A_PER = (BK + A_RSTEP - 1) / A_RSTEP, B_PER = (BK + B_RSTEP - 1) / B_RSTEP;
    __shared__ __attribute__((aligned(16))) float As[2][BK * BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * BN];

    const int t = threadIdx.x;
    const int j0 = blockIdx.x * BN, co0 = blockIdx.y * BM;
    const int mbeg = blockIdx.z * p.mper;
    const int mend = min(p.M, mbeg + p.mper);
    const int Hs = p.H >> p.ups, Ws = p.W >> p.ups;

    const int ac = (t % AU) * VEC, ak = t / AU;
    const bool a_cok = (co0 + ac) < p.Cout;
    const int bc = (t % BU) * VEC, bk = t / BU;
    const int j = j0 + bc;
    const bool b_jok = j < p.Kcols;
    const int tap = b_jok ? j / p.Cin : 0;
    const int ci = b_jok ? j - tap * p.Cin : 0;
    const int kh = tap / p.KH, kw = tap - kh * p.KH;
    const int dh = kh - p.pad, dw = kw - p.pad;
    const bool do_bias = p.bias_slabs != nullptr && blockIdx.x == 0;

    float ra[A_PER][VEC], rb[B_PER][VEC];
    float bsum[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) bsum[e] = 0.f;

    auto load_tiles = [&](int mb) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int row = ak + i * A_RSTEP;
            const int m = mb + row;
            const bool v = a_cok && row < BK && m < mend;
            if constexpr (VEC == 4) {
                f32x4 val = {0.f, 0.f, 0.f, 0.f};
                if (v) val = *reinterpret_cast<const f32x4*>(p.dy + (long long)m * p.Cout + co0 + ac);
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[i][e] = val[e];
            } else {
                ra[i][0] = v ? p.dy[(long long)m * p.Cout + co0 + ac] : 0.f;
            }
            if (do_bias) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) bsum[e] += ra[i][e];
            }
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int row = bk + i * B_RSTEP;
            const int m = mb + row;
            const int n = m >> (p.logH + p.logW);
            const int ih = ((m >> p.logW) & (p.H - 1)) + dh;
            const int iw = (m & (p.W - 1)) + dw;
            const bool v = b_jok && row < BK && m < mend && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            const long long off = ((long long)(n * Hs + (ih >> p.ups)) * Ws + (iw >> p.ups)) * p.Cin + ci;
            if constexpr (VEC == 4) {
                f32x4 val = {0.f, 0.f, 0.f, 0.f};
                if (v) val = *reinterpret_cast<const f32x4*>(p.x + off);
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[i][e] = lrelu_f(val[e], p.pre_slope);
            } else {
                rb[i][0] = v ? lrelu_f(p.x[off], p.pre_slope) : 0.f;
            }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int row = ak + i * A_RSTEP;
            if (row < BK) {
                if constexpr (VEC == 4) {
                    f32x4 val = {ra[i][0], ra[i][1], ra[i][2], ra[i][3]};
                    *reinterpret_cast<f32x4*>(&As[buf][row * BM + ac]) = val;
                } else {
                    As[buf][row * BM + ac] = ra[i][0];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int row = bk + i * B_RSTEP;
            if (row < BK) {
                if constexpr (VEC == 4) {
                    f32x4 val = {rb[i][0], rb[i][1], rb[i][2], rb[i][3]};
                    *reinterpret_cast<f32x4*>(&Bs[buf][row * BN + bc]) = val;
                } else {
                    Bs[buf][row * BN + bc] = rb[i][0];
                }
            }
        }
    };

    const int lane = t & 63, r = lane & 31, h = lane >> 5, wv = t >> 6;
    const int wm0 = (wv / WAVES_N) * 32 * TM, wn0 = (wv % WAVES_N) * 32 * TN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < TN; ++jj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;

    const int nk = (mend > mbeg) ? (mend - mbeg + BK - 1) / BK : 0;
    if (nk > 0) {
        load_tiles(mbeg);
        store_tiles(0);
    }
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) load_tiles(mbeg + (ks + 1) * BK);
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[buf][(2 * kp + h) * BM + wm0 + 32 * i + r];
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) b[jj] = Bs[buf][(2 * kp + h) * BN + wn0 + 32 * jj + r];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jj = 0; jj < TN; ++jj)
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[jj], acc[i][jj], 0, 0, 0);
        }
        if (ks + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    float* out = p.slabs + (long long)blockIdx.z * p.Cout * p.Kcols;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < TN; ++jj) {
            const int col = j0 + wn0 + 32 * jj + r;
            if (col >= p.Kcols) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + wm0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (co < p.Cout) out[(long long)co * p.Kcols + col] = acc[i][jj][e];
            }
        }

    if (do_bias) {  // block-uniform; As is free after the loop's last barrier
        float* red = &As[0][0];  // needs (256 / AU) * BM = 256 * VEC <= 2 * BK * BM floats
#pragma unroll
        for (int e = 0; e < VEC; ++e) red[ak * BM + ac + e] = bsum[e];
        __syncthreads();
        if (t < BM && co0 + t < p.Cout) {
            float sacc = 0.f;
            for (int rr = 0; rr < 256 / AU; ++rr) sacc += red[rr * BM + t];
            p.bias_slabs[(long long)blockIdx.z * p.Cout + co0 + t] = sacc;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
static int fill_common(const gim_conv_shape* s, int* logH, int* logW) {
    GIM_CHECK_ARG(s != nullptr, "conv: null shape");
    GIM_CHECK_ARG(s->N > 0 && s->Cin > 0 && s->Cout > 0, "conv: non-positive dims");
    GIM_CHECK_ARG(s->KH == 1 || s->KH == 3 || s->KH == 9 || (s->KH > 0 && (s->KH & 1)), "conv: KH must be odd");
    GIM_CHECK_ARG(s->ups == 0 || s->ups == 1, "conv: ups must be 0 or 1");
    *logH = ilog2_exact(s->H);
    *logW = ilog2_exact(s->W);
    GIM_CHECK_ARG(*logH >= 0 && *logW >= 0, "conv: H and W must be powers of two");
    GIM_CHECK_ARG(!s->ups || (s->H >= 2 && s->W >= 2), "conv: ups needs H, W >= 2");
    GIM_CHECK_ARG((long long)s->N * s->H * s->W < (1ll << 31), "conv: too many output pixels");
    return GIM_OK;
}

// K step 32 measured SLOWER on MI355X (104 vs 111 episodes/s: 73 KB of LDS per workgroup leaves 2 waves per
// SIMD instead of 3); kept as an opt-in for A/B runs.
static const bool g_force_kb16 = getenv("GIM_CONV_KB32") == nullptr;

// Small-M layers (4x4x512 maps, the decoder head, linears on <= 240 rows) have too few output tiles to fill
// 256 CUs and are bound by the latency of their long K loop: slice K over grid.z until there are ~2 workgroups
// per CU, each keeping >= 8 K-steps.
static int plan_ksplit(long long wgs, int nk) {
    if (wgs >= 192 || nk < 16) return 1;
    long long ks = (256 + wgs - 1) / wgs;
    if (ks > nk / 8) ks = nk / 8;
    if (ks > 32) ks = 32;
    return ks < 1 ? 1 : (int)ks;
}

template <int BM, int BN, int TM, int TN, int BMODE, int GEN, int KB>
static void launch_cfg_kb(ConvP p, hipStream_t st) {
    const int gx = (p.M + BM - 1) / BM, gy = (p.Cb + BN - 1) / BN;
    const int nk = (p.Ktot + KB - 1) / KB;
    p.ksplit = plan_ksplit((long long)gx * gy, nk);
    p.kper = (nk + p.ksplit - 1) / p.ksplit;
    p.ksplit = (nk + p.kper - 1) / p.kper;
    if (p.ksplit > 1) (void)hipMemsetAsync(p.y, 0, (size_t)p.M * p.Cb * sizeof(float), st);
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, TM, TN, BMODE, GEN, KB>), dim3(gx, gy, p.ksplit), dim3(256), 0, st, p);
}

// K step 32 halves the barriers and LDS round trips per FLOP; it needs the gathered channel count to be a
// multiple of 32 (and the vector paths); the large 128-row tiles are the ones that profit.
template <int BM, int BN, int TM, int TN, int BMODE, int GEN>
static void launch_cfg(const ConvP& p, hipStream_t st) {
    if constexpr ((GEN & 1) == 0 && BM == 128) {
        if (p.Ca % 32 == 0 && !g_force_kb16) {
            launch_cfg_kb<BM, BN, TM, TN, BMODE, GEN, 32>(p, st);
            return;
        }
    }
    launch_cfg_kb<BM, BN, TM, TN, BMODE, GEN, 16>(p, st);
}

template <int BMODE, int GEN>
static void launch_igemm(const ConvP& p, hipStream_t st) {
    // tile by shape (largest accumulator block the channel count fills); parallelism for small M comes from split-K
    const int M = p.M, Cb = p.Cb;
    if (Cb > 64) {
        if (M <= 64) launch_cfg<64, 128, 1, 2, BMODE, GEN>(p, st);
        else launch_cfg<128, 128, 2, 2, BMODE, GEN>(p, st);
    } else if (Cb > 32) {
        if (M <= 64) launch_cfg<64, 64, 1, 1, BMODE, GEN>(p, st);
        else launch_cfg<128, 64, 2, 1, BMODE, GEN>(p, st);
    } else {
        launch_cfg<128, 32, 1, 1, BMODE, GEN>(p, st);
    }
}

extern "C" int gim_conv2d_fwd(const float* x, const float* w, const float* bias, const float* sigma, const float* residual,
                              float* y, const gim_conv_shape* s, void* stream) {
    int logH, logW;
    int rc = fill_common(s, &logH, &logW);
    if (rc) return rc;
    GIM_CHECK_ARG(x && w && y, "conv fwd: null pointer");
    ConvP p;
    p.x = x; p.w = w; p.bias = bias; p.sigma = sigma; p.res = residual; p.mask_x = nullptr; p.y = y;
    p.N = s->N; p.H = s->H; p.W = s->W; p.logH = logH; p.logW = logW;
    p.Ca = s->Cin; p.Cb = s->Cout; p.KH = s->KH; p.pad = (s->KH - 1) / 2; p.ups = s->ups; p.T = s->KH * s->KH;
    p.Cin_w = s->Cin; p.M = s->N * s->H * s->W; p.Ktot = p.T * s->Cin;
    p.pre_slope = s->pre_slope; p.mask_slope = 1.f;
    const bool gen = (s->Cin % BK) != 0 || ((uintptr_t)x & 15) || ((uintptr_t)w & 15);
    if (gen) launch_igemm<0, 1>(p, (hipStream_t)stream);
    else launch_igemm<0, 0>(p, (hipStream_t)stream);
    return gim_check_launch("gim_conv2d_fwd");
}

extern "C" int gim_conv2d_dgrad(const float* dy, const float* w, const float* sigma, const float* mask_x, float* dx,
                                const gim_conv_shape* s, void* stream) {
    int logH, logW;
    int rc = fill_common(s, &logH, &logW);
    if (rc) return rc;
    GIM_CHECK_ARG(dy && w && dx, "conv dgrad: null pointer");
    GIM_CHECK_ARG(!(mask_x && s->ups), "conv dgrad: mask_x is only legal for ups == 0");
    ConvP p;
    p.x = dy; p.w = w; p.bias = nullptr; p.sigma = sigma; p.res = nullptr; p.mask_x = mask_x; p.y = dx;
    p.N = s->N; p.H = s->H; p.W = s->W; p.logH = logH; p.logW = logW;
    p.Ca = s->Cout; p.Cb = s->Cin; p.KH = s->KH; p.pad = (s->KH - 1) / 2; p.ups = 0; p.T = s->KH * s->KH;
    p.Cin_w = s->Cin; p.M = s->N * s->H * s->W; p.Ktot = p.T * s->Cout;
    p.pre_slope = 1.f; p.mask_slope = s->pre_slope;
    const bool gen = (s->Cout % BK) != 0 || ((uintptr_t)dy & 15);
    const bool bscalar = (s->Cin % 4) != 0 || ((uintptr_t)w & 15);
    hipStream_t st = (hipStream_t)stream;
    if (gen) { if (bscalar) launch_igemm<1, 3>(p, st); else launch_igemm<1, 1>(p, st); }
    else     { if (bscalar) launch_igemm<1, 2>(p, st); else launch_igemm<1, 0>(p, st); }
    return gim_check_launch("gim_conv2d_dgrad");
}

static void wgrad_plan(const gim_conv_shape* s, int* bm, int* bn, int* nslab, int* mper) {
    const int Kcols = s->KH * s->KH * s->Cin;
    const long long M = (long long)s->N * s->H * s->W;
    *bm = s->Cout > 64 ? 128 : (s->Cout > 32 ? 64 : 32);
    *bn = (*bm == 32) ? 128 : (Kcols > 64 ? 128 : 64);
    const long long tiles = (long long)((Kcols + *bn - 1) / *bn) * ((s->Cout + *bm - 1) / *bm);
    // about two workgroups per CU in total, and at least 32 K-steps (512 pixels) per workgroup so that the
    // slab write + later slab reduction stay small next to the MFMA work
    long long S = (512 + tiles - 1) / tiles;
    const long long maxS = (M + 511) / 512;
    if (S > maxS) S = maxS;
    if (S > 256) S = 256;
    if (S < 1) S = 1;
    long long mp = (M + S - 1) / S;
    mp = (mp + BK - 1) / BK * BK;
    *mper = (int)mp;
    *nslab = (int)((M + mp - 1) / mp);
}

extern "C" int gim_conv2d_wgrad_slabs(const gim_conv_shape* s) {
    int logH, logW;
    if (fill_common(s, &logH, &logW)) return GIM_E_BADARG;
    int bm, bn, ns, mper;
    wgrad_plan(s, &bm, &bn, &ns, &mper);
    return ns;
}

template <int VEC>
static void launch_wgrad(const WgP& p, int bm, int bn, dim3 g, hipStream_t st) {
    if (bm == 128 && bn == 128) hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 2, 2, VEC>), g, dim3(256), 0, st, p);
    else if (bm == 128 && bn == 64) hipLaunchKernelGGL((conv_wgrad_kernel<128, 64, 2, 1, VEC>), g, dim3(256), 0, st, p);
    else if (bm == 64 && bn == 128) hipLaunchKernelGGL((conv_wgrad_kernel<64, 128, 1, 2, VEC>), g, dim3(256), 0, st, p);
    else if (bm == 64 && bn == 64) hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, 1, 1, VEC>), g, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<32, 128, 1, 1, VEC>), g, dim3(256), 0, st, p);
}

extern "C" int gim_conv2d_wgrad(const float* dy, const float* x, float* slabs, float* bias_slabs, int n_slabs,
                                const gim_conv_shape* s, void* stream) {
    int logH, logW;
    int rc = fill_common(s, &logH, &logW);
    if (rc) return rc;
    GIM_CHECK_ARG(dy && x && slabs, "conv wgrad: null pointer");
    int bm, bn, ns, mper;
    wgrad_plan(s, &bm, &bn, &ns, &mper);
    GIM_CHECK_ARG(n_slabs == ns, "conv wgrad: n_slabs must equal gim_conv2d_wgrad_slabs(shape)");
    WgP p;
    p.dy = dy; p.x = x; p.slabs = slabs; p.bias_slabs = bias_slabs;
    p.N = s->N; p.H = s->H; p.W = s->W; p.logH = logH; p.logW = logW;
    p.Cin = s->Cin; p.Cout = s->Cout; p.KH = s->KH; p.pad = (s->KH - 1) / 2; p.ups = s->ups; p.T = s->KH * s->KH;
    p.M = s->N * s->H * s->W; p.Kcols = p.T * s->Cin; p.mper = mper; p.pre_slope = s->pre_slope;
    dim3 g((p.Kcols + bn - 1) / bn, (s->Cout + bm - 1) / bm, ns);
    const bool vec = (s->Cin % 4 == 0) && (s->Cout % 4 == 0) && !(((uintptr_t)dy | (uintptr_t)x) & 15);
    if (vec) launch_wgrad<4>(p, bm, bn, g, (hipStream_t)stream);
    else launch_wgrad<1>(p, bm, bn, g, (hipStream_t)stream);
    return gim_check_launch("gim_conv2d_wgrad");
}

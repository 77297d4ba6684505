// fp32 implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), NHWC.
//
//   forward / dgrad :  Y[m][n] = sum_k A[m][k] * B[k][n]
//        m = output pixel (n_img, oy, ox), k = (tap, channel of the gathered tensor), n = output channel
//        A is gathered on the fly from the NHWC activation (im2col never materialised); the gather
//        applies the fused prologue  nearest-up2( leaky_relu(x) )  and zero padding.
//        B comes straight from the weight tensor [Cout][KF][KF][Cin]:
//          forward: k-contiguous rows  (BMODE 0, LDS image [n][k], operands by ds_read_b128)
//          dgrad  : the same tensor read "k-major" with the taps re-mapped (BMODE 1, LDS image [k][n])
//   wgrad : C[co][j] = sum_m dY[m][co] * A[m][j],  j = (tap, ci), split over pixel slices into slabs.
//
// One gather/output geometry (struct Geo) covers four families, so that pooling and upsampling never cost
// convolution FLOPs (FOLDS):
//   plain      : iy = oy + ta - pad
//   S2         : iy = 2*oy + ta - pad over the (K+1)^2-tap folded weights F = sum of 2x2-shifted copies of W
//                = avgpool2(conv_KxK(x)) forward [16/36 of the FLOPs for K=3, 100/324 for K=9], and the dgrad
//                and wgrad of the sub-pixel form below
//   PC kind 0  : the 4 output-parity classes of conv_KxK(nearest_up2(x)): each class is a ((K+1)/2)^2-tap
//                convolution of the LOW-resolution x with a strided subset of F's taps
//   PC kind 1  : the 4 input-parity classes of the dgrad of S2
//
// Tiles: 256 threads = 4 waves, each wave owns a (32*TM) x (32*TN) block of 32x32 MFMA accumulators;
// K step 16; LDS double-buffered, register-staged (global -> VGPR -> LDS) so that the loads of step
// k+1 are in flight under the MFMAs of step k; one barrier per K step.
// k-contiguous LDS rows are padded 16 -> 20 floats: every 16-lane group of a ds_read_b128 then
// touches 16 distinct 16-byte bank slots (conflict-free, MI355X_MICROARCH LDS table).
#include <stdlib.h>

#include "common.h"
#include <type_traits>

#define PM_MAX_PIXELS 16    // position-major rows (Geo.pm) for logical maps up to 4 x 4 (larger maps: measured slower, profiles/r04_p_*) ...
#define PM_MIN_IMAGES 32    // ... of at least this many images (a 64-row tile then holds <= 3 pixel positions)
#define BK 16  // wgrad pixel step; also the K granularity the fast paths require (channels % 16 == 0)

struct Geo {
    int N, H, W, logH, logW;  // logical output grid (per parity class in PC mode)
    int Hin, Win;             // stored size of the gathered tensor (before the on-the-fly nearest upsample)
    int ups;                  // gathered coordinates are >> ups
    int s_in;                 // gather stride (1 or 2)
    int s_in_x;               // ... along x (= s_in except in the x-folded dgrad: stride J along x, 1 along y)
    int off_y, off_x;         // iy = oy * s_in + ta + off_y
    int Th, Tw;               // taps per dimension
    int KF;                   // taps per dimension of the weight tensor in memory
    int KFw;                  // ... along x (= KF except in the x-folded dgrad: K + J - 1)
    int wa_base, wa_step, wb_base, wb_step;  // weight tap (a, b) = (wa_base + wa_step * ta, wb_base + wb_step * tb)
    int os, py, px;           // output pixel = (oy * os + py, ox * os + px) in an (H*os) x (W*os) image
    int pc, pc_kind, pc_K;    // PC mode: blockIdx.z & 3 = class; class parameters derived in the kernel
    int pm;                   // POSITION-MAJOR rows (maps of <= 16 logical pixels, conv_igemm_kernel fast path): GEMM row m = slot * N + image,
                              // pixel = nibble `slot` of pm_perm - the rows of a tile share (at most a few) pixel positions, so the taps that
                              // fall into the zero padding for ALL of them are skipped: 31 % of the MACs of a 3x3 convolution on a 4x4 map,
                              // 23 % of the pool fold of an 8x8 map, 75 % of the sub-pixel classes of a 1x1 -> 2x2 map
    unsigned long long pm_perm;       // slot -> pixel, 4 bits each: pixels with the SAME set of valid taps sit next to each other (a tile that
    unsigned long long pm_perm_cls[4];  // straddles two slots then skips what both skip); per parity class in PC mode
};

__host__ __device__ __forceinline__ int pm_pixel(unsigned long long perm, int slot) { return (int)((perm >> (4 * slot)) & 15ull); }


// GEMM row -> (image, logical pixel)
__device__ __forceinline__ void geo_row(const Geo& g, int m, int& n, int& oy, int& ox) {
    if (g.pm) {
        const int slot = m / g.N;
        n = m - slot * g.N;
        const int pix = pm_pixel(g.pm_perm, slot);
        oy = pix >> g.logW;
        ox = pix & (g.W - 1);
    } else {
        n = m >> (g.logH + g.logW);
        oy = (m >> g.logW) & (g.H - 1);
        ox = m & (g.W - 1);
    }
}

// class-dependent part of a PC geometry (uniform per workgroup)
__host__ __device__ __forceinline__ void geo_select_class(Geo& g, int cls) {
    const int py = cls >> 1, px = cls & 1;
    const int K = g.pc_K, pd = (K - 1) / 2;
    g.py = py;
    g.px = px;
    if (g.pc_kind == 0) {  // conv_KxK(up2(x)), output parity (py, px)
        const int oy = (py - pd) >> 1, ox = (px - pd) >> 1;  // floor
        g.off_y = oy;
        g.off_x = ox;
        g.wa_base = pd + 1 - py + 2 * oy;
        g.wb_base = pd + 1 - px + 2 * ox;
        g.wa_step = g.wb_step = 2;
    } else {  // dgrad of the stride-2 folded conv, input parity (py, px)
        const int th = (K + 1) / 2;
        const int a0 = (py + pd) & 1, b0 = (px + pd) & 1;
        g.off_y = ((py + pd - a0) >> 1) - (th - 1);
        g.off_x = ((px + pd - b0) >> 1) - (th - 1);
        g.wa_base = a0 + 2 * (th - 1);
        g.wb_base = b0 + 2 * (th - 1);
        g.wa_step = g.wb_step = -2;
    }
}

struct ConvP {
    Geo g;
    const float* x;
    const float* w;
    const float* bias;
    const float* sigma;
    const float* res;
    const float* mask_x;
    float* y;
    int Ca, Cb;
    int Cin_w;
    int M;
    int Ktot;
    const float* zero;  // 16 bytes of zeros (out-of-range lanes load from here)
    unsigned x_bytes;   // size of the gathered tensor (buffer-resource range of the fast path)
    int tune_ks;        // host only: split-K factor (caller's, else the tuning table's; 0 = heuristic)
    int tune_tile;      // host only: tile configuration forced by the caller (0 = table / heuristic, < 0 = heuristic only)
    int y_zeroed;       // host only: the caller guarantees y holds zeros (split-K launches then skip their memset)
    int tune_kind;      // host only: row kind of the launch-tuning table (0 fwd, 1 dgrad k-major, 4 dgrad on transposed weights)
    float pos_inf;      // +infinity as a run-time value
    float pre_slope, mask_slope, out_scale;
    float res_scale;    // dgrad with a half-resolution residual (epilogue MODE 4): its factor
    float post_slope;   // forward: leaky-relu on the stored output (1 = none; never with split-K: the slices are combined by addition)
    int res_ups;  // residual stored at half the output resolution (nearest-upsampled on the fly)
    int ksplit;   // > 1: K-slices over grid.z, partial results combined with float atomics into a pre-zeroed y
    int kper;     // K-steps per slice
    int pix;      // floats between consecutive PIXELS of the gathered tensor: = Ca, except in the row-contiguous form of the <= 8-channel
                  // image layers (gim_conv2d_fwd_rows: Ca = the padded length of one tap ROW, K * Cin rounded up to 16, pix = Cin)
    int f16;      // host only: gim_conv_shape.prec == 1 - fp16 operands on v_mfma_f32_32x32x16_f16 where the launch is eligible (conv_f16.inc)
};

// GENF bit 0: generic K (channel count of the gathered tensor not a multiple of 16, or unaligned base)
// GENF bit 1: (BMODE 1 only) scalar loads of the k-major weight tile (output channels not a multiple of 4)
//
// __launch_bounds__(256, 4): cap the allocation at 128 registers (accumulators included) so that 4 workgroups
// (40 KB of LDS each at K step 16) share a CU: 4 waves per SIMD hide the global->LDS->MFMA latency of the
// one-barrier-per-K-step pipeline better than 3 (no spills: ~120 VGPRs).
//
// Address generation is kept off the critical path (PMC: at ~1 wave per SIMD the non-MFMA instruction stream, not
// memory, bounded the mid-size layers): per thread and tile row ONE 32-bit element offset is computed up front;
// per K step only a wave-uniform (tap, channel) offset advances - incrementally, no integer divisions - and the
// per-row work is two adds, two compares and the load.  Loaded values are not touched until store_tiles (any
// consumer would pull an s_waitcnt vmcnt(0) in front of the MFMA block).
// Out-of-range lanes (zero padding, ragged tile edges) load from this zero page instead of being masked off: the
// main loop then has no exec-mask branches around its global loads (one basic block, loads issue back to back).
// (Passed to the kernels as a pointer argument so that the select stays a GLOBAL load; a direct reference to the symbol
// turns the loads into flat loads, which also count against lgkmcnt and stall the LDS reads.)
__device__ __attribute__((aligned(16))) float g_zero4[4] = {0.f, 0.f, 0.f, 0.f};

static const float* zero_page() {
    static const float* z = [] {
        void* ptr = nullptr;
        if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(g_zero4)) != hipSuccess) ptr = nullptr;
        return (const float*)ptr;
    }();
    return z;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define BUF_OOB 0x80000000u   // byte offset beyond any num_records: the buffer load returns 0 for that lane
#define BUF_MAX_BYTES 0x7FFFFFF0ull   // largest operand one launch addresses (the launchers halve the batch beyond it)
// 16-byte load through a raw buffer resource: address = base + voff (per lane) + soff (wave-uniform, SGPR); lanes whose
// voff is outside [0, num_records) read zeros.  The per-K-step address work is ONE scalar add: ordinary VALU instructions
// are not free next to MFMA (tools/micro/mfma_valu.hip: each one takes ~4 cycles from the matrix pipe).
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// Epilogue of one accumulator block: NE values of ONE output channel `co` on rows mbase + (e & 3) + 8 * (e >> 2) (the C/D layout of
// the 32x32 and, for NE = 4, of the 16x16 MFMA).  MODE 0: y = a * scale + bias; 1: + residual; 2: + residual stored at half
// resolution (nearest-upsampled on the fly); 3: * leaky_relu'(mask).  All loads of the block are issued before the first is used
// (a load - wait - store chain per element made the epilogue of a short-K workgroup as long as its K loop), every offset is a
// per-lane VGPR offset (the SGPR offset of a raw buffer access is not bounds-checked: rows beyond a ragged tile edge must be
// dropped by the lane offset), and nothing in the element loops branches.
struct EpiCtx {
    __amdgpu_buffer_rsrc_t ry, rr, rm;
    float scale, mask_slope, post_slope;
    float res_scale;   // MODE 4: factor of the half-resolution residual (0.25: the average pool's backward)
    int M, Cb, logH, logW, Hm1, Wm1, os, py, px, Ho, Wo;
    int pmN;           // > 0: position-major rows (Geo.pm), m = slot * pmN + image, pixel = nibble `slot` of pm_perm
    unsigned long long pm_perm;
    bool atom, remap;
};

// byte offset in y of logical element (m, co) [and, MODE 2, of its half-resolution residual], rows beyond M not handled here
template <int MODE>
__device__ __forceinline__ void epi_elem_off(const EpiCtx& c, int m, int co, unsigned& o, unsigned& ro) {
    const int n = m >> (c.logH + c.logW);
    const int oy = ((m >> c.logW) & c.Hm1) * c.os + c.py;
    const int ox = (m & c.Wm1) * c.os + c.px;
    o = (unsigned)((((n * c.Ho + oy) * c.Wo + ox) * c.Cb + co) * 4);
    if constexpr (MODE == 2 || MODE == 4) ro = (unsigned)((((n * (c.Ho >> 1) + (oy >> 1)) * (c.Wo >> 1) + (ox >> 1)) * c.Cb + co) * 4);
}

template <int NE, int MODE>
__device__ __forceinline__ void epi_block(const EpiCtx& c, const float (&a)[NE], int mbase, int co, bool cok, float bv) {
    constexpr int CH = NE < 8 ? NE : 8;   // elements in flight: 8 loads per lane cover the latency, more would cost occupancy
    const unsigned rowb = (unsigned)c.Cb * 4u;
#pragma unroll
    for (int e0 = 0; e0 < NE; e0 += CH) {
        unsigned off[CH], roff[(MODE == 2 || MODE == 4) ? CH : 1];
        // rows come in groups of 4 consecutive m (mbase % 4 == 0): one full address computation per group where a group stays in
        // one image row (W >= 4), plain arithmetic otherwise; out-of-range lanes get the high bit (>= num_records: dropped)
        if (!c.remap && !c.pmN && MODE != 2 && MODE != 4) {
            const unsigned base = (unsigned)((mbase * c.Cb + co) * 4);
#pragma unroll
            for (int q = 0; q < CH; ++q) off[q] = base + (unsigned)(((e0 + q) & 3) + 8 * ((e0 + q) >> 2)) * rowb;
        } else if (c.Wm1 >= 3 && !c.pmN) {
#pragma unroll
            for (int g4 = 0; g4 < CH; g4 += 4) {
                unsigned o, ro = 0;
                epi_elem_off<MODE>(c, mbase + 8 * ((e0 + g4) >> 2), co, o, ro);
#pragma unroll
                for (int k = 0; k < 4 && g4 + k < CH; ++k) {
                    off[g4 + k] = o + (unsigned)(k * c.os) * rowb;
                    if constexpr (MODE == 2 || MODE == 4) roff[g4 + k] = ro + (unsigned)(((c.px + k * c.os) >> 1) - (c.px >> 1)) * rowb;
                }
            }
        } else if (c.pmN) {
            // position-major rows (m = pixel * N + image, N >= 32): ONE division for the block's first row; the other rows of the chunk
            // are < 32 rows further on, i.e. in the same or the next pixel position
            const int slot0 = mbase / c.pmN, n0 = mbase - slot0 * c.pmN;
            const int pixa = pm_pixel(c.pm_perm, slot0), pixb = pm_pixel(c.pm_perm, (slot0 + 1) & 15);
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const int d = ((e0 + q) & 3) + 8 * ((e0 + q) >> 2);
                const bool wrap = n0 + d >= c.pmN;
                const int n = wrap ? n0 + d - c.pmN : n0 + d, pix = wrap ? pixb : pixa;
                const int oy = ((pix >> c.logW) & c.Hm1) * c.os + c.py, ox = (pix & c.Wm1) * c.os + c.px;
                off[q] = (unsigned)((((n * c.Ho + oy) * c.Wo + ox) * c.Cb + co) * 4);
                if constexpr (MODE == 2 || MODE == 4) roff[q] = (unsigned)((((n * (c.Ho >> 1) + (oy >> 1)) * (c.Wo >> 1) + (ox >> 1)) * c.Cb + co) * 4);
            }
        } else {
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                unsigned ro = 0;
                epi_elem_off<MODE>(c, mbase + ((e0 + q) & 3) + 8 * ((e0 + q) >> 2), co, off[q], ro);
                if constexpr (MODE == 2 || MODE == 4) roff[q] = ro;
            }
        }
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const unsigned bad = (cok && mbase + ((e0 + q) & 3) + 8 * ((e0 + q) >> 2) < c.M) ? 0u : BUF_OOB;
            off[q] |= bad;
            if constexpr (MODE == 2 || MODE == 4) roff[q] |= bad;
        }
        float ld[MODE == 0 ? 1 : CH], ld2[MODE == 4 ? CH : 1];
        if constexpr (MODE == 4) {   // the input-gradient fan-in of a ResBlockDown: mask (full resolution) and the pooled skip gradient (half)
#pragma unroll
            for (int q = 0; q < CH; ++q) ld[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.rm, off[q], 0, 0));
#pragma unroll
            for (int q = 0; q < CH; ++q) ld2[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.rr, roff[q], 0, 0));
        } else if constexpr (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int q = 0; q < CH; ++q) ld[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.rr, MODE == 2 ? roff[q] : off[q], 0, 0));
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int q = 0; q < CH; ++q) ld[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(c.rm, off[q], 0, 0));
        }
        float v[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            v[q] = a[e0 + q] * c.scale + bv;
            if constexpr (MODE == 1 || MODE == 2) v[q] += ld[q];
            if constexpr (MODE == 3 || MODE == 4) v[q] *= (ld[q] > 0.f ? 1.0f : c.mask_slope);
            if constexpr (MODE == 4) v[q] += c.res_scale * ld2[q];
            if constexpr (MODE != 3 && MODE != 4) v[q] = fmaxf(v[q], v[q] * c.post_slope);   // post_slope = 1: identity (0 < slope <= 1)
        }
        if (c.atom) {
#pragma unroll
            for (int q = 0; q < CH; ++q) (void)__builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v[q], c.ry, off[q], 0, 0);
        } else {
#pragma unroll
            for (int q = 0; q < CH; ++q) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[q]), c.ry, off[q], 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);   // one chunk's offsets / loads at a time: the chunks must not pile up in registers
    }
}

// LDS bytes of the fp32 path's two double-buffered tiles (k-rows padded by 4 floats; the k-major B tile of BMODE 1 is unpadded)
constexpr int igemm_lds_bytes(int BM, int BN, int KB, int BMODE) {
    return 2 * (BM * (KB + 4) + (BMODE == 0 ? BN * (KB + 4) : KB * BN)) * 4;
}

template <int BM, int BN, int TM, int TN, int BMODE, int GENF, int KB>
__global__ __launch_bounds__(256, igemm_lds_bytes(BM, BN, KB, BMODE) <= 40960 ? 4 : 2) void conv_igemm_kernel(const ConvP p) {
    constexpr bool GEN = (GENF & 1) != 0;
    constexpr bool BSCALAR = (GENF & 2) != 0;
    // BN == 16: narrow outputs (<= 16 channels: RGB layers, the 6-channel image pair) use the 16x16x4 MFMA - a 32-wide tile
    // would spend 81-91 % of its MFMA work on padding columns.  One wave = 32*TM rows x 16 columns = 2*TM accumulator blocks.
    constexpr bool N16 = BN == 16;
    constexpr int WAVES_N = N16 ? 1 : BN / (32 * TN);
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
    constexpr int LDA = LDK;                       // dwords per k-row of an LDS tile
    constexpr int A_SZ = BM * LDA;
    constexpr int B_SZ = (BMODE == 0) ? BN * LDA : KB * BN;
    __shared__ __attribute__((aligned(16))) float lds[2 * A_SZ + 2 * B_SZ];
    float* As = lds;
    float* Bs = lds + 2 * A_SZ;

    Geo g = p.g;
    int kslice = blockIdx.z;
    if (g.pc) {
        geo_select_class(g, blockIdx.z & 3);
        g.pm_perm = p.g.pm_perm_cls[blockIdx.z & 3];   // (indexed in the kernel-argument segment: a dynamic index into the local copy would put it in scratch)
        kslice = blockIdx.z >> 2;
    }
    const int t = threadIdx.x;
    // Plain tile order.  (XCD-aware orders - one XCD per contiguous chunk of M tiles, or of weight tiles - were measured in
    // round 1: they cut the L2 misses of the dominant layer by 13 % and were not faster, profiles/r01_i_xcd_modes.txt.)
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int arow = t / QPR, aq = (t % QPR) * 4;
    const int He = g.Hin << g.ups, We = g.Win << g.ups;  // extent of the (virtually upsampled) gathered image
    const int KF2 = g.KF * g.KFw;

    // per-row constants of the A gather: pixel origin (for the bounds test) and element offset without the tap
    int a_oy[A_ROWS], a_ox[A_ROWS];
    long long a_off[A_ROWS];
    bool a_ok[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        const int m = m0 + arow + RP * i;
        a_ok[i] = m < p.M;
        const int n = m >> (g.logH + g.logW);
        a_oy[i] = ((m >> g.logW) & (g.H - 1)) * g.s_in + g.off_y;
        a_ox[i] = (m & (g.W - 1)) * g.s_in_x + g.off_x;
        a_off[i] = (long long)n * g.Hin * g.Win * p.Ca + (GEN ? 0 : aq);
        if (!g.ups) a_off[i] += ((long long)a_oy[i] * g.Win + a_ox[i]) * p.Ca;
    }
    // per-thread constants of the B tile
    long long b_off0[B_ROWS];
    bool b_ok0[B_ROWS];
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) {
        const int row = arow + RP * i;
        b_ok0[i] = row < BN && (n0 + row) < p.Cb;
        b_off0[i] = (long long)(n0 + row) * KF2 * p.Cin_w + aq;
    }
    const int b4_krow = t / B_U, b4_col = (t % B_U) * 4;
    const bool b4_cok = (n0 + b4_col) < p.Cb;  // Cb % 4 == 0 on the vector path: the whole quad is in range

    // ---- fast path (!GEN): buffer loads, per-lane byte offsets fixed per tap, wave-uniform K-step offsets in SGPRs ----
    // Rows beyond the ragged tile edge (m >= M, output channel >= Cb) are clamped to a valid row instead of zeroed: their
    // results are never stored.  Only the spatial zero padding needs zeros (BUF_OOB).
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0xFFFFFFFFu, 0x00020000);
    unsigned a_base[A_ROWS], a_cur[A_ROWS];
    unsigned b_voff[BMODE == 0 ? B_ROWS : B_PER4];
    if constexpr (!GEN) {
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            const int m = min(m0 + arow + RP * i, p.M - 1);
            int n, oy, ox;
            geo_row(g, m, n, oy, ox);
            a_oy[i] = oy * g.s_in + g.off_y;
            a_ox[i] = ox * g.s_in_x + g.off_x;
            a_base[i] = (unsigned)((n * g.Hin * g.Win * p.pix + aq) * 4);
            a_cur[i] = BUF_OOB;
        }
        if constexpr (BMODE == 0) {
#pragma unroll
            for (int i = 0; i < B_ROWS; ++i) {
                const int row = min(n0 + min(arow + RP * i, BN - 1), p.Cb - 1);
                b_voff[i] = (unsigned)((row * KF2 * p.Cin_w + aq) * 4);
            }
        } else if constexpr (!BSCALAR) {
#pragma unroll
            for (int i = 0; i < B_PER4; ++i) {
                const int krow = min(b4_krow + i * B_RSTEP, KB - 1);
                const int col = min(n0 + b4_col, p.Cb - 4);
                b_voff[i] = (unsigned)((krow * KF2 * p.Cin_w + col) * 4);
            }
        }
    }
    // per-tap refresh of the A offsets (runs when the uniform tap changes: every Ca / KB K-steps)
    auto set_tap = [&](int ta, int tb) {
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            const int iy = a_oy[i] + ta, ix = a_ox[i] + tb;
            const bool v = (unsigned)iy < (unsigned)He && (unsigned)ix < (unsigned)We;
            const int py = g.ups ? (iy >> 1) : iy, px = g.ups ? (ix >> 1) : ix;
            a_cur[i] = v ? a_base[i] + (unsigned)((py * g.Win + px) * p.pix * 4) : BUF_OOB;
        }
    };

    // staging registers (global -> VGPR -> LDS)
    f32x4 ra[A_ROWS];
    f32x4 rb0[B_ROWS];
    float rb1[B_PER];
    f32x4 rb4[B_PER4];

    // position-major rows: the taps that reach at least one of the tile's pixel positions (block-uniform; scalar code)
    unsigned tapmask = 0xFFFFFFFFu;
    if constexpr (!GEN) {
        if (g.pm) {
            tapmask = 0;
            const int slot0 = m0 / g.N, slot1 = min(m0 + BM - 1, p.M - 1) / g.N;
            for (int slot = slot0; slot <= slot1; ++slot) {
                const int pix = pm_pixel(g.pm_perm, slot);
                const int oy = (pix >> g.logW) * g.s_in + g.off_y, ox = (pix & (g.W - 1)) * g.s_in_x + g.off_x;
                for (int ta = 0; ta < g.Th; ++ta)
                    for (int tb = 0; tb < g.Tw; ++tb)
                        if ((unsigned)(oy + ta) < (unsigned)He && (unsigned)(ox + tb) < (unsigned)We) tapmask |= 1u << (ta * g.Tw + tb);
            }
        }
    }
    // wave-uniform K position, advanced incrementally by load_tiles (fast path)
    int k_c0 = 0, k_ta = 0, k_tb = 0;
    // position-major rows: K steps are counted over the VALID taps only; tap index of the v-th valid tap / of the next one
    auto pm_seek = [&](int vstep) {
        const int cps = p.Ca / KB;
        int q = vstep / cps, tap = 0;
        k_c0 = (vstep - q * cps) * KB;
        for (;; ++tap)
            if ((tapmask >> tap) & 1u) { if (q == 0) break; --q; }
        k_ta = tap / g.Tw;
        k_tb = tap - k_ta * g.Tw;
    };
    auto pm_next_tap = [&]() {
        int tap = k_ta * g.Tw + k_tb;
        const int T = g.Th * g.Tw;
        do { ++tap; } while (tap < T && !((tapmask >> tap) & 1u));
        if (tap >= T) tap = T - 1;     // behind the last valid tap: the step that would use it is never loaded for the MFMAs
        k_ta = tap / g.Tw;
        k_tb = tap - k_ta * g.Tw;
    };
    // K order: tap-major (all channel chunks of a tap, then the next tap).  (A channel-group-major order halved the L2 misses of
    // the dominant layer in round 1 and was not faster: profiles/r01_k_conv_k_order.txt.)
    auto seek = [&](int k0) {
        const int tap = k0 / p.Ca;
        k_c0 = k0 - tap * p.Ca;
        k_ta = tap / g.Tw;
        k_tb = tap - k_ta * g.Tw;
        set_tap(k_ta, k_tb);
    };

    auto load_tiles = [&](int k0) {
        if constexpr (!GEN) {
            const int ta = k_ta, tb = k_tb, c0 = k_c0;
            const int wtap = (g.wa_base + g.wa_step * ta) * g.KFw + g.wb_base + g.wb_step * tb;
            // ---- A: gathered activations (offsets of this tap are in a_cur; the channel offset is wave-uniform) ----
            const unsigned sa = (unsigned)(c0 * 4);
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) ra[i] = buf_load4(rx, a_cur[i], sa);
            // ---- B: weights ----
            if constexpr (BMODE == 0) {
                const unsigned sb = (unsigned)((wtap * p.Cin_w + c0) * 4);
#pragma unroll
                for (int i = 0; i < B_ROWS; ++i) rb0[i] = buf_load4(rw, b_voff[i], sb);
            } else {
                if constexpr (BSCALAR) {
                    const float* wb = p.w + (((long long)c0 * KF2 + wtap) * p.Cin_w + n0);
#pragma unroll
                    for (int i = 0; i < B_PER; ++i) {
                        const int idx = t + 256 * i;
                        const int krow = idx / BN, col = idx % BN;
                        rb1[i] = (n0 + col) < p.Cb ? wb[(long long)krow * KF2 * p.Cin_w + col] : 0.f;
                    }
                } else {
                    const unsigned sb = (unsigned)(((c0 * KF2 + wtap) * p.Cin_w) * 4);
#pragma unroll
                    for (int i = 0; i < B_PER4; ++i) rb4[i] = buf_load4(rw, b_voff[i], sb);
                }
            }
            // advance the uniform K position by one step (Ca % KB == 0 on this path)
            k_c0 += KB;
            if (k_c0 == p.Ca) {
                k_c0 = 0;
                if (g.pm) pm_next_tap();
                else if (++k_tb == g.Tw) { k_tb = 0; ++k_ta; }
                set_tap(k_ta, k_tb);
            }
        } else {
            // ---- generic K: per-element (tap, channel) decode; small layers only (Cin in {1,2,3,6}, Cout = 3) ----
#pragma unroll
            for (int i = 0; i < A_ROWS; ++i) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kf = k0 + aq + e;
                    const int tap = kf / p.Ca;
                    const int c = kf - tap * p.Ca;
                    const int ta = tap / g.Tw, tb = tap - ta * g.Tw;
                    const int iy = (a_oy[i] + ta), ix = (a_ox[i] + tb);
                    const bool v = a_ok[i] && kf < p.Ktot && (unsigned)iy < (unsigned)He && (unsigned)ix < (unsigned)We;
                    float val = 0.f;
                    if (v) {
                        const long long off = g.ups ? a_off[i] + (long long)((iy >> 1) * g.Win + (ix >> 1)) * p.Ca + c
                                                    : a_off[i] + (long long)(ta * g.Win + tb) * p.Ca + c;
                        val = p.x[off];
                    }
                    ra[i][e] = val;
                }
            }
            if constexpr (BMODE == 0) {
#pragma unroll
                for (int i = 0; i < B_ROWS; ++i) {
                    f32x4 val = {0.f, 0.f, 0.f, 0.f};
                    if (b_ok0[i]) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int kf = k0 + aq + e;
                            if (kf < p.Ktot) {
                                const int tap = kf / p.Ca, c = kf - tap * p.Ca;
                                const int ta = tap / g.Tw, tb = tap - ta * g.Tw;
                                const int wt = (g.wa_base + g.wa_step * ta) * g.KFw + g.wb_base + g.wb_step * tb;
                                val[e] = p.w[b_off0[i] - aq + (long long)wt * p.Cin_w + c];
                            }
                        }
                    }
                    rb0[i] = val;
                }
            } else {
                auto wrow = [&](int krow, bool& v) -> long long {  // offset of weight row (channel ca, mapped tap)
                    const int kf = k0 + krow;
                    const int tap = kf / p.Ca;
                    const int ca = kf - tap * p.Ca;
                    const int ta = tap / g.Tw, tb = tap - ta * g.Tw;
                    const int wt = (g.wa_base + g.wa_step * ta) * g.KFw + g.wb_base + g.wb_step * tb;
                    v = v && kf < p.Ktot;
                    return ((long long)ca * KF2 + wt) * p.Cin_w;
                };
                if constexpr (BSCALAR) {
#pragma unroll
                    for (int i = 0; i < B_PER; ++i) {
                        const int idx = t + 256 * i;
                        const int krow = idx / BN, col = idx % BN;
                        bool v = (n0 + col) < p.Cb;
                        const long long ro = wrow(krow, v);
                        rb1[i] = v ? p.w[ro + n0 + col] : 0.f;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < B_PER4; ++i) {
                        const int krow = b4_krow + i * B_RSTEP;
                        bool v = krow < KB && b4_cok;
                        const long long ro = wrow(krow, v);
                        f32x4 val = {0.f, 0.f, 0.f, 0.f};
                        if (v) val = *reinterpret_cast<const f32x4*>(p.w + ro + n0 + b4_col);
                        rb4[i] = val;
                    }
                }
            }
        }
    };

    const bool has_act = p.pre_slope != 1.0f;   // wave-uniform
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            // fused leaky-relu max(x, slope*x) (slope 1 = identity), IN PLACE and branch-free: a conditional copy of the
            // staged registers gets hoisted above the MFMA block and drags the s_waitcnt vmcnt(0) with it
            // (tools/isa_waitcnt_check.py), which exposes the global-load latency every K-step
            // (slope 1 - every dgrad, the 1x1 skip convs - skips the 2 VALU per element behind a wave-uniform branch: ordinary
            // VALU work costs matrix-pipe time)
            if (has_act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[i][e] = __builtin_amdgcn_fmed3f(ra[i][e], ra[i][e] * p.pre_slope, p.pos_inf);  // med3(x, s*x, +inf) = max(x, s*x) in 2 VALU (a literal inf folds back into the 3-op canonicalising max)
            }
            *reinterpret_cast<f32x4*>(&As[buf * A_SZ + (arow + RP * i) * LDK + aq]) = ra[i];
        }
        if constexpr (BMODE == 0) {
#pragma unroll
            for (int i = 0; i < B_ROWS; ++i) {
                const int row = arow + RP * i;
                if (BN % RP == 0 || row < BN) *reinterpret_cast<f32x4*>(&Bs[buf * B_SZ + row * LDK + aq]) = rb0[i];
            }
        } else {
            if constexpr (BSCALAR) {
#pragma unroll
                for (int i = 0; i < B_PER; ++i) Bs[buf * B_SZ + t + 256 * i] = rb1[i];  // [krow][col] row-major == idx
            } else {
#pragma unroll
                for (int i = 0; i < B_PER4; ++i) {
                    const int krow = b4_krow + i * B_RSTEP;
                    if (krow < KB) *reinterpret_cast<f32x4*>(&Bs[buf * B_SZ + krow * BN + b4_col]) = rb4[i];
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
    constexpr int NB16 = 2 * TM;                   // N16: 16-row accumulator blocks per wave
    const int r16 = lane & 15, q16 = lane >> 4;    // N16: row / column within a block, k quad
    f32x4 acc16[NB16];
#pragma unroll
    for (int i = 0; i < NB16; ++i) acc16[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk_all = (p.Ktot + KB - 1) / KB;
    int ks0 = kslice * p.kper;
    int nk = min(nk_all, ks0 + p.kper);
    bool pm_started = false;
    if constexpr (!GEN) {
        if (g.pm) {   // the slices divide the VALID steps of this tile
            const int nv = __builtin_popcount(tapmask) * (p.Ca / KB);
            const int per = (nv + p.ksplit - 1) / p.ksplit;
            ks0 = kslice * per;
            nk = min(nv, ks0 + per);
            if (ks0 >= nk) {
                if (kslice != 0) return;   // (block-uniform) nothing left for this slice; slice 0 always has the centre tap
            }
            pm_seek(ks0);
            set_tap(k_ta, k_tb);
            pm_started = true;
        }
    }
    if constexpr (!GEN) { if (!pm_started) seek(ks0 * KB); }
    load_tiles(ks0 * KB);
    store_tiles(0);
    __syncthreads();
    auto kstep = [&](int ks, auto BUFC, auto MAINC) {
        constexpr int buf = decltype(BUFC)::value;   // compile-time LDS buffer: offsets fold into the ds_read / ds_write immediates
        // MAIN: a step of the steady-state loop, whose loads and LDS stores are unconditional
        constexpr bool MAIN = decltype(MAINC)::value;
        if (MAIN || ks + 1 < nk) load_tiles((ks + 1) * KB);
        __builtin_amdgcn_sched_barrier(0);  // nothing that touches the staged registers may move into the MFMA block
        const float* Ab = As + buf * A_SZ;
        const float* Bb = Bs + buf * B_SZ;
        if constexpr (N16) {
            // v_mfma_f32_16x16x4_f32: lane (r16, q16) feeds A[row r16][k] and B[k][col r16] with k = 4*q16 + t in step t: one
            // ds_read_b128 per operand block serves 4 MFMAs (same k permutation on both operands)
#pragma unroll
            for (int kk = 0; kk < KB / 16; ++kk) {
                f32x4 a[NB16], b;
#pragma unroll
                for (int i = 0; i < NB16; ++i) a[i] = *reinterpret_cast<const f32x4*>(&Ab[(wm0 + 16 * i + r16) * LDK + 16 * kk + 4 * q16]);
                if constexpr (BMODE == 0) {
                    b = *reinterpret_cast<const f32x4*>(&Bb[r16 * LDK + 16 * kk + 4 * q16]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[e] = Bb[(16 * kk + 4 * q16 + e) * BN + r16];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < NB16; ++i) acc16[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[e], acc16[i], 0, 0, 0);
            }
        } else {
            // all operand fragments of the K step first, in registers of their own (with per-group arrays the compiler re-used the
            // same registers for the second half of a 32-deep step and could issue its LDS reads only after the first 8 MFMAs)
            f32x4 a[KB / 8][TM], b[KB / 8][TN];
#pragma unroll
            for (int kk = 0; kk < KB / 8; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[kk][i] = *reinterpret_cast<const f32x4*>(&Ab[(wm0 + 32 * i + r) * LDK + 8 * kk + 4 * h]);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (BMODE == 0) {
                        b[kk][j] = *reinterpret_cast<const f32x4*>(&Bb[(wn0 + 32 * j + r) * LDK + 8 * kk + 4 * h]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) b[kk][j][e] = Bb[(8 * kk + 4 * h + e) * BN + wn0 + 32 * j + r];
                    }
                }
            }
            // (128x128 tiles hold 64 accumulators per lane: there the barrier would cost spills, the compiler interleaves as before)
            if constexpr (TM * TN <= 2) __builtin_amdgcn_sched_barrier(0);   // keep the reads in front: the MFMA block below waits for them in order
#pragma unroll
            for (int kk = 0; kk < KB / 8; ++kk)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][i][e], b[kk][j][e], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MAIN || ks + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    };
    int ks = ks0;
    for (; ks + 2 < nk; ks += 2) {   // steady state: both steps have a successor to load - no conditions around the loads / stores
        kstep(ks, std::integral_constant<int, 0>(), std::true_type());
        kstep(ks + 1, std::integral_constant<int, 1>(), std::true_type());
    }
    for (; ks < nk; ks += 2) {
        kstep(ks, std::integral_constant<int, 0>(), std::false_type());
        if (ks + 1 < nk) kstep(ks + 1, std::integral_constant<int, 1>(), std::false_type());
    }

    // ---- epilogue: out_scale/sigma, bias, residual, activation mask; logical pixel -> stored pixel ----
    EpiCtx ec;
    ec.pmN = 0;
    ec.scale = p.out_scale * (p.sigma ? 1.0f / p.sigma[0] : 1.0f);
    ec.mask_slope = p.mask_slope;
    ec.post_slope = p.post_slope;
    ec.res_scale = p.res_scale;
    const bool first = kslice == 0;
    const bool has_res = p.res != nullptr && first, has_mask = p.mask_x != nullptr;   // block-uniform
    ec.atom = p.ksplit > 1;
    ec.remap = g.os != 1;
    ec.M = p.M; ec.Cb = p.Cb;
    ec.logH = g.logH; ec.logW = g.logW; ec.Hm1 = g.H - 1; ec.Wm1 = g.W - 1; ec.os = g.os; ec.py = g.py; ec.px = g.px;
    ec.Ho = g.H * g.os; ec.Wo = g.W * g.os;
    ec.pmN = g.pm ? g.N : 0;
    ec.pm_perm = g.pm_perm;
    // y (and the mask, which has y's shape) in bytes: the host guarantees < 2 GiB per launch
    const unsigned ybytes = (unsigned)g.N * (unsigned)ec.Ho * (unsigned)ec.Wo * (unsigned)p.Cb * 4u;
    ec.ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, ybytes, 0x00020000);
    ec.rr = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, has_res ? ((p.res_ups || has_mask) ? ybytes >> 2 : ybytes) : 0u, 0x00020000);
    ec.rm = __builtin_amdgcn_make_buffer_rsrc((void*)p.mask_x, 0, has_mask ? ybytes : 0u, 0x00020000);
    auto run = [&](auto MODEC) {
        constexpr int MODE = decltype(MODEC)::value;
        if constexpr (N16) {   // accumulator block i, register e: row 16*i + 4*q16 + e, column r16
            const int co = n0 + r16;
            const bool cok = co < p.Cb;
            const float bv = (p.bias && first && cok) ? p.bias[co] : 0.f;
#pragma unroll
            for (int i = 0; i < NB16; ++i) {
                const float a[4] = {acc16[i][0], acc16[i][1], acc16[i][2], acc16[i][3]};
                epi_block<4, MODE>(ec, a, m0 + wm0 + 16 * i + 4 * q16, co, cok, bv);
            }
        } else {               // block (i, j), register e: row 32*i + (e & 3) + 8*(e >> 2) + 4*h, column 32*j + r
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int co = n0 + wn0 + 32 * j + r;
                    const bool cok = co < p.Cb;
                    const float bv = (p.bias && first && cok) ? p.bias[co] : 0.f;
                    float a[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) a[e] = acc[i][j][e];
                    epi_block<16, MODE>(ec, a, m0 + wm0 + 32 * i + 4 * h, co, cok, bv);
                }
        }
    };
    if (has_res && has_mask) {     // dgrad with the pooled skip gradient folded in (gim_conv2d_dgrad_res): mask, then + res_scale * up2(res)
        run(std::integral_constant<int, 4>());
    } else if (has_res) {
        if (p.res_ups) run(std::integral_constant<int, 2>());
        else run(std::integral_constant<int, 1>());
    } else if (has_mask) {
        run(std::integral_constant<int, 3>());
    } else {
        run(std::integral_constant<int, 0>());
    }
}

// -------------------------------------------------------------------------------------------------
// Patch-resident form of the fast path for plain 3x3 convolutions (forward, dgrad on k-major or transposed weights).
// The taps of a tile of BM consecutive output pixels (whole image rows, or part of one) read the same (rows + 2) x (columns + 2)
// input patch: it is loaded ONCE per 16-channel chunk (K order: channel chunk outer, taps inner), activated once, and every tap's
// A fragment is that LDS image at a wave-uniform offset.  Against the tap-major loop above that is 9x fewer activation loads,
// ds_writes, LeakyReLUs and per-row address updates per K step - the instructions that wait for gaps between the other waves'
// MFMAs - and the halo rows are fetched once per workgroup instead of once per tap
// (tools/micro/igemm_lab.hip, profiles/r03_k_patch_resident_lab.txt: +6-8 % on 32x32 64->128, +20 % on 16x16 128->256, 0.92 of
// the MFMA peak at K = 4608).  Only the weight tile is staged per K step.
// Host guarantees: plain geometry (stride 1, no upsample, no parity classes), 3x3 taps, Ca % 16 == 0, H * W >= BM (a tile never
// crosses an image), vector weight loads, split-K in whole chunks (kper = chunks per slice * 9).
// -------------------------------------------------------------------------------------------------
template <int BM, int BN, int TM, int TN, int BMODE>
__global__ __launch_bounds__(256, TM * TN <= 2 ? 3 : 2) void conv_igemm_patch_kernel(const ConvP p) {
    constexpr int KB = 16, LDK = KB + 4;
    constexpr int WAVES_N = BN / (32 * TN), WAVES_M = BM / (32 * TM);
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    constexpr int P_PER = BM == 64 ? 4 : 5;          // patch quads per thread: (BM / W + 2) * (min(W, BM) + 2) pixels <= 198 (BM 64) / 264 (BM 128, W <= 64)
    constexpr int B_ROWS = BN / 64;                  // BMODE 0: weight rows per thread
    constexpr int B_U = BN / 4, B_RSTEP = 256 / B_U; // BMODE 1: float4 units per k-row, k-rows per pass
    constexpr int B_PER4 = (KB + B_RSTEP - 1) / B_RSTEP;
    constexpr int B_SZ = (BMODE == 0) ? BN * LDK : KB * BN;
    extern __shared__ __attribute__((aligned(16))) float patch_lds[];
    const Geo& g = p.g;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int kslice = blockIdx.z;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    // patch geometry (workgroup-uniform)
    const int Wt = min(g.W, BM), logWt = 31 - __builtin_clz(Wt);
    const int PW = Wt + 2, PP = (BM / Wt + 2) * PW;
    const int P_SZ = (PP * LDK + 3) & ~3;
    float* Ps = patch_lds;
    float* Bs = patch_lds + 2 * P_SZ;
    const int n_img = m0 >> (g.logH + g.logW);
    const int oy0 = ((m0 >> g.logW) & (g.H - 1)) + g.off_y, ox0 = (m0 & (g.W - 1)) + g.off_x;
    const int KF2 = g.KF * g.KFw;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0xFFFFFFFFu, 0x00020000);
    unsigned p_voff[P_PER];
    int p_lds[P_PER];
#pragma unroll
    for (int i = 0; i < P_PER; ++i) {
        const int q = t + 256 * i, pp = q >> 2, quad = q & 3;
        const int pr = pp / PW, pc = pp - pr * PW;
        const int iy = oy0 + pr, ix = ox0 + pc;
        const bool v = pp < PP && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        p_voff[i] = v ? (unsigned)((((n_img * g.Hin + iy) * g.Win + ix) * p.Ca + quad * 4) * 4) : BUF_OOB;
        p_lds[i] = pp < PP ? pp * LDK + quad * 4 : -1;
    }
    unsigned b_voff[BMODE == 0 ? B_ROWS : B_PER4];
    const int brow = t >> 2, bq = (t & 3) * 4;                 // BMODE 0
    const int b4_krow = t / B_U, b4_col = (t % B_U) * 4;       // BMODE 1
    if constexpr (BMODE == 0) {
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) b_voff[i] = (unsigned)((min(n0 + brow + 64 * i, p.Cb - 1) * KF2 * p.Cin_w + bq) * 4);
    } else {
#pragma unroll
        for (int i = 0; i < B_PER4; ++i)
            b_voff[i] = (unsigned)((min(b4_krow + i * B_RSTEP, KB - 1) * KF2 * p.Cin_w + min(n0 + b4_col, p.Cb - 4)) * 4);
    }
    const bool has_act = p.pre_slope != 1.0f;
    f32x4 rp[P_PER], rb[BMODE == 0 ? B_ROWS : B_PER4];
    auto load_patch = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < P_PER; ++i) rp[i] = buf_load4(rx, p_voff[i], (unsigned)(c0 * 4));
    };
    int p_wr = 0;   // float offset of the patch buffer the next store_patch fills (alternates 0 / P_SZ)
    auto store_patch = [&](int) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < P_PER; ++i) {
            if (has_act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) rp[i][e] = __builtin_amdgcn_fmed3f(rp[i][e], rp[i][e] * p.pre_slope, p.pos_inf);
            }
            if (p_lds[i] >= 0) *reinterpret_cast<f32x4*>(&Ps[p_wr + p_lds[i]]) = rp[i];
        }
        p_wr = p_wr ? 0 : P_SZ;
    };
    auto load_b = [&](int ta, int tb, int c0) __attribute__((always_inline)) {
        const int wtap = (g.wa_base + g.wa_step * ta) * g.KFw + g.wb_base + g.wb_step * tb;
        if constexpr (BMODE == 0) {
            const unsigned sb = (unsigned)((wtap * p.Cin_w + c0) * 4);
#pragma unroll
            for (int i = 0; i < B_ROWS; ++i) rb[i] = buf_load4(rw, b_voff[i], sb);
        } else {
            const unsigned sb = (unsigned)(((c0 * KF2 + wtap) * p.Cin_w) * 4);
#pragma unroll
            for (int i = 0; i < B_PER4; ++i) rb[i] = buf_load4(rw, b_voff[i], sb);
        }
    };
    auto store_b = [&](int buf) __attribute__((always_inline)) {
        if constexpr (BMODE == 0) {
#pragma unroll
            for (int i = 0; i < B_ROWS; ++i) *reinterpret_cast<f32x4*>(&Bs[buf * B_SZ + (brow + 64 * i) * LDK + bq]) = rb[i];
        } else {
#pragma unroll
            for (int i = 0; i < B_PER4; ++i) {
                const int krow = b4_krow + i * B_RSTEP;
                if (krow < KB) *reinterpret_cast<f32x4*>(&Bs[buf * B_SZ + krow * BN + b4_col]) = rb[i];
            }
        }
    };
    const int r = lane & 31, h = lane >> 5;
    const int wm0 = (wv / WAVES_N) * 32 * TM, wn0 = (wv % WAVES_N) * 32 * TN;
    // lane constants of the A fragment: tile-local pixel (ty, tx) of row wm0 + 32 i + r sits at patch position (ty + ta) * PW + tx + tb
    // for tap (ta, tb): one base per tap ROW (the patch buffer of the current chunk folded in), tb and the k quad are immediates
    int a_row[3][TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int q = wm0 + 32 * i + r;
        const int base = ((q >> logWt) * PW + (q & (Wt - 1))) * LDK + 4 * h;
#pragma unroll
        for (int ta = 0; ta < 3; ++ta) a_row[ta][i] = base + ta * PW * LDK;
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // split-K in whole chunks: kper = chunks per slice * 9
    const int nchunk = p.Ca / KB;
    const int ch0 = kslice * (p.kper / 9);
    const int ch1 = min(nchunk, ch0 + p.kper / 9);
    load_patch(ch0 * KB);
    load_b(0, 0, ch0 * KB);
    store_patch(0);
    store_b(0);
    __syncthreads();
    // one K step: tap TAP of chunk ch (weight buffer BB: compile-time, the patch buffer is folded into a_row); prefetches the next
    // step's weights and - on the first tap - the next chunk's patch, which goes to LDS behind the last tap's MFMAs
    auto kstep = [&](int ch, bool more_chunks, auto TAPC, auto BBC) __attribute__((always_inline)) {
        constexpr int TAP = decltype(TAPC)::value, BB = decltype(BBC)::value;
        constexpr int TA = TAP / 3, TB = TAP % 3;
        if (TAP == 0 && more_chunks) load_patch((ch + 1) * KB);
        if (TAP < 8) load_b((TAP + 1) / 3, (TAP + 1) % 3, ch * KB);
        else if (more_chunks) load_b(0, 0, (ch + 1) * KB);
        __builtin_amdgcn_sched_barrier(0);
        const float* Bb = Bs + BB * B_SZ;
        f32x4 a[2][TM], b[2][TN];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[kk][i] = *reinterpret_cast<const f32x4*>(&Ps[a_row[TA][i] + TB * LDK + 8 * kk]);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (BMODE == 0) {
                    b[kk][j] = *reinterpret_cast<const f32x4*>(&Bb[(wn0 + 32 * j + r) * LDK + 8 * kk + 4 * h]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[kk][j][e] = Bb[(8 * kk + 4 * h + e) * BN + wn0 + 32 * j + r];
                }
            }
        }
        if constexpr (TM * TN <= 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][i][e], b[kk][j][e], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (TAP < 8 || more_chunks) store_b(BB ^ 1);
        if (TAP == 8 && more_chunks) store_patch(-1);
        __syncthreads();
    };
    int pbuf = 0;   // patch buffer of the chunk being computed
    auto chunk = [&](int ch, auto B0C) __attribute__((always_inline)) {   // nine taps; the weight buffer alternates, starting at B0
        constexpr int B0 = decltype(B0C)::value;
        using Ba = std::integral_constant<int, B0>;
        using Bb_ = std::integral_constant<int, B0 ^ 1>;
        const bool more_chunks = ch + 1 < ch1;
        kstep(ch, more_chunks, std::integral_constant<int, 0>(), Ba());
        kstep(ch, more_chunks, std::integral_constant<int, 1>(), Bb_());
        kstep(ch, more_chunks, std::integral_constant<int, 2>(), Ba());
        kstep(ch, more_chunks, std::integral_constant<int, 3>(), Bb_());
        kstep(ch, more_chunks, std::integral_constant<int, 4>(), Ba());
        kstep(ch, more_chunks, std::integral_constant<int, 5>(), Bb_());
        kstep(ch, more_chunks, std::integral_constant<int, 6>(), Ba());
        kstep(ch, more_chunks, std::integral_constant<int, 7>(), Bb_());
        kstep(ch, more_chunks, std::integral_constant<int, 8>(), Ba());
        // the next chunk's patch is in the other buffer: move the A bases there
        const int d = pbuf ? -P_SZ : P_SZ;
        pbuf ^= 1;
#pragma unroll
        for (int ta = 0; ta < 3; ++ta)
#pragma unroll
            for (int i = 0; i < TM; ++i) a_row[ta][i] += d;
    };
    int ch = ch0;
    for (; ch + 1 < ch1; ch += 2) {   // pairs: after nine steps the weight buffer parity has flipped
        chunk(ch, std::integral_constant<int, 0>());
        chunk(ch + 1, std::integral_constant<int, 1>());
    }
    if (ch < ch1) chunk(ch, std::integral_constant<int, 0>());

    // ---- epilogue (as conv_igemm_kernel) ----
    EpiCtx ec;
    ec.pmN = 0;
    ec.scale = p.out_scale * (p.sigma ? 1.0f / p.sigma[0] : 1.0f);
    ec.mask_slope = p.mask_slope;
    ec.post_slope = p.post_slope;
    ec.res_scale = p.res_scale;
    const bool first = kslice == 0;
    const bool has_res = p.res != nullptr && first, has_mask = p.mask_x != nullptr;
    ec.atom = p.ksplit > 1;
    ec.remap = false;
    ec.M = p.M; ec.Cb = p.Cb;
    ec.logH = g.logH; ec.logW = g.logW; ec.Hm1 = g.H - 1; ec.Wm1 = g.W - 1; ec.os = 1; ec.py = 0; ec.px = 0;
    ec.Ho = g.H; ec.Wo = g.W;
    const unsigned ybytes = (unsigned)g.N * (unsigned)g.H * (unsigned)g.W * (unsigned)p.Cb * 4u;
    ec.ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, ybytes, 0x00020000);
    ec.rr = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, has_res ? ((p.res_ups || has_mask) ? ybytes >> 2 : ybytes) : 0u, 0x00020000);
    ec.rm = __builtin_amdgcn_make_buffer_rsrc((void*)p.mask_x, 0, has_mask ? ybytes : 0u, 0x00020000);
    auto run = [&](auto MODEC) {
        constexpr int MODE = decltype(MODEC)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int co = n0 + wn0 + 32 * j + r;
                const bool cok = co < p.Cb;
                const float bv = (p.bias && first && cok) ? p.bias[co] : 0.f;
                float a[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) a[e] = acc[i][j][e];
                epi_block<16, MODE>(ec, a, m0 + wm0 + 32 * i + 4 * h, co, cok, bv);
            }
    };
    if (has_res && has_mask) {     // dgrad with the pooled skip gradient folded in (gim_conv2d_dgrad_res): mask, then + res_scale * up2(res)
        run(std::integral_constant<int, 4>());
    } else if (has_res) {
        if (p.res_ups) run(std::integral_constant<int, 2>());
        else run(std::integral_constant<int, 1>());
    } else if (has_mask) {
        run(std::integral_constant<int, 3>());
    } else {
        run(std::integral_constant<int, 0>());
    }
}

// -------------------------------------------------------------------------------------------------
// wgrad
// -------------------------------------------------------------------------------------------------
struct WgP {
    Geo g;             // logical grid = pixels of dy; gather geometry of x (plain or stride 2)
    const float* dy;
    const float* x;
    float* slabs;
    float* bias_slabs;
    int Cin, Cout;     // channels of x / of dy (roles as seen by this kernel)
    int M;
    int Kcols;
    int mper;
    float pre_slope;   // leaky-relu on the gathered x operand
    float a_slope;     // leaky-relu on the dy operand (role-swapped use: sub-pixel conv wgrad)
    const float* zero; // 16 bytes of zeros (out-of-range lanes load from here)
    unsigned x_bytes;  // size of the gathered tensor x (buffer-resource range)
    float pos_inf;     // +infinity as a run-time value (keeps med3(x, s*x, inf) from folding back into a 3-op max)
    int atomic;        // 1: all pixel slices add into ONE pre-zeroed slab with float atomics (no reduce pass)
    int ns;            // pixel slices
    int xcd;           // 1: XCD-aware block order (the grid's z extent is then ns rounded up to a multiple of 8, see wgrad_block)
    int pix;           // floats between consecutive pixels of x: = Cin, except in the row-contiguous form (gim_conv2d_wgrad_rows_acc:
                       // "Cin" = padded tap-row length, pix = the image's channel count)
};

// Block -> (column tile, row tile, pixel slice), XCD-aware.  Every tile of one pixel slice streams the SAME dY rows and x rows
// (a 3x3 layer with 512 channels has 72 column tiles x 4 row tiles per slice), and each of the 8 XCDs has an L2 of its own while
// the hardware hands consecutive workgroups of a launch to the XCDs round-robin: with the plain (x, y, z) order the tiles of a
// slice are spread over all eight L2s and every XCD fetches every slice.  Here all tiles of a slice get linear ids that are
// congruent mod 8 - one XCD, consecutive in its dispatch order - so a slice's operands (a few hundred KB) are fetched into one L2
// once (measured: -14 GB of L2 misses per training step, profiles/r02_y_*).  Only for launches with many slices (the host sets
// p.xcd): a slice lives on ONE XCD, so 8x8-map layers with 2-20 slices would leave most of the chip idle (measured 3.5x slower).
// Returns false for the padding blocks (slice >= ns): they exit before any barrier.
__device__ __forceinline__ bool wgrad_block(const WgP& p, int& bx, int& by, int& bz) {
    if (!p.xcd) {
        bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
        return true;
    }
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned xcd = L & 7u, q = L >> 3, ntiles = gx * gy;
    // (the XCDs walk the tiles in rotated orders: in the same order they would all add into the same output tile at the same time
    //  and their float atomics would queue up on its cache lines)
    const unsigned tile = (q + xcd * ((ntiles + 7u) / 8u)) % ntiles, slice = (q / ntiles) * 8u + xcd;
    bx = (int)(tile % gx); by = (int)(tile / gx); bz = (int)slice;
    return slice < (unsigned)p.ns;
}

// VA / VB = 4: the dY rows (A) / the gathered x rows (B) are fetched as float4 (their channel count % 4 == 0, 16-byte aligned
// base); = 1: scalar fallback for that operand alone (3- and 6-channel image layers, 1-channel Omniglot: the 64-channel dY of
// the first encoder conv still streams as float4 while its 3-channel input is gathered element by element).
// Workgroups of the first column tile also produce the bias gradient sum_m dY[m][co] of their pixel slice from
// the dY values they stream anyway (bias_slabs[slice][Cout]).
// WBK: pixels per K step (16; 32 for the 64x64 tile of tile code 6432 - that tile does 8 MFMAs per wave and step, so the per-step
// costs weigh twice as much as on the larger tiles; 32 KB of LDS, still 4 workgroups per CU).
template <int BM, int BN, int TM, int TN, int VA, int VB, bool FASTB, int WBK = BK>
__global__ __launch_bounds__(256, 1) void conv_wgrad_kernel(const WgP p) {
    constexpr int WAVES_N = BN / (32 * TN);
    constexpr int WAVES_M = BM / (32 * TM);
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    static_assert(!FASTB || VB == 4, "fast B addressing: vector path only");
    constexpr int AU = BM / VA, BU = BN / VB;                   // load units per tile row
    constexpr int A_RSTEP = 256 / AU, B_RSTEP = 256 / BU;       // tile rows covered per pass
    constexpr int A_PER = (WBK + A_RSTEP - 1) / A_RSTEP, B_PER = (WBK + B_RSTEP - 1) / B_RSTEP;
    constexpr int A_FLOATS = WBK * BM, B_FLOATS = WBK * BN;
    __shared__ __attribute__((aligned(16))) float As[2][A_FLOATS];
    __shared__ __attribute__((aligned(16))) float Bs[2][B_FLOATS];

    const Geo& g = p.g;
    const int t = threadIdx.x;
    int bx, by, bz;
    if (!wgrad_block(p, bx, by, bz)) return;
    const int j0 = bx * BN, co0 = by * BM;
    const int mbeg = bz * p.mper;
    const int mend = min(p.M, mbeg + p.mper);
    const int He = g.Hin << g.ups, We = g.Win << g.ups;

    const int ac = (t % AU) * VA, ak = t / AU;
    const bool a_cok = (co0 + ac) < p.Cout;
    const int bc = (t % BU) * VB, bk = t / BU;
    const int j = j0 + bc;
    const bool b_jok = j < p.Kcols;
    const int tap = b_jok ? j / p.Cin : 0;
    const int ci = b_jok ? j - tap * p.Cin : 0;
    const int ta = tap / g.Tw, tb = tap - ta * g.Tw;
    const int dh = ta + g.off_y, dw = tb + g.off_x;
    const bool do_bias = p.bias_slabs != nullptr && bx == 0;

    // ---- vector-path fast addressing (tools/micro/mfma_valu.hip: VALU work is paid in matrix-pipe time) ----
    // A (dy rows): buffer resource whose BASE advances by WBK rows per step and whose num_records shrinks to the rows left
    //   in this pixel slice: lane offsets are constants and rows beyond the slice read zeros - no per-step VALU at all.
    // B (gathered x): when the WBK pixels of a K step lie in ONE image (H * W % WBK == 0, no on-the-fly upsample) pixel `row` of
    //   the step sits at (oy0 + (row >> logW), ox0 + (row & (W - 1))) with (oy0, ox0) wave-uniform, so the element offset is
    //   U(n, oy0, ox0)  [SALU]  +  L(tap, channel, tile row)  [lane constant]; only the zero-padding test is per lane (two adds,
    //   two compares, one select per tile row).  Otherwise (1x1 and 2x2 maps) the generic per-row address path runs.
    constexpr bool fastb = FASTB;  // host: VB == 4 && ups == 0 && H * W % WBK == 0
    unsigned a_v[A_PER], b_l[B_PER];
    int b_dx[B_PER], b_dy[B_PER];
    // the most negative lane constant is shifted into the base pointer so that every offset is a non-negative 32-bit value
    const int b_bias = ((g.off_y < 0 ? -g.off_y : 0) * g.Win + (g.off_x < 0 ? -g.off_x : 0)) * p.pix * 4;
    const __amdgpu_buffer_rsrc_t rxb = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x - b_bias), 0, p.x_bytes + (unsigned)b_bias, 0x00020000);
    if constexpr (VA == 4) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int row = ak + i * A_RSTEP;
            a_v[i] = row < WBK ? (unsigned)((row * p.Cout + min(co0 + ac, p.Cout - 4)) * 4) : BUF_OOB;
        }
    }
    if constexpr (VB == 4) {
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int row = bk + i * B_RSTEP;
            b_dy[i] = (row >> g.logW) * g.s_in + dh;
            b_dx[i] = (row & (g.W - 1)) * g.s_in + dw;
            b_l[i] = (b_jok && row < WBK) ? (unsigned)(((b_dy[i] * g.Win + b_dx[i]) * p.pix + ci) * 4 + b_bias) : BUF_OOB;
        }
    }

    // staging registers
    float ra[A_PER][VA], rb[B_PER][VB];
    float bsum[VA];
#pragma unroll
    for (int e = 0; e < VA; ++e) bsum[e] = 0.f;

    auto load_tiles = [&](int mb) {
        if constexpr (VA == 4) {
            const int left = mend - mb;  // > 0
            const __amdgpu_buffer_rsrc_t ra_rs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(p.dy + (long long)mb * p.Cout), 0, (unsigned)min(left, WBK) * (unsigned)p.Cout * 4u, 0x00020000);
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const f32x4 val = buf_load4(ra_rs, a_v[i], 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[i][e] = val[e];
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const int row = ak + i * A_RSTEP;
                const int m = mb + row;
                const bool v = a_cok && row < WBK && m < mend;
                ra[i][0] = v ? p.dy[(long long)m * p.Cout + co0 + ac] : 0.f;
            }
        }
        if constexpr (VB == 4) {
            if constexpr (fastb) {
                const int n = mb >> (g.logH + g.logW);
                const int oys = ((mb >> g.logW) & (g.H - 1)) * g.s_in, oxs = (mb & (g.W - 1)) * g.s_in;
                const unsigned u = (unsigned)((((n * g.Hin + oys) * g.Win) + oxs) * p.pix * 4);
#pragma unroll
                for (int i = 0; i < B_PER; ++i) {
                    const bool v = (unsigned)(oys + b_dy[i]) < (unsigned)He && (unsigned)(oxs + b_dx[i]) < (unsigned)We;
                    const f32x4 val = buf_load4(rxb, v ? b_l[i] : BUF_OOB, u);
#pragma unroll
                    for (int e = 0; e < 4; ++e) rb[i][e] = val[e];
                }
            } else {
#pragma unroll
                for (int i = 0; i < B_PER; ++i) {
                    const int row = bk + i * B_RSTEP;
                    const int m = mb + row;
                    const int n = m >> (g.logH + g.logW);
                    const int iy = ((m >> g.logW) & (g.H - 1)) * g.s_in + dh;
                    const int ix = (m & (g.W - 1)) * g.s_in + dw;
                    const bool v = b_jok && row < WBK && m < mend && (unsigned)iy < (unsigned)He && (unsigned)ix < (unsigned)We;
                    const unsigned off = (unsigned)((((n * g.Hin + (iy >> g.ups)) * g.Win + (ix >> g.ups)) * p.pix + ci) * 4 + b_bias);
                    const f32x4 val = buf_load4(rxb, v ? off : BUF_OOB, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) rb[i][e] = val[e];
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                const int row = bk + i * B_RSTEP;
                const int m = mb + row;
                const int n = m >> (g.logH + g.logW);
                const int iy = ((m >> g.logW) & (g.H - 1)) * g.s_in + dh;
                const int ix = (m & (g.W - 1)) * g.s_in + dw;
                const bool v = b_jok && row < WBK && m < mend && (unsigned)iy < (unsigned)He && (unsigned)ix < (unsigned)We;
                const long long off = ((long long)(n * g.Hin + (iy >> g.ups)) * g.Win + (ix >> g.ups)) * p.pix + ci;
                rb[i][0] = v ? p.x[off] : 0.f;
            }
        }
    };
    // activations and the bias sums consume the loaded values here, after the MFMA block, never in load_tiles
    const bool act_a = p.a_slope != 1.0f, act_b = p.pre_slope != 1.0f;  // block-uniform
    auto store_tiles = [&](int buf) {
        if (act_a) {
#pragma unroll
            for (int i = 0; i < A_PER; ++i)
#pragma unroll
                for (int e = 0; e < VA; ++e) ra[i][e] = __builtin_amdgcn_fmed3f(ra[i][e], ra[i][e] * p.a_slope, p.pos_inf);
        }
        if (do_bias) {   // block-uniform (first column tile only)
            asm volatile("" ::: "memory");   // keeps this a real branch: if-converted, its adds + selects ran in EVERY workgroup's K loop
#pragma unroll
            for (int i = 0; i < A_PER; ++i)
#pragma unroll
                for (int e = 0; e < VA; ++e) bsum[e] += ra[i][e];
        }
        if (act_b) {
#pragma unroll
            for (int i = 0; i < B_PER; ++i)
#pragma unroll
                for (int e = 0; e < VB; ++e) rb[i][e] = __builtin_amdgcn_fmed3f(rb[i][e], rb[i][e] * p.pre_slope, p.pos_inf);
        }
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int row = ak + i * A_RSTEP;
            if (row < WBK) {
                if constexpr (VA == 4) {
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
            if (row < WBK) {
                if constexpr (VB == 4) {
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
    const int nk = (mend > mbeg) ? (mend - mbeg + WBK - 1) / WBK : 0;
    if (nk > 0) {
        load_tiles(mbeg);
        store_tiles(0);
    }
    __syncthreads();
    // the LDS buffer of a step is a compile-time constant (two steps per trip): its offset folds into the ds_read / ds_write
    // immediates instead of costing address VALU next to the MFMAs (every VALU instruction is paid in matrix-pipe time)
    const float* a_rd = &As[0][0] + h * BM + wm0 + r;
    const float* b_rd = &Bs[0][0] + h * BN + wn0 + r;
    auto kstep32 = [&](int ks, auto BUFC, auto MAINC) {
        constexpr int buf = decltype(BUFC)::value;
        constexpr bool MAIN = decltype(MAINC)::value;   // a step of the steady-state loop: it has a successor, loads / stores unconditional
        if (MAIN || ks + 1 < nk) load_tiles(mbeg + (ks + 1) * WBK);
        __builtin_amdgcn_sched_barrier(0);  // keep every consumer of the staged registers behind the MFMA block
#pragma unroll
        for (int kp = 0; kp < WBK / 2; ++kp) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = a_rd[buf * A_FLOATS + 2 * kp * BM + 32 * i];
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) b[jj] = b_rd[buf * B_FLOATS + 2 * kp * BN + 32 * jj];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jj = 0; jj < TN; ++jj)
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[jj], acc[i][jj], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MAIN || ks + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    };
    int ks = 0;
    for (; ks + 2 < nk; ks += 2) {   // steady state: whole pairs, every step has a successor
        kstep32(ks, std::integral_constant<int, 0>(), std::true_type());
        kstep32(ks + 1, std::integral_constant<int, 1>(), std::true_type());
    }
    for (; ks + 1 < nk; ks += 2) {   // whole pairs: no control flow between the two steps (a branch there makes the compiler carry
        kstep32(ks, std::integral_constant<int, 0>(), std::false_type());       // the accumulators in VGPRs and copy them to AGPRs every trip)
        kstep32(ks + 1, std::integral_constant<int, 1>(), std::false_type());
    }
    if (ks < nk) kstep32(ks, std::integral_constant<int, 0>(), std::false_type());

    // each accumulator register is two 128-byte row segments per wave: the shape float atomics run at full rate for
    float* out = p.slabs + (p.atomic ? 0 : (long long)bz * p.Cout * p.Kcols);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < TN; ++jj) {
            const int col = j0 + wn0 + 32 * jj + r;
            if (col >= p.Kcols) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + wm0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (co < p.Cout) {
                    if (p.atomic) atomicAdd(&out[(long long)co * p.Kcols + col], acc[i][jj][e]);
                    else out[(long long)co * p.Kcols + col] = acc[i][jj][e];
                }
            }
        }

    if (do_bias) {  // block-uniform; As is free after the loop's last barrier
        float* red = &As[0][0];  // needs (256 / AU) * BM = 256 * VA <= 2 * WBK * BM floats
#pragma unroll
        for (int e = 0; e < VA; ++e) red[ak * BM + ac + e] = bsum[e];
        __syncthreads();
        if (t < BM && co0 + t < p.Cout) {
            float sacc = 0.f;
            for (int rr = 0; rr < 256 / AU; ++rr) sacc += red[rr * BM + t];
            if (p.atomic) atomicAdd(&p.bias_slabs[co0 + t], sacc);
            else p.bias_slabs[(long long)bz * p.Cout + co0 + t] = sacc;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Row-resident weight gradient of plain 3x3 convolutions: dW[co][ta][tb][ci] = sum_pixels dY[p][co] * act(x)[p + (ta, tb)][ci].
// A workgroup owns 128 output channels x ONE tap row (ta; its three taps tb) x 32 input channels (wave w: channels 32 w .. 32 w + 31,
// three 32 x 32 accumulator blocks) and walks its pixel slice 16 pixels at a time: the three taps read the same (columns + 2)-wide
// rows of x, loaded once per step (18 pixels instead of 3 x 16) and activated once; every tap's B fragment is that LDS image at an
// immediate offset, and the dY tile is read once for three taps.  24 MFMAs behind 32 scalar LDS reads per wave and step, 10 KB
// staged per 16 pixels: 38 FLOP per staged byte against 16 for the column-tile kernel above.  (All nine taps per workgroup - 144
// accumulator registers per lane, one or two waves per SIMD - was measured and is slower: profiles/r03_o_*.)
// WT = min(W, 16): a step is 16 / WT image rows of WT pixels.
// Host guarantees: plain geometry, 3x3, Cin % 32 == 0, Cout % 128 == 0, H * W >= 16, slices of whole steps, atomic combine.
// -------------------------------------------------------------------------------------------------
template <int WT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_row_kernel(const WgP p) {
    constexpr int BMC = 128, CI = 32, PXS = 16, TRS = PXS / WT, PW = WT + 2, PP = TRS * PW;
    constexpr int P_SZ = PP * CI, A_SZ = PXS * BMC;
    static_assert(PP * 8 <= 256, "one patch quad per thread");
    __shared__ __attribute__((aligned(16))) float Ps[2][P_SZ];
    __shared__ __attribute__((aligned(16))) float As[2][A_SZ];
    const Geo& g = p.g;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, r = lane & 31, h = lane >> 5;
    const int ta = blockIdx.x % 3, ci0 = (blockIdx.x / 3) * CI, co0 = blockIdx.y * BMC;
    const int mbeg = blockIdx.z * p.mper, mend = min(p.M, mbeg + p.mper);
    // ---- x rows: 8 quads per pixel; lane constant relative to (row + ta - 1, column - 1) of the step's first pixel ----
    const int b_bias = (g.Win + 1) * p.Cin * 4;   // shifted into the base pointer: every offset stays >= 0
    const __amdgpu_buffer_rsrc_t rxb = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x - b_bias), 0, p.x_bytes + (unsigned)b_bias, 0x00020000);
    const int pp = t >> 3, quad = t & 7;
    const int pr = pp / PW, pc = pp - pr * PW;
    const int p_dy = pr + ta - 1, p_dx = pc - 1;
    const unsigned p_l = (unsigned)((((pr + ta) * g.Win + pc) * p.Cin + ci0 + quad * 4) * 4);
    const int p_lds = pp < PP ? pp * CI + quad * 4 : -1;
    // ---- dY rows: 32 quads per pixel row (128 channels) ----
    const int a_row = t >> 5, a_cq = (t & 31) * 4;
    unsigned a_v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a_v[i] = (unsigned)(((a_row + 8 * i) * p.Cout + co0 + a_cq) * 4);
    const bool do_bias = p.bias_slabs != nullptr && blockIdx.x == 0;   // block-uniform
    const bool act_b = p.pre_slope != 1.0f;
    f32x4 ra[2], rp;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    auto load_tiles = [&](int mb) __attribute__((always_inline)) {
        const int left = mend - mb;
        const __amdgpu_buffer_rsrc_t ra_rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.dy + (long long)mb * p.Cout), 0, (unsigned)min(left, PXS) * (unsigned)p.Cout * 4u, 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i) ra[i] = buf_load4(ra_rs, a_v[i], 0);
        const int n = mb >> (g.logH + g.logW);
        const int oy = (mb >> g.logW) & (g.H - 1), ox = mb & (g.W - 1);
        const unsigned u = (unsigned)((((n * g.Hin + oy) * g.Win) + ox) * p.Cin * 4);
        const bool v = p_lds >= 0 && (unsigned)(oy + p_dy) < (unsigned)g.Hin && (unsigned)(ox + p_dx) < (unsigned)g.Win;
        rp = buf_load4(rxb, v ? p_l : BUF_OOB, u);
    };
    auto store_tiles = [&](int buf) __attribute__((always_inline)) {
        if (do_bias) {
            asm volatile("" ::: "memory");   // a real branch (see conv_wgrad_kernel)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) bsum[e] += ra[i][e];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(&As[buf][(a_row + 8 * i) * BMC + a_cq]) = ra[i];
        if (act_b) {
#pragma unroll
            for (int e = 0; e < 4; ++e) rp[e] = __builtin_amdgcn_fmed3f(rp[e], rp[e] * p.pre_slope, p.pos_inf);
        }
        if (p_lds >= 0) *reinterpret_cast<f32x4*>(&Ps[buf][p_lds]) = rp;
    };
    f32x16 acc[3];
#pragma unroll
    for (int tb = 0; tb < 3; ++tb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[tb][e] = 0.f;
    const int nk = (mend > mbeg) ? (mend - mbeg + PXS - 1) / PXS : 0;
    if (nk > 0) {
        load_tiles(mbeg);
        store_tiles(0);
    }
    __syncthreads();
    const float* a_rd = &As[0][0] + h * BMC + wv * 32 + r;
    const float* b_rd = &Ps[0][0] + h * CI + r;
    auto kstep = [&](int ks, auto BUFC, auto MAINC) __attribute__((always_inline)) {
        constexpr int buf = decltype(BUFC)::value;
        constexpr bool MAIN = decltype(MAINC)::value;
        if (MAIN || ks + 1 < nk) load_tiles(mbeg + (ks + 1) * PXS);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kp = 0; kp < PXS / 2; ++kp) {
            const float a = a_rd[buf * A_SZ + 2 * kp * BMC];
            constexpr int dummy = 0; (void)dummy;
            const int pos = ((2 * kp) / WT) * PW + (2 * kp) % WT;   // patch position of pixel 2 kp for tap column 0; pixel 2 kp + 1 is the next one
#pragma unroll
            for (int tb = 0; tb < 3; ++tb) {
                const float b = b_rd[buf * P_SZ + (pos + tb) * CI];
                acc[tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[tb], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MAIN || ks + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    };
    int ks = 0;
    for (; ks + 2 < nk; ks += 2) {
        kstep(ks, std::integral_constant<int, 0>(), std::true_type());
        kstep(ks + 1, std::integral_constant<int, 1>(), std::true_type());
    }
    for (; ks + 1 < nk; ks += 2) {
        kstep(ks, std::integral_constant<int, 0>(), std::false_type());
        kstep(ks + 1, std::integral_constant<int, 1>(), std::false_type());
    }
    if (ks < nk) kstep(ks, std::integral_constant<int, 0>(), std::false_type());
    // out[co][ta][tb][ci]: each accumulator register is two 128-byte row segments per wave
#pragma unroll
    for (int tb = 0; tb < 3; ++tb) {
        const int col = (ta * 3 + tb) * p.Cin + ci0 + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = co0 + wv * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            atomicAdd(&p.slabs[(long long)co * p.Kcols + col], acc[tb][e]);
        }
    }
    if (do_bias) {   // As is free after the loop's last barrier
        float* red = &As[0][0];
#pragma unroll
        for (int e = 0; e < 4; ++e) red[a_row * BMC + a_cq + e] = bsum[e];
        __syncthreads();
        if (t < BMC) {
            float sacc = 0.f;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) sacc += red[rr * BMC + t];
            atomicAdd(&p.bias_slabs[co0 + t], sacc);
        }
    }
}

#include "conv_f16.inc"

// -------------------------------------------------------------------------------------------------
// folded weights: F[co][a][b][ci] = sum_{dh,dw in {0,1}} W[co][a-dh][b-dw][ci],  a, b in [0, K]
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fold_weights_kernel(const float* __restrict__ w, float* __restrict__ f, int Cout, int Cin, int K) {
    const int KF = K + 1;
    const long long n = (long long)Cout * KF * KF * Cin;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int ci = (int)(i % Cin);
        long long rr = i / Cin;
        const int b = (int)(rr % KF); rr /= KF;
        const int a = (int)(rr % KF);
        const int co = (int)(rr / KF);
        float s = 0.f;
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int dw = 0; dw < 2; ++dw) {
                const int kh = a - dh, kw = b - dw;
                if (kh >= 0 && kh < K && kw >= 0 && kw < K) s += w[(((long long)co * K + kh) * K + kw) * Cin + ci];
            }
        f[i] = s;
    }
}

// all folds of a model in one launch: block b works on chunk tab[2b+1] (of 65536 elements) of job tab[2b].
// (Round 4: 32-bit indices, the tap count a compile-time constant - its divisions become multiplies - and 16-byte loads / stores
// along the channels when Cin % 4 == 0: the first version's three 64-bit divisions per ELEMENT made the two launches of a step
// 0.53 ms at the head of the forward passes, ~4x the time of the bytes they move: profiles/r04_c_fold_weights_kernel.txt.)
template <int KF_>   // taps per dimension of the folded kernel (K + 1); 0 = run-time value
__device__ __forceinline__ void fold_chunk(const gim_fold_job& jb, unsigned lo, unsigned hi) {
    const unsigned K = (unsigned)jb.KH, KF = KF_ ? (unsigned)KF_ : K + 1u, Cin = (unsigned)jb.Cin;
    if ((Cin & 3u) == 0 && (((uintptr_t)jb.w | (uintptr_t)jb.f) & 15) == 0) {
        const unsigned Cq = Cin >> 2;
        const bool pow2 = (Cq & (Cq - 1u)) == 0;
        const unsigned sh = 31u - (unsigned)__builtin_clz(Cq);
        const f32x4* __restrict__ w4 = reinterpret_cast<const f32x4*>(jb.w);
        f32x4* __restrict__ f4 = reinterpret_cast<f32x4*>(jb.f);
        for (unsigned q = (lo >> 2) + threadIdx.x; q < (hi >> 2); q += 256) {
            const unsigned rr = pow2 ? q >> sh : q / Cq, cq = pow2 ? q & (Cq - 1u) : q - rr * Cq;
            const unsigned r2 = rr / KF, b = rr - r2 * KF;
            const unsigned co = r2 / KF, a = r2 - co * KF;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (unsigned dh = 0; dh < 2; ++dh)
#pragma unroll
                for (unsigned dw = 0; dw < 2; ++dw) {
                    const unsigned kh = a - dh, kw = b - dw;   // (unsigned wrap-around = out of range)
                    if (kh < K && kw < K) acc += w4[((co * K + kh) * K + kw) * Cq + cq];
                }
            f4[q] = acc;
        }
        return;
    }
    for (unsigned i = lo + threadIdx.x; i < hi; i += 256) {
        const unsigned rr = i / Cin, ci = i - rr * Cin;
        const unsigned r2 = rr / KF, b = rr - r2 * KF;
        const unsigned co = r2 / KF, a = r2 - co * KF;
        float acc = 0.f;
#pragma unroll
        for (unsigned dh = 0; dh < 2; ++dh)
#pragma unroll
            for (unsigned dw = 0; dw < 2; ++dw) {
                const unsigned kh = a - dh, kw = b - dw;
                if (kh < K && kw < K) acc += jb.w[((co * K + kh) * K + kw) * Cin + ci];
            }
        jb.f[i] = acc;
    }
}

__global__ __launch_bounds__(256) void fold_weights_batched_kernel(const gim_fold_job* __restrict__ jobs, const int* __restrict__ tab) {
    const gim_fold_job jb = jobs[tab[2 * blockIdx.x]];
    const unsigned KF = (unsigned)jb.KH + 1u;
    const unsigned n = (unsigned)jb.Cout * KF * KF * (unsigned)jb.Cin;     // < 2^31: checked on the host
    const unsigned lo = (unsigned)tab[2 * blockIdx.x + 1] * 65536u, hi = lo + 65536u < n ? lo + 65536u : n;
    if (KF == 4) fold_chunk<4>(jb, lo, hi);            // 3 x 3 convolutions
    else if (KF == 10) fold_chunk<10>(jb, lo, hi);     // 9 x 9
    else if (KF == 2) fold_chunk<2>(jb, lo, hi);       // 1 x 1
    else fold_chunk<0>(jb, lo, hi);
}

extern "C" int gim_conv2d_fold_weights_batched(const gim_fold_job* jobs, const int32_t* tab, int n_blocks, void* stream) {
    GIM_CHECK_ARG(jobs && tab && n_blocks > 0, "fold_weights_batched: bad args");
    hipLaunchKernelGGL(fold_weights_batched_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, jobs, (const int*)tab);
    return gim_check_launch("gim_conv2d_fold_weights_batched");
}

extern "C" int gim_conv2d_fold_weights(const float* w, float* f, int Cout, int Cin, int KH, void* stream) {
    GIM_CHECK_ARG(w && f && Cout > 0 && Cin > 0 && KH > 0 && (KH & 1), "fold_weights: bad args");
    const long long n = (long long)Cout * (KH + 1) * (KH + 1) * Cin;
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fold_weights_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, w, f, Cout, Cin, KH);
    return gim_check_launch("gim_conv2d_fold_weights");
}

// -------------------------------------------------------------------------------------------------
// transposed weights for the k-contiguous dgrad: WT[ci][a][b][co] = W[co][a][b][ci]  (W plain or folded, T = KF*KF taps)
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_weights_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin, int T) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = co0 + ty + 8 * r, ci = ci0 + tx;
        tile[ty + 8 * r][tx] = (co < Cout && ci < Cin) ? w[((long long)co * T + tap) * Cin + ci] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ci = ci0 + ty + 8 * r, co = co0 + tx;
        if (ci < Cin && co < Cout) wt[((long long)ci * T + tap) * Cout + co] = tile[tx][ty + 8 * r];
    }
}

extern "C" int gim_conv2d_transpose_weights(const float* w, float* wt, int Cout, int Cin, int KF, void* stream) {
    GIM_CHECK_ARG(w && wt && Cout > 0 && Cin > 0 && KF > 0 && KF * KF <= 65535, "transpose_weights: bad args");
    hipLaunchKernelGGL(transpose_weights_kernel, dim3((Cin + 31) / 32, (Cout + 31) / 32, KF * KF), dim3(256), 0, (hipStream_t)stream,
                       w, wt, Cout, Cin, KF * KF);
    return gim_check_launch("gim_conv2d_transpose_weights");
}

// -------------------------------------------------------------------------------------------------
// x-folded transposed weights for the gradient w.r.t. images (<= 8 input channels).  J horizontally adjacent dx pixels share all
// but J - 1 of their K input columns, so the J * Cin values dx[y][J x' + j][ci] are the output CHANNELS of one stride-(1, J)
// convolution of dy with K x (K + J - 1) taps - and [N, H, W / J, J * Cin] IS [N, H, W, Cin] in memory:
//   WX[(j, ci)][a][u][co] = W[co][K - 1 - a][K - 1 - (u - j)][ci]  if 0 <= u - j < K, else 0     (taps already flipped)
// The 16-column MFMA tile then carries 12 useful columns (Cin = 3: J = 4, Cin = 6: J = 2) instead of 3 or 6.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void xfold_weights_kernel(const float* __restrict__ w, float* __restrict__ wx, int Cout, int Cin, int K, int J) {
    const int KW = K + J - 1;
    const long long total = (long long)J * Cin * K * KW * Cout;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int co = (int)(i % Cout);
        long long r = i / Cout;
        const int u = (int)(r % KW); r /= KW;
        const int a = (int)(r % K); r /= K;
        const int ci = (int)(r % Cin), j = (int)(r / Cin);
        const int b = u - j;
        wx[i] = (b >= 0 && b < K) ? w[(((long long)co * K + (K - 1 - a)) * K + (K - 1 - b)) * Cin + ci] : 0.f;
    }
}

extern "C" int gim_conv2d_xfold_weights(const float* w, float* wx, int Cout, int Cin, int KH, int J, void* stream) {
    GIM_CHECK_ARG(w && wx && Cout > 0 && Cin > 0 && KH > 0 && (KH & 1) && J >= 2 && (J & (J - 1)) == 0, "xfold_weights: bad args");
    const long long total = (long long)J * Cin * KH * (KH + J - 1) * Cout;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(xfold_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wx, Cout, Cin, KH, J);
    return gim_check_launch("gim_conv2d_xfold_weights");
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
static int check_shape(const gim_conv_shape* s) {
    GIM_CHECK_ARG(s != nullptr, "conv: null shape");
    GIM_CHECK_ARG(s->N > 0 && s->Cin > 0 && s->Cout > 0, "conv: non-positive dims");
    GIM_CHECK_ARG(s->KH > 0 && (s->KH & 1), "conv: KH must be odd");
    GIM_CHECK_ARG(s->ups == 0 || s->ups == 1, "conv: ups must be 0 or 1");
    GIM_CHECK_ARG(ilog2_exact(s->H) >= 0 && ilog2_exact(s->W) >= 0, "conv: H and W must be powers of two");
    GIM_CHECK_ARG(!s->ups || (s->H >= 2 && s->W >= 2), "conv: ups needs H, W >= 2");
    GIM_CHECK_ARG(!s->pool || (!s->ups && s->H >= 2 && s->W >= 2 && s->wfold), "conv: pool needs ups == 0, H, W >= 2 and folded weights");
    GIM_CHECK_ARG(!s->wfold || s->pool || s->ups, "conv: wfold only with pool or ups");
    GIM_CHECK_ARG((long long)s->N * s->H * s->W < (1ll << 31), "conv: too many output pixels");
    GIM_CHECK_ARG(s->prec == 0 || s->prec == 1, "conv: prec must be 0 (fp32 MFMA) or 1 (fp16 operands, fp32 accumulate)");
    return GIM_OK;
}

static void geo_grid(Geo& g, int N, int H, int W) {
    g.N = N; g.H = H; g.W = W; g.logH = ilog2_exact(H); g.logW = ilog2_exact(W);
}

static Geo geo_plain(const gim_conv_shape* s, bool flip) {
    Geo g{};
    const int pad = (s->KH - 1) / 2;
    geo_grid(g, s->N, s->H, s->W);
    g.ups = flip ? 0 : s->ups;
    g.Hin = s->H >> g.ups; g.Win = s->W >> g.ups;
    g.s_in = g.s_in_x = 1; g.off_y = g.off_x = -pad; g.Th = g.Tw = g.KF = g.KFw = s->KH;
    g.wa_base = g.wb_base = flip ? s->KH - 1 : 0;
    g.wa_step = g.wb_step = flip ? -1 : 1;
    g.os = 1;
    return g;
}

// stride-2 gather over the (K+1)^2 folded taps: logical grid (H/2, W/2), gathered image (H, W)
static Geo geo_s2(const gim_conv_shape* s, bool flip) {
    Geo g{};
    const int pad = (s->KH - 1) / 2, KF = s->KH + 1;
    geo_grid(g, s->N, s->H / 2, s->W / 2);
    g.Hin = s->H; g.Win = s->W; g.ups = 0;
    g.s_in = g.s_in_x = 2; g.off_y = g.off_x = -pad; g.Th = g.Tw = g.KF = g.KFw = KF;
    g.wa_base = g.wb_base = flip ? KF - 1 : 0;
    g.wa_step = g.wb_step = flip ? -1 : 1;
    g.os = 1;
    return g;
}

// 4 parity classes: logical grid (H/2, W/2), output image (H, W), gathered image (H/2, W/2)
static Geo geo_pc(const gim_conv_shape* s, int kind) {
    Geo g{};
    geo_grid(g, s->N, s->H / 2, s->W / 2);
    g.Hin = s->H / 2; g.Win = s->W / 2; g.ups = 0;
    g.s_in = g.s_in_x = 1; g.Th = g.Tw = (s->KH + 1) / 2; g.KF = g.KFw = s->KH + 1;
    g.os = 2; g.pc = 1; g.pc_kind = kind; g.pc_K = s->KH;
    return g;
}

// Launch configuration.  No process-wide switches: everything a launch depends on is in its gim_conv_shape (tune_*) or in
// the read-only table below.  (Round 1's environment A/B switches - K step 32, XCD tile orders, channel-group K order, tile /
// split-K forcing - were removed with the measurements recorded in DESIGN.md section 5 and profiles/r01_*.)
//
// Layers whose output tiles do not fill the chip (8x8 and smaller maps, the decoder head, linears) are sliced along K over
// grid.z.  Measured sweep on MI355X (profiles/r01_ksplit_sweep.txt): best when the launch has about two workgroups per CU
// (~512 of the 128x128 tiles, proportionally more of the smaller ones); beyond 6 slices the float atomics of the combine cost
// more than they buy.
#define GIM_SMALL_TILES 24   // below this many 128x128 output tiles a launch uses 64x64 tiles (4x the workgroups, 4x shorter MFMA chain per K step)

// Launch configurations measured per layer shape on an MI355X (tools/conv_autotune.py writes conv_tune_table.inc):
// {kind (0 fwd-style, 1 dgrad-style, 2 wgrad, 4 dgrad on transposed weights), M, Ca, Cb, Ktot, parity classes,
//  tile config, split-K | wgrad slice target}.
// Shapes that are not in the table use the heuristics below.
struct TuneEntry { int kind, M, Ca, Cb, Ktot, pc, tile, ks; };
static const TuneEntry g_tune[] = {
#include "conv_tune_table.inc"
    {-1, 0, 0, 0, 0, 0, 0, 0}};
static const TuneEntry* tune_lookup(int kind, int M, int Ca, int Cb, int Ktot, int pc) {
    for (const TuneEntry* e = g_tune; e->kind >= 0; ++e)
        if (e->kind == kind && e->M == M && e->Ca == Ca && e->Cb == Cb && e->Ktot == Ktot && e->pc == pc) return e;
    return nullptr;
}

// Position-major rows (Geo.pm): the slot -> pixel order of a launch (of each parity class in PC mode): pixels sorted by their set of
// valid taps, so that neighbouring slots - the ones a tile straddles - skip the same taps
static unsigned long long pm_perm_of(const Geo& g) {
    const int P = g.H * g.W, He = g.Hin << g.ups, We = g.Win << g.ups;
    unsigned mask[16];
    int order[16];
    for (int pix = 0; pix < P; ++pix) {
        const int oy = (pix >> g.logW) * g.s_in + g.off_y, ox = (pix & (g.W - 1)) * g.s_in_x + g.off_x;
        unsigned mk = 0;
        for (int ta = 0; ta < g.Th; ++ta)
            for (int tb = 0; tb < g.Tw; ++tb)
                if ((unsigned)(oy + ta) < (unsigned)He && (unsigned)(ox + tb) < (unsigned)We) mk |= 1u << (ta * g.Tw + tb);
        mask[pix] = mk;
        order[pix] = pix;
    }
    for (int i = 1; i < P; ++i)   // stable insertion sort by mask
        for (int j = i; j > 0 && mask[order[j - 1]] > mask[order[j]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    unsigned long long perm = 0;
    for (int sl = 0; sl < P; ++sl) perm |= (unsigned long long)order[sl] << (4 * sl);
    return perm;
}
static void pm_make_perms(Geo& g) {
    if (g.pc) {
        for (int cls = 0; cls < 4; ++cls) {
            Geo c = g;
            geo_select_class(c, cls);
            g.pm_perm_cls[cls] = pm_perm_of(c);
        }
    } else {
        g.pm_perm = pm_perm_of(g);
    }
}

// what the kernel's tap mask leaves of a launch's K steps, in 1/1000 (1000 without position-major rows): for gim_conv_launch_plan
static int pm_valid_permille(const Geo& g0, int M, int BM) {
    if (!g0.pm) return 1000;
    long long valid = 0, total = 0;
    for (int cls = 0; cls < (g0.pc ? 4 : 1); ++cls) {
        Geo g = g0;
        if (g.pc) { geo_select_class(g, cls); g.pm_perm = g0.pm_perm_cls[cls]; }
        const int He = g.Hin << g.ups, We = g.Win << g.ups;
        for (int m0 = 0; m0 < M; m0 += BM) {
            unsigned mask = 0;
            const int last = m0 + BM - 1 < M - 1 ? m0 + BM - 1 : M - 1;
            for (int slot = m0 / g.N; slot <= last / g.N; ++slot) {
                const int pix = pm_pixel(g.pm_perm, slot);
                const int oy = (pix >> g.logW) * g.s_in + g.off_y, ox = (pix & (g.W - 1)) * g.s_in_x + g.off_x;
                for (int ta = 0; ta < g.Th; ++ta)
                    for (int tb = 0; tb < g.Tw; ++tb)
                        if ((unsigned)(oy + ta) < (unsigned)He && (unsigned)(ox + tb) < (unsigned)We) mask |= 1u << (ta * g.Tw + tb);
            }
            valid += __builtin_popcount(mask);
            total += g.Th * g.Tw;
        }
    }
    return (int)((valid * 1000 + total / 2) / total);
}

// split-K factor: explicit (shape->tune_ksplit), else the table's, else enough workgroups to give every CU several, never
// fewer than 8 K-steps per split
static int plan_ksplit(long long wgs, int nk, int tile_area, int want_ks) {
    if (want_ks > 0) return want_ks > nk ? nk : want_ks;
    const long long target = tile_area >= 128 * 128 ? 512 : 1024;
    if (wgs >= target - target / 8 || nk < 16) return 1;
    long long ks = (target + wgs / 2) / wgs;
    if (ks > nk / 8) ks = nk / 8;
    if (ks > 6) ks = 6;
    return ks < 1 ? 1 : (int)ks;
}

// gim_conv_launch_plan: when this thread-local pointer is set, the launchers below record what they WOULD launch
// ({table row found, BM, BN, split-K | wgrad slices, grid x, y, z, matrix path}) and launch nothing.
static thread_local int32_t* t_plan_out = nullptr;
// an argument error found only once the launch configuration is known (set by the launcher, returned by the entry point)
static thread_local bool t_launch_refused = false;

template <int BM, int BN, int TM, int TN, int BMODE, int GEN, int KB = 16>
static void launch_cfg_kb(ConvP p, size_t y_elems, hipStream_t st, bool table_hit) {
    const int gx = (p.M + BM - 1) / BM, gy = (p.Cb + BN - 1) / BN;
    const int ncls = p.g.pc ? 4 : 1;
    const int nk = (p.Ktot + KB - 1) / KB;
#ifndef GIM_NO_PM   // (a second build of the library for same-box A/B runs)
    // small maps: position-major rows, padding taps skipped (Geo.pm) - a tile of BM rows then holds BM / N pixel positions
    if constexpr (GEN == 0) {
        const int taps = p.g.Th * p.g.Tw;
        p.g.pm = (taps > 1 && taps <= 32 && p.g.H * p.g.W <= PM_MAX_PIXELS && p.g.N >= PM_MIN_IMAGES && p.pix == p.Ca && p.Ca % KB == 0) ? 1 : 0;
        if (p.g.pm) pm_make_perms(p.g);
    }
#endif
    p.ksplit = plan_ksplit((long long)gx * gy * ncls, nk, BM * BN, p.tune_ks);
    p.kper = (nk + p.ksplit - 1) / p.ksplit;
    p.ksplit = (nk + p.kper - 1) / p.kper;
    if (t_plan_out) {
        // out[7] bits 8..: the share of the launch's K steps that is SKIPPED, in 1/1000 (position-major rows skip padding taps; else 0)
        const int32_t v[8] = {table_hit ? 1 : 0, BM, BN, p.ksplit, gx, gy, p.ksplit * ncls, (1000 - pm_valid_permille(p.g, p.M, BM)) << 8};
        for (int i = 0; i < 8; ++i) t_plan_out[i] = v[i];
        return;
    }
    if (p.ksplit > 1 && p.post_slope != 1.f) {   // the K slices are combined by addition: no nonlinearity behind them
        gim_set_error("conv fwd: post_slope with a launch that splits K (ask gim_conv_launch_plan first)");
        t_launch_refused = true;
        return;
    }
    if (p.ksplit > 1 && !p.y_zeroed) (void)hipMemsetAsync(p.y, 0, y_elems * sizeof(float), st);
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, TM, TN, BMODE, GEN, KB>), dim3(gx, gy, p.ksplit * ncls), dim3(256), 0, st, p);
}

template <int BM, int BN, int TM, int TN, int BMODE, int GEN>
static void launch_cfg(const ConvP& p, size_t y_elems, hipStream_t st, bool table_hit) {
    launch_cfg_kb<BM, BN, TM, TN, BMODE, GEN>(p, y_elems, st, table_hit);
}

// 64x64 tile (tile code 64), optionally with K step 32 (tile code 6432: fp32 MFMA fast path with Ca % 32 == 0 only): the small
// tile does 8 MFMAs per wave and K step, so the per-step costs (barrier, loop, address update, LDS round trip) weigh twice as
// much as on the 64x128 tile; a 32-deep step halves them at 36 KB of LDS (still 4 workgroups per CU).
template <int BMODE, int GEN>
static void launch_64x64(const ConvP& p, size_t y_elems, hipStream_t st, bool table_hit, bool kb32) {
    if constexpr (GEN == 0) {
        if (kb32 && p.Ca % 32 == 0) {
            launch_cfg_kb<64, 64, 1, 1, BMODE, GEN, 32>(p, y_elems, st, table_hit);
            return;
        }
    }
    launch_cfg<64, 64, 1, 1, BMODE, GEN>(p, y_elems, st, table_hit);
}

// Patch-resident launch (conv_igemm_patch_kernel) of a plain 3x3 convolution on the fast path, when the geometry allows it; the
// tile and split-K choice are the caller's / the table row's (tile code + 20000) / the heuristic's.  Returns false when the launch
// is not eligible (the tap-major kernel runs).  tools/patch_autotune.py compares the two kernels per layer.
template <int BM, int BN, int TM, int TN, int BMODE>
static void launch_patch_cfg(ConvP p, size_t y_elems, hipStream_t st, bool table_hit) {
    const int gx = (p.M + BM - 1) / BM, gy = (p.Cb + BN - 1) / BN;
    const int chunks = p.Ca / 16, T = 9;
    int ks = plan_ksplit((long long)gx * gy, chunks * T, BM * BN, p.tune_ks);
    if (ks > chunks) ks = chunks;
    const int cps = (chunks + ks - 1) / ks;
    p.ksplit = (chunks + cps - 1) / cps;
    p.kper = cps * T;
    if (t_plan_out) {
        const int32_t v[8] = {table_hit ? 1 : 0, BM, BN, p.ksplit, gx, gy, p.ksplit, 1};
        for (int i = 0; i < 8; ++i) t_plan_out[i] = v[i];
        return;
    }
    if (p.ksplit > 1 && p.post_slope != 1.f) {
        gim_set_error("conv fwd: post_slope with a launch that splits K (ask gim_conv_launch_plan first)");
        t_launch_refused = true;
        return;
    }
    if (p.ksplit > 1 && !p.y_zeroed) (void)hipMemsetAsync(p.y, 0, y_elems * sizeof(float), st);
    const int Wt = p.g.W < BM ? p.g.W : BM;
    const int PP = (BM / Wt + 2) * (Wt + 2);
    const int P_SZ = (PP * 20 + 3) & ~3;
    const int B_SZ = (BMODE == 0) ? BN * 20 : 16 * BN;
    const size_t lds = (size_t)(2 * P_SZ + 2 * B_SZ) * sizeof(float);
    hipLaunchKernelGGL((conv_igemm_patch_kernel<BM, BN, TM, TN, BMODE>), dim3(gx, gy, p.ksplit), dim3(256), lds, st, p);
}

template <int BMODE>
static bool launch_patch(const ConvP& p, size_t y_elems, hipStream_t st, int want, bool table_hit) {
    const Geo& g = p.g;
    if (g.pc || g.ups || g.s_in != 1 || g.s_in_x != 1 || g.os != 1 || g.Th != 3 || g.Tw != 3) return false;
    if (p.Ca % 16 != 0 || p.pix != p.Ca || p.Cb < 32 || ((uintptr_t)p.x & 15) || ((uintptr_t)p.w & 15)) return false;
    if (BMODE == 1 && p.Cb % 4 != 0) return false;
    const int HW = g.H * g.W;
    if (HW < 64) return false;
    if (g.W < 2) return false;   // a one-pixel-wide map of >= 128 rows: (BM / W + 2) * (W + 2) patch pixels would exceed the P_PER quads a thread stages
    // tile: the caller's / table's choice, else as launch_igemm's heuristic; 128-row tiles need H * W >= 128 and W <= 64
    int cfg = want;
    if (cfg == 6432) cfg = 64;
    if (cfg != 128 && cfg != 641 && cfg != 1264 && cfg != 64) {
        if (p.Cb > 64) {
            const long long t128 = (long long)((p.M + 127) / 128) * ((p.Cb + 127) / 128);
            cfg = p.M <= 64 ? 641 : (t128 < GIM_SMALL_TILES ? 64 : 641);
        } else {
            cfg = p.M <= 64 ? 64 : 1264;
        }
    }
    if (p.Cb <= 64 && (cfg == 128 || cfg == 641)) cfg = cfg == 128 ? 1264 : 64;
    if ((cfg == 128 || cfg == 1264) && (HW < 128 || g.W > 64)) cfg = cfg == 128 ? 641 : 64;
    if (p.Cb <= 64 && cfg == 641) cfg = 64;
    if (cfg == 128) launch_patch_cfg<128, 128, 2, 2, BMODE>(p, y_elems, st, table_hit);
    else if (cfg == 641) launch_patch_cfg<64, 128, 1, 2, BMODE>(p, y_elems, st, table_hit);
    else if (cfg == 1264) launch_patch_cfg<128, 64, 2, 1, BMODE>(p, y_elems, st, table_hit);
    else launch_patch_cfg<64, 64, 1, 1, BMODE>(p, y_elems, st, table_hit);
    return true;
}

// fp16-operand launch (conv_igemm_f16_kernel): K steps of 32, tile by shape - the kernel is bound by operand traffic, so the largest
// tile that still fills the chip; split-K as for the fp32 kernels.  No table rows (the table was measured on the fp32 kernels).
template <int BM, int BN, int TM, int TN>
static void launch_f16_cfg(ConvP p, size_t y_elems, hipStream_t st) {
    const int gx = (p.M + BM - 1) / BM, gy = (p.Cb + BN - 1) / BN;
    const int ncls = p.g.pc ? 4 : 1;
    const int nk = p.Ktot / 32;
    p.ksplit = plan_ksplit((long long)gx * gy * ncls, nk, BM * BN, p.tune_ks);
    p.kper = (nk + p.ksplit - 1) / p.ksplit;
    p.ksplit = (nk + p.kper - 1) / p.kper;
    if (t_plan_out) {
        const int32_t v[8] = {0, BM, BN, p.ksplit, gx, gy, p.ksplit * ncls, 2};
        for (int i = 0; i < 8; ++i) t_plan_out[i] = v[i];
        return;
    }
    if (p.ksplit > 1 && p.post_slope != 1.f) {
        gim_set_error("conv fwd: post_slope with a launch that splits K (ask gim_conv_launch_plan first)");
        t_launch_refused = true;
        return;
    }
    if (p.ksplit > 1 && !p.y_zeroed) (void)hipMemsetAsync(p.y, 0, y_elems * sizeof(float), st);
    hipLaunchKernelGGL((conv_igemm_f16_kernel<BM, BN, TM, TN>), dim3(gx, gy, p.ksplit * ncls), dim3(256), 0, st, p);
}

// patch-resident fp16 launch of a plain 3x3 convolution (conv_igemm_patch_f16_kernel); split-K in whole 32-channel chunks
template <int BM, int BN, int TM, int TN>
static void launch_patch_f16_cfg(ConvP p, size_t y_elems, hipStream_t st) {
    const int gx = (p.M + BM - 1) / BM, gy = (p.Cb + BN - 1) / BN;
    const int chunks = p.Ca / 32, T = 9;
    int ks = plan_ksplit((long long)gx * gy, chunks * T, BM * BN, p.tune_ks);
    if (ks > chunks) ks = chunks;
    const int cps = (chunks + ks - 1) / ks;
    p.ksplit = (chunks + cps - 1) / cps;
    p.kper = cps * T;
    if (t_plan_out) {
        const int32_t v[8] = {0, BM, BN, p.ksplit, gx, gy, p.ksplit, 2};
        for (int i = 0; i < 8; ++i) t_plan_out[i] = v[i];
        return;
    }
    if (p.ksplit > 1 && p.post_slope != 1.f) {
        gim_set_error("conv fwd: post_slope with a launch that splits K (ask gim_conv_launch_plan first)");
        t_launch_refused = true;
        return;
    }
    if (p.ksplit > 1 && !p.y_zeroed) (void)hipMemsetAsync(p.y, 0, y_elems * sizeof(float), st);
    const int Wt = p.g.W < BM ? p.g.W : BM;
    const int PP = (BM / Wt + 2) * (Wt + 2);
    const int P_SZ = (PP * 40 + 7) & ~7;
    const size_t lds = (size_t)(2 * P_SZ + 2 * BN * 40) * sizeof(_Float16);
    hipLaunchKernelGGL((conv_igemm_patch_f16_kernel<BM, BN, TM, TN>), dim3(gx, gy, p.ksplit), dim3(256), lds, st, p);
}

static bool launch_patch_f16(const ConvP& p, size_t y_elems, hipStream_t st) {
    const Geo& g = p.g;
    if (g.pc || g.ups || g.s_in != 1 || g.s_in_x != 1 || g.os != 1 || g.Th != 3 || g.Tw != 3) return false;
    const int HW = g.H * g.W;
    if (HW < 64 || g.W < 2) return false;
    const long long t128 = (long long)((p.M + 127) / 128) * ((p.Cb + 127) / 128);
    const bool big = HW >= 128 && g.W <= 64 && t128 >= 256;      // 128-row tiles: whole image rows, enough of them to fill the chip
    if (p.Cb > 64) {
        if (big) launch_patch_f16_cfg<128, 128, 2, 2>(p, y_elems, st);
        else if (p.M > 64 && t128 >= GIM_SMALL_TILES) launch_patch_f16_cfg<64, 128, 1, 2>(p, y_elems, st);
        else launch_patch_f16_cfg<64, 64, 1, 1>(p, y_elems, st);
    } else {
        if (big || (HW >= 128 && g.W <= 64 && t128 >= GIM_SMALL_TILES)) launch_patch_f16_cfg<128, 64, 2, 1>(p, y_elems, st);
        else launch_patch_f16_cfg<64, 64, 1, 1>(p, y_elems, st);
    }
    return true;
}

static bool launch_f16(const ConvP& p, size_t y_elems, hipStream_t st) {
    if (p.Ca % 32 != 0 || p.pix != p.Ca || p.Cb < 32 || ((uintptr_t)p.x & 15) || ((uintptr_t)p.w & 15)) return false;
    if (launch_patch_f16(p, y_elems, st)) return true;
    const long long t128 = (long long)((p.M + 127) / 128) * ((p.Cb + 127) / 128) * (p.g.pc ? 4 : 1);
    if (p.Cb > 64) {
        if (p.M <= 64 || t128 < GIM_SMALL_TILES) launch_f16_cfg<64, 64, 1, 1>(p, y_elems, st);
        else if (t128 < 256) launch_f16_cfg<64, 128, 1, 2>(p, y_elems, st);
        else launch_f16_cfg<128, 128, 2, 2>(p, y_elems, st);
    } else {
        if (p.M <= 64 || t128 < GIM_SMALL_TILES) launch_f16_cfg<64, 64, 1, 1>(p, y_elems, st);
        else launch_f16_cfg<128, 64, 2, 1>(p, y_elems, st);
    }
    return true;
}

// tile by shape (largest accumulator block the channel count fills); parallelism for small M comes from split-K.
// tune_tile / tune_ks: the caller's explicit choice (gim_conv_shape.tune_tile / tune_ksplit; tune_tile < 0 = heuristics only),
// else the table row of this shape, else the heuristic.
template <int BMODE, int GEN>
static void launch_igemm(const ConvP& p, size_t y_elems, hipStream_t st) {
    const int M = p.M, Cb = p.Cb;
    if constexpr (BMODE == 0 && GEN == 0) {
        if (p.f16 && launch_f16(p, y_elems, st)) return;
    }
    ConvP pt = p;
    const bool forced = p.tune_tile != 0 || p.tune_ks > 0;
    const TuneEntry* te = forced ? nullptr : tune_lookup(p.tune_kind, M, p.Ca, Cb, p.Ktot, p.g.pc);
    if (te) pt.tune_ks = te->ks;
    const bool hit = te != nullptr;
    int want = p.tune_tile > 0 ? p.tune_tile : (te ? te->tile : 0);   // 0: heuristic
    // tile code + 20000: the patch-resident kernel (plain 3x3 layers); rows and caller choices without it keep the tap-major loop
    // they were tuned on; a launch with neither row nor caller choice takes the patch-resident kernel where the geometry allows it
    const bool patch_row = want >= 20000;
    if (patch_row) want -= 20000;
    if constexpr (GEN == 0) {
        if ((patch_row || (!te && p.tune_tile == 0 && p.tune_ks == 0)) && launch_patch<BMODE>(pt, y_elems, st, want, hit)) return;
    }
    if (Cb > 64) {
        const long long t128 = (long long)((M + 127) / 128) * ((Cb + 127) / 128) * (p.g.pc ? 4 : 1);
        int cfg = want ? want : (M <= 64 ? 641 : (t128 < GIM_SMALL_TILES ? 64 : 641));
        if (cfg == 641) launch_cfg<64, 128, 1, 2, BMODE, GEN>(pt, y_elems, st, hit);
        else if (cfg == 1264) launch_cfg<128, 64, 2, 1, BMODE, GEN>(pt, y_elems, st, hit);
        else if (cfg == 64 || cfg == 6432) launch_64x64<BMODE, GEN>(pt, y_elems, st, hit, cfg == 6432);
        else launch_cfg<128, 128, 2, 2, BMODE, GEN>(pt, y_elems, st, hit);
    } else if (Cb > 32) {
        int cfg = (want == 64 || want == 6432 || want == 1264) ? want : (M <= 64 ? 64 : 1264);
        if (cfg == 64 || cfg == 6432) launch_64x64<BMODE, GEN>(pt, y_elems, st, hit, cfg == 6432);
        else launch_cfg<128, 64, 2, 1, BMODE, GEN>(pt, y_elems, st, hit);
    } else if (Cb > 16) {
        launch_cfg<128, 32, 1, 1, BMODE, GEN>(pt, y_elems, st, hit);
    } else {
        launch_cfg<128, 16, 1, 1, BMODE, GEN>(pt, y_elems, st, hit);   // 16x16x4 MFMA tile for <= 16 output channels
    }
}

extern "C" int gim_conv2d_fwd(const float* x, const float* w, const float* bias, const float* sigma, const float* residual,
                              float* y, const gim_conv_shape* s, void* stream) {
    int rc = check_shape(s);
    if (rc) return rc;
    GIM_CHECK_ARG(x && w && y, "conv fwd: null pointer");
    // 3 -> 3 / 1 -> 1 image layers: direct convolution (conv_tiny.hip) unless the caller chose a launch (tune_* / deterministic mode)
    if (s->tune_tile == 0 && s->tune_ksplit == 0 && gim_tiny_fwd(x, w, bias, sigma, residual, y, s, (hipStream_t)stream, t_plan_out))
        return t_plan_out ? GIM_OK : gim_check_launch("gim_conv2d_fwd");
    {   // operands beyond the 32-bit buffer-offset range: halve the batch (images are independent)
        const size_t xi = (size_t)(s->H >> s->ups) * (s->W >> s->ups) * s->Cin;        // elements per image
        const size_t yi = (size_t)(s->H >> s->pool) * (s->W >> s->pool) * s->Cout;
        const size_t ri = s->res_ups ? yi / 4 : yi;
        if (s->N > 1 && (xi > yi ? xi : yi) * s->N * sizeof(float) > BUF_MAX_BYTES) {
            gim_conv_shape a = *s, b = *s;
            a.N = s->N / 2; b.N = s->N - a.N;
            rc = gim_conv2d_fwd(x, w, bias, sigma, residual, y, &a, stream);
            if (rc) return rc;
            return gim_conv2d_fwd(x + a.N * xi, w, bias, sigma, residual ? residual + a.N * ri : nullptr, y + a.N * yi, &b, stream);
        }
    }
    ConvP p{};
    p.zero = zero_page();
    p.pos_inf = __builtin_inff();
    const bool up_fold = s->ups && s->wfold;
    p.g = s->pool ? geo_s2(s, false) : (up_fold ? geo_pc(s, 0) : geo_plain(s, false));
    p.x = x; p.w = w; p.bias = bias; p.sigma = sigma; p.res = residual; p.mask_x = nullptr; p.y = y;
    p.Ca = s->Cin; p.Cb = s->Cout; p.Cin_w = s->Cin; p.pix = s->Cin;
    p.M = p.g.N * p.g.H * p.g.W; p.Ktot = p.g.Th * p.g.Tw * s->Cin;
    {
        const unsigned long long xb = (unsigned long long)p.g.N * p.g.Hin * p.g.Win * p.Ca * 4ull;
        GIM_CHECK_ARG(xb <= 0x7FFFFFF0ull, "conv: one image of the gathered tensor exceeds 2 GiB (32-bit buffer offsets)");
        p.x_bytes = (unsigned)xb;
    }
    p.pre_slope = s->pre_slope; p.mask_slope = 1.f; p.out_scale = s->pool ? 0.25f : 1.f; p.res_ups = s->res_ups;
    GIM_CHECK_ARG(s->post_slope >= 0.f && s->post_slope <= 1.f, "conv fwd: post_slope must be in [0, 1] (0 or 1 = none)");
    p.post_slope = (s->post_slope > 0.f) ? s->post_slope : 1.f;
    p.tune_kind = 0;
    p.tune_tile = s->tune_tile; p.tune_ks = s->tune_ksplit; p.y_zeroed = s->out_zeroed;
    p.f16 = s->prec == 1;
    const size_t y_elems = (size_t)s->N * (s->H >> s->pool) * (s->W >> s->pool) * s->Cout;
    GIM_CHECK_ARG(y_elems * sizeof(float) <= 0x7FFFFFF0ull, "conv: one image of the output exceeds 2 GiB (32-bit buffer offsets)");
    const bool gen = (s->Cin % BK) != 0 || ((uintptr_t)x & 15) || ((uintptr_t)w & 15);
    t_launch_refused = false;
    if (gen) launch_igemm<0, 1>(p, y_elems, (hipStream_t)stream);
    else launch_igemm<0, 0>(p, y_elems, (hipStream_t)stream);
    if (t_launch_refused) return GIM_E_BADARG;
    if (t_plan_out) return GIM_OK;
    return gim_check_launch("gim_conv2d_fwd");
}

static int dgrad_impl(const float* dy, const float* w, const float* sigma, const float* mask_x, float* dx,
                      const gim_conv_shape* s, void* stream, bool transposed, const float* res_half = nullptr, float res_scale = 0.f) {
    int rc = check_shape(s);
    if (rc) return rc;
    GIM_CHECK_ARG(dy && w && dx, "conv dgrad: null pointer");
    if (!transposed && !res_half && s->tune_tile == 0 && s->tune_ksplit == 0 && gim_tiny_dgrad(dy, w, sigma, mask_x, dx, s, (hipStream_t)stream, t_plan_out))
        return t_plan_out ? GIM_OK : gim_check_launch("gim_conv2d_dgrad");
    const bool up_fold = s->ups && s->wfold;
    GIM_CHECK_ARG(!(mask_x && s->ups && !up_fold), "conv dgrad: mask_x with ups == 1 needs folded weights");
    {
        const size_t yi = (size_t)(s->H >> s->pool) * (s->W >> s->pool) * s->Cout;             // dy elements per image
        const size_t xi = (size_t)(s->H >> (up_fold ? 1 : 0)) * (s->W >> (up_fold ? 1 : 0)) * s->Cin;   // dx (and mask) per image
        if (s->N > 1 && (xi > yi ? xi : yi) * s->N * sizeof(float) > BUF_MAX_BYTES) {
            gim_conv_shape a = *s, b = *s;
            a.N = s->N / 2; b.N = s->N - a.N;
            rc = dgrad_impl(dy, w, sigma, mask_x, dx, &a, stream, transposed, res_half, res_scale);
            if (rc) return rc;
            return dgrad_impl(dy + a.N * yi, w, sigma, mask_x ? mask_x + a.N * xi : nullptr, dx + a.N * xi, &b, stream, transposed,
                              res_half ? res_half + a.N * (xi / 4) : nullptr, res_scale);
        }
    }
    ConvP p{};
    p.zero = zero_page();
    p.pos_inf = __builtin_inff();
    // pool: dx [N,H,W,Cin] from dy [N,H/2,W/2,Cout] by input-parity classes; sub-pixel (ups+wfold): dx
    // [N,H/2,W/2,Cin] directly from dy [N,H,W,Cout] by a stride-2 gather; plain: dx at the conv's resolution
    p.g = s->pool ? geo_pc(s, 1) : (up_fold ? geo_s2(s, true) : geo_plain(s, true));
    p.x = dy; p.w = w; p.bias = nullptr; p.sigma = sigma; p.res = res_half; p.res_scale = res_scale; p.mask_x = mask_x; p.y = dx;
    p.Ca = s->Cout; p.Cb = s->Cin; p.Cin_w = s->Cin; p.pix = s->Cout;
    p.M = p.g.N * p.g.H * p.g.W; p.Ktot = p.g.Th * p.g.Tw * s->Cout;
    {
        const unsigned long long xb = (unsigned long long)p.g.N * p.g.Hin * p.g.Win * p.Ca * 4ull;
        GIM_CHECK_ARG(xb <= 0x7FFFFFF0ull, "conv: one image of the gathered tensor exceeds 2 GiB (32-bit buffer offsets)");
        p.x_bytes = (unsigned)xb;
    }
    p.pre_slope = 1.f; p.mask_slope = s->pre_slope; p.out_scale = s->pool ? 0.25f : 1.f; p.res_ups = 0;
    p.post_slope = 1.f;
    const size_t y_elems = (size_t)s->N * (s->H >> (up_fold ? 1 : 0)) * (s->W >> (up_fold ? 1 : 0)) * s->Cin;
    GIM_CHECK_ARG(y_elems * sizeof(float) <= 0x7FFFFFF0ull, "conv: one image of the output exceeds 2 GiB (32-bit buffer offsets)");
    const bool gen = (s->Cout % BK) != 0 || ((uintptr_t)dy & 15);
    const bool bscalar = (s->Cin % 4) != 0 || ((uintptr_t)w & 15);
    hipStream_t st = (hipStream_t)stream;
    p.tune_kind = 1;
    p.tune_tile = s->tune_tile; p.tune_ks = s->tune_ksplit; p.y_zeroed = s->out_zeroed;
    p.f16 = transposed && s->prec == 1;   // fp16 operands: the k-contiguous (transposed-weights) form only - ops.py routes fp16 dgrads there
    if (transposed) {
        // WT[ci][a][b][co]: the weight rows are k-contiguous (k = (tap, co)), i.e. the forward kernel's operand layout
        GIM_CHECK_ARG(!gen && !((uintptr_t)w & 15), "conv dgrad (transposed weights): Cout % 16 == 0 and 16-byte aligned operands required");
        p.Cin_w = s->Cout;
        p.tune_kind = 4;
        launch_igemm<0, 0>(p, y_elems, st);
        if (t_plan_out) return GIM_OK;
    return gim_check_launch("gim_conv2d_dgrad_t");
    }
    if (gen) { if (bscalar) launch_igemm<1, 3>(p, y_elems, st); else launch_igemm<1, 1>(p, y_elems, st); }
    else     { if (bscalar) launch_igemm<1, 2>(p, y_elems, st); else launch_igemm<1, 0>(p, y_elems, st); }
    if (t_plan_out) return GIM_OK;
    return gim_check_launch("gim_conv2d_dgrad");
}

extern "C" int gim_conv2d_dgrad(const float* dy, const float* w, const float* sigma, const float* mask_x, float* dx,
                                const gim_conv_shape* s, void* stream) {
    return dgrad_impl(dy, w, sigma, mask_x, dx, s, stream, false);
}

// dx = lrelu'(mask_x) * conv^T(dy, w) / sigma + res_scale * nearest_up2(res_half): the input gradient of a ResBlockDown's first conv
// with the gradient of the block's POOLED skip path (res_half [N, H/2, W/2, Cin]; res_scale = 0.25 = the average pool's backward)
// added in the epilogue - the fan-in that a separate pass over the full-resolution tensor did (gim_add_avgpool2_bwd).  Plain stride-1
// convolutions, mask_x required.
extern "C" int gim_conv2d_dgrad_res(const float* dy, const float* w, const float* sigma, const float* mask_x, const float* res_half,
                                    float res_scale, float* dx, const gim_conv_shape* s, void* stream) {
    GIM_CHECK_ARG(s && !s->ups && !s->pool && !s->wfold && s->H >= 2 && s->W >= 2, "conv dgrad (+ half-resolution residual): plain convolutions on maps of >= 2 x 2 only");
    GIM_CHECK_ARG(mask_x && res_half, "conv dgrad (+ half-resolution residual): mask_x and res_half required");
    return dgrad_impl(dy, w, sigma, mask_x, dx, s, stream, false, res_half, res_scale);
}

extern "C" int gim_conv2d_dgrad_t(const float* dy, const float* wt, const float* sigma, const float* mask_x, float* dx,
                                  const gim_conv_shape* s, void* stream) {
    return dgrad_impl(dy, wt, sigma, mask_x, dx, s, stream, true);
}

// dgrad of a plain convolution with <= 8 input channels on x-folded weights (gim_conv2d_xfold_weights): the forward kernel's
// operand path over dy with stride (1, J), K x (K + J - 1) taps and J * Cin output columns.
extern "C" int gim_conv2d_dgrad_xfold(const float* dy, const float* wx, const float* sigma, const float* mask_x, float* dx,
                                      const gim_conv_shape* s, int J, void* stream) {
    int rc = check_shape(s);
    if (rc) return rc;
    GIM_CHECK_ARG(dy && wx && dx, "conv dgrad (x-folded): null pointer");
    GIM_CHECK_ARG(!s->ups && !s->pool && !s->wfold, "conv dgrad (x-folded): plain convolutions only");
    GIM_CHECK_ARG(J >= 2 && (J & (J - 1)) == 0 && s->W % J == 0 && J * s->Cin <= 32, "conv dgrad (x-folded): J must be a power of two dividing W with J * Cin <= 32");
    GIM_CHECK_ARG(s->Cout % BK == 0 && !((uintptr_t)dy & 15) && !((uintptr_t)wx & 15), "conv dgrad (x-folded): Cout % 16 == 0 and 16-byte aligned operands required");
    {
        const size_t yi = (size_t)s->H * s->W * s->Cout, xi = (size_t)s->H * s->W * s->Cin;
        if (s->N > 1 && (xi > yi ? xi : yi) * s->N * sizeof(float) > BUF_MAX_BYTES) {
            gim_conv_shape a = *s, b = *s;
            a.N = s->N / 2; b.N = s->N - a.N;
            rc = gim_conv2d_dgrad_xfold(dy, wx, sigma, mask_x, dx, &a, J, stream);
            if (rc) return rc;
            return gim_conv2d_dgrad_xfold(dy + a.N * yi, wx, sigma, mask_x ? mask_x + a.N * xi : nullptr, dx + a.N * xi, &b, J, stream);
        }
    }
    ConvP p{};
    p.zero = zero_page();
    p.pos_inf = __builtin_inff();
    const int pad = (s->KH - 1) / 2;
    Geo g{};
    geo_grid(g, s->N, s->H, s->W / J);
    g.Hin = s->H; g.Win = s->W; g.ups = 0;
    g.s_in = 1; g.s_in_x = J; g.off_y = g.off_x = -pad;
    g.Th = g.KF = s->KH; g.Tw = g.KFw = s->KH + J - 1;
    g.wa_base = g.wb_base = 0; g.wa_step = g.wb_step = 1;
    g.os = 1;
    p.g = g;
    p.x = dy; p.w = wx; p.bias = nullptr; p.sigma = sigma; p.res = nullptr; p.mask_x = mask_x; p.y = dx;
    p.Ca = s->Cout; p.Cb = J * s->Cin; p.Cin_w = s->Cout; p.pix = s->Cout;
    p.M = g.N * g.H * g.W; p.Ktot = g.Th * g.Tw * s->Cout;
    p.x_bytes = (unsigned)((unsigned long long)s->N * s->H * s->W * s->Cout * 4ull);
    p.pre_slope = 1.f; p.mask_slope = s->pre_slope; p.out_scale = 1.f; p.res_ups = 0; p.post_slope = 1.f;
    const size_t y_elems = (size_t)s->N * s->H * s->W * s->Cin;
    p.tune_kind = 5;   // no table rows: heuristics (or the caller's tune_* fields)
    p.tune_tile = s->tune_tile; p.tune_ks = s->tune_ksplit; p.y_zeroed = s->out_zeroed;
    launch_igemm<0, 0>(p, y_elems, (hipStream_t)stream);
    if (t_plan_out) return GIM_OK;
    return gim_check_launch("gim_conv2d_dgrad_xfold");
}

// -------------------------------------------------------------------------------------------------
// Row-contiguous K for the image layers (<= 8 input channels; plain stride-1 convolutions; K * Cin <= 64): in NHWC memory the K
// taps of one tap ROW of an output pixel are K * Cin CONTIGUOUS floats (pixels x - pad .. x + pad, all channels).  On a zero-padded,
// already activated copy of the image (gim_pad_image) the convolution is therefore an implicit GEMM with K "taps" (the rows) of
// CaP = K * Cin rounded up to 16 "channels" each, whose pixel stride is Cin floats instead of CaP: the fast path's 16-byte
// vector loads and wave-uniform K-step offsets, where the generic-K path gathers scalars with a (tap, channel) decode per element
// (9x9 6->64: 0.35 ms at 58 TFLOP/s; 3x3 3->64: 20-40 TFLOP/s).  The columns beyond K * Cin of a row read the next pixels' data and
// meet zero weights (gim_conv2d_pack_rows_weights); reads beyond the tensor return 0 (buffer bounds).  The same view gives the
// weight gradient its B operand (gim_conv2d_wgrad_rows_acc): dW slot [Cout][K][CaP], un-padded by the batched finish (fold = 3).
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_rows_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int K, int CaP) {
    const int total = Cout * K * CaP;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int c = i % CaP, r = i / CaP;     // r = co * K + ta
        wp[i] = c < K * Cin ? w[r * K * Cin + c] : 0.f;
    }
}

extern "C" int gim_conv2d_pack_rows_weights(const float* w, float* wp, int Cout, int Cin, int KH, void* stream) {
    GIM_CHECK_ARG(w && wp && Cout > 0 && Cin > 0 && KH > 0 && (KH & 1) && KH * Cin <= 64, "pack_rows_weights: bad args");
    const int CaP = (KH * Cin + 15) & ~15, total = Cout * KH * CaP;
    hipLaunchKernelGGL(pack_rows_weights_kernel, dim3((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024), dim3(256), 0, (hipStream_t)stream,
                       w, wp, Cout, Cin, KH, CaP);
    return gim_check_launch("gim_conv2d_pack_rows_weights");
}

// xp[n][y + pad][x + pad][c] = lrelu(x[n][y][x][c], slope), zero border of `pad` pixels: [N, H + 2 pad, W + 2 pad, C]
__global__ __launch_bounds__(256) void pad_image_kernel(const float* __restrict__ x, float* __restrict__ xp, int N, int H, int W, int C, int pad, float slope) {
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const long long total = (long long)N * Hp * Wp * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int xx = (int)(r % Wp) - pad; r /= Wp;
        const int yy = (int)(r % Hp) - pad;
        const int n = (int)(r / Hp);
        float v = 0.f;
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) v = x[(((long long)n * H + yy) * W + xx) * C + c];
        xp[i] = fmaxf(v, v * slope);
    }
}

extern "C" int gim_pad_image(const float* x, float* xp, int N, int H, int W, int C, int pad, float slope, void* stream) {
    GIM_CHECK_ARG(x && xp && N > 0 && H > 0 && W > 0 && C > 0 && pad >= 0 && slope > 0.f && slope <= 1.f, "pad_image: bad args");
    const long long total = (long long)N * (H + 2 * pad) * (W + 2 * pad) * C;
    hipLaunchKernelGGL(pad_image_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)stream,
                       x, xp, N, H, W, C, pad, slope);
    return gim_check_launch("gim_pad_image");
}

// -------------------------------------------------------------------------------------------------
// Sub-pixel convolution to <= 4 output channels (the generators' 9x9 64->3 image layer): for K = 5, 9, 13 ((K - 1) / 2 even) the four
// output-parity classes of conv_KxK(up2(x)) gather the SAME ((K+1)/2)^2 window of the low-resolution x (offset -(K-1)/4), each with
// its own strided subset of the folded taps F.  As four classes of a 3-column GEMM they fill 3 of the 16 columns of the narrowest MFMA
// tile (0.19 ms, 0.11 of peak); stacked, they are ONE plain ((K+1)/2)-tap convolution to 4 * Cout channels - the same executed FLOPs on
// 12 of 16 columns - followed by a depth-to-space copy (gim_depth_to_space2, which also adds the bias):
//   WM[(2 py + px) * Cout + co][ta][tb][ci] = F[co][1 - py + 2 ta][1 - px + 2 tb][ci]
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_subpixel_weights_kernel(const float* __restrict__ f, float* __restrict__ wm, int Cout, int Cin, int K) {
    const int KF = K + 1, T = KF / 2;
    const int total = 4 * Cout * T * T * Cin;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int ci = i % Cin;
        int r = i / Cin;
        const int tb = r % T; r /= T;
        const int ta = r % T; r /= T;
        const int co = r % Cout, cls = r / Cout;
        const int py = cls >> 1, px = cls & 1;
        wm[i] = f[((co * KF + 1 - py + 2 * ta) * KF + 1 - px + 2 * tb) * Cin + ci];
    }
}

extern "C" int gim_conv2d_pack_subpixel_weights(const float* wf, float* wm, int Cout, int Cin, int KH, void* stream) {
    GIM_CHECK_ARG(wf && wm && Cout > 0 && Cin > 0 && KH >= 5 && (KH & 3) == 1, "pack_subpixel_weights: KH must be 5, 9, 13, ...");
    const int T = (KH + 1) / 2, total = 4 * Cout * T * T * Cin;
    hipLaunchKernelGGL(pack_subpixel_weights_kernel, dim3((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024), dim3(256), 0, (hipStream_t)stream,
                       wf, wm, Cout, Cin, KH);
    return gim_check_launch("gim_conv2d_pack_subpixel_weights");
}

static int rows_shape_ok(const gim_conv_shape* s) {
    int rc = check_shape(s);
    if (rc) return rc;
    GIM_CHECK_ARG(!s->ups && !s->pool && !s->wfold && !s->res_ups, "conv rows form: plain stride-1 convolutions only");
    GIM_CHECK_ARG(s->Cin <= 8 && s->KH * s->Cin <= 64 && s->Cout >= 16, "conv rows form: <= 8 input channels, K * Cin <= 64, >= 16 output channels");
    return GIM_OK;
}

static Geo geo_rows(const gim_conv_shape* s) {
    Geo g{};
    const int pad = (s->KH - 1) / 2;
    geo_grid(g, s->N, s->H, s->W);
    g.ups = 0; g.Hin = s->H + 2 * pad; g.Win = s->W + 2 * pad;
    g.s_in = g.s_in_x = 1; g.off_y = g.off_x = 0;
    g.Th = g.KF = s->KH; g.Tw = g.KFw = 1;
    g.wa_base = 0; g.wa_step = 1; g.wb_base = 0; g.wb_step = 0;
    g.os = 1;
    return g;
}

// y = conv(xp, w) / sigma + bias + residual on the padded, activated copy xp = gim_pad_image(x) and wp = gim_conv2d_pack_rows_weights(w)
extern "C" int gim_conv2d_fwd_rows(const float* xp, const float* wp, const float* bias, const float* sigma, const float* residual,
                                   float* y, const gim_conv_shape* s, void* stream) {
    int rc = rows_shape_ok(s);
    if (rc) return rc;
    GIM_CHECK_ARG(xp && wp && y && !((uintptr_t)xp & 15) && !((uintptr_t)wp & 15), "conv fwd (rows form): null or unaligned pointer");
    ConvP p{};
    p.zero = zero_page();
    p.pos_inf = __builtin_inff();
    p.g = geo_rows(s);
    const int CaP = (s->KH * s->Cin + 15) & ~15;
    p.x = xp; p.w = wp; p.bias = bias; p.sigma = sigma; p.res = residual; p.mask_x = nullptr; p.y = y;
    p.Ca = CaP; p.Cb = s->Cout; p.Cin_w = CaP; p.pix = s->Cin;
    p.M = p.g.N * p.g.H * p.g.W; p.Ktot = s->KH * CaP;
    const unsigned long long xb = (unsigned long long)p.g.N * p.g.Hin * p.g.Win * s->Cin * 4ull;
    GIM_CHECK_ARG(xb <= 0x7FFFFFF0ull, "conv fwd (rows form): the padded image batch exceeds 2 GiB (32-bit buffer offsets): split the batch");
    p.x_bytes = (unsigned)xb;
    p.pre_slope = 1.f; p.mask_slope = 1.f; p.out_scale = 1.f; p.res_ups = 0;
    GIM_CHECK_ARG(s->post_slope >= 0.f && s->post_slope <= 1.f, "conv fwd: post_slope must be in [0, 1] (0 or 1 = none)");
    p.post_slope = (s->post_slope > 0.f) ? s->post_slope : 1.f;
    p.tune_kind = 6;   // no table rows: heuristics (or the caller's tune_* fields)
    p.tune_tile = s->tune_tile; p.tune_ks = s->tune_ksplit; p.y_zeroed = s->out_zeroed;
    p.f16 = 0;
    const size_t y_elems = (size_t)s->N * s->H * s->W * s->Cout;
    GIM_CHECK_ARG(y_elems * sizeof(float) <= 0x7FFFFFF0ull, "conv: the output exceeds 2 GiB (32-bit buffer offsets): split the batch");
    t_launch_refused = false;
    launch_igemm<0, 0>(p, y_elems, (hipStream_t)stream);
    if (t_launch_refused) return GIM_E_BADARG;
    if (t_plan_out) return GIM_OK;
    return gim_check_launch("gim_conv2d_fwd_rows");
}

// wgrad roles.  plain: A = dy [N,H,W,Cout], B = gathered x.  pool: A = dy [N,H/2,W/2,Cout], B = x gathered with
// stride 2 over the (K+1)^2 folded taps -> slabs in F layout [Cout][KF][KF][Cin].  sub-pixel (ups + wfold), roles
// swapped: A = leaky_relu(x) [N,H/2,W/2,Cin], B = dy [N,H,W,Cout] gathered with stride 2 -> slabs
// G[Cin][KF][KF][Cout] with G[ci][ta][tb][co] = dF[co][K-ta][K-tb][ci] (gim_wgrad_finish un-transposes).
struct WgPlan { int bm, bn, ns, mper, rows, cols, M, table_hit, bk, patch, patch_target, f16; };

static WgPlan wgrad_plan(const gim_conv_shape* s) {
    WgPlan q{};
    const bool up_fold = s->ups && s->wfold;
    const int KF = s->wfold ? s->KH + 1 : s->KH;
    q.rows = up_fold ? s->Cin : s->Cout;
    q.cols = KF * KF * (up_fold ? s->Cout : s->Cin);
    const long long M = (long long)s->N * (s->H >> (s->wfold ? 1 : 0)) * (s->W >> (s->wfold ? 1 : 0));
    q.M = (int)M;
    // launch choice: the caller's (gim_conv_shape.tune_tile / tune_wgrad), else the table row of this shape, else the heuristic
    int target = s->tune_wgrad > 0 ? s->tune_wgrad : 0;
    int tile = s->tune_tile > 0 ? s->tune_tile : 0;
    if (!target && !tile && s->tune_tile == 0) {
        const int pcw = (s->pool ? 1 : 0) + (up_fold ? 2 : 0);
        const TuneEntry* te = tune_lookup(2, (int)M, q.rows, q.cols, s->KH, pcw);
        if (te) { target = te->ks; tile = te->tile; q.table_hit = 1; }
    }
    if (tile >= 20000) { q.patch = 1; q.patch_target = target; tile -= 20000; }   // row-resident kernel (plain 3x3 layers)
    // fp16 operands (conv_wgrad_f16_kernel): one tile shape, 32 pixels per K step; the table rows were measured on the fp32 kernels
    const bool f16 = s->prec == 1 && q.rows % 4 == 0 && (up_fold ? s->Cout : s->Cin) % 4 == 0 && q.rows >= 32 && q.cols >= 64;
    if (f16) { q.patch = 0; tile = 128; if (!(s->tune_wgrad > 0)) target = 0; q.f16 = 1; q.table_hit = 0; }
    q.bm = q.rows > 64 ? 128 : (q.rows > 32 ? 64 : 32);
    q.bn = (q.bm == 32) ? 128 : (q.cols > 64 ? 128 : 64);
    if (tile == 128) { q.bm = 128; q.bn = 128; }
    else if (tile == 641) { q.bm = 64; q.bn = 128; }
    else if (tile == 1264) { q.bm = 128; q.bn = 64; }
    else if (tile == 64 || tile == 6432) { q.bm = 64; q.bn = 64; }
    else if (tile == 32128) { q.bm = 32; q.bn = 128; }
    // 32-pixel K steps: fp32 path, both operands on 16-byte loads (the launcher falls back to 16 otherwise)
    q.bk = ((f16 || tile == 6432) && q.rows % 4 == 0 && (up_fold ? s->Cout : s->Cin) % 4 == 0) ? 32 : BK;
    const long long tiles = (long long)((q.cols + q.bn - 1) / q.bn) * ((q.rows + q.bm - 1) / q.bm);
    // Heuristic: about four workgroups per CU in total and at least 32 K-steps (512 pixels) per workgroup, so that the float
    // atomics of the combine stay small next to the MFMA work - unless that leaves most CUs idle (1x1 convs and linears on small
    // maps: few tiles, few pixels): a launch with one workgroup per CU runs one wave per SIMD and pays the full load latency every
    // K step (~1.8 us), so short slices on many CUs win although they add more partial tiles (measured: tools/conv_autotune.py).
    // An explicit / table target lifts the 512-pixel floor to 64 (4 K-steps).
    long long minpix = 512;
    // (fp16 kernel: the same slicing - half as many, longer slices were measured 13 % slower over the config-5 step, 43.5 vs 49.8
    //  episodes/s: the kernel is bound by operand traffic and wants the waves in flight, not fewer epilogues)
    if (!target) {
        target = 1024;
        if (tiles * ((M + 511) / 512) < 256) minpix = 128;
    } else {
        minpix = 64;
    }
    long long S = (target + tiles - 1) / tiles;
    const long long maxS = (M + minpix - 1) / minpix;
    if (S > maxS) S = maxS;
    if (S > 1024) S = 1024;
    if (S < 1) S = 1;
    long long mp = (M + S - 1) / S;
    mp = (mp + q.bk - 1) / q.bk * q.bk;
    q.mper = (int)mp;
    q.ns = (int)((M + mp - 1) / mp);
    return q;
}

// The pixel slices of one weight gradient are combined either with float atomics in the kernel epilogue (n_slabs = 1: the
// caller sees ONE slab, no separate reduce pass - at 16 episodes per GPU that pass cost 7 % of the step; the sum order then
// varies from run to run in the last bits) or, deterministically, as gim_conv2d_wgrad_slabs(shape) separate slabs that
// gim_wgrad_finish adds up in a fixed order.  The caller chooses per call through n_slabs.
extern "C" int gim_conv2d_wgrad_slabs(const gim_conv_shape* s) {
    if (check_shape(s)) return GIM_E_BADARG;
    return wgrad_plan(s).ns;
}

template <int VA, int VB, bool FASTB>
static void launch_wgrad(const WgP& p, int bm, int bn, int bk, dim3 g, hipStream_t st) {
    if constexpr (VA == 4 && VB == 4) {
        if (bk == 32 && bm == 64 && bn == 64) {
            hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, 1, 1, 4, 4, FASTB, 32>), g, dim3(256), 0, st, p);
            return;
        }
    }
    if (bm == 128 && bn == 128) hipLaunchKernelGGL((conv_wgrad_kernel<128, 128, 2, 2, VA, VB, FASTB>), g, dim3(256), 0, st, p);
    else if (bm == 128 && bn == 64) hipLaunchKernelGGL((conv_wgrad_kernel<128, 64, 2, 1, VA, VB, FASTB>), g, dim3(256), 0, st, p);
    else if (bm == 64 && bn == 128) hipLaunchKernelGGL((conv_wgrad_kernel<64, 128, 1, 2, VA, VB, FASTB>), g, dim3(256), 0, st, p);
    else if (bm == 64 && bn == 64) hipLaunchKernelGGL((conv_wgrad_kernel<64, 64, 1, 1, VA, VB, FASTB>), g, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<32, 128, 1, 1, VA, VB, FASTB>), g, dim3(256), 0, st, p);
}

// Weight gradient of a 1x1 convolution with <= 8 channels on ONE side (the skip convs on images: 3 -> 64, 6 -> 64, 64 -> 3): an outer
// product of a wide row (64+ channels) and a handful of scalars per pixel, summed over ~10^5 pixels - HBM-bound (the wide tensor is
// read once: 21 MB at 80 images of 32 x 32 x 64), for which the MFMA kernel above needs 48-65 us (scalar gathers of the narrow
// operand, 0.7 TFLOP/s).  Thread = one wide channel x one of 16 pixel phases, eight pixels in flight per thread (the loop is bound by
// load latency, not by arithmetic); the narrow row is one address for the whole wave.  out[co][ci] (+ bias gradient), combined with
// float atomics into the pre-zeroed arena slot like every other slice.
constexpr int NARROW_PARTS = 16, NARROW_MPER = 512, NARROW_U = 8, NARROW_ROWS = NARROW_MPER / NARROW_PARTS;
template <int CN, bool EXACT>   // EXACT: Cn == CN (the narrow rows of a round are one compile-time-strided block); else Cn <= CN
__global__ __launch_bounds__(64 * NARROW_PARTS) void wgrad_1x1_narrow_kernel(const float* __restrict__ wide, const float* __restrict__ nar,
                                                                             float* __restrict__ out, float* __restrict__ bias_out, int M, int Cw,
                                                                             int Cn_, float slope_x, int wide_is_dy) {
    __shared__ float red[NARROW_PARTS][9][64];
    const int Cn = EXACT ? CN : Cn_;
    const int lane = threadIdx.x & 63, part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: the narrow rows go through scalar loads
    const int c = blockIdx.y * 64 + lane, cc = min(c, Cw - 1);
    const int m0 = blockIdx.x * NARROW_MPER, m1 = min(M, m0 + NARROW_MPER);
    float acc[CN], bsum = 0.f;
#pragma unroll
    for (int j = 0; j < CN; ++j) acc[j] = 0.f;
    // wave `part` owns NARROW_ROWS consecutive pixels, NARROW_U per round (M is a multiple of NARROW_U: the host checks): branch-free,
    // all loads of a round issue back to back
    const int mw = m0 + part * NARROW_ROWS, me = min(m1, mw + NARROW_ROWS);
    for (int mb = mw; mb < me; mb += NARROW_U) {
        float wv[NARROW_U], nv[NARROW_U][CN];
        const float* nrow = nar + (long long)mb * Cn;
#pragma unroll
        for (int u = 0; u < NARROW_U; ++u) {
            wv[u] = wide[(long long)(mb + u) * Cw + cc];
#pragma unroll
            for (int j = 0; j < CN; ++j) nv[u][j] = nrow[u * Cn + (EXACT ? j : min(j, Cn - 1))];
        }
#pragma unroll
        for (int u = 0; u < NARROW_U; ++u) {
            float w_ = wv[u];
            bsum += w_;
            if (!wide_is_dy) w_ = fmaxf(w_, w_ * slope_x);          // the wide operand is x: leaky-relu (slope 1 = identity)
#pragma unroll
            for (int j = 0; j < CN; ++j) {
                float n_ = nv[u][j];
                if (wide_is_dy) n_ = fmaxf(n_, n_ * slope_x);
                acc[j] += w_ * n_;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[part][j][lane] = j < CN ? acc[j < CN ? j : 0] : 0.f;
    red[part][8][lane] = bsum;
    __syncthreads();
    // the 16 partials are summed and added to the slot with CONSECUTIVE lanes on consecutive addresses (one cache line per wave-wide
    // atomic): out[c][j] of a wide dy is read across the LDS rows, out[j][c] of a wide x is wave j as it stands; wave 8 = bias
    if (wide_is_dy) {
        const int t = threadIdx.x, cl = t / Cn, j = t - cl * Cn, cg = blockIdx.y * 64 + cl;
        if (t < 64 * Cn && cg < Cw) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < NARROW_PARTS; ++q) v += red[q][j][cl];
            atomicAdd(&out[(long long)cg * Cn + j], v);
        } else if (part == 8 && bias_out && c < Cw) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < NARROW_PARTS; ++q) v += red[q][8][lane];
            atomicAdd(&bias_out[c], v);
        }
    } else if (part < Cn && c < Cw) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < NARROW_PARTS; ++q) v += red[q][part][lane];
        atomicAdd(&out[(long long)part * Cw + c], v);
    }
    // narrow dy (64 -> 3): its bias gradient is the column sum of the narrow rows; the first workgroup column does it, 64 pixel
    // phases per channel folded through LDS
    if (bias_out && !wide_is_dy && blockIdx.y == 0) {
        __syncthreads();
        float sb = 0.f;
        if (part < Cn)
            for (int m = m0 + lane; m < m1; m += 64) sb += nar[(long long)m * Cn + part];
        if (part < 8) red[0][part][lane] = sb;
        __syncthreads();
        if (threadIdx.x < Cn) {
            float v = 0.f;
            for (int q = 0; q < 64; ++q) v += red[0][threadIdx.x][q];
            atomicAdd(&bias_out[threadIdx.x], v);
        }
    }
}

// prezeroed: the caller guarantees slabs / bias_slabs hold zeros (or a partial sum to add to): pixel slices are combined
// with float atomics and nothing is cleared here (gim_conv2d_wgrad_acc).
static int wgrad_impl(const float* dy, const float* x, float* slabs, float* bias_slabs, int n_slabs, const gim_conv_shape* s,
                      void* stream, bool prezeroed, bool rows = false) {
    int rc = check_shape(s);
    if (rc) return rc;
    GIM_CHECK_ARG(dy && x && slabs, "conv wgrad: null pointer");
    WgPlan q = wgrad_plan(s);
    const int CaP = (s->KH * s->Cin + 15) & ~15;
    if (rows) {   // row-contiguous form: K tap rows of CaP columns each; table rows do not apply (other column count)
        gim_conv_shape v = *s;
        v.tune_tile = s->tune_tile ? s->tune_tile : -1;
        v.prec = 0;
        q = wgrad_plan(&v);
        WgPlan q2 = q;
        q2.cols = s->KH * CaP;
        q2.bn = q2.cols > 64 ? 128 : 64;
        if (q2.bm == 32) q2.bn = 128;
        q = q2;
    }
    const bool atomic = prezeroed ? true : (n_slabs == 1 && q.ns > 1);
    if (!prezeroed) GIM_CHECK_ARG(n_slabs == 1 || n_slabs == q.ns, "conv wgrad: n_slabs must be 1 (atomic combine) or gim_conv2d_wgrad_slabs(shape)");
    const bool up_fold = s->ups && s->wfold;
    GIM_CHECK_ARG(!(up_fold && bias_slabs), "conv wgrad: the sub-pixel form does not produce the bias gradient (use gim_colsum)");
    WgP p{};
    p.zero = zero_page();
    p.pos_inf = __builtin_inff();
    if (rows) p.g = geo_rows(s);
    else if (s->pool) p.g = geo_s2(s, false);
    else if (up_fold) p.g = geo_s2(s, false);
    else p.g = geo_plain(s, false);
    if (up_fold) {
        p.dy = x; p.x = dy; p.Cin = s->Cout; p.Cout = s->Cin; p.pre_slope = 1.f; p.a_slope = s->pre_slope; p.pix = s->Cout;
    } else {
        p.dy = dy; p.x = x; p.Cin = s->Cin; p.Cout = s->Cout; p.pre_slope = s->pre_slope; p.a_slope = 1.f; p.pix = s->Cin;
    }
    p.slabs = slabs; p.bias_slabs = bias_slabs;
    p.M = q.M; p.Kcols = q.cols; p.mper = q.mper; p.atomic = atomic ? 1 : 0;
    if (rows) { p.Cin = CaP; p.pix = s->Cin; p.pre_slope = 1.f; }   // columns per tap row; the padded copy is already activated
    {
        const unsigned long long xb = (unsigned long long)p.g.N * p.g.Hin * p.g.Win * p.pix * 4ull;
        GIM_CHECK_ARG(xb <= 0x7FFFFFF0ull, "conv wgrad: gathered tensor larger than 2 GiB (32-bit buffer offsets): split the batch");
        p.x_bytes = (unsigned)xb;
    }
    if (atomic && !prezeroed && !t_plan_out) {
        (void)hipMemsetAsync(slabs, 0, (size_t)q.rows * q.cols * sizeof(float), (hipStream_t)stream);
        if (bias_slabs) (void)hipMemsetAsync(bias_slabs, 0, (size_t)q.rows * sizeof(float), (hipStream_t)stream);
    }
    // 3 -> 3 / 1 -> 1 image layers, slices combined by atomics: the direct kernel (conv_tiny.hip)
    if (!rows && (prezeroed || n_slabs == 1) && s->tune_tile == 0 && s->tune_wgrad == 0 && gim_tiny_shape(s)) {
        if (!prezeroed && !atomic && !t_plan_out) {   // (atomic: the memsets above have run)
            (void)hipMemsetAsync(slabs, 0, (size_t)q.rows * q.cols * sizeof(float), (hipStream_t)stream);
            if (bias_slabs) (void)hipMemsetAsync(bias_slabs, 0, (size_t)q.rows * sizeof(float), (hipStream_t)stream);
        }
        (void)gim_tiny_wgrad_acc(dy, x, slabs, bias_slabs, s, (hipStream_t)stream, t_plan_out);
        return t_plan_out ? GIM_OK : gim_check_launch("gim_conv2d_wgrad");
    }
    // plain 3x3, row-resident kernel: on request (tile code >= 20000 from the caller or the table row; ks / target = workgroups wanted)
    // (W >= 4: the narrowest instantiation walks rows of 4 pixels - a non-square 8 x 2 map passes H * W >= 16 and must not run it)
    if (!rows && q.patch && atomic && s->KH == 3 && !s->wfold && !s->ups && !s->pool && s->Cin % 32 == 0 && s->Cout % 128 == 0 && s->H * s->W >= 16 &&
        s->W >= 4 && !((uintptr_t)dy & 15) && !((uintptr_t)x & 15)) {
        const int tiles = 3 * (s->Cin / 32) * (s->Cout / 128);
        long long S = ((q.patch_target > 0 ? q.patch_target : 1024) + tiles - 1) / tiles;
        const long long maxS = (q.M + 127) / 128;          // at least 8 steps per slice
        if (S > maxS) S = maxS;
        if (S < 1) S = 1;
        long long mp = (q.M + S - 1) / S;
        mp = (mp + 15) / 16 * 16;
        const int nsp = (int)((q.M + mp - 1) / mp);
        if (t_plan_out) {
            const int32_t v[8] = {q.table_hit, 128, 96, nsp, 3 * (s->Cin / 32), s->Cout / 128, nsp, 1};
            for (int i = 0; i < 8; ++i) t_plan_out[i] = v[i];
            return GIM_OK;
        }
        p.mper = (int)mp; p.ns = nsp;
        const dim3 gp(3 * (s->Cin / 32), s->Cout / 128, nsp);
        const int Wt = s->W < 16 ? s->W : 16;
        if (Wt == 16) hipLaunchKernelGGL(conv_wgrad_row_kernel<16>, gp, dim3(256), 0, (hipStream_t)stream, p);
        else if (Wt == 8) hipLaunchKernelGGL(conv_wgrad_row_kernel<8>, gp, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(conv_wgrad_row_kernel<4>, gp, dim3(256), 0, (hipStream_t)stream, p);
        return gim_check_launch("gim_conv2d_wgrad");
    }
    // 1x1 convolution with <= 8 channels on one side, atomic combine: the outer-product kernel (no table row: nothing to choose)
    if (!rows && s->KH == 1 && !s->wfold && !s->ups && !s->pool && atomic && (s->Cin <= 8 || s->Cout <= 8) && (s->Cin >= 16 || s->Cout >= 16) && q.M % NARROW_U == 0 &&
        s->tune_tile == 0 && s->tune_wgrad == 0) {   // an explicit tile / slice choice of the caller means the MFMA kernel
        const bool wide_is_dy = s->Cin <= 8;
        const int Cw = wide_is_dy ? s->Cout : s->Cin, Cn = wide_is_dy ? s->Cin : s->Cout;
        const int nsl = (q.M + NARROW_MPER - 1) / NARROW_MPER;
        if (t_plan_out) {
            const int32_t v[8] = {0, 0, 0, nsl, nsl, (Cw + 63) / 64, 1, 0};
            for (int i = 0; i < 8; ++i) t_plan_out[i] = v[i];
            return GIM_OK;
        }
        const dim3 g(nsl, (Cw + 63) / 64), b(64 * NARROW_PARTS);
        const float *wd = wide_is_dy ? dy : x, *nr = wide_is_dy ? x : dy;
        if (Cn == 3) hipLaunchKernelGGL((wgrad_1x1_narrow_kernel<3, true>), g, b, 0, (hipStream_t)stream, wd, nr, slabs, bias_slabs, q.M, Cw, Cn, s->pre_slope, wide_is_dy ? 1 : 0);
        else if (Cn == 6) hipLaunchKernelGGL((wgrad_1x1_narrow_kernel<6, true>), g, b, 0, (hipStream_t)stream, wd, nr, slabs, bias_slabs, q.M, Cw, Cn, s->pre_slope, wide_is_dy ? 1 : 0);
        else hipLaunchKernelGGL((wgrad_1x1_narrow_kernel<8, false>), g, b, 0, (hipStream_t)stream, wd, nr, slabs, bias_slabs, q.M, Cw, Cn, s->pre_slope, wide_is_dy ? 1 : 0);
        return gim_check_launch("gim_conv2d_wgrad");
    }
    p.ns = q.ns;
    p.xcd = (q.ns >= 64 || (q.ns >= 8 && q.ns % 8 == 0)) ? 1 : 0;   // every XCD gets (nearly) the same number of slices
    dim3 g((q.cols + q.bn - 1) / q.bn, (q.rows + q.bm - 1) / q.bm, p.xcd ? (q.ns + 7) / 8 * 8 : q.ns);   // z padded: wgrad_block
    if (t_plan_out) {
        const int32_t v[8] = {q.table_hit, q.bm, q.bn, q.ns, (int32_t)g.x, (int32_t)g.y, (int32_t)g.z, q.f16 ? 2 : 0};
        for (int i = 0; i < 8; ++i) t_plan_out[i] = v[i];
        return GIM_OK;
    }
    // per operand as the kernel sees it (the sub-pixel form swaps the roles): A = p.dy with p.Cout channels, B = p.x with p.Cin
    const bool va = (p.Cout % 4 == 0) && !((uintptr_t)p.dy & 15);
    const bool vb = (p.Cin % 4 == 0) && !((uintptr_t)p.x & 15);
    const int bk = (q.bk == 32 && va && vb) ? 32 : BK;
    const bool fastb = vb && p.g.ups == 0 && ((p.g.H * p.g.W) & (bk - 1)) == 0;   // a K step stays inside one image
    hipStream_t st = (hipStream_t)stream;
    if (q.f16 && va && vb && bk == 32) {
        if (fastb) hipLaunchKernelGGL(conv_wgrad_f16_kernel<true>, g, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(conv_wgrad_f16_kernel<false>, g, dim3(256), 0, st, p);
        return gim_check_launch("gim_conv2d_wgrad");
    }
    if (va && fastb) launch_wgrad<4, 4, true>(p, q.bm, q.bn, bk, g, st);
    else if (va && vb) launch_wgrad<4, 4, false>(p, q.bm, q.bn, bk, g, st);
    else if (va) launch_wgrad<4, 1, false>(p, q.bm, q.bn, bk, g, st);
    else if (vb) launch_wgrad<1, 4, false>(p, q.bm, q.bn, bk, g, st);
    else launch_wgrad<1, 1, false>(p, q.bm, q.bn, bk, g, st);
    return gim_check_launch("gim_conv2d_wgrad");
}

extern "C" int gim_conv2d_wgrad(const float* dy, const float* x, float* slabs, float* bias_slabs, int n_slabs,
                                const gim_conv_shape* s, void* stream) {
    return wgrad_impl(dy, x, slabs, bias_slabs, n_slabs, s, stream, false);
}

extern "C" int gim_conv2d_wgrad_acc(const float* dy, const float* x, float* acc, float* bias_acc, const gim_conv_shape* s, void* stream) {
    if (check_shape(s) == 0) {   // operands beyond the 32-bit buffer-offset range: halve the batch, both halves ADD into acc
        const size_t yi = (size_t)(s->H >> s->pool) * (s->W >> s->pool) * s->Cout;
        const size_t xi = (size_t)(s->H >> s->ups) * (s->W >> s->ups) * s->Cin;
        if (s->N > 1 && (xi > yi ? xi : yi) * s->N * sizeof(float) > BUF_MAX_BYTES) {
            gim_conv_shape a = *s, b = *s;
            a.N = s->N / 2; b.N = s->N - a.N;
            const int rc = gim_conv2d_wgrad_acc(dy, x, acc, bias_acc, &a, stream);
            if (rc) return rc;
            return gim_conv2d_wgrad_acc(dy + a.N * yi, x + a.N * xi, acc, bias_acc, &b, stream);
        }
    }
    return wgrad_impl(dy, x, acc, bias_acc, 1, s, stream, true);
}

// dW slot [Cout][K][CaP] (+)= dy^T * rows(xp), bias slot += column sums of dy: the weight gradient of the row-contiguous form
// (xp = gim_pad_image(x): padded AND activated); slices combine by float atomics into the caller's pre-zeroed slot, the batched
// finish un-pads it (gim_wgrad_job.fold = 3).
extern "C" int gim_conv2d_wgrad_rows_acc(const float* dy, const float* xp, float* acc, float* bias_acc, const gim_conv_shape* s, void* stream) {
    int rc = rows_shape_ok(s);
    if (rc) return rc;
    GIM_CHECK_ARG(s->Cout % 4 == 0 && dy && xp && !((uintptr_t)dy & 15) && !((uintptr_t)xp & 15), "conv wgrad (rows form): Cout % 4 == 0 and 16-byte aligned operands required");
    return wgrad_impl(dy, xp, acc, bias_acc, 1, s, stream, true, true);
}

// The launch a conv entry point would make for `shape`, without launching anything (tests, tools/conv_autotune.py).
//   kind 0 = gim_conv2d_fwd, 1 = gim_conv2d_dgrad, 2 = gim_conv2d_dgrad_t, 3 = gim_conv2d_wgrad_acc
//   out[8] = {1 if a row of the compiled-in launch table matched this shape, tile rows BM, tile columns BN,
//             split-K factor (wgrad: pixel slices), grid x, grid y, grid z, loop form (0 tap-major, 1 patch-resident, 2 fp16 operands,
//             3 direct image-layer kernel) | skipped share of the K steps in 1/1000 << 8 (position-major rows, Geo.pm)}
extern "C" int gim_conv_launch_plan(const gim_conv_shape* s, int kind, int32_t* out) {
    GIM_CHECK_ARG(s && out && kind >= 0 && kind <= 3, "conv_launch_plan: bad args");
    float* const fake = reinterpret_cast<float*>(uintptr_t(4096));   // aligned, never dereferenced: nothing is launched
    for (int i = 0; i < 8; ++i) out[i] = -1;
    t_plan_out = out;
    int rc;
    if (kind == 0) rc = gim_conv2d_fwd(fake, fake, nullptr, nullptr, nullptr, fake, s, nullptr);
    else if (kind == 1) rc = gim_conv2d_dgrad(fake, fake, nullptr, nullptr, fake, s, nullptr);
    else if (kind == 2) rc = gim_conv2d_dgrad_t(fake, fake, nullptr, nullptr, fake, s, nullptr);
    else rc = gim_conv2d_wgrad_acc(fake, fake, fake, nullptr, s, nullptr);
    t_plan_out = nullptr;
    return rc;
}

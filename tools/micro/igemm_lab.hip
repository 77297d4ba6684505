// Loop-structure laboratory for the forward / dgrad implicit-GEMM kernel (csrc/conv_igemm.hip, fast path, BMODE 0).
// A stand-alone copy of that kernel's K loop on a convolution-shaped problem - Y[m][n] = sum_{tap, c} act(X[m + off(tap)][c]) *
// W[n][tap][c], every X row read by all taps like the rows of an NHWC image - so that loop variants can be built in seconds,
// run without Python, verified against a naive kernel and timed side by side in ONE process (same device, same clocks):
//   VAR 0  register-staged (global -> VGPR -> ds_write_b128), k-rows padded by 4 floats       [the production loop]
//   VAR 1  LDS-DMA (buffer_load ... lds, 16 B per lane), unpadded 128-byte k-rows, XOR-swizzled 16-byte quads on both sides,
//          activation on the fragment after the LDS read; K step 32
// and, for any variant, PRIO (s_setprio by workgroup parity class) and STAMP (s_memtime per phase, per wave; perturbs the loop).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o igemm_lab igemm_lab.hip && ./igemm_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define BUF_OOB 0x80000000u
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct LabP {
    const float* x;
    const float* w;
    float* y;
    int M, N, Ca, T, Wimg;   // K = T * Ca; tap t reads X row m + (t / 3) * Wimg + (t % 3)
    unsigned x_bytes;
    unsigned long long* stamps;   // [workgroups * 4 waves][8]
    float slope, pos_inf;
    int prio;      // 1: s_setprio(blockIdx.x & 3); 2: priority 3 everywhere but in the MFMA blocks; 3: priority 3 in prologue and epilogue only
    int oob_test;  // 1: tap 1 of rows with m % 5 == 0 is "zero padding" (lane offset BUF_OOB)
};

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
typedef __attribute__((address_space(3))) void* lds_vp;

constexpr int lab_lds_bytes(int BM, int BN, int KB, int VAR) { return 2 * (BM + BN) * (VAR == 1 ? KB : KB + 4) * 4; }

template <int BM, int BN, int TM, int TN, int KB, int VAR, bool STAMP>
__global__ __launch_bounds__(256, lab_lds_bytes(BM, BN, KB, VAR) <= 40960 ? 4 : (lab_lds_bytes(BM, BN, KB, VAR) <= 53248 ? 3 : 2)) void lab_kernel(const LabP p) {
    constexpr bool DMA = VAR == 1;
    static_assert(!DMA || KB == 32, "LDS-DMA variant: 128-byte k-rows");
    constexpr int WAVES_N = BN / (32 * TN), WAVES_M = BM / (32 * TM);
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    constexpr int LDK = DMA ? KB : KB + 4;
    constexpr int QPR = KB / 4, RP = 256 / QPR, A_ROWS = BM / RP, B_ROWS = BN / RP;
    static_assert(A_ROWS >= 1 && B_ROWS >= 1 && BM % RP == 0 && BN % RP == 0, "tile rows per pass");
    constexpr int A_SZ = BM * LDK, B_SZ = BN * LDK;
    __shared__ __attribute__((aligned(1024))) float lds[2 * A_SZ + 2 * B_SZ];
    float* As = lds;
    float* Bs = lds + 2 * A_SZ;

    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (p.prio >= 2) __builtin_amdgcn_s_setprio(3);
    if (p.prio == 1) {
        const int pr = __builtin_amdgcn_readfirstlane(blockIdx.x) & 3;
        if (pr == 1) __builtin_amdgcn_s_setprio(1);
        else if (pr == 2) __builtin_amdgcn_s_setprio(2);
        else if (pr == 3) __builtin_amdgcn_s_setprio(3);
    }
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int arow = t / QPR;
    // 16-byte quad of the k-row this thread fetches: DMA - the quad that belongs at LDS position (row, t % 8) of the swizzled image
    const int aq = DMA ? (((t & 7) ^ ((t >> 4) & 7)) * 4) : (t % QPR) * 4;
    const int K = p.T * p.Ca;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0xFFFFFFFFu, 0x00020000);
    unsigned a_base[A_ROWS], a_cur[A_ROWS], b_voff[B_ROWS];
    bool a_pad[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        const int m = min(m0 + arow + RP * i, p.M - 1);
        a_base[i] = (unsigned)((m * p.Ca + aq) * 4);
        a_pad[i] = p.oob_test && (m % 5 == 0);
        a_cur[i] = a_base[i];
    }
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) b_voff[i] = (unsigned)((min(n0 + arow + RP * i, p.N - 1) * K + aq) * 4);
    int k_c0 = 0, k_tap = 0;
    auto set_tap = [&](int tap) {
        const unsigned off = (unsigned)(((tap / 3) * p.Wimg + (tap % 3)) * p.Ca * 4);
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) a_cur[i] = (a_pad[i] && tap == 1) ? BUF_OOB : a_base[i] + off;
    };
    set_tap(0);
    auto advance = [&]() {
        k_c0 += KB;
        if (k_c0 == p.Ca) { k_c0 = 0; ++k_tap; set_tap(k_tap); }
    };

    f32x4 ra[A_ROWS], rb[B_ROWS];
    auto load_tiles = [&]() {   // VAR 0
        const unsigned sa = (unsigned)(k_c0 * 4), sb = (unsigned)((k_tap * p.Ca + k_c0) * 4);
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) ra[i] = buf_load4(rx, a_cur[i], sa);
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) rb[i] = buf_load4(rw, b_voff[i], sb);
        advance();
    };
    const bool has_act = p.slope != 1.0f;
    auto store_tiles = [&](int buf) {   // VAR 0
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            if (has_act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[i][e] = __builtin_amdgcn_fmed3f(ra[i][e], ra[i][e] * p.slope, p.pos_inf);
            }
            *reinterpret_cast<f32x4*>(&As[buf * A_SZ + (arow + RP * i) * LDK + aq]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) *reinterpret_cast<f32x4*>(&Bs[buf * B_SZ + (arow + RP * i) * LDK + aq]) = rb[i];
    };
    auto dma_tiles = [&](int buf) {   // VAR 1: wave wv writes rows RP * i + 8 * wv .. + 7 of each pass, 1 KiB per instruction
        const unsigned sa = (unsigned)(k_c0 * 4), sb = (unsigned)((k_tap * p.Ca + k_c0) * 4);
#if __HIP_DEVICE_COMPILE__   // (the host pass of hipcc drops a kernel template whose body names this builtin, silently: no launch stub)
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_vp)(As + buf * A_SZ + (RP * i + 8 * wv) * LDK), 16, a_cur[i], sa, 0, 0);
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_vp)(Bs + buf * B_SZ + (RP * i + 8 * wv) * LDK), 16, b_voff[i], sb, 0, 0);
#endif
        advance();
    };

    const int r = lane & 31, h = lane >> 5;
    const int wm0 = (wv / WAVES_N) * 32 * TM, wn0 = (wv % WAVES_N) * 32 * TN;
    // DMA image: 16-byte quad q of row R sits at quad position q ^ ((R >> 1) & 7); rows of a wave's fragment are wm0 + 32 i + r
    // with wm0, 32 i multiples of 16: the swizzle term is a lane constant
    int swoff[KB / 8];
#pragma unroll
    for (int kk = 0; kk < KB / 8; ++kk) swoff[kk] = DMA ? (((2 * kk + h) ^ ((r >> 1) & 7)) * 4) : 8 * kk + 4 * h;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto now = [&]() -> unsigned long long { return STAMP ? __builtin_amdgcn_s_memtime() : 0ull; };
    const unsigned long long c_begin = now(), r_begin = STAMP ? __builtin_amdgcn_s_memrealtime() : 0ull;   // shader cycles / 100 MHz ticks

    const int nk = K / KB;
    if constexpr (DMA) {
        dma_tiles(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        load_tiles();
        store_tiles(0);
    }
    __syncthreads();
    if (p.prio == 3) __builtin_amdgcn_s_setprio(0);
    auto kstep = [&](int ks, auto BUFC, auto MAINC) {
        constexpr int buf = decltype(BUFC)::value;
        constexpr bool MAIN = decltype(MAINC)::value;
        const unsigned long long t0 = now();
        if (MAIN || ks + 1 < nk) {
            if constexpr (DMA) dma_tiles(buf ^ 1);
            else load_tiles();
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t1 = now();
        const float* Ab = As + buf * A_SZ;
        const float* Bb = Bs + buf * B_SZ;
        f32x4 a[KB / 8][TM], b[KB / 8][TN];
#pragma unroll
        for (int kk = 0; kk < KB / 8; ++kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[kk][i] = *reinterpret_cast<const f32x4*>(&Ab[(wm0 + 32 * i + r) * LDK + swoff[kk]]);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[kk][j] = *reinterpret_cast<const f32x4*>(&Bb[(wn0 + 32 * j + r) * LDK + swoff[kk]]);
        }
        if constexpr (DMA) {
            if (has_act) {
#pragma unroll
                for (int kk = 0; kk < KB / 8; ++kk)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) a[kk][i][e] = __builtin_amdgcn_fmed3f(a[kk][i][e], a[kk][i][e] * p.slope, p.pos_inf);
            }
        }
        if constexpr (STAMP) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (TM * TN <= 2 || STAMP) __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t2 = now();
        if (p.prio == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int kk = 0; kk < KB / 8; ++kk)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][i][e], b[kk][j][e], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (p.prio == 2) __builtin_amdgcn_s_setprio(3);
        const unsigned long long t3 = now();
        unsigned long long t4 = t3, t5 = t3;
        if constexpr (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t4 = t5 = now();
        } else {
            if (MAIN || ks + 1 < nk) {
                if constexpr (STAMP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t4 = now(); }
                store_tiles(buf ^ 1);
                if constexpr (STAMP) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t5 = now(); }
            }
        }
        __syncthreads();
        if constexpr (STAMP) {
            const unsigned long long t6 = now();
            st[0] += t1 - t0; st[1] += t2 - t1; st[2] += t3 - t2; st[3] += t5 - t3; st[5] += t6 - t5; st[6] += 1;
        }
    };
    int ks = 0;
    for (; ks + 2 < nk; ks += 2) {
        kstep(ks, std::integral_constant<int, 0>(), std::true_type());
        kstep(ks + 1, std::integral_constant<int, 1>(), std::true_type());
    }
    for (; ks < nk; ks += 2) {
        kstep(ks, std::integral_constant<int, 0>(), std::false_type());
        if (ks + 1 < nk) kstep(ks + 1, std::integral_constant<int, 1>(), std::false_type());
    }
    if (p.prio == 3) __builtin_amdgcn_s_setprio(3);
    const unsigned long long te0 = now();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn0 + 32 * j + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < p.M && col < p.N) p.y[(long long)row * p.N + col] = acc[i][j][e];
            }
        }
    if constexpr (STAMP) {
        st[7] = now() - te0;
        // in-kernel shader clock of this wave's lifetime: cycles per 100 MHz tick, x1000 (slot 5 is re-used: barrier time moves out)
        const unsigned long long dc = now() - c_begin, dr = __builtin_amdgcn_s_memrealtime() - r_begin;
        st[4] = dr ? dc * 1000ull / dr : 0ull;   // (slot 4 = lds-write time is folded into slot 3 for this probe)
        if (lane == 0) {
            unsigned long long* o = p.stamps + ((long long)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv) * 8;
            for (int i = 0; i < 8; ++i) o[i] = st[i];
        }
    }
}

// VAR 3: PERSISTENT workgroups.  The grid is (compute units x 4) workgroups; each walks output tiles
// tile, tile + grid, ... of the staged loop (VAR 0).  PREFETCH: in the LAST K step of a tile the workgroup sets up the next tile's
// addresses and issues ITS first operand loads, which land in LDS behind that step's MFMAs - the next tile's K loop starts without a
// prologue; the epilogue (accumulators copied out first) follows in program order.  Without PREFETCH the loop only removes the
// workgroup launches.  K / KB must be even (every tile then starts on LDS buffer 0: compile-time buffers in the loop).
template <int BM, int BN, int KB, bool PREFETCH>
__global__ __launch_bounds__(256, 4) void lab_persist_kernel(const LabP p, int tiles_m, int ntiles) {
    constexpr int LDK = KB + 4;
    constexpr int QPR = KB / 4, RP = 256 / QPR, A_ROWS = BM / RP, B_ROWS = BN / RP;
    constexpr int A_SZ = BM * LDK, B_SZ = BN * LDK;
    static_assert(BM == 64 && BN == 64, "one 32 x 32 accumulator per wave");
    __shared__ __attribute__((aligned(1024))) float lds[2 * A_SZ + 2 * B_SZ];
    float* As = lds;
    float* Bs = lds + 2 * A_SZ;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int arow = t / QPR, aq = (t % QPR) * 4;
    const int K = p.T * p.Ca, nk = K / KB;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0xFFFFFFFFu, 0x00020000);
    unsigned a_base[A_ROWS], a_cur[A_ROWS], b_voff[B_ROWS];
    int k_c0 = 0, k_tap = 0, m0 = 0, n0 = 0;
    auto set_tap = [&](int tap) {
        const unsigned off = (unsigned)(((tap / 3) * p.Wimg + (tap % 3)) * p.Ca * 4);
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) a_cur[i] = a_base[i] + off;
    };
    auto setup = [&](int tile) {
        m0 = (tile % tiles_m) * BM;
        n0 = (tile / tiles_m) * BN;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) a_base[i] = (unsigned)((min(m0 + arow + RP * i, p.M - 1) * p.Ca + aq) * 4);
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) b_voff[i] = (unsigned)((min(n0 + arow + RP * i, p.N - 1) * K + aq) * 4);
        k_c0 = 0; k_tap = 0;
        set_tap(0);
    };
    f32x4 ra[A_ROWS], rb[B_ROWS];
    auto load_tiles = [&]() {
        const unsigned sa = (unsigned)(k_c0 * 4), sb = (unsigned)((k_tap * p.Ca + k_c0) * 4);
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) ra[i] = buf_load4(rx, a_cur[i], sa);
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) rb[i] = buf_load4(rw, b_voff[i], sb);
        k_c0 += KB;
        if (k_c0 == p.Ca) { k_c0 = 0; ++k_tap; set_tap(k_tap); }
    };
    const bool has_act = p.slope != 1.0f;
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            if (has_act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) ra[i][e] = __builtin_amdgcn_fmed3f(ra[i][e], ra[i][e] * p.slope, p.pos_inf);
            }
            *reinterpret_cast<f32x4*>(&As[buf * A_SZ + (arow + RP * i) * LDK + aq]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) *reinterpret_cast<f32x4*>(&Bs[buf * B_SZ + (arow + RP * i) * LDK + aq]) = rb[i];
    };
    const int r = lane & 31, h = lane >> 5;
    const int wm0 = (wv >> 1) * 32, wn0 = (wv & 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    auto mfmas = [&](auto BUFC) {
        constexpr int buf = decltype(BUFC)::value;
        const float* Ab = As + buf * A_SZ;
        const float* Bb = Bs + buf * B_SZ;
        f32x4 a[KB / 8], b[KB / 8];
#pragma unroll
        for (int kk = 0; kk < KB / 8; ++kk) {
            a[kk] = *reinterpret_cast<const f32x4*>(&Ab[(wm0 + r) * LDK + 8 * kk + 4 * h]);
            b[kk] = *reinterpret_cast<const f32x4*>(&Bb[(wn0 + r) * LDK + 8 * kk + 4 * h]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < KB / 8; ++kk)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][e], b[kk][e], acc, 0, 0, 0);
    };
    auto kstep = [&](auto BUFC) {      // a step with a successor INSIDE its tile
        constexpr int buf = decltype(BUFC)::value;
        load_tiles();
        __builtin_amdgcn_sched_barrier(0);
        mfmas(BUFC);
        __builtin_amdgcn_sched_barrier(0);
        store_tiles(buf ^ 1);
        __syncthreads();
    };
    int tile = blockIdx.x;
    setup(tile);
    load_tiles();
    store_tiles(0);
    __syncthreads();
    for (;;) {
        for (int ks = 0; ks + 2 < nk; ks += 2) {
            kstep(std::integral_constant<int, 0>());
            kstep(std::integral_constant<int, 1>());
        }
        kstep(std::integral_constant<int, 0>());
        // last K step of the tile (LDS buffer 1)
        const int next = tile + (int)gridDim.x;
        const bool has_next = next < ntiles;
        const int em0 = m0, en0 = n0;
        if (PREFETCH && has_next) {
            setup(next);
            load_tiles();
        }
        __builtin_amdgcn_sched_barrier(0);
        mfmas(std::integral_constant<int, 1>());
        __builtin_amdgcn_sched_barrier(0);
        if (PREFETCH && has_next) store_tiles(0);
        __syncthreads();
        // epilogue of the finished tile
        f32x16 out = acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const int col = en0 + wn0 + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = em0 + wm0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (row < p.M && col < p.N) p.y[(long long)row * p.N + col] = out[e];
        }
        if (!has_next) break;
        if (!PREFETCH) {
            setup(next);
            load_tiles();
            store_tiles(0);
            __syncthreads();
        }
        tile = next;
    }
}

template <int KB, bool PREFETCH>
static void run_persist(LabP p, const std::vector<double>& ref, int ref_rows, int wg_per_cu) {
    constexpr int BM = 64, BN = 64;
    if ((p.T * p.Ca / KB) % 2 != 0 || p.Ca % KB != 0) return;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
    // wg_per_cu 0: the balanced grid - every workgroup walks the same number of tiles (ceil(tiles / 1024) of them)
    const int rounds = (ntiles + 1023) / 1024;
    const int grid = wg_per_cu == 0 ? (ntiles + rounds - 1) / rounds : (ntiles < 256 * wg_per_cu ? ntiles : 256 * wg_per_cu);
    CK(hipMemset(p.y, 0xFF, (size_t)p.M * p.N * 4));
    hipLaunchKernelGGL((lab_persist_kernel<BM, BN, KB, PREFETCH>), dim3(grid), dim3(256), 0, 0, p, tiles_m, ntiles);
    CK(hipDeviceSynchronize());
    std::vector<float> y((size_t)ref_rows * p.N);
    CK(hipMemcpy(y.data(), p.y, y.size() * 4, hipMemcpyDeviceToHost));
    double emax = 0, rmax = 0;
    for (size_t i = 0; i < y.size(); ++i) { emax = fmax(emax, fabs((double)y[i] - ref[i])); rmax = fmax(rmax, fabs(ref[i])); }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((lab_persist_kernel<BM, BN, KB, PREFETCH>), dim3(grid), dim3(256), 0, 0, p, tiles_m, ntiles);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((lab_persist_kernel<BM, BN, KB, PREFETCH>), dim3(grid), dim3(256), 0, 0, p, tiles_m, ntiles);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-44s %4dx%-3d KB%-2d  %8.4f ms %7.1f TF  err %.1e  (%d workgroups for %d tiles)\n",
           PREFETCH ? "persistent + cross-tile prefetch" : "persistent (no prefetch)", BM, BN, KB, ms, 2.0 * p.M * p.N * (double)p.T * p.Ca / ms / 1e9, emax / rmax, grid, ntiles);
}

// VAR 4: WAVE-SPECIALISED workgroups.  4 MFMA waves (one 32 x 32 accumulator each, 64 x 64 tile) that only
// ds_read and multiply, plus NL loader waves (NL = 1: A and B tiles; NL = 2: one each) that issue every global load, the activation and
// every ds_write of the workgroup: an MFMA wave's issue queue then holds nothing that has to wait for a gap between other waves' MFMAs
// (the fp32 MFMA occupies the vector ALUs).  One barrier per K step for all waves; two LDS buffers; the loader holds ONE step in
// registers, loaded a full barrier interval before it is written.  K / KB even.
template <int KB, int NL>
__global__ __launch_bounds__(256 + 64 * NL, 4) void lab_ws_kernel(const LabP p) {
    constexpr int BM = 64, BN = 64, LDK = KB + 4, A_SZ = BM * LDK, B_SZ = BN * LDK;
    constexpr int QPR = KB / 4, RPW = 64 / QPR, NR = 64 / RPW;    // a loader wave covers RPW rows per pass, NR passes per tile
    __shared__ __attribute__((aligned(1024))) float lds[2 * A_SZ + 2 * B_SZ];
    float* As = lds;
    float* Bs = lds + 2 * A_SZ;
    const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int K = p.T * p.Ca, nk = K / KB;
    if (wv >= 4) {
        // ---------------- loader
        const bool doA = NL == 1 || wv == 4, doB = NL == 1 || wv == 5;
        const int lrow = lane / QPR, aq = (lane % QPR) * 4;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0xFFFFFFFFu, 0x00020000);
        unsigned a_base[NR], a_cur[NR], b_voff[NR];
        bool a_pad[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int m = min(m0 + lrow + RPW * i, p.M - 1);
            a_base[i] = (unsigned)((m * p.Ca + aq) * 4);
            a_pad[i] = p.oob_test && (m % 5 == 0);
            a_cur[i] = a_base[i];
            b_voff[i] = (unsigned)((min(n0 + lrow + RPW * i, p.N - 1) * K + aq) * 4);
        }
        int k_c0 = 0, k_tap = 0;
        f32x4 ra[NR], rb[NR];
        auto load_tiles = [&]() {
            const unsigned sa = (unsigned)(k_c0 * 4), sb = (unsigned)((k_tap * p.Ca + k_c0) * 4);
            if (doA) {
#pragma unroll
                for (int i = 0; i < NR; ++i) ra[i] = buf_load4(rx, a_cur[i], sa);
            }
            if (doB) {
#pragma unroll
                for (int i = 0; i < NR; ++i) rb[i] = buf_load4(rw, b_voff[i], sb);
            }
            k_c0 += KB;
            if (k_c0 == p.Ca) {
                k_c0 = 0; ++k_tap;
                const unsigned off = (unsigned)(((k_tap / 3) * p.Wimg + (k_tap % 3)) * p.Ca * 4);
#pragma unroll
                for (int i = 0; i < NR; ++i) a_cur[i] = (a_pad[i] && k_tap == 1) ? BUF_OOB : a_base[i] + off;
            }
        };
        const bool has_act = p.slope != 1.0f;
        auto store_tiles = [&](int buf) {
            if (doA) {
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    if (has_act) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) ra[i][e] = __builtin_amdgcn_fmed3f(ra[i][e], ra[i][e] * p.slope, p.pos_inf);
                    }
                    *reinterpret_cast<f32x4*>(&As[buf * A_SZ + (lrow + RPW * i) * LDK + aq]) = ra[i];
                }
            }
            if (doB) {
#pragma unroll
                for (int i = 0; i < NR; ++i) *reinterpret_cast<f32x4*>(&Bs[buf * B_SZ + (lrow + RPW * i) * LDK + aq]) = rb[i];
            }
        };
        load_tiles();
        store_tiles(0);
        load_tiles();                 // step 1 (nk >= 2)
        __syncthreads();              // barrier 0: buffer 0 holds step 0
        for (int ks = 0; ks + 2 < nk; ks += 2) {
            store_tiles(1);           // step ks + 1
            load_tiles();             // step ks + 2
            __syncthreads();
            store_tiles(0);           // step ks + 2
            load_tiles();             // step ks + 3 (exists: nk even)
            __syncthreads();
        }
        store_tiles(1);               // step nk - 1
        __syncthreads();
        return;
    }
    // ---------------- MFMA waves
    const int r = lane & 31, h = lane >> 5;
    const int wm0 = (wv >> 1) * 32, wn0 = (wv & 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    auto step = [&](auto BUFC) {
        constexpr int buf = decltype(BUFC)::value;
        __syncthreads();
        const float* Ab = As + buf * A_SZ;
        const float* Bb = Bs + buf * B_SZ;
        f32x4 a[KB / 8], b[KB / 8];
#pragma unroll
        for (int kk = 0; kk < KB / 8; ++kk) {
            a[kk] = *reinterpret_cast<const f32x4*>(&Ab[(wm0 + r) * LDK + 8 * kk + 4 * h]);
            b[kk] = *reinterpret_cast<const f32x4*>(&Bb[(wn0 + r) * LDK + 8 * kk + 4 * h]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < KB / 8; ++kk)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][e], b[kk][e], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int ks = 0; ks < nk; ks += 2) {
        step(std::integral_constant<int, 0>());
        step(std::integral_constant<int, 1>());
    }
    const int col = n0 + wn0 + r;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < p.M && col < p.N) p.y[(long long)row * p.N + col] = acc[e];
    }
}

template <int KB, int NL>
static void run_ws(LabP p, const std::vector<double>& ref, int ref_rows) {
    constexpr int BM = 64, BN = 64;
    if ((p.T * p.Ca / KB) % 2 != 0 || p.Ca % KB != 0) return;
    const dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN), block(256 + 64 * NL);
    CK(hipMemset(p.y, 0xFF, (size_t)p.M * p.N * 4));
    hipLaunchKernelGGL((lab_ws_kernel<KB, NL>), grid, block, 0, 0, p);
    CK(hipDeviceSynchronize());
    std::vector<float> y((size_t)ref_rows * p.N);
    CK(hipMemcpy(y.data(), p.y, y.size() * 4, hipMemcpyDeviceToHost));
    double emax = 0, rmax = 0;
    for (size_t i = 0; i < y.size(); ++i) { emax = fmax(emax, fabs((double)y[i] - ref[i])); rmax = fmax(rmax, fabs(ref[i])); }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((lab_ws_kernel<KB, NL>), grid, block, 0, 0, p);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((lab_ws_kernel<KB, NL>), grid, block, 0, 0, p);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-44s %4dx%-3d KB%-2d  %8.4f ms %7.1f TF  err %.1e\n", NL == 1 ? "wave-specialised: 4 MFMA waves + 1 loader" : "wave-specialised: 4 MFMA waves + 2 loaders",
           BM, BN, KB, ms, 2.0 * p.M * p.N * (double)p.T * p.Ca / ms / 1e9, emax / rmax);
}

// VAR 2: input patch resident in LDS.  The 3 x 3 taps of a tile of BM consecutive output rows read the BM + 2 * WIMG + 2 input rows
// m0 .. m0 + BM + 2 * WIMG + 1: they are loaded ONCE per 16-channel chunk (K order: channel chunk outer, taps inner) and every tap's
// A fragment is the same LDS image at a compile-time row offset (ds_read immediate) - 9x fewer activation loads, ds_writes,
// activations and address updates per K step; only the weight tile is staged per step.
template <int BM, int BN, int TM, int TN, int WIMG>
__global__ __launch_bounds__(256, 3) void lab_patch_kernel(const LabP p) {
    constexpr int KB = 16, LDK = KB + 4;
    constexpr int WAVES_N = BN / (32 * TN), WAVES_M = BM / (32 * TM);
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    constexpr int PR = BM + 2 * WIMG + 2;             // patch rows
    constexpr int PQ = PR * 4;                        // 16-byte quads per patch
    constexpr int P_PER = (PQ + 255) / 256;           // quads per thread
    constexpr int B_ROWS = BN / 64;                   // weight rows per thread (64 rows per pass: 4 quads per row)
    constexpr int P_SZ = PR * LDK, B_SZ = BN * LDK;
    __shared__ __attribute__((aligned(16))) float lds[2 * P_SZ + 2 * B_SZ];
    float* Ps = lds;
    float* Bs = lds + 2 * P_SZ;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int K = p.T * p.Ca;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0xFFFFFFFFu, 0x00020000);
    unsigned p_voff[P_PER];
    int p_lds[P_PER];
#pragma unroll
    for (int i = 0; i < P_PER; ++i) {
        const int q = t + 256 * i, row = q >> 2, quad = q & 3;
        p_voff[i] = q < PQ ? (unsigned)(((m0 + row) * p.Ca + quad * 4) * 4) : BUF_OOB;
        p_lds[i] = (q < PQ ? row : 0) * LDK + quad * 4;
    }
    const int brow = t >> 2, bq = (t & 3) * 4;
    unsigned b_voff[B_ROWS];
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) b_voff[i] = (unsigned)((min(n0 + brow + 64 * i, p.N - 1) * K + bq) * 4);
    const bool has_act = p.slope != 1.0f;
    f32x4 rp[P_PER], rb[B_ROWS];
    auto load_patch = [&](int c0) {
#pragma unroll
        for (int i = 0; i < P_PER; ++i) rp[i] = buf_load4(rx, p_voff[i], (unsigned)(c0 * 4));
    };
    auto store_patch = [&](int pb) {
#pragma unroll
        for (int i = 0; i < P_PER; ++i) {
            if (has_act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) rp[i][e] = __builtin_amdgcn_fmed3f(rp[i][e], rp[i][e] * p.slope, p.pos_inf);
            }
            if (P_PER * 256 == PQ || t + 256 * i < PQ) *reinterpret_cast<f32x4*>(&Ps[pb * P_SZ + p_lds[i]]) = rp[i];
        }
    };
    auto load_b = [&](int tap, int c0) {
        const unsigned sb = (unsigned)((tap * p.Ca + c0) * 4);
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) rb[i] = buf_load4(rw, b_voff[i], sb);
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) *reinterpret_cast<f32x4*>(&Bs[buf * B_SZ + (brow + 64 * i) * LDK + bq]) = rb[i];
    };
    const int r = lane & 31, h = lane >> 5;
    const int wm0 = (wv / WAVES_N) * 32 * TM, wn0 = (wv % WAVES_N) * 32 * TN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int nchunk = p.Ca / KB;
    load_patch(0);
    load_b(0, 0);
    store_patch(0);
    store_b(0);
    __syncthreads();
    // one K step: tap TAP of chunk `ch` (patch buffer PB, weight buffer BB); prefetches the next step's weights, and - on the
    // first tap - the next chunk's patch, which is written to LDS behind the last tap's MFMAs
    auto kstep = [&](int ch, auto TAPC, auto PBC, auto BBC) {
        constexpr int TAP = decltype(TAPC)::value, PB = decltype(PBC)::value, BB = decltype(BBC)::value;
        const bool more_chunks = ch + 1 < nchunk;
        if (TAP == 0 && more_chunks) load_patch((ch + 1) * KB);
        if (TAP < 8) load_b(TAP + 1, ch * KB);
        else if (more_chunks) load_b(0, (ch + 1) * KB);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int TOFF = ((TAP / 3) * WIMG + (TAP % 3)) * LDK;
        const float* Ab = Ps + PB * P_SZ + TOFF;
        const float* Bb = Bs + BB * B_SZ;
        f32x4 a[2][TM], b[2][TN];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[kk][i] = *reinterpret_cast<const f32x4*>(&Ab[(wm0 + 32 * i + r) * LDK + 8 * kk + 4 * h]);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[kk][j] = *reinterpret_cast<const f32x4*>(&Bb[(wn0 + 32 * j + r) * LDK + 8 * kk + 4 * h]);
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
        if (TAP == 8 && more_chunks) store_patch(PB ^ 1);
        __syncthreads();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto chunk = [&](int ch, auto PBC, auto B0C) {   // nine taps; the weight buffer alternates, starting at B0
        constexpr int B0 = decltype(B0C)::value;
        using Ba = std::integral_constant<int, B0>;
        using Bb_ = std::integral_constant<int, B0 ^ 1>;
        kstep(ch, std::integral_constant<int, 0>(), PBC, Ba());
        kstep(ch, std::integral_constant<int, 1>(), PBC, Bb_());
        kstep(ch, std::integral_constant<int, 2>(), PBC, Ba());
        kstep(ch, std::integral_constant<int, 3>(), PBC, Bb_());
        kstep(ch, std::integral_constant<int, 4>(), PBC, Ba());
        kstep(ch, std::integral_constant<int, 5>(), PBC, Bb_());
        kstep(ch, std::integral_constant<int, 6>(), PBC, Ba());
        kstep(ch, std::integral_constant<int, 7>(), PBC, Bb_());
        kstep(ch, std::integral_constant<int, 8>(), PBC, Ba());
    };
    int ch = 0;
    for (; ch + 1 < nchunk; ch += 2) {   // pairs: after nine steps the weight buffer parity has flipped
        chunk(ch, I0(), I0());
        chunk(ch + 1, I1(), I1());
    }
    if (ch < nchunk) chunk(ch, I0(), I0());
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn0 + 32 * j + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < p.M && col < p.N) p.y[(long long)row * p.N + col] = acc[i][j][e];
            }
        }
}

template <int BM, int BN, int TM, int TN, int WIMG>
static void run_patch(LabP p, const std::vector<double>& ref, int ref_rows) {
    dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN);
    CK(hipMemset(p.y, 0xFF, (size_t)p.M * p.N * 4));
    hipLaunchKernelGGL((lab_patch_kernel<BM, BN, TM, TN, WIMG>), grid, dim3(256), 0, 0, p);
    CK(hipDeviceSynchronize());
    std::vector<float> y((size_t)ref_rows * p.N);
    CK(hipMemcpy(y.data(), p.y, y.size() * 4, hipMemcpyDeviceToHost));
    double emax = 0, rmax = 0;
    for (size_t i = 0; i < y.size(); ++i) { emax = fmax(emax, fabs((double)y[i] - ref[i])); rmax = fmax(rmax, fabs(ref[i])); }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((lab_patch_kernel<BM, BN, TM, TN, WIMG>), grid, dim3(256), 0, 0, p);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((lab_patch_kernel<BM, BN, TM, TN, WIMG>), grid, dim3(256), 0, 0, p);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %4dx%-3d KB16  %8.4f ms %7.1f TF  err %.1e\n", "LDS-resident patch (chunk-major K)", BM, BN, ms / reps,
           2.0 * p.M * p.N * (double)p.T * p.Ca / (ms / reps) / 1e9, emax / rmax);
}

// naive reference for rows [0, rows): double accumulation
__global__ void ref_kernel(const LabP p, int rows, double* out) {
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long long)rows * p.N) return;
    const int m = (int)(id / p.N), n = (int)(id % p.N);
    double s = 0.0;
    for (int tap = 0; tap < p.T; ++tap) {
        if (p.oob_test && (m % 5 == 0) && tap == 1) continue;
        const float* xr = p.x + (long long)(m + (tap / 3) * p.Wimg + (tap % 3)) * p.Ca;
        const float* wr = p.w + ((long long)n * p.T + tap) * p.Ca;
        for (int c = 0; c < p.Ca; ++c) {
            float v = xr[c];
            v = v > 0.f ? v : v * p.slope;
            s += (double)v * (double)wr[c];
        }
    }
    out[id] = s;
}

struct Result { double ms, tf, err; double ph[8]; };

template <int BM, int BN, int TM, int TN, int KB, int VAR, bool STAMP>
static Result run(LabP p, const char* name, const std::vector<double>& ref, int ref_rows, bool print = true) {
    dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN);
    const long long nw = (long long)grid.x * grid.y * 4;
    unsigned long long* stamps = nullptr;
    if (STAMP) { CK(hipMalloc(&stamps, nw * 8 * sizeof(unsigned long long))); CK(hipMemset(stamps, 0, nw * 8 * sizeof(unsigned long long))); }
    p.stamps = stamps;
    CK(hipMemset(p.y, 0xFF, (size_t)p.M * p.N * 4));
    hipLaunchKernelGGL((lab_kernel<BM, BN, TM, TN, KB, VAR, STAMP>), grid, dim3(256), 0, 0, p);
    CK(hipDeviceSynchronize());
    // verify
    std::vector<float> y((size_t)ref_rows * p.N);
    CK(hipMemcpy(y.data(), p.y, y.size() * 4, hipMemcpyDeviceToHost));
    double emax = 0, rmax = 0;
    for (size_t i = 0; i < y.size(); ++i) { emax = fmax(emax, fabs((double)y[i] - ref[i])); rmax = fmax(rmax, fabs(ref[i])); }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((lab_kernel<BM, BN, TM, TN, KB, VAR, STAMP>), grid, dim3(256), 0, 0, p);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((lab_kernel<BM, BN, TM, TN, KB, VAR, STAMP>), grid, dim3(256), 0, 0, p);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    Result res{};
    res.ms = ms / reps;
    res.tf = 2.0 * p.M * p.N * (double)p.T * p.Ca / res.ms / 1e9;
    res.err = emax / rmax;
    if (STAMP) {
        std::vector<unsigned long long> hs(nw * 8);
        CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
        double tot[8] = {0};
        for (long long w = 0; w < nw; ++w) for (int i = 0; i < 8; ++i) tot[i] += (double)hs[w * 8 + i];
        for (int i = 0; i < 6; ++i) res.ph[i] = tot[i] / tot[6];
        res.ph[4] = tot[4] / nw / 10.0;   // MHz (mean over waves): cycles per 10 ns tick x 100
        res.ph[7] = tot[7] / nw;
        CK(hipFree(stamps));
    }
    if (print) {
        printf("%-44s %4dx%-3d KB%-2d  %8.4f ms %7.1f TF  err %.1e", name, BM, BN, KB, res.ms, res.tf, res.err);
        if (STAMP) printf("  cycles/K-step/wave: issue-loads %.0f | lds-read %.0f | mfma %.0f | vmcnt %.0f | in-kernel clock %.0f MHz | barrier %.0f ; epilogue %.0f",
                          res.ph[0], res.ph[1], res.ph[2], res.ph[3], res.ph[4], res.ph[5], res.ph[7]);
        printf("\n");
    }
    return res;
}

int main(int argc, char** argv) {
    struct Shape { int M, N, Ca, T, Wimg; const char* what; };
    const Shape shapes[] = {
        {81920, 128, 64, 9, 32, "80 img 32x32 64->128 3x3 (dominant)"},
        {163840, 128, 64, 9, 32, "160 img 32x32 64->128 3x3"},
        {20480, 256, 128, 9, 16, "80 img 16x16 128->256 3x3"},
        {163840, 64, 64, 16, 32, "160 img 64x64 64->64 3x3 pool-folded (4x4 taps on the pooled grid)"},
        {327680, 128, 512, 9, 32, "320 img 32x32 512->128 3x3 (long K)"},
    };
    for (const Shape& s : shapes) {
        LabP p{};
        p.M = s.M; p.N = s.N; p.Ca = s.Ca; p.T = s.T; p.Wimg = s.Wimg;
        const long long xrows = (long long)s.M + (s.T / 3 + 1) * s.Wimg + 8;
        const size_t xn = (size_t)xrows * s.Ca, wn = (size_t)s.N * s.T * s.Ca;
        std::vector<float> hx(xn), hw(wn);
        unsigned seed = 12345u;
        auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
        for (auto& v : hx) v = rnd();
        for (auto& v : hw) v = rnd() * 0.05f;
        float *dx, *dw, *dy;
        CK(hipMalloc(&dx, xn * 4)); CK(hipMalloc(&dw, wn * 4)); CK(hipMalloc(&dy, (size_t)s.M * s.N * 4));
        CK(hipMemcpy(dx, hx.data(), xn * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dw, hw.data(), wn * 4, hipMemcpyHostToDevice));
        p.x = dx; p.w = dw; p.y = dy; p.x_bytes = (unsigned)(xn * 4); p.pos_inf = INFINITY;
        const int ref_rows = 2048;
        double* dref;
        CK(hipMalloc(&dref, (size_t)ref_rows * s.N * 8));
        printf("== %s: M=%d N=%d K=%d (%.2f GFLOP)\n", s.what, s.M, s.N, s.T * s.Ca, 2.0 * s.M * s.N * s.T * s.Ca / 1e9);
        for (int pass = 0; pass < 3; ++pass) {
            p.slope = pass == 1 ? 1.0f : 0.2f;
            p.oob_test = pass == 0 ? 1 : 0;
            p.prio = 0;
            hipLaunchKernelGGL(ref_kernel, dim3((ref_rows * s.N + 255) / 256), dim3(256), 0, 0, p, ref_rows, dref);
            std::vector<double> ref((size_t)ref_rows * s.N);
            CK(hipMemcpy(ref.data(), dref, ref.size() * 8, hipMemcpyDeviceToHost));
            printf("-- pre-activation slope %.1f%s\n", p.slope, p.oob_test ? ", zero-padding lanes (BUF_OOB) on tap 1" : "");
            for (int round = 0; round < 3; ++round) {
                run<64, 64, 1, 1, 32, 0, false>(p, "staged", ref, ref_rows);
                if (s.N >= 128) run<64, 128, 1, 2, 16, 0, false>(p, "staged", ref, ref_rows);
                run<128, 64, 2, 1, 16, 0, false>(p, "staged", ref, ref_rows);
                run_ws<32, 1>(p, ref, ref_rows);
                run_ws<32, 2>(p, ref, ref_rows);
                run_ws<16, 1>(p, ref, ref_rows);
                if (!p.oob_test) {
                    run_persist<32, false>(p, ref, ref_rows, 4);
                    run_persist<32, true>(p, ref, ref_rows, 4);
                    run_persist<32, true>(p, ref, ref_rows, 0);
                    run_persist<32, true>(p, ref, ref_rows, 2);
                }
                if (round == 0) {
                    run<64, 64, 1, 1, 32, 0, true>(p, "staged, STAMPED", ref, ref_rows);
                    if (s.N >= 128) run<64, 128, 1, 2, 16, 0, true>(p, "staged, STAMPED", ref, ref_rows);
                }
                if (!p.oob_test && s.T == 9) {
                    if (s.Wimg == 32) {
                        run_patch<64, 64, 1, 1, 32>(p, ref, ref_rows);
                        if (s.N >= 128) run_patch<64, 128, 1, 2, 32>(p, ref, ref_rows);
                        run_patch<128, 64, 2, 1, 32>(p, ref, ref_rows);
                    } else if (s.Wimg == 16) {
                        run_patch<64, 64, 1, 1, 16>(p, ref, ref_rows);
                        if (s.N >= 128) run_patch<64, 128, 1, 2, 16>(p, ref, ref_rows);
                        run_patch<128, 64, 2, 1, 16>(p, ref, ref_rows);
                    }
                }
            }
        }
        CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dy)); CK(hipFree(dref));
    }
    return 0;
}

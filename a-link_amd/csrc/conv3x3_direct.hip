// conv3x3_direct.hip — 3x3 / stride 1 / pad 1 NHWC convolution with the INPUT TILE RESIDENT IN LDS.
//
// 95 % of the backbone FLOPs are stride-1 3x3 convolutions (device work behind
// reference code/face_model.py:90).  The generic implicit-GEMM kernel (conv_igemm.hip) re-stages an
// im2col pixel tile for each of the 9 taps, so its L2->LDS traffic (64 FLOP/B at 128x128) sits at the
// per-CU LDS-fill limit.  This kernel is the conv-native form:
//
//   * One workgroup (8 waves) owns R output rows x W columns of ONE image and BN output channels.
//     For each 64-channel input chunk it stages the (R+2) x (W+2) input halo region ONCE into LDS
//     (zero frame included: padding costs nothing and needs no per-tap masks), then walks the 9 taps
//     reading MFMA B-fragments from the SAME region at shifted addresses.  Only the [BN][64] weight
//     tile is streamed per tap (double-buffered LDS-DMA).  L2->LDS traffic: >= 200 FLOP/B.
//   * MFMA tiles are ROW-ALIGNED: a 16-pixel tile is 16 consecutive columns of one image row, so a
//     fragment read touches 16 consecutive 128-B pixel rows of LDS.  Lane->pixel inside a tile is
//     permuted (delta(): even columns on lanes {0-3,12-15}, odd on {4-11}) so that, with the XOR
//     swizzle chunk ^ ((col>>1)&7), every ds_read_b128 lane group hits 16 distinct 16-B slots for
//     ANY tap shift — conflict-free without padding rows.
//   * All tap/tile address arithmetic folds into ds_read immediates: per lane 18 precomputed
//     offsets (3 ky x 3 kx x 2 k-halves) serve every tile and tap; zero VALU per fragment read.
//   * Tiles map 1:1 onto image rows, e.g. stage 3 (14x14) = one image per workgroup: 256 images fill
//     the 256 CUs exactly (the 128x128 im2col tiling gave 3.06 rounds).
//   * Epilogue identical to conv_igemm: folded-BN bias with 9 border classes, PReLU, residual, store
//     of 16 (TCW=4) or 8 (TCW=2) consecutive channels per lane.
#include "alink_common.h"

namespace alink {
namespace {

template <typename T> struct Vec8;
template <> struct Vec8<__bf16>   { typedef bf16x8 type; };
template <> struct Vec8<_Float16> { typedef f16x8 type; };

template <typename T>
__device__ __forceinline__ f32x4 mfma16(typename Vec8<T>::type a, typename Vec8<T>::type b, f32x4 c);
template <>
__device__ __forceinline__ f32x4 mfma16<__bf16>(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mfma16<_Float16>(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void dma16(const void* gsrc, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)gsrc,
        (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// column (0..15) inside a 16-pixel MFMA tile handled by MFMA column lr
__device__ __forceinline__ int delta(int lr) { return lr < 4 ? 2 * lr : (lr < 12 ? 2 * (lr - 4) + 1 : 2 * (lr - 8)); }

// Tile geometry of one instantiation.  A workgroup tile is R rows x TR column-blocks of 16.
template <int WPX, int TPW, int TR>
struct Geo {
    static constexpr bool modeA = (TR % WPX) == 0;           // groups split the column blocks of every row
    static constexpr bool modeB = !modeA && (TR == TPW);     // one row per group
    static constexpr bool modeC = !modeA && !modeB;          // TR == 1: groups split rows
    static_assert(modeA || modeB || (TR == 1), "unsupported tile geometry");
    static constexpr int CPG = modeA ? TR / WPX : 1;
    static constexpr int R = modeA ? TPW / CPG : (modeB ? WPX : TPW * WPX);
    // tile u of wave group g -> (row, column block)
    __device__ static constexpr int row(int g, int u) { return modeA ? u / CPG : (modeB ? g : g * TPW + u); }
    __device__ static constexpr int cb(int g, int u)  { return modeA ? g * CPG + u % CPG : (modeB ? u : 0); }
};

template <int N>
__device__ __forceinline__ void wait_dma_then_barrier() {
    // counted wait: all but the N youngest LDS-DMA instructions of this wave have landed; then the
    // workgroup barrier.  One asm statement with a memory clobber: no LDS access moves across it and
    // hipcc adds no vmcnt(0) of its own (cdna_hip_programming.md §5 "Pipelining across barriers").
    // lgkmcnt(0): the wave's own LDS reads have returned before it arrives — the compiler sinks the MFMAs that
    // consume a step's last fragments (and their wait) below this statement, and a read still queued at the barrier
    // can be overtaken by another wave's DMA into the buffer it is aimed at (conv3x3_linear.hip has the full story).
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int A, int B>
struct IC2 { static constexpr int a = A, b = B; };

// NWB = number of weight buffers in LDS (2; a 3-buffer counted-vmcnt pipeline measured slower, see
// DESIGN.md "what did not work").
// NWV = waves per workgroup: 8 (one workgroup per CU, X double-buffered) or 4 (XDB = false: single X
// buffer, <= 80 KB of LDS, so TWO independent workgroups share a CU and cover each other's DMA waits,
// barriers, prologue and epilogue).
template <typename T, int WPX, int WCO, int TPW, int TCW, int TR, int PITCH, int NWB, int NWV, bool XDB>
__global__ __launch_bounds__(NWV * 64, 2) void conv3x3_direct_kernel(const ConvParams p) {
    typedef typename Vec8<T>::type vec8;
    typedef Geo<WPX, TPW, TR> G;
    constexpr int NT = NWV * 64;
    static_assert(WPX * WCO == NWV, "wave grid");
    static_assert(NWB == 2, "two weight buffers");
    constexpr int R = G::R;
    constexpr int BN = WCO * TCW * 16;
    constexpr int XPIX = (R + 2) * PITCH;                       // pixels in the staged region
    constexpr int XSLOTS = (XPIX * 8 + NT - 1) / NT;            // 16-B DMA slots per thread
    constexpr int XSTRIDE = XSLOTS * NT * 16;                   // bytes per X buffer (every slot in bounds)
    constexpr int WBYTES = BN * 128;
    constexpr int WSLOTS = BN * 8 / NT;
    static_assert(WSLOTS >= 1, "BN >= 64");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wpx = wave / WCO, wco = wave % WCO;
    const int q = lane >> 4, lr = lane & 15;

    const int H = p.H, W = p.W, Cin = p.Cin;
    const int ncc = Cin >> 6;
    const int nk = ncc * 9;
    const bool xdouble = XDB && ncc > 1;
    const int woff0 = (xdouble ? 2 : 1) * XSTRIDE;              // weight buffers follow the X buffers

    const int ntn = p.Cout / BN;
    const int tiles_per_img = (H + R - 1) / R;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % ntn, tile_m = lid / ntn;
    const int n_img = tile_m / tiles_per_img, r0 = (tile_m - n_img * tiles_per_img) * R;
    const int n0 = tile_n * BN;
    const int K = 9 * Cin;

    const T* __restrict__ gin = (const T*)p.in;
    const T* __restrict__ gw = (const T*)p.wgt;
    const T* __restrict__ gz = (const T*)p.zero;

    // ---- X staging: slot s = tid + 512*i -> LDS pixel (s>>3), chunk position (s&7) ----------------
    int xoff[XSLOTS];
#pragma unroll
    for (int i = 0; i < XSLOTS; ++i) {
        const int s = tid + NT * i;
        const int idx = s >> 3, pos = s & 7;
        const int rr = idx / PITCH, pc = idx % PITCH;
        const int c16 = pos ^ ((pc >> 1) & 7);
        const int iy = r0 - 1 + rr, ix = pc - 1;
        const bool ok = idx < XPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        xoff[i] = ok ? ((n_img * H + iy) * W + ix) * Cin + c16 * 8 : -1;
    }
    auto stage_x_slot = [&](int i, int buf, int cc) {
        const T* src = xoff[i] >= 0 ? gin + (xoff[i] + cc * 64) : gz + (lane & 7) * 8;
        dma16(src, smem + buf * XSTRIDE + (NT * i + wave * 64) * 16);
    };
    // ---- W staging.  Weights of direct-variant layers are stored in K-step order
    // [cout][chunk][tap][64], so the source of K-step kt is a per-thread pointer advanced by 64
    // elements per step.  slot s -> row (s>>3), position (s&7); logical chunk = pos ^ ((row>>1)&7).
    const T* wp[WSLOTS];
#pragma unroll
    for (int i = 0; i < WSLOTS; ++i) {
        const int s = tid + NT * i;
        const int row = s >> 3, pos = s & 7;
        wp[i] = gw + ((size_t)(n0 + row) * K + (pos ^ ((row >> 1) & 7)) * 8);
    }
    auto stage_w = [&](int bufoff) {   // stages the next not-yet-staged K-step
#pragma unroll
        for (int i = 0; i < WSLOTS; ++i) {
            dma16(wp[i], smem + woff0 + bufoff + (NT * i + wave * 64) * 16);
            wp[i] += 64;
        }
    };

    // ---- per-lane fragment offsets ------------------------------------------------------------------
    // pixel fragment: lane reads LDS pixel (row, col0 + delta + kx), logical chunk 4*ks + q; the ky row
    // shift and the tile offset are compile-time and ride in the ds_read immediate
    const int dl = delta(lr);
    const int gbase = (G::modeA ? wpx * G::CPG * 16 * 128
                                : (G::modeB ? wpx * PITCH * 128 : wpx * TPW * PITCH * 128));
    int xl[3][2];   // [kx][ks]
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int pc = dl + kx;
        const int f = (pc >> 1) & 7;
        const int a = gbase + pc * 128;
        xl[kx][0] = a + (((0 | q) ^ f) << 4);
        xl[kx][1] = a + (((4 | q) ^ f) << 4);
    }
    // weight fragment: tile t row lr of this wave's block, logical chunk 4*ks + q
    int wl[2];
    {
        const int rowb = wco * (16 * TCW) + lr;
        const int f = (lr >> 1) & 7;
        wl[0] = woff0 + rowb * 128 + (((0 | q) ^ f) << 4);
        wl[1] = woff0 + rowb * 128 + (((4 | q) ^ f) << 4);
    }

    f32x4 acc[TCW][TPW];
#pragma unroll
    for (int t = 0; t < TCW; ++t)
#pragma unroll
        for (int u = 0; u < TPW; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- main loop: per K-step (one tap x 64 channels): issue the next step's weight DMA (and a slice
    // of the next X chunk), compute both 32-deep halves from LDS, then vmcnt(0) + barrier.
    // (Measured alternatives that were NOT faster — 3 weight buffers with counted vmcnt, and
    // register-double-buffered fragments with the barrier mid-step — are recorded in DESIGN.md.)
    // epilogue tables -> LDS now, so the epilogue has no dependent global-load chain:
    // bias [ncls][BN] f32 then alpha [BN] f32, after the weight buffers
    const int ncls = p.border_cls ? 9 : 1;
    float* const ebias = (float*)(smem + woff0 + NWB * WBYTES);
    float* const ealpha = ebias + 9 * BN;
    for (int i = tid; i < ncls * BN; i += NT) ebias[i] = p.bias[(i / BN) * p.Cout + n0 + (i % BN)];
    if (p.alpha)
        for (int i = tid; i < BN; i += NT) ealpha[i] = p.alpha[n0 + i];
    unsigned long long* stamps = (unsigned long long*)p.stamps;
    if (stamps && tid == 0) stamps[(size_t)blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < XSLOTS; ++i) stage_x_slot(i, 0, 0);
    stage_w(0);
    wait_dma_then_barrier<0>();
    if (stamps && tid == 0) stamps[(size_t)blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();

    int wtog = 0;                       // byte offset of the weight buffer holding K-step t
    for (int cc = 0; cc < ncc; ++cc) {
        const int xcur = (xdouble ? (cc & 1) : 0) * XSTRIDE;
        const int xnext_buf = xdouble ? ((cc & 1) ^ 1) : 0;
        const bool more_x = cc + 1 < ncc;
        auto step = [&](auto tapc) {
            constexpr int tap = decltype(tapc)::a;
            constexpr int ky = tap / 3, kx = tap % 3;
            const int t = cc * 9 + tap;
            if (t + 1 < nk) stage_w(wtog ^ WBYTES);
            if (XDB && more_x) {
#pragma unroll
                for (int i = tap; i < XSLOTS; i += 9) stage_x_slot(i, xnext_buf, cc + 1);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                vec8 wf[TCW], pf[TPW];
#pragma unroll
                for (int tt = 0; tt < TCW; ++tt) wf[tt] = *(const vec8*)(smem + (wl[ks] + wtog) + tt * 2048);
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    const int imm = (ky * PITCH + (G::modeA ? ((u / G::CPG) * PITCH + (u % G::CPG) * 16)
                                                            : (G::modeB ? u * 16 : u * PITCH))) * 128;
                    pf[u] = *(const vec8*)(smem + (xl[kx][ks] + xcur) + imm);
                }
#pragma unroll
                for (int tt = 0; tt < TCW; ++tt)
#pragma unroll
                    for (int u = 0; u < TPW; ++u) acc[tt][u] = mfma16<T>(wf[tt], pf[u], acc[tt][u]);
            }
            wait_dma_then_barrier<0>();
            wtog ^= WBYTES;
        };
        step(IC2<0, 0>{}); step(IC2<1, 0>{}); step(IC2<2, 0>{});
        step(IC2<3, 0>{}); step(IC2<4, 0>{}); step(IC2<5, 0>{});
        step(IC2<6, 0>{}); step(IC2<7, 0>{}); step(IC2<8, 0>{});
        if (!XDB && more_x) {
            // single X buffer: every wave is past its last read of this chunk (barrier above); refill
            // and wait — the co-resident workgroup keeps the matrix cores busy meanwhile
#pragma unroll
            for (int i = 0; i < XSLOTS; ++i) stage_x_slot(i, 0, cc + 1);
            wait_dma_then_barrier<0>();
        }
    }

    if (stamps && tid == 0) stamps[(size_t)blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime();
    // ---- epilogue ---------------------------------------------------------------------------------------
    // All residual loads are issued first (independent, one exposed latency), bias/alpha come from LDS.
    // A lane holds CPL = 4 TCW channels of each of its pixels, as runs of 8 consecutive channels
    // (one 16-B store each).  TCW = 2: one run, channels 8q ..; the four q-lanes of a pixel write 64
    // contiguous bytes.  TCW = 4: two runs, channels 32h + 8q .. (h = 0, 1) — the weight rows are
    // permuted for exactly this (perm64b_row_of_channel) so that, per store instruction, the four
    // q-lanes of a pixel again cover 64 CONTIGUOUS bytes instead of four 16-B pieces 32 B apart.
    constexpr int CPL = TCW * 4;
    const int wbase = wco * (16 * TCW);                            // wave's channel block inside BN
    auto chan_t = [&](int t) { return wbase + (TCW == 4 ? 32 * (t >> 1) + 8 * q + 4 * (t & 1) : 8 * q + 4 * t); };
    auto chan_h = [&](int h) { return wbase + (TCW == 4 ? 32 * h + 8 * q : 8 * q); };
    const int cbase = n0;
    size_t off[TPW];
    bool ok[TPW];
    int cls[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const int oy = r0 + G::row(wpx, u);
        const int ox = G::cb(wpx, u) * 16 + dl;
        ok[u] = oy < H && ox < W;
        const int oyc = ok[u] ? oy : r0, oxc = ok[u] ? ox : 0;      // clamped: loads stay in bounds
        off[u] = ((size_t)(n_img * H + oyc) * W + oxc) * p.Cout + cbase;
        const int rc = (oy == 0) ? 0 : ((oy == H - 1) ? 2 : 1);
        const int cc = (ox == 0) ? 0 : ((ox == W - 1) ? 2 : 1);
        cls[u] = p.border_cls ? rc * 3 + cc : 0;
    }
    vec8 res[TPW][CPL / 8];      // the residual, or (backward mode) the stored forward activation
    const T* extra = (const T*)(p.dact ? p.dact : p.resid);
    if (extra) {
#pragma unroll
        for (int u = 0; u < TPW; ++u)
#pragma unroll
            for (int h = 0; h < CPL / 8; ++h) res[u][h] = *(const vec8*)(extra + off[u] + chan_h(h));
    }
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        float v[CPL];
#pragma unroll
        for (int t = 0; t < TCW; ++t) {
            const f32x4 b4 = *(const f32x4*)(ebias + cls[u] * BN + chan_t(t));
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 * t + j] = acc[t][u][j] + b4[j];
        }
        if (p.dact) {
            // v *= PReLU'(z): 1 where the stored activation is positive, the slope elsewhere
#pragma unroll
            for (int t = 0; t < TCW; ++t) {
                const f32x4 a4 = *(const f32x4*)(ealpha + chan_t(t));
#pragma unroll
                for (int j = 0; j < 4; ++j) v[4 * t + j] *= (float)res[u][(4 * t + j) / 8][(4 * t + j) % 8] > 0.f ? 1.f : a4[j];
            }
        } else {
            if (p.alpha) {
#pragma unroll
                for (int t = 0; t < TCW; ++t) {
                    const f32x4 a4 = *(const f32x4*)(ealpha + chan_t(t));
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[4 * t + j] = v[4 * t + j] > 0.f ? v[4 * t + j] : v[4 * t + j] * a4[j];
                }
            }
            if (p.resid) {
#pragma unroll
                for (int h = 0; h < CPL / 8; ++h)
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[8 * h + i] += (float)res[u][h][i];
            }
        }
        if (p.post_relu) {
#pragma unroll
            for (int i = 0; i < CPL; ++i) v[i] = relu_keep_nan(v[i]);
        }
        if (ok[u]) {
#pragma unroll
            for (int h = 0; h < CPL / 8; ++h) {
                vec8 o8;
#pragma unroll
                for (int i = 0; i < 8; ++i) o8[i] = (T)v[8 * h + i];
                *(vec8*)((T*)p.out + off[u] + chan_h(h)) = o8;
            }
        }
    }
    if (stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamps[(size_t)blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime();
    }
}

template <int WPX, int WCO, int TPW, int TCW, int TR, int PITCH, int NWB, int NWV = 8, bool XDB = true>
struct Variant {
    static constexpr int R = Geo<WPX, TPW, TR>::R;
    static constexpr int BN = WCO * TCW * 16;
    static constexpr int NT = NWV * 64;
    static size_t lds_bytes(int Cin) {
        const size_t xslots = ((size_t)(R + 2) * PITCH * 8 + NT - 1) / NT;
        const size_t xb = xslots * NT * 16, wb = (size_t)BN * 128;
        // fragment reads of dummy columns run up to 2 pixels past a row: they land in the weight
        // buffers that follow, still inside the allocation
        return ((XDB && Cin > 64) ? 2 : 1) * xb + NWB * wb + 10 * BN * 4;   // + bias[9][BN] + alpha[BN]
    }
    template <typename T>
    static hipError_t launch(const ConvParams& p, hipStream_t st) {
        const int tiles_m = p.N * ((p.H + R - 1) / R), ntn = p.Cout / BN;
        dim3 grid(tiles_m * ntn, 1, 1), block(NT, 1, 1);
        hipLaunchKernelGGL((conv3x3_direct_kernel<T, WPX, WCO, TPW, TCW, TR, PITCH, NWB, NWV, XDB>), grid, block,
                           lds_bytes(p.Cin), st, p);
        return hipGetLastError();
    }
    static bool fits(int Cin) { return lds_bytes(Cin) <= 160 * 1024; }
    template <typename T>
    static hipError_t set_attr() {
        const size_t want = fits(128) ? lds_bytes(128) : lds_bytes(64);
        return hipFuncSetAttribute((const void*)conv3x3_direct_kernel<T, WPX, WCO, TPW, TCW, TR, PITCH, NWB, NWV, XDB>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)want);
    }
};

typedef Variant<2, 4, 7, 4, 1, 16, 2>  D1;   // W = 14,  Cout % 256 == 0   (stage 3, stage-4 unit-1 conv1)
typedef Variant<2, 4, 7, 4, 2, 32, 2>  D2;   // W = 28,  Cout % 256 == 0   (stage-3 unit-1 conv1)
typedef Variant<2, 4, 7, 2, 2, 32, 2>  D3;   // W = 28,  Cout % 128 == 0   (stage 2)
typedef Variant<2, 4, 8, 2, 4, 64, 2>  D4;   // W = 56,  Cout % 128 == 0   (stage-2 unit-1 conv1)
typedef Variant<4, 2, 7, 2, 4, 64, 2>  D5;   // W = 56,  Cout % 64 == 0    (stage 1)
typedef Variant<4, 2, 7, 2, 7, 128, 2> D6;   // W = 112, Cout % 64 == 0    (stage-1 unit-1 conv1)
// 4-wave, single-X-buffer variants: two workgroups per CU
typedef Variant<2, 2, 7, 4, 1, 16, 2, 4, false> S1;   // W = 14, Cout % 128 == 0  (69 KB)
typedef Variant<2, 2, 7, 4, 2, 32, 2, 4, false> S3;   // W = 28, Cout % 128 == 0  (73 KB)

bool g_use_direct = true;
bool g_use_pair = true;     // prefer the two-workgroups-per-CU variants where they exist

}  // namespace

extern "C" void alink_debug_set_direct(int on) { g_use_direct = on != 0; }
extern "C" void alink_debug_set_pair(int on) { g_use_pair = on != 0; }

// Which direct variant (1..6) serves this convolution, 0 = none (use conv_igemm).
// forward convolutions of the IR backbone: the rolling-row kernels (conv3x3_c64.hip, variant 21; its stride-2 form
// conv3x3_s2c64.hip, variant 25) where they apply, else the tile kernels below (0 = none: conv_igemm).  Callers with other epilogues (backward pass: PReLU'; VGGFace2 / VGG16: post-ReLU) use
// direct_variant_tiles.
int direct_variant(int ksz, int stride, int pad, int H, int W, int Cin, int Cout) {
    if (g_use_direct) {
        if (const int v = c64_variant(ksz, stride, pad, H, W, Cin, Cout)) return v;
        if (const int v = s2c64_variant(ksz, stride, pad, H, W, Cin, Cout)) return v;
    }
    return direct_variant_tiles(ksz, stride, pad, H, W, Cin, Cout);
}
int direct_variant_tiles(int ksz, int stride, int pad, int H, int W, int Cin, int Cout) {
    if (!g_use_direct || ksz != 3 || stride != 1 || pad != 1 || Cin % 64 || Cout % 64) return 0;
    if (const int lv = linear_variant(ksz, stride, pad, H, W, Cin, Cout)) return lv;     // 11..13: conv3x3_linear.hip
    if (W > 14 * 8 || H < 1) return 0;
    const int tr = (W + 15) / 16, pitch = (W + 2 + 15) / 16 * 16;
    if (g_use_pair) {
        if (tr == 1 && pitch == 16 && W >= 12 && Cout % 128 == 0 && S1::lds_bytes(Cin) <= 80 * 1024) return 7;
        if (tr == 2 && pitch == 32 && Cout % 128 == 0 && S3::lds_bytes(Cin) <= 80 * 1024) return 8;
    }
    if (tr == 1 && pitch == 16 && W >= 12 && Cout % 256 == 0 && D1::fits(Cin)) return 1;
    if (tr == 2 && pitch == 32 && Cout % 256 == 0 && D2::fits(Cin)) return 2;
    if (tr == 2 && pitch == 32 && Cout % 128 == 0 && D3::fits(Cin)) return 3;
    if (tr == 4 && pitch == 64 && Cout % 128 == 0 && D4::fits(Cin)) return 4;
    if (tr == 4 && pitch == 64 && D5::fits(Cin)) return 5;
    if (tr == 7 && pitch == 128 && W == 112 && D6::fits(Cin)) return 6;
    return 0;
}
// weight-row permutation code of a variant (permuted_row): 17 -> perm64b (TCW = 4), 8 -> perm32 (TCW = 2)
int direct_variant_cpl(int v) { return (v == 21 || v == 25) ? 8 : (v >= 11 ? linear_variant_cpl(v) : ((v == 1 || v == 2 || v == 7 || v == 8) ? 17 : 8)); }

hipError_t direct_set_attributes() {
    hipError_t e;
#define A(V)                                                         \
    if ((e = V::set_attr<__bf16>()) != hipSuccess) return e;         \
    if ((e = V::set_attr<_Float16>()) != hipSuccess) return e;
    A(D1) A(D2) A(D3) A(D4) A(D5) A(D6) A(S1) A(S3)
#undef A
    if ((e = c64_set_attributes()) != hipSuccess) return e;
    if ((e = front_c64_set_attributes()) != hipSuccess) return e;
    if ((e = s2c64_set_attributes()) != hipSuccess) return e;
    return linear_set_attributes();
}

hipError_t launch_conv3x3_direct(int variant, int dtype, const ConvParams& p, hipStream_t st) {
    if (variant == 21) return launch_conv3x3_c64(variant, dtype, p, st);
    if (variant == 25) return launch_conv3x3_s2c64(variant, dtype, p, st);
    if (variant >= 11) return launch_conv3x3_linear(variant, dtype, p, st);
    if (dtype != ALINK_DT_BF16 && dtype != ALINK_DT_F16) return hipErrorInvalidValue;   // no split-precision form of these
    if (p.ksz != 3 || p.stride != 1 || p.pad != 1 || p.splitk != 1) return hipErrorInvalidValue;
    if ((long long)p.N * p.H * p.W * p.Cin >= (1ll << 31)) return hipErrorInvalidValue;
#define L(V) (dtype == ALINK_DT_BF16 ? V::launch<__bf16>(p, st) : V::launch<_Float16>(p, st))
    switch (variant) {
        case 1: return L(D1);
        case 2: return L(D2);
        case 3: return L(D3);
        case 4: return L(D4);
        case 5: return L(D5);
        case 6: return L(D6);
        case 7: return L(S1);
        case 8: return L(S3);
    }
#undef L
    return hipErrorInvalidValue;
}

}  // namespace alink

// conv3x3_linear.hip — 3x3 / stride 1 / pad 1 NHWC convolution on LINEAR pixel tiles (no dummy columns).
//
// The row-aligned tiles of conv3x3_direct.hip spend 2 of every 16 MFMA columns on pixels that do not
// exist (maps are 14, 28, 56 wide): 12.5 % of the matrix-core work — and, on a part that runs this
// workload at its package power limit, 12.5 % of the energy — buys nothing.  Here the
// whole batch is one tall image of N*H rows of W pixels, NHWC being contiguous across images, and a
// workgroup owns 224 CONSECUTIVE pixels of it (16 rows of 14, 8 rows of 28 or 4 rows of 56) = exactly
// fourteen 16-pixel MFMA tiles, as two wave groups of seven:
//
//   * Input staging is a plain contiguous copy: the 224 pixels plus W + 1 before and after (one row of
//     halo each side, one pixel of slack) for the current 64-channel chunk, by LDS-DMA, XOR-swizzled.
//     No zero frame is built.
//   * A lane's pixel for tile t is 16 t + d, its operand for tap (ky, kx) sits at LDS position
//     16 t + d + ky W + kx: 16 consecutive positions for any tap, so the delta() lane permutation and
//     the chunk ^ ((pos >> 1) & 7) swizzle of the row-aligned kernel keep every ds_read_b128 conflict-free;
//     (16 t) vanishes from the swizzle term, so nine per-lane offsets serve all tiles through the ds_read
//     immediate, and the upper K half is the same address ^ 64.
//   * What the zero frame did is done by ADDRESS: a lane whose tap would read across a row end (left /
//     right image border) or across an image boundary (top / bottom) gets its border bits added to the
//     operand address far above bit 17, i.e. outside the workgroup's LDS allocation, where a DS read
//     returns zero.  Four border bits per tile and lane (28 bits, one VGPR) are set up once; per (tile, tap)
//     that costs an and and a shift-add — against eight MFMAs.
//   * Output (and residual / stored activation) addresses are linear too: pixel index x Cout.
//
// Weights, K-step order, epilogue and the two-workgroups-per-CU residency are those of the 4-wave
// variants of conv3x3_direct.hip (same row permutation codes, so the packed weights are interchangeable).
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

// pixel (0..15) inside a 16-pixel MFMA tile handled by MFMA column lr (see conv3x3_direct.hip)
__device__ __forceinline__ int delta(int lr) { return lr < 4 ? 2 * lr : (lr < 12 ? 2 * (lr - 4) + 1 : 2 * (lr - 8)); }

// End of a K-step (and of an input refill): this wave's LDS-DMAs have landed (vmcnt) AND its own LDS reads have returned
// (lgkmcnt) before it arrives at the barrier.  The second wait is not optional: the MFMAs that consume a step's last
// fragments carry no memory dependence, so the compiler sinks them — and the lgkmcnt wait they imply — BELOW this
// statement; a wave then reaches the barrier with ds_reads still queued, a faster wave passes the barrier and issues the
// DMA that re-fills the buffer those reads are aimed at, and the reads return the NEXT tile's bytes.  Seen as rare
// wrong 224-pixel groups under multi-stream load (round 3: 1 row in 10^4 in the 128-channel forms, percent-level in the
// register-rich 64-channel forms, where the compiler hoists eight reads across); single-stream runs never showed it.
__device__ __forceinline__ void wait_dma_then_barrier() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int A>
struct IC { static constexpr int a = A; };

constexpr int GPX = 224;        // pixels per workgroup: 14 tiles
constexpr int TPW = 7;          // tiles per wave group (2 groups)
constexpr int NT = 256;         // 4 waves: 2 pixel groups x 2 channel groups

template <int W, int TCW>
struct Lin {
    static constexpr int BN = 2 * TCW * 16;
    static constexpr int XPIX = GPX + 2 * W + 2;                       // halo row + slack pixel on both sides
    static constexpr int XSLOTS = (XPIX * 8 + NT - 1) / NT;
    static constexpr int XBYTES = XSLOTS * NT * 16;
    static constexpr int WOFF = XBYTES;
    static constexpr int WBYTES = BN * 128;
    static constexpr size_t lds_bytes() { return (size_t)WOFF + 2 * WBYTES + 10 * BN * 4; }
    static_assert(GPX % W == 0, "a workgroup covers whole rows");
};

// EPI: the epilogue a launch needs, fixed at compile time for the two forms the forward pass uses — 1: bias + PReLU
// (a unit's conv1), 2: bias + residual (conv2) — so that neither carries the other's loads, selects and branches;
// 0: every mode by run-time flags (PReLU', post-ReLU, split partial sums, both at once).
//
// SP (T = _Float16 only): the split-precision mode ALINK_DT_F16X2 — every value an f16 pair hi + lo.  Activations are
// [pixel][2 Cin], each 64-channel chunk stored as [hi 64 | lo 64]; weights [Cout][chunk][hi: 9 taps x 64 | lo: 9 taps x 64].
// A chunk is walked three times — X = hi against W_hi, X = hi against W_lo, X = lo against W_hi — into the same f32
// accumulators (27 K-steps per chunk instead of 9, two input refills instead of one); the dropped lo x W_lo term is 2^-22
// of the sum.  Power-of-two scales per tensor (ConvParams::acc_scale, bias_scale, res_scale) keep hi in range and lo normal.
// NP (SP only): matrix-core products per multiplication — 3 = the exact mode; 1 = X_hi W_hi alone (ConvParams::nprod: the
// screening form on a split-precision handle): nine K-steps per chunk, the lo halves of input and weights never fetched.
template <typename T, int W, int TCW, int EPI, bool SP, int NP = 3>
__global__ __launch_bounds__(NT, 2) void conv3x3_linear_kernel(const ConvParams p) {
    typedef typename Vec8<T>::type vec8;
    typedef Lin<W, TCW> G;
    constexpr int BN = G::BN, XSLOTS = G::XSLOTS, WBYTES = G::WBYTES, WSLOTS = BN * 8 / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wpx = wave >> 1, wco = wave & 1;
    if (p.stagger > 0) {
        // Two workgroups share a CU, one wave each per SIMD.  In a launch that fills every slot at once they run in
        // lockstep: the same K-step, the same barrier, the same input refill and the same epilogue at the same time, so
        // nothing of one covers the other's waits.  The workgroup whose first wave sits in an odd wave slot of its SIMD
        // (HW_ID.WAVE_ID) starts late by `stagger` x 1024 cycles; the K-step barriers then keep the offset.
        __shared__ int s_late;
        if (tid == 0) s_late = (int)(__builtin_amdgcn_s_getreg(6148) & 1u);       // hwreg(HW_REG_HW_ID, 0, 4) = WAVE_ID
        __syncthreads();
        if (s_late)
            for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(16);
    }
    const int q = lane >> 4, lr = lane & 15;
    const int H = p.H, Cin = p.Cin;
    // diagnostic (alink_debug_set_stamps; nullptr in every product call): 8 x u64 per workgroup — start, main loop
    // entered, main loop left, end (stores drained), cycles inside the input refills, number of refills, residual
    // loads returned, last store issued
    unsigned long long* const stamps = (unsigned long long*)p.stamps;
    unsigned long long refill_cycles = 0;
    if (stamps && tid == 0) stamps[(size_t)blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memtime();
    // split mode (small batches: too few workgroups to fill the chip): blockIdx.y owns ncc / splitk of the
    // 64-channel input chunks and leaves its f32 partial sums in slab blockIdx.y (conv_split_finish_kernel
    // adds the slabs in order and applies the epilogue)
    constexpr int NPH = SP ? NP : 1;
    const int CinP = SP ? 2 * Cin : Cin;                               // pixel pitch of the input tensor in elements
    const int ncc_all = Cin >> 6;
    const int ncc = ncc_all / p.splitk, cc_first = (int)blockIdx.y * ncc;
    const int nk = ncc * 9 * NPH;
    const int K = 9 * CinP;                                            // weight row pitch
    const long long totpix = (long long)p.N * H * W;

    const int ntn = p.Cout / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % ntn, grp = lid / ntn;
    const int n0 = tile_n * BN;
    const long long gp0 = (long long)grp * GPX;                        // first pixel of the group

    const T* __restrict__ gin = (const T*)p.in;
    const T* __restrict__ gw = (const T*)p.wgt;
    const T* __restrict__ gz = (const T*)p.zero;

    // ---- X staging: LDS position idx <- pixel gp0 - W - 1 + idx, a contiguous span of the tensor.  Slot
    // s = tid + 256 i -> position (tid >> 3) + 32 i, 16-B piece tid & 7; the swizzle term ((position >> 1) & 7)
    // is the same for every i, so one offset per thread serves all slots ------------------------------------------
    const int xpos0 = tid >> 3;
    const int xc16 = (tid & 7) ^ ((tid >> 4) & 7);
    const long long xgp0 = gp0 - W - 1 + xpos0;                        // pixel of slot 0
    auto stage_x = [&](int cc, int part = 0) {     // part (SP): 0 = the chunk's hi half, 1 = its lo half
        long long x0 = xgp0;
        // SP refills twice per chunk from two source planes: left to itself the compiler keeps BOTH sets of per-slot
        // 64-bit source addresses live across the chunk (32 registers the kernel does not have: they went to scratch);
        // made opaque here, the addresses are rebuilt at each refill (a few dozen VALU operations per 27 K-steps)
        if (SP) asm volatile("" : "+v"(x0));
#pragma unroll
        for (int i = 0; i < XSLOTS; ++i) {
            const long long gp = x0 + 32 * i;
            const bool ok = xpos0 + 32 * i < G::XPIX && gp >= 0 && gp < totpix;
            const T* src = ok ? gin + ((size_t)gp * CinP + (SP ? 2 * cc + part : cc) * 64 + xc16 * 8) : gz + (lane & 7) * 8;
            dma16(src, smem + (NT * i + wave * 64) * 16);
        }
    };
    // ---- W staging (K-step order [cout][chunk][tap][64], as conv3x3_direct): slot s = tid + 256 i -> row
    // (tid >> 3) + 32 i, piece tid & 7, again with an i-independent swizzle term ------------------------------------
    const unsigned woff0 = (unsigned)((tid >> 3) * K + (((tid & 7) ^ ((tid >> 4) & 7)) * 8));
    const T* wstep = gw + (size_t)n0 * K + (size_t)cc_first * (SP ? 18 : 9) * 64;   // advanced by 64 elements per K-step (uniform)
    int wr = 0;       // SP: position of the K-step being staged inside its chunk's 27 (hi taps, lo taps, hi taps again)
    auto stage_w = [&](int bufoff) {
#pragma unroll
        for (int i = 0; i < WSLOTS; ++i)
            dma16(wstep + (woff0 + (unsigned)(32 * i * K)), smem + G::WOFF + bufoff + (NT * i + wave * 64) * 16);
        if (SP && NP == 1) {
            // the hi blocks 0..8 only, then over the chunk's lo blocks to the next chunk (18)
            wstep += wr == 8 ? 10 * 64 : 64;
            wr = wr == 8 ? 0 : wr + 1;
        } else if (SP) {
            // blocks of the chunk: 0..8 hi, 9..17 lo; the walk is 0..17, back to 0..8, then on to the next chunk (18)
            wstep += wr == 17 ? -17 * 64 : (wr == 26 ? 10 * 64 : 64);
            wr = wr == 26 ? 0 : wr + 1;
        } else {
            wstep += 64;
        }
    };

    // The first input span and the first weight tile are requested NOW: the per-lane border bits (integer divisions),
    // the accumulator clear and the bias / slope tables below all run under their flight instead of before it
    // (stamps: the prologue of a 512-workgroup stage-3 launch took 9.0 k of the workgroup's 100 k cycles).
    stage_x(cc_first);
    stage_w(0);

    // ---- per-lane fragment offsets and border bits --------------------------------------------------------------
    const int dl = delta(lr);
    // operand offset of tap (ky, kx), lower K half (the upper half is ^ 64): position s = dl + ky W + kx,
    // byte s * 128 + ((q ^ ((s >> 1) & 7)) << 4).  Recomputed per K-step from dl (four VALU operations) rather
    // than kept in nine registers.
    const int xbase = wpx * (TPW * 2048);
    // bit 4u + {0: top row of an image, 1: bottom row, 2: left column, 3: right column} for tile u
    unsigned border = 0;
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        // all by the compile-time W: the maps are square (H == W is a launch condition) and a group starts on a row
        // boundary (GPX % W == 0), so the group's first row is grp * (GPX / W).  (With the runtime H and the 64-bit
        // pixel index this was ~130 instructions of integer division per tile: 5 k of the prologue's 9 k cycles.)
        const int pl = 16 * (wpx * TPW + u) + dl;
        const int col = pl % W;
        const int y = (int)(((unsigned)grp * (unsigned)(GPX / W) + (unsigned)(pl / W)) % (unsigned)W);
        border |= ((y == 0 ? 1u : 0u) | (y == H - 1 ? 2u : 0u) | (col == 0 ? 4u : 0u) | (col == W - 1 ? 8u : 0u)) << (4 * u);
    }
    int wl[2];
    {
        const int rowb = wco * (16 * TCW) + lr;
        const int f = (lr >> 1) & 7;
        wl[0] = G::WOFF + rowb * 128 + (((0 | q) ^ f) << 4);
        wl[1] = G::WOFF + rowb * 128 + (((4 | q) ^ f) << 4);
    }

    f32x4 acc[TCW][TPW];
#pragma unroll
    for (int t = 0; t < TCW; ++t)
#pragma unroll
        for (int u = 0; u < TPW; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ncls = p.border_cls ? 9 : 1;
    float* const ebias = (float*)(smem + G::WOFF + 2 * WBYTES);
    float* const ealpha = ebias + 9 * BN;
    for (int i = tid; i < ncls * BN; i += NT) ebias[i] = p.bias[(i / BN) * p.Cout + n0 + (i % BN)] * (SP ? p.bias_scale : 1.f);
    if (p.alpha)
        for (int i = tid; i < BN; i += NT) ealpha[i] = p.alpha[n0 + i];
    wait_dma_then_barrier();
    if (stamps && tid == 0) stamps[(size_t)blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memtime();

    int wtog = 0;
    for (int cc = 0; cc < ncc; ++cc) {
#pragma unroll 1
      for (int ph = 0; ph < NPH; ++ph) {
        auto step = [&](auto tapc) {
            constexpr int tap = decltype(tapc)::a;
            constexpr int ky = tap / 3, kx = tap % 3;
            constexpr unsigned tapbits = (ky == 0 ? 1u : 0u) | (ky == 2 ? 2u : 0u) | (kx == 0 ? 4u : 0u) | (kx == 2 ? 8u : 0u);
            const int t = (cc * NPH + ph) * 9 + tap;
            if (t + 1 < nk) stage_w(wtog ^ WBYTES);
            // A lane whose pixel lies on the border this tap crosses must read zeros: its border bits, shifted
            // up to 2^18 and beyond, are ADDED to its operand address, which then lies outside the workgroup's
            // LDS allocation — and a DS read out of range returns zero (alink_debug_lds_oob_probe checks this
            // contract).  Two VALU operations per (tile, tap) and the tile offset stays in the ds_read immediate.
            // bnow / sdl are made opaque so that nothing derived from them is hoisted out of the chunk loop
            // into registers the kernel does not have.
            unsigned bnow = border;
            int sdl = dl;
            asm volatile("" : "+v"(bnow), "+v"(sdl));
            const int spos = sdl + ky * W + kx;
            const int xt0 = xbase + spos * 128 + ((q ^ ((spos >> 1) & 7)) << 4);
            const int xt1 = xt0 ^ 64;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                vec8 wf[TCW], pf[TPW];
#pragma unroll
                for (int tt = 0; tt < TCW; ++tt) wf[tt] = *(const vec8*)(smem + (wl[ks] + wtog) + tt * 2048);
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    const unsigned t = tapbits ? (bnow & (tapbits << (4 * u))) : 0u;
                    const unsigned far = 4 * u < 18 ? t << (18 - (4 * u < 18 ? 4 * u : 0)) : t;
                    pf[u] = *(const vec8*)(smem + ((ks ? xt1 : xt0) + (int)far) + 2048 * u);
                }
#pragma unroll
                for (int tt = 0; tt < TCW; ++tt)
#pragma unroll
                    for (int u = 0; u < TPW; ++u) acc[tt][u] = mfma16<T>(wf[tt], pf[u], acc[tt][u]);
            }
            wait_dma_then_barrier();
            wtog ^= WBYTES;
        };
        step(IC<0>{}); step(IC<1>{}); step(IC<2>{});
        step(IC<3>{}); step(IC<4>{}); step(IC<5>{});
        step(IC<6>{}); step(IC<7>{}); step(IC<8>{});
        // SP: the hi half serves phases 0 and 1, the lo half phase 2
        const bool next_chunk = ph == NPH - 1;
        if (next_chunk ? cc + 1 < ncc : ph == 1) {
            // single X buffer: every wave is past its last read of this chunk; refill and wait — the
            // co-resident workgroup keeps the matrix cores busy meanwhile
            const unsigned long long t0 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
            if (next_chunk) stage_x(cc_first + cc + 1, 0);
            else            stage_x(cc_first + cc, 1);
            wait_dma_then_barrier();
            if (stamps) refill_cycles += __builtin_amdgcn_s_memtime() - t0;
        }
      }
    }
    if (stamps && tid == 0) {
        stamps[(size_t)blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memtime();
        stamps[(size_t)blockIdx.x * 8 + 4] = refill_cycles;
        stamps[(size_t)blockIdx.x * 8 + 5] = (unsigned long long)(ncc - 1);
    }

    // ---- epilogue (as conv3x3_direct: bias by border class, PReLU | PReLU', residual, 16-B stores) -----------
    constexpr int CPL = TCW * 4;
    const int wbase = wco * (16 * TCW);
    auto chan_t = [&](int t) { return wbase + (TCW == 4 ? 32 * (t >> 1) + 8 * q + 4 * (t & 1) : 8 * q + 4 * t); };
    auto chan_h = [&](int h) { return wbase + (TCW == 4 ? 32 * h + 8 * q : 8 * q); };
    // split-K slabs: a run-time branch of the generic 16-bit epilogue; for split precision a compile-time form of its own
    // (EPI == 3) — as a run-time branch there it cost the main form 120 spilled registers
    if (SP ? EPI == 3 : (EPI == 0 && p.splitk > 1)) {     // (the SP slabs hold raw accumulators, conv_split_finish applies the scales)
        float* slab = (float*)p.out + (size_t)blockIdx.y * (size_t)totpix * p.Cout;
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const long long gp = gp0 + 16 * (wpx * TPW + u) + dl;
            if (gp < totpix) {
#pragma unroll
                for (int t = 0; t < TCW; ++t) *(f32x4*)(slab + (size_t)gp * p.Cout + n0 + chan_t(t)) = acc[t][u];
            }
        }
        return;
    }
    if constexpr (SP) {
        // split-precision epilogue: out = [pixel][2 Cout], the lane's 8-channel runs at chunk * 128 + (channel & 63), lo
        // half 64 elements on.  The finished value replaces the accumulator; the residual's hi and lo halves are added
        // in two passes (one batch of loads each: the same registers as the 16-bit kernel's single pass).
        size_t off[TPW];
        bool ok[TPW];
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            const long long gp = gp0 + 16 * (wpx * TPW + u) + dl;
            ok[u] = gp < totpix;
            off[u] = (size_t)(ok[u] ? gp : 0) * (2 * p.Cout);
            const unsigned b4 = (border >> (4 * u)) & 15u;
            const int rc = (b4 & 1u) ? 0 : ((b4 & 2u) ? 2 : 1);
            const int ccl = (b4 & 4u) ? 0 : ((b4 & 8u) ? 2 : 1);
            const int cls = p.border_cls ? rc * 3 + ccl : 0;
#pragma unroll
            for (int t = 0; t < TCW; ++t) {
                const f32x4 b = *(const f32x4*)(ebias + cls * BN + chan_t(t));
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t][u][j] = fmaf(acc[t][u][j], p.acc_scale, b[j]);
            }
        }
        if (EPI == 1 || (EPI == 0 && p.alpha)) {
#pragma unroll
            for (int t = 0; t < TCW; ++t) {
                const f32x4 a4 = *(const f32x4*)(ealpha + chan_t(t));
#pragma unroll
                for (int u = 0; u < TPW; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t][u][j] = acc[t][u][j] > 0.f ? acc[t][u][j] : acc[t][u][j] * a4[j];
            }
        }
        int so[CPL / 8];
#pragma unroll
        for (int h = 0; h < CPL / 8; ++h) so[h] = ((n0 + chan_h(h)) >> 6) * 128 + ((n0 + chan_h(h)) & 63);
        // EPI == 2 keeps this a RUN-TIME branch (always taken there): as straight-line code the residual's loads are scheduled
        // into the bias / scale pass above and the 128-channel form needs 444 bytes of scratch per lane
        if (EPI != 1 && p.resid) {
            const T* r = (const T*)p.resid;
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                vec8 res[TPW][CPL / 8];
#pragma unroll
                for (int u = 0; u < TPW; ++u)
#pragma unroll
                    for (int h = 0; h < CPL / 8; ++h) res[u][h] = *(const vec8*)(r + off[u] + so[h] + 64 * part);
#pragma unroll
                for (int u = 0; u < TPW; ++u)
#pragma unroll
                    for (int h = 0; h < CPL / 8; ++h)
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            acc[2 * h + (i >> 2)][u][i & 3] = fmaf((float)res[u][h][i], p.res_scale, acc[2 * h + (i >> 2)][u][i & 3]);
            }
        }
#pragma unroll
        for (int u = 0; u < TPW; ++u) {
            if (!ok[u]) continue;
#pragma unroll
            for (int h = 0; h < CPL / 8; ++h) {
                vec8 o8, l8;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float v = acc[2 * h + (i >> 2)][u][i & 3];
                    if (EPI == 0 && p.post_relu) v = relu_keep_nan(v);
                    o8[i] = (T)v;
                    l8[i] = (T)(v - (float)o8[i]);
                }
                *(vec8*)((T*)p.out + off[u] + so[h]) = o8;
                *(vec8*)((T*)p.out + off[u] + so[h] + 64) = l8;
            }
        }
        return;
    }
    size_t off[TPW];
    bool ok[TPW];
    int cls[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const long long gp = gp0 + 16 * (wpx * TPW + u) + dl;
        ok[u] = gp < totpix;
        off[u] = (size_t)(ok[u] ? gp : 0) * p.Cout + n0;
        const unsigned b4 = (border >> (4 * u)) & 15u;
        const int rc = (b4 & 1u) ? 0 : ((b4 & 2u) ? 2 : 1);
        const int ccl = (b4 & 4u) ? 0 : ((b4 & 8u) ? 2 : 1);
        cls[u] = p.border_cls ? rc * 3 + ccl : 0;
    }
    vec8 res[TPW][CPL / 8];
    const T* extra = EPI == 1 ? (const T*)nullptr : (const T*)((EPI == 0 && p.dact) ? p.dact : p.resid);
    if (EPI == 2 || (EPI == 0 && extra)) {
#pragma unroll
        for (int u = 0; u < TPW; ++u)
#pragma unroll
            for (int h = 0; h < CPL / 8; ++h) res[u][h] = *(const vec8*)(extra + off[u] + chan_h(h));
    }
    if (stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) stamps[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memtime();
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
        if (EPI == 0 && p.dact) {
#pragma unroll
            for (int t = 0; t < TCW; ++t) {
                const f32x4 a4 = *(const f32x4*)(ealpha + chan_t(t));
#pragma unroll
                for (int j = 0; j < 4; ++j) v[4 * t + j] *= (float)res[u][(4 * t + j) / 8][(4 * t + j) % 8] > 0.f ? 1.f : a4[j];
            }
        } else {
            if (EPI == 1 || (EPI == 0 && p.alpha)) {
#pragma unroll
                for (int t = 0; t < TCW; ++t) {
                    const f32x4 a4 = *(const f32x4*)(ealpha + chan_t(t));
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[4 * t + j] = v[4 * t + j] > 0.f ? v[4 * t + j] : v[4 * t + j] * a4[j];
                }
            }
            if (EPI == 2 || (EPI == 0 && p.resid)) {
#pragma unroll
                for (int h = 0; h < CPL / 8; ++h)
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[8 * h + i] += (float)res[u][h][i];
            }
        }
        if (EPI == 0 && p.post_relu) {
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
        stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamps[(size_t)blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memtime();
    }
}

bool g_generic_epilogue = false;      // A/B: every launch on the run-time-flag epilogue (EPI = 0)

template <typename T, int W, int TCW, bool SP = false>
hipError_t launch_one(const ConvParams& p, hipStream_t st) {
    typedef Lin<W, TCW> G;
    const long long totpix = (long long)p.N * p.H * W;
    const long long groups = (totpix + GPX - 1) / GPX;
    const long long nwg = groups * (p.Cout / G::BN);
    if (nwg <= 0 || nwg >= (1ll << 31)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)nwg, (unsigned)p.splitk);
    if (SP) {
        if (p.dact) return hipErrorInvalidValue;
        if (p.nprod == 1) {
            if (p.splitk > 1) hipLaunchKernelGGL((conv3x3_linear_kernel<T, W, TCW, 3, SP, 1>), grid, dim3(NT), G::lds_bytes(), st, p);
            else              hipLaunchKernelGGL((conv3x3_linear_kernel<T, W, TCW, 0, SP, 1>), grid, dim3(NT), G::lds_bytes(), st, p);
            return hipGetLastError();
        }
        // round 5: the exact mode's two forward forms get compile-time epilogues too (1: bias + PReLU, 2: bias + residual),
        // as the 16-bit kernels did in round 2 — same arithmetic in the same order, bit-identical (tests/test_gpu_conv.py)
        const bool plain_sp = p.splitk == 1 && !p.post_relu && !p.stamps && !g_generic_epilogue;
        if (p.splitk > 1) hipLaunchKernelGGL((conv3x3_linear_kernel<T, W, TCW, 3, SP>), grid, dim3(NT), G::lds_bytes(), st, p);
        else if (plain_sp && p.alpha && !p.resid)
            hipLaunchKernelGGL((conv3x3_linear_kernel<T, W, TCW, 1, SP>), grid, dim3(NT), G::lds_bytes(), st, p);
        else if (plain_sp && !p.alpha && p.resid)
            hipLaunchKernelGGL((conv3x3_linear_kernel<T, W, TCW, 2, SP>), grid, dim3(NT), G::lds_bytes(), st, p);
        else              hipLaunchKernelGGL((conv3x3_linear_kernel<T, W, TCW, 0, SP>), grid, dim3(NT), G::lds_bytes(), st, p);
        return hipGetLastError();
    }
    const bool plain = p.splitk == 1 && !p.dact && !p.post_relu && !p.stamps && !g_generic_epilogue;
    if (plain && p.alpha && !p.resid)
        hipLaunchKernelGGL((conv3x3_linear_kernel<T, W, TCW, 1, false>), grid, dim3(NT), G::lds_bytes(), st, p);
    else if (plain && !p.alpha && p.resid)
        hipLaunchKernelGGL((conv3x3_linear_kernel<T, W, TCW, 2, false>), grid, dim3(NT), G::lds_bytes(), st, p);
    else
        hipLaunchKernelGGL((conv3x3_linear_kernel<T, W, TCW, 0, false>), grid, dim3(NT), G::lds_bytes(), st, p);
    return hipGetLastError();
}
template <typename T, int W, int TCW>
hipError_t set_attr_one() {
    const int lds = (int)Lin<W, TCW>::lds_bytes();
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_linear_kernel<T, W, TCW, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_linear_kernel<T, W, TCW, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv3x3_linear_kernel<T, W, TCW, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    return e;
}
template <int W, int TCW>
hipError_t set_attr_sp() {
    hipError_t e = hipFuncSetAttribute((const void*)conv3x3_linear_kernel<_Float16, W, TCW, 0, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)Lin<W, TCW>::lds_bytes());
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv3x3_linear_kernel<_Float16, W, TCW, 3, true>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)Lin<W, TCW>::lds_bytes());
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv3x3_linear_kernel<_Float16, W, TCW, 1, true>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)Lin<W, TCW>::lds_bytes());
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv3x3_linear_kernel<_Float16, W, TCW, 2, true>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)Lin<W, TCW>::lds_bytes());
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv3x3_linear_kernel<_Float16, W, TCW, 0, true, 1>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)Lin<W, TCW>::lds_bytes());
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv3x3_linear_kernel<_Float16, W, TCW, 3, true, 1>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)Lin<W, TCW>::lds_bytes());
    return e;
}

// Which map widths take the linear-tile kernel: bit 0 = 56, bit 1 = 28, bit 2 = 14, bit 3 = 7, bit 4 = 112 (split
// precision only: the 16-bit modes have the row-aligned 8-wave kernel there, 6 % faster alone) (default: all).
// Measured on MI355X: a single 256-image launch at 28 or 14 wide has 448 workgroups for 512
// slots, so alone it takes as long as the row-aligned kernel (12.5 % fewer MFMAs, idle slots instead of
// time); with launches overlapping on several streams the embedding rate rises by 4.3 %.
int g_linear_mode = 31;

}  // namespace

extern "C" void alink_debug_set_linear(int mode) { g_linear_mode = mode; }
extern "C" void alink_debug_set_generic_epilogue(int on) { g_generic_epilogue = on != 0; }

// Hardware contract probe: a DS read whose address lies beyond the workgroup's LDS allocation returns zero
// (and does not fault).  The kernel above adds border bits 18..21 (tiles 0-4), 20..23 (tile 5) or 24..27 (tile 6)
// to an in-range operand address — one or two of them at a time (a corner pixel crosses a row AND a column
// border) — and reads with the tile's immediate offset 2048 u, u <= 6.  The probe reads, from a 1 KB allocation:
//   lanes  0.. 9   base + 2^(18 + lane)                                  every single bit the kernel can set
//   lanes 10..54   base + 2^a + 2^b, 18 <= a < b <= 27                   every pair (a superset of the kernel's)
//   lanes 55..63   base + all ten bits, base + 2^27 + 2^26 + ..., etc.   wider sums than the kernel ever forms
// each once with immediate offset 0 and once with offset:12288 (= 2048 * 6), from a per-lane in-range base;
// out: 64 x 8 floats that must all be zero, then 4 floats read IN range that must be the stored 1, 2, 3, 4
// (if the DS unit dropped high address bits before its bounds check, the far reads would alias those).
namespace {
__global__ void lds_oob_probe_kernel(float* out) {
    __shared__ __attribute__((aligned(16))) float buf[256];
    for (int i = threadIdx.x; i < 256; i += 64) buf[i] = 1.0f + i;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)buf;   // escapes: the stores stay
    const int lane = threadIdx.x;
    unsigned bits;
    if (lane < 10) {
        bits = 1u << (18 + lane);
    } else if (lane < 55) {
        int k = lane - 10, a = 18;
        while (k >= 27 - a) { k -= 27 - a; ++a; }                       // pair index -> (a, b), a < b
        bits = (1u << a) | (1u << (a + 1 + k));
    } else {
        bits = 0x0FFC0000u >> (lane - 55) & 0x0FFC0000u;                // runs of the ten bits
    }
    const unsigned far = base + (lane & 15) * 16u + bits, near = base + (lane & 3) * 16u;
    f32x4 a0, a1, b;
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:12288\n\tds_read_b128 %2, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(b) : "v"(far), "v"(near) : "memory");
    *(f32x4*)(out + lane * 8) = a0;
    *(f32x4*)(out + lane * 8 + 4) = a1;
    if (lane == 0) *(f32x4*)(out + 512) = b;
}
constexpr int PROBE_FLOATS = 516;
bool g_contract_ok[64] = {};      // per device: set by the probe at alink_init / finalize on that device
}  // namespace
// Run once per device (init_kernels): the linear-tile kernel is used on a device only where the contract it rests
// on holds there (otherwise every width takes the row-aligned / implicit-GEMM kernels, which need no such contract).
hipError_t linear_check_contract() {
    const int dev = current_device();
    float* d = nullptr;
    hipError_t e = hipMalloc((void**)&d, PROBE_FLOATS * sizeof(float));
    if (e != hipSuccess) return e;
    float h[PROBE_FLOATS];
    for (int i = 0; i < PROBE_FLOATS; ++i) h[i] = 7.f;
    e = hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(lds_oob_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)0, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return e;
    bool ok = h[512] == 1.f && h[513] == 2.f && h[514] == 3.f && h[515] == 4.f;
    for (int i = 0; i < 512; ++i) ok = ok && h[i] == 0.f;
    if (dev >= 0 && dev < 64) g_contract_ok[dev] = ok;
    return hipSuccess;
}

extern "C" int alink_debug_lds_oob_probe(float* dev_out516, void* stream) {
    DeviceGuard dg(device_of_pointer(dev_out516));
    hipLaunchKernelGGL(lds_oob_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dev_out516);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
// 1 if the probe passed on the current device (the debug mode setter cannot override a failed probe)
extern "C" int alink_debug_linear_contract_ok(void) {
    const int dev = current_device();
    return dev >= 0 && dev < 64 && g_contract_ok[dev] ? 1 : 0;
}

// 11 / 12 / 13 / 14: linear-tile kernel for 14 / 28 / 56 / 7-wide square maps (0 = not applicable)
int linear_variant(int ksz, int stride, int pad, int H, int W, int Cin, int Cout) {
    if (ksz != 3 || stride != 1 || pad != 1 || H != W || Cin % 64) return 0;
    const int dev = current_device();
    if (dev < 0 || dev >= 64 || !g_contract_ok[dev]) return 0;        // the probe failed (or never ran) on this device
    if ((g_linear_mode & 4) && W == 14 && Cout % 128 == 0) return 11;
    if ((g_linear_mode & 2) && W == 28 && Cout % 128 == 0) return 12;
    if ((g_linear_mode & 1) && W == 56 && Cout % 64 == 0) return 13;
    if ((g_linear_mode & 8) && W == 7 && Cout % 128 == 0) return 14;
    return 0;
}
// split precision: the same, plus the 112-wide layer (variant 15: 2 rows per workgroup, 64-channel tiles, 80 KB of LDS, two
// workgroups per CU) — the 16-bit modes take the row-aligned kernel there, split precision would otherwise fall to the
// implicit GEMM, which stages the hi half of every input tile twice
int linear_variant_x2(int ksz, int stride, int pad, int H, int W, int Cin, int Cout) {
    if (const int v = linear_variant(ksz, stride, pad, H, W, Cin, Cout)) return v;
    if (ksz != 3 || stride != 1 || pad != 1 || H != W || Cin % 64) return 0;
    const int dev = current_device();
    if (dev < 0 || dev >= 64 || !g_contract_ok[dev]) return 0;
    if ((g_linear_mode & 16) && W == 112 && Cout % 64 == 0) return 15;
    return 0;
}
int linear_variant_cpl(int v) { return (v == 13 || v == 15) ? 8 : 17; }

hipError_t linear_set_attributes() {
    hipError_t e;
#define A(T)                                                              \
    if ((e = set_attr_one<T, 14, 4>()) != hipSuccess) return e;           \
    if ((e = set_attr_one<T, 28, 4>()) != hipSuccess) return e;           \
    if ((e = set_attr_one<T, 56, 2>()) != hipSuccess) return e;           \
    if ((e = set_attr_one<T, 7, 4>()) != hipSuccess) return e;            \
    if ((e = set_attr_one<T, 14, 2>()) != hipSuccess) return e;           \
    if ((e = set_attr_one<T, 28, 2>()) != hipSuccess) return e;           \
    if ((e = set_attr_one<T, 7, 2>()) != hipSuccess) return e;
    A(__bf16) A(_Float16)
#undef A
#define B(W_, TCW_) if ((e = set_attr_sp<W_, TCW_>()) != hipSuccess) return e;
    B(14, 4) B(28, 4) B(56, 2) B(7, 4) B(14, 2) B(28, 2) B(7, 2) B(112, 2)
#undef B
    return hipSuccess;
}

hipError_t launch_conv3x3_linear(int variant, int dtype, const ConvParams& p, hipStream_t st) {
    if (p.ksz != 3 || p.stride != 1 || p.pad != 1 || p.H != p.W) return hipErrorInvalidValue;
    // a handful of images: the latency form (same packed weights, same sums in the same order: bit-identical)
    if (conv3x3_lat_applies(dtype, p)) return launch_conv3x3_lat(dtype, p, st);
    if (p.splitk < 1 || (p.Cin / 64) % p.splitk) return hipErrorInvalidValue;      // whole chunks per split
    if ((long long)p.N * p.H * p.W * p.Cin * (dtype == ALINK_DT_F16X2 ? 2 : 1) >= (1ll << 31)) return hipErrorInvalidValue;
#define L(W_, TCW_) (dtype == ALINK_DT_BF16 ? launch_one<__bf16, W_, TCW_>(p, st) : (dtype == ALINK_DT_F16X2 ? launch_one<_Float16, W_, TCW_, true>(p, st) : launch_one<_Float16, W_, TCW_>(p, st)))
    // p.fine: the 64-channel form of the same kernel (weight rows are packed per 32-channel block, so both
    // forms read the same tensor, and every output is the same sum in the same order)
    if (p.fine) {
        switch (variant) {
            case 11: return p.W == 14 ? L(14, 2) : hipErrorInvalidValue;
            case 12: return p.W == 28 ? L(28, 2) : hipErrorInvalidValue;
            case 14: return p.W == 7 ? L(7, 2) : hipErrorInvalidValue;
        }
    }
    switch (variant) {
        case 11: return p.W == 14 ? L(14, 4) : hipErrorInvalidValue;
        case 12: return p.W == 28 ? L(28, 4) : hipErrorInvalidValue;
        case 13: return p.W == 56 ? L(56, 2) : hipErrorInvalidValue;
        case 14: return p.W == 7 ? L(7, 4) : hipErrorInvalidValue;
        case 15: return p.W == 112 && dtype == ALINK_DT_F16X2 ? launch_one<_Float16, 112, 2, true>(p, st) : hipErrorInvalidValue;
    }
#undef L
    return hipErrorInvalidValue;
}

}  // namespace alink

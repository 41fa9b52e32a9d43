// sgemm.hip — exact-f32 GEMM on the f32-input matrix cores, with the operand gathers SmallRes needs.
//
// SmallRes (reference code/siamese.py:134-170) is trained end to end in float32 by Keras; its tower
// is four small 3x3 convolutions and one wide Dense layer, forward AND backward.  All of that is
// C[M][N] = A[M][K] . B[K][N] with different ways of finding A and B in memory:
//
//   A_ROW    A[m][k] row-major                     dense forward / dense input-gradient (A = activations, dz)
//   A_COL    A given as [K][M] row-major           dense weight-gradient (a^T . dz)
//   A_CONV   A[m][k] = im2col(NHWC image): m = (n, oy, ox), k = (ky, kx, ci), zero outside the image,
//            optional SmallRes.preprocess (x-128)/128 on load             conv forward, conv input-gradient
//   A_CONVT  A^T of the above, reduction over pixels: C[k][co] = sum_p im2col[p][k] dz[p][co], with one
//            extra row k = 9 Ci of ones (-> the bias gradient)            conv weight-gradient
//   B_ROW    B[k][n] row-major                     Keras kernels (in, out) / (ky, kx, ci, co)
//   B_COLT   B given as [N][K] row-major           dense input-gradient (dz . W^T)
//   B_FLIP   B[(ky',kx',co)][ci] = w[2-ky'][2-kx'][ci][co]                conv input-gradient
//
// One workgroup = a 64 x 64 tile of C, 4 waves of 32 x 32 (v_mfma_f32_32x32x2_f32: bit-for-bit an
// ordered fmaf chain, so results match an f32 CPU matmul to rounding), K walked in stages of 16 or 64
// through LDS (K-major tiles, pitch 96 floats so the two half-waves of an MFMA operand read hit
// disjoint banks), next tile prefetched into registers while the current one is multiplied.
// Optional split along K (grid.z) into f32 partial slabs, reduced in slab order by
// splitk_reduce_kernel (deterministic), which also applies the epilogue.
// Epilogue: + bias[n], ReLU, and/or the ReLU mask of a stored activation (act[m][n] > 0).
#include "alink_common.h"
#include "sgemm.h"

namespace alink {
namespace {

constexpr int BK_SMALL = 16, BK_DEEP = 64;

// what an absent 16-byte piece reads: zeros, so that it needs no fix-up on its way into LDS
__device__ __attribute__((aligned(16))) float g_gemm_zeros[4] = {0.f, 0.f, 0.f, 0.f};      // (not const: a constant-address-space pointer selected against a global one makes every load a FLAT one)


// a / d for 0 <= a < 2^23, d > 0 given rd = 1.f / d: one multiply and a correction instead of the ~40 instructions of an
// integer division by a run-time divisor (the loaders below decode an im2col index per 16 bytes they fetch: with one
// workgroup per CU there is no other wave to hide that arithmetic behind — it WAS the kernel's time until round 6)
__device__ __forceinline__ int fdiv(int a, int d, float rd, int& rem) {
    int q = (int)((float)a * rd);
    rem = a - q * d;
    if (rem < 0) { --q; rem += d; }
    else if (rem >= d) { ++q; rem -= d; }
    return q;
}

// WN = waves along N: 2 -> 64 x 64 tile (2 x 2 waves), 1 -> 128 x 32 tile (4 x 1 waves; layers with <= 32 outputs).
// BK = reduction steps per LDS stage: 16 (12 KB of LDS, many workgroups per CU: the large grids of the f32 backbone) or
// 64 (48-64 KB, one round trip to memory per 64 steps: the SmallRes layers, whose grids do not fill the chip — a
// workgroup alone on its CU has nothing to hide a load behind, so its time is the NUMBER of round trips; round 6).
template <int AMODE, int BMODE, int WN, int BK, bool VEC>
__global__ __launch_bounds__(256) void gemm32_kernel(const GemmP p) {
    constexpr int WM = 4 / WN, BM = 32 * WM, BN = 32 * WN;
    constexpr int LDA = BM + 32, LDB = 96;               // pitch = 32 mod 64 floats: the two half-waves of an operand read hit disjoint banks
    constexpr int AQ = BM * BK / 1024;                   // float4 pieces per thread for the A tile
    constexpr int BQ = (BN * BK + 1023) / 1024;          // ... and for the B tile (BN x BK / 4 pieces over 256 threads)
    constexpr int AKQ = 256 / BM;                        // row loaders: k-quads covered per pass
    constexpr int AKR = 1024 / BM;                       // col loaders: k rows covered per pass
    constexpr int BKR = 1024 / BN, BKQ = 256 / BN;       // the same for B_ROW / B_COLT, B_FLIP
    extern __shared__ __attribute__((aligned(16))) float gemm_lds[];      // (dynamic: 12 ... 64 KB by tile form and stage depth)
    float* const As0 = gemm_lds;
    float* const Bs0 = gemm_lds + BK * LDA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    // [probe:0]
    const int kbeg = blockIdx.z * p.kper;
    const int kend = min(p.K, kbeg + p.kper);

    // ---- per-thread loader geometry -----------------------------------------------------------------
    // A tile (BM x BK): "row" loaders (A_ROW, A_CONV) take row ale and k-quad alkq0 + j * AKQ;
    // "col" loaders (A_COL, A_CONVT) take k row alkk0 + j * AKR and 4 consecutive columns at al4.
    // B tile (BN x BK): piece j of a thread is k row blkk + j * BKR (B_ROW) or k-quad blkq + j * BKQ; with
    // BN x BK / 4 < 256 pieces only the first threads load.
    const int ale = tid & (BM - 1), alkq0 = tid / BM;
    const int alkk0 = tid / (BM / 4), al4 = (tid % (BM / 4)) * 4;
    const int ble = tid & (BN - 1), blkq = tid / BN;
    const int blkk = tid / (BN / 4), bl4 = (tid % (BN / 4)) * 4;
    const bool b_active = tid < BN * BK / 4;

    // A_CONV: decode this thread's pixel once
    int cn = 0, coy = 0, cox = 0;
    bool cvalid = false;
    const float rWo = 1.f / (float)(p.Wo > 0 ? p.Wo : 1), rHo = 1.f / (float)(p.Ho > 0 ? p.Ho : 1), rCi = 1.f / (float)(p.Ci > 0 ? p.Ci : 1);
    // (index spaces here stay far below 2^23: M = images x pixels <= 512 x 128 x 128 is checked by the launcher)
    if (AMODE == A_CONV) {
        const int m = m0 + ale;
        cvalid = m < p.M;
        const int mm = cvalid ? m : 0;
        const int r = fdiv(mm, p.Wo, rWo, cox);
        cn = fdiv(r, p.Ho, rHo, coy);
    }
    // the image this thread gathers from may live in the second buffer (GemmP::A2)
    const bool a_second = AMODE == A_CONV && p.A2 && cn >= p.a_split;
    const float* Aimg = a_second ? p.A2 : p.A;
    if (a_second) cn -= p.a_split;
    // A_CONV, 16-byte form: what a piece needs splits into this thread's pixel (fixed: py, px, pixbase) and the piece's
    // (tap, channel) — the same for every lane of the wave, so it is computed on the scalar unit (k0, the wave's k-quad and
    // the magic division are uniform) and a piece costs the vector unit two adds, two compares and the address
    const int ks = p.ks ? p.ks : 3, cstr = p.cstride ? p.cstride : 1;      // A_CONV geometry (defaults: 3x3, stride 1)
    const int py = coy * cstr - p.pad, px = cox * cstr - p.pad;
    const int pixbase = ((cn * p.H + py) * p.W + px) * p.Ci;
    const int akq_u = __builtin_amdgcn_readfirstlane(alkq0);               // BM >= 64: one k-quad column per wave
    const int Kc = 9 * p.Ci;   // im2col width (conv modes)
    // A_CONVT: this thread's four im2col columns (tap, channel) do not change along the reduction: decoded once
    int tdy[4] = {0, 0, 0, 0}, tdx[4] = {0, 0, 0, 0}, tci[4] = {0, 0, 0, 0}, tkind[4] = {0, 0, 0, 0};   // kind: 0 none, 1 gather, 2 the ones row
    if (AMODE == A_CONVT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kc = m0 + al4 + i;
            if (kc > Kc) continue;
            if (kc == Kc) { tkind[i] = 2; continue; }
            const int tap = kc / p.Ci;
            tci[i] = kc - tap * p.Ci;
            tdy[i] = tap / 3 - p.pad;
            tdx[i] = tap % 3 - p.pad;
            tkind[i] = 1;                      // (once per thread: plain divisions)
        }
    }

    // A_CONVT, 16-byte form: where this thread's walk over the pixels starts (see load_a); rows shorter than half a step (tiny
    // images) and a gathered tensor in two buffers take the 4-byte form, which decodes every piece (vec_ok)
    int t_ox = 0, t_oy = 0, t_off = 0;
    const int t_drow = (p.W - p.Wo) * p.Ci, t_dimg = (p.H - p.Ho) * p.W * p.Ci;      // offset steps at a row's / an image's end
    if (AMODE == A_CONVT && VEC) {
        const int r = fdiv(kbeg + alkk0, p.Wo, rWo, t_ox);
        const int n = fdiv(r, p.Ho, rHo, t_oy);
        t_off = ((n * p.H + t_oy + tdy[0]) * p.W + t_ox + tdx[0]) * p.Ci + tci[0];
    }

    // The loaders are BRANCH-FREE: every piece computes its address and whether it exists, loads from the address or from a
    // harmless one (the start of the operand), and keeps the value or zero.  With the `if (inside) load` form they had until
    // round 6 the compiler could not issue a stage's loads together — each sat in its own block behind a branch, followed by
    // its own s_waitcnt vmcnt(0): a stage cost ten round trips to memory one after the other, which was the whole kernel.
    // VEC (host-checked): K, M, N and the leading dimensions allow the 16-byte forms; otherwise four 4-byte loads per piece.
    // pixel pre-scaling as one subtract-multiply with kernel-uniform constants (no branch behind a load): none = (x - 0) * 1,
    // SmallRes.preprocess = (x - 128) * 2^-7 (the same bits as / 128), the backbone's = (x - pre_sub) * pre_mul
    const float psub = p.prescale == 2 ? p.pre_sub : (p.prescale ? 128.f : 0.f);
    const float pmul = p.prescale == 2 ? p.pre_mul : (p.prescale ? 0.0078125f : 1.f);
    // (offsets are 32-bit element counts from the operand's base — every operand here is far below 2^31 elements, checked by
    // the launcher — and an absent piece reads element 0: a select on an int, which the compiler keeps a select)
    // A loader only FETCHES: it leaves the raw 16 bytes (or four raw 4-byte values) in the piece's registers and a mask of
    // which of them exist — bits 0..3 — plus bit 4 for the ones row of a weight gradient.  Nothing touches the values until
    // the piece is written to LDS a stage later (fix below): a load whose value is selected, scaled or masked right behind it
    // is waited for right behind it, and a stage then costs one trip to memory per piece.
    auto ld4 = [](const float* base, int off, bool ok, float v[4]) -> int {
        const f32x4 t = *(const f32x4*)(ok ? base + off : g_gemm_zeros);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
        return ok ? 15 : 0;
    };
    auto ld1 = [](const float* base, int off, bool ok) -> float { return base[ok ? off : 0]; };
    auto load_a = [&](int k0, int j, float v[4]) -> int {
        if (AMODE == A_ROW) {
            const int m = m0 + ale, k = k0 + 4 * (alkq0 + j * AKQ);
            const int off = m * p.lda + k;
            if (VEC) return ld4(p.A, off, m < p.M && k < kend, v);
            int mk = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const bool ok = m < p.M && k + i < kend; v[i] = ld1(p.A, off + i, ok); mk |= ok ? 1 << i : 0; }
            return mk;
        } else if (AMODE == A_COL) {
            const int k = k0 + alkk0 + j * AKR, m = m0 + al4;
            const int off = k * p.lda + m;
            if (VEC) return ld4(p.A, off, k < kend && m < p.M, v);
            int mk = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const bool ok = k < kend && m + i < p.M; v[i] = ld1(p.A, off + i, ok); mk |= ok ? 1 << i : 0; }
            return mk;
        } else if (AMODE == A_CONV) {
            const int k = k0 + 4 * (alkq0 + j * AKQ);
            if (VEC) {                                          // Ci % 4 == 0 (and no pre-scaling): one pixel's four consecutive channels
                const int ku = k0 + 4 * (akq_u + j * AKQ);      // == k, known to be uniform
                const int tap = p.ci_magic ? (int)__umulhi((unsigned)ku, p.ci_magic) : ku;
                const int ci = ku - tap * p.Ci;
                const int ty = (tap * 11) >> 5, tx = tap - ty * ks;      // tap / 3 for tap < 9 (a 1x1 kernel has tap 0 only)
                const int tapoff = (ty * p.W + tx) * p.Ci + ci;
                const bool ok = cvalid && ku < kend && (unsigned)(py + ty) < (unsigned)p.H && (unsigned)(px + tx) < (unsigned)p.W;
                return ld4(Aimg, pixbase + tapoff, ok, v);
            }
            int mk = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kk = k + i;
                int ci;
                const int tap = fdiv(kk, p.Ci, rCi, ci);
                const int ty = (tap * 11) >> 5, tx = tap - ty * ks;
                const int iy = coy * cstr + ty - p.pad, ix = cox * cstr + tx - p.pad;
                const bool ok = cvalid && kk < kend && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                v[i] = ld1(Aimg, ((cn * p.H + iy) * p.W + ix) * p.Ci + ci, ok);
                mk |= ok ? 1 << i : 0;
            }
            return mk;
        } else {   // A_CONVT: reduction index = pixel, M index = (tap, ci) or the ones row
            const int pix = k0 + alkk0 + j * AKR;
            const bool live = pix < kend;
            if (VEC) {                                          // (the launcher takes this form only where the walk holds: vec_ok)
                // The pieces of a thread walk the pixels kbeg + alkk0, + AKR, + 2 AKR, ... in exactly the order the kernel asks
                // for them (AQ pieces per stage, BK = AQ x AKR), and its four columns are ONE tap's consecutive channels: the
                // element offset of (pixel + tap, channel) is carried from piece to piece — an add, and selects at a row's and
                // an image's end — instead of rebuilt from two divisions and three multiplies per piece.  That arithmetic was
                // the stage: ~100 vector instructions a piece, eight pieces, 8.5 k cycles around 2 k of MFMA
                // (tools/experiments/gemm_probe.sh phases).
                const bool ok = live && tkind[0] == 1 && (unsigned)(t_oy + tdy[0]) < (unsigned)p.H && (unsigned)(t_ox + tdx[0]) < (unsigned)p.W;
                const int mk = ld4(p.A, t_off, ok, v) | ((live && tkind[0] == 2) ? 16 : 0);
                t_ox += AKR; t_off += AKR * p.Ci;
#pragma unroll
                for (int w = 0; w < 2; ++w) {                    // (AKR <= 2 Wo: at most two row ends per step)
                    const bool c = t_ox >= p.Wo;
                    t_ox -= c ? p.Wo : 0; t_off += c ? t_drow : 0; t_oy += c ? 1 : 0;
                    const bool c2 = t_oy >= p.Ho;
                    t_oy = c2 ? 0 : t_oy; t_off += c2 ? t_dimg : 0;
                }
                return mk;
            }
            int ox, oy;
            const int r = fdiv(live ? pix : 0, p.Wo, rWo, ox);
            const int n = fdiv(r, p.Ho, rHo, oy);
            const bool second = p.A2 && n >= p.a_split;
            const float* Ap = second ? p.A2 : p.A;
            const int nn = second ? n - p.a_split : n;
            int mk = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int iy = oy + tdy[i], ix = ox + tdx[i];
                const bool ok = live && tkind[i] == 1 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                v[i] = ld1(Ap, ((nn * p.H + iy) * p.W + ix) * p.Ci + tci[i], ok);
                mk |= (ok ? 1 << i : 0) | ((live && tkind[i] == 2) ? 16 << i : 0);       // bits 4..7: the ones row
            }
            return mk;
        }
    };
    auto load_b = [&](int k0, int j, float v[4]) -> int {
        if (BMODE == B_ROW) {
            const int k = k0 + blkk + j * BKR, n = n0 + bl4;
            const int off = k * p.ldb + n;
            if (VEC) return ld4(p.B, off, b_active && k < kend && n < p.N, v);
            int mk = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const bool ok = b_active && k < kend && n + i < p.N; v[i] = ld1(p.B, off + i, ok); mk |= ok ? 1 << i : 0; }
            return mk;
        } else if (BMODE == B_COLT) {
            const int n = n0 + ble, k = k0 + 4 * (blkq + j * BKQ);
            const int off = n * p.ldb + k;
            if (VEC) return ld4(p.B, off, b_active && n < p.N && k < kend, v);
            int mk = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) { const bool ok = b_active && n < p.N && k + i < kend; v[i] = ld1(p.B, off + i, ok); mk |= ok ? 1 << i : 0; }
            return mk;
        } else {   // B_FLIP: k = (tap', co) over the dz channels p.Ci, n = ci of the weights (p.N of them)
            const int n = n0 + ble, k = k0 + 4 * (blkq + j * BKQ);
            const int tap = p.ci_magic ? (int)__umulhi((unsigned)k, p.ci_magic) : k;      // p.Ci = channels of A (= Co of the layer)
            const int co = k - tap * p.Ci;
            return ld4(p.B, ((8 - tap) * p.N + n) * p.Ci + co, b_active && n < p.N && k < kend, v);
        }
    };
    // what a fetched piece becomes on its way into LDS: absent values 0, pixels pre-scaled (conv modes), the ones row 1
    const bool a_scaled = AMODE == A_CONV || AMODE == A_CONVT;
    // (The 16-byte forms fetch all four values of a piece or none — none = from a 16-byte block of zeros — and the im2col one is
    // never pre-scaled (vec_ok): nothing to do to a piece on its way into LDS, where the 4-byte forms scale, test a bit and select per
    // VALUE.  This kernel's time is its vector instruction count — the exact-f32 MFMA shares the vector unit's issue: see the stage loop.)
    auto fix_a = [&](float v[4], int mk) {
        if (VEC) {                                               // an absent piece was fetched from g_gemm_zeros: nothing to select
            if (AMODE == A_CONVT && p.prescale) {                // (uniform; an absent value must stay 0)
                const bool have = mk & 1;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = have ? (v[i] - psub) * pmul : 0.f;
            }
            if (AMODE == A_CONVT) v[0] += (mk & 16) ? 1.f : 0.f;                   // the ones row: the quad's first column
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x = a_scaled ? (v[i] - psub) * pmul : v[i];
            v[i] = (mk >> i & 1) ? x : 0.f;
            if (AMODE == A_CONVT) v[i] += (mk >> (4 + i) & 1) ? 1.f : 0.f;
        }
    };
    auto fix_b = [&](float v[4], int mk) {
        if (VEC || BMODE == B_FLIP) return;                      // 16-byte pieces: absent ones are zeros already
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (mk >> i & 1) ? v[i] : 0.f;
    };
    auto store_a = [&](float* As, int j, const float v[4]) {
        if (AMODE == A_ROW || AMODE == A_CONV) {
            const int kq = alkq0 + j * AKQ;
#pragma unroll
            for (int i = 0; i < 4; ++i) As[(4 * kq + i) * LDA + ale] = v[i];
        } else {
            *(f32x4*)(As + (alkk0 + j * AKR) * LDA + al4) = f32x4{v[0], v[1], v[2], v[3]};
        }
    };
    auto store_b = [&](float* Bs, int j, const float v[4]) {
        if (!b_active) return;
        if (BMODE == B_ROW) {
            *(f32x4*)(Bs + (blkk + j * BKR) * LDB + bl4) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
            const int kq = blkq + j * BKQ;
#pragma unroll
            for (int i = 0; i < 4; ++i) Bs[(4 * kq + i) * LDB + ble] = v[i];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int l31 = lane & 31, hh = lane >> 5;

    // A stage: the pieces fetched a stage ago go into LDS (fixed up on the way), barrier, the stage's operands leave LDS for
    // registers in one batch, then its matrix instructions run with the NEXT stage's fetches dealt out between them — the
    // MFMAs form one dependent chain (64 cycles each) that a wave alone on its SIMD would otherwise sit out, and a fetch has
    // no consumer until the next stage's top, so nothing waits for memory in between.  (A second LDS buffer with the stores
    // dealt out as well measured SLOWER — 17.4 against 14.9 us on SmallRes' conv2: tools/experiments/gemm_probe.sh.)
    float ra[AQ][4], rb[BQ][4];
    int ma[AQ], mb[BQ];
    // [probe:1]
#pragma unroll
    for (int j = 0; j < AQ; ++j) ma[j] = load_a(kbeg, j, ra[j]);
#pragma unroll
    for (int j = 0; j < BQ; ++j) mb[j] = load_b(kbeg, j, rb[j]);
    // [probe:2]
    float* const As = As0;
    float* const Bs = Bs0;
    // (the last stage stands after the loop, not in a branch of it: with three ways through one loop body the accumulator was
    // copied register to register at every join — 32 moves a stage)
    int k0 = kbeg;
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < AQ; ++j) { fix_a(ra[j], ma[j]); store_a(As, j, ra[j]); }
#pragma unroll
        for (int j = 0; j < BQ; ++j) { fix_b(rb[j], mb[j]); store_b(Bs, j, rb[j]); }
    };
    float av[BK / 2], bv[BK / 2];
    auto operands = [&]() {                      // (read one step at a time, every MFMA waited ~130 cycles for its own ds_read)
#pragma unroll
        for (int i = 0; i < BK / 2; ++i) {
            av[i] = As[(2 * i + hh) * LDA + wm * 32 + l31];
            bv[i] = Bs[(2 * i + hh) * LDB + wn * 32 + l31];
        }
    };
    auto mfma_step = [&](int kk) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk >> 1], bv[kk >> 1], acc, 0, 0, 0); };
    for (; k0 + BK < kend; k0 += BK) {
        stash();
        // [probe:3]
        __syncthreads();
        // [probe:4]
        operands();
        if (BK == BK_SMALL) {
            // many workgroups per CU: other waves fill this one's shadows, and the scheduler does better left alone
#pragma unroll
            for (int j = 0; j < AQ; ++j) ma[j] = load_a(k0 + BK, j, ra[j]);
#pragma unroll
            for (int j = 0; j < BQ; ++j) mb[j] = load_b(k0 + BK, j, rb[j]);
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) mfma_step(kk);
        } else {
            constexpr int NP = AQ + BQ, STEPS = BK / 2, PER = STEPS / NP > 0 ? STEPS / NP : 1;
            int kk = 0;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                if (q < AQ) ma[q] = load_a(k0 + BK, q, ra[q]);
                else mb[q - AQ] = load_b(k0 + BK, q - AQ, rb[q - AQ]);
#pragma unroll
                for (int t = 0; t < PER; ++t)
                    if (kk < BK) { mfma_step(kk); kk += 2; }
                __builtin_amdgcn_sched_barrier(0);               // keep the deal as written
                // (Written straight instead — fetches, LDS operand reads, the chain — with the interleave handed to the scheduler
                // as sched_group_barrier groups of one MFMA, two LDS reads, 6-12 vector instructions and a load: the same 4.3-5.1 k
                // cycles a stage.  The skeleton WITHOUT MFMAs and loads already takes 2.0-2.5 k of them (gemm_probe.sh
                // phases_skeleton) and the two do not overlap inside one wave: the phase is their sum, and a wave alone on its SIMD
                // has no other wave to overlap with.  Two workgroups per CU run 2x the work in 1.47x the time.
                // And another wave does not overlap with it either: a WAVE-SPECIALISED form — 4 waves that only multiply, 4 that
                // only fetch into the other of two LDS buffers, one barrier a stage; bit-equal, tested — ran a stage in 4.6 k cycles,
                // the multiplying waves' 1.9 k PLUS the fetching waves' instructions, and lost the gain to its longer prologue
                // (conv2 forward 16.6 -> 17.7 us).  v_mfma_f32_32x32x2_f32 runs at the vector unit's own FP32 rate (157 TFLOP/s
                // both): the exact-f32 MFMA and the vector instructions share the SIMD's issue, whatever wave they come from.
                // What this kernel can still save is vector INSTRUCTIONS, not their placement.)
            }
#pragma unroll
            for (; kk < BK; kk += 2) mfma_step(kk);
        }
        // [probe:6]
        __syncthreads();
    }
    {
        // [probe:5]
        // the last stage: the steps that exist (a reduction shorter than the stage — conv1's K = 27 — stops early: the rest of
        // the stage is zeros; an odd tail is one more zero step)
        stash();
        __syncthreads();
        operands();
        const int steps = min(BK, (kend - k0 + 1) & ~1);
        if (BK == BK_SMALL || steps == BK) {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) mfma_step(kk);
        } else {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2)
                if (kk < steps) mfma_step(kk);                   // (uniform; unrolled so that av / bv stay registers)
        }
    }
    // [probe:7]

    // ---- epilogue -----------------------------------------------------------------------------------
    // what the 16 rows need from memory is fetched first, all of it, then combined, then stored: a load behind a store to a
    // pointer the compiler cannot tell apart waits for the store (16 round trips one after the other until round 6)
    const int col = n0 + wn * 32 + l31;
    if (col >= p.N) return;
    float* C = p.C + (p.splitk > 1 ? (size_t)blockIdx.z * p.M * p.ldc : 0);
    const bool fin = p.splitk == 1;
    const float bias = (fin && p.bias) ? p.bias[col] : 0.f;
    const float alpha = (fin && p.alpha) ? p.alpha[col] : 1.f;
    // (four rows at a time: the fetched values are live registers, and the 16-deep form runs many waves per SIMD — 64 of them
    // for all 16 rows cost it its occupancy)
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += 4) {
        float va[4], vr[4], vc[4];
        int idxs[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + q;
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            idxs[q] = (row < p.M ? row : 0) * p.ldc + col;
            va[q] = 1.f; vr[q] = 0.f; vc[q] = 0.f;
        }
        if (fin && p.act) {                     // (one uniform branch around four loads, not four branches around one each)
#pragma unroll
            for (int q = 0; q < 4; ++q) va[q] = p.act[idxs[q]];
        }
        if (fin && p.resid) {
#pragma unroll
            for (int q = 0; q < 4; ++q) vr[q] = p.resid[idxs[q]];
        }
        if (fin && p.accumulate) {
#pragma unroll
            for (int q = 0; q < 4; ++q) vc[q] = C[idxs[q]];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = acc[r0 + q];
            if (fin) {
                v += bias;
                if (p.alpha) v = v > 0.f ? v : v * alpha;
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.act) v = va[q] > 0.f ? v : 0.f;
                if (p.resid) v += vr[q];
                if (p.accumulate) v += vc[q];
            }
            va[q] = v;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + q;
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (row < p.M) C[(size_t)row * p.ldc + col] = va[q];
        }
    }
    // [probe:8]
}

// out[m][n] = epilogue( sum_z part[z][m][n] ): 32 outputs x 8 slab groups per workgroup — group g adds slabs g, g + 8, ...
// in ascending order, the eight group sums are added in group order (a fixed order whatever the grid: deterministic); the
// loads of a group are independent, so a reduction over 128 slabs is 16 dependent adds per thread, not 128 round trips.
// (Tried in round 6 and dropped: the tile's last slab to arrive — an arrival counter per tile — summing the slabs inside the
// GEMM launch.  Same bits, one launch less, but slower: with a __threadfence() per workgroup every L2 is written back and
// invalidated under the kernels running beside it (a conv input gradient 18 -> 94 us, the side stream's Adadelta 20 -> 88 us);
// with device-scope slab stores and loads instead, the one finishing workgroup's round trips outweigh this launch (Dense
// forward 18.6 + 5.0 -> 43.6 us, conv4 12.6 + 5.9 -> 21.2 us).  The sum is parallel work; a launch is how it stays parallel.)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, long long MN, int ldc,
                                                            int N, int S, const float* __restrict__ bias, int relu,
                                                            const float* __restrict__ act, int accumulate, const float* __restrict__ alpha,
                                                            const float* __restrict__ resid) {
    __shared__ float red[8][32];
    const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
    const long long i = (long long)blockIdx.x * 32 + o;
    float s = 0.f;
    if (i < MN)
        for (int z = g; z < S; z += 8) s += part[(size_t)z * MN + i];
    red[g][o] = s;
    __syncthreads();
    if (g != 0 || i >= MN) return;
    s = red[0][o];
#pragma unroll
    for (int q = 1; q < 8; ++q) s += red[q][o];
    if (bias) s += bias[i % ldc];
    if (alpha) s = s > 0.f ? s : s * alpha[i % ldc];
    if (relu) s = fmaxf(s, 0.f);
    if (act) s = act[i] > 0.f ? s : 0.f;
    if (resid) s += resid[i];
    if (accumulate) s += out[i];
    (void)N;
    out[i] = s;
}

inline bool narrow(const GemmP& p) { return p.N <= 32; }      // 128 x 32 tiles for layers with <= 32 outputs
inline long long tiles_of(const GemmP& p) {
    const int bm = narrow(p) ? 128 : 64, bn = narrow(p) ? 32 : 64;
    return (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
}
// deep stages for grids that cannot hide a load behind other workgroups (at most 2 per CU), and a reduction worth a stage
inline bool deep(const GemmP& p) { return tiles_of(p) * (p.splitk > 0 ? p.splitk : 1) <= 512 && p.K >= 24; }

// may every piece be one 16-byte load?  (operand bases are 16-byte aligned: device allocations and whole rows of them)
inline bool vec_ok(const GemmP& p) {
    bool a = true, b = true;
    if (p.amode == A_ROW) a = (p.K % 4) == 0 && (p.lda % 4) == 0;
    else if (p.amode == A_COL) a = (p.M % 4) == 0 && (p.lda % 4) == 0;
    else if (p.amode == A_CONV) a = (p.Ci % 4) == 0 && p.prescale == 0;      // (pre-scaled pixels are 3-channel images: the 4-byte form)
    else if (p.amode == A_CONVT) a = (p.Ci % 4) == 0 && 2 * p.Wo >= (narrow(p) ? 8 : 16) && !p.A2;      // (the pixel walk: at most two row ends per step of 1024 / BM pixels)
    if (p.bmode == B_ROW) b = (p.N % 4) == 0 && (p.ldb % 4) == 0;
    else if (p.bmode == B_COLT) b = (p.K % 4) == 0 && (p.ldb % 4) == 0;
    return a && b && ((((uintptr_t)p.A) | ((uintptr_t)p.B) | ((uintptr_t)(p.A2 ? p.A2 : p.A))) & 15) == 0;
}

template <int WNv, int BKv>
constexpr size_t lds_bytes() { return (size_t)BKv * ((32 * (4 / WNv) + 32) + 96) * sizeof(float); }

// more than 64 KB of dynamic LDS needs the function attribute, once per kernel and device
template <int AM, int BMo, int WNv, int BKv, bool V>
hipError_t ensure_lds() {
    static unsigned long long done = 0;                          // one bit per device
    if (lds_bytes<WNv, BKv>() <= 65536) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 64 && (done >> dev & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute((const void*)gemm32_kernel<AM, BMo, WNv, BKv, V>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes<WNv, BKv>());
    if (e == hipSuccess && dev < 64) done |= 1ull << dev;
    return e;
}

template <int AM, int BMo, int WNv, int BKv>
hipError_t launch_v(const GemmP& p, dim3 grid, hipStream_t st) {
    hipError_t e;
    if (vec_ok(p)) {
        if ((e = ensure_lds<AM, BMo, WNv, BKv, true>()) != hipSuccess) return e;
        hipLaunchKernelGGL((gemm32_kernel<AM, BMo, WNv, BKv, true>), grid, dim3(256), (lds_bytes<WNv, BKv>()), st, p);
    } else {
        if ((e = ensure_lds<AM, BMo, WNv, BKv, false>()) != hipSuccess) return e;
        hipLaunchKernelGGL((gemm32_kernel<AM, BMo, WNv, BKv, false>), grid, dim3(256), (lds_bytes<WNv, BKv>()), st, p);
    }
    return hipSuccess;
}

template <int AM, int BMo>
hipError_t launch_t(const GemmP& p, hipStream_t st) {
    const bool dp = deep(p) && (p.kper % BK_DEEP) == 0;
    if (narrow(p)) {
        dim3 grid((p.N + 31) / 32, (p.M + 127) / 128, p.splitk);
        const hipError_t e = dp ? launch_v<AM, BMo, 1, BK_DEEP>(p, grid, st) : launch_v<AM, BMo, 1, BK_SMALL>(p, grid, st);
        if (e != hipSuccess) return e;
    } else {
        dim3 grid((p.N + 63) / 64, (p.M + 63) / 64, p.splitk);
        const hipError_t e = dp ? launch_v<AM, BMo, 2, BK_DEEP>(p, grid, st) : launch_v<AM, BMo, 2, BK_SMALL>(p, grid, st);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

}  // namespace

size_t gemm32_workspace_floats(const GemmP& p) { return p.splitk > 1 ? (size_t)p.splitk * p.M * p.ldc : 0; }

// Chooses a split along K so that a small (M, N) problem still fills the chip; kper is a multiple of the stage depth (64).
void gemm32_plan_split(GemmP& p, int max_split) {
    const long long tiles = tiles_of(p);
    int s = 1;
    while (s < max_split && tiles * s < 384 && p.K / (s * 2) >= 2 * BK_DEEP) s *= 2;
    p.kper = ((p.K + s - 1) / s + BK_DEEP - 1) / BK_DEEP * BK_DEEP;
    p.splitk = (p.K + p.kper - 1) / p.kper;
}

hipError_t launch_gemm32(const GemmP& p, float* workspace, hipStream_t st) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.splitk < 1 || p.kper <= 0 || (p.kper % BK_SMALL) != 0) return hipErrorInvalidValue;
    // the loaders' reciprocal divisions (fdiv) are exact below 2^23
    if ((p.amode == A_CONV && p.M >= (1 << 23)) || (p.amode == A_CONVT && p.K >= (1 << 23)) || p.K >= (1 << 23)) {
        if (p.amode == A_CONV || p.amode == A_CONVT) return hipErrorInvalidValue;
    }
    if (p.splitk > 1 && !workspace) return hipErrorInvalidValue;
    if (p.bmode == B_FLIP && ((p.Ci & 3) || p.amode != A_CONV)) return hipErrorInvalidValue;
    if (p.splitk > 1 && p.ldc != p.N) return hipErrorInvalidValue;     // slabs are dense [M][N]
    GemmP q = p;
    q.ci_magic = p.Ci > 1 ? (unsigned)((0x100000000ull + (unsigned)p.Ci - 1) / (unsigned)p.Ci) : 0u;      // exact for k < 2^32 / Ci
    float* out = p.C;
    if (p.splitk > 1) q.C = workspace;
    hipError_t e = hipErrorInvalidValue;
#define CASE(a, b) if (p.amode == a && p.bmode == b) e = launch_t<a, b>(q, st);
    CASE(A_ROW, B_ROW) CASE(A_ROW, B_COLT) CASE(A_COL, B_ROW) CASE(A_CONV, B_ROW) CASE(A_CONV, B_FLIP) CASE(A_CONVT, B_ROW)
#undef CASE
    if (e != hipSuccess) return e;
    if (p.splitk > 1) {
        const long long MN = (long long)p.M * p.N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((MN + 31) / 32)), dim3(256), 0, st, workspace, out, MN,
                           p.ldc, p.N, p.splitk, p.bias, p.relu, p.act, p.accumulate, p.alpha, p.resid);
        e = hipGetLastError();
    }
    return e;
}

hipError_t launch_slab_sum(const float* part, float* out, long long MN, int S, hipStream_t st) {
    if (!part || !out || MN <= 0 || S <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((MN + 31) / 32)), dim3(256), 0, st, part, out, MN, (int)MN, (int)MN, S,
                       (const float*)nullptr, 0, (const float*)nullptr, 0, (const float*)nullptr, (const float*)nullptr);
    return hipGetLastError();
}

}  // namespace alink

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
// ordered fmaf chain, so results match an f32 CPU matmul to rounding), K walked in steps of 16
// through LDS (K-major tiles, pitch 96 floats so the two half-waves of an MFMA operand read hit
// disjoint banks), next tile prefetched into registers while the current one is multiplied.
// Optional split along K (grid.z) into f32 partial slabs, reduced in slab order by
// splitk_reduce_kernel (deterministic), which also applies the epilogue.
// Epilogue: + bias[n], ReLU, and/or the ReLU mask of a stored activation (act[m][n] > 0).
#include "alink_common.h"
#include "sgemm.h"

namespace alink {
namespace {

constexpr int BK = 16;

__device__ __forceinline__ float prescale_px(float x) { return (x - 128.f) / 128.f; }
__device__ __forceinline__ float prescale_px(float x, const GemmP& p) { return p.prescale == 2 ? (x - p.pre_sub) * p.pre_mul : (x - 128.f) / 128.f; }

// WN = waves along N: 2 -> 64 x 64 tile (2 x 2 waves), 1 -> 128 x 32 tile (4 x 1 waves; layers with <= 32 outputs)
template <int AMODE, int BMODE, int WN>
__global__ __launch_bounds__(256) void gemm32_kernel(const GemmP p) {
    constexpr int WM = 4 / WN, BM = 32 * WM, BN = 32 * WN;
    constexpr int LDA = BM + 32, LDB = 96;               // pitch = 32 mod 64 floats: the two half-waves of an operand read hit disjoint banks
    constexpr int AQ = BM / 64;                          // float4 pieces per thread for the A tile
    __shared__ float As[BK * LDA];
    __shared__ float Bs[BK * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * p.kper;
    const int kend = min(p.K, kbeg + p.kper);

    // ---- per-thread loader geometry -----------------------------------------------------------------
    // A tile (BM x 16): "row" loaders (A_ROW, A_CONV) take row ale and k-quad alkq0 + j * (256 / BM);
    // "col" loaders (A_COL, A_CONVT) take k row alkk0 + j * (1024 / BM) and 4 consecutive columns at al4.
    // B tile (BN x 16): one piece per thread; with BN = 32 only the first 128 threads load.
    const int ale = tid & (BM - 1), alkq0 = tid / BM;
    const int alkk0 = tid / (BM / 4), al4 = (tid % (BM / 4)) * 4;
    const int ble = tid & (BN - 1), blkq = tid / BN;
    const int blkk = tid / (BN / 4), bl4 = (tid % (BN / 4)) * 4;
    const bool b_active = tid < BN * 4;

    // A_CONV: decode this thread's pixel once
    int cn = 0, coy = 0, cox = 0;
    bool cvalid = false;
    if (AMODE == A_CONV) {
        const int m = m0 + ale;
        cvalid = m < p.M;
        const int mm = cvalid ? m : 0;
        cox = mm % p.Wo;
        const int r = mm / p.Wo;
        coy = r % p.Ho;
        cn = r / p.Ho;
    }
    const int Kc = 9 * p.Ci;   // im2col width (conv modes)
    const int ks = p.ks ? p.ks : 3, cstr = p.cstride ? p.cstride : 1;      // A_CONV geometry (defaults: 3x3, stride 1)

    auto load_a = [&](int k0, int j, float v[4]) {
        v[0] = v[1] = v[2] = v[3] = 0.f;
        if (AMODE == A_ROW) {
            const int m = m0 + ale, k = k0 + 4 * (alkq0 + j * (256 / BM));
            if (m < p.M && k < kend) {
                const float* src = p.A + (size_t)m * p.lda + k;
                if (k + 3 < kend) { const f32x4 t = *(const f32x4*)src; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
                else for (int i = 0; i < 4 && k + i < kend; ++i) v[i] = src[i];
            }
        } else if (AMODE == A_COL) {
            const int k = k0 + alkk0 + j * (1024 / BM), m = m0 + al4;
            if (k < kend && m < p.M) {
                const float* src = p.A + (size_t)k * p.lda + m;
                if (m + 3 < p.M) { const f32x4 t = *(const f32x4*)src; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
                else for (int i = 0; i < 4 && m + i < p.M; ++i) v[i] = src[i];
            }
        } else if (AMODE == A_CONV) {
            if (!cvalid) return;
            const int k = k0 + 4 * (alkq0 + j * (256 / BM));
            if ((p.Ci & 3) == 0) {
                if (k >= kend) return;
                const int tap = k / p.Ci, ci = k - tap * p.Ci;
                const int iy = coy * cstr + tap / ks - p.pad, ix = cox * cstr + tap % ks - p.pad;
                if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
                    const f32x4 t = *(const f32x4*)(p.A + (((size_t)cn * p.H + iy) * p.W + ix) * p.Ci + ci);
                    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
                    if (p.prescale) for (int i = 0; i < 4; ++i) v[i] = prescale_px(v[i], p);
                }
            } else {
                for (int i = 0; i < 4; ++i) {
                    const int kk = k + i;
                    if (kk >= kend) break;
                    const int tap = kk / p.Ci, ci = kk - tap * p.Ci;
                    const int iy = coy * cstr + tap / ks - p.pad, ix = cox * cstr + tap % ks - p.pad;
                    if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
                        const float x = p.A[(((size_t)cn * p.H + iy) * p.W + ix) * p.Ci + ci];
                        v[i] = p.prescale ? prescale_px(x, p) : x;
                    }
                }
            }
        } else {   // A_CONVT: reduction index = pixel, M index = (tap, ci) or the ones row
            const int pix = k0 + alkk0 + j * (1024 / BM);
            if (pix >= kend) return;
            const int ox = pix % p.Wo;
            const int r = pix / p.Wo;
            const int oy = r % p.Ho, n = r / p.Ho;
            for (int i = 0; i < 4; ++i) {
                const int kc = m0 + al4 + i;
                if (kc > Kc) break;
                if (kc == Kc) { v[i] = 1.f; break; }
                const int tap = kc / p.Ci, ci = kc - tap * p.Ci;
                const int iy = oy + tap / 3 - p.pad, ix = ox + tap % 3 - p.pad;
                if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
                    const float x = p.A[(((size_t)n * p.H + iy) * p.W + ix) * p.Ci + ci];
                    v[i] = p.prescale ? prescale_px(x) : x;
                }
            }
        }
    };
    auto load_b = [&](int k0, float v[4]) {
        v[0] = v[1] = v[2] = v[3] = 0.f;
        if (!b_active) return;
        if (BMODE == B_ROW) {
            const int k = k0 + blkk, n = n0 + bl4;
            if (k < kend && n < p.N) {
                const float* src = p.B + (size_t)k * p.ldb + n;
                if (n + 3 < p.N) { const f32x4 t = *(const f32x4*)src; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
                else for (int i = 0; i < 4 && n + i < p.N; ++i) v[i] = src[i];
            }
        } else if (BMODE == B_COLT) {
            const int n = n0 + ble, k = k0 + 4 * blkq;
            if (n < p.N && k < kend) {
                const float* src = p.B + (size_t)n * p.ldb + k;
                if (k + 3 < kend) { const f32x4 t = *(const f32x4*)src; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
                else for (int i = 0; i < 4 && k + i < kend; ++i) v[i] = src[i];
            }
        } else {   // B_FLIP: k = (tap', co) over the dz channels p.Ci, n = ci of the weights (p.N of them)
            const int n = n0 + ble, k = k0 + 4 * blkq;
            if (n < p.N && k < kend) {
                const int tap = k / p.Ci, co = k - tap * p.Ci;          // p.Ci = channels of A (= Co of the layer)
                const f32x4 t = *(const f32x4*)(p.B + ((size_t)(8 - tap) * p.N + n) * p.Ci + co);
                v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
            }
        }
    };
    auto store_a = [&](int j, const float v[4]) {
        if (AMODE == A_ROW || AMODE == A_CONV) {
            const int kq = alkq0 + j * (256 / BM);
#pragma unroll
            for (int i = 0; i < 4; ++i) As[(4 * kq + i) * LDA + ale] = v[i];
        } else {
            *(f32x4*)(As + (alkk0 + j * (1024 / BM)) * LDA + al4) = f32x4{v[0], v[1], v[2], v[3]};
        }
    };
    auto store_b = [&](const float v[4]) {
        if (!b_active) return;
        if (BMODE == B_ROW) {
            *(f32x4*)(Bs + blkk * LDB + bl4) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) Bs[(4 * blkq + i) * LDB + ble] = v[i];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int l31 = lane & 31, hh = lane >> 5;

    float ra[AQ][4], rb[4];
#pragma unroll
    for (int j = 0; j < AQ; ++j) load_a(kbeg, j, ra[j]);
    load_b(kbeg, rb);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
        for (int j = 0; j < AQ; ++j) store_a(j, ra[j]);
        store_b(rb);
        __syncthreads();
        if (k0 + BK < kend) {
#pragma unroll
            for (int j = 0; j < AQ; ++j) load_a(k0 + BK, j, ra[j]);
            load_b(k0 + BK, rb);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a = As[(kk + hh) * LDA + wm * 32 + l31];
            const float b = Bs[(kk + hh) * LDB + wn * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue -----------------------------------------------------------------------------------
    const int col = n0 + wn * 32 + l31;
    if (col >= p.N) return;
    float* C = p.C + (p.splitk > 1 ? (size_t)blockIdx.z * p.M * p.ldc : 0);
    const float bias = (p.splitk == 1 && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (row >= p.M) continue;
        float v = acc[r];
        const size_t idx = (size_t)row * p.ldc + col;
        if (p.splitk == 1) {
            v += bias;
            if (p.alpha) v = v > 0.f ? v : v * p.alpha[col];
            if (p.relu) v = fmaxf(v, 0.f);
            if (p.act) v = p.act[idx] > 0.f ? v : 0.f;
            if (p.resid) v += p.resid[idx];
            if (p.accumulate) v += C[idx];
        }
        C[idx] = v;
    }
}

// out[m][n] = epilogue( sum_z part[z][m][n] ), z ascending
__global__ void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, long long MN, int ldc,
                                     int N, int S, const float* __restrict__ bias, int relu,
                                     const float* __restrict__ act, int accumulate, const float* __restrict__ alpha,
                                     const float* __restrict__ resid) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= MN) return;
    float s = 0.f;
    for (int z = 0; z < S; ++z) s += part[(size_t)z * MN + i];
    if (bias) s += bias[i % ldc];
    if (alpha) s = s > 0.f ? s : s * alpha[i % ldc];
    if (relu) s = fmaxf(s, 0.f);
    if (act) s = act[i] > 0.f ? s : 0.f;
    if (resid) s += resid[i];
    if (accumulate) s += out[i];
    (void)N;
    out[i] = s;
}

// gb[c] = sum_r dz[r][c] (fixed order)
__global__ void colsum_kernel(const float* __restrict__ dz, float* __restrict__ gb, int n, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int r = 0; r < n; ++r) s += dz[(size_t)r * C + c];
    gb[c] = s;
}

inline bool narrow(const GemmP& p) { return p.N <= 32; }      // 128 x 32 tiles for layers with <= 32 outputs

template <int AM, int BMo>
hipError_t launch_t(const GemmP& p, hipStream_t st) {
    if (narrow(p)) {
        dim3 grid((p.N + 31) / 32, (p.M + 127) / 128, p.splitk);
        hipLaunchKernelGGL((gemm32_kernel<AM, BMo, 1>), grid, dim3(256), 0, st, p);
    } else {
        dim3 grid((p.N + 63) / 64, (p.M + 63) / 64, p.splitk);
        hipLaunchKernelGGL((gemm32_kernel<AM, BMo, 2>), grid, dim3(256), 0, st, p);
    }
    return hipGetLastError();
}

}  // namespace

size_t gemm32_workspace_floats(const GemmP& p) { return p.splitk > 1 ? (size_t)p.splitk * p.M * p.ldc : 0; }

// Chooses a split along K so that a small (M, N) problem still fills the chip; kper is a multiple of BK.
void gemm32_plan_split(GemmP& p, int max_split) {
    const int bm = narrow(p) ? 128 : 64, bn = narrow(p) ? 32 : 64;
    const long long tiles = (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
    int s = 1;
    while (s < max_split && tiles * s < 512 && p.K / (s * 2) >= 4 * BK) s *= 2;
    p.kper = ((p.K + s - 1) / s + BK - 1) / BK * BK;
    p.splitk = (p.K + p.kper - 1) / p.kper;
}

hipError_t launch_gemm32(const GemmP& p, float* workspace, hipStream_t st) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.splitk < 1 || p.kper <= 0 || (p.kper % BK) != 0) return hipErrorInvalidValue;
    if (p.splitk > 1 && !workspace) return hipErrorInvalidValue;
    if (p.bmode == B_FLIP && ((p.Ci & 3) || p.amode != A_CONV)) return hipErrorInvalidValue;
    if (p.splitk > 1 && p.ldc != p.N) return hipErrorInvalidValue;     // slabs are dense [M][N]
    GemmP q = p;
    float* out = p.C;
    if (p.splitk > 1) q.C = workspace;
    hipError_t e = hipErrorInvalidValue;
#define CASE(a, b) if (p.amode == a && p.bmode == b) e = launch_t<a, b>(q, st);
    CASE(A_ROW, B_ROW) CASE(A_ROW, B_COLT) CASE(A_COL, B_ROW) CASE(A_CONV, B_ROW) CASE(A_CONV, B_FLIP) CASE(A_CONVT, B_ROW)
#undef CASE
    if (e != hipSuccess) return e;
    if (p.splitk > 1) {
        const long long MN = (long long)p.M * p.N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((MN + 255) / 256)), dim3(256), 0, st, workspace, out, MN,
                           p.ldc, p.N, p.splitk, p.bias, p.relu, p.act, p.accumulate, p.alpha, p.resid);
        e = hipGetLastError();
    }
    return e;
}

hipError_t launch_colsum(const float* dz, float* gb, int n, int C, hipStream_t st) {
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 255) / 256), dim3(256), 0, st, dz, gb, n, C);
    return hipGetLastError();
}

}  // namespace alink

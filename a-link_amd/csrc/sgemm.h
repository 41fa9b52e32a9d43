// sgemm.h — launch descriptor of the exact-f32 MFMA GEMM (sgemm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace alink {

enum { A_ROW = 0, A_COL = 1, A_CONV = 2, A_CONVT = 3 };
enum { B_ROW = 0, B_COLT = 1, B_FLIP = 2 };

struct GemmP {
    const float* A;
    const float* B;
    float* C;
    const float* A2;          // A_CONV / A_CONVT: images a_split, a_split + 1, ... of the gathered tensor live in this second buffer
    int a_split;              //   (the two sides of a siamese batch, handed over as two arrays); nullptr: one buffer
    int M, N, K;              // C[M][N] = A[M][K] . B[K][N]
    int lda, ldb, ldc;        // leading dimensions of the plain layouts
    int amode, bmode;
    // convolution geometry (A_CONV / A_CONVT): the gathered tensor is [n][H][W][Ci]; (Ho, Wo) are the
    // spatial dims of the OTHER side (output pixels for A_CONV, dz pixels for A_CONVT); 3x3 taps, `pad`
    int H, W, Ci, Ho, Wo, pad, prescale;
    int ks, cstride;          // A_CONV only: kernel size 3 (0 = 3) or 1, and stride 1 (0 = 1) or 2 (the f32 backbone mode)
    float pre_sub, pre_mul;   // prescale = 2: (x - pre_sub) * pre_mul on load instead of SmallRes' (x - 128) / 128
    // epilogue (applied by the kernel, or by the reduction when split): + bias[n], ReLU, mask act[m][n] > 0
    const float* bias;
    const float* act;
    const float* alpha;       // per-column PReLU slopes applied after the bias: v > 0 ? v : v * alpha[n]
    const float* resid;       // [M][ldc] added last (residual connection)
    int relu;
    int accumulate;           // C += result (after the epilogue terms) instead of C = result
    int splitk, kper;         // grid.z slabs of kper (multiple of 16) reduction steps
    unsigned ci_magic;        // set by launch_gemm32: ceil(2^32 / Ci) (0 for Ci = 1): k / Ci as one multiply-high in the loaders
};

void       gemm32_plan_split(GemmP& p, int max_split);
size_t     gemm32_workspace_floats(const GemmP& p);
hipError_t launch_gemm32(const GemmP& p, float* workspace, hipStream_t st);
// out[i] = sum_z part[z][i], i < MN, z < S — the split-K slab sum on its own (fixed order: 8 interleaved groups, then the groups)
hipError_t launch_slab_sum(const float* part, float* out, long long MN, int S, hipStream_t st);

}  // namespace alink

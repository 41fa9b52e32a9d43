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
    int M, N, K;              // C[M][N] = A[M][K] . B[K][N]
    int lda, ldb, ldc;        // leading dimensions of the plain layouts
    int amode, bmode;
    // convolution geometry (A_CONV / A_CONVT): the gathered tensor is [n][H][W][Ci]; (Ho, Wo) are the
    // spatial dims of the OTHER side (output pixels for A_CONV, dz pixels for A_CONVT); 3x3 taps, `pad`
    int H, W, Ci, Ho, Wo, pad, prescale;
    // epilogue (applied by the kernel, or by the reduction when split): + bias[n], ReLU, mask act[m][n] > 0
    const float* bias;
    const float* act;
    int relu;
    int accumulate;           // C += result (after the epilogue terms) instead of C = result
    int splitk, kper;         // grid.z slabs of kper (multiple of 16) reduction steps
};

void       gemm32_plan_split(GemmP& p, int max_split);
size_t     gemm32_workspace_floats(const GemmP& p);
hipError_t launch_gemm32(const GemmP& p, float* workspace, hipStream_t st);
hipError_t launch_colsum(const float* dz, float* gb, int n, int C, hipStream_t st);

}  // namespace alink

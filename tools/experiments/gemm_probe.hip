// Where does a SmallRes-sized gemm32 launch spend its time?  Built on the GPU box against a sed-modified copy of
// csrc/sgemm.hip (tools/experiments/gemm_probe.sh): variants without the MFMAs / without the global loads / both.
#include "alink_common.h"
#include "sgemm.h"
#include <cstdio>
#include <vector>
using namespace alink;
namespace alink { void set_error(const char*, ...) {} int hip_fail(hipError_t e, const char* w, const char*, int) { printf("HIP error %d in %s\n", (int)e, w); return -1; } }
static float time_gemm(GemmP g, int max_split, float* ws, int reps) {
    gemm32_plan_split(g, max_split);
#ifdef PROBE_PHASES
    launch_gemm32(g, ws, 0); launch_gemm32(g, ws, 0);
    hipDeviceSynchronize();
    long long h[16];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_probe), sizeof(h));
    printf("  cycles: prologue %lld, first loads issued %lld | FIRST stage: wait + LDS stores %lld, barrier %lld, next loads issued %lld, mfma %lld | all other stages %lld | epilogue %lld | total %lld\n",
           h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6], h[8] - h[7], h[8] - h[0]);
#endif
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch_gemm32(g, ws, 0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch_gemm32(g, ws, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / reps;
}
int main() {
    const int nb = 32;
    float *in, *w, *out, *ws;
    hipMalloc(&in, (size_t)2 * nb * 32 * 32 * 64 * 4); hipMalloc(&w, 9 * 64 * 64 * 4 + (1 << 20)); hipMalloc(&out, (size_t)2 * nb * 32 * 32 * 64 * 4);
    hipMalloc(&ws, (size_t)64 << 20);
    hipMemset(in, 0, (size_t)2 * nb * 32 * 32 * 64 * 4); hipMemset(w, 0, 9 * 64 * 64 * 4);
    {   // conv2 forward: 32 images 32x32x32 -> 30x30x32 (valid)
        GemmP g{}; g.A = in; g.B = w; g.C = out; g.Ho = 30; g.Wo = 30; g.M = nb * 900; g.N = 32; g.K = 288; g.ldb = 32; g.ldc = 32;
        g.amode = A_CONV; g.bmode = B_ROW; g.H = 32; g.W = 32; g.Ci = 32; g.pad = 0; g.relu = 1; g.bias = w;
        printf("conv2 fwd  M=%d N=32 K=288: %.1f us\n", g.M, time_gemm(g, 1, ws, 50));
    }
    {   // the same on 64 images: 450 workgroups = two per CU (2 x 64 KB of LDS), i.e. two waves per SIMD — does the launch take twice as long?
        GemmP g{}; g.A = in; g.B = w; g.C = out; g.Ho = 30; g.Wo = 30; g.M = 2 * nb * 900; g.N = 32; g.K = 288; g.ldb = 32; g.ldc = 32;
        g.amode = A_CONV; g.bmode = B_ROW; g.H = 32; g.W = 32; g.Ci = 32; g.pad = 0; g.relu = 1; g.bias = w;
        printf("conv2 fwd x2  M=%d N=32 K=288: %.1f us\n", g.M, time_gemm(g, 1, ws, 50));
    }
    {   // conv4 forward: 15x15x64 -> 13x13x64
        GemmP g{}; g.A = in; g.B = w; g.C = out; g.Ho = 13; g.Wo = 13; g.M = nb * 169; g.N = 64; g.K = 576; g.ldb = 64; g.ldc = 64;
        g.amode = A_CONV; g.bmode = B_ROW; g.H = 15; g.W = 15; g.Ci = 64; g.pad = 0; g.relu = 1; g.bias = w;
        printf("conv4 fwd  M=%d N=64 K=576: %.1f us\n", g.M, time_gemm(g, 1, ws, 50));
    }
    {   // conv1 forward: 32x32x3 -> 32x32x32 (same)
        GemmP g{}; g.A = in; g.B = w; g.C = out; g.Ho = 32; g.Wo = 32; g.M = nb * 1024; g.N = 32; g.K = 27; g.ldb = 32; g.ldc = 32;
        g.amode = A_CONV; g.bmode = B_ROW; g.H = 32; g.W = 32; g.Ci = 3; g.pad = 1; g.relu = 1; g.bias = w;
        printf("conv1 fwd  M=%d N=32 K=27: %.1f us\n", g.M, time_gemm(g, 1, ws, 50));
    }
    {   // wgrad layer 1: in a1 [32][32][32][32], dz [32][30][30][32]
        GemmP g{}; g.A = in; g.B = out; g.C = w; g.Ho = 30; g.Wo = 30; g.M = 289; g.N = 32; g.K = nb * 900; g.ldb = 32; g.ldc = 32;
        g.amode = A_CONVT; g.bmode = B_ROW; g.H = 32; g.W = 32; g.Ci = 32; g.pad = 0;
        printf("wgrad(1)   M=289 N=32 K=%d: %.1f us (split %d)\n", g.K, time_gemm(g, 128, ws, 50), 0);
    }
    {   // dgrad layer 1: dz [32][30][30][32] -> din [32][32][32][32]
        GemmP g{}; g.A = out; g.B = w; g.C = in; g.H = 30; g.W = 30; g.Ci = 32; g.Ho = 32; g.Wo = 32; g.pad = 2;
        g.M = nb * 1024; g.N = 32; g.K = 288; g.ldc = 32; g.amode = A_CONV; g.bmode = B_FLIP; g.act = in;
        printf("dgrad(1)   M=%d N=32 K=288: %.1f us\n", g.M, time_gemm(g, 1, ws, 50));
    }
    return 0;
}

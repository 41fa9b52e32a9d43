// wino_budget.hip — can a CU keep its matrix pipes busy under the LDS / VALU load a fused Winograd F(2x2,3x3) kernel in split
// precision would put on it?  A micro-benchmark of the INNER STRUCTURE only (no real data flow: operands are random f16 in LDS,
// results are summed into a sink), record: profiles/experiments/r04_winograd_sp.txt.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/bin/wino_budget tools/experiments/wino_budget.hip && wino_budget
//
// One 256-thread workgroup per CU (4 waves, one per SIMD, 512 registers each), or two (launch_bounds 2).  Per "stage" a wave
// issues M MFMAs (16x16x32 f16) fed by R operand ds_read_b128, plus the side traffic of the variant:
//   X  extra ds_read_b128 (the raw pixels the input transform reads)     W  ds_write_b128 (the transformed operands it writes)
//   D  LDS-DMA pieces of 1 KiB (the weight stream)                        V  v_fma_f32 on live values (the transform arithmetic)
// and ends the stage with s_waitcnt + s_barrier like the product kernels.  Reported: cycles per stage (s_memtime, median
// workgroup), MFMA-busy share = M x 16 / cycles, and the TFLOP/s that share means at the clock the chip held.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ void dma16(const void* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int M, int R, int X, int W, int D, int V, int WGS, int NTH = 256>
__global__ __launch_bounds__(NTH, WGS) void stage_kernel(const _Float16* __restrict__ gsrc, float* __restrict__ sink,
                                                         unsigned long long* __restrict__ cyc, int stages) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int OPB = WGS == 1 ? 65536 : 32768, OPM = OPB - 1, DMAOFF = OPB, WROFF = OPB + 16384, WRM = WGS == 1 ? 0x7FFF : 0x3FFF;
    // fill the operand area with data (random f16 from global)
    for (int i = tid; i < OPB / 16; i += NTH) *(f16x8*)(smem + i * 16) = *(const f16x8*)(gsrc + (size_t)((blockIdx.x * 4096 + i) & 0xFFFFF) * 8);
    __syncthreads();
    constexpr int NACC = M < 32 ? M : 32;
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float vv[8] = {1.f, 1.0001f, 0.9999f, 1.0002f, 0.9998f, 1.0003f, 0.9997f, 1.0004f};
    // conflict-free operand reads: lane l reads 16 B at row (l & 15) * 128 + swizzled 16-B piece, like the product kernels
    const int q = lane >> 4, lr = lane & 15;
    const int fbase = lr * 128 + ((q ^ ((lr >> 1) & 7)) << 4);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < stages; ++s) {
        const int rot = (s & 7) * 2048;
        f16x8 fr[R > 0 ? R : 1];
#pragma unroll
        for (int r = 0; r < R; ++r) fr[r] = *(const f16x8*)(smem + ((fbase + rot + r * 2048 + wave * 512) & OPM));
#pragma unroll
        for (int d = 0; d < D; ++d) dma16(gsrc + (size_t)(((s * D + d) * 256 + tid) & 0xFFFFF) * 8, smem + DMAOFF + ((d * 4 + wave) & 15) * 1024);
        f16x8 xr[X > 0 ? X : 1];
#pragma unroll
        for (int x = 0; x < X; ++x) xr[x] = *(const f16x8*)(smem + ((fbase + rot + OPB / 2 + x * 2048 + wave * 512) & OPM));
#pragma unroll
        for (int m = 0; m < M; ++m) {
            acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[m % (R > 0 ? R : 1)], fr[(m * 7 + 3) % (R > 0 ? R : 1)], acc[m % NACC], 0, 0, 0);
            // the transform arithmetic spread between the MFMAs
            if (V > 0) {
#pragma unroll
                for (int v = 0; v < (V + M - 1) / M; ++v) vv[(m + v) & 7] = fmaf(vv[(m + v) & 7], vv[(m + v + 3) & 7], (float)xr[(m + v) % (X > 0 ? X : 1)][v & 7]);
            }
        }
#pragma unroll
        for (int w = 0; w < W; ++w) {
            f16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (_Float16)(vv[j] + (X > 0 ? (float)xr[w % (X > 0 ? X : 1)][j] : 0.f));
            *(f16x8*)(smem + WROFF + ((w * 4096 + tid * 16) & WRM)) = o;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int j = 0; j < 8; ++j) t += vv[j];
    sink[blockIdx.x * NTH + tid] = t;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int M, int R, int X, int W, int D, int V, int WGS, int NTH = 256>
void run(const char* name, const _Float16* g, float* sink, unsigned long long* cyc, int ncu) {
    const int stages = 2000, grid = ncu * WGS;
    auto k = stage_kernel<M, R, X, W, D, V, WGS, NTH>;
    const size_t shm = WGS == 1 ? 65536 + 16384 + 32768 : 32768 + 16384 + 16384;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<grid, NTH, shm>>>(g, sink, cyc, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<grid, NTH, shm>>>(g, sink, cyc, stages);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    if (hipGetLastError() != hipSuccess) { printf("%-62s launch failed\n", name); return; }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cps = (double)h[grid / 2] / stages;                 // s_memtime ticks at 100 MHz on gfx950? report both views
    const double flops = (double)grid * (NTH / 64) * stages * M * 16.0 * 16 * 32 * 2;
    printf("%-62s %8.3f ms  %7.1f TFLOP/s issued   (%d MFMA / wave / stage, %.2f reads + %.2f writes + %.2f DMA per MFMA)\n", name, ms,
           flops / (ms * 1e-3) / 1e12, M, (double)(R + X) / M, (double)W / M, (double)D / M);
    (void)cps;
}

int main() {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount;
    _Float16* g;
    float* sink;
    unsigned long long* cyc;
    hipMalloc(&g, (size_t)(1 << 20) * 16 + 65536);
    hipMalloc(&sink, (size_t)ncu * 2 * 512 * 4);
    hipMalloc(&cyc, (size_t)ncu * 2 * 8);
    std::vector<_Float16> h((size_t)(1 << 20) * 8 + 32768);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    hipMemcpy(g, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    printf("%d CUs; every variant: 2000 stages per workgroup, random f16 operands\n", ncu);
    //            M   R   X  W  D   V  WGS
    run<56, 22, 0, 0, 4, 0, 2>("direct SP K-step (product kernel's shape), 2 WG/CU", g, sink, cyc, ncu);
    run<56, 22, 0, 0, 4, 0, 1>("direct SP K-step, 1 WG/CU", g, sink, cyc, ncu);
    run<56, 0, 0, 0, 0, 0, 1>("bare MFMA (operands in registers), 1 WG/CU", g, sink, cyc, ncu);
    run<48, 24, 0, 0, 0, 0, 1>("winograd operands only: 24 reads / 48 MFMA, 1 WG/CU", g, sink, cyc, ncu);
    run<48, 24, 0, 0, 8, 0, 1>("  + weight stream: 8 DMA pieces", g, sink, cyc, ncu);
    run<48, 24, 16, 4, 8, 0, 1>("  + transform traffic: 16 raw reads, 4 writes", g, sink, cyc, ncu);
    run<48, 24, 16, 4, 8, 300, 1>("  + transform arithmetic: 300 v_fma", g, sink, cyc, ncu);
    run<48, 24, 16, 4, 8, 150, 1>("  + transform arithmetic: 150 v_fma (all 16 planes at once)", g, sink, cyc, ncu);
    run<24, 16, 8, 2, 4, 150, 2>("half stage (32 ch), 2 WG/CU: 16+8 reads, 2 writes, 4 DMA, 150 fma / 24 MFMA", g, sink, cyc, ncu);
    run<12, 8, 0, 0, 0, 0, 1>("all-planes-in-registers block: 8 reads / 12 MFMA", g, sink, cyc, ncu);
    // ONE 8-wave workgroup per CU (two waves per SIMD share the big LDS images; 256 registers each: 32 x 32 blocks)
    run<56, 22, 0, 0, 4, 0, 1, 512>("direct SP K-step, ONE 8-wave WG/CU", g, sink, cyc, ncu);
    run<24, 16, 0, 0, 0, 0, 1, 512>("winograd 8-wave WG: 16 reads / 24 MFMA, operands only", g, sink, cyc, ncu);
    run<24, 16, 0, 0, 4, 0, 1, 512>("  + weight stream: 4 DMA pieces per wave", g, sink, cyc, ncu);
    run<24, 16, 8, 2, 4, 0, 1, 512>("  + transform traffic: 8 raw reads, 2 writes per wave", g, sink, cyc, ncu);
    run<24, 16, 8, 2, 4, 75, 1, 512>("  + transform arithmetic: 75 v_fma per wave (all planes at once)", g, sink, cyc, ncu);
    run<24, 16, 8, 2, 4, 150, 1, 512>("  + transform arithmetic: 150 v_fma per wave (plane by plane)", g, sink, cyc, ncu);
    return 0;
}

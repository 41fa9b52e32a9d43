// Fixed cost of a launch as a function of the LDS a workgroup reserves and of what the kernel's ends do.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDSF>
__global__ __launch_bounds__(256) void k_lds(float* out, int n) {
    __shared__ float s[LDSF > 0 ? LDSF : 1];
    if (LDSF > 0) { s[threadIdx.x] = (float)threadIdx.x; __syncthreads(); }
    if (n < 0) out[blockIdx.x * 256 + threadIdx.x] = LDSF > 0 ? s[(threadIdx.x + 1) & 255] : 1.f;
}
template <int LDSF>
__global__ __launch_bounds__(256) void k_store(float* out, int rows) {      // the gemm epilogue's store pattern: 16 rows x 32 columns per wave
    __shared__ float s[LDSF > 0 ? LDSF : 1];
    if (LDSF > 0) { s[threadIdx.x] = (float)threadIdx.x; __syncthreads(); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hh = lane >> 5;
    const int m0 = blockIdx.x * 128 + wave * 32;
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (row < rows) out[(size_t)row * 32 + l31] = LDSF > 0 ? s[(threadIdx.x + r) & 255] : (float)r;
    }
}
template <typename F>
static float timeit(F f, int reps = 100) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / reps;
}
int main() {
    float* out; hipMalloc(&out, 64 << 20);
    printf("empty, 225 blocks, no LDS:      %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_lds<0>, dim3(225), dim3(256), 0, 0, out, 1); }));
    printf("empty, 225 blocks, 12 KB LDS:   %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_lds<3072>, dim3(225), dim3(256), 0, 0, out, 1); }));
    printf("empty, 225 blocks, 48 KB LDS:   %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_lds<12288>, dim3(225), dim3(256), 0, 0, out, 1); }));
    printf("empty, 225 blocks, 64 KB LDS:   %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_lds<16384>, dim3(225), dim3(256), 0, 0, out, 1); }));
    printf("empty, 2048 blocks, 64 KB LDS:  %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_lds<16384>, dim3(2048), dim3(256), 0, 0, out, 1); }));
    printf("store 4 MB, 256 blocks, no LDS: %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_store<0>, dim3(256), dim3(256), 0, 0, out, 32768); }));
    printf("store 4 MB, 256 blocks, 64 KB:  %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_store<16384>, dim3(256), dim3(256), 0, 0, out, 32768); }));
    return 0;
}

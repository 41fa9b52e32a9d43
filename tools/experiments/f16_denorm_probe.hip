// Does v_mfma_f32_16x16x32_f16 keep f16 subnormal INPUTS, and does the f32 -> f16 conversion produce them?
// (decides where the split-precision mode may let lo halves fall: DESIGN.md §4)   hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void probe(float* out, const float* in) {
    const int lane = threadIdx.x;
    f16x8 a, b;
    unsigned short sub = 0x0001;               // smallest f16 subnormal, 2^-24
    _Float16 s;
    __builtin_memcpy(&s, &sub, 2);
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.f; b[j] = (_Float16)0.f; }
    if ((lane >> 4) == 0) { a[0] = s; b[0] = (_Float16)1024.f; }      // k = 0 only: A[row][0] = 2^-24, B[0][col] = 1024
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (lane == 0) out[0] = c[0];                                       // expect 2^-14 = 6.1035e-05 if kept
    // conversion: f32 1e-7 -> f16 (subnormal 0x0002 if kept, 0 if flushed), read back as f32
    _Float16 h = (_Float16)in[0];
    unsigned short hb;
    __builtin_memcpy(&hb, &h, 2);
    if (lane == 0) { out[1] = (float)hb; out[2] = (float)h; }
    // f16 subnormal -> f32
    if (lane == 0) out[3] = (float)s;
}
int main() {
    float *d, *di, h[4], hi = 1e-7f;
    hipMalloc(&d, 16); hipMalloc(&di, 4);
    hipMemcpy(di, &hi, 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, di);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("mfma(2^-24 * 1024) = %.6e (kept: 6.103516e-05)\ncvt f16(1e-7f) bits = %g value %.6e (kept: 2, 1.192093e-07)\nf32(f16 subnormal 0x0001) = %.6e (kept: 5.960464e-08)\n",
           h[0], h[1], h[2], h[3]);
    return 0;
}

// Does "a DS read beyond the workgroup's LDS allocation returns zero" hold for workgroups whose allocation does NOT start
// at LDS offset 0 (several workgroups per CU)?  Many workgroups of LDSK KB each; every lane reads base + far bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int KB>
__global__ void probe(float* out, int spin) {
    extern __shared__ __attribute__((aligned(16))) float buf[];
    const int n = KB * 256;
    for (int i = threadIdx.x; i < n; i += 64) buf[i] = 1.0f + (i & 1023);
    __syncthreads();
    // keep the workgroup resident for a while so that later ones land behind it
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)buf;
    const int lane = threadIdx.x;
    unsigned bits;
    if (lane < 10) bits = 1u << (18 + lane);
    else if (lane < 55) { int k = lane - 10, a = 18; while (k >= 27 - a) { k -= 27 - a; ++a; } bits = (1u << a) | (1u << (a + 1 + k)); }
    else bits = 0x0FFC0000u >> (lane - 55) & 0x0FFC0000u;
    // in-range bases spread over the allocation (top of it included)
    const unsigned inr = ((lane * 977u) % (unsigned)(KB * 1024 - 16384)) & ~15u;
    const unsigned far = base + inr + bits;
    f32x4 a0, a1;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:12288\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a0), "=&v"(a1) : "v"(far) : "memory");
    float s = 0.f;
    for (int j = 0; j < 4; ++j) s += fabsf(a0[j]) + fabsf(a1[j]);
    out[blockIdx.x * 64 + lane] = s;
}
template <int KB>
int run(int blocks) {
    float* d;
    hipMalloc(&d, blocks * 64 * 4);
    hipFuncSetAttribute((const void*)probe<KB>, hipFuncAttributeMaxDynamicSharedMemorySize, KB * 1024);
    probe<KB><<<blocks, 64, KB * 1024>>>(d, 200);
    std::vector<float> h(blocks * 64);
    hipError_t e = hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0, badblocks = 0;
    for (int b = 0; b < blocks; ++b) { int c = 0; for (int l = 0; l < 64; ++l) c += h[b * 64 + l] != 0.f; bad += c; badblocks += c > 0; }
    printf("LDS %3d KB per workgroup, %d workgroups: err=%d non-zero far reads %d in %d workgroups\n", KB, blocks, (int)e, bad, badblocks);
    hipFree(d);
    return bad;
}
int main() {
    run<20>(4096); run<40>(2048); run<52>(2048); run<56>(2048); run<70>(2048); run<80>(1024);
    return 0;
}

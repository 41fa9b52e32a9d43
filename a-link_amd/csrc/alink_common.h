// alink_common.h — shared host/device declarations for libalink_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>

#include "../../include/alink_hip.h"

namespace alink {

// ---- error plumbing (never throw across the C ABI) ---------------------------------------------
void set_error(const char* fmt, ...);
int  hip_fail(hipError_t e, const char* what, const char* file, int line);

#define ALINK_HIP(call)                                                          \
    do {                                                                         \
        hipError_t e__ = (call);                                                 \
        if (e__ != hipSuccess) return ::alink::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

#define ALINK_REQUIRE(cond, code, ...)          \
    do {                                        \
        if (!(cond)) {                          \
            ::alink::set_error(__VA_ARGS__);    \
            return (code);                      \
        }                                       \
    } while (0)

// ---- device ownership (one process per GPU, but several devices may be visible to a process) -------------
// Every handle remembers the device that was current when it was created; every entry point that allocates,
// copies or launches makes that device current for the duration of the call and restores the caller's
// device afterwards.  Entry points without a handle take the device from the allocation they are handed.
// The caller's `stream` must belong to the same device (torch: the stream of the tensor's device).
static inline int current_device() {
    int d = -1;
    return hipGetDevice(&d) == hipSuccess ? d : -1;
}
static inline int device_of_pointer(const void* p) {
    if (p) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, p) == hipSuccess && a.type == hipMemoryTypeDevice) return a.device;
        (void)hipGetLastError();                        // host / unregistered pointer: not an error here
    }
    return current_device();
}
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (dev < 0 || hipGetDevice(&prev) != hipSuccess || prev == dev) return;
        switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// ---- vector types -------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16   bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float    f32x4;
typedef __attribute__((ext_vector_type(16))) float   f32x16;

// ---- implicit-GEMM convolution launch descriptor -------------------------------------------------
// Activations are NHWC in T (bf16 | f16); weights are [Cout][ksz*ksz*Cin] in T with the rows of
// every 64-row block permuted by `perm64` (see conv_igemm.hip) so that a lane's 16 accumulator
// values are 16 consecutive output channels.
struct ConvParams {
    const void*  in;      // [N][H][W][Cin]
    const void*  wgt;     // [Cout][K]  (row-permuted)
    const float* bias;    // [ncls][Cout] f32, natural channel order; ncls = border_cls ? 9 : 1
    const float* alpha;   // [Cout] PReLU slopes, or nullptr
    const void*  resid;   // [M][Cout] T, or nullptr
    const void*  dact;    // [M][Cout] T, or nullptr: backward mode — instead of the forward PReLU the result is
                          // multiplied by PReLU'(z) read off the stored forward activation: 1 if dact > 0 else alpha[c]
    void*        out;     // [M][Cout] T;  split-K: f32 slabs [splitk][M][Cout]
    const void*  zero;    // >= 256 B of zeros (source for padded taps / tail rows)
    // conv_igemm only: a fused 1x1 projection shortcut (the first unit of a stage: out = conv(in) + conv1x1_stride(in2)).
    // in2 = [N][H][W][Cin2] (same grid as `in`), sampled at (oy * stride, ox * stride); its weights follow the ksz*ksz*Cin
    // columns of every weight row (row pitch ksz*ksz*Cin + Cin2); its K-steps run after the last tap.  nullptr = none.
    const void*  in2;
    int Cin2;
    int in2_compact;      // in2 holds only the sampled pixels: [N][Ho][Wo][Cin2] (the fused front kernel writes it that way)
    int N, H, W, Cin, Cout, Ho, Wo, stride, ksz, pad, M;
    int border_cls;       // bias class chosen by output position (3x3, stride 1, pad 1 only)
    int splitk;           // 1 = fused epilogue; >1 = f32 partial slabs
    int ksteps_per_split;
    int post_relu;        // ReLU after the residual add (bottleneck units: relu(conv + bias + shortcut))
    int fine;             // linear-tile kernel: 1 = 64-channel workgroups instead of 128 (small batches: twice the
                          // workgroups, same weights, bit-identical sums)
    int stagger;          // linear-tile kernel: the workgroup in the CU's second wave slot starts this many x 1024 cycles late
    int ablate;           // diagnostic timing-only modes of conv3x3_direct (0 = normal)
    // ALINK_DT_F16X2 only (powers of two, exact): stored value = true value x 2^e, e per tensor.
    float acc_scale;      // 2^(e_out - e_in - e_w): accumulator -> output units
    float bias_scale;     // 2^e_out
    float res_scale;      // 2^(e_out - e_resid)
    int nprod;            // ALINK_DT_F16X2: 3 (0 = 3) = hi x W_hi + hi x W_lo + lo x W_hi, the exact mode; 1 = hi x W_hi only — the
                          // SCREENING form on the same handle: one matrix-core product like plain f16, but under the handle's
                          // power-of-two scales (no range to leave) and with the residual stream still carried as hi + lo
    void* stamps;         // diagnostic: 4 x u64 s_memtime stamps per workgroup, or nullptr
};

// Launchers (host).  All enqueue on `stream` and return a HIP error.
hipError_t launch_conv_igemm(int dtype, const ConvParams& p, hipStream_t stream);
double     conv_flops(const ConvParams& p);

struct StemParams {
    const void*  in;      // pixels, layout per `layout`
    const void*  wgt;     // [64'][64] T: k = ky*16 + kx*3 + c (27 real, 37 zero: two MFMA K steps), rows permuted
    const float* bias;    // [C0]
    const float* alpha;   // [C0]
    void*        out;     // [N][H][W][C0] T
    int N, H, W, C0, layout;
    // input normalisation in the loader: v = (pixel[c] - sub[c']) * mul, stored as network channel c'
    // (c' = 2 - c when flip: RGB pixels into a BGR-trained network).  IR-ResNet: sub = 127.5, mul = 1/128.
    float sub[3], mul;
    int flip;
    // ALINK_DT_F16X2: wgt = [64'][hi 32 | lo 32]; the loader stores normalised pixels x 2^8; out = [N][H][W][hi 64 | lo 64]
    float acc_scale, bias_scale;
};
hipError_t launch_stem(int dtype, const StemParams& p, hipStream_t stream);

struct FcFinishParams {
    const float* slabs;   // [S][M][E]
    const float* bias;    // [E]
    float*       out;     // [M][E], L2-normalised rows
    float*       norms;   // [M] row norms before normalisation (as divided by: 1 for a zero row), or nullptr
    int S, M, E;
    float scale;          // the slab sums are multiplied by this before the bias (ALINK_DT_F16X2: 2^-(e_in + e_w); else 1)
    int*  nonfinite;      // or nullptr: set to 1 when a row comes out non-finite (16-bit range left somewhere upstream)
};
hipError_t launch_fc_finish(const FcFinishParams& p, hipStream_t stream);
hipError_t launch_absmax_f16(const void* x, size_t n_elements, unsigned* out_bits, hipStream_t stream);
// epilogue of a split-K convolution: p as for the fused launch (out = the real output), slabs [S][M][Cout] f32
hipError_t launch_conv_split_finish(int dtype, const ConvParams& p, const float* slabs, int S, hipStream_t stream);

// position of natural channel c (0..63 within its 64-block) in the permuted weight rows
static inline int perm64_row_of_channel(int c) {
    // MFMA tile t (0..3), row 4q+j  <->  channel 16q + 4t + j
    int q = c >> 4, t = (c >> 2) & 3, j = c & 3;
    return 16 * t + 4 * q + j;
}
// same for kernels whose waves own 32 channels (2 MFMA tiles): tile t (0..1), row 4q+j <-> 8q + 4t + j
static inline int perm32_row_of_channel(int c) {
    int q = c >> 3, t = (c >> 2) & 1, j = c & 3;
    return 16 * t + 4 * q + j;
}
// conv3x3_direct with 4 MFMA tiles per wave: tile t = 2 th + tl, row 4q+j  <->  channel 32 th + 8q + 4 tl + j,
// i.e. a lane holds two runs of 8 consecutive channels, 32 apart
static inline int perm64b_row_of_channel(int c) {
    int th = c >> 5, q = (c >> 3) & 3, tl = (c >> 2) & 1, j = c & 3;
    return 16 * (2 * th + tl) + 4 * q + j;
}
// weight row of output channel co; cpl: 16 = perm64 (conv_igemm, stems), 17 = perm64b, 8 = perm32
static inline int permuted_row(int co, int cpl) {
    if (cpl == 17) return (co & ~63) + perm64b_row_of_channel(co & 63);
    return cpl == 16 ? (co & ~63) + perm64_row_of_channel(co & 63) : (co & ~31) + perm32_row_of_channel(co & 31);
}

// conv3x3_direct.hip
int        direct_variant(int ksz, int stride, int pad, int H, int W, int Cin, int Cout);
int        direct_variant_tiles(int ksz, int stride, int pad, int H, int W, int Cin, int Cout);   // without the rolling-row kernel
// conv3x3_c64.hip: rolling-row kernel for 64 -> 64 channel layers, weights in registers (variant 21)
int        c64_variant(int ksz, int stride, int pad, int H, int W, int Cin, int Cout);
hipError_t c64_set_attributes();
hipError_t launch_conv3x3_c64(int variant, int dtype, const ConvParams& p, hipStream_t st);
// conv3x3_s2c64.hip: the direct stride-2 kernel for 112 x 112 x 64 -> 56 x 56 x 64, projection shortcut as extra K-steps (variant 25)
int        s2c64_variant(int ksz, int stride, int pad, int H, int W, int Cin, int Cout);
hipError_t s2c64_set_attributes();
hipError_t launch_conv3x3_s2c64(int variant, int dtype, const ConvParams& p, hipStream_t st);
// front_c64.hip: stem + the first unit's conv1 in one rolling-row launch (the stem's activation stays in LDS; the quarter
// of it the unit's projection shortcut samples goes to `xs`, [N][56][56][64])
bool       front_c64_applies(int dtype, int H, int W, int C0, int Cout);
hipError_t front_c64_set_attributes();
hipError_t launch_front_c64(int dtype, const ConvParams& conv1, const StemParams& stem, void* xs, bool slopes_le_1, hipStream_t st);
int        direct_variant_cpl(int v);
hipError_t direct_set_attributes();
hipError_t launch_conv3x3_direct(int variant, int dtype, const ConvParams& p, hipStream_t st);
// linear-tile variants (conv3x3_linear.hip), reached through the four functions above as variants 11..13
int        linear_variant(int ksz, int stride, int pad, int H, int W, int Cin, int Cout);
int        linear_variant_x2(int ksz, int stride, int pad, int H, int W, int Cin, int Cout);   // split precision: + the 112-wide layer
int        linear_variant_cpl(int v);
hipError_t linear_set_attributes();
hipError_t linear_check_contract();   // probes the LDS out-of-range read contract; disables the linear kernel if it fails
hipError_t launch_conv3x3_linear(int variant, int dtype, const ConvParams& p, hipStream_t st);
// conv3x3_lat.hip: the latency form for launches of a handful of images (one wave per 32 x 32 output block, operands straight from L2)
bool       conv3x3_lat_applies(int dtype, const ConvParams& p);
hipError_t launch_conv3x3_lat(int dtype, const ConvParams& p, hipStream_t stream);
bool       conv_gemm_lat_applies(int dtype, const ConvParams& p);     // a lone image's stride-2 3x3 (+ fused shortcut) and 1x1 layers
hipError_t launch_conv_gemm_lat(int dtype, const ConvParams& p, hipStream_t stream);

// backward.hip (input-gradient pass)
hipError_t launch_l2norm_bwd(int dtype, const float* g, const float* e, const float* norms, void* dz, int M, int E,
                             hipStream_t st);
hipError_t launch_zero_insert(int dtype, const void* in, void* out, int N, int H, int W, int Ho, int Wo, int C,
                              hipStream_t st);
hipError_t launch_stem_bwd(int dtype, const void* dy0, const void* y0, const float* w, const float* alpha, float* dpix,
                           int N, int H, int W, float mul, int nchw, hipStream_t st);

// host-side conversions
uint16_t f32_to_bf16_rne(float f);
uint16_t f32_to_f16_rne(float f);

int init_kernels();   // function attributes (dynamic LDS sizes)

// ReLU and max that KEEP a NaN (v_max_f32 returns the other operand): 16-bit float storage that left its range turns into
// inf - inf = NaN in the next convolution, and a ReLU written as fmaxf(v, 0) would turn that NaN into a plausible 0 — the
// range flag is raised where a NaN reaches the pooled features, so it must get there (PReLU's select already keeps it).
__device__ __forceinline__ float relu_keep_nan(float v) { return v < 0.f ? 0.f : v; }
__device__ __forceinline__ float max_keep_nan(float a, float b) { return (a < b || b != b) ? b : a; }

}  // namespace alink
#include <map>
#include <string>
#include <vector>
namespace alink {
// backbone_f32.hip: the float32 precision mode of the IR backbone (alink_ir_cfg.dtype = ALINK_DT_F32)
struct F32Net;
F32Net* f32net_build(const std::map<std::string, std::vector<float>>& raw, const alink_ir_cfg& cfg);
void    f32net_destroy(F32Net* n);
size_t  f32net_workspace_bytes(const F32Net* n, int N);
int     f32net_embed(const F32Net* n, const void* dev_in, int layout, int N, float* dev_out, void* ws, size_t ws_bytes,
                     hipStream_t st);

}  // namespace alink

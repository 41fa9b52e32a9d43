// select.hip — pool scoring and top-k on device (HBM-bound integer/compare work, no MFMA).
//
// Replaces, for large pools, the host NumPy at
//   code/uncertainty.py:15-60   (_proba_uncertainty / _proba_margin / _proba_entropy)
//   code/uncertainty.py:155,183,213 (multi_argmax top-n)   and   code/ALINK_arc.py:170-181
//   (disparity = -|M2[:,c] - M1[:,c]| ; argsort(...)[:int(P*ratio)]).
//
// top-k = exact radix SELECT on unique 64-bit keys (order-preserving score bits << 32 | index, so
// ties break towards the lower index deterministically — the reference's np.argsort/argpartition
// tie order is unspecified), then a bitonic sort of just the k survivors.  Each select pass streams
// the P keys once (8 B/key), histogramming in LDS.
#include "alink_common.h"

namespace alink {
namespace {

struct SelState {
    unsigned long long prefix;   // bits decided so far (high bits), rest zero
    unsigned long long mask;     // mask of decided bits
    unsigned long long krem;     // rank still to find inside the matching set (1-based)
    unsigned int count;          // compaction counter
    unsigned int pad;
    unsigned int hist[256];
};

__device__ __forceinline__ unsigned int ord32(float f) {
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // ascending float order as unsigned
}

__global__ void score_kernel(int kind, const float* __restrict__ a, const float* __restrict__ b, int col,
                             long long P, int C, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float* p = a + i * C;
    float r;
    if (kind == ALINK_SCORE_UNCERTAINTY) {
        float m = p[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, p[c]);
        r = 1.f - m;
    } else if (kind == ALINK_SCORE_MARGIN) {
        if (C == 1) { r = 0.f; }
        else {
            float m1 = -INFINITY, m2 = -INFINITY;
            for (int c = 0; c < C; ++c) {
                const float v = p[c];
                if (v > m1) { m2 = m1; m1 = v; } else if (v > m2) { m2 = v; }
            }
            r = m1 - m2;
        }
    } else if (kind == ALINK_SCORE_ENTROPY) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += p[c];
        float e = 0.f;
        for (int c = 0; c < C; ++c) {
            const float q = p[c] / s;
            e += (q > 0.f) ? -q * logf(q) : 0.f;
        }
        r = e;
    } else {
        r = -fabsf(b[i * C + col] - p[col]);
    }
    out[i] = r;
}

__global__ void make_keys_kernel(const float* __restrict__ s, long long P, int largest,
                                 unsigned long long* __restrict__ keys, SelState* st, unsigned long long k) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i == 0) {
        st->prefix = 0; st->mask = 0; st->krem = k; st->count = 0;
    }
    if (i < 256) st->hist[i] = 0;
    if (i >= P) return;
    unsigned int o = ord32(s[i]);
    if (largest) o = ~o;
    keys[i] = ((unsigned long long)o << 32) | (unsigned long long)(unsigned int)i;
}

__global__ __launch_bounds__(256) void hist_kernel(const unsigned long long* __restrict__ keys, long long P,
                                                   SelState* st, int shift) {
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long prefix = st->prefix, mask = st->mask;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long long)gridDim.x * 256) {
        const unsigned long long kx = keys[i];
        if ((kx & mask) == prefix) atomicAdd(&h[(kx >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], h[threadIdx.x]);
}

__global__ void scan_kernel(SelState* st, int shift) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long k = st->krem, cum = 0;
    int d = 0;
    for (; d < 256; ++d) {
        const unsigned long long c = st->hist[d];
        if (cum + c >= k) break;
        cum += c;
    }
    if (d > 255) d = 255;
    st->krem = k - cum;
    st->prefix |= (unsigned long long)d << shift;
    st->mask |= 255ull << shift;
    for (int i = 0; i < 256; ++i) st->hist[i] = 0;
}

__global__ void compact_kernel(const unsigned long long* __restrict__ keys, long long P, SelState* st,
                               unsigned long long* __restrict__ out, unsigned int cap) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const unsigned long long kx = keys[i];
    if (kx <= st->prefix) {
        const unsigned int slot = atomicAdd(&st->count, 1u);
        if (slot < cap) out[slot] = kx;
    }
}

__global__ void pad_kernel(unsigned long long* out, unsigned int k, unsigned int n2) {
    const unsigned int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= k && i < n2) out[i] = ~0ull;
}

constexpr int CHUNK = 4096;   // keys sorted per workgroup in LDS (32 KB)

__device__ __forceinline__ void cmpx(unsigned long long* s, unsigned int i, unsigned int j, bool asc) {
    const unsigned long long a = s[i], b = s[j];
    if ((a > b) == asc) { s[i] = b; s[j] = a; }
}

// sizes 2 .. min(n2, CHUNK): full bitonic build inside LDS
__global__ __launch_bounds__(256) void bitonic_local_sort(unsigned long long* keys, unsigned int n2) {
    __shared__ unsigned long long s[CHUNK];
    const unsigned int base = blockIdx.x * CHUNK, n = min((unsigned int)CHUNK, n2);
    for (unsigned int i = threadIdx.x; i < n; i += 256) s[i] = keys[base + i];
    __syncthreads();
    for (unsigned int size = 2; size <= n; size <<= 1)
        for (unsigned int stride = size >> 1; stride > 0; stride >>= 1) {
            for (unsigned int t = threadIdx.x; t < n / 2; t += 256) {
                const unsigned int i = 2 * t - (t & (stride - 1)), j = i + stride;
                cmpx(s, i, j, ((base + i) & size) == 0);
            }
            __syncthreads();
        }
    for (unsigned int i = threadIdx.x; i < n; i += 256) keys[base + i] = s[i];
}
__global__ void bitonic_global_step(unsigned long long* keys, unsigned int n2, unsigned int size,
                                    unsigned int stride) {
    const unsigned int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n2 / 2) return;
    const unsigned int i = 2 * t - (t & (stride - 1)), j = i + stride;
    const unsigned long long a = keys[i], b = keys[j];
    const bool asc = (i & size) == 0;
    if ((a > b) == asc) { keys[i] = b; keys[j] = a; }
}
// strides CHUNK/2 .. 1 of the merge for `size`
__global__ __launch_bounds__(256) void bitonic_local_merge(unsigned long long* keys, unsigned int size) {
    __shared__ unsigned long long s[CHUNK];
    const unsigned int base = blockIdx.x * CHUNK;
    for (unsigned int i = threadIdx.x; i < CHUNK; i += 256) s[i] = keys[base + i];
    __syncthreads();
    for (unsigned int stride = CHUNK >> 1; stride > 0; stride >>= 1) {
        for (unsigned int t = threadIdx.x; t < CHUNK / 2; t += 256) {
            const unsigned int i = 2 * t - (t & (stride - 1)), j = i + stride;
            cmpx(s, i, j, ((base + i) & size) == 0);
        }
        __syncthreads();
    }
    for (unsigned int i = threadIdx.x; i < CHUNK; i += 256) keys[base + i] = s[i];
}

__global__ void emit_kernel(const unsigned long long* __restrict__ keys, const float* __restrict__ scores,
                            int k, int32_t* __restrict__ idx, float* __restrict__ vals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= k) return;
    const unsigned int id = (unsigned int)(keys[i] & 0xffffffffull);
    idx[i] = (int32_t)id;
    if (vals) vals[i] = scores[id];
}

unsigned int next_pow2(unsigned int x) {
    unsigned int p = 1;
    while (p < x) p <<= 1;
    return p;
}

}  // namespace
}  // namespace alink

using namespace alink;

extern "C" {

int alink_score(int kind, const float* dev_probs, const float* dev_b, int col, int64_t P, int C,
                float* dev_scores, void* stream) {
    ALINK_REQUIRE(dev_probs && dev_scores, ALINK_EINVAL, "NULL argument");
    DeviceGuard dg(device_of_pointer(dev_probs));
    ALINK_REQUIRE(kind >= 0 && kind <= 3, ALINK_EINVAL, "unknown score kind %d", kind);
    ALINK_REQUIRE(C >= 1, ALINK_EINVAL, "C must be >= 1");
    ALINK_REQUIRE(kind != ALINK_SCORE_DISPARITY || (dev_b && col >= 0 && col < C), ALINK_EINVAL,
                  "disparity needs dev_b and 0 <= col < C");
    if (P == 0) return ALINK_OK;
    ALINK_REQUIRE(P > 0, ALINK_EINVAL, "negative P");
    hipLaunchKernelGGL(score_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, kind,
                       dev_probs, dev_b, col, (long long)P, C, dev_scores);
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

size_t alink_topk_scratch_bytes(int64_t P, int k) {
    if (P <= 0 || k <= 0) return 0;
    const size_t n2 = next_pow2((unsigned int)k);
    return (((size_t)P * 8 + 255) & ~(size_t)255) + ((n2 * 8 + 255) & ~(size_t)255) + 2048;
}

int alink_topk(const float* dev_scores, int64_t P, int k, int largest, int32_t* dev_idx, float* dev_vals,
               void* dev_scratch, void* stream) {
    ALINK_REQUIRE(dev_scores && dev_idx && dev_scratch, ALINK_EINVAL, "NULL argument");
    DeviceGuard dg(device_of_pointer(dev_scores));
    ALINK_REQUIRE(P > 0 && P < (1ll << 31), ALINK_EINVAL, "P=%lld outside 1..2^31-1", (long long)P);
    ALINK_REQUIRE(k > 0 && k <= P, ALINK_EINVAL, "k=%d outside 1..P", k);
    ALINK_REQUIRE(((uintptr_t)dev_scratch & 255) == 0, ALINK_EINVAL, "scratch must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const unsigned int n2 = next_pow2((unsigned int)k);
    char* base = (char*)dev_scratch;
    unsigned long long* keys = (unsigned long long*)base;
    unsigned long long* out = (unsigned long long*)(base + (((size_t)P * 8 + 255) & ~(size_t)255));
    SelState* state = (SelState*)((char*)out + (((size_t)n2 * 8 + 255) & ~(size_t)255));
    const unsigned nb = (unsigned)((P + 255) / 256);
    hipLaunchKernelGGL(make_keys_kernel, dim3(nb), dim3(256), 0, st, dev_scores, (long long)P, largest, keys,
                       state, (unsigned long long)k);
    const unsigned hb = nb < 2048 ? nb : 2048;
    for (int shift = 56; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(hist_kernel, dim3(hb), dim3(256), 0, st, keys, (long long)P, state, shift);
        hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(64), 0, st, state, shift);
    }
    hipLaunchKernelGGL(compact_kernel, dim3(nb), dim3(256), 0, st, keys, (long long)P, state, out, (unsigned)k);
    if (n2 > (unsigned)k)
        hipLaunchKernelGGL(pad_kernel, dim3((n2 + 255) / 256), dim3(256), 0, st, out, (unsigned)k, n2);
    const unsigned chunks = n2 > CHUNK ? n2 / CHUNK : 1;
    hipLaunchKernelGGL(bitonic_local_sort, dim3(chunks), dim3(256), 0, st, out, n2);
    for (unsigned int size = 2 * CHUNK; size <= n2 && size != 0; size <<= 1) {
        for (unsigned int stride = size >> 1; stride >= CHUNK; stride >>= 1)
            hipLaunchKernelGGL(bitonic_global_step, dim3((n2 / 2 + 255) / 256), dim3(256), 0, st, out, n2, size,
                               stride);
        hipLaunchKernelGGL(bitonic_local_merge, dim3(n2 / CHUNK), dim3(256), 0, st, out, size);
    }
    hipLaunchKernelGGL(emit_kernel, dim3((k + 255) / 256), dim3(256), 0, st, out, dev_scores, k, dev_idx,
                       dev_vals);
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

}  // extern "C"

// noise.hip — the A2-LINK perturbation stage on device (HBM-bound elementwise / scatter work).
//
// Replaces the host NumPy of reference code/noise.py (Gaussian :33-45, SaltPepper :48-65, Poisson
// :68-76, Speckle :79-88, Perlin :91-150), code/attack.py:5-29 (perturb_image) and the cv2.resize of
// code/committee.py:22-26.  The reference draws from the unseeded global np.random stream
// (SURVEY.md §5), so there is no stream contract to keep: random numbers here come from a
// counter-based Philox4x32-10 generator keyed by (seed, element), which makes every kernel
// independent of its launch shape and lets the CPU oracle reproduce the exact same draws.
//
// Every kernel streams its input once and writes its output once (4 B + 4 B per element, 16-B lane
// accesses); nothing is staged through LDS because nothing is reused.
#include "alink_common.h"
#include "philox.h"

namespace alink {
namespace {

// 53-bit-ish uniform in (0, 1) from two words (double precision consumers: Poisson)
__device__ __forceinline__ double u01d(unsigned int hi, unsigned int lo) {
    const unsigned long long v = ((unsigned long long)(hi >> 5) << 26) | (lo >> 6);   // 27 + 26 bits
    return ((double)v + 0.5) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ void box_muller(unsigned int a, unsigned int b, float& z0, float& z1) {
    const float r = sqrtf(-2.f * logf(u01(a)));
    const float th = 6.283185307179586f * u01(b);
    z0 = r * cosf(th);
    z1 = r * sinf(th);
}

// ---- Gaussian / Speckle ---------------------------------------------------------------------------
// mode 0 (code/noise.py:40-44):  out = x + (mean + sigma * z)
// mode 1 (code/noise.py:84-87):  out = x + x * (z / div)
// mode 2 (PGD's random start, an extension): out = x + (lo + (hi - lo) * u), u the element's 24-bit uniform
// Element e (= offset + i) takes normal number (e & 3) of Philox block e >> 2: a row range of a logical array (a rank's
// shard of the pair batch) draws what the whole array would have drawn there, whatever the offset's alignment.
struct AffineNoise {
    const float* in;
    float* out;
    long long count;
    unsigned long long seed, offset;
    float p0, p1;
    int mode;
    int vec;              // offset % 4 == 0 and both pointers 16-byte aligned: 16-byte lane accesses
};

__device__ __forceinline__ float apply_noise(const AffineNoise& p, float x, float z) {
    return p.mode == 0 ? x + (p.p0 + p.p1 * z) : (p.mode == 1 ? x + x * (z / p.p0) : x + (p.p0 + (p.p1 - p.p0) * z));
}

__global__ __launch_bounds__(256) void affine_noise_kernel(const AffineNoise p) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;     // Philox block (offset >> 2) + g: 4 elements
    const long long i0 = g * 4 - (long long)(p.offset & 3);           // local index of the block's first element
    if (i0 >= p.count) return;
    const U4 r = draw(p.seed, (p.offset >> 2) + (unsigned long long)g, 0, ST_NORMAL);
    float z[4];
    if (p.mode == 2) {
        z[0] = u01(r.x); z[1] = u01(r.y); z[2] = u01(r.z); z[3] = u01(r.w);
    } else {
        box_muller(r.x, r.y, z[0], z[1]);
        box_muller(r.z, r.w, z[2], z[3]);
    }
    if (p.vec && i0 + 4 <= p.count) {
        const f32x4 x = *(const f32x4*)(p.in + i0);
        f32x4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = apply_noise(p, x[j], z[j]);
        *(f32x4*)(p.out + i0) = y;
    } else {
        for (int j = 0; j < 4; ++j)
            if (i0 + j >= 0 && i0 + j < p.count) p.out[i0 + j] = apply_noise(p, p.in[i0 + j], z[j]);
    }
}

// ---- Bernoulli keep-masks (Dropout of the SmallRes tower, code/siamese.py:146,153) -----------------------------
// (philox.h: keep_mask_block)
// `first`: the element out[0] stands for — a rank that trains rows lo : hi of a batch draws those rows' masks exactly
__global__ __launch_bounds__(256) void keep_mask_kernel(unsigned char* __restrict__ out, long long count, float keep,
                                                        unsigned long long seed, unsigned long long first) {
    keep_mask_block(out, count, keep, seed, first, (first >> 2) + (unsigned long long)blockIdx.x * 256 + threadIdx.x);
}

// ---- Salt & pepper (code/noise.py:54-65, tuple-index semantics of NumPy < 1.23) --------------------
// One workgroup per image: copy, then n_salt writes of 1 at (r, c, ch) with r in [0, H-2],
// c in [0, W-2], ch in [0, C-2] (np.random.randint(0, i - 1) excludes the high end), then n_pepper
// writes of 0 — pepper wins where both land, as in the reference's statement order.
struct SaltPepperP {
    const float* in;
    float* out;
    int H, W, C, n_salt, n_pepper;
    unsigned long long seed, first_image;
};

__device__ __forceinline__ int bounded(unsigned int x, int range) {
    return (int)(((unsigned long long)x * (unsigned long long)(unsigned int)range) >> 32);
}

__global__ __launch_bounds__(256) void saltpepper_kernel(const SaltPepperP p) {
    const int img = blockIdx.x, tid = threadIdx.x;
    const long long per = (long long)p.H * p.W * p.C;
    const float* src = p.in + img * per;
    float* dst = p.out + img * per;
    if (src != dst)
        for (long long i = tid; i < per; i += 256) dst[i] = src[i];
    __syncthreads();
#pragma unroll 1
    for (int phase = 0; phase < 2; ++phase) {
        const int n = phase == 0 ? p.n_salt : p.n_pepper;
        for (int k = tid; k < n; k += 256) {
            const U4 r = draw(p.seed, ((p.first_image + (unsigned long long)img) << 32) | (unsigned int)k, (unsigned int)phase, ST_SALTPEPPER);
            const int rr = bounded(r.x, p.H - 1), cc = bounded(r.y, p.W - 1), ch = bounded(r.z, p.C - 1);
            dst[((long long)rr * p.W + cc) * p.C + ch] = phase == 0 ? 1.f : 0.f;
        }
        __syncthreads();
    }
}

// ---- Perlin (code/noise.py:95-150) --------------------------------------------------------------------
// noise(y, x) = sum over 3 octaves ns of  sum_{r,s in {0,1}} wa[r] * wb[s] * <offset_rs, v[i + r][j + s]>
//   i = y / ns, a = y % ns, j = x / ns, b = x % ns, wa = (1 - q(a/ns), q(a/ns)), wb likewise for b,
//   offset_rs = (b - s*ns, a - r*ns), q(t) = t^3 (t (6 t - 15) + 10), v = unit vectors on a
//   (size/ns + 1)^2 grid (:100-107).  The same noise is added to every channel (:148).
struct PerlinP {
    const float* in;
    float* out;
    const float* vec;     // [n_images][nodes_total][2] unit vectors, octave after octave
    int size, C, ns[3], goff[3], nodes_total;
};

__device__ __forceinline__ float quintic(float t) { return t * t * t * (t * (t * 6.f - 15.f) + 10.f); }

__global__ __launch_bounds__(256) void perlin_kernel(const PerlinP p) {
    const int img = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= p.size * p.size) return;
    const int y = pix / p.size, x = pix - y * p.size;
    const float* vimg = p.vec + (size_t)img * p.nodes_total * 2;
    float noise = 0.f;
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        const int ns = p.ns[o], gs = p.size / ns + 1;
        const int i = y / ns, a = y - i * ns, j = x / ns, b = x - j * ns;
        const float qa = quintic((float)a / (float)ns), qb = quintic((float)b / (float)ns);
        const float* v = vimg + (size_t)p.goff[o] * 2;
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float* node = v + ((size_t)(i + r) * gs + (j + s)) * 2;
                const float d = (float)(b - s * ns) * node[0] + (float)(a - r * ns) * node[1];
                acc += (r ? qa : 1.f - qa) * (s ? qb : 1.f - qb) * d;
            }
        noise += acc;
    }
    const size_t base = ((size_t)img * p.size * p.size + pix) * p.C;
    for (int c = 0; c < p.C; ++c) p.out[base + c] = p.in[base + c] + noise;
}

// node e = first + i of the logical [all images][nodes] array takes word (e & 3) of Philox block e >> 2
__global__ void perlin_vectors_kernel(float* __restrict__ vec, long long n, unsigned long long seed, unsigned long long first) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;      // 4 nodes per Philox block
    const long long i0 = g * 4 - (long long)(first & 3);
    if (i0 >= n) return;
    const U4 r = draw(seed, (first >> 2) + (unsigned long long)g, 0, ST_PERLIN);
    const unsigned int w[4] = {r.x, r.y, r.z, r.w};
    for (int j = 0; j < 4; ++j) {
        if (i0 + j < 0 || i0 + j >= n) continue;
        const float phi = 6.283185307179586f * u01(w[j]);       // np.random.uniform(0, 2 pi) (:102)
        vec[(i0 + j) * 2 + 0] = cosf(phi);
        vec[(i0 + j) * 2 + 1] = sinf(phi);
    }
}

// ---- Poisson (code/noise.py:72-76) -------------------------------------------------------------------
// vals = 2^ceil(log2(number of unique values of the image)); out = Poisson(x * vals) / vals.
//
// Pass 1 (unique_count_kernel, one 1024-thread workgroup per image, everything on chip): images reaching this noise are
// mostly integer-valued 0..255 (PIL -> float32, code/readDFW.py:82): every lane marks seen[(int)x] with a plain LDS store
// (all writers store 1: no atomic) and the count is the number of marks — one streaming read, no global atomic.  An
// image with any other value (a bilinear resize makes them, code/readDFW.py:82) is counted EXACTLY by an open-addressing
// hash set of the float bit patterns in LDS (32768 slots = 128 KiB, `ds_cmpst` probes), the key space cut into 2^p parts by
// hash bits so that a part's load stays below 1/2; the image is re-read once per part (from L2).  Round 4's form kept the
// table in global memory (512 KiB of caller scratch per image, one L2 atomic per element): 1.5-2.2 ms per 1,024 images.
// Pass 2 (poisson_kernel) samples: lam < 10 by the product-of-uniforms method, lam >= 10 by Hörmann's transformed
// rejection (PTRS) — the two methods numpy's legacy generator uses.  Round 5: for lam < 2^24 (x <= 255 with vals <= 65536:
// every image the loop produces) PTRS runs in FLOAT32 — lam = x * vals is exact there, the proposal
// k = floor(lam) + floor((2a/us + b) U + frac(lam) + 0.43) is formed without ever adding a small number to a large one, and
// the acceptance test compares V invalpha / (a / us^2 + b) with the Poisson pmf written without cancellation:
//   log pmf(k; lam) = -lam h(d / lam) - log sqrt(2 pi k) - 1/(12 k) + 1/(360 k^3),  d = k - lam,
//   h(x) = (1 + x) log(1 + x) - x = x^2 (1/2 - x/6 + x^2/12 - ...)          (Stirling; exact table of log k! below k = 8)
// so float32's hardware sqrt / rcp / exp / log (1 ulp) serve where float64 needed ~40-instruction software sequences:
// the float64 sampler ran at 0.19 TB/s.  The setup and the proposal use correctly rounded operations only (__fsqrt_rn,
// __fdiv_rn, no contraction), so the CPU oracle reproduces k bit for bit except where an acceptance test lands within
// ~1e-6 of equality.  lam >= 2^24 keeps the float64 form.
struct UniqueP {
    const float* in;
    unsigned int* count;      // [n_images] number of unique values
    float* vals_out;          // optional [n_images]: 2^ceil(log2(count))
    long long per;
    int parts_log2;           // hash parts of the general path
};

constexpr int kSeen = 256;
constexpr unsigned int kIntegerImage = 0x80000000u;          // flag in count[]: every element an integer in 0..255
constexpr unsigned int kHashSlots = 32768;
constexpr unsigned int kEmpty = 0xFFFFFFFFu;

__device__ __forceinline__ unsigned int float_key(float x) {
    if (x == 0.f) x = 0.f;                                    // -0.0 == 0.0 for np.unique
    unsigned int key = __float_as_uint(x);
    if (key == kEmpty) key = 0x7FC00000u;                     // a NaN pattern that collides with "empty"
    return key;
}

__global__ __launch_bounds__(1024) void unique_count_kernel(const UniqueP p) {
    extern __shared__ unsigned int lds[];                     // [kSeen] marks, [4] scalars, [kHashSlots] table
    unsigned int* seen = lds;
    unsigned int* scal = lds + kSeen;                         // 0: not-all-integers flag, 1: count, 2: probe overflow
    unsigned int* tab = lds + kSeen + 4;
    const int img = blockIdx.x, tid = threadIdx.x;
    const float* src = p.in + (size_t)img * p.per;
    if (tid < kSeen) seen[tid] = 0;
    if (tid < 4) scal[tid] = 0;
    __syncthreads();
    // ---- integer path: one streaming read, plain LDS stores
    bool other = false;
    const long long per4 = ((((uintptr_t)src) & 15) == 0) ? p.per / 4 : 0;
    for (long long i = tid; i < per4; i += 1024) {
        const f32x4 v = *(const f32x4*)(src + 4 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = v[j];
            const int q = (int)x;
            if (x >= 0.f && x <= 255.f && (float)q == x) seen[q] = 1; else other = true;
        }
    }
    for (long long i = 4 * per4 + tid; i < p.per; i += 1024) {
        const float x = src[i];
        const int q = (int)x;
        if (x >= 0.f && x <= 255.f && (float)q == x) seen[q] = 1; else other = true;
    }
    if (other) scal[0] = 1;
    __syncthreads();
    if (scal[0] == 0) {
        if (tid < kSeen && seen[tid]) atomicAdd(&scal[1], 1u);
        __syncthreads();
    } else {
        // ---- general path: exact count of distinct bit patterns, one hash part at a time
        const int parts = 1 << p.parts_log2;
        unsigned int mine = 0;
        for (int part = 0; part < parts; ++part) {
            for (unsigned int s = tid; s < kHashSlots; s += 1024) tab[s] = kEmpty;
            __syncthreads();
            for (long long i = tid; i < p.per; i += 1024) {
                const unsigned int key = float_key(src[i]);
                const unsigned int h = key * 2654435761u;
                if ((int)(h >> (32 - p.parts_log2 - 1) >> 1) != part && p.parts_log2) continue;      // top parts_log2 bits
                // the slot comes from the 15 bits just BELOW the part bits: the high bits of a multiplicative hash depend on every
                // key bit.  (Until round 6: (h >> 2) & 32767 — bits 2..16 of the product, which only key bits 0..16 reach: float
                // patterns with short mantissas — quarter steps, bf16 / f16-rounded pixels — all landed in a handful of home
                // slots and linear probing degenerated into thousands of contended atomicCAS per element)
                unsigned int slot = (h >> (32 - p.parts_log2 - 15)) & (kHashSlots - 1);
                for (unsigned int probes = 0;; ++probes) {
                    const unsigned int old = atomicCAS(&tab[slot], kEmpty, key);
                    if (old == kEmpty) { ++mine; break; }
                    if (old == key) break;
                    if (probes >= kHashSlots) { scal[2] = 1; break; }     // a part that does not fit: never hang, see below
                    slot = (slot + 1) & (kHashSlots - 1);
                }
            }
            __syncthreads();
        }
        if (scal[2]) {
            // a hash part overflowed the table (an adversarial set of bit patterns: never seen) — count by brute force
            // instead: an element counts if no earlier element equals it.  Slow (per^2 / 1024 compares per lane) but exact.
            mine = 0;
            for (long long i = tid; i < p.per; i += 1024) {
                const unsigned int key = float_key(src[i]);
                bool first = true;
                for (long long j = 0; j < i && first; ++j) first = float_key(src[j]) != key;
                mine += first ? 1u : 0u;
            }
        }
        if (mine) atomicAdd(&scal[1], mine);
        __syncthreads();
    }
    if (tid == 0) {
        const unsigned int c = scal[1];
        p.count[img] = c | (scal[0] == 0 ? kIntegerImage : 0u);
        if (p.vals_out) {
            unsigned int v = 1;                              // vals = 2 ** np.ceil(np.log2(n_unique))
            while (v < c) v <<= 1;
            p.vals_out[img] = (float)v;
        }
    }
}

__device__ __forceinline__ double loggam(double x) { return lgamma(x); }

struct PoissonP {
    const float* in;
    float* out;
    long long per;
    unsigned long long seed, first_image;
};

// float64 form (lam < 10: product of uniforms; lam >= 2^24: PTRS)
__device__ double poisson_sample(double lam, unsigned long long seed, unsigned long long elem) {
    if (!(lam >= 0.0)) return __longlong_as_double(0x7FF8000000000000ll);     // numpy raises for lam < 0
    if (lam == 0.0) return 0.0;
    unsigned int sub = 0;
    if (lam < 10.0) {
        // product of uniforms until it falls below exp(-lam)
        const double enlam = exp(-lam);
        double prod = 1.0;
        long long k = 0;
        for (;;) {
            const U4 r = draw(seed, elem, sub++, ST_POISSON);
            prod *= u01d(r.x, r.y);
            if (prod <= enlam) return (double)k;
            ++k;
            prod *= u01d(r.z, r.w);
            if (prod <= enlam) return (double)k;
            ++k;
        }
    }
    const double slam = sqrt(lam);
    const double b = 0.931 + 2.53 * slam;
    const double a = -0.059 + 0.02483 * b;
    const double vr = 0.9277 - 3.6224 / (b - 2.0);
    for (;;) {
        const U4 r = draw(seed, elem, sub++, ST_POISSON);
        const double U = u01d(r.x, r.y) - 0.5;
        const double V = u01d(r.z, r.w);
        const double us = 0.5 - fabs(U);
        const double k = floor((2.0 * a / us + b) * U + lam + 0.43);
        if (us >= 0.07 && V <= vr) return k;
        if (k < 0.0 || (us < 0.013 && V > us)) continue;
        // the squeeze above accepts ~9 draws in 10; the logarithms are only needed here
        const double invalpha = 1.1239 + 1.1328 / (b - 3.4);
        if (log(V) + log(invalpha) - log(a / (us * us) + b) <= -lam + k * log(lam) - loggam(k + 1.0)) return k;
    }
}

// log(k!) for k < 8 (float32)
__device__ __forceinline__ float small_logfact(int k) {
    const float t[8] = {0.f, 0.f, 0.6931471806f, 1.791759469f, 3.178053830f, 4.787491743f, 6.579251212f, 8.525161361f};
    float v = t[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) v = (k == i) ? t[i] : v;
    return v;
}

// float32 PTRS for 10 <= lam < 2^24 (header above), in two pieces: the per-lambda constants and one attempt.
// Setup and proposal are correctly rounded single operations in this order (the oracle repeats them in NumPy float32).
struct PtrsC { float b, a, a2, vr, invalpha, inv_lam; };

__device__ __forceinline__ PtrsC ptrs_setup(float lam) {
    PtrsC c;
    const float slam = __fsqrt_rn(lam);
    c.b = __fadd_rn(0.931f, __fmul_rn(2.53f, slam));
    c.a = __fadd_rn(-0.059f, __fmul_rn(0.02483f, c.b));
    c.invalpha = __fadd_rn(1.1239f, __fdiv_rn(1.1328f, __fsub_rn(c.b, 3.4f)));
    c.vr = __fsub_rn(0.9277f, __fdiv_rn(3.6224f, __fsub_rn(c.b, 2.0f)));
    c.a2 = __fmul_rn(2.0f, c.a);
    c.inv_lam = 1.0f / lam;
    return c;
}

// one attempt with the uniforms of words (wu, wv): true = accepted, kf = k - floor(lam)
__device__ __forceinline__ bool ptrs_attempt(const PtrsC& c, float lam, float Li, float Lf043, unsigned int wu, unsigned int wv, float& kf) {
    const float U = __fsub_rn(u01(wu), 0.5f);
    const float V = u01(wv);
    const float us = __fsub_rn(0.5f, fabsf(U));
    kf = floorf(__fadd_rn(__fmul_rn(__fadd_rn(__fdiv_rn(c.a2, us), c.b), U), Lf043));
    if (us >= 0.07f && V <= c.vr) return true;
    const float k = Li + kf;                                   // the acceptance test may round it
    if (k < 0.f || (us < 0.013f && V > us)) return false;
    // V invalpha / (a / us^2 + b) <= pmf(k; lam), pmf without cancellation (header)
    const float us2 = us * us;
    const float lhs = V * c.invalpha * us2 / (c.a + c.b * us2);
    float logp;
    if (k < 8.f) {
        logp = -lam + k * __logf(lam) - small_logfact((int)k);
    } else {
        const float d = kf - (lam - Li);                       // k - lam
        const float x = d * c.inv_lam;
        float h;                                               // (1 + x) log(1 + x) - x
        if (fabsf(x) < 0.125f) {
            float s = -1.f / 110.f;                            // x^2 (1/2 - x/6 + x^2/12 - ... + x^8/90 - x^9/110)
            s = fmaf(s, x, 1.f / 90.f);
            s = fmaf(s, x, -1.f / 72.f);
            s = fmaf(s, x, 1.f / 56.f);
            s = fmaf(s, x, -1.f / 42.f);
            s = fmaf(s, x, 1.f / 30.f);
            s = fmaf(s, x, -1.f / 20.f);
            s = fmaf(s, x, 1.f / 12.f);
            s = fmaf(s, x, -1.f / 6.f);
            s = fmaf(s, x, 0.5f);
            h = s * x * x;
        } else {
            h = (1.f + x) * log1pf(x) - x;
        }
        const float rk = 1.0f / k;
        logp = -lam * h - 0.5f * __logf(6.283185307179586f * k) - rk * (1.f / 12.f) + rk * rk * rk * (1.f / 360.f);
    }
    return __logf(lhs) <= logp;
}

// element by element (poisson_rest_kernel, and the definition the oracle follows): Philox block (elem, sub) serves two
// attempts, (x, y) then (z, w).  Returns the count as a double: Li + kf can exceed 2^24 by a hair.
__device__ __forceinline__ double poisson_ptrs_f32(float lam, unsigned long long seed, unsigned long long elem) {
    const PtrsC c = ptrs_setup(lam);
    const float Li = floorf(lam);
    const float Lf043 = __fadd_rn(__fsub_rn(lam, Li), 0.43f);
    float kf;
    for (unsigned int sub = 0;; ++sub) {
        const U4 r = draw(seed, elem, sub, ST_POISSON);
        if (ptrs_attempt(c, lam, Li, Lf043, r.x, r.y, kf)) break;
        if (ptrs_attempt(c, lam, Li, Lf043, r.z, r.w, kf)) break;
    }
    return (double)Li + (double)kf;
}

// Pass 2.  A wave owns kWaveRange consecutive elements of an image and hands them to its lanes ONE AT A TIME: a lane whose
// element was accepted takes the next unclaimed one in the same round (ballot + prefix count), so that every round nearly
// all 64 lanes make a real attempt.  (One 16-wave workgroup per image drawing from a counter in LDS — fewer drains — was
// 13 % slower: profiles/experiments/r05_poisson_kernel.txt.)  (A lane-per-element loop runs until the slowest of its 64 lanes accepts: 12 % of the
// attempts fail, so almost every wave paid a second and a third round for a handful of lanes — 3.2 attempts and 1.6 Philox
// blocks per element-slot against 1.14 here.)  The per-element stream is untouched: element e draws Philox blocks
// (e, 0), (e, 1), ... and uses (x, y) then (z, w) of each.  For an integer-valued image (unique_count_kernel marks it) lam
// takes at most 256 values: their constants — a square root and three divisions each — come from a table in LDS built
// once per workgroup.  Elements outside the float32 sampler's range (0 < lam < 10, lam >= 2^24, negative, NaN) are left to
// poisson_rest_kernel through a bitmap; a negative one also raises the flag the host turns into numpy's ValueError.
constexpr int kWaveRange = 1536;      // sweep (1,024 images of 112 x 112 x 3): 512..2048 within 3 %, 3072+ 10-15 % slower

struct PoissonScratch {
    unsigned int* count;       // [n] unique values | kIntegerImage
    unsigned int* any_rest;    // [n] image has elements for poisson_rest_kernel
    unsigned int* negative;    // [1] some element was negative (np.random.poisson raises)
    unsigned int* bitmap;      // [n][words_per_image]
};

__global__ __launch_bounds__(256) void poisson_kernel(const PoissonP p, const PoissonScratch sc, int words_per_image, long long range) {
    __shared__ float tab[256 * 6];
    const int img = blockIdx.y, tid = threadIdx.x;
    const unsigned int craw = sc.count[img];
    const bool int_image = (craw & kIntegerImage) != 0;
    unsigned int c = craw & ~kIntegerImage, v = 1;            // vals = 2 ** np.ceil(np.log2(n_unique))
    while (v < c) v <<= 1;
    const float valsf = (float)v;
    const double inv_vals = 1.0 / (double)v;                  // a power of two: k * inv_vals == k / vals exactly
    if (int_image) {
        const float lam = (float)tid * valsf;
        if (lam >= 10.f && lam < 16777216.f) {
            const PtrsC t = ptrs_setup(lam);
            float* q = tab + tid * 6;
            q[0] = t.b; q[1] = t.a; q[2] = t.a2; q[3] = t.vr; q[4] = t.invalpha; q[5] = t.inv_lam;
        }
    }
    __syncthreads();
    const long long start = ((long long)blockIdx.x * 4 + (tid >> 6)) * range;         // `range` elements per WAVE
    if (start >= p.per) return;
    const long long end = start + range < p.per ? start + range : p.per;
    const float* src = p.in + (size_t)img * p.per;
    float* dst = p.out + (size_t)img * p.per;
    const unsigned long long elem0 = (p.first_image + (unsigned long long)img) * (unsigned long long)p.per;
    long long next = start;                                   // wave-uniform
    bool active = false;
    long long e = 0;
    float lam = 0.f, Li = 0.f, Lf043 = 0.f, kf = 0.f;
    PtrsC k{};
    unsigned int sub = 0, rz = 0, rw = 0;
    bool second = false;
    for (;;) {
        // ---- lanes without an element claim the next ones
        const unsigned long long want = next < end ? __ballot(!active) : 0ull;
        if (want) {
            const int pre = __builtin_amdgcn_mbcnt_hi((unsigned int)(want >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)want, 0u));
            const long long mine = next + pre;
            if (!active && mine < end) {
                e = mine;
                const float x = src[e];
                lam = x * valsf;                              // exact (a power of two) unless it overflows
                if (lam == 0.f) {
                    dst[e] = 0.f;
                } else if (lam >= 10.f && lam < 16777216.f) {
                    Li = floorf(lam);
                    Lf043 = __fadd_rn(__fsub_rn(lam, Li), 0.43f);
                    if (int_image) {
                        const float* q = tab + (int)x * 6;
                        k.b = q[0]; k.a = q[1]; k.a2 = q[2]; k.vr = q[3]; k.invalpha = q[4]; k.inv_lam = q[5];
                    } else {
                        k = ptrs_setup(lam);
                    }
                    sub = 0;
                    second = false;
                    active = true;
                } else {
                    atomicOr(&sc.bitmap[(size_t)img * words_per_image + (e >> 5)], 1u << (e & 31));
                    sc.any_rest[img] = 1;
                    if (x < 0.f) sc.negative[0] = 1;
                }
            }
            next += __popcll(want);
        }
        if (!__any(active)) {
            if (next >= end) break;
            continue;
        }
        // ---- one attempt per lane that holds an element
        if (active) {
            unsigned int wu, wv;
            if (!second) {
                const U4 r = draw(p.seed, elem0 + (unsigned long long)e, sub, ST_POISSON);
                wu = r.x; wv = r.y; rz = r.z; rw = r.w;
            } else {
                wu = rz; wv = rw;
                ++sub;
            }
            second = !second;
            if (ptrs_attempt(k, lam, Li, Lf043, wu, wv, kf)) {
                dst[e] = (float)(((double)Li + (double)kf) * inv_vals);
                active = false;
            }
        }
    }
}

// what poisson_kernel left in the bitmap: the float64 samplers (product of uniforms below 10, PTRS from 2^24, NaN for a
// negative rate).  grid (slices, n_images); an image without such elements costs one load.
__global__ __launch_bounds__(256) void poisson_rest_kernel(const PoissonP p, const PoissonScratch sc, int words_per_image) {
    const int img = blockIdx.y;
    if (!sc.any_rest[img]) return;
    unsigned int c = sc.count[img] & ~kIntegerImage, v = 1;
    while (v < c) v <<= 1;
    const double vals = (double)v;
    const unsigned int* bm = sc.bitmap + (size_t)img * words_per_image;
    const unsigned long long elem0 = (p.first_image + (unsigned long long)img) * (unsigned long long)p.per;
    for (int w = blockIdx.x * 256 + threadIdx.x; w < words_per_image; w += gridDim.x * 256) {
        unsigned int bits = bm[w];
        while (bits) {
            const int bit = __ffs(bits) - 1;
            bits &= bits - 1;
            const long long e = (long long)w * 32 + bit;
            const float x = p.in[(size_t)img * p.per + e];
            const double k = poisson_sample((double)x * vals, p.seed, elem0 + (unsigned long long)e);
            p.out[(size_t)img * p.per + e] = (float)(k / vals);
        }
    }
}

// ---- bilinear resize (cv2.resize INTER_LINEAR, code/committee.py:22-26) -------------------------------
// OpenCV's pixel-centre mapping: f = (d + 0.5) * (src / dst) - 0.5, s = floor(f), f -= s, clamped so
// that s in [0, src - 1] with weight 0 on the clamped side; horizontal pass then vertical pass.
struct ResizeP {
    const float* in;
    float* out;
    int N, H, W, C, Ho, Wo;
};

__device__ __forceinline__ void src_coord(int d, int src, int dst, int& s0, int& s1, float& w) {
    float f = (float)(((double)d + 0.5) * ((double)src / (double)dst) - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= src - 1) { s = src - 1; f = 0.f; }
    s0 = s;
    s1 = s + 1 < src ? s + 1 : src - 1;
    w = f;
}

__global__ __launch_bounds__(256) void resize_kernel(const ResizeP p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)p.N * p.Ho * p.Wo * p.C;
    if (i >= total) return;
    const int c = (int)(i % p.C);
    long long t = i / p.C;
    const int ox = (int)(t % p.Wo); t /= p.Wo;
    const int oy = (int)(t % p.Ho);
    const int n = (int)(t / p.Ho);
    int x0, x1, y0, y1;
    float fx, fy;
    src_coord(ox, p.W, p.Wo, x0, x1, fx);
    src_coord(oy, p.H, p.Ho, y0, y1, fy);
    const float* im = p.in + (size_t)n * p.H * p.W * p.C;
    const float a00 = im[((size_t)y0 * p.W + x0) * p.C + c], a01 = im[((size_t)y0 * p.W + x1) * p.C + c];
    const float a10 = im[((size_t)y1 * p.W + x0) * p.C + c], a11 = im[((size_t)y1 * p.W + x1) * p.C + c];
    const float r0 = a00 * (1.f - fx) + a01 * fx;
    const float r1 = a10 * (1.f - fx) + a11 * fx;
    p.out[i] = r0 * (1.f - fy) + r1 * fy;
}

// ---- perturb_image (code/attack.py:5-29) -----------------------------------------------------------------
// n candidates, each k x (x, y, r, g, b): copy the base image, then img[int(x), int(y)] = (r, g, b)
// in list order (a later pixel overwrites an earlier one at the same position).  With split != 0
// the output is laid out [2][n][Hc/2][W][3] — all top halves, then all bottom halves — which is what
// noise.PredictionWrappedModel.predict (code/noise.py:158-168) slices before embedding.
struct PerturbP {
    const float* img;      // [n_img][Hc][W][3]
    const int* img_of;     // image of candidates [g * group, (g + 1) * group): img_of[g]; nullptr: image 0 for all
    const double* xs;      // [n][5k]
    float* out;
    int n, k, Hc, W, split, group;
};

__global__ __launch_bounds__(256) void perturb_kernel(const PerturbP p) {
    const int cand = blockIdx.x, tid = threadIdx.x;
    const long long per = (long long)p.Hc * p.W * 3, half = per / 2;
    const float* img = p.img + (p.img_of ? (size_t)p.img_of[cand / p.group] * per : 0);
    float* top = p.split ? p.out + (size_t)cand * half : p.out + (size_t)cand * per;
    float* bot = p.split ? p.out + ((size_t)p.n + cand) * half : top + half;
    if ((half & 3) == 0 && ((((uintptr_t)img) | ((uintptr_t)p.out)) & 15) == 0) {      // 16-byte lanes (112-wide pairs: always)
        const float4* s0 = (const float4*)img;
        const float4* s1 = (const float4*)(img + half);
        float4* d0 = (float4*)top;
        float4* d1 = (float4*)bot;
        for (long long i = tid; i < half / 4; i += 256) { d0[i] = s0[i]; d1[i] = s1[i]; }
    } else {
        for (long long i = tid; i < half; i += 256) { top[i] = img[i]; bot[i] = img[half + i]; }
        if ((per & 1) && tid == 0) bot[half] = img[per - 1];      // odd Hc never splits (host refuses)
    }
    __syncthreads();
    if (tid == 0) {
        const double* x = p.xs + (size_t)cand * 5 * p.k;
        for (int j = 0; j < p.k; ++j) {
            const long long r = (long long)x[5 * j], c = (long long)x[5 * j + 1];     // astype(int): truncation
            if (r < 0 || r >= p.Hc || c < 0 || c >= p.W) continue;
            const long long o = (r * p.W + c) * 3;
            float* d = o < half ? top + o : bot + (o - half);
            d[0] = (float)(long long)x[5 * j + 2];
            d[1] = (float)(long long)x[5 * j + 3];
            d[2] = (float)(long long)x[5 * j + 4];
        }
    }
}

inline dim3 g1(long long n) { return dim3((unsigned)((n + 255) / 256), 1, 1); }
inline int affine_vec_ok(const float* in, const float* out, unsigned long long offset) {
    return ((offset & 3) == 0 && ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0) ? 1 : 0;
}

}  // namespace
}  // namespace alink

using namespace alink;

extern "C" {

int alink_noise_gaussian(const float* dev_in, float* dev_out, int64_t count, float mean, float sigma,
                         uint64_t seed, uint64_t offset, void* stream) {
    ALINK_REQUIRE(dev_in && dev_out && count >= 0, ALINK_EINVAL, "bad argument");
    DeviceGuard dg(device_of_pointer(dev_out));
    if (count == 0) return ALINK_OK;
    AffineNoise p{dev_in, dev_out, count, seed, offset, mean, sigma, 0, affine_vec_ok(dev_in, dev_out, offset)};
    hipLaunchKernelGGL(affine_noise_kernel, g1(((long long)(offset & 3) + count + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

int alink_noise_speckle(const float* dev_in, float* dev_out, int64_t count, float divisor, uint64_t seed,
                        uint64_t offset, void* stream) {
    ALINK_REQUIRE(dev_in && dev_out && count >= 0 && divisor != 0.f, ALINK_EINVAL, "bad argument");
    DeviceGuard dg(device_of_pointer(dev_out));
    if (count == 0) return ALINK_OK;
    AffineNoise p{dev_in, dev_out, count, seed, offset, divisor, 0.f, 1, affine_vec_ok(dev_in, dev_out, offset)};
    hipLaunchKernelGGL(affine_noise_kernel, g1(((long long)(offset & 3) + count + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

int alink_noise_uniform(const float* dev_in, float* dev_out, int64_t count, float lo, float hi, uint64_t seed,
                        uint64_t offset, void* stream) {
    ALINK_REQUIRE(dev_in && dev_out && count >= 0 && lo <= hi, ALINK_EINVAL, "bad argument");
    DeviceGuard dg(device_of_pointer(dev_out));
    if (count == 0) return ALINK_OK;
    AffineNoise p{dev_in, dev_out, count, seed, offset, lo, hi, 2, affine_vec_ok(dev_in, dev_out, offset)};
    hipLaunchKernelGGL(affine_noise_kernel, g1(((long long)(offset & 3) + count + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

int alink_keep_masks(uint8_t* dev_out, int64_t count, float keep, uint64_t seed, void* stream) {
    return alink_keep_masks_at(dev_out, count, keep, seed, 0, stream);
}

int alink_keep_masks_at(uint8_t* dev_out, int64_t count, float keep, uint64_t seed, uint64_t first, void* stream) {
    ALINK_REQUIRE(dev_out && count >= 0 && keep >= 0.f && keep <= 1.f, ALINK_EINVAL, "bad argument");
    DeviceGuard dg(device_of_pointer(dev_out));
    if (count == 0) return ALINK_OK;
    hipLaunchKernelGGL(keep_mask_kernel, g1(((long long)(first & 3) + count + 3) / 4), dim3(256), 0, (hipStream_t)stream, dev_out,
                       (long long)count, keep, seed, (unsigned long long)first);
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

int alink_noise_saltpepper(const float* dev_in, float* dev_out, int n_images, int H, int W, int C, int n_salt,
                           int n_pepper, uint64_t seed, uint64_t first_image, void* stream) {
    ALINK_REQUIRE(dev_in && dev_out && n_images >= 0, ALINK_EINVAL, "bad argument");
    DeviceGuard dg(device_of_pointer(dev_out));
    // np.random.randint(0, i - 1) needs i - 1 > 0 for every axis (code/noise.py:59,63)
    ALINK_REQUIRE(H >= 2 && W >= 2 && C >= 2, ALINK_EINVAL, "low >= high: image %dx%dx%d has an axis shorter than 2", H, W, C);
    ALINK_REQUIRE(n_salt >= 0 && n_pepper >= 0, ALINK_EINVAL, "negative counts");
    if (n_images == 0) return ALINK_OK;
    ALINK_REQUIRE(first_image + (uint64_t)n_images <= (1ull << 32), ALINK_EINVAL, "image index beyond 2^32");
    SaltPepperP p{dev_in, dev_out, H, W, C, n_salt, n_pepper, seed, first_image};
    hipLaunchKernelGGL(saltpepper_kernel, dim3(n_images), dim3(256), 0, (hipStream_t)stream, p);
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

/* number of grid nodes per image for the three octaves: sum (size/ns + 1)^2 */
int alink_perlin_nodes(int size, const int* ns3) {
    if (!ns3 || size <= 0) return -1;
    int t = 0;
    for (int o = 0; o < 3; ++o) {
        if (ns3[o] <= 0) return -1;
        const int gs = size / ns3[o] + 1;
        t += gs * gs;
    }
    return t;
}

int alink_perlin_vectors(int n_images, int nodes_total, uint64_t seed, uint64_t first_image, float* dev_vec, void* stream) {
    ALINK_REQUIRE(dev_vec && n_images >= 0 && nodes_total > 0, ALINK_EINVAL, "bad argument");
    DeviceGuard dg(device_of_pointer(dev_vec));
    const long long n = (long long)n_images * nodes_total;
    if (n == 0) return ALINK_OK;
    const unsigned long long first = first_image * (unsigned long long)nodes_total;
    hipLaunchKernelGGL(perlin_vectors_kernel, g1(((long long)(first & 3) + n + 3) / 4), dim3(256), 0, (hipStream_t)stream, dev_vec, n, seed, first);
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

int alink_noise_perlin(const float* dev_in, float* dev_out, int n_images, int size, int C, const int* ns3,
                       const float* dev_vec, void* stream) {
    ALINK_REQUIRE(dev_in && dev_out && dev_vec && ns3 && n_images >= 0 && size > 0 && C > 0, ALINK_EINVAL, "bad argument");
    DeviceGuard dg(device_of_pointer(dev_out));
    PerlinP p{};
    p.in = dev_in; p.out = dev_out; p.vec = dev_vec; p.size = size; p.C = C;
    int off = 0;
    for (int o = 0; o < 3; ++o) {
        // m.reshape(nc, ns, nc, ns) (code/noise.py:130) needs size == int(size/ns) * ns
        ALINK_REQUIRE(ns3[o] > 0 && (size / ns3[o]) * ns3[o] == size, ALINK_EINVAL,
                      "cannot reshape array of size %d into shape (%d,%d,%d,%d)", size * size, size / ns3[o], ns3[o],
                      size / ns3[o], ns3[o]);
        p.ns[o] = ns3[o];
        p.goff[o] = off;
        const int gs = size / ns3[o] + 1;
        off += gs * gs;
    }
    p.nodes_total = off;
    if (n_images == 0) return ALINK_OK;
    for (int i0 = 0; i0 < n_images; i0 += 32768) {            // grid.y carries the image index
        const int m = n_images - i0 < 32768 ? n_images - i0 : 32768;
        PerlinP q = p;
        q.in = dev_in + (size_t)i0 * size * size * C;
        q.out = dev_out + (size_t)i0 * size * size * C;
        q.vec = dev_vec + (size_t)i0 * p.nodes_total * 2;
        hipLaunchKernelGGL(perlin_kernel, dim3((size * size + 255) / 256, m), dim3(256), 0, (hipStream_t)stream, q);
    }
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

static inline size_t poisson_words_per_image(int64_t per_image) { return (size_t)((per_image + 31) / 32); }

size_t alink_noise_poisson_scratch_bytes(int n_images, int64_t per_image) {
    if (n_images <= 0 || per_image <= 0) return 0;
    // per image: the unique-value count, the "has elements for the float64 sampler" word, one bit per element; + the
    // negative-rate flag (first word of the scratch: the host reads it to raise numpy's ValueError)
    return 16 + (size_t)n_images * 8 + (size_t)n_images * poisson_words_per_image(per_image) * 4;
}

int alink_noise_poisson(const float* dev_in, float* dev_out, int n_images, int64_t per_image, uint64_t seed,
                        uint64_t first_image, void* dev_scratch, size_t scratch_bytes, float* dev_vals, void* stream) {
    ALINK_REQUIRE(dev_in && dev_out && n_images >= 0 && per_image > 0, ALINK_EINVAL, "bad argument");
    DeviceGuard dg(device_of_pointer(dev_out));
    ALINK_REQUIRE(per_image < (1ll << 30), ALINK_EINVAL, "image of %lld elements too large", (long long)per_image);
    if (n_images == 0) return ALINK_OK;
    const size_t need = alink_noise_poisson_scratch_bytes(n_images, per_image);
    ALINK_REQUIRE(dev_scratch && scratch_bytes >= need && (((uintptr_t)dev_scratch) & 3) == 0, ALINK_ENOMEM, "scratch %zu < %zu bytes (or unaligned)", scratch_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    const int wpi = (int)poisson_words_per_image(per_image);
    PoissonScratch sc;
    sc.negative = (unsigned int*)dev_scratch;
    sc.count = sc.negative + 4;
    sc.any_rest = sc.count + n_images;
    sc.bitmap = sc.any_rest + n_images;
    ALINK_HIP(hipMemsetAsync(dev_scratch, 0, need, st));
    // hash parts of the general path: a part's expected load of the 32768-slot table stays at or below 1/2
    int parts_log2 = 0;
    while ((16384ll << parts_log2) < per_image) ++parts_log2;
    UniqueP u{dev_in, sc.count, dev_vals, per_image, parts_log2};
    const size_t lds = (kSeen + 4 + kHashSlots) * sizeof(unsigned int);
    static bool attr_set[64] = {};
    const int dev = device_of_pointer(dev_out);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        ALINK_HIP(hipFuncSetAttribute((const void*)unique_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev] = true;
    }
    // grid.y carries the image index in the sampling kernels: at most 32768 images per launch (pool-scale batches take several)
    const unsigned int nblk = (unsigned int)((per_image + 4 * kWaveRange - 1) / (4 * kWaveRange));
    const unsigned int slices = (unsigned int)((wpi + 255) / 256);
    for (int i0 = 0; i0 < n_images; i0 += 32768) {
        const int m = n_images - i0 < 32768 ? n_images - i0 : 32768;
        UniqueP uc = u;
        uc.in = dev_in + (size_t)i0 * per_image;
        uc.count = sc.count + i0;
        uc.vals_out = dev_vals ? dev_vals + i0 : nullptr;
        hipLaunchKernelGGL(unique_count_kernel, dim3(m), dim3(1024), lds, st, uc);
        PoissonP p{dev_in + (size_t)i0 * per_image, dev_out + (size_t)i0 * per_image, per_image, seed, first_image + (uint64_t)i0};
        PoissonScratch sci = sc;
        sci.count = sc.count + i0;
        sci.any_rest = sc.any_rest + i0;
        sci.bitmap = sc.bitmap + (size_t)i0 * wpi;
        hipLaunchKernelGGL(poisson_kernel, dim3(nblk, m), dim3(256), 0, st, p, sci, wpi, (long long)kWaveRange);
        hipLaunchKernelGGL(poisson_rest_kernel, dim3(slices < 8 ? slices : 8, m), dim3(256), 0, st, p, sci, wpi);
    }
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

// ---- signed-gradient step + projection (FGSM / PGD extension, a-link_amd/noise.py) -------------------------------------
// adv <- clip(clip(adv + step * sign(grad), clean - eps, clean + eps), lo, hi), in place; 16 B per lane, HBM-bound
// (three float32 streams in, one out)
namespace alink {
namespace {
__global__ __launch_bounds__(256) void pgd_step_kernel(float* __restrict__ adv, const float* __restrict__ clean,
                                                       const float* __restrict__ grad, long long n4, long long n,
                                                       float step, float eps, float lo, float hi) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    auto one = [&](float a, float c, float g) {
        const float s = (g > 0.f ? 1.f : 0.f) - (g < 0.f ? 1.f : 0.f);
        a = fmaf(step, s, a);
        a = fminf(fmaxf(a, c - eps), c + eps);
        return fminf(fmaxf(a, lo), hi);
    };
    if (i < n4) {
        float4 a = ((const float4*)adv)[i];
        const float4 c = ((const float4*)clean)[i], g = ((const float4*)grad)[i];
        a.x = one(a.x, c.x, g.x); a.y = one(a.y, c.y, g.y); a.z = one(a.z, c.z, g.z); a.w = one(a.w, c.w, g.w);
        ((float4*)adv)[i] = a;
    } else if (i == n4) {
        for (long long k = 4 * n4; k < n; ++k) adv[k] = one(adv[k], clean[k], grad[k]);
    }
}
__global__ __launch_bounds__(256) void pgd_step_scalar_kernel(float* __restrict__ adv, const float* __restrict__ clean,
                                                              const float* __restrict__ grad, long long n,
                                                              float step, float eps, float lo, float hi) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g = grad[i], c = clean[i];
    const float s = (g > 0.f ? 1.f : 0.f) - (g < 0.f ? 1.f : 0.f);
    float a = fmaf(step, s, adv[i]);
    a = fminf(fmaxf(a, c - eps), c + eps);
    adv[i] = fminf(fmaxf(a, lo), hi);
}
}  // namespace
}  // namespace alink

int alink_pgd_step(float* dev_adv, const float* dev_clean, const float* dev_grad, int64_t n, float step, float eps,
                   float lo, float hi, void* stream) {
    ALINK_REQUIRE(dev_adv && dev_clean && dev_grad && n >= 0, ALINK_EINVAL, "bad argument");
    if (n == 0) return ALINK_OK;
    DeviceGuard dg(device_of_pointer(dev_adv));
    // 16-byte lane accesses when all three buffers are 16-byte aligned; a chunk of an odd-sized batch is not (its start
    // is s * H * W * C * 4 bytes): the same kernel then takes every element in its scalar tail — one block per 256 elements
    const bool aligned = (((uintptr_t)dev_adv | (uintptr_t)dev_clean | (uintptr_t)dev_grad) & 15) == 0;
    if (!aligned) {
        hipLaunchKernelGGL(alink::pgd_step_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           dev_adv, dev_clean, dev_grad, (long long)n, step, eps, lo, hi);
        ALINK_HIP(hipGetLastError());
        return ALINK_OK;
    }
    const long long n4 = n / 4;
    hipLaunchKernelGGL(alink::pgd_step_kernel, dim3((unsigned)((n4 + 1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       dev_adv, dev_clean, dev_grad, n4, (long long)n, step, eps, lo, hi);
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

int alink_resize_bilinear(const float* dev_in, float* dev_out, int n, int H, int W, int C, int Ho, int Wo,
                          void* stream) {
    ALINK_REQUIRE(dev_in && dev_out && n >= 0 && H > 0 && W > 0 && C > 0 && Ho > 0 && Wo > 0, ALINK_EINVAL, "bad argument");
    DeviceGuard dg(device_of_pointer(dev_out));
    const long long total = (long long)n * Ho * Wo * C;
    if (total == 0) return ALINK_OK;
    ALINK_REQUIRE(total < (1ll << 39), ALINK_EINVAL, "output too large");
    ResizeP p{dev_in, dev_out, n, H, W, C, Ho, Wo};
    hipLaunchKernelGGL(resize_kernel, g1(total), dim3(256), 0, (hipStream_t)stream, p);
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

int alink_perturb_images(const float* dev_img, const double* dev_xs, int n, int k, int Hc, int W, int split,
                         float* dev_out, void* stream) {
    return alink_perturb_images_multi(dev_img, nullptr, 1, dev_xs, n, k, Hc, W, split, dev_out, stream);
}

int alink_perturb_images_multi(const float* dev_imgs, const int* dev_img_of, int group, const double* dev_xs, int n, int k,
                               int Hc, int W, int split, float* dev_out, void* stream) {
    ALINK_REQUIRE(dev_imgs && dev_xs && dev_out && n >= 0 && k >= 0 && Hc > 0 && W > 0 && group > 0, ALINK_EINVAL, "bad argument");
    DeviceGuard dg(device_of_pointer(dev_out));
    ALINK_REQUIRE(!split || (Hc % 2) == 0, ALINK_EINVAL, "split needs an even number of rows, got %d", Hc);
    if (n == 0) return ALINK_OK;
    PerturbP p{dev_imgs, dev_img_of, dev_xs, dev_out, n, k, Hc, W, split, group};
    hipLaunchKernelGGL(perturb_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, p);
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

}  // extern "C"

// head.hip — the siamese pair scorer and its fine-tune step.
//
// Replaces the Keras graph built at reference code/siamese.py:24-35
//     L1 = abs(left - right); Dense(h1, relu); Dense(h2, relu); Dense(2); softmax
//     compile(loss="binary_crossentropy", optimizer=Adadelta(lr), metrics=['accuracy'])
// and the calls made on it: predict (code/siamese.py:130-131), fit/train_on_batch/test_on_batch
// (code/siamese.py:57,103,107) and the committee mean (code/committee.py:13-20).
//
// f32 by default (bf16 compute mode: see `qmode` below).  The large-P forward runs on the f32-input matrix cores
// (v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered fmaf chain, so results match a CPU f32 matmul
// to rounding).  Pairs are gathered by index straight from the embedding matrices (row = 2 KB at
// D = 512, fully coalesced), so the N^2 score-matrix workload (utilities/generateMatrixDFW.py:25-36)
// never materialises pair tensors.  The fine-tune step works on Keras-sized batches (16 rows):
// latency-bound, so it is a short chain of plain VALU kernels, gradients kept in one flat buffer
// for the data-parallel all-reduce.
//
// Keras 2.1.2 semantics restated (SURVEY.md §8c "files a CPU restatement must follow"):
//   binary_crossentropy: p clipped to [1e-7, 1-1e-7], converted to a logit, sigmoid CE on it, mean
//   over the 2 outputs; sample-weighted batch loss = mean(w*l) / mean(w != 0); 'accuracy' resolves
//   to binary_accuracy = mean(round(p) == y) (round half to even), not weighted;
//   Adadelta: a = rho*a + (1-rho)*g^2; u = g*sqrt(d+eps)/sqrt(a+eps); p -= lr*u; d = rho*d+(1-rho)*u^2.
#include "alink_common.h"

#include <algorithm>
#include <vector>

using namespace alink;

struct alink_head {
    int device = -1;             // the device the handle's memory lives on (current at create)
    int D, h1, h2;
    int od = 2;                  // outputs: 2 = Dense(2)+softmax (code/siamese.py:31-32), 1 = Dense(1, sigmoid) (code/siamese3.py:25)
    float lr, rho, eps;
    size_t nparams;
    size_t oW1, ob1, oW2, ob2, oW3, ob3;
    float* d_params = nullptr;   // Keras layout: kernel (in,out) row-major, bias; per layer
    float* d_grads = nullptr;
    float* d_acc = nullptr;      // Adadelta accumulators
    float* d_dacc = nullptr;     // Adadelta delta accumulators
    float* d_w1p = nullptr;      // W1 packed for the MFMA forward: [D/8][h1][2][4]
    float* d_w2p = nullptr;      // W2 packed: [h1/8][h2][2][4]
    bool packed_dirty = true;
    // bf16 compute mode (alink_head_set_compute_dtype; BASELINE configs[4] "bf16 fine-tune", mixed precision):
    // master parameters, gradients and Adadelta state stay f32; the forward / backward GEMM operands are bf16 —
    // weights from d_pq (2-byte copy of the flat parameter vector, kept current by the update kernels of the
    // batch<=32 step, else re-quantised lazily), activations and activation gradients rounded to bf16 where they are
    // produced; products are exact in f32 and accumulate in f32.  d_pqf holds the same bf16 values widened to f32 for
    // the large-batch chain and the MFMA predict kernel (a derived cache like d_w1p / d_w2p).  Biases are not quantised.
    int qmode = 0;
    __bf16* d_pq = nullptr;
    float* d_pqf = nullptr;
    __bf16* d_wt = nullptr;      // bf16 mode, predict on the bf16 matrix cores: W1^T [h1][D] then W2^T [h2][h1] (re-derived with the packed copies)
    bool pq_dirty = true, pqf_dirty = true;
    // train/eval scratch for up to `cap` rows
    int cap = 4096;
    float *d_dm = nullptr, *d_z1 = nullptr, *d_z2 = nullptr, *d_dz1 = nullptr, *d_dz2 = nullptr,
          *d_dz3 = nullptr, *d_p = nullptr;
    float* d_tiny = nullptr;     // per-row-group partials of the tiny-batch train step
    float* d_mini = nullptr;     // Dense1 partial sums of SmallRes' three-launch step ([D / 32][n][128]); only for that head shape
    unsigned* d_counter = nullptr;   // row groups finished (tiny_eval), 0 between launches
    std::vector<void*> allocs;
    // hipGraph cache of the fine-tune step (a launch-bound chain of 8-9 small kernels): one executable
    // graph per distinct (operand pointers, n, grad_scale, apply); replayed while the caller keeps
    // feeding the same buffers (DenseHead stages every batch into persistent tensors for that purpose)
    struct StepGraph {
        const void *L, *R, *y, *sw, *metrics;
        int n, apply;
        float grad_scale, lr;
        hipGraphExec_t exec;
    };
    std::vector<StepGraph> graphs;
    bool use_graph = false;     // measured on MI355X: replay 67.8 us vs 66.2 us of plain launches — the chain is
                                // bound by kernel-to-kernel dependency latency, not by launch cost; kept as an option
    ~alink_head() {
        for (auto& g : graphs) (void)hipGraphExecDestroy(g.exec);
        for (void* p : allocs) (void)hipFree(p);
    }
};

namespace {

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
// round to bf16 and back (RNE): the value a bf16 store of x would hold
__device__ __forceinline__ float qbf(float x) { return (float)(__bf16)x; }
template <bool Q> __device__ __forceinline__ float qa(float x) { return Q ? qbf(x) : x; }
__device__ __forceinline__ float qa(float x, int q) { return q ? qbf(x) : x; }
// 4 consecutive weights starting at element `i` of a flat f32 (Q = false) or bf16 (Q = true) parameter vector
template <bool Q> __device__ __forceinline__ f32x4 ldw4(const void* base, size_t i) {
    if (Q) {
        const bf16x4 v = *(const bf16x4*)((const __bf16*)base + i);
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
    return *(const f32x4*)((const float*)base + i);
}
template <bool Q> __device__ __forceinline__ float ldw1(const void* base, size_t i) {
    return Q ? (float)((const __bf16*)base)[i] : ((const float*)base)[i];
}

// master f32 -> bf16 copy and its widened f32 image
__global__ void quantize_kernel(const float* __restrict__ prm, __bf16* __restrict__ pq, float* __restrict__ pqf, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const __bf16 b = (__bf16)prm[i];
    pq[i] = b;
    pqf[i] = (float)b;
}

constexpr int TP = 32;          // pairs per workgroup in the MFMA forward
constexpr int ROWP = TP * 8 + 4;  // floats per k8 block in LDS (pad 4: conflict-free b128 writes)
constexpr int KC = 512;         // K chunk staged per pass

struct HeadFwd {
    const float *L, *R;
    const int32_t *li, *ri;
    long long P;
    const float *w1p, *b1, *w2p, *b2, *w3, *b3;
    float* probs;
    int D, h1, h2;
    int accumulate;     // add to probs already there (committee member > 0)
    float final_div;    // > 0: divide by it after adding (last committee member)
    int matN;           // > 0: score-matrix mode, pair p = (mat_row0 + p / matN, p % matN) of one matrix L == R
    int mat_row0;
    int out_col;        // >= 0: write only this softmax column, probs is [P] (else [P][od])
    int od;             // 2: softmax over Dense(2); 1: sigmoid of Dense(1)
    int q;              // bf16 compute mode: |l - r|, a1, a2 rounded to bf16 (the weights handed in are already)
};

// out-of-place repack W (in,out) row-major -> [in/8][out][2][4]:  k = 8*k8 + 2*s + h
__global__ void pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int in, int out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)in * out) return;
    const int k = (int)(i / out), c = (int)(i - (long long)k * out);
    const int k8 = k >> 3, kk = k & 7, h = kk & 1, s = kk >> 1;
    wp[((size_t)k8 * out + c) * 8 + h * 4 + s] = w[i];
}

template <int CT>
__global__ __launch_bounds__(256, 2) void head_fwd_kernel(const HeadFwd p) {
    extern __shared__ __attribute__((aligned(16))) float buf[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long p0 = (long long)blockIdx.x * TP;
    const int D = p.D, h1 = p.h1, h2 = p.h2;
    const int l31 = lane & 31, hh = lane >> 5;
    const int colbase = wave * (32 * CT);

    f32x16 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

    // rows of the embedding matrices this tile's 32 pairs read (computed once, kept in LDS)
    __shared__ long long s_lrow[TP], s_rrow[TP];
    if (tid < TP) {
        long long pp = p0 + tid;
        if (pp >= p.P) pp = p.P - 1;
        long long lrow, rrow;
        if (p.matN > 0) {
            const long long qd = pp / p.matN;
            lrow = p.mat_row0 + qd;
            rrow = pp - qd * p.matN;
        } else {
            lrow = p.li ? (long long)p.li[pp] : pp;
            rrow = p.ri ? (long long)p.ri[pp] : pp;
        }
        s_lrow[tid] = lrow;
        s_rrow[tid] = rrow;
    }
    for (int k0 = 0; k0 < D; k0 += KC) {
        const int kc = min(KC, D - k0), nk8 = kc >> 3;
        __syncthreads();
        for (int idx = tid; idx < TP * nk8; idx += 256) {
            const int row = idx / nk8, k8 = idx - row * nk8;
            const long long lrow = s_lrow[row], rrow = s_rrow[row];
            const float* lp = p.L + lrow * D + k0 + k8 * 8;
            const float* rp = p.R + rrow * D + k0 + k8 * 8;
            const f32x4 l0 = *(const f32x4*)lp, l1 = *(const f32x4*)(lp + 4);
            const f32x4 r0 = *(const f32x4*)rp, r1 = *(const f32x4*)(rp + 4);
            f32x4 e, o;   // even k (h = 0) and odd k (h = 1), s = 0..3
            e[0] = fabsf(l0[0] - r0[0]); o[0] = fabsf(l0[1] - r0[1]);
            e[1] = fabsf(l0[2] - r0[2]); o[1] = fabsf(l0[3] - r0[3]);
            e[2] = fabsf(l1[0] - r1[0]); o[2] = fabsf(l1[1] - r1[1]);
            e[3] = fabsf(l1[2] - r1[2]); o[3] = fabsf(l1[3] - r1[3]);
            if (p.q) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { e[j] = qbf(e[j]); o[j] = qbf(o[j]); }
            }
            float* d = buf + k8 * ROWP + row * 8;
            *(f32x4*)d = e;
            *(f32x4*)(d + 4) = o;
        }
        __syncthreads();
        const float* wbase = p.w1p + ((size_t)(k0 >> 3) * h1 + colbase + l31) * 8 + hh * 4;
        for (int k8 = 0; k8 < nk8; ++k8) {
            const f32x4 a = *(const f32x4*)(buf + k8 * ROWP + l31 * 8 + hh * 4);
            f32x4 b[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) b[c] = *(const f32x4*)(wbase + ((size_t)k8 * h1 + c * 32) * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[c][s], acc[c], 0, 0, 0);
        }
    }
    __syncthreads();
    // A1 = relu(Z1 + b1) -> LDS in the packed A-operand layout of layer 2
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int col = colbase + c * 32 + l31;
        const float bb = p.b1[col];
        float* d = buf + (col >> 3) * ROWP + (col & 1) * 4 + ((col & 7) >> 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
            d[row * 8] = qa(fmaxf(acc[c][r] + bb, 0.f), p.q);
        }
    }
    __syncthreads();
    // layer 2: wave = (column tile, K split)
    const int nt2 = h2 >> 5, tile = wave % nt2, ks = wave / nt2, nsplit = 4 / nt2;
    const int nk8_2 = h1 >> 3, kper = nk8_2 / nsplit;
    f32x16 acc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[i] = 0.f;
    {
        const float* w2b = p.w2p + ((size_t)tile * 32 + l31) * 8 + hh * 4;
        for (int k8 = ks * kper; k8 < (ks + 1) * kper; ++k8) {
            const f32x4 a = *(const f32x4*)(buf + k8 * ROWP + l31 * 8 + hh * 4);
            const f32x4 b = *(const f32x4*)(w2b + (size_t)k8 * h2 * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc2, 0, 0, 0);
        }
    }
    __syncthreads();
    // partials [ks][row][col]
    {
        float* part = buf + ks * (TP * h2);
        const int col = tile * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
            part[row * h2 + col] = acc2[r];
        }
    }
    __syncthreads();
    float* a2 = buf + 4 * TP * 64;      // after the (at most 4 x 32 x 64) partials
    for (int i = tid; i < TP * h2; i += 256) {
        float s = 0.f;
        for (int z = 0; z < nsplit; ++z) s += buf[z * (TP * h2) + i];
        a2[i] = qa(fmaxf(s + p.b2[i % h2], 0.f), p.q);
    }
    __syncthreads();
    if (tid < 2 * TP && p.od == 2) {
        const int row = tid >> 1, cls = tid & 1;
        float z = 0.f;
        for (int c = 0; c < h2; ++c) z = fmaf(a2[row * h2 + c], p.w3[c * 2 + cls], z);
        z += p.b3[cls];
        const float zo = __shfl_xor(z, 1, 64);
        const float m = fmaxf(z, zo);
        const float e = expf(z - m), eo = expf(zo - m);
        float pr = e / (e + eo);
        const long long pp = p0 + row;
        if (pp < p.P && (p.out_col < 0 || p.out_col == cls)) {
            float* o = p.out_col < 0 ? p.probs + pp * 2 + cls : p.probs + pp;
            if (p.accumulate) pr += *o;
            if (p.final_div > 0.f) pr = pr / p.final_div;
            *o = pr;
        }
    } else if (tid < TP && p.od == 1) {
        const int row = tid;
        float z = 0.f;
        for (int c = 0; c < h2; ++c) z = fmaf(a2[row * h2 + c], p.w3[c], z);
        z += p.b3[0];
        float pr = 1.f / (1.f + expf(-z));
        const long long pp = p0 + row;
        if (pp < p.P) {
            float* o = p.probs + pp;
            if (p.accumulate) pr += *o;
            if (p.final_div > 0.f) pr = pr / p.final_div;
            *o = pr;
        }
    }
}


// ------------------------------- bf16 compute mode: pair scoring on the bf16 matrix cores ------------------------------
// The predict path of the bf16 compute mode for the head the reference builds (h1 = 512, h2 = 64, 2-way softmax, D a
// multiple of 512: code/siamese.py:27-32): same arithmetic as head_fwd_kernel with p.q (operands rounded to bf16,
// exact products, f32 sums) on v_mfma_f32_16x16x32_bf16 instead of the f32-input instruction (16x the rate), so that
// pool-scale scoring (1.6 M pairs per GPU and pass in configs[2], 60 M in the DFW score matrix) is bound by the
// gather of the embedding rows, not by the multiply.
//   workgroup = 64 pairs, 4 waves; |l - r| of the current 512-deep K chunk as bf16 in LDS ([pair][k], 16-B pieces
//   XOR-swizzled by the pair so that every ds_read_b128 lane group hits 16 distinct bank slots); layer 1: wave w owns
//   output channels 128 w .. 128 w + 127 (8 MFMA tiles) x 64 pairs (4 tiles) = 32 accumulators, weight fragments
//   straight from global (W1^T [h1][D] bf16, L2-resident, next K-step's fragments in flight); a1 = relu(z1 + b1) as bf16
//   back into the same LDS image; layer 2: wave w owns pair tile w x 64 channels; layer 3 + softmax on the VALU.
typedef __attribute__((ext_vector_type(8))) __bf16 hbf16x8;
struct HeadFwdQ {
    const float *L, *R;
    const int32_t *li, *ri;
    long long P;
    const __bf16 *w1t, *w2t;        // [512][D], [64][512]
    const float *b1, *b2, *w3, *b3;  // w3: the bf16 values widened to f32, [64][2]
    float* probs;
    int D, accumulate;
    float final_div;
};
constexpr int QP = 64;                 // pairs per workgroup
__device__ __forceinline__ int q_lds_off(int pair, int k) {      // byte offset of element (pair, k) of the [64][512] bf16 image
    return pair * 1024 + ((((k >> 3) ^ (pair & 15)) << 4) | ((k & 7) << 1));
}
__global__ __launch_bounds__(256, 2) void head_fwd_bf16_kernel(const HeadFwdQ p) {
    extern __shared__ __attribute__((aligned(16))) char qs[];    // 64 KB image + row indices
    long long* const s_lrow = (long long*)(qs + 65536);
    long long* const s_rrow = s_lrow + QP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ln = lane & 15, kb = lane >> 4;
    const long long p0 = (long long)blockIdx.x * QP;
    const int D = p.D;
    if (tid < QP) {
        long long pp = p0 + tid;
        if (pp >= p.P) pp = p.P - 1;
        s_lrow[tid] = p.li ? (long long)p.li[pp] : pp;
        s_rrow[tid] = p.ri ? (long long)p.ri[pp] : pp;
    }
    f32x4 acc[8][4];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[c][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    // weight fragment of MFMA tile c at K-step ks: rows (channels) 128 wave + 16 c + ln, k = 32 ks + 8 kb .. + 7
    const __bf16* w1row = p.w1t + (size_t)(wave * 128 + ln) * D + kb * 8;
    for (int k0 = 0; k0 < D; k0 += 512) {
        __syncthreads();                                           // row indices ready / previous chunk's reads done
        // gather: 64 pairs x 64 pieces of 8 k; thread -> (pair = i >> 6, piece = i & 63), i = tid + 256 j
#pragma unroll 4
        for (int j = 0; j < 16; ++j) {
            const int i = tid + 256 * j, pair = i >> 6, piece = i & 63;
            const float* lp = p.L + s_lrow[pair] * D + k0 + piece * 8;
            const float* rp = p.R + s_rrow[pair] * D + k0 + piece * 8;
            const f32x4 l0 = *(const f32x4*)lp, l1 = *(const f32x4*)(lp + 4), r0 = *(const f32x4*)rp, r1 = *(const f32x4*)(rp + 4);
            hbf16x8 d;
#pragma unroll
            for (int e = 0; e < 4; ++e) { d[e] = (__bf16)fabsf(l0[e] - r0[e]); d[4 + e] = (__bf16)fabsf(l1[e] - r1[e]); }
            *(hbf16x8*)(qs + pair * 1024 + ((piece ^ (pair & 15)) << 4)) = d;
        }
        __syncthreads();
        hbf16x8 wa[8], wb[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) wa[c] = *(const hbf16x8*)(w1row + (size_t)c * 16 * D + k0);
#pragma unroll 1
        for (int ks = 0; ks < 16; ks += 2) {
            // two K-steps per trip, fragments of the second in flight under the first's MFMAs and vice versa
#pragma unroll
            for (int c = 0; c < 8; ++c) wb[c] = *(const hbf16x8*)(w1row + (size_t)c * 16 * D + k0 + (ks + 1) * 32);
            {
                hbf16x8 pf[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) pf[u] = *(const hbf16x8*)(qs + (16 * u + ln) * 1024 + ((((ks * 4 + kb)) ^ ln) << 4));
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[c][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[c], pf[u], acc[c][u], 0, 0, 0);
            }
            if (ks + 2 < 16) {
#pragma unroll
                for (int c = 0; c < 8; ++c) wa[c] = *(const hbf16x8*)(w1row + (size_t)c * 16 * D + k0 + (ks + 2) * 32);
            }
            {
                hbf16x8 pf[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) pf[u] = *(const hbf16x8*)(qs + (16 * u + ln) * 1024 + (((((ks + 1) * 4 + kb)) ^ ln) << 4));
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[c][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[c], pf[u], acc[c][u], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    // a1 = relu(z1 + b1) as bf16 into the image: accumulator element j of tile (c, u) is channel 128 wave + 16 c + 4 kb + j,
    // pair 16 u + ln -> 4 consecutive channels = one 8-byte store
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int ch = wave * 128 + 16 * c + 4 * kb;
        const f32x4 bb = *(const f32x4*)(p.b1 + ch);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            typedef __attribute__((ext_vector_type(4))) __bf16 b4;
            b4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (__bf16)fmaxf(acc[c][u][j] + bb[j], 0.f);
            *(b4*)(qs + q_lds_off(16 * u + ln, ch)) = o;
        }
    }
    __syncthreads();
    // layer 2: pair tile `wave`, 4 channel tiles, K = 512
    f32x4 acc2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc2[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const __bf16* w2row = p.w2t + (size_t)ln * 512 + kb * 8;
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {
        const hbf16x8 pf = *(const hbf16x8*)(qs + (16 * wave + ln) * 1024 + (((ks * 4 + kb) ^ ln) << 4));
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const hbf16x8 wf = *(const hbf16x8*)(w2row + (size_t)c * 16 * 512 + ks * 32);
            acc2[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, pf, acc2[c], 0, 0, 0);
        }
    }
    __syncthreads();                                               // every wave is done reading a1: the image's start is free
    float* const a2 = (float*)qs;                                  // [64 pairs][64 channels] f32 (bf16-valued)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = 16 * c + 4 * kb;
        const f32x4 bb = *(const f32x4*)(p.b2 + ch);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = qbf(fmaxf(acc2[c][j] + bb[j], 0.f));
        *(f32x4*)(a2 + (16 * wave + ln) * 64 + ch) = o;
    }
    __syncthreads();
    if (tid < 2 * QP) {
        const int row = tid >> 1, cls = tid & 1;
        float z = 0.f;
#pragma unroll 8
        for (int c = 0; c < 64; ++c) z = fmaf(a2[row * 64 + c], p.w3[c * 2 + cls], z);
        z += p.b3[cls];
        const float zo = __shfl_xor(z, 1, 64);
        const float m = fmaxf(z, zo);
        const float e = expf(z - m), eo = expf(zo - m);
        float pr = e / (e + eo);
        const long long pp = p0 + row;
        if (pp < p.P) {
            float* o = p.probs + pp * 2 + cls;
            if (p.accumulate) pr += *o;
            if (p.final_div > 0.f) pr = pr / p.final_div;
            *o = pr;
        }
    }
}
// wt[c][k] = pq[k][c]: W (in, out) row-major bf16 -> W^T
__global__ void transpose_bf16_kernel(const __bf16* __restrict__ w, __bf16* __restrict__ wt, int in, int out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)in * out) return;
    const int c = (int)(i / in), k = (int)(i - (long long)c * in);
    wt[i] = w[(size_t)k * out + c];
}

// ------------------------------- small-batch train / eval kernels --------------------------------
__global__ void absdiff_kernel(const float* __restrict__ L, const float* __restrict__ R,
                               float* __restrict__ dm, int n, int D) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n * D) dm[i] = fabsf(L[i] - R[i]);
}

// z[r][c] = sum_k a[r][k] * w[k][c] + b[c];  one thread per (r, c), c fastest
__global__ void dense_fwd_kernel(const float* __restrict__ a, const float* __restrict__ w,
                                 const float* __restrict__ b, float* __restrict__ z, int n, int K, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n * C) return;
    const int r = i / C, c = i - r * C;
    const float* ar = a + (size_t)r * K;
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < K; ++k) s = fmaf(ar[k], w[(size_t)k * C + c], s);
    z[i] = s + b[c];
}
// same but the input is relu(zin) (the stored pre-activation of the previous layer)
__global__ void dense_fwd_relu_in_kernel(const float* __restrict__ zin, const float* __restrict__ w,
                                         const float* __restrict__ b, float* __restrict__ z, int n, int K,
                                         int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n * C) return;
    const int r = i / C, c = i - r * C;
    const float* ar = zin + (size_t)r * K;
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < K; ++k) s = fmaf(fmaxf(ar[k], 0.f), w[(size_t)k * C + c], s);
    z[i] = s + b[c];
}

// Small-batch dense forward with enough waves to hide latency: block = (one row, 64 columns) x 8
// K-slices (one wave each, coalesced weight rows, broadcast activation), fixed-order LDS reduce.
// z[r][c] = sum_k act(a[r][k]) * w[k][c] + b[c]   (act = relu when relu_in).  K % 8 == 0.
// With Rm != nullptr the input is |a - Rm| computed on the fly (the Lambda layer of code/siamese.py:27),
// and the first column block also stores it to dm_out for the weight gradient.
__global__ __launch_bounds__(512) void dense_fwd_tiled_kernel(const float* __restrict__ a,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ b, float* __restrict__ z,
                                                             int n, int K, int C, int relu_in,
                                                             const float* __restrict__ Rm, float* __restrict__ dm_out, int q,
                                                             const int* __restrict__ li = nullptr, const int* __restrict__ ri = nullptr) {
    // li / ri (first layer only): row r of the batch is row li[r] of `a` against row ri[r] of `Rm` (a table of features)
    __shared__ float part[8][64];
    const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, r = blockIdx.y;
    const int kper = K >> 3;
    const float* ar = a + (size_t)(li ? li[r] : r) * K + ks * kper;
    const float* wp = w + (size_t)(ks * kper) * C + c;
    float s = 0.f;
    if (Rm) {
        const float* rr = Rm + (size_t)(ri ? ri[r] : r) * K + ks * kper;
        if (blockIdx.x == 0)
            for (int k = lane; k < kper; k += 64) dm_out[(size_t)r * K + ks * kper + k] = qa(fabsf(ar[k] - rr[k]), q);
        if (c < C) {
#pragma unroll 8
            for (int k = 0; k < kper; ++k) s = fmaf(qa(fabsf(ar[k] - rr[k]), q), wp[(size_t)k * C], s);
        }
    } else if (c < C) {
#pragma unroll 8
        for (int k = 0; k < kper; ++k) {
            float av = ar[k];
            if (relu_in) av = fmaxf(av, 0.f);
            s = fmaf(av, wp[(size_t)k * C], s);
        }
    }
    part[ks][lane] = s;
    __syncthreads();
    if (ks == 0 && c < C) {
        float t = part[0][lane];
#pragma unroll
        for (int i = 1; i < 8; ++i) t += part[i][lane];
        z[(size_t)r * C + c] = qa(t + b[c], q);     // bf16 mode: relu(z) is the next GEMM's operand, rounding commutes with relu
    }
}

struct HeadLoss {
    const float *z2, *w3, *b3, *y, *sw;
    float *probs, *dz3, *dz2, *gw3, *gb3, *metrics;
    int n, h2, want_grads, od;
    float grad_scale;   // <= 0: 1 / count(sw != 0)
    int q;              // bf16 compute mode: dz3 / dz2 rounded to bf16 where produced
};

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// One workgroup: layer 3, softmax, Keras BCE + metrics, and (if want_grads) dZ3, dW3, db3, dZ2.
__global__ __launch_bounds__(256) void head_loss_kernel(const HeadLoss p) {
    __shared__ float red[4];
    const int tid = threadIdx.x, n = p.n, h2 = p.h2;
    float cnt = 0.f;
    for (int i = tid; i < n; i += 256) cnt += (p.sw ? (p.sw[i] != 0.f) : 1.f);
    cnt = block_sum(cnt, red);
    const float scale = p.grad_scale > 0.f ? p.grad_scale : 1.f / cnt;

    float lsum = 0.f, asum = 0.f;
    const int od = p.od;
    for (int i = tid; i < n; i += 256) {
        const float* a = p.z2 + (size_t)i * h2;
        const float w = p.sw ? p.sw[i] : 1.f;
        if (od == 1) {
            // Dense(1, sigmoid) + binary_crossentropy over one output (code/siamese3.py:25-28)
            float z = 0.f;
            for (int c = 0; c < h2; ++c) z = fmaf(fmaxf(a[c], 0.f), p.w3[c], z);
            z += p.b3[0];
            const float pr = 1.f / (1.f + expf(-z));
            p.probs[i] = pr;
            const float y = p.y[i];
            const float pc = fminf(fmaxf(pr, 1e-7f), 1.f - 1e-7f);
            const float x = logf(pc / (1.f - pc));
            lsum += (fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)))) * w;
            asum += (rintf(pr) == y) ? 1.f : 0.f;
            if (p.want_grads) {
                const bool inside = pr >= 1e-7f && pr <= 1.f - 1e-7f;
                const float dp = inside ? w * scale * (pc - y) / (pc * (1.f - pc)) : 0.f;
                p.dz3[i] = qa(dp * pr * (1.f - pr), p.q);
            }
            continue;
        }
        float z0 = 0.f, z1 = 0.f;
        for (int c = 0; c < h2; ++c) {
            const float v = fmaxf(a[c], 0.f);
            z0 = fmaf(v, p.w3[c * 2 + 0], z0);
            z1 = fmaf(v, p.w3[c * 2 + 1], z1);
        }
        z0 += p.b3[0];
        z1 += p.b3[1];
        const float m = fmaxf(z0, z1);
        const float e0 = expf(z0 - m), e1 = expf(z1 - m);
        const float pr[2] = {e0 / (e0 + e1), e1 / (e0 + e1)};
        p.probs[i * 2 + 0] = pr[0];
        p.probs[i * 2 + 1] = pr[1];
        float li = 0.f, acc = 0.f, dp[2];
        for (int c = 0; c < 2; ++c) {
            const float y = p.y[i * 2 + c];
            const float pc = fminf(fmaxf(pr[c], 1e-7f), 1.f - 1e-7f);
            const float x = logf(pc / (1.f - pc));
            li += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
            acc += (rintf(pr[c]) == y) ? 1.f : 0.f;
            const bool inside = pr[c] >= 1e-7f && pr[c] <= 1.f - 1e-7f;
            // d/dp of BCE(y, clip(p)) through logit + sigmoid CE = (pc - y) / (pc (1 - pc))
            dp[c] = inside ? 0.5f * w * scale * (pc - y) / (pc * (1.f - pc)) : 0.f;
        }
        lsum += 0.5f * li * w;
        asum += 0.5f * acc;
        if (p.want_grads) {
            // through softmax: dz_j = p_j * (dp_j - sum_c dp_c p_c)
            const float dot = dp[0] * pr[0] + dp[1] * pr[1];
            p.dz3[i * 2 + 0] = qa(pr[0] * (dp[0] - dot), p.q);
            p.dz3[i * 2 + 1] = qa(pr[1] * (dp[1] - dot), p.q);
        }
    }
    lsum = block_sum(lsum, red);
    asum = block_sum(asum, red);
    if (tid == 0) {
        p.metrics[0] = lsum * (p.grad_scale > 0.f ? p.grad_scale : 1.f / cnt);
        p.metrics[1] = asum / (float)n;
    }
    if (!p.want_grads) return;
    __syncthreads();   // dz3 written by this block (global, same block -> visible after barrier)
    // dW3[c][j] = sum_i relu(z2[i][c]) * dz3[i][j];  db3[j] = sum_i dz3[i][j]
    for (int t = tid; t < h2 * od + od; t += 256) {
        float s = 0.f;
        if (t < h2 * od) {
            const int c = t / od, j = t - c * od;
            for (int i = 0; i < n; ++i) s = fmaf(fmaxf(p.z2[(size_t)i * h2 + c], 0.f), p.dz3[i * od + j], s);
            p.gw3[t] = s;
        } else {
            const int j = t - h2 * od;
            for (int i = 0; i < n; ++i) s += p.dz3[i * od + j];
            p.gb3[j] = s;
        }
    }
    // dZ2[i][c] = (z2 > 0) * sum_j dz3[i][j] * w3[c][j]
    for (int t = tid; t < n * h2; t += 256) {
        const int i = t / h2, c = t - i * h2;
        float g = 0.f;
        for (int j = 0; j < od; ++j) g = fmaf(p.dz3[i * od + j], p.w3[c * od + j], g);
        p.dz2[t] = p.z2[t] > 0.f ? qa(g, p.q) : 0.f;
    }
}

// gw[k][c] = sum_i act(a[i][k]) * dz[i][c];  gb[c] = sum_i dz[i][c]   (blockIdx.y == 0 part)
// relu_a: a holds pre-activations, use relu(a)
__global__ void dense_wgrad_kernel(const float* __restrict__ a, const float* __restrict__ dz,
                                   float* __restrict__ gw, float* __restrict__ gb, int n, int K, int C,
                                   int relu_a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < K * C) {
        const int k = i / C, c = i - k * C;
        float s = 0.f;
        for (int r = 0; r < n; ++r) {
            float av = a[(size_t)r * K + k];
            if (relu_a) av = fmaxf(av, 0.f);
            s = fmaf(av, dz[(size_t)r * C + c], s);
        }
        gw[i] = s;
    } else if (i < K * C + C) {
        const int c = i - K * C;
        float s = 0.f;
        for (int r = 0; r < n; ++r) s += dz[(size_t)r * C + c];
        gb[c] = s;
    }
}
// dzin[r][k] = (zin[r][k] > 0) * sum_c dz[r][c] * w[k][c]
__global__ void dense_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                   const float* __restrict__ zin, float* __restrict__ dzin, int n, int K,
                                   int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n * K) return;
    const int r = i / K, k = i - r * K;
    const float* wr = w + (size_t)k * C;
    const float* dr = dz + (size_t)r * C;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s = fmaf(dr[c], wr[c], s);
    dzin[i] = zin[i] > 0.f ? s : 0.f;
}

// d|l-r|/dl: dL[r][k] = sgn(L-R) * sum_c dz1[r][c] * W1[k][c];  dR = -dL   (tf.abs gradient: sign, 0 at 0)
// dL[r][k] = sign(L - R) * sum_c dZ1[r][c] W1[k][c], dR = -dL.  Workgroup = 64 inputs k x 16 rows, W1 and dZ1 staged through
// LDS 128 columns at a time (a thread per (r, k) walking W1 row k straight from memory read it at a 4 * h1-byte stride from
// lane to lane: 19.8 us for 8 MFLOP at n = 32, D = 2048 until round 6); c ascending per output, as before.
// relu_in: L and R are the outputs of a ReLU (SmallRes' tower ends in one) and the gradients wanted are those w.r.t. its
// pre-activations: the mask (x > 0) is applied here instead of by a launch of its own.
__global__ __launch_bounds__(256) void head_input_grad_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                              const float* __restrict__ dz1, const float* __restrict__ w1,
                                                              float* __restrict__ dL, float* __restrict__ dR, int n, int D, int h1,
                                                              int relu_in) {
    __shared__ float Ws[64 * 129];
    __shared__ __attribute__((aligned(16))) float Dz[16 * 128];
    const int tid = threadIdx.x, kl = tid & 63, rg = tid >> 6;
    const int k0 = blockIdx.x * 64, r0 = blockIdx.y * 16;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int cc = 0; cc < h1; cc += 128) {
        for (int i = tid; i < 64 * 32; i += 256) {
            const int row = i >> 5, c4 = (i & 31) * 4;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (k0 + row < D && cc + c4 < h1) t = *(const f32x4*)(w1 + (size_t)(k0 + row) * h1 + cc + c4);
            float* d = Ws + row * 129 + c4;
            d[0] = t[0]; d[1] = t[1]; d[2] = t[2]; d[3] = t[3];
        }
        for (int i = tid; i < 16 * 32; i += 256) {
            const int row = i >> 5, c4 = (i & 31) * 4;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (r0 + row < n && cc + c4 < h1) t = *(const f32x4*)(dz1 + (size_t)(r0 + row) * h1 + cc + c4);
            *(f32x4*)(Dz + row * 128 + c4) = t;
        }
        __syncthreads();
        const int cend = min(128, h1 - cc);
        for (int c = 0; c < cend; ++c) {
            const float w = Ws[kl * 129 + c];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = fmaf(Dz[(rg * 4 + q) * 128 + c], w, acc[q]);
        }
        __syncthreads();
    }
    const int k = k0 + kl;
    if (k >= D) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = r0 + rg * 4 + q;
        if (r >= n) continue;
        const size_t i = (size_t)r * D + k;
        const float l = L[i], r_ = R[i], d = l - r_;
        const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        const float gl = acc[q] * sg, gr = -acc[q] * sg;
        dL[i] = (relu_in && !(l > 0.f)) ? 0.f : gl;
        dR[i] = (relu_in && !(r_ > 0.f)) ? 0.f : gr;
    }
}

// One launch for the two independent halves of the middle of the backward pass:
//   blocks [0, nb_w):   gW2[k][c] = sum_i relu(z1[i][k]) dz2[i][c], gb2        (dense_wgrad_kernel's body)
//   blocks [nb_w, ...): dz1[r][k] = (z1 > 0) sum_c dz2[r][c] W2[k][c]          (dense_dgrad_kernel's body)
__global__ void head_bwd_mid_kernel(const float* __restrict__ z1, const float* __restrict__ dz2,
                                    const float* __restrict__ w2, float* __restrict__ gw2, float* __restrict__ gb2,
                                    float* __restrict__ dz1, int n, int h1, int h2, int nb_w, int q) {
    if ((int)blockIdx.x < nb_w) {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i < h1 * h2) {
            const int k = i / h2, c = i - k * h2;
            float s = 0.f;
            for (int r = 0; r < n; ++r) s = fmaf(fmaxf(z1[(size_t)r * h1 + k], 0.f), dz2[(size_t)r * h2 + c], s);
            gw2[i] = s;
        } else if (i < h1 * h2 + h2) {
            const int c = i - h1 * h2;
            float s = 0.f;
            for (int r = 0; r < n; ++r) s += dz2[(size_t)r * h2 + c];
            gb2[c] = s;
        }
        return;
    }
    const int i = (blockIdx.x - nb_w) * 256 + threadIdx.x;
    if (i >= n * h1) return;
    const int r = i / h1, k = i - r * h1;
    const float* wr = w2 + (size_t)k * h2;
    const float* dr = dz2 + (size_t)r * h2;
    float s = 0.f;
    for (int c = 0; c < h2; ++c) s = fmaf(dr[c], wr[c], s);
    dz1[i] = z1[i] > 0.f ? qa(s, q) : 0.f;
}

__device__ __forceinline__ void adadelta_one(float* prm, float* a, float* d, size_t i, float gi, float lr, float rho,
                                             float eps) {
    const float na = rho * a[i] + (1.f - rho) * gi * gi;
    const float u = gi * sqrtf(d[i] + eps) / sqrtf(na + eps);
    prm[i] = prm[i] - lr * u;
    d[i] = rho * d[i] + (1.f - rho) * u * u;
    a[i] = na;
}

// Last launch of the fused train step: the first-layer weight/bias gradient (gW1 = dm^T dz1, gb1) with
// its Adadelta update applied in the same thread, and — in the remaining blocks — the Adadelta update
// of every later parameter from the gradients the earlier launches stored.  All gradients were taken
// with the old weights (nothing upstream reads W1 after this point; W2/W3 are updated only here).
__global__ void head_wgrad1_update_kernel(const float* __restrict__ dm, const float* __restrict__ dz1,
                                          float* __restrict__ prm, float* __restrict__ g, float* __restrict__ a,
                                          float* __restrict__ d, int n, int D, int h1, size_t n_first, size_t nparams,
                                          int nb_first, float lr, float rho, float eps) {
    if ((int)blockIdx.x < nb_first) {
        const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
        if (i >= n_first) return;
        float s = 0.f;
        if (i < (size_t)D * h1) {
            const int k = (int)(i / h1), c = (int)(i - (size_t)k * h1);
            for (int r = 0; r < n; ++r) s = fmaf(dm[(size_t)r * D + k], dz1[(size_t)r * h1 + c], s);
        } else {
            const int c = (int)(i - (size_t)D * h1);
            for (int r = 0; r < n; ++r) s += dz1[(size_t)r * h1 + c];
        }
        g[i] = s;
        adadelta_one(prm, a, d, i, s, lr, rho, eps);
        return;
    }
    const size_t i = n_first + (size_t)(blockIdx.x - nb_first) * 256 + threadIdx.x;
    if (i < nparams) adadelta_one(prm, a, d, i, g[i], lr, rho, eps);
}

__global__ void adadelta_kernel(float* __restrict__ prm, const float* __restrict__ g, float* __restrict__ a,
                                float* __restrict__ d, size_t n, float lr, float rho, float eps) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float na = rho * a[i] + (1.f - rho) * gi * gi;
    const float u = gi * sqrtf(d[i] + eps) / sqrtf(na + eps);
    prm[i] = prm[i] - lr * u;
    d[i] = rho * d[i] + (1.f - rho) * u * u;
    a[i] = na;
}

// One launch for the update of TWO parameter blocks (an end-to-end model's tower, then the head): workgroups [0, nb2) take block 2
__global__ void adadelta_two_kernel(float* __restrict__ p2, const float* __restrict__ g2, float* __restrict__ a2, float* __restrict__ d2,
                                    size_t n2, float* __restrict__ p1, const float* __restrict__ g1v, float* __restrict__ a1,
                                    float* __restrict__ d1, size_t n1, int nb2, float lr, float rho, float eps) {
    const bool second = (int)blockIdx.x < nb2;
    const size_t i = (size_t)(second ? blockIdx.x : blockIdx.x - nb2) * 256 + threadIdx.x;
    if (i >= (second ? n2 : n1)) return;
    float* prm = second ? p2 : p1;
    float* a = second ? a2 : a1;
    float* d = second ? d2 : d1;
    const float gi = second ? g2[i] : g1v[i];
    const float na = rho * a[i] + (1.f - rho) * gi * gi;
    const float u = gi * sqrtf(d[i] + eps) / sqrtf(na + eps);
    prm[i] = prm[i] - lr * u;
    d[i] = rho * d[i] + (1.f - rho) * u * u;
    a[i] = na;
}

inline dim3 g1(long long n) { return dim3((unsigned)((n + 255) / 256), 1, 1); }

int head_alloc(alink_head* h, float** p, size_t count) {
    ALINK_HIP(hipMalloc((void**)p, count * sizeof(float)));
    h->allocs.push_back(*p);
    ALINK_HIP(hipMemset(*p, 0, count * sizeof(float)));
    return ALINK_OK;
}

// bf16 mode: bring the bf16 copy (and, if asked, its widened image) up to date with the f32 master parameters
int ensure_q(alink_head* h, hipStream_t st, bool need_widened) {
    if (!h->qmode) return ALINK_OK;
    if (!h->pq_dirty && !(need_widened && h->pqf_dirty)) return ALINK_OK;
    hipLaunchKernelGGL(quantize_kernel, g1((long long)h->nparams), dim3(256), 0, st, h->d_params, h->d_pq, h->d_pqf,
                       h->nparams);
    ALINK_HIP(hipGetLastError());
    h->pq_dirty = h->pqf_dirty = false;
    return ALINK_OK;
}

int ensure_packed(alink_head* h, hipStream_t st) {
    if (!h->packed_dirty) return ALINK_OK;
    if (h->qmode) { const int rc = ensure_q(h, st, true); if (rc) return rc; }
    const float* src = h->qmode ? h->d_pqf : h->d_params;
    hipLaunchKernelGGL(pack_kernel, g1((long long)h->D * h->h1), dim3(256), 0, st, src + h->oW1,
                       h->d_w1p, h->D, h->h1);
    hipLaunchKernelGGL(pack_kernel, g1((long long)h->h1 * h->h2), dim3(256), 0, st, src + h->oW2,
                       h->d_w2p, h->h1, h->h2);
    if (h->qmode && h->d_wt) {      // W1^T, W2^T in bf16 for head_fwd_bf16_kernel
        hipLaunchKernelGGL(transpose_bf16_kernel, g1((long long)h->D * h->h1), dim3(256), 0, st, h->d_pq + h->oW1, h->d_wt,
                           h->D, h->h1);
        hipLaunchKernelGGL(transpose_bf16_kernel, g1((long long)h->h1 * h->h2), dim3(256), 0, st, h->d_pq + h->oW2,
                           h->d_wt + (size_t)h->D * h->h1, h->h1, h->h2);
    }
    ALINK_HIP(hipGetLastError());
    h->packed_dirty = false;
    return ALINK_OK;
}

bool g_use_bf16_mfma = true;       // A/B: bf16 mode's predict on the bf16 matrix cores (else the f32-input kernel with rounding)

size_t fwd_lds_bytes(int h1) {
    const size_t a = (size_t)(KC / 8) * ROWP, b = (size_t)(h1 / 8) * ROWP, c = 4 * TP * 64 + TP * 64;
    size_t m = a > b ? a : b;
    if (c > m) m = c;
    return m * sizeof(float);
}

int launch_fwd(alink_head* h, const float* L, const float* R, const int32_t* li, const int32_t* ri,
               long long P, float* probs, int accumulate, float final_div, hipStream_t st, int matN = 0,
               int mat_row0 = 0, int out_col = -1) {
    int rc = ensure_packed(h, st);
    if (rc) return rc;
    if (h->qmode && g_use_bf16_mfma && h->d_wt && h->h1 == 512 && h->h2 == 64 && h->od == 2 && h->D % 512 == 0 && matN == 0 &&
        out_col < 0) {
        HeadFwdQ q{};
        q.L = L; q.R = R; q.li = li; q.ri = ri; q.P = P;
        q.w1t = h->d_wt; q.w2t = h->d_wt + (size_t)h->D * h->h1;
        q.b1 = h->d_params + h->ob1; q.b2 = h->d_params + h->ob2; q.w3 = h->d_pqf + h->oW3; q.b3 = h->d_params + h->ob3;
        q.probs = probs; q.D = h->D; q.accumulate = accumulate; q.final_div = final_div;
        hipLaunchKernelGGL(head_fwd_bf16_kernel, dim3((unsigned)((P + QP - 1) / QP)), dim3(256), 65536 + 2 * QP * 8, st, q);
        ALINK_HIP(hipGetLastError());
        return ALINK_OK;
    }
    HeadFwd p{};
    p.L = L; p.R = R; p.li = li; p.ri = ri; p.P = P;
    p.w1p = h->d_w1p; p.b1 = h->d_params + h->ob1; p.w2p = h->d_w2p; p.b2 = h->d_params + h->ob2;
    p.w3 = (h->qmode ? h->d_pqf : h->d_params) + h->oW3; p.b3 = h->d_params + h->ob3; p.probs = probs;
    p.q = h->qmode ? 1 : 0;
    p.D = h->D; p.h1 = h->h1; p.h2 = h->h2; p.accumulate = accumulate; p.final_div = final_div;
    p.matN = matN; p.mat_row0 = mat_row0; p.out_col = out_col; p.od = h->od;
    const size_t lds = fwd_lds_bytes(h->h1);
    const dim3 grid((unsigned)((P + TP - 1) / TP)), block(256);
    switch (h->h1 / 128) {
        case 1: hipLaunchKernelGGL(head_fwd_kernel<1>, grid, block, lds, st, p); break;
        case 2: hipLaunchKernelGGL(head_fwd_kernel<2>, grid, block, lds, st, p); break;
        case 3: hipLaunchKernelGGL(head_fwd_kernel<3>, grid, block, lds, st, p); break;
        case 4: hipLaunchKernelGGL(head_fwd_kernel<4>, grid, block, lds, st, p); break;
        default: set_error("h1=%d unsupported", h->h1); return ALINK_EINVAL;
    }
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

bool g_use_tiny = true;

// ------------------------------- tiny-batch train step (n <= 32) -------------------------------------
// The fine-tune step of code/siamese.py:52-58 runs at batch 16: far too little work to fill the chip,
// so its cost is the length of its chain of DEPENDENT launches and the memory round trips inside each.
// Rows are independent up to dZ1, and only z1 -> z2 and dZ1 -> dW1 need every column / row of the
// previous stage, so three launches suffice, each issuing all of its global loads up front:
//   A  tiny_dense1_kernel    |l - r| (stored for dW1), Dense1 pre-activations            (h1/64 x n/4 blocks)
//   B  tiny_dense2_loss_kernel  Dense2, Dense3, softmax/sigmoid, BCE, metrics partials, dZ3, dZ2, dZ1
//                            (the whole per-row backward), per-row-group partials of dW3 / db3   (n/4 blocks)
//   C  tiny_wgrad_update_kernel  dW1 tile (64 columns x 32 k) + Adadelta in place; the k-part-0 blocks
//                            also db1 and dW2 rows (+ update), block (0,0) db2, dW3, db3, metrics
//                                                                                   (h1/64 x D/32 blocks)
// No parameter is read in C except by the thread that updates it (W1 is last read in A, W2 / W3 / b* in
// B), so the in-place update races with nothing.
constexpr int TINY_N = 64;      // rows the tiny path takes (32 until round 6: customTrainModel's balanced batches are 16..48 rows)
constexpr int TINY_RG = 4;      // rows per workgroup in A and B
constexpr int TINY_KB = 32;     // W1 rows (k) per workgroup in C
constexpr int TINY_PART = 2 + 64 * 2 + 2 + 4;   // floats one row group leaves: loss, acc, dW3, db3, [132] = rows that count

// acc over K for TINY_RG rows x 4 columns per thread: 512 threads = 16 column lanes x 32 K-slices.
// a_s: [TINY_RG][K] activations in LDS; w: [K][C] row-major; the thread's columns are c0 + 4*cl ...
// All 16 weight loads of a 16-deep K chunk are in flight together.  K % 512 == 0.
// Result: out[r * 64 + c] (c < 64) in `fin` (LDS, TINY_RG * 64 floats), summed in a fixed order.
// sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), result in every lane: four v_add_f32 with a
// DPP operand (quad xor 1, quad xor 2, half-row mirror, row mirror) — no LDS crossbar (ds_bpermute) traffic
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// element offset (inside a K x C weight matrix) of this thread's first 4 weights
__device__ __forceinline__ size_t tiny_w_off(int K, int C, int c0) {
    const int tid = threadIdx.x, cl = tid & 15, ks = tid >> 4;
    return (size_t)(ks * (K >> 5)) * C + c0 + cl * 4;
}
// Q: the weights are the bf16 copy (8-byte loads, widened in registers)
template <bool Q>
__device__ __forceinline__ void tiny_load_w(f32x4 (&wv)[16], const void* wbase, size_t off, int C) {
#pragma unroll
    for (int i = 0; i < 16; ++i) wv[i] = ldw4<Q>(wbase, off + (size_t)i * C);
}
// `wv` holds the first 16-deep chunk (loaded by the caller before it staged a_s, so the weight loads and
// the activation loads share one round trip).
template <bool Q>
__device__ __forceinline__ void tiny_dense_core(const float* a_s, int K, const void* wbase, size_t woff, int C,
                                                f32x4 (&wv)[16], float* red /* [8][TINY_RG][64] */, float* fin) {
    const int tid = threadIdx.x, cl = tid & 15, ks = tid >> 4;
    const int kper = K >> 5;
    float acc[TINY_RG][4];
#pragma unroll
    for (int r = 0; r < TINY_RG; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[r][j] = 0.f;
    const float* ap = a_s + ks * kper;
    for (int k0 = 0; k0 < kper; k0 += 16) {
        if (k0) tiny_load_w<Q>(wv, wbase, woff + (size_t)k0 * C, C);
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
#pragma unroll
            for (int r = 0; r < TINY_RG; ++r) {
                const f32x4 av = *(const f32x4*)(ap + r * K + k0 + i4 * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[r][j] = fmaf(av[i], wv[i4 * 4 + i][j], acc[r][j]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < TINY_RG; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[r][j];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[r][j] = v;
        }
    if ((tid & 63) < 16) {
#pragma unroll
        for (int r = 0; r < TINY_RG; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[((tid >> 6) * TINY_RG + r) * 64 + cl * 4 + j] = acc[r][j];
    }
    __syncthreads();
    if (tid < TINY_RG * 64) {
        float s = red[tid];
#pragma unroll
        for (int wi = 1; wi < 8; ++wi) s += red[wi * TINY_RG * 64 + tid];
        fin[tid] = s;
    }
    __syncthreads();
}

template <bool Q>
__global__ __launch_bounds__(512) void tiny_dense1_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                         const void* __restrict__ w1, const float* __restrict__ b1,
                                                         float* __restrict__ z1, float* __restrict__ dm, int n, int D,
                                                         int h1, const int* __restrict__ li = nullptr,
                                                         const int* __restrict__ ri = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float tiny_lds[];
    float* a_s = tiny_lds;                        // [TINY_RG][D]
    float* red = a_s + TINY_RG * D;               // [8][TINY_RG][64]
    float* fin = red + 8 * TINY_RG * 64;          // [TINY_RG][64]
    const int tid = threadIdx.x, r0 = blockIdx.y * TINY_RG, c0 = blockIdx.x * 64;
    const size_t woff = tiny_w_off(D, h1, c0);
    f32x4 wv[16];
    tiny_load_w<Q>(wv, w1, woff, h1);
    const float bias = tid < TINY_RG * 64 ? b1[c0 + (tid & 63)] : 0.f;
    for (int i = tid; i < TINY_RG * D / 4; i += 512) {
        const int r = (i * 4) / D, k = (i * 4) - r * D, row = r0 + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < n) {
            const f32x4 l = *(const f32x4*)(L + (size_t)(li ? li[row] : row) * D + k),
                        q = *(const f32x4*)(R + (size_t)(ri ? ri[row] : row) * D + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = qa<Q>(fabsf(l[j] - q[j]));
            if (blockIdx.x == 0) *(f32x4*)(dm + (size_t)row * D + k) = v;
        }
        *(f32x4*)(a_s + i * 4) = v;
    }
    __syncthreads();
    tiny_dense_core<Q>(a_s, D, w1, woff, h1, wv, red, fin);
    if (tid < TINY_RG * 64) {
        const int r = tid >> 6, c = tid & 63, row = r0 + r;
        if (row < n) z1[(size_t)row * h1 + c0 + c] = qa<Q>(fin[tid] + bias);     // relu(q(z)) == q(relu(z))
    }
}

struct TinyLoss {
    const void* wbase;        // flat parameter vector the WEIGHTS are read from: f32 master, or its bf16 copy (Q)
    size_t oW2, oW3;
    const float *z1, *b2, *b3, *y, *sw;
    float *z2, *probs, *dz2, *dz3, *dz1, *part;
    int n, h1, od, want_grads;
    float grad_scale;
    float* eval_metrics;      // want_grads == 0: {loss, accuracy} written by the LAST row group to finish (fixed summation order)
    unsigned* counter;        // ... which is found by this counter (left at 0 again)
};

// h2 == 64.  One workgroup per TINY_RG rows.
template <bool Q>
__global__ __launch_bounds__(512) void tiny_dense2_loss_kernel(const TinyLoss p) {
    extern __shared__ __attribute__((aligned(16))) float tiny_lds[];
    const int h1 = p.h1, n = p.n, od = p.od;
    float* a_s = tiny_lds;                        // [TINY_RG][h1]  relu(z1)
    float* red = a_s + TINY_RG * h1;
    float* z2s = red + 8 * TINY_RG * 64;          // [TINY_RG][64]
    float* dz3s = z2s + TINY_RG * 64;             // [TINY_RG][2]
    float* rowm = dz3s + TINY_RG * 2;             // [TINY_RG][2]  loss, accuracy of the row
    float* cnts = rowm + TINY_RG * 2;             // [1]
    float* w3s = cnts + 4;                        // [64 * od + od]  W3 then b3 (contiguous parameters)
    float* ys = w3s + 64 * 2 + 4;                 // [TINY_RG][2]
    float* sws = ys + TINY_RG * 2;                // [TINY_RG]
    const int tid = threadIdx.x, r0 = blockIdx.x * TINY_RG;
    // every small operand is fetched now, together with the weights: nothing later waits on global memory
    const size_t woff = p.oW2 + tiny_w_off(h1, 64, 0);
    f32x4 wv[16];
    tiny_load_w<Q>(wv, p.wbase, woff, 64);
    const float bias = tid < TINY_RG * 64 ? p.b2[tid & 63] : 0.f;
    if (tid >= 64 && tid < 64 + 64 * od + od) {       // W3 (quantised in Q mode) then b3 (never quantised)
        const int t = tid - 64;
        w3s[t] = t < 64 * od ? ldw1<Q>(p.wbase, p.oW3 + t) : p.b3[t - 64 * od];
    }
    if (tid >= 256 && tid < 256 + TINY_RG * od) {
        const int t = tid - 256, row = r0 + t / od;
        ys[t] = row < n ? p.y[(size_t)row * od + t % od] : 0.f;
    }
    if (tid >= 320 && tid < 320 + TINY_RG) {
        const int row = r0 + tid - 320;
        sws[tid - 320] = (row < n && p.sw) ? p.sw[row] : 1.f;
    }
    for (int i = tid; i < TINY_RG * h1 / 4; i += 512) {
        const int r = (i * 4) / h1, k = (i * 4) - r * h1, row = r0 + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < n) {
            v = *(const f32x4*)(p.z1 + (size_t)row * h1 + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        *(f32x4*)(a_s + i * 4) = v;
    }
    if (tid < 64) {   // number of rows that count (sample weight != 0), over the WHOLE batch
        float c = (tid < n) ? (p.sw ? (p.sw[tid] != 0.f ? 1.f : 0.f) : 1.f) : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (tid == 0) cnts[0] = c;
    }
    __syncthreads();
    tiny_dense_core<Q>(a_s, h1, p.wbase, woff, 64, wv, red, z2s);
    // dZ1 below walks W2 in chunks of 128 rows staged through LDS (coalesced 16-B loads here, padded rows
    // read back per thread): the first chunk is fetched now, under the loss computation
    const bool w2_in_regs = h1 == 512;       // one 16-deep chunk per thread: wv still holds this thread's share of W2
    f32x4 wc[4];
    if (p.want_grads && !w2_in_regs) {
#pragma unroll
        for (int i = 0; i < 4; ++i) wc[i] = ldw4<Q>(p.wbase, p.oW2 + (size_t)(tid + 512 * i) * 4);
    }
    if (tid < TINY_RG * 64) {
        const int r = tid >> 6, c = tid & 63, row = r0 + r;
        const float v = qa<Q>(z2s[tid] + bias);
        z2s[tid] = v;
        if (row < n) p.z2[(size_t)row * 64 + c] = v;
    }
    __syncthreads();
    const float scale = p.grad_scale > 0.f ? p.grad_scale : 1.f / cnts[0];
    if (tid < TINY_RG * 16) {
        // layer 3 of row r: 16 lanes x 4 channels, xor-reduced; then the row's loss on lane 0
        const int r = tid >> 4, l = tid & 15, row = r0 + r;
        float z[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = l * 4 + i;
            const float v = fmaxf(z2s[r * 64 + c], 0.f);
            z[0] = fmaf(v, w3s[c * od + 0], z[0]);
            if (od == 2) z[1] = fmaf(v, w3s[c * od + 1], z[1]);
        }
        z[0] = row16_sum(z[0]);
        z[1] = row16_sum(z[1]);
        if (l == 0) {
            float li = 0.f, ai = 0.f, d3[2] = {0.f, 0.f};
            if (row < n) {
                const float w = sws[r];
                const float* b3s = w3s + 64 * od;
                if (od == 1) {
                    // Dense(1, sigmoid) + binary_crossentropy over one output (code/siamese3.py:25-28)
                    const float zz = z[0] + b3s[0];
                    const float pr = 1.f / (1.f + expf(-zz));
                    p.probs[row] = pr;
                    const float y = ys[r];
                    const float pc = fminf(fmaxf(pr, 1e-7f), 1.f - 1e-7f);
                    const float x = logf(pc / (1.f - pc));
                    li = (fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)))) * w;
                    ai = (rintf(pr) == y) ? 1.f : 0.f;
                    const bool inside = pr >= 1e-7f && pr <= 1.f - 1e-7f;
                    const float dp = inside ? w * scale * (pc - y) / (pc * (1.f - pc)) : 0.f;
                    d3[0] = qa<Q>(dp * pr * (1.f - pr));
                } else {
                    const float z0 = z[0] + b3s[0], z1v = z[1] + b3s[1];
                    const float m = fmaxf(z0, z1v);
                    const float e0 = expf(z0 - m), e1 = expf(z1v - m);
                    const float pr[2] = {e0 / (e0 + e1), e1 / (e0 + e1)};
                    p.probs[row * 2 + 0] = pr[0];
                    p.probs[row * 2 + 1] = pr[1];
                    float lsum = 0.f, acc = 0.f, dp[2];
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const float y = ys[r * 2 + c];
                        const float pc = fminf(fmaxf(pr[c], 1e-7f), 1.f - 1e-7f);
                        const float x = logf(pc / (1.f - pc));
                        lsum += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
                        acc += (rintf(pr[c]) == y) ? 1.f : 0.f;
                        const bool inside = pr[c] >= 1e-7f && pr[c] <= 1.f - 1e-7f;
                        dp[c] = inside ? 0.5f * w * scale * (pc - y) / (pc * (1.f - pc)) : 0.f;
                    }
                    li = 0.5f * lsum * w;
                    ai = 0.5f * acc;
                    const float dot = dp[0] * pr[0] + dp[1] * pr[1];
                    d3[0] = qa<Q>(pr[0] * (dp[0] - dot));
                    d3[1] = qa<Q>(pr[1] * (dp[1] - dot));
                }
                if (p.want_grads) {
                    p.dz3[row * od + 0] = d3[0];
                    if (od == 2) p.dz3[row * od + 1] = d3[1];
                }
            }
            dz3s[r * 2 + 0] = d3[0];
            dz3s[r * 2 + 1] = d3[1];
            rowm[r * 2 + 0] = li;
            rowm[r * 2 + 1] = ai;
        }
    }
    __syncthreads();
    float* part = p.part + (size_t)blockIdx.x * TINY_PART;
    if (tid == 0) {
        float ls = 0.f, as = 0.f;
#pragma unroll
        for (int r = 0; r < TINY_RG; ++r) { ls += rowm[r * 2]; as += rowm[r * 2 + 1]; }
        part[0] = ls;
        part[1] = as;
        part[132] = cnts[0];
    }
    if (!p.want_grads) {
        // test_on_batch: no third launch — the row group that finishes last adds the groups' partials in group order
        if (p.eval_metrics) {
            __shared__ int last;
            if (tid == 0) {
                __threadfence();
                last = atomicAdd(p.counter, 1u) == gridDim.x - 1 ? 1 : 0;
            }
            __syncthreads();
            if (last && tid == 0) {
                __threadfence();
                const volatile float* pv = p.part;
                float ls = 0.f, as = 0.f;
                for (unsigned g = 0; g < gridDim.x; ++g) { ls += pv[(size_t)g * TINY_PART]; as += pv[(size_t)g * TINY_PART + 1]; }
                p.eval_metrics[0] = ls * (p.grad_scale > 0.f ? p.grad_scale : 1.f / cnts[0]);
                p.eval_metrics[1] = as / (float)n;
                *p.counter = 0u;
            }
        }
        return;
    }
    float* dz2s = red;                             // [TINY_RG][64], the reduction scratch is free again
    if (tid < TINY_RG * 64) {
        // dZ2[row][c] = (z2 > 0) * sum_j dz3[row][j] * w3[c][j]
        const int r = tid >> 6, c = tid & 63, row = r0 + r;
        float g = dz3s[r * 2] * w3s[c * od];
        if (od == 2) g = fmaf(dz3s[r * 2 + 1], w3s[c * od + 1], g);
        g = z2s[tid] > 0.f ? qa<Q>(g) : 0.f;
        dz2s[tid] = g;
        if (row < n) p.dz2[(size_t)row * 64 + c] = g;
    } else if (tid < TINY_RG * 64 + 64 * od + od) {
        // this row group's share of dW3[c][j] = sum_r relu(z2[r][c]) dz3[r][j] and db3[j] (rows >= n carry dz3 = 0)
        const int t = tid - TINY_RG * 64;
        float s = 0.f;
        if (t < 64 * od) {
            const int c = t / od, j = t - c * od;
#pragma unroll
            for (int r = 0; r < TINY_RG; ++r) s = fmaf(fmaxf(z2s[r * 64 + c], 0.f), dz3s[r * 2 + j], s);
        } else {
            const int j = t - 64 * od;
#pragma unroll
            for (int r = 0; r < TINY_RG; ++r) s += dz3s[r * 2 + j];
        }
        part[2 + t] = s;
    }
    __syncthreads();
    // dZ1[row][k] = (z1 > 0) * sum_j dZ2[row][j] W2[k][j].
    if (w2_in_regs) {
        // The forward left W2[16 ks + i][4 cl .. 4 cl + 3] (i < 16) in this thread's registers: its four columns'
        // share of the dot for its 16 rows k, summed over the 16 column lanes (one DPP row) by row16_sum — no
        // second pass over W2 and no LDS traffic beyond the 4 x 16 B of dZ2.
        const int cl = tid & 15, ks = tid >> 4;
#pragma unroll
        for (int r = 0; r < TINY_RG; ++r) {
            const f32x4 dv = *(const f32x4*)(dz2s + r * 64 + cl * 4);
            float mine = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float sdot = dv[0] * wv[i][0];
                sdot = fmaf(dv[1], wv[i][1], sdot);
                sdot = fmaf(dv[2], wv[i][2], sdot);
                sdot = fmaf(dv[3], wv[i][3], sdot);
                sdot = row16_sum(sdot);
                if (cl == i) mine = sdot;         // lane i of the row keeps (and stores) W2 row 16 ks + i
            }
            const int k = ks * 16 + cl;
            if (r0 + r < n) p.dz1[(size_t)(r0 + r) * h1 + k] = a_s[r * h1 + k] > 0.f ? qa<Q>(mine) : 0.f;
        }
        return;
    }
    // general h1: chunks of 128 W2 rows in LDS with a row pitch of 68 floats (16-B reads of 16 consecutive rows
    // then fall on distinct banks); thread = (k, row).
    float* w2s = a_s + TINY_RG * h1 + 8 * TINY_RG * 64 + TINY_RG * 64 + 192;     // [128][68], after everything else
    const int kq = tid & 127, rq = tid >> 7;
    for (int kc = 0; kc < h1; kc += 128) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = (tid + 512 * i) * 4;                                    // element of the 128 x 64 chunk
            *(f32x4*)(w2s + (e >> 6) * 68 + (e & 63)) = wc[i];
        }
        __syncthreads();
        if (kc + 128 < h1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wc[i] = ldw4<Q>(p.wbase, p.oW2 + (size_t)(kc + 128) * 64 + (size_t)(tid + 512 * i) * 4);
        }
        float sdot = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const f32x4 wv4 = *(const f32x4*)(w2s + kq * 68 + i * 4);
            const f32x4 dv = *(const f32x4*)(dz2s + rq * 64 + i * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) sdot = fmaf(dv[j], wv4[j], sdot);
        }
        if (r0 + rq < n) p.dz1[(size_t)(r0 + rq) * h1 + kc + kq] = a_s[rq * h1 + kc + kq] > 0.f ? qa<Q>(sdot) : 0.f;
        __syncthreads();
    }
}

struct TinyBwd {
    const float *z1, *dz1, *dz2, *dm, *part, *sw;
    float *prm, *g, *a, *d, *metrics;
    __bf16* pq;               // bf16 copy of the parameters, refreshed with every update (nullptr in f32 mode)
    size_t oW1, ob1, oW2, ob2, oW3, ob3;
    int n, D, h1, od, apply, ngroups;
    float lr, rho, eps, grad_scale;
};

__device__ __forceinline__ float tiny_adadelta(float p, float g, float& a, float& d, float lr, float rho, float eps) {
    const float na = rho * a + (1.f - rho) * g * g;
    const float u = g * sqrtf(d + eps) / sqrtf(na + eps);
    d = rho * d + (1.f - rho) * u * u;
    a = na;
    return p - lr * u;
}
// Adadelta on 4 consecutive parameters whose old values were loaded up front
__device__ __forceinline__ void tiny_update4(const TinyBwd& p, size_t i, const f32x4& g, f32x4 w, f32x4 a, f32x4 d) {
    *(f32x4*)(p.g + i) = g;
    if (p.apply) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float aj = a[j], dj = d[j];
            w[j] = tiny_adadelta(w[j], g[j], aj, dj, p.lr, p.rho, p.eps);
            a[j] = aj;
            d[j] = dj;
        }
        *(f32x4*)(p.prm + i) = w;
        *(f32x4*)(p.a + i) = a;
        *(f32x4*)(p.d + i) = d;
        if (p.pq) *(bf16x4*)(p.pq + i) = bf16x4{(__bf16)w[0], (__bf16)w[1], (__bf16)w[2], (__bf16)w[3]};
    }
}

// grid (h1 / 64, D / TINY_KB + 2), 256 threads; h2 == 64, n <= TINY_N.  Roles by blockIdx.y:
//   y <  D / TINY_KB : dW1 tile (64 columns x TINY_KB k rows) + update
//   y == D / TINY_KB : db1 and dW2 rows of column block x + update (k index of W2 = column of layer 1)
//   y == last, x == 0: db2, dW3, db3 + update, and the metrics, from the row groups' partials
// so every block's chain is load -> one short reduction -> store, and they all run side by side.
__global__ __launch_bounds__(256) void tiny_wgrad_update_kernel(const TinyBwd p) {
    __shared__ __attribute__((aligned(16))) float sa[TINY_N * 64];
    __shared__ __attribute__((aligned(16))) float sb[TINY_N * 64];
    __shared__ float acc_sum;
    const int tid = threadIdx.x, n = p.n, D = p.D, h1 = p.h1;
    const int c0 = blockIdx.x * 64, nkp = D / TINY_KB;
    const int cl = tid & 15, kk = tid >> 4;
    if ((int)blockIdx.y < nkp) {
        // ---- dW1[kb + 2 kk + q][c0 + 4 cl + j] = sum_r dm[r][k] dZ1[r][c] -----------------------------------------
        const int kb = blockIdx.y * TINY_KB;
        float* dz1s = sa;                          // [n][64]
        float* dms = sb;                           // [n][TINY_KB]
        f32x4 pw[2], pa[2], pd[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const size_t i = p.oW1 + (size_t)(kb + kk * 2 + q) * h1 + c0 + cl * 4;
            if (p.apply) {
                pw[q] = *(const f32x4*)(p.prm + i);
                pa[q] = *(const f32x4*)(p.a + i);
                pd[q] = *(const f32x4*)(p.d + i);
            }
        }
        for (int i = tid; i < n * 16; i += 256) {
            const int r = i >> 4, j = (i & 15) * 4;
            *(f32x4*)(dz1s + r * 64 + j) = *(const f32x4*)(p.dz1 + (size_t)r * h1 + c0 + j);
        }
        for (int i = tid; i < n * (TINY_KB / 4); i += 256) {
            const int r = i / (TINY_KB / 4), j = (i % (TINY_KB / 4)) * 4;
            *(f32x4*)(dms + r * TINY_KB + j) = *(const f32x4*)(p.dm + (size_t)r * D + kb + j);
        }
        __syncthreads();
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        for (int r = 0; r < n; ++r) {
            const f32x4 dv = *(const f32x4*)(dz1s + r * 64 + cl * 4);
            const float m0 = dms[r * TINY_KB + kk * 2], m1 = dms[r * TINY_KB + kk * 2 + 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][j] = fmaf(m0, dv[j], acc[0][j]);
                acc[1][j] = fmaf(m1, dv[j], acc[1][j]);
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
            tiny_update4(p, p.oW1 + (size_t)(kb + kk * 2 + q) * h1 + c0 + cl * 4, acc[q], pw[q], pa[q], pd[q]);
        return;
    }
    if ((int)blockIdx.y == nkp) {
        // ---- db1[c0 ..] and dW2[c0 + 4 kk + q][4 cl + j] = sum_r relu(z1[r][c0 + 4 kk + q]) dZ2[r][4 cl + j] -----
        float* dz2s = sa;                          // [n][64]
        float* z1s = sb;                           // [n][64]
        f32x4 w2w[4], w2a[4], w2d[4];
        float b1w = 0.f, b1a = 0.f, b1d = 0.f;
        if (p.apply) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t i = p.oW2 + (size_t)(c0 + kk * 4 + q) * 64 + cl * 4;
                w2w[q] = *(const f32x4*)(p.prm + i);
                w2a[q] = *(const f32x4*)(p.a + i);
                w2d[q] = *(const f32x4*)(p.d + i);
            }
            if (tid < 64) { b1w = p.prm[p.ob1 + c0 + tid]; b1a = p.a[p.ob1 + c0 + tid]; b1d = p.d[p.ob1 + c0 + tid]; }
        }
        float gb1 = 0.f;                            // db1 straight from global: column c0 + tid of dZ1
        if (tid < 64) {
            float v[TINY_N];
#pragma unroll
            for (int r = 0; r < TINY_N; ++r) v[r] = r < n ? p.dz1[(size_t)r * h1 + c0 + tid] : 0.f;
#pragma unroll
            for (int r = 0; r < TINY_N; ++r) gb1 += v[r];
        }
        for (int i = tid; i < n * 16; i += 256) {
            const int r = i >> 4, j = (i & 15) * 4;
            *(f32x4*)(dz2s + r * 64 + j) = *(const f32x4*)(p.dz2 + (size_t)r * 64 + j);
            *(f32x4*)(z1s + r * 64 + j) = *(const f32x4*)(p.z1 + (size_t)r * h1 + c0 + j);
        }
        __syncthreads();
        if (tid < 64) {
            p.g[p.ob1 + c0 + tid] = gb1;
            if (p.apply) {
                const float nb1 = tiny_adadelta(b1w, gb1, b1a, b1d, p.lr, p.rho, p.eps);
                p.prm[p.ob1 + c0 + tid] = nb1;
                if (p.pq) p.pq[p.ob1 + c0 + tid] = (__bf16)nb1;
                p.a[p.ob1 + c0 + tid] = b1a;
                p.d[p.ob1 + c0 + tid] = b1d;
            }
        }
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < n; ++r) {
            const f32x4 dv = *(const f32x4*)(dz2s + r * 64 + cl * 4);
            const f32x4 zv = *(const f32x4*)(z1s + r * 64 + kk * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float a = fmaxf(zv[q], 0.f);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[q][j] = fmaf(a, dv[j], acc[q][j]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            tiny_update4(p, p.oW2 + (size_t)(c0 + kk * 4 + q) * 64 + cl * 4, acc[q], w2w[q], w2a[q], w2d[q]);
        return;
    }
    if (blockIdx.x != 0) return;
    // ---- db2 (threads 0..63), dW3 / db3 (threads 64..) and the metrics (thread 0), fixed summation order -----------
    const int od = p.od;
    const bool is_b2 = tid < 64, is_w3 = tid >= 64 && tid < 64 + 64 * od + od;
    const size_t pi = is_b2 ? p.ob2 + tid : p.oW3 + (tid - 64);      // oW3 .. ob3 + od - 1 are contiguous
    float w_ = 0.f, a_ = 0.f, d_ = 0.f;
    if ((is_b2 || is_w3) && p.apply) { w_ = p.prm[pi]; a_ = p.a[pi]; d_ = p.d[pi]; }
    float partv[TINY_N / TINY_RG], lossv[TINY_N / TINY_RG];
    const float cnt = tid == 0 ? p.part[132] : 1.f;
#pragma unroll
    for (int gI = 0; gI < TINY_N / TINY_RG; ++gI) {
        partv[gI] = (is_w3 && gI < p.ngroups) ? p.part[(size_t)gI * TINY_PART + 2 + (tid - 64)] : 0.f;
        lossv[gI] = (tid < 2 && gI < p.ngroups) ? p.part[(size_t)gI * TINY_PART + tid] : 0.f;
    }
    float s = 0.f;
    if (is_b2) {
        float v[TINY_N];
#pragma unroll
        for (int r = 0; r < TINY_N; ++r) v[r] = r < n ? p.dz2[(size_t)r * 64 + tid] : 0.f;
#pragma unroll
        for (int r = 0; r < TINY_N; ++r) s += v[r];
    } else {
#pragma unroll
        for (int gI = 0; gI < TINY_N / TINY_RG; ++gI) s += partv[gI];
    }
    float ls = 0.f;
#pragma unroll
    for (int gI = 0; gI < TINY_N / TINY_RG; ++gI) ls += lossv[gI];
    if (tid == 1) acc_sum = ls;
    if (is_b2 || is_w3) {
        p.g[pi] = s;
        if (p.apply) {
            const float nw = tiny_adadelta(w_, s, a_, d_, p.lr, p.rho, p.eps);
            p.prm[pi] = nw;
            if (p.pq) p.pq[pi] = (__bf16)nw;
            p.a[pi] = a_;
            p.d[pi] = d_;
        }
    }
    __syncthreads();
    if (tid == 0) {
        p.metrics[0] = ls * (p.grad_scale > 0.f ? p.grad_scale : 1.f / cnt);
        p.metrics[1] = acc_sum / (float)n;
    }
}

size_t tiny_lds_bytes(int K, bool with_w2 = false) {
    return (size_t)(TINY_RG * K + 8 * TINY_RG * 64 + TINY_RG * 64 + 192 + (with_w2 ? 128 * 68 : 0)) * sizeof(float);
}

bool tiny_ok(const alink_head* h, int n) {
    return g_use_tiny && n <= TINY_N && h->h2 == 64 && h->h1 % 512 == 0 && h->D % 512 == 0 && h->D <= 2048 &&
           h->h1 <= 1024 && h->oW1 == 0 && (!h->qmode || h->h1 == 512);     // LDS: 43 KB (A at D = 2048), 61 KB (B at h1 = 1024) of the 64 KB default
}

// the three launches; gradients are always left in d_grads, parameters updated when `apply`
int tiny_train(alink_head* h, const float* L, const float* R, const float* y, const float* sw, int n,
               float grad_scale, bool apply, float* metrics, hipStream_t st, const int* li = nullptr, const int* ri = nullptr) {
    const int D = h->D, h1 = h->h1;
    float* P = h->d_params;
    const bool Q = h->qmode != 0;
    if (Q) { const int rc = ensure_q(h, st, false); if (rc) return rc; }
    const void* WB = Q ? (const void*)h->d_pq : (const void*)P;      // where the WEIGHTS are read from (oW1 == 0)
    const int ngroups = (n + TINY_RG - 1) / TINY_RG;
    if (Q) hipLaunchKernelGGL(tiny_dense1_kernel<true>, dim3(h1 / 64, ngroups), dim3(512), tiny_lds_bytes(D), st, L, R, WB,
                              P + h->ob1, h->d_z1, h->d_dm, n, D, h1, li, ri);
    else   hipLaunchKernelGGL(tiny_dense1_kernel<false>, dim3(h1 / 64, ngroups), dim3(512), tiny_lds_bytes(D), st, L, R, WB,
                              P + h->ob1, h->d_z1, h->d_dm, n, D, h1, li, ri);
    TinyLoss lp{};
    lp.wbase = WB; lp.oW2 = h->oW2; lp.oW3 = h->oW3;
    lp.z1 = h->d_z1; lp.b2 = P + h->ob2; lp.b3 = P + h->ob3; lp.y = y;
    lp.sw = sw; lp.z2 = h->d_z2; lp.probs = h->d_p; lp.dz2 = h->d_dz2; lp.dz3 = h->d_dz3; lp.dz1 = h->d_dz1;
    lp.part = h->d_tiny; lp.n = n; lp.h1 = h1; lp.od = h->od; lp.want_grads = 1; lp.grad_scale = grad_scale;
    if (Q) hipLaunchKernelGGL(tiny_dense2_loss_kernel<true>, dim3(ngroups), dim3(512), tiny_lds_bytes(h1, true), st, lp);
    else   hipLaunchKernelGGL(tiny_dense2_loss_kernel<false>, dim3(ngroups), dim3(512), tiny_lds_bytes(h1, true), st, lp);
    TinyBwd bp{};
    bp.z1 = h->d_z1; bp.dz1 = h->d_dz1; bp.dz2 = h->d_dz2; bp.dm = h->d_dm; bp.part = h->d_tiny; bp.sw = sw;
    bp.prm = P; bp.g = h->d_grads; bp.a = h->d_acc; bp.d = h->d_dacc; bp.metrics = metrics;
    bp.pq = Q ? h->d_pq : nullptr;
    bp.oW1 = h->oW1; bp.ob1 = h->ob1; bp.oW2 = h->oW2; bp.ob2 = h->ob2; bp.oW3 = h->oW3; bp.ob3 = h->ob3;
    bp.n = n; bp.D = D; bp.h1 = h1; bp.od = h->od; bp.apply = apply ? 1 : 0; bp.ngroups = ngroups;
    bp.lr = h->lr; bp.rho = h->rho; bp.eps = h->eps; bp.grad_scale = grad_scale;
    hipLaunchKernelGGL(tiny_wgrad_update_kernel, dim3(h1 / 64, D / TINY_KB + 2), dim3(256), 0, st, bp);
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

// test_on_batch on n <= TINY_N rows: launches A and B of the step above (no gradients), metrics from B's last row group
int tiny_eval(alink_head* h, const float* L, const float* R, const float* y, int n, float* metrics, hipStream_t st,
              const int* li = nullptr, const int* ri = nullptr) {
    const int D = h->D, h1 = h->h1;
    float* P = h->d_params;
    const bool Q = h->qmode != 0;
    if (Q) { const int rc = ensure_q(h, st, false); if (rc) return rc; }
    const void* WB = Q ? (const void*)h->d_pq : (const void*)P;
    const int ngroups = (n + TINY_RG - 1) / TINY_RG;
    if (Q) hipLaunchKernelGGL(tiny_dense1_kernel<true>, dim3(h1 / 64, ngroups), dim3(512), tiny_lds_bytes(D), st, L, R, WB,
                              P + h->ob1, h->d_z1, h->d_dm, n, D, h1, li, ri);
    else   hipLaunchKernelGGL(tiny_dense1_kernel<false>, dim3(h1 / 64, ngroups), dim3(512), tiny_lds_bytes(D), st, L, R, WB,
                              P + h->ob1, h->d_z1, h->d_dm, n, D, h1, li, ri);
    TinyLoss lp{};
    lp.wbase = WB; lp.oW2 = h->oW2; lp.oW3 = h->oW3;
    lp.z1 = h->d_z1; lp.b2 = P + h->ob2; lp.b3 = P + h->ob3; lp.y = y;
    lp.sw = nullptr; lp.z2 = h->d_z2; lp.probs = h->d_p; lp.dz2 = h->d_dz2; lp.dz3 = h->d_dz3; lp.dz1 = h->d_dz1;
    lp.part = h->d_tiny; lp.n = n; lp.h1 = h1; lp.od = h->od; lp.want_grads = 0; lp.grad_scale = 0.f;
    lp.eval_metrics = metrics; lp.counter = h->d_counter;
    if (Q) hipLaunchKernelGGL(tiny_dense2_loss_kernel<true>, dim3(ngroups), dim3(512), tiny_lds_bytes(h1, true), st, lp);
    else   hipLaunchKernelGGL(tiny_dense2_loss_kernel<false>, dim3(ngroups), dim3(512), tiny_lds_bytes(h1, true), st, lp);
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

// --------------- SmallRes' head: train step + input gradients in three launches (round 6) -------------------
// SmallRes (code/siamese.py:156-168) ends in |l - r| -> Dense(128) -> Dense(32) -> Dense(2) on 2048 features at batch 16.  On
// the generic chain above that is seven dependent launches (two forwards, loss, middle, two weight gradients' worth, input
// gradients: 57 us of a 339 us step, every one of them a few workgroups waiting on memory).  The dependencies allow three:
//   A  mini_dense1_kernel   one workgroup per 32 inputs k: partial Dense1 sums of every row over its W1 rows (one round trip)
//   B  mini_mid_kernel      ONE workgroup: adds the partials, Dense2, Dense3, softmax, BCE, metrics, dZ3, dZ2, dZ1, dW3, dW2
//                           (everything whose operands fit in LDS: 16 x 128 activations, the 16 KB of W2)
//   C  mini_wgrad1_kernel   one workgroup per 32 inputs k again: its rows of dW1 and its columns of dL / dR (the same W1 rows)
// Sums run in a fixed order (k ascending inside a slice, slices in eight interleaved groups; rows / columns ascending as in
// the generic kernels), so a step is reproducible; against the generic chain the Dense1 / Dense2 sums differ by rounding.
constexpr int MINI_N = 32;       // rows (pairs) the path takes
constexpr int MINI_KS = 32;      // inputs per workgroup in A and C
bool g_use_mini = true;

template <int NJ>                // rows per thread: 8 (n <= 16) or 16
__global__ __launch_bounds__(256) void mini_dense1_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                          const float* __restrict__ W1, float* __restrict__ part, int n, int D) {
    __shared__ __attribute__((aligned(16))) float Ws[MINI_KS * 128];
    __shared__ __attribute__((aligned(16))) float dms[2 * NJ * MINI_KS];
    const int tid = threadIdx.x, k0 = blockIdx.x * MINI_KS;
    f32x4 wv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wv[j] = *(const f32x4*)(W1 + (size_t)k0 * 128 + (size_t)(tid + 256 * j) * 4);     // 32 rows of W1 are contiguous
    const int row = tid >> 3, k4 = (tid & 7) * 4;
    f32x4 lv = {0.f, 0.f, 0.f, 0.f}, rv = {0.f, 0.f, 0.f, 0.f};
    if (row < n) {
        lv = *(const f32x4*)(L + (size_t)row * D + k0 + k4);
        rv = *(const f32x4*)(R + (size_t)row * D + k0 + k4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) *(f32x4*)(Ws + (tid + 256 * j) * 4) = wv[j];
    if (row < 2 * NJ) {
        f32x4 d;
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = fabsf(lv[i] - rv[i]);
        *(f32x4*)(dms + row * MINI_KS + k4) = d;
    }
    __syncthreads();
    const int c = tid & 127, half = tid >> 7;         // thread: column c, rows half, half + 2, ...
    float acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = 0.f;
#pragma unroll 2
    for (int k = 0; k < MINI_KS; k += 4) {
        const float w0 = Ws[k * 128 + c], w1 = Ws[(k + 1) * 128 + c], w2 = Ws[(k + 2) * 128 + c], w3 = Ws[(k + 3) * 128 + c];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const f32x4 d = *(const f32x4*)(dms + (half + 2 * j) * MINI_KS + k);
            acc[j] = fmaf(d[0], w0, acc[j]);
            acc[j] = fmaf(d[1], w1, acc[j]);
            acc[j] = fmaf(d[2], w2, acc[j]);
            acc[j] = fmaf(d[3], w3, acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int r = half + 2 * j;
        if (r < n) part[((size_t)blockIdx.x * n + r) * 128 + c] = acc[j];
    }
}

struct MiniMid {
    const float *part, *b1, *W2, *b2, *W3, *b3, *y, *sw;
    float *gW2, *gb2, *gW3, *gb3, *dz1, *probs, *metrics;
    int n, S;                // rows; slices of A (a multiple of 8)
    float grad_scale;        // <= 0: 1 / count(sw != 0)
};

__global__ __launch_bounds__(1024) void mini_mid_kernel(const MiniMid p) {
    __shared__ __attribute__((aligned(16))) float z1s[MINI_N * 128];
    __shared__ __attribute__((aligned(16))) float W2s[128 * 36];      // [k][c2], pitch 36: 16-byte rows for the dZ1 phase
    __shared__ __attribute__((aligned(16))) float W2T[32 * 132];      // [c2][k], pitch 132: 16-byte reads along k for Dense2
    __shared__ __attribute__((aligned(16))) float z2s[MINI_N * 32], dz2s[MINI_N * 32];
    __shared__ __attribute__((aligned(16))) float w3s[68];
    __shared__ float dz3s[MINI_N * 2], rowm[MINI_N * 2], b2s[32], sws[MINI_N], ys[MINI_N * 2];
    const int tid = threadIdx.x, n = p.n, S = p.S;
    const int nz1 = n * 128;
    // ---- every global load of the forward first: W2 and the small vectors ride under the partial sums
    const f32x4 w2v = *(const f32x4*)(p.W2 + tid * 4);
    float small = 0.f;
    if (tid < 64) small = p.W3[tid];
    else if (tid < 66) small = p.b3[tid - 64];
    else if (tid >= 128 && tid < 160) small = p.b2[tid - 128];
    else if (tid >= 192 && tid < 192 + n) small = p.sw ? p.sw[tid - 192] : 1.f;
    else if (tid >= 256 && tid < 256 + 2 * n) small = p.y[tid - 256];
    // z1[r][c] = b1[c] + sum over slices: eight interleaved groups (slices g, g + 8, ... ascending), then the groups pairwise.
    // Two outputs at a time, 16 slices of each per batch of loads: 32 loads in flight per thread (one output and 8 slices at a
    // time was 16 dependent round trips, most of this kernel's 14 us).
    for (int o = tid; o < nz1; o += 2048) {
        const bool two = o + 1024 < nz1;
        const float* pa = p.part + o;
        const float* pb = p.part + (two ? o + 1024 : o);
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int s0 = 0;
        for (; s0 + 16 <= S; s0 += 16) {
            float ta[16], tb[16];
#pragma unroll
            for (int g = 0; g < 16; ++g) { ta[g] = pa[(size_t)(s0 + g) * nz1]; tb[g] = pb[(size_t)(s0 + g) * nz1]; }
#pragma unroll
            for (int g = 0; g < 16; ++g) { a[g & 7] += ta[g]; b[g & 7] += tb[g]; }
        }
        if (s0 < S) {                              // (S is a multiple of 8)
            float ta[8], tb[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) { ta[g] = pa[(size_t)(s0 + g) * nz1]; tb[g] = pb[(size_t)(s0 + g) * nz1]; }
#pragma unroll
            for (int g = 0; g < 8; ++g) { a[g] += ta[g]; b[g] += tb[g]; }
        }
        z1s[o] = (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) + p.b1[o & 127];
        if (two) z1s[o + 1024] = (((b[0] + b[1]) + (b[2] + b[3])) + ((b[4] + b[5]) + (b[6] + b[7]))) + p.b1[o & 127];
    }
    {
        const int k = tid >> 3, c = (tid & 7) * 4;
        *(f32x4*)(W2s + k * 36 + c) = w2v;
        W2T[(c + 0) * 132 + k] = w2v[0]; W2T[(c + 1) * 132 + k] = w2v[1]; W2T[(c + 2) * 132 + k] = w2v[2]; W2T[(c + 3) * 132 + k] = w2v[3];
    }
    if (tid < 66) w3s[tid] = small;
    else if (tid >= 128 && tid < 160) b2s[tid - 128] = small;
    else if (tid >= 192 && tid < 192 + n) sws[tid - 192] = small;
    else if (tid >= 256 && tid < 256 + 2 * n) ys[tid - 256] = small;
    __syncthreads();
    // ---- Dense2 on relu(z1): one thread per (row, column), k ascending
    if (tid < n * 32) {
        const int r = tid >> 5, c2 = tid & 31;
        float s = 0.f;
        // (16-byte LDS reads: one per four steps from each operand — with 4-byte reads this phase was 4.2 of the kernel's 14 us)
#pragma unroll 4
        for (int k = 0; k < 128; k += 4) {
            const f32x4 a = *(const f32x4*)(z1s + r * 128 + k), w = *(const f32x4*)(W2T + c2 * 132 + k);
            s = fmaf(fmaxf(a[0], 0.f), w[0], s);
            s = fmaf(fmaxf(a[1], 0.f), w[1], s);
            s = fmaf(fmaxf(a[2], 0.f), w[2], s);
            s = fmaf(fmaxf(a[3], 0.f), w[3], s);
        }
        z2s[tid] = s + b2s[c2];
    }
    __syncthreads();
    // ---- per row: Dense3, softmax, Keras binary_crossentropy on the clipped probabilities, accuracy, dZ3 (head_loss_kernel's arithmetic)
    float cnt = 0.f;
    for (int i = 0; i < n; ++i) cnt += sws[i] != 0.f ? 1.f : 0.f;
    const float scale = p.grad_scale > 0.f ? p.grad_scale : 1.f / cnt;
    if (tid < n) {
        const float* a = z2s + tid * 32;
        const float w = sws[tid];
        float z0 = 0.f, z1 = 0.f;
#pragma unroll
        for (int c = 0; c < 32; c += 4) {
            const f32x4 av = *(const f32x4*)(a + c), wa = *(const f32x4*)(w3s + c * 2), wb = *(const f32x4*)(w3s + c * 2 + 4);
            z0 = fmaf(fmaxf(av[0], 0.f), wa[0], z0); z1 = fmaf(fmaxf(av[0], 0.f), wa[1], z1);
            z0 = fmaf(fmaxf(av[1], 0.f), wa[2], z0); z1 = fmaf(fmaxf(av[1], 0.f), wa[3], z1);
            z0 = fmaf(fmaxf(av[2], 0.f), wb[0], z0); z1 = fmaf(fmaxf(av[2], 0.f), wb[1], z1);
            z0 = fmaf(fmaxf(av[3], 0.f), wb[2], z0); z1 = fmaf(fmaxf(av[3], 0.f), wb[3], z1);
        }
        z0 += w3s[64];
        z1 += w3s[65];
        const float m = fmaxf(z0, z1);
        const float e0 = expf(z0 - m), e1 = expf(z1 - m);
        const float pr[2] = {e0 / (e0 + e1), e1 / (e0 + e1)};
        p.probs[tid * 2 + 0] = pr[0];
        p.probs[tid * 2 + 1] = pr[1];
        float li = 0.f, acc = 0.f, dp[2];
        for (int c = 0; c < 2; ++c) {
            const float y = ys[tid * 2 + c];
            const float pc = fminf(fmaxf(pr[c], 1e-7f), 1.f - 1e-7f);
            const float x = logf(pc / (1.f - pc));
            li += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
            acc += (rintf(pr[c]) == y) ? 1.f : 0.f;
            const bool inside = pr[c] >= 1e-7f && pr[c] <= 1.f - 1e-7f;
            dp[c] = inside ? 0.5f * w * scale * (pc - y) / (pc * (1.f - pc)) : 0.f;
        }
        rowm[tid * 2 + 0] = 0.5f * li * w;
        rowm[tid * 2 + 1] = 0.5f * acc;
        const float dot = dp[0] * pr[0] + dp[1] * pr[1];
        dz3s[tid * 2 + 0] = pr[0] * (dp[0] - dot);
        dz3s[tid * 2 + 1] = pr[1] * (dp[1] - dot);
    }
    __syncthreads();
    // ---- metrics (rows ascending), dW3 / db3, dZ2
    if (tid == 1023) {
        float ls = 0.f, as = 0.f;
        for (int i = 0; i < n; ++i) { ls += rowm[i * 2]; as += rowm[i * 2 + 1]; }
        p.metrics[0] = ls * scale;
        p.metrics[1] = as / (float)n;
    }
    if (tid >= 896 && tid < 896 + 66) {
        const int t = tid - 896;
        float s = 0.f;
        if (t < 64) {
            const int c = t >> 1, j = t & 1;
            for (int i = 0; i < n; ++i) s = fmaf(fmaxf(z2s[i * 32 + c], 0.f), dz3s[i * 2 + j], s);
            p.gW3[t] = s;
        } else {
            const int j = t - 64;
            for (int i = 0; i < n; ++i) s += dz3s[i * 2 + j];
            p.gb3[j] = s;
        }
    }
    if (tid < n * 32) {
        const int i = tid >> 5, c = tid & 31;
        float g = 0.f;
        g = fmaf(dz3s[i * 2 + 0], w3s[c * 2 + 0], g);
        g = fmaf(dz3s[i * 2 + 1], w3s[c * 2 + 1], g);
        dz2s[tid] = z2s[tid] > 0.f ? g : 0.f;
    }
    __syncthreads();
    // ---- dW2[k][c] = sum_i relu(z1[i][k]) dZ2[i][c], db2; dZ1[r][k] = (z1 > 0) sum_c dZ2[r][c] W2[k][c]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int o = tid + 1024 * j, k = o >> 5, c2 = o & 31;
        float s = 0.f;
        for (int i = 0; i < n; ++i) s = fmaf(fmaxf(z1s[i * 128 + k], 0.f), dz2s[i * 32 + c2], s);
        p.gW2[o] = s;
    }
    if (tid < 32) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += dz2s[i * 32 + tid];
        p.gb2[tid] = s;
    }
    for (int o = tid; o < nz1; o += 1024) {
        const int r = o >> 7, k = o & 127;
        float g = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < 32; c2 += 4) {
            const f32x4 dv = *(const f32x4*)(dz2s + r * 32 + c2), w = *(const f32x4*)(W2s + k * 36 + c2);
            g = fmaf(dv[0], w[0], g);
            g = fmaf(dv[1], w[1], g);
            g = fmaf(dv[2], w[2], g);
            g = fmaf(dv[3], w[3], g);
        }
        p.dz1[o] = z1s[o] > 0.f ? g : 0.f;
    }
}

// C: rows k0 .. k0 + 31 of dW1 (= |l - r|^T dZ1, rows ascending) and columns k0 .. k0 + 31 of dL / dR (= +-sign(l - r) dZ1 W1^T,
// c ascending; relu_in: times (input > 0), see head_input_grad_kernel); workgroup 0 also db1
template <int NI>                // row passes of the input gradients: 2 (n <= 16) or 4
__global__ __launch_bounds__(256) void mini_wgrad1_kernel(const float* __restrict__ L, const float* __restrict__ R,
                                                          const float* __restrict__ W1, const float* __restrict__ dz1,
                                                          float* __restrict__ gW1, float* __restrict__ gb1, float* __restrict__ dL,
                                                          float* __restrict__ dR, float* __restrict__ colsum, int n, int D, int relu_in) {
    __shared__ float Ws[MINI_KS * 129];
    __shared__ __attribute__((aligned(16))) float dz1s[8 * NI * 128];
    __shared__ __attribute__((aligned(16))) float ls[8 * NI * MINI_KS], rs[8 * NI * MINI_KS], dms[8 * NI * MINI_KS];
    const int tid = threadIdx.x, k0 = blockIdx.x * MINI_KS;
    f32x4 wv[4], dv[NI];
#pragma unroll
    for (int j = 0; j < 4; ++j) wv[j] = *(const f32x4*)(W1 + (size_t)k0 * 128 + (size_t)(tid + 256 * j) * 4);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int i4 = (tid + 256 * j) * 4;                   // dz1 is [n][128]: rows >= n read as zeros
        dv[j] = i4 < n * 128 ? *(const f32x4*)(dz1 + i4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int row = tid >> 3, k4 = (tid & 7) * 4;
    f32x4 lv = {0.f, 0.f, 0.f, 0.f}, rv = {0.f, 0.f, 0.f, 0.f};
    if (row < n) {
        lv = *(const f32x4*)(L + (size_t)row * D + k0 + k4);
        rv = *(const f32x4*)(R + (size_t)row * D + k0 + k4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = (tid + 256 * j) * 4, k = i >> 7, c = i & 127;
        float* d = Ws + k * 129 + c;
        d[0] = wv[j][0]; d[1] = wv[j][1]; d[2] = wv[j][2]; d[3] = wv[j][3];
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) *(f32x4*)(dz1s + (tid + 256 * j) * 4) = dv[j];
    if (row < 8 * NI) {
        f32x4 d;
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = fabsf(lv[i] - rv[i]);
        *(f32x4*)(dms + row * MINI_KS + k4) = d;
        *(f32x4*)(ls + row * MINI_KS + k4) = lv;
        *(f32x4*)(rs + row * MINI_KS + k4) = rv;
    }
    __syncthreads();
    {   // dW1: thread = column c, rows kh * 16 .. + 15 of the slice
        const int c = tid & 127, kh = tid >> 7;
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        for (int r = 0; r < n; ++r) {
            const float d = dz1s[r * 128 + c];
#pragma unroll
            for (int j4 = 0; j4 < 16; j4 += 4) {
                const f32x4 a = *(const f32x4*)(dms + r * MINI_KS + kh * 16 + j4);
                acc[j4 + 0] = fmaf(a[0], d, acc[j4 + 0]);
                acc[j4 + 1] = fmaf(a[1], d, acc[j4 + 1]);
                acc[j4 + 2] = fmaf(a[2], d, acc[j4 + 2]);
                acc[j4 + 3] = fmaf(a[3], d, acc[j4 + 3]);
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) gW1[(size_t)(k0 + kh * 16 + j) * 128 + c] = acc[j];
        if (blockIdx.x == 0 && tid < 128) {
            float s = 0.f;
            for (int r = 0; r < n; ++r) s += dz1s[r * 128 + tid];
            gb1[tid] = s;
        }
    }
    {   // input gradients: thread = input k, rows rg, rg + 8, ...
        const int k = tid & 31, rg = tid >> 5;
        float acc[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i] = 0.f;
        for (int c = 0; c < 128; ++c) {
            const float w = Ws[k * 129 + c];
#pragma unroll
            for (int i = 0; i < NI; ++i) acc[i] = fmaf(dz1s[(rg + 8 * i) * 128 + c], w, acc[i]);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int r = rg + 8 * i;
            if (r >= n) continue;
            const float l = ls[r * MINI_KS + k], r_ = rs[r * MINI_KS + k], d = l - r_;
            const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            const float gl = acc[i] * sg, gr = -acc[i] * sg;
            const size_t o = (size_t)r * D + k0 + k;
            const float vl = (relu_in && !(l > 0.f)) ? 0.f : gl, vr = (relu_in && !(r_ > 0.f)) ? 0.f : gr;
            dL[o] = vl;
            dR[o] = vr;
            ls[r * MINI_KS + k] = vl;              // (this thread's own slot: nobody else reads l, r of (r, k))
            rs[r * MINI_KS + k] = vr;
        }
    }
    if (!colsum) return;                           // (uniform)
    __syncthreads();
    if (tid < MINI_KS) {                           // column sums over the 2n rows [dL ; dR], rows ascending (colsum_two_kernel's order)
        float t = 0.f;
        for (int r = 0; r < n; ++r) t += ls[r * MINI_KS + tid];
        for (int r = 0; r < n; ++r) t += rs[r * MINI_KS + tid];
        colsum[k0 + tid] = t;
    }
}

// out[k] = sum_r dL[r][k] + sum_r dR[r][k], rows ascending, dL's first: the column sums of the 2n x D matrix [dL ; dR]
__global__ void colsum_two_kernel(const float* __restrict__ dL, const float* __restrict__ dR, float* __restrict__ out, int n, int D) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= D) return;
    float t = 0.f;
    for (int r = 0; r < n; ++r) t += dL[(size_t)r * D + k];
    for (int r = 0; r < n; ++r) t += dR[(size_t)r * D + k];
    out[k] = t;
}

bool mini_ok(const alink_head* h, int n) {
    return g_use_mini && h->d_mini && !h->qmode && n <= MINI_N && h->h1 == 128 && h->h2 == 32 && h->od == 2 && h->oW1 == 0 &&
           h->D % (8 * MINI_KS) == 0;
}

int mini_step(alink_head* h, const float* L, const float* R, const float* y, const float* sw, int n, float grad_scale,
              int relu_in, float* dL, float* dR, float* colsum, float* metrics, hipStream_t st) {
    const int D = h->D, S = D / MINI_KS;
    float* P = h->d_params;
    float* G = h->d_grads;
    if (n <= 16) hipLaunchKernelGGL(mini_dense1_kernel<8>, dim3(S), dim3(256), 0, st, L, R, P + h->oW1, h->d_mini, n, D);
    else hipLaunchKernelGGL(mini_dense1_kernel<16>, dim3(S), dim3(256), 0, st, L, R, P + h->oW1, h->d_mini, n, D);
    MiniMid mp{};
    mp.part = h->d_mini; mp.b1 = P + h->ob1; mp.W2 = P + h->oW2; mp.b2 = P + h->ob2; mp.W3 = P + h->oW3; mp.b3 = P + h->ob3;
    mp.y = y; mp.sw = sw; mp.gW2 = G + h->oW2; mp.gb2 = G + h->ob2; mp.gW3 = G + h->oW3; mp.gb3 = G + h->ob3; mp.dz1 = h->d_dz1;
    mp.probs = h->d_p; mp.metrics = metrics; mp.n = n; mp.S = S; mp.grad_scale = grad_scale;
    hipLaunchKernelGGL(mini_mid_kernel, dim3(1), dim3(1024), 0, st, mp);
    if (n <= 16) hipLaunchKernelGGL(mini_wgrad1_kernel<2>, dim3(S), dim3(256), 0, st, L, R, P + h->oW1, h->d_dz1, G + h->oW1, G + h->ob1, dL, dR, colsum, n, D, relu_in);
    else hipLaunchKernelGGL(mini_wgrad1_kernel<4>, dim3(S), dim3(256), 0, st, L, R, P + h->oW1, h->d_dz1, G + h->oW1, G + h->ob1, dL, dR, colsum, n, D, relu_in);
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}


// forward (+ optional backward) on a small batch with the VALU kernels
// The step is a chain of dependent small kernels (each ~7 us of pure latency at batch 16), so the chain
// is kept short: |l - r| inside the first Dense, the two independent middle gradients in one launch,
// and (fused_update) the first-layer weight gradient together with the whole Adadelta update: 5 launches.
int small_pass(alink_head* h, const float* L, const float* R, const float* y, const float* sw, int n,
               float grad_scale, bool want_grads, float* metrics, hipStream_t st, bool fused_update = false,
               const int* li = nullptr, const int* ri = nullptr) {
    ALINK_REQUIRE(n > 0 && n <= h->cap, ALINK_EINVAL, "batch of %d rows outside 1..%d", n, h->cap);
    const int D = h->D, h1 = h->h1, h2 = h->h2;
    float* P = h->d_params;
    float* G = h->d_grads;
    const int q = h->qmode ? 1 : 0;
    if (q) { const int rc = ensure_q(h, st, true); if (rc) return rc; }
    const float* WF = q ? h->d_pqf : P;            // weights: the bf16 values widened to f32 in bf16 mode; biases: master
    hipLaunchKernelGGL(dense_fwd_tiled_kernel, dim3((h1 + 63) / 64, n), dim3(512), 0, st, L, WF + h->oW1,
                       P + h->ob1, h->d_z1, n, D, h1, 0, R, h->d_dm, q, li, ri);
    hipLaunchKernelGGL(dense_fwd_tiled_kernel, dim3((h2 + 63) / 64, n), dim3(512), 0, st, h->d_z1, WF + h->oW2,
                       P + h->ob2, h->d_z2, n, h1, h2, 1, (const float*)nullptr, (float*)nullptr, q, (const int*)nullptr,
                       (const int*)nullptr);
    HeadLoss lp{};
    lp.q = q;
    lp.z2 = h->d_z2; lp.w3 = WF + h->oW3; lp.b3 = P + h->ob3; lp.y = y; lp.sw = sw; lp.probs = h->d_p;
    lp.dz3 = h->d_dz3; lp.dz2 = h->d_dz2; lp.gw3 = G + h->oW3; lp.gb3 = G + h->ob3; lp.metrics = metrics;
    lp.n = n; lp.h2 = h2; lp.want_grads = want_grads ? 1 : 0; lp.grad_scale = grad_scale; lp.od = h->od;
    hipLaunchKernelGGL(head_loss_kernel, dim3(1), dim3(256), 0, st, lp);
    if (want_grads) {
        const int nb_w = (h1 * h2 + h2 + 255) / 256, nb_d = (n * h1 + 255) / 256;
        hipLaunchKernelGGL(head_bwd_mid_kernel, dim3(nb_w + nb_d), dim3(256), 0, st, h->d_z1, h->d_dz2, WF + h->oW2,
                           G + h->oW2, G + h->ob2, h->d_dz1, n, h1, h2, nb_w, q);
        if (fused_update) {
            const size_t n_first = (size_t)D * h1 + h1;            // W1 and b1 are the first parameters (oW1 = 0)
            const int nb_first = (int)((n_first + 255) / 256), nb_rest = (int)((h->nparams - n_first + 255) / 256);
            hipLaunchKernelGGL(head_wgrad1_update_kernel, dim3(nb_first + nb_rest), dim3(256), 0, st, h->d_dm, h->d_dz1,
                               P, G, h->d_acc, h->d_dacc, n, D, h1, n_first, h->nparams, nb_first, h->lr, h->rho, h->eps);
        } else {
            hipLaunchKernelGGL(dense_wgrad_kernel, g1((long long)D * h1 + h1), dim3(256), 0, st, h->d_dm, h->d_dz1,
                               G + h->oW1, G + h->ob1, n, D, h1, 0);
        }
    }
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

unsigned long long g_head_attr_done = 0;      // one bit per device: function attributes are per device
int head_init_attrs() {
    const int dev = current_device();
    if (dev >= 0 && dev < 64 && (g_head_attr_done >> dev & 1ull)) return ALINK_OK;
    const int lds = (int)fwd_lds_bytes(512);
    ALINK_HIP(hipFuncSetAttribute((const void*)head_fwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    ALINK_HIP(hipFuncSetAttribute((const void*)head_fwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    ALINK_HIP(hipFuncSetAttribute((const void*)head_fwd_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    ALINK_HIP(hipFuncSetAttribute((const void*)head_fwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    ALINK_HIP(hipFuncSetAttribute((const void*)head_fwd_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 2 * QP * 8));
    if (dev >= 0 && dev < 64) g_head_attr_done |= 1ull << dev;
    return ALINK_OK;
}

}  // namespace

extern "C" {

alink_head_t* alink_head_create(int d_in, int h1, int h2, float lr, float rho, float eps) {
    return alink_head_create_ex(d_in, h1, h2, 2, lr, rho, eps);
}

alink_head_t* alink_head_create_ex(int d_in, int h1, int h2, int out_dim, float lr, float rho, float eps) {
    if (out_dim != 1 && out_dim != 2) { set_error("out_dim=%d must be 1 (sigmoid) or 2 (softmax)", out_dim); return nullptr; }
    if (d_in <= 0 || d_in % 8) { set_error("d_in=%d must be a positive multiple of 8", d_in); return nullptr; }
    if (h1 < 128 || h1 > 512 || h1 % 128) { set_error("h1=%d must be 128, 256, 384 or 512", h1); return nullptr; }
    if (h2 != 32 && h2 != 64) { set_error("h2=%d must be 32 or 64", h2); return nullptr; }
    if (head_init_attrs()) return nullptr;
    alink_head* h = new alink_head();
    h->device = current_device();
    h->D = d_in; h->h1 = h1; h->h2 = h2; h->od = out_dim; h->lr = lr; h->rho = rho; h->eps = eps;
    h->oW1 = 0; h->ob1 = (size_t)d_in * h1; h->oW2 = h->ob1 + h1; h->ob2 = h->oW2 + (size_t)h1 * h2;
    h->oW3 = h->ob2 + h2; h->ob3 = h->oW3 + (size_t)h2 * out_dim; h->nparams = h->ob3 + out_dim;
    int rc = 0;
    rc |= head_alloc(h, &h->d_params, h->nparams);
    rc |= head_alloc(h, &h->d_grads, h->nparams + 4);   // + 4 spare floats: see alink_head_grads_dev
    rc |= head_alloc(h, &h->d_acc, h->nparams);
    rc |= head_alloc(h, &h->d_dacc, h->nparams);
    rc |= head_alloc(h, &h->d_w1p, (size_t)d_in * h1);
    rc |= head_alloc(h, &h->d_w2p, (size_t)h1 * h2);
    rc |= head_alloc(h, &h->d_dm, (size_t)h->cap * d_in);
    rc |= head_alloc(h, &h->d_z1, (size_t)h->cap * h1);
    rc |= head_alloc(h, &h->d_dz1, (size_t)h->cap * h1);
    rc |= head_alloc(h, &h->d_z2, (size_t)h->cap * h2);
    rc |= head_alloc(h, &h->d_dz2, (size_t)h->cap * h2);
    rc |= head_alloc(h, &h->d_dz3, (size_t)h->cap * 2);
    rc |= head_alloc(h, &h->d_p, (size_t)h->cap * 2);
    rc |= head_alloc(h, &h->d_tiny, (size_t)(TINY_N / TINY_RG) * TINY_PART);
    rc |= head_alloc(h, (float**)&h->d_counter, 4);
    if (h1 == 128 && h2 == 32 && out_dim == 2 && d_in % (8 * MINI_KS) == 0)
        rc |= head_alloc(h, &h->d_mini, (size_t)(d_in / MINI_KS) * MINI_N * 128);
    if (rc) { delete h; return nullptr; }
    return h;
}

void alink_head_destroy(alink_head_t* h) {
    if (!h) return;
    DeviceGuard dg(h->device);
    delete h;
}
size_t alink_head_num_params(const alink_head_t* h) { return h ? h->nparams : 0; }

int alink_head_set_params(alink_head_t* h, const float* host_params, size_t count) {
    ALINK_REQUIRE(h && host_params, ALINK_EINVAL, "NULL argument");
    ALINK_REQUIRE(count == h->nparams, ALINK_EINVAL, "expected %zu parameters, got %zu", h->nparams, count);
    DeviceGuard dg(h->device);
    ALINK_HIP(hipMemcpy(h->d_params, host_params, count * sizeof(float), hipMemcpyHostToDevice));
    h->packed_dirty = h->pq_dirty = h->pqf_dirty = true;
    return ALINK_OK;
}
int alink_head_get_params(const alink_head_t* h, float* host_params, size_t count) {
    ALINK_REQUIRE(h && host_params, ALINK_EINVAL, "NULL argument");
    ALINK_REQUIRE(count == h->nparams, ALINK_EINVAL, "expected %zu parameters, got %zu", h->nparams, count);
    DeviceGuard dg(h->device);
    ALINK_HIP(hipDeviceSynchronize());
    ALINK_HIP(hipMemcpy(host_params, h->d_params, count * sizeof(float), hipMemcpyDeviceToHost));
    return ALINK_OK;
}
int alink_head_reset_optimizer(alink_head_t* h) {
    ALINK_REQUIRE(h, ALINK_EINVAL, "NULL head");
    DeviceGuard dg(h->device);
    ALINK_HIP(hipMemset(h->d_acc, 0, h->nparams * sizeof(float)));
    ALINK_HIP(hipMemset(h->d_dacc, 0, h->nparams * sizeof(float)));
    return ALINK_OK;
}
int alink_head_set_lr(alink_head_t* h, float lr) {
    ALINK_REQUIRE(h && lr >= 0.f, ALINK_EINVAL, "bad lr");
    h->lr = lr;
    return ALINK_OK;
}
float alink_head_get_lr(const alink_head_t* h) { return h ? h->lr : 0.f; }
float* alink_head_params_dev(alink_head_t* h) {
    if (!h) return nullptr;
    h->packed_dirty = h->pq_dirty = h->pqf_dirty = true;      // the caller may write through it: re-derive every weight copy on next use
    return h->d_params;
}
float* alink_head_grads_dev(alink_head_t* h) { return h ? h->d_grads : nullptr; }

int alink_head_forward(alink_head_t* h, const float* dev_L, const float* dev_R, const int32_t* dev_li,
                       const int32_t* dev_ri, int64_t P, float* dev_probs, void* stream) {
    ALINK_REQUIRE(h && dev_L && dev_R && dev_probs, ALINK_EINVAL, "NULL argument");
    DeviceGuard dg(h->device);
    if (P == 0) return ALINK_OK;
    ALINK_REQUIRE(P > 0, ALINK_EINVAL, "negative pair count");
    return launch_fwd(h, dev_L, dev_R, dev_li, dev_ri, P, dev_probs, 0, 0.f, (hipStream_t)stream);
}

int alink_committee_forward(alink_head_t* const* heads, int n_heads, const float* dev_L, const float* dev_R,
                            const int32_t* dev_li, const int32_t* dev_ri, int64_t P, float* dev_probs,
                            void* dev_scratch, void* stream) {
    (void)dev_scratch;
    ALINK_REQUIRE(heads && n_heads > 0 && dev_L && dev_R && dev_probs, ALINK_EINVAL, "bad argument");
    ALINK_REQUIRE(heads[0], ALINK_EINVAL, "NULL committee member 0");
    DeviceGuard dg(heads[0]->device);
    if (P == 0) return ALINK_OK;
    for (int m = 0; m < n_heads; ++m) {
        ALINK_REQUIRE(heads[m], ALINK_EINVAL, "NULL committee member %d", m);
        // np.sum over members then / len(models) (code/committee.py:18): sequential f32 adds, one divide
        const int rc = launch_fwd(heads[m], dev_L, dev_R, dev_li, dev_ri, P, dev_probs, m > 0,
                                  m == n_heads - 1 ? (float)n_heads : 0.f, (hipStream_t)stream);
        if (rc) return rc;
    }
    return ALINK_OK;
}

int alink_committee_forward_multi(alink_head_t* const* heads, int n_heads, const float* const* dev_L,
                                  const float* const* dev_R, const int32_t* dev_li, const int32_t* dev_ri, int64_t P,
                                  float* dev_probs, void* stream) {
    ALINK_REQUIRE(heads && n_heads > 0 && dev_L && dev_R && dev_probs, ALINK_EINVAL, "bad argument");
    ALINK_REQUIRE(heads[0], ALINK_EINVAL, "NULL committee member 0");
    DeviceGuard dg(heads[0]->device);
    if (P == 0) return ALINK_OK;
    ALINK_REQUIRE(P > 0, ALINK_EINVAL, "negative pair count");
    for (int m = 0; m < n_heads; ++m) {
        ALINK_REQUIRE(heads[m] && dev_L[m] && dev_R[m], ALINK_EINVAL, "NULL committee member / matrix %d", m);
        ALINK_REQUIRE(heads[m]->device == heads[0]->device, ALINK_EINVAL, "committee members live on different devices");
        // same accumulation as alink_committee_forward: member softmaxes added in member order, one divide
        const int rc = launch_fwd(heads[m], dev_L[m], dev_R[m], dev_li, dev_ri, P, dev_probs, m > 0,
                                  m == n_heads - 1 ? (float)n_heads : 0.f, (hipStream_t)stream);
        if (rc) return rc;
    }
    return ALINK_OK;
}

int alink_pair_scores_matrix(alink_head_t* const* heads, int n_heads, const float* dev_emb, int n, int row0,
                             int nrows, int col, float* dev_scores, void* stream) {
    ALINK_REQUIRE(heads && n_heads > 0 && dev_emb && dev_scores, ALINK_EINVAL, "bad argument");
    ALINK_REQUIRE(heads[0], ALINK_EINVAL, "NULL committee member 0");
    DeviceGuard dg(heads[0]->device);
    ALINK_REQUIRE(n > 0 && row0 >= 0 && nrows >= 0 && row0 + nrows <= n, ALINK_EINVAL,
                  "rows [%d, %d) outside a %d x %d matrix", row0, row0 + nrows, n, n);
    ALINK_REQUIRE(col >= 0 && col < heads[0]->od, ALINK_EINVAL, "col=%d outside the model's %d output(s)", col, heads[0]->od);
    if (nrows == 0) return ALINK_OK;
    for (int m = 0; m < n_heads; ++m) {
        ALINK_REQUIRE(heads[m], ALINK_EINVAL, "NULL committee member %d", m);
        const int rc = launch_fwd(heads[m], dev_emb, dev_emb, nullptr, nullptr, (long long)nrows * n, dev_scores,
                                  m > 0, (n_heads > 1 && m == n_heads - 1) ? (float)n_heads : 0.f,
                                  (hipStream_t)stream, n, row0, heads[m]->od == 1 ? -1 : col);
        if (rc) return rc;
    }
    return ALINK_OK;
}

static int train_step_launches(alink_head_t* h, const float* dev_L, const float* dev_R, const float* dev_y,
                               const float* dev_sw, int n, float grad_scale, int apply, float* dev_metrics,
                               hipStream_t st) {
    int rc = tiny_ok(h, n) ? tiny_train(h, dev_L, dev_R, dev_y, dev_sw, n, grad_scale, apply != 0, dev_metrics, st)
                           : small_pass(h, dev_L, dev_R, dev_y, dev_sw, n, grad_scale, true, dev_metrics, st, apply != 0);
    if (rc) return rc;
    if (apply) {
        h->packed_dirty = h->pqf_dirty = true;
        if (!tiny_ok(h, n)) h->pq_dirty = true;        // the batch<=32 step writes the bf16 copy itself
    }
    return ALINK_OK;
}

void alink_debug_set_tiny_step(int on) { g_use_tiny = on != 0; }
void alink_debug_set_mini_step(int on) { g_use_mini = on != 0; }
void alink_debug_set_head_bf16_mfma(int on) { g_use_bf16_mfma = on != 0; }

int alink_head_set_graph(alink_head_t* h, int on) {
    ALINK_REQUIRE(h, ALINK_EINVAL, "NULL head");
    h->use_graph = on != 0;
    return ALINK_OK;
}

int alink_head_train_step(alink_head_t* h, const float* dev_L, const float* dev_R, const float* dev_y,
                          const float* dev_sw, int n, float grad_scale, int apply, float* dev_metrics,
                          void* stream) {
    ALINK_REQUIRE(h && dev_L && dev_R && dev_y && dev_metrics, ALINK_EINVAL, "NULL argument");
    ALINK_REQUIRE(n > 0 && n <= h->cap, ALINK_EINVAL, "batch of %d rows outside 1..%d", n, h->cap);
    DeviceGuard dg(h->device);
    hipStream_t st = (hipStream_t)stream;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    // the legacy default stream cannot be captured, and a stream the caller is already capturing must
    // simply receive the launches
    const bool can_graph = h->use_graph && !h->qmode && st != nullptr &&
                           hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone;
    if (!can_graph) return train_step_launches(h, dev_L, dev_R, dev_y, dev_sw, n, grad_scale, apply, dev_metrics, st);
    for (auto& g : h->graphs)
        if (g.L == dev_L && g.R == dev_R && g.y == dev_y && g.sw == dev_sw && g.metrics == dev_metrics && g.n == n &&
            g.apply == apply && g.grad_scale == grad_scale && g.lr == h->lr) {
            ALINK_HIP(hipGraphLaunch(g.exec, st));
            if (apply) h->packed_dirty = true;
            return ALINK_OK;
        }
    hipGraph_t graph = nullptr;
    ALINK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = train_step_launches(h, dev_L, dev_R, dev_y, dev_sw, n, grad_scale, apply, dev_metrics, st);
    const hipError_t ee = hipStreamEndCapture(st, &graph);
    if (rc || ee != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        if (rc) return rc;
        h->use_graph = false;                       // capture is not available here: plain launches from now on
        return train_step_launches(h, dev_L, dev_R, dev_y, dev_sw, n, grad_scale, apply, dev_metrics, st);
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) {
        (void)hipGetLastError();
        h->use_graph = false;
        return train_step_launches(h, dev_L, dev_R, dev_y, dev_sw, n, grad_scale, apply, dev_metrics, st);
    }
    if (h->graphs.size() >= 16) {
        (void)hipGraphExecDestroy(h->graphs.front().exec);
        h->graphs.erase(h->graphs.begin());
    }
    h->graphs.push_back({dev_L, dev_R, dev_y, dev_sw, dev_metrics, n, apply, grad_scale, h->lr, exec});
    ALINK_HIP(hipGraphLaunch(exec, st));
    if (apply) h->packed_dirty = true;
    return ALINK_OK;
}

int alink_head_apply_update(alink_head_t* h, void* stream) {
    ALINK_REQUIRE(h, ALINK_EINVAL, "NULL head");
    DeviceGuard dg(h->device);
    hipLaunchKernelGGL(adadelta_kernel, g1((long long)h->nparams), dim3(256), 0, (hipStream_t)stream,
                       h->d_params, h->d_grads, h->d_acc, h->d_dacc, h->nparams, h->lr, h->rho, h->eps);
    ALINK_HIP(hipGetLastError());
    h->packed_dirty = h->pq_dirty = h->pqf_dirty = true;
    return ALINK_OK;
}

int alink_head_apply_update_with(alink_head_t* h, float* dev_params2, const float* dev_grads2, float* dev_acc2, float* dev_dacc2,
                                 size_t n2, void* stream) {
    ALINK_REQUIRE(h && dev_params2 && dev_grads2 && dev_acc2 && dev_dacc2 && n2 > 0, ALINK_EINVAL, "NULL argument");
    DeviceGuard dg(h->device);
    const int nb2 = (int)((n2 + 255) / 256), nb1 = (int)((h->nparams + 255) / 256);
    hipLaunchKernelGGL(adadelta_two_kernel, dim3(nb2 + nb1), dim3(256), 0, (hipStream_t)stream, dev_params2, dev_grads2, dev_acc2,
                       dev_dacc2, n2, h->d_params, h->d_grads, h->d_acc, h->d_dacc, h->nparams, nb2, h->lr, h->rho, h->eps);
    ALINK_HIP(hipGetLastError());
    h->packed_dirty = h->pq_dirty = h->pqf_dirty = true;
    return ALINK_OK;
}

int alink_head_set_compute_dtype(alink_head_t* h, int dtype) {
    ALINK_REQUIRE(h, ALINK_EINVAL, "NULL head");
    ALINK_REQUIRE(dtype == ALINK_DT_F32 || dtype == ALINK_DT_BF16, ALINK_EINVAL, "compute dtype must be ALINK_DT_F32 or ALINK_DT_BF16");
    DeviceGuard dg(h->device);
    if (dtype == ALINK_DT_BF16 && !h->d_pq) {
        // all three buffers or none: a half-allocated set must not leave d_pq set (a retry would skip this block and
        // the bf16 kernels would dereference the missing ones)
        void *pq = nullptr, *pqf = nullptr, *wt = nullptr;
        hipError_t e = hipMalloc(&pq, h->nparams * sizeof(__bf16));
        if (e == hipSuccess) e = hipMalloc(&pqf, h->nparams * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(&wt, ((size_t)h->D * h->h1 + (size_t)h->h1 * h->h2) * sizeof(__bf16));
        if (e != hipSuccess) {
            (void)hipFree(pq); (void)hipFree(pqf); (void)hipFree(wt);
            return hip_fail(e, "hipMalloc (bf16 compute mode buffers)", __FILE__, __LINE__);
        }
        h->d_pq = (decltype(h->d_pq))pq; h->d_pqf = (decltype(h->d_pqf))pqf; h->d_wt = (decltype(h->d_wt))wt;
        h->allocs.push_back(pq); h->allocs.push_back(pqf); h->allocs.push_back(wt);
    }
    h->qmode = dtype == ALINK_DT_BF16 ? 1 : 0;
    h->packed_dirty = h->pq_dirty = h->pqf_dirty = true;
    return ALINK_OK;
}
int alink_head_get_compute_dtype(const alink_head_t* h) { return h && h->qmode ? ALINK_DT_BF16 : ALINK_DT_F32; }

static int input_grads(alink_head_t* h, const float* dev_L, const float* dev_R, int n, float* dev_dL, float* dev_dR, int relu_in,
                       void* stream) {
    ALINK_REQUIRE(h && dev_L && dev_R && dev_dL && dev_dR, ALINK_EINVAL, "NULL argument");
    ALINK_REQUIRE(n > 0 && n <= h->cap, ALINK_EINVAL, "batch of %d rows outside 1..%d", n, h->cap);
    ALINK_REQUIRE(!h->qmode, ALINK_ESTATE, "input gradients (the SmallRes tower) are float32 only");
    DeviceGuard dg(h->device);
    hipLaunchKernelGGL(head_input_grad_kernel, dim3((h->D + 63) / 64, (n + 15) / 16), dim3(256), 0, (hipStream_t)stream, dev_L,
                       dev_R, h->d_dz1, h->d_params + h->oW1, dev_dL, dev_dR, n, h->D, h->h1, relu_in);
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}
int alink_head_input_grads(alink_head_t* h, const float* dev_L, const float* dev_R, int n, float* dev_dL,
                           float* dev_dR, void* stream) {
    return input_grads(h, dev_L, dev_R, n, dev_dL, dev_dR, 0, stream);
}
int alink_head_input_grads_relu(alink_head_t* h, const float* dev_L, const float* dev_R, int n, float* dev_dL,
                                float* dev_dR, void* stream) {
    return input_grads(h, dev_L, dev_R, n, dev_dL, dev_dR, 1, stream);
}

int alink_head_train_step_input_grads(alink_head_t* h, const float* dev_L, const float* dev_R, const float* dev_y,
                                      const float* dev_sw, int n, float grad_scale, int relu_inputs, float* dev_dL,
                                      float* dev_dR, float* dev_colsum, float* dev_metrics, void* stream) {
    ALINK_REQUIRE(h && dev_L && dev_R && dev_y && dev_metrics && dev_dL && dev_dR, ALINK_EINVAL, "NULL argument");
    ALINK_REQUIRE(n > 0 && n <= h->cap, ALINK_EINVAL, "batch of %d rows outside 1..%d", n, h->cap);
    ALINK_REQUIRE(!h->qmode, ALINK_ESTATE, "input gradients (the SmallRes tower) are float32 only");
    if (mini_ok(h, n)) {
        DeviceGuard dg(h->device);
        return mini_step(h, dev_L, dev_R, dev_y, dev_sw, n, grad_scale, relu_inputs != 0, dev_dL, dev_dR, dev_colsum, dev_metrics, (hipStream_t)stream);
    }
    int rc = alink_head_train_step(h, dev_L, dev_R, dev_y, dev_sw, n, grad_scale, 0, dev_metrics, stream);
    if (rc) return rc;
    if ((rc = input_grads(h, dev_L, dev_R, n, dev_dL, dev_dR, relu_inputs != 0, stream))) return rc;
    if (dev_colsum) {
        DeviceGuard dg(h->device);
        hipLaunchKernelGGL(colsum_two_kernel, g1(h->D), dim3(256), 0, (hipStream_t)stream, dev_dL, dev_dR, dev_colsum, n, h->D);
        ALINK_HIP(hipGetLastError());
    }
    return ALINK_OK;
}

int alink_head_eval(alink_head_t* h, const float* dev_L, const float* dev_R, const float* dev_y, int n,
                    float* dev_metrics, void* stream) {
    ALINK_REQUIRE(h && dev_L && dev_R && dev_y && dev_metrics, ALINK_EINVAL, "NULL argument");
    DeviceGuard dg(h->device);
    if (tiny_ok(h, n)) return tiny_eval(h, dev_L, dev_R, dev_y, n, dev_metrics, (hipStream_t)stream);
    return small_pass(h, dev_L, dev_R, dev_y, nullptr, n, 0.f, false, dev_metrics, (hipStream_t)stream);
}

int alink_head_custom_train_steps(alink_head_t* h, const float* dev_table, const int32_t* dev_idx, const float* dev_vals,
                                  const int64_t* host_desc, int steps, int with_weights, float* dev_metrics, void* stream) {
    ALINK_REQUIRE(h && dev_table && dev_idx && dev_vals && host_desc && dev_metrics && steps >= 0, ALINK_EINVAL, "NULL argument");
    DeviceGuard dg(h->device);
    hipStream_t st = (hipStream_t)stream;
    for (int s = 0; s < steps; ++s) {
        const int64_t io = host_desc[4 * s], fo = host_desc[4 * s + 1], n = host_desc[4 * s + 2], nh = host_desc[4 * s + 3];
        ALINK_REQUIRE(io >= 0 && fo >= 0 && nh >= 0 && n > nh && n - nh <= h->cap && nh <= h->cap, ALINK_EINVAL,
                      "step %d: %lld rows, %lld held out (at most %d each)", s, (long long)n, (long long)nh, h->cap);
        const int nu = (int)(n - nh);
        const int32_t *li = dev_idx + io, *ri = li + n;
        const float* y = dev_vals + fo;
        const float* sw = with_weights ? y + (size_t)n * h->od : nullptr;
        float* m = dev_metrics + 4 * (size_t)s;
        // train_on_batch on the rows that were not held out (the same kernels alink_head_train_step launches for a batch of this size)
        int rc = tiny_ok(h, nu) ? tiny_train(h, dev_table, dev_table, y + (size_t)nh * h->od, sw, nu, 0.f, true, m, st, li + nh, ri + nh)
                                : small_pass(h, dev_table, dev_table, y + (size_t)nh * h->od, sw, nu, 0.f, true, m, st, true, li + nh, ri + nh);
        if (rc) return rc;
        h->packed_dirty = h->pqf_dirty = true;
        if (!tiny_ok(h, nu)) h->pq_dirty = true;
        // test_on_batch on the held-out rows, with the parameters that step left
        if (nh > 0) {
            rc = tiny_ok(h, (int)nh) ? tiny_eval(h, dev_table, dev_table, y, (int)nh, m + 2, st, li, ri)
                                     : small_pass(h, dev_table, dev_table, y, nullptr, (int)nh, 0.f, false, m + 2, st, false, li, ri);
            if (rc) return rc;
        }
    }
    return ALINK_OK;
}

}  // extern "C"

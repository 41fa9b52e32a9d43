// evaluate.hip — DFW verification protocol counts on device (HBM-bound integer/compare work).
//
// Replaces the Python loops of reference utilities/ROC_precompute.py:19-63: walk the strict upper
// triangle of the N x N score matrix, sort every pair into genuine / impostor / unused by the
// protocol mask and `roc_case` (:27-44), and for every threshold count the genuine (true positive)
// and impostor (false positive) scores >= threshold (:51-63).
//
// One streaming pass: each score is binary-searched into the ascending thresholds — c = number of
// thresholds <= score — and counted into a (T+1)-bin histogram per class, held in LDS and flushed
// once per workgroup.  TP[t] (t = rank of the threshold) is then the suffix sum over bins > t — a
// T-sized host loop.  Counts are integers: results are exact, independent of the launch shape.
// Scores are f32 (what alink_pair_scores_matrix writes) widened to f64 for the comparison, which is
// what the reference does by round-tripping them through np.savetxt / np.loadtxt.
#include "alink_common.h"

namespace alink {
namespace {

struct RocParams {
    const float* scores;           // [N][N]
    const unsigned char* mask;     // [N][N] protocol codes 0..255
    const double* thr;             // [T] ascending
    unsigned long long* hist;      // [2][T+1]: genuine bins then impostor bins
    int N, T, roc_case, use_lds;
};

__device__ __forceinline__ int klass(int m, int roc_case) {
    // 0 = genuine, 1 = impostor, -1 = unused   (utilities/ROC_precompute.py:27-44)
    if (roc_case == 3) return (m == 1 || m == 2) ? 0 : ((m == 3 || m == 4) ? 1 : -1);
    if (roc_case == 2) return m == 2 ? 0 : (m == 4 ? 1 : -1);
    return m == 1 ? 0 : (m == 3 ? 1 : -1);
}

__global__ __launch_bounds__(256) void roc_hist_kernel(const RocParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int T = p.T, N = p.N, tid = threadIdx.x;
    double* sthr = (double*)sm;                                  // [T]
    unsigned int* shist = (unsigned int*)(sm + (size_t)T * 8);  // [2][T+1]
    if (p.use_lds) {
        for (int i = tid; i < T; i += 256) sthr[i] = p.thr[i];
        for (int i = tid; i < 2 * (T + 1); i += 256) shist[i] = 0u;
        __syncthreads();
    }
    const double* thr = p.use_lds ? sthr : p.thr;
    // rows are dealt round-robin to workgroups; a row's columns j > i are read coalesced
    for (int i = blockIdx.x; i < N - 1; i += gridDim.x) {
        const float* srow = p.scores + (size_t)i * N;
        const unsigned char* mrow = p.mask + (size_t)i * N;
        for (int j = i + 1 + tid; j < N; j += 256) {
            const int k = klass(mrow[j], p.roc_case);
            if (k < 0) continue;
            const double s = (double)srow[j];
            int lo = 0, hi = T;                 // c = #{t : thr[t] <= s}; NaN compares false -> 0
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (thr[mid] <= s) lo = mid + 1; else hi = mid;
            }
            if (p.use_lds) atomicAdd(&shist[k * (T + 1) + lo], 1u);
            else atomicAdd(&p.hist[(size_t)k * (T + 1) + lo], 1ull);
        }
    }
    if (p.use_lds) {
        __syncthreads();
        for (int i = tid; i < 2 * (T + 1); i += 256)
            if (shist[i]) atomicAdd(&p.hist[i], (unsigned long long)shist[i]);
    }
}

}  // namespace
}  // namespace alink

using namespace alink;

extern "C" int alink_roc_counts(const float* dev_scores, const uint8_t* dev_mask, int n,
                                const double* dev_thr_sorted, int n_thr, int roc_case,
                                unsigned long long* dev_hist, void* stream) {
    ALINK_REQUIRE(dev_scores && dev_mask && dev_thr_sorted && dev_hist, ALINK_EINVAL, "NULL argument");
    DeviceGuard dg(device_of_pointer(dev_scores));
    ALINK_REQUIRE(n >= 0 && n_thr > 0, ALINK_EINVAL, "n=%d, n_thr=%d", n, n_thr);
    ALINK_REQUIRE(roc_case >= 1 && roc_case <= 3, ALINK_EINVAL, "roc_case=%d must be 1, 2 or 3", roc_case);
    hipStream_t st = (hipStream_t)stream;
    ALINK_HIP(hipMemsetAsync(dev_hist, 0, sizeof(unsigned long long) * 2 * ((size_t)n_thr + 1), st));
    if (n < 2) return ALINK_OK;
    RocParams p{dev_scores, dev_mask, dev_thr_sorted, dev_hist, n, n_thr, roc_case, 0};
    const size_t lds = (size_t)n_thr * 8 + 2 * ((size_t)n_thr + 1) * 4;
    p.use_lds = lds <= 64 * 1024;
    // per-workgroup row walk: one block per row up to 4 blocks per CU (u32 LDS bins cannot overflow:
    // a workgroup sees < 2^32 pairs for any n that fits an int)
    const int grid = n - 1 < 1024 ? n - 1 : 1024;
    hipLaunchKernelGGL(roc_hist_kernel, dim3(grid), dim3(256), p.use_lds ? lds : 0, st, p);
    ALINK_HIP(hipGetLastError());
    return ALINK_OK;
}

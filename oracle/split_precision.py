"""oracle.split_precision — CPU restatement (NumPy) of the SPLIT-PRECISION arithmetic of the product's `dtype="f16x2"`
mode (include/alink_hip.h: ALINK_DT_F16X2; csrc/conv3x3_linear.hip, conv_igemm.hip, stem_tail.hip, SP forms).

TEST INFRASTRUCTURE ONLY: nothing under a-link_amd/ imports this file.  It exists so that the numerical claims DESIGN.md §5
makes for the mode can be checked on the CPU (tests/test_oracle_split_precision.py) without a GPU:

    x = hi + lo,  hi = RN16(x * 2^e),  lo = RN16(x * 2^e - hi)           an f16 pair: 22 significant bits while lo is normal
    sum_k x_k w_k  ~=  2^-(ex+ew) * sum_k ( xh_k wh_k + xh_k wl_k + xl_k wh_k )      products of f16 values are exact in f32,
                                                                                    accumulated in f32; xl wl is dropped

The reference computes the same layers in float32 (MXNet, reference code/face_model.py:90); the mode's claim is "the accuracy
of float32" — here measured against float64.
"""
import numpy as np


def scale_exp(maxabs):
    """e with maxabs * 2^e in [1024, 2048) — csrc/backbone.hip scale_exp."""
    if not np.isfinite(maxabs) or maxabs <= 0:
        return 0
    return 10 - int(np.floor(np.log2(maxabs)))


def split(x, e):
    """-> (hi, lo) float16 arrays of x * 2^e."""
    s = np.ldexp(np.asarray(x, np.float64), e)
    hi = s.astype(np.float16)
    lo = (s - hi.astype(np.float64)).astype(np.float16)
    return hi, lo


def join(hi, lo, e):
    return np.ldexp(hi.astype(np.float64) + lo.astype(np.float64), -e)


def matmul(x, w, ex=None, ew=None):
    """x (M, K) @ w (K, N) the way the SP kernels do it: three f32 GEMMs of f16 operands into one f32 accumulator (the
    device adds them K-step by K-step in its MFMA accumulators; the order of f32 additions differs, the error level not),
    scaled back by 2^-(ex+ew).  Returns float32."""
    ex = scale_exp(np.abs(x).max()) if ex is None else ex
    ew = scale_exp(np.abs(w).max()) if ew is None else ew
    xh, xl = split(x, ex)
    wh, wl = split(w, ew)
    f = np.float32
    acc = xl.astype(f) @ wh.astype(f)
    acc = acc + xh.astype(f) @ wl.astype(f)
    acc = acc + xh.astype(f) @ wh.astype(f)
    return np.ldexp(acc, -(ex + ew)).astype(np.float32)

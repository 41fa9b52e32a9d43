"""CPU: the numerical claims DESIGN.md §5 makes for the split-precision mode (dtype="f16x2"), on the NumPy restatement of
its arithmetic (oracle/split_precision.py).  The GPU tests check the kernels against the f32 oracle; these check the
ALGORITHM against float64 and against plain float32."""
import numpy as np

from oracle import split_precision as SP


def test_pair_represents_22_bits_and_sums_to_f32_precision():
    rng = np.random.default_rng(0)
    x = rng.standard_normal(200000) * np.exp(rng.uniform(-3, 3, 200000))
    e = SP.scale_exp(np.abs(x).max())
    hi, lo = SP.split(x, e)
    assert np.isfinite(hi.astype(np.float64)).all() and np.abs(hi.astype(np.float64)).max() < 2048.5
    s = np.ldexp(x, e)
    err = np.abs(s - (hi.astype(np.float64) + lo.astype(np.float64)))
    normal = np.abs(lo.astype(np.float64)) >= 2.0 ** -14                      # lo a normal f16
    assert (err[normal] <= 2.0 ** -21 * np.abs(s[normal])).all()                # 22 significant bits (round to nearest: 2^-22 typical)
    assert (err <= np.maximum(2.0 ** -21 * np.abs(s), 2.0 ** -25)).all()        # below that: gradual, an absolute floor of 2^-25
    # the epilogue's residual add forms hi + lo in float32: usually exact, and never worse than one f32 rounding (lo can sit
    # more than 13 bits below hi's last place when hi happens to land close to x)
    s32 = hi.astype(np.float32) + lo.astype(np.float32)
    exact = hi.astype(np.float64) + lo.astype(np.float64)
    assert (np.abs(s32.astype(np.float64) - exact) <= 1.0001 * 2.0 ** -24 * np.abs(exact)).all()
    assert (s32.astype(np.float64) == exact).mean() > 0.8


def test_three_product_gemm_is_as_accurate_as_float32():
    """K = 2304 (a 3x3 convolution over 256 channels): error against float64, relative to sum |x w|, of plain float32 and
    of the three-product scheme — the same level (both are dominated by f32 accumulation, not by operand rounding)."""
    rng = np.random.default_rng(1)
    x = rng.standard_normal((64, 2304))
    w = rng.standard_normal((2304, 48)) / 48.0
    ref = x @ w
    mag = np.abs(x) @ np.abs(w)
    e32 = np.abs((x.astype(np.float32) @ w.astype(np.float32)).astype(np.float64) - ref) / mag
    esp = np.abs(SP.matmul(x, w).astype(np.float64) - ref) / mag
    e16 = np.abs((x.astype(np.float16).astype(np.float32) @ w.astype(np.float16).astype(np.float32)).astype(np.float64) - ref) / mag
    print("error / sum|xw|: float32 max %.1e mean %.1e | split precision max %.1e mean %.1e | plain f16 operands max %.1e"
          % (e32.max(), e32.mean(), esp.max(), esp.mean(), e16.max()))
    assert esp.max() < 4e-7 and esp.mean() < 3 * max(e32.mean(), 2e-8)
    assert e16.max() > 20 * esp.max()                                          # what the pair buys over one f16


def test_power_of_two_scales_move_a_result_only_through_subnormal_lo_halves():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((32, 576))
    w = rng.standard_normal((576, 32))
    a = SP.matmul(x, w, ex=9, ew=10)
    b = SP.matmul(x, w, ex=4, ew=10)             # 32x smaller stored values: more lo halves below 2^-14
    c = SP.matmul(x, w, ex=9, ew=7)
    mag = (np.abs(x) @ np.abs(w))
    assert (np.abs(a.astype(np.float64) - b) / mag).max() < 1e-7
    assert (np.abs(a.astype(np.float64) - c) / mag).max() < 1e-7
    # with every lo half a normal number the scale changes nothing at all
    xq = np.round(x * 64) / 64 + 8.0 * np.sign(x)                               # values in 8..14, lo = O(2^-9) of them: never subnormal at e >= 4
    assert np.array_equal(SP.matmul(xq, w, ex=6, ew=10), SP.matmul(xq, w, ex=4, ew=10))

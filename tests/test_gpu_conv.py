"""GPU parity of the implicit-GEMM convolution kernel (conv_igemm.hip) through the C ABI
(alink_conv_nhwc) against a CPU f32 convolution of the same (bf16/f16-rounded) operands."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # N, H, W, Cin, Cout, ksz, stride, pad, border, alpha, resid
    (2, 14, 14, 64, 64, 3, 1, 1, 1, 1, 0),      # unit conv1: border classes + PReLU, 256x64 tile
    (2, 14, 14, 64, 64, 3, 2, 1, 0, 0, 1),      # unit conv2 stride 2 + residual
    (1, 12, 10, 128, 128, 3, 1, 1, 1, 1, 0),    # 128x128 tile, ragged M (120 pixels)
    (3, 8, 8, 64, 128, 1, 2, 0, 0, 0, 0),       # 1x1 stride-2 shortcut
    (2, 7, 7, 256, 256, 3, 1, 1, 0, 0, 1),      # stage-4 like, K = 2304
    (1, 9, 9, 128, 64, 3, 2, 1, 0, 0, 0),       # odd size, stride 2
    (5, 16, 16, 64, 192, 3, 1, 1, 1, 1, 1),     # Cout not a multiple of 128 -> 256x64 tile path
    # shapes served by conv3x3_direct.hip (input tile resident in LDS), one per variant D1..D6
    (3, 14, 14, 256, 256, 3, 1, 1, 1, 1, 0),    # D1: stage-3 conv1 (border classes + PReLU)
    (2, 14, 14, 128, 512, 3, 1, 1, 0, 0, 1),    # D1: two channel tiles, residual, Cin = 2 chunks
    (2, 28, 28, 128, 256, 3, 1, 1, 1, 1, 0),    # D2
    (2, 28, 28, 128, 128, 3, 1, 1, 0, 0, 1),    # D3: stage-2 conv2-like
    (1, 28, 28, 64, 128, 3, 1, 1, 1, 1, 1),     # D3 with a single input chunk
    (2, 56, 56, 64, 128, 3, 1, 1, 1, 1, 0),     # D4
    (2, 56, 56, 64, 64, 3, 1, 1, 1, 1, 1),      # D5: stage 1
    (1, 112, 112, 64, 64, 3, 1, 1, 1, 1, 0),    # stage-1 unit-1 conv1: the rolling-row kernel (conv3x3_c64.hip; D6 with linear = 0)
    (3, 112, 112, 64, 64, 3, 1, 1, 0, 0, 1),    # the same kernel: three images (bands of several images per workgroup), residual
    (37, 112, 112, 64, 64, 3, 1, 1, 1, 1, 1),   # 259 bands for 256 persistent workgroups: some take two; PReLU + residual
    (41, 28, 28, 128, 128, 3, 1, 1, 0, 0, 1),   # stage-2 conv2 over 41 images: 144 groups of 224 pixels, most of them across image boundaries
    (21, 56, 56, 64, 64, 3, 1, 1, 1, 1, 0),     # stage-1 unit over 21 images: 294 groups of 224 pixels across image boundaries
    (3, 112, 112, 64, 64, 3, 2, 1, 0, 0, 1),    # the direct stride-2 kernel (conv3x3_s2c64.hip), plain form: residual; one output row per pass
    (9, 112, 112, 64, 64, 3, 2, 1, 0, 1, 0),    # the same: 504 output rows for 256 workgroups (runs of 1-2 rows across images), PReLU
    (2, 13, 14, 64, 256, 3, 1, 1, 1, 0, 0),     # D1 with a ragged row count (H not a multiple of R)
    (1, 30, 28, 64, 128, 3, 1, 1, 0, 1, 0),     # D3 ragged rows
]


def _ref(x, w, bias, alpha, resid, stride, pad, border):
    y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), stride=stride, padding=pad).permute(0, 2, 3, 1)
    N, Ho, Wo, Co = y.shape
    if border:
        rc = torch.ones(Ho, dtype=torch.long); rc[0] = 0; rc[-1] = 2
        cc = torch.ones(Wo, dtype=torch.long); cc[0] = 0; cc[-1] = 2
        cls = rc[:, None] * 3 + cc[None, :]
        y = y + bias[cls]                      # (Ho,Wo,Co) broadcast over N
    else:
        y = y + bias[0]
    if alpha is not None:
        y = torch.where(y > 0, y, y * alpha)
    if resid is not None:
        y = y + resid
    return y


# linear-tile kernel (conv3x3_linear.hip): image borders and image boundaries inside a 224-pixel group,
# the tail group of a batch, every epilogue mode
LINEAR_CASES = [
    (11, 14, 14, 128, 256, 3, 1, 1, 1, 1, 1),   # 11 images: groups straddle image boundaries, ragged tail
    (3, 28, 28, 64, 128, 3, 1, 1, 1, 0, 1),
    (1, 56, 56, 64, 64, 3, 1, 1, 0, 1, 0),
    (9, 7, 7, 128, 128, 3, 1, 1, 1, 1, 1),      # 16-pixel tiles span three rows; 4.6 images per group
]


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("dma,linear", [(1, -1), (0, -1), (1, 0)])
def test_conv_matches_cpu(gpu, dt, dma, linear):
    lib = gpu.load()
    lib.alink_debug_set_dma(dma)
    if linear >= 0:
        lib.alink_debug_set_linear(linear)      # 0: row-aligned / implicit-GEMM kernels only (library default: linear tiles)
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    code = gpu.DT_BF16 if dt == "bf16" else gpu.DT_F16
    g = torch.Generator().manual_seed(1234)
    try:
        for (N, H, W, Ci, Co, k, s, p, border, use_alpha, use_resid) in CASES + LINEAR_CASES:
            x = (torch.randn(N, H, W, Ci, generator=g)).to(tdt)
            w = (torch.randn(Co, k, k, Ci, generator=g) * (1.0 / np.sqrt(k * k * Ci))).to(tdt)
            ncls = 9 if border else 1
            bias = torch.randn(ncls, Co, generator=g)
            alpha = torch.rand(Co, generator=g) * 0.5 if use_alpha else None
            Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
            resid = torch.randn(N, Ho, Wo, Co, generator=g).to(tdt) if use_resid else None
            ref = _ref(x.float(), w.float(), bias, alpha, resid.float() if use_resid else None, s, p, border)
            xd, wd, bd = x.cuda(), w.cuda(), bias.cuda()
            ad = alpha.cuda() if use_alpha else None
            rd = resid.cuda() if use_resid else None
            out = torch.full((N, Ho, Wo, Co), float("nan"), dtype=tdt, device="cuda")
            is_linear_case = (N, H, W, Ci, Co, k, s, p, border, use_alpha, use_resid) in LINEAR_CASES
            for fine in ((0, 1) if is_linear_case else (-1,)):      # both forms of the linear-tile kernel
                out.fill_(float("nan"))
                rc = lib.alink_conv_nhwc(code, gpu.ptr(xd), gpu.ptr(wd), gpu.ptr(bd), gpu.ptr(ad), gpu.ptr(rd),
                                         gpu.ptr(out), N, H, W, Ci, Co, k, s, p, border, fine, None)
                gpu.check(rc, "alink_conv_nhwc")
                _check_conv(out, ref, dt, (N, H, W, Ci, Co, k, s, p, border, fine), dma)
    finally:
        lib.alink_debug_set_dma(1)
        lib.alink_debug_set_linear(31)          # library default: linear tiles at every width they support


def _check_conv(out, ref, dt, what, dma):
    got = out.float().cpu()
    assert torch.isfinite(got).all(), what
    # output rounding of T (2^-9 bf16 / 2^-12 f16 relative) + accumulation-order noise
    rel = 2.0 ** -8 if dt == "bf16" else 2.0 ** -10
    err = (got - ref).abs()
    tol = rel * ref.abs() + 2e-3
    assert (err <= tol).all(), "case %s dt=%s dma=%d: max err %.4g" % (what, dt, dma, float((err - tol).max()))


def test_split_precision_conv_matches_f64(gpu):
    """ALINK_DT_F16X2 forms of the implicit-GEMM and linear-tile kernels (values as f16 pairs hi + lo, three products,
    f32 accumulation): float32 operands in, float32 out, against a float64 convolution of the SAME float32 operands.
    Error budget: 2^-22 relative per stored operand (x 2 operands + the dropped lo x lo term) and f32 accumulation —
    asserted at 4e-6 of the per-output sum of |products| scale (the 16-bit kernels are asserted at 2^-8 / 2^-10).
    Two sets of scale exponents per case: a power-of-two scale moves a result only through the few lo halves that fall
    into the f16 subnormals (values below 2^-14 of the tensor's stored unit keep fewer than 22 bits) — asserted below 1e-7
    of the same scale; the 64- and 128-channel forms of the linear-tile kernel must agree bit for bit."""
    lib = gpu.load()
    g = torch.Generator().manual_seed(4321)
    worst = 0.0
    for (N, H, W, Ci, Co, k, s, p, border, use_alpha, use_resid) in CASES[:7] + LINEAR_CASES + [(1, 112, 112, 64, 64, 3, 1, 1, 1, 1, 0)]:
        x = torch.randn(N, H, W, Ci, generator=g)
        w = torch.randn(Co, k, k, Ci, generator=g) * (1.0 / np.sqrt(k * k * Ci))
        ncls = 9 if border else 1
        bias = torch.randn(ncls, Co, generator=g)
        alpha = torch.rand(Co, generator=g) * 0.5 if use_alpha else None
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        resid = torch.randn(N, Ho, Wo, Co, generator=g) if use_resid else None
        ref = _ref(x.double(), w.double(), bias.double(), alpha.double() if use_alpha else None,
                   resid.double() if use_resid else None, s, p, border)
        mag = _ref(x.double().abs(), w.double().abs(), bias.double().abs(), None, resid.double().abs() if use_resid else None, s, p, border)
        xd, wd, bd = x.cuda(), w.cuda(), bias.cuda()
        ad = alpha.cuda() if use_alpha else None
        rd = resid.cuda() if use_resid else None
        is_linear_case = (N, H, W, Ci, Co, k, s, p, border, use_alpha, use_resid) in LINEAR_CASES
        outs = {}
        for fine in ((0, 1) if is_linear_case else (-1,)):
            for (e_in, e_w, e_out, e_res) in ((9, 14, 8, 10), (4, 11, 5, 7)):
                out = torch.full((N, Ho, Wo, Co), float("nan"), dtype=torch.float32, device="cuda")
                rc = lib.alink_conv_nhwc_x2(gpu.ptr(xd), gpu.ptr(wd), gpu.ptr(bd), gpu.ptr(ad), gpu.ptr(rd), gpu.ptr(out),
                                            N, H, W, Ci, Co, k, s, p, border, fine, e_in, e_w, e_out, e_res, None)
                gpu.check(rc, "alink_conv_nhwc_x2")
                got = out.cpu()
                assert torch.isfinite(got).all(), (N, H, W, Ci, Co, k, s, p)
                err = ((got.double() - ref).abs() / mag).max().item()
                worst = max(worst, err)
                assert err < 4e-6, ((N, H, W, Ci, Co, k, s, p, border, fine), err)
                outs[(fine, e_in)] = got
        if is_linear_case:
            for e in (9, 4):
                assert torch.equal(outs[(0, e)], outs[(1, e)]), "the 64-channel form changed a bit: %s" % ((N, H, W, Ci, Co, k, s, p),)
        f0 = 0 if is_linear_case else -1
        drift = ((outs[(f0, 9)].double() - outs[(f0, 4)].double()).abs() / mag).max().item()
        assert drift < 1e-7, ((N, H, W, Ci, Co, k, s, p), drift)
    print("split-precision conv: worst error / sum|products| = %.2e" % worst)


def test_conv_rejects_bad_shapes(gpu):
    lib = gpu.load()
    x = torch.zeros(1, 4, 4, 32, dtype=torch.bfloat16, device="cuda")
    rc = lib.alink_conv_nhwc(0, gpu.ptr(x), gpu.ptr(x), gpu.ptr(x), None, None, gpu.ptr(x),
                             1, 4, 4, 32, 64, 3, 1, 1, 0, -1, None)
    assert rc == -1 and b"multiples of 64" in lib.alink_last_error()


def test_lds_out_of_range_read_is_zero(gpu):
    """conv3x3_linear.hip adds border bits 18..27 (one or two at a time) to a lane's operand address, which then
    lies beyond the workgroup's LDS allocation, and relies on the DS read returning zero there (no fault, no
    aliasing of in-range data).  The probe reads every single bit, every pair of bits and wider sums, each with
    immediate offset 0 and 12288 (the largest tile immediate): check that hardware contract directly, and that
    the library recorded it for this device (the linear-tile kernel is only selected where it holds)."""
    import ctypes as C
    lib = gpu.load()
    out = torch.full((516,), 7.0, device="cuda")
    assert lib.alink_debug_lds_oob_probe(C.c_void_p(out.data_ptr()), None) == 0
    torch.cuda.synchronize()
    o = out.cpu()
    assert (o[:512] == 0).all()                          # 64 address forms x 2 immediates x 16 B
    assert o[512:516].tolist() == [1.0, 2.0, 3.0, 4.0]   # the in-range read of the same instruction group
    assert lib.alink_debug_linear_contract_ok() == 1


def test_linear_kernel_fuzz(gpu):
    """Seeded sweep of the linear-tile kernel: every width it serves, batch sizes whose pixel count is not a
    multiple of the 224-pixel group (tail groups, image boundaries at every phase inside a group), one to
    four input chunks, all epilogue modes."""
    lib = gpu.load()
    rng = np.random.RandomState(2024)
    g = torch.Generator().manual_seed(99)
    for case in range(24):
        W = [7, 14, 28, 56][case % 4]
        N = int(rng.randint(1, {7: 40, 14: 24, 28: 7, 56: 3}[W] + 1))
        Ci = 64 * int(rng.randint(1, 5 if W <= 14 else 3))
        Co = 64 if W == 56 else 128 * int(rng.randint(1, 3))
        border, use_alpha, use_resid = int(rng.randint(0, 2)), int(rng.randint(0, 2)), int(rng.randint(0, 2))
        x = torch.randn(N, W, W, Ci, generator=g).to(torch.bfloat16)
        w = (torch.randn(Co, 3, 3, Ci, generator=g) * (1.0 / np.sqrt(9 * Ci))).to(torch.bfloat16)
        bias = torch.randn(9 if border else 1, Co, generator=g)
        alpha = torch.rand(Co, generator=g) * 0.5 if use_alpha else None
        resid = torch.randn(N, W, W, Co, generator=g).to(torch.bfloat16) if use_resid else None
        ref = _ref(x.float(), w.float(), bias, alpha, resid.float() if use_resid else None, 1, 1, border)
        out = torch.full((N, W, W, Co), float("nan"), dtype=torch.bfloat16, device="cuda")
        xd, wd, bd = x.cuda(), w.cuda(), bias.cuda()
        ad = alpha.cuda() if use_alpha else None
        rd = resid.cuda() if use_resid else None
        # both workgroup forms of the kernel (128- and 64-channel; the 56-wide variant has one form): each against
        # the CPU reference, and against each other BIT FOR BIT — alink_embed switches between them by batch size
        # (ConvParams::fine) and promises embeddings that do not depend on the batch an image arrives in
        forms = []
        for fine in (0, 1):
            out.fill_(float("nan"))
            gpu.check(lib.alink_conv_nhwc(gpu.DT_BF16, gpu.ptr(xd), gpu.ptr(wd), gpu.ptr(bd), gpu.ptr(ad), gpu.ptr(rd),
                                          gpu.ptr(out), N, W, W, Ci, Co, 3, 1, 1, border, fine, None), "alink_conv_nhwc")
            forms.append(out.clone())
        assert torch.equal(forms[0].view(torch.int16), forms[1].view(torch.int16)), (case, N, W, Ci, Co)
        got = out.float().cpu()
        assert torch.isfinite(got).all(), (case, N, W, Ci, Co)
        err = (got - ref).abs()
        tol = 2.0 ** -8 * ref.abs() + 2e-3
        assert (err <= tol).all(), "case %d N=%d W=%d Ci=%d Co=%d border=%d: excess %.4g" % (
            case, N, W, Ci, Co, border, float((err - tol).max()))

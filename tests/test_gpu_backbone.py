"""GPU parity of the whole IR-ResNet feature extractor (stem -> residual stages -> FC -> L2 norm)
against the unfused CPU oracle (oracle/ir_resnet.py), through the reference's own API surface
(siamese.ArcFace.process / FaceModel.get_feature).

Tolerance: north_star — 512-d embeddings within 1e-3 cosine of the reference arithmetic (f32 on
CPU); the HIP path stores activations in bf16 (f16 optional) and accumulates in f32.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

COS_TOL = 1e-3


def _cos_dist(a, b):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    return 1.0 - (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))


def _pixels(n, size, seed=0):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (n, size[0], size[1], 3)).astype(np.float32)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_small_net_all_layouts(gpu, dtype):
    """units (2,2,2,2) at 32x32: every layer type, every input layout, ragged tiles everywhere."""
    from a_link_amd import weights as W
    from a_link_amd.backbone import IRBackbone
    from oracle import ir_resnet
    size = (32, 32)
    params = W.synthetic_ir_params((2, 2, 2, 2), size=size, seed=3)
    bb = IRBackbone(params, image_size=size, dtype=dtype, max_batch=8)
    x = _pixels(5, size, seed=1)
    ref = ir_resnet.embed(params, x)
    got = bb.embed(x)
    assert got.shape == (5, 512) and got.dtype == np.float32
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    d = _cos_dist(got, ref)
    assert d.max() < COS_TOL, d
    # NCHW (FaceModel.get_feature's layout) and u8 inputs give the same result bit for bit
    got_nchw = bb.embed(np.ascontiguousarray(np.transpose(x, (0, 3, 1, 2))))
    got_u8 = bb.embed(x.astype(np.uint8))
    assert np.array_equal(got, got_nchw)
    assert np.array_equal(got, got_u8)
    # batch invariance: an image embeds identically alone and inside a batch (no cross-image reduction)
    single = bb.embed(x[2:3])
    assert np.array_equal(single[0], got[2])
    # chunking (max_batch = 8): 19 images in 3 chunks equals per-image results
    x19 = _pixels(19, size, seed=7)
    g19 = bb.embed(x19)
    assert np.array_equal(g19[9], bb.embed(x19[9:10])[0])


def test_non_integer_and_out_of_range_pixels(gpu):
    """Noisy images are not clipped back to 0..255 before embedding (reference code/noise.py:44,
    code/ALINK_arc.py:161-164): negative / >255 / fractional pixels must flow through."""
    from a_link_amd import weights as W
    from a_link_amd.backbone import IRBackbone
    from oracle import ir_resnet
    size = (32, 32)
    params = W.synthetic_ir_params((1, 1, 1, 1), size=size, seed=5)
    bb = IRBackbone(params, image_size=size, max_batch=4)
    rng = np.random.default_rng(2)
    x = (_pixels(3, size, 4) + rng.normal(10, np.sqrt(10), (3, 32, 32, 3))).astype(np.float32)
    x[0, :2] = -20.0
    x[1, -2:] = 300.0
    d = _cos_dist(bb.embed(x), ir_resnet.embed(params, x))
    assert d.max() < COS_TOL, d


def test_r50_112_api_surface(gpu):
    """ResNet-50-IR at the reference's 112x112 through ArcFace.process / FaceModel.get_feature."""
    from a_link_amd import siamese, weights as W
    from oracle import ir_resnet
    fm = siamese.ArcFace((112, 112), "synthetic:r50:1", max_batch=4)
    params = W.synthetic_ir_params(W.R50_UNITS, seed=1)
    x = _pixels(3, (112, 112), seed=0)
    got = fm.process(x)
    ref = ir_resnet.embed(params, x, batch=3)
    d = _cos_dist(got, ref)
    assert d.max() < COS_TOL, d
    # the reference's per-image path: get_input (HWC->CHW) + get_feature, batch 1
    one = fm.model.get_feature(fm.model.get_input(x[1]))
    assert one.shape == (512,)
    assert np.array_equal(one, got[1])
    assert fm.process(np.zeros((0, 112, 112, 3), np.float32)).shape == (0, 512)


def test_profile_reports_every_launch(gpu):
    from a_link_amd import weights as W
    from a_link_amd.backbone import IRBackbone
    from oracle import ir_resnet
    size = (32, 32)
    units = (1, 2, 1, 1)
    params = W.synthetic_ir_params(units, size=size, seed=3)
    bb = IRBackbone(params, image_size=size, max_batch=4)
    x = torch.from_numpy(_pixels(4, size)).cuda()
    prof = bb.profile(x)
    kinds = [k for k, _, _ in prof]
    assert kinds[0] == 0 and kinds[-1] == 3 and kinds[-2] == 2
    assert kinds.count(1) == sum(units) * 2          # the four projection shortcuts ride inside their unit's conv2 launch
    total = sum(f for _, _, f in prof)                 # ... and their FLOPs are still counted
    assert abs(total / 4 - ir_resnet.flops_per_image(units, size=32)) < 1e-6 * total
    assert all(ms >= 0 for _, ms, _ in prof)
    # A/B hook: shortcuts as launches of their own (what the gradient pass and split precision use) — same embeddings
    # up to the rounding of the shortcut's output to 16 bits, which the fused form skips
    lib = gpu.load()
    lib.alink_debug_set_fuse_shortcut(0)
    try:
        unfused = IRBackbone(params, image_size=size, max_batch=4)
    finally:
        lib.alink_debug_set_fuse_shortcut(1)
    assert [k for k, _, _ in unfused.profile(x)].count(1) == sum(units) * 2 + 4
    a, b = bb.embed_device(x).cpu().numpy(), unfused.embed_device(x).cpu().numpy()
    assert _cos_dist(a, b).max() < 2e-4
    ref = ir_resnet.embed(params, x.cpu().numpy())
    assert _cos_dist(a, ref).max() <= _cos_dist(b, ref).max() * 1.5 + 1e-6


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_fused_front_is_bit_identical_to_stem_and_conv1_launches(gpu, dtype):
    """csrc/front_c64.hip: stem + stage1_unit1 conv1 in one rolling-row launch (the stem's activation stays in LDS, the
    projection shortcut reads a compact quarter-resolution copy) against the two launches it replaces — the same
    arithmetic in the same order, so the embeddings are equal bit for bit.  Every pixel layout; batches that give a
    workgroup one band (3 images), runs that cross image boundaries (83 images) and runs of whole images (300)."""
    from a_link_amd import weights as W
    from a_link_amd.backbone import IRBackbone
    units = (1, 1, 1, 1)
    params = W.synthetic_ir_params(units, seed=5, normalized=True)
    lib = gpu.load()
    bb = IRBackbone(params, dtype=dtype, max_batch=300)
    rng = np.random.default_rng(11)
    for n in (3, 83, 300):
        u8 = torch.from_numpy(rng.integers(0, 256, (n, 112, 112, 3), dtype=np.uint8)).cuda()
        for x in (u8, u8.float(), u8.float().permute(0, 3, 1, 2).contiguous()):
            prof = bb.profile(x)
            assert prof[0][0] == 1 and [k for k, _, _ in prof].count(0) == 0      # no stem launch: it rides in conv1's
            fused = bb.embed_device(x).clone()
            lib.alink_debug_set_fuse_stem(0)
            try:
                assert bb.profile(x)[0][0] == 0
                plain = bb.embed_device(x).clone()
            finally:
                lib.alink_debug_set_fuse_stem(1)
            assert torch.isfinite(fused).all()
            assert torch.equal(fused, plain), (n, x.dtype, tuple(x.shape), (fused - plain).abs().max().item())
    # FLOPs of the merged launch = stem + conv1
    from oracle import ir_resnet
    total = sum(f for _, _, f in bb.profile(u8))
    assert abs(total / 300 - ir_resnet.flops_per_image(units, size=112)) < 1e-6 * total


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_direct_stride2_kernel_with_fused_shortcut_matches_implicit_gemm(gpu, dtype):
    """csrc/conv3x3_s2c64.hip: stage1_unit1's stride-2 conv2 with the projection shortcut as two more K-steps, as a
    rolling-row kernel (input rows de-interleaved into odd / even planes in LDS, weights in registers) against the same
    launch on the implicit-GEMM kernel: the same products summed in the same K order.  With the fused front kernel (the
    shortcut reads its compact quarter-resolution copy) and without (it samples the stem's activation at stride 2)."""
    from a_link_amd import weights as W
    from a_link_amd.backbone import IRBackbone
    units = (1, 1, 1, 1)
    params = W.synthetic_ir_params(units, seed=7, normalized=True)
    lib = gpu.load()
    direct = IRBackbone(params, dtype=dtype, max_batch=300)
    lib.alink_debug_set_s2direct(0)
    try:
        igemm = IRBackbone(params, dtype=dtype, max_batch=300)
    finally:
        lib.alink_debug_set_s2direct(1)
    rng = np.random.default_rng(13)
    for n in (2, 37, 300):
        x = torch.from_numpy(rng.integers(0, 256, (n, 112, 112, 3), dtype=np.uint8)).cuda()
        for fuse_stem in (1, 0):
            lib.alink_debug_set_fuse_stem(fuse_stem)
            try:
                a, b = direct.embed_device(x).clone(), igemm.embed_device(x).clone()
            finally:
                lib.alink_debug_set_fuse_stem(1)
            assert torch.isfinite(a).all()
            assert torch.equal(a, b), (n, fuse_stem, (a - b).abs().max().item())


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_front_kernels_on_four_streams_equal_one_stream(gpu, dtype):
    """Regression for a store-data race in the first form of csrc/front_c64.hip (round 3): its quarter-resolution copy went
    out through buffer stores whose unwanted lanes carried an out-of-range offset; with launches overlapping on four streams a
    buffer_store_dwordx4 now and then sent what its data registers held some fifteen instructions later (the compiler reuses
    them at once) — one wrong pixel quad in ~10^5 images, never on one stream, so neither the bit-identity test above nor two
    full suite runs saw it.  Ordinary predicated global stores since.  A shallow net (the front is a fifth of its work) over
    2,048 images, 40 times on 4 streams, against the one-stream result: every bit equal."""
    from a_link_amd import weights as W
    from a_link_amd.backbone import IRBackbone
    p = W.synthetic_ir_params((1, 1, 1, 1), seed=1, normalized=True)
    x = torch.randint(0, 256, (2048, 112, 112, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).cuda()
    ref = IRBackbone(p, dtype=dtype, max_batch=292, streams=1, shards_per_call=1, lazy_range_check=True).embed_device(x).clone()
    bb = IRBackbone(p, dtype=dtype, max_batch=292, streams=4, lazy_range_check=True)
    for rep in range(40):
        got = bb.embed_device(x)
        torch.cuda.synchronize()
        bad = torch.nonzero((got != ref).any(1)).flatten().tolist()
        assert not bad, (dtype, rep, len(bad), bad[:16])


def test_error_paths(gpu):
    from a_link_amd import weights as W
    from a_link_amd.backbone import IRBackbone
    size = (32, 32)
    params = W.synthetic_ir_params((1, 1, 1, 1), size=size)
    bad = dict(params)
    del bad["stage2_unit1_sc_beta"]
    with pytest.raises(KeyError):
        IRBackbone(bad, image_size=size)
    bad = dict(params)
    bad["bn0_gamma"] = np.zeros(63, np.float32)
    with pytest.raises(gpu.AlinkError):
        IRBackbone(bad, image_size=size)
    bb = IRBackbone(params, image_size=size)
    with pytest.raises(ValueError):
        bb.embed(np.zeros((2, 31, 32, 3), np.float32))


def test_stream_sharding_is_invisible(gpu):
    """alink_embed splits batches >= 128 into shards on internal streams: results must not depend on
    the shard count, and must equal per-image results."""
    from a_link_amd import weights as W
    from a_link_amd.backbone import IRBackbone
    size = (32, 32)
    params = W.synthetic_ir_params((1, 1, 1, 1), size=size, seed=9)
    x = _pixels(300, size, seed=3)
    outs = []
    for shards in (1, 2, 4, 8):
        bb = IRBackbone(params, image_size=size, max_batch=300, shards_per_call=shards)
        outs.append(bb.embed(x))
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])
    # chunked over side streams (the pool-inference path): 300 images as 64-image launches on 3 streams
    bb = IRBackbone(params, image_size=size, max_batch=64, streams=3)
    xd = torch.from_numpy(x).cuda()
    for _ in range(3):                                  # repeated calls reuse workspaces / streams safely
        got = bb.embed_device(xd)
        assert np.array_equal(got.cpu().numpy(), outs[0])
    bb = IRBackbone(params, image_size=size, max_batch=8, streams=4)
    assert np.array_equal(bb.embed(x[137:138])[0], outs[0][137])


def test_headline_depth_parity_r100_and_f16_range(gpu):
    """IR-ResNet-100 at 112 x 112 — the network of the headline metric — against the unfused float32 CPU
    oracle: 1 - cos within the north-star bar of 1e-3 in bf16 (measured 6e-4 .. 8e-4 with the SURVEY §8d
    synthetic weights, whose activations grow to ~1e8 over 49 units; 1.4e-4 for IR-50).  The same growth
    leaves the float16 range, which must be reported, not returned as NaN."""
    from a_link_amd import _abi, weights as W
    from a_link_amd.backbone import IRBackbone
    from oracle import ir_resnet
    params = W.synthetic_ir_params(W.ARCH_UNITS["r100"], seed=1)
    x = np.random.default_rng(0).integers(0, 256, (4, 112, 112, 3)).astype(np.float32)
    ref = ir_resnet.embed(params, x, batch=4).astype(np.float64)
    got = IRBackbone(params, dtype="bf16", max_batch=4).embed(x).astype(np.float64)
    assert (1.0 - (got * ref).sum(1)).max() < 1e-3
    with pytest.raises(_abi.AlinkError):
        IRBackbone(params, dtype="f16", max_batch=4).embed(x)


def test_small_batch_split_k_matches_fused(gpu):
    """Small batches run their few-workgroup convolutions split over K into f32 slabs that
    conv_split_finish_kernel adds in order (csrc/backbone.hip plan_split): same embeddings as the fused
    launches up to float summation order, at IR-50 depth, for batch 1, 5 and 32."""
    from a_link_amd import _abi, weights as W
    from a_link_amd.backbone import IRBackbone
    params = W.synthetic_ir_params(W.ARCH_UNITS["r50"], seed=2)
    x = np.random.default_rng(3).integers(0, 256, (32, 112, 112, 3)).astype(np.float32)
    fused = IRBackbone(params, dtype="bf16", max_batch=32)
    split = IRBackbone(params, dtype="bf16", max_batch=32, small_batch_split=True)
    from oracle import ir_resnet
    ref = ir_resnet.embed(params, x[:2], batch=2).astype(np.float64)
    for n in (1, 5, 32):
        a = split.embed(x[:n]).astype(np.float64)
        b = fused.embed(x[:n]).astype(np.float64)
        assert np.allclose(np.linalg.norm(a, axis=1), 1.0, atol=1e-5)
        # two bf16 computations that round differently: each within ~1.4e-4 of the f32 oracle at this depth
        assert (1.0 - (a * b).sum(1)).max() < 5e-4, n
        assert (1.0 - (a[:2] * ref[:min(n, 2)]).sum(1)).max() < 1e-3, n
    # the split path did run (at 16 images; a lone image's layers take the bit-identical latency form instead, which the split
    # yields to — conv3x3_lat.hip — except the 64-channel front, which is still split)
    assert not np.array_equal(split.embed(x[:16]), fused.embed(x[:16]))
    # split precision has the same latency mode: slabs of raw accumulators, scales applied by its own finish kernel —
    # both forms at float32 accuracy, different only in summation order
    fused2 = IRBackbone(params, dtype="f16x2", max_batch=32)
    split2 = IRBackbone(params, dtype="f16x2", max_batch=32, small_batch_split=True)
    for n in (1, 5, 32):
        a, b = split2.embed(x[:n]), fused2.embed(x[:n])
        assert np.isfinite(a).all() and np.abs(a - b).max() < 5e-6, (n, np.abs(a - b).max())
        assert np.abs(a[:2] - ref[:min(n, 2)]).max() < 1e-5, n
    assert not np.array_equal(split2.embed(x[:16]), fused2.embed(x[:16]))


def _embed_in_launches(bb, xd, streams):
    """What bench.py times: the device-resident batch cut into bb.max_batch-image launches on `streams` streams."""
    bb.n_streams = streams
    return bb.embed_device(xd).cpu().numpy()


def test_benchmarked_form_unit_net_at_112(gpu):
    """The kernel forms bench.py times, end to end, on a network small enough for the CPU oracle: units (1,1,1,1)
    at 112x112 runs the linear-tile kernel at every width it serves (56 / 28 / 14 / 7: ~84 % of the benchmark's
    kernel time) — the 32x32 test nets have 16/8/4/2-wide maps and never reach it.  Batches 1 and 5 take the
    64-channel `fine` form, 292 the 128-channel form (backbone.hip: nwg128 <= 384 ? fine), 600 = 292 + 292 + 16
    on 4 streams mixes both in one call.  Checked: against the unfused f32 oracle on a subsample, and BIT-equal
    between an image embedded alone and inside each batch — i.e. across the fine <-> coarse switch."""
    from a_link_amd import weights as W
    from a_link_amd.backbone import IRBackbone
    from oracle import ir_resnet
    size = (112, 112)
    params = W.synthetic_ir_params((1, 1, 1, 1), size=size, seed=4)
    x = _pixels(600, size, seed=11)
    xd = torch.from_numpy(x).cuda()
    bb = IRBackbone(params, image_size=size, max_batch=292, streams=4)
    e600 = _embed_in_launches(bb, xd, 4)
    assert e600.shape == (600, 512) and np.isfinite(e600).all()
    probe = [0, 1, 4, 137, 291, 292, 583, 584, 599]                  # both sides of every launch boundary
    ref = ir_resnet.embed(params, x[probe])
    d = _cos_dist(e600[probe], ref)
    assert d.max() < COS_TOL, d
    singles = np.concatenate([bb.embed_device(xd[i:i + 1]).cpu().numpy() for i in probe])    # batch 1: fine form
    assert np.array_equal(singles, e600[probe])
    e5 = bb.embed_device(xd[:5]).cpu().numpy()                                                 # batch 5: fine form
    assert np.array_equal(e5, e600[:5])
    e292 = _embed_in_launches(bb, xd[:292], 1)                                                 # one 128-channel launch
    assert np.array_equal(e292, e600[:292])
    e600_1 = _embed_in_launches(bb, xd, 1)                                                     # same launches, one stream
    assert np.array_equal(e600_1, e600)
    # the threshold itself: 190 images is the last batch on the fine form at 14 wide, 196 the first on the coarse one
    for n in (190, 196, 256):
        assert np.array_equal(bb.embed_device(xd[:n]).cpu().numpy(), e600[:n]), n


def test_benchmarked_form_r50_at_112_292_images(gpu):
    """BASELINE configs[1]'s network at the launch batch bench.py uses: IR-50, 292 images, 128-channel linear
    tiles.  8 of the 292 against the f32 oracle; every one of those 8 bit-equal to its batch-1 (fine-form) embedding."""
    from a_link_amd import weights as W
    from a_link_amd.backbone import IRBackbone
    from oracle import ir_resnet
    params = W.synthetic_ir_params(W.R50_UNITS, seed=1)
    x = _pixels(292, (112, 112), seed=5)
    xd = torch.from_numpy(x).cuda()
    bb = IRBackbone(params, max_batch=292)
    e = bb.embed_device(xd).cpu().numpy()
    probe = [0, 1, 57, 145, 146, 200, 290, 291]
    ref = ir_resnet.embed(params, x[probe], batch=8)
    d = _cos_dist(e[probe], ref)
    assert d.max() < COS_TOL, d
    singles = np.concatenate([bb.embed_device(xd[i:i + 1]).cpu().numpy() for i in probe])
    assert np.array_equal(singles, e[probe])
    # and 256 images (configs[1] as worded: one 256-image batch) equal the same rows
    assert np.array_equal(bb.embed_device(xd[:256]).cpu().numpy(), e[:256])


@pytest.mark.parametrize("arch,dtype", [("r100", "bf16"), ("r100", "f16"), ("r50", "bf16")])
def test_parity_on_calibrated_weights(gpu, capsys, arch, dtype):
    """Parity where a real checkpoint lives: weights whose BatchNorm statistics match their activations
    (oracle/calibrate.py: activations O(10) instead of the ~1e8 the uncalibrated SURVEY §8d weights grow to).  There
    float16 storage works at full depth, and the bf16 margin against the 1e-3 bar is what a trained model would see
    rather than the worst case of test_headline_depth_parity_r100_and_f16_range."""
    from a_link_amd import weights as W
    from a_link_amd.backbone import IRBackbone
    from oracle import calibrate, ir_resnet
    params = calibrate.calibrated_ir_params(W.ARCH_UNITS[arch], seed=1, n_cal=16)
    x = calibrate.calibration_pixels(4, (112, 112), seed=77)          # images of the calibrated distribution
    x = np.concatenate([x, _pixels(4, (112, 112), seed=0)])           # and uniform-noise images off it
    ref = ir_resnet.embed(params, x, batch=8)
    got = IRBackbone(params, dtype=dtype, max_batch=8).embed(x)
    d = _cos_dist(got, ref)
    with capsys.disabled():
        print("\n[calibrated %s %s] 1-cos max %.2e (calibration-like images %.2e, uniform-noise images %.2e), "
              "max |activation| %.0f" % (arch, dtype, d.max(), d[:4].max(), d[4:].max(), calibrate.activation_range(params, x[:2])))
    assert d.max() < (2e-4 if dtype == "f16" else COS_TOL), d


def test_normalized_synthetic_weights_f16_and_auto_dtype(gpu, capsys):
    """weights.synthetic_ir_params(normalized=True): the SURVEY §8d draw with BatchNorm statistics set (closed-form
    moment propagation, product side, no data) to match the activations — IR-100 then stays inside float16.
    dtype="auto" picks f16 there and bf16 for the un-normalised draw (whose activations reach ~1e8)."""
    from a_link_amd import weights as W
    from a_link_amd.backbone import IRBackbone
    from oracle import ir_resnet
    x = _pixels(4, (112, 112), seed=0)
    pn = W.synthetic_ir_params(W.R100_UNITS, seed=1, normalized=True)
    ref = ir_resnet.embed(pn, x, batch=4)
    auto = IRBackbone(pn, dtype="auto", max_batch=4)
    assert auto.dtype == "f16"
    d16 = _cos_dist(auto.embed(x), ref)
    dbf = _cos_dist(IRBackbone(pn, dtype="bf16", max_batch=4).embed(x), ref)
    with capsys.disabled():
        print("\n[normalized r100] 1-cos max: f16 %.2e, bf16 %.2e" % (d16.max(), dbf.max()))
    assert d16.max() < 2e-5 and dbf.max() < COS_TOL
    ps = W.synthetic_ir_params(W.R100_UNITS, seed=1)
    assert IRBackbone(ps, dtype="auto", max_batch=4).dtype == "bf16"


def test_float32_precision_mode(gpu, capsys):
    """IRBackbone(dtype="f32"): the reference's own precision (exact-f32 MFMA GEMMs, bn1 unfused like the symbol).
    Against the f32 CPU oracle the two differ only by summation order: 1 - cos at the 1e-7 level at IR-50 / IR-100
    depth on either kind of weights, every layout, batch-invariant bit for bit."""
    from a_link_amd import _abi, weights as W
    from a_link_amd.backbone import IRBackbone
    from oracle import ir_resnet
    size = (32, 32)
    params = W.synthetic_ir_params((2, 2, 2, 2), size=size, seed=3)
    bb = IRBackbone(params, image_size=size, dtype="f32", max_batch=8)
    x = _pixels(5, size, seed=1)
    ref = ir_resnet.embed(params, x)
    got = bb.embed(x)
    assert np.abs(got - ref).max() < 5e-6 and _cos_dist(got, ref).max() < 1e-6
    assert np.array_equal(got, bb.embed(np.ascontiguousarray(np.transpose(x, (0, 3, 1, 2)))))      # NCHW
    assert np.array_equal(got, bb.embed(x.astype(np.uint8)))                                         # u8
    assert np.array_equal(bb.embed(x[2:3])[0], got[2])                                               # batch-invariant
    x19 = _pixels(19, size, seed=7)
    assert np.array_equal(bb.embed(x19)[9], bb.embed(x19[9:10])[0])                                  # chunked (max_batch 8)
    big = IRBackbone(params, image_size=size, dtype="f32", max_batch=300)
    x300 = _pixels(300, size, seed=8)
    assert np.array_equal(big.embed(x300)[[0, 150, 299]], np.concatenate([bb.embed(x300[i:i + 1]) for i in (0, 150, 299)]))
    with pytest.raises(_abi.AlinkError):
        bb.profile(torch.from_numpy(x).cuda())
    with pytest.raises(_abi.AlinkError):
        IRBackbone(params, image_size=size, dtype="f32", enable_grad=True)
    out = {}
    for arch, normalized in (("r50", False), ("r100", True), ("r100", False)):
        p = W.synthetic_ir_params(W.ARCH_UNITS[arch], seed=1, normalized=normalized)
        xi = _pixels(6, (112, 112), seed=2)
        r = ir_resnet.embed(p, xi, batch=6)
        g = IRBackbone(p, dtype="f32", max_batch=6).embed(xi)
        out[(arch, normalized)] = (float(_cos_dist(g, r).max()), float(np.abs(g - r).max()))
    with capsys.disabled():
        print("\n[f32 mode vs f32 oracle @112] " + "; ".join("%s %s: 1-cos %.1e max|d| %.1e" % (a, "normalized" if nrm else "survey", c, d)
                                                            for (a, nrm), (c, d) in out.items()))
    for (a, nrm), (c, d) in out.items():
        assert c < 2e-6 and d < (2e-5 if nrm else 2e-4), (a, nrm, c, d)


def test_split_precision_mode(gpu, capsys):
    """IRBackbone(dtype="f16x2"): f16 pairs hi + lo with power-of-two scales per tensor, three products on the f16 matrix
    cores.  Must sit where the float32 mode sits against the f32 CPU oracle (they differ from it by summation order and,
    here, 2^-22 operand rounding): max |d| a few 1e-6, 1 - cos at the 1e-7 level — on the survey weights too, whose
    activations (~1e8) no plain f16 storage holds: the calibrated scales do.  Every layout, batch-invariant bit for bit."""
    from a_link_amd import _abi, weights as W
    from a_link_amd.backbone import IRBackbone
    from oracle import ir_resnet
    size = (32, 32)
    params = W.synthetic_ir_params((2, 2, 2, 2), size=size, seed=3)
    bb = IRBackbone(params, image_size=size, dtype="f16x2", max_batch=8)
    x = _pixels(5, size, seed=1)
    ref = ir_resnet.embed(params, x)
    got = bb.embed(x)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() < 1e-5 and _cos_dist(got, ref).max() < 1e-6, (np.abs(got - ref).max(), _cos_dist(got, ref).max())
    assert np.array_equal(got, bb.embed(np.ascontiguousarray(np.transpose(x, (0, 3, 1, 2)))))      # NCHW
    assert np.array_equal(got, bb.embed(x.astype(np.uint8)))                                         # u8
    assert np.array_equal(bb.embed(x[2:3])[0], got[2])                                               # batch-invariant
    x19 = _pixels(19, size, seed=7)
    assert np.array_equal(bb.embed(x19)[9], bb.embed(x19[9:10])[0])                                  # chunked (max_batch 8)
    # the scales are powers of two: re-calibrating on other images (here: 8x brighter-than-possible ones) moves a result
    # only through lo halves that reach the f16 subnormals — far below the mode's own error
    bb.calibrate(_pixels(4, size, seed=5) * 8.0)
    drift = float(np.abs(bb.embed(x) - got).max())
    with capsys.disabled():
        print("\n[f16x2 small net] max|d| vs oracle %.2e, 1-cos %.1e; drift after re-calibration on 8x brighter images %.1e"
              % (np.abs(got - ref).max(), _cos_dist(got, ref).max(), drift))
    assert drift < 1e-6
    # noisy / fractional / out-of-range pixels keep the accuracy (the stem splits the normalised pixel too)
    xn = x + np.random.default_rng(3).normal(0, 40, x.shape).astype(np.float32)
    gn, rn = bb.embed(xn), ir_resnet.embed(params, xn)
    assert np.abs(gn - rn).max() < 1e-5, np.abs(gn - rn).max()
    with pytest.raises(_abi.AlinkError):
        IRBackbone(params, image_size=size, dtype="f16x2", enable_grad=True)
    # a batch far outside what the scales were calibrated for (pixels x 100: activations ~100x the probe images', beyond
    # the 32x headroom): the FC-finish kernel raises the range flag, the host re-calibrates on the offending batch
    # (scales only go down) and re-runs — the caller sees finite, accurate embeddings ...
    xb = x * 100.0
    fresh = IRBackbone(params, image_size=size, dtype="f16x2", max_batch=8)
    gb, rb = fresh.embed(xb), ir_resnet.embed(params, xb)
    assert np.isfinite(gb).all() and np.abs(gb - rb).max() < 1e-5, np.abs(gb - rb).max()
    assert np.abs(fresh.embed(x) - ref).max() < 1e-5           # and ordinary images still embed to f32 accuracy afterwards
    # ... or, with lazy_range_check (bench.py: one check per timed region), an error at check_range(), never silent NaNs
    lazy = IRBackbone(params, image_size=size, dtype="f16x2", max_batch=8, lazy_range_check=True)
    lazy.embed_device(torch.from_numpy(xb).cuda())
    with pytest.raises(_abi.AlinkError):
        lazy.check_range()
    lazy.embed_device(torch.from_numpy(x).cuda())
    lazy.check_range()                                          # the flag was reset by the failed check
    out = {}
    for arch, normalized in (("r50", False), ("r100", True), ("r100", False)):
        p = W.synthetic_ir_params(W.ARCH_UNITS[arch], seed=1, normalized=normalized)
        xi = _pixels(6, (112, 112), seed=2)
        r = ir_resnet.embed(p, xi, batch=6)
        g = IRBackbone(p, dtype="f16x2", max_batch=6).embed(xi)
        out[(arch, normalized)] = (float(_cos_dist(g, r).max()), float(np.abs(g - r).max()))
    with capsys.disabled():
        print("\n[f16x2 mode vs f32 oracle @112] " + "; ".join("%s %s: 1-cos %.1e max|d| %.1e" % (a, "normalized" if nrm else "survey", c, d)
                                                              for (a, nrm), (c, d) in out.items()))
    for (c, d) in out.values():
        assert c < 1e-6 and d < 2e-5, out


@pytest.mark.parametrize("dtype,fine_max", [("f16x2", -1), ("bf16", 100000), ("f16", -1)])
def test_concurrent_launches_are_bit_identical_to_serial(gpu, dtype, fine_max):
    """Regression for a write-after-read race between a K-step's last LDS reads and the next DMA into the same buffer
    (csrc/conv3x3_linear.hip wait_dma_then_barrier: the wait for the wave's own reads was missing, and the compiler
    sinks the consuming MFMAs below the barrier).  It only showed with launches overlapping on several streams: rare
    wrong 224-pixel groups, per-cent level in the register-rich 64-channel forms (forced everywhere here for bf16 via
    fine_max; split precision uses them at 7 wide by itself).  2,048 IR-50 images on 4 streams, four times, against the
    one-stream result: every bit equal."""
    from a_link_amd import weights as W
    from a_link_amd.backbone import IRBackbone
    lib = gpu.load()
    if fine_max >= 0:
        lib.alink_debug_set_fine_max(fine_max)
    try:
        p = W.synthetic_ir_params(W.R50_UNITS, seed=1, normalized=True)
        x = torch.randint(0, 256, (2048, 112, 112, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0)).cuda()
        ref = IRBackbone(p, dtype=dtype, max_batch=292, streams=1).embed_device(x).clone()
        bb = IRBackbone(p, dtype=dtype, max_batch=292, streams=4)
        for rep in range(4):
            got = bb.embed_device(x)
            torch.cuda.synchronize()
            bad = torch.nonzero((got != ref).any(1)).flatten().tolist()
            assert not bad, (dtype, rep, len(bad), bad[:16])
    finally:
        lib.alink_debug_set_fine_max(384)


def test_one_product_screening_form_of_the_split_precision_handle(gpu):
    """alink_backbone_set_products(1): X_hi W_hi alone on the split-precision handle — a SCREENING form (same weights,
    scales, workspace).  Accuracy of the f16 class (far finer than bf16) on a network whose activations plain f16 cannot
    hold (the SURVEY draw: BatchNorm statistics that do not match the activations), batch-invariant bit for bit, and the
    handle returns to the exact form afterwards."""
    from a_link_amd import _abi, weights as W
    from a_link_amd.backbone import IRBackbone
    from oracle import ir_resnet
    size = (32, 32)
    params = W.synthetic_ir_params((2, 3, 4, 2), size=size, seed=5)
    rng = np.random.default_rng(1)
    x = rng.integers(0, 256, (40, 32, 32, 3)).astype(np.float32)
    ref = ir_resnet.embed(params, x)
    bb = IRBackbone(params, image_size=size, max_batch=16, dtype="f16x2")
    exact = bb.embed(x)
    view = bb.screening_view()
    one = view.embed(x)
    again = bb.embed(x)
    assert np.array_equal(exact, again)                                   # back in the exact form
    e_exact, e_one = np.abs(exact - ref).max(), np.abs(one - ref).max()
    e_bf16 = np.abs(IRBackbone(params, image_size=size, max_batch=16, dtype="bf16").embed(x) - ref).max()
    assert e_exact < 5e-6 and 20 * e_exact < e_one < e_bf16 / 4, (e_exact, e_one, e_bf16)
    assert np.array_equal(view.embed(x[7:8]), one[7:8]) and np.array_equal(view.embed(x[:16])[3], one[3])
    with pytest.raises(_abi.AlinkError):
        IRBackbone(params, image_size=size, max_batch=16, dtype="bf16").set_products(1)
    with pytest.raises(_abi.AlinkError):
        bb.set_products(2)


@pytest.mark.parametrize("dtype", ["bf16", "f16", "f16x2"])
def test_latency_form_of_small_launches_is_bit_identical_to_the_tile_kernels(gpu, dtype):
    """conv3x3_lat.hip: launches of a handful of images run the 3x3 stride-1 convolutions one WAVE per 16-pixel x 16/32-channel
    block, operands straight from L2 by buffer loads with 18 K-sub-steps in flight, no LDS and no barrier (FaceModel.get_feature, reference
    code/face_model.py:86-93, is a batch-1 call: 2.07 -> 1.04 ms in bf16, 6.05 -> 2.28 ms in split precision).  Every output is the
    same sum in the same order, and the same epilogue operations, as in the tile kernels: embeddings of 1 / 2 / 3 / 4 / 5 images —
    both block shapes, partial pixel tiles, every border class, PReLU and residual epilogues, 28-, 14- and 7-wide layers — equal
    the tile kernels' bit for bit, equal their rows of a 292-image batch, and so does the one-product screening form.  The same
    switch covers the implicit-GEMM layers of a lone image (conv_gemm_lat_kernel: the stride-2 3x3 convolution of a stage's first
    unit, with its fused 1x1 shortcut in the 16-bit modes, and split precision's stand-alone shortcuts: conv_igemm's walk and epilogue)."""
    from a_link_amd import _abi, weights as W
    from a_link_amd.backbone import IRBackbone
    lib = _abi.load()
    params = W.synthetic_ir_params((1, 2, 2, 2), seed=9, normalized=True)
    rng = np.random.default_rng(2)
    x = rng.integers(0, 256, (300, 112, 112, 3), dtype=np.uint8)
    bb = IRBackbone(params, dtype=dtype, max_batch=292)
    big = bb.embed(x)                                              # 292 + 8 images: the tile kernels' throughput forms
    views = [("exact", bb)] + ([("one product", bb.screening_view())] if dtype == "f16x2" else [])
    try:
        for name, m in views:
            ref_big = m.embed(x) if name != "exact" else big
            for n in (1, 2, 3, 4, 5):
                lib.alink_debug_set_latency_form(1600)
                lat = m.embed(x[:n])
                lib.alink_debug_set_latency_form(0)
                tile = m.embed(x[:n])
                assert np.array_equal(lat, tile), (dtype, name, n, np.abs(lat - tile).max())
                assert np.array_equal(lat, ref_big[:n]), (dtype, name, n)
            for form in (0, 1, 2):                                 # each block shape alone, 3 images
                lib.alink_debug_set_latency_form(1600)
                lib.alink_debug_set_latency_tiles(form)
                assert np.array_equal(m.embed(x[:3]), ref_big[:3]), (dtype, name, form)
                lib.alink_debug_set_latency_tiles(-1)
    finally:
        lib.alink_debug_set_latency_form(1600)
        lib.alink_debug_set_latency_tiles(-1)


@pytest.mark.parametrize("dtype", ["bf16", "f16x2"])
def test_compile_time_epilogues_are_bit_identical_to_the_generic_one(gpu, dtype):
    """csrc/conv3x3_linear.hip: a unit's conv1 (bias + PReLU) and conv2 (bias + residual) run on instantiations whose
    epilogue is fixed at compile time — since round 5 in split precision too — against the run-time-flag epilogue
    (alink_debug_set_generic_epilogue): the same operations in the same order, so embeddings are equal bit for bit.  64- and
    128-channel forms (40 and 292 images)."""
    from a_link_amd import weights as W
    from a_link_amd.backbone import IRBackbone
    params = W.synthetic_ir_params((1, 2, 2, 1), seed=6, normalized=True)
    lib = gpu.load()
    bb = IRBackbone(params, dtype=dtype, max_batch=292)
    rng = np.random.default_rng(12)
    x = torch.from_numpy(rng.integers(0, 256, (292, 112, 112, 3), dtype=np.uint8)).cuda()
    if dtype == "f16x2":
        bb.calibrate(x[:64])
    for n in (40, 292):
        fixed = bb.embed_device(x[:n]).clone()
        lib.alink_debug_set_generic_epilogue(1)
        try:
            generic = bb.embed_device(x[:n]).clone()
        finally:
            lib.alink_debug_set_generic_epilogue(0)
        assert torch.isfinite(fixed).all() and torch.equal(fixed, generic), (dtype, n, (fixed - generic).abs().max().item())


def test_l2_epilogue_equals_sklearn_normalize(gpu):
    """The reference's literal call after the forward is sklearn.preprocessing.normalize(embedding) (code/face_model.py:92;
    sklearn IS installed here).  The product normalises in the FC-finish kernel: its embeddings must equal sklearn's
    normalisation of the oracle's RAW fc1 output (f32 mode: 2e-6 — the exact mode's bar) and be fixed points of it."""
    preprocessing = pytest.importorskip("sklearn.preprocessing")
    from a_link_amd import weights as W
    from a_link_amd.backbone import IRBackbone
    from oracle import ir_resnet
    size = (32, 32)
    params = W.synthetic_ir_params((1, 2, 1, 1), size=size, seed=3)
    x = _pixels(6, size, seed=2)
    with torch.no_grad():
        raw = ir_resnet.forward_raw(params, np.transpose(x, (0, 3, 1, 2))).numpy()
    want = preprocessing.normalize(raw)                                   # the dependency itself, on the raw features
    for dt, tol in (("f32", 2e-6), ("f16x2", 1e-5)):
        got = IRBackbone(params, image_size=size, max_batch=8, dtype=dt).embed(x)
        assert np.abs(got - want).max() < tol, (dt, np.abs(got - want).max())
        np.testing.assert_allclose(preprocessing.normalize(got), got, rtol=0, atol=2e-7)     # already unit rows
        np.testing.assert_allclose(np.linalg.norm(got.astype(np.float64), axis=1), 1.0, atol=3e-7)

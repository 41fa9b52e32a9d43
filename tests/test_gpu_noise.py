"""GPU: the perturbation stage (csrc/noise.hip) against the oracle fed from the same Philox stream,
distribution checks against the reference's np.random semantics, perturb_image / resize exactness,
and the few-pixel attack's fused device objective against the generic route."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _imgs(n, h, w, seed, integer=True):
    rng = np.random.RandomState(seed)
    x = rng.randint(0, 256, (n, h, w, 3)).astype(np.float32)
    return x if integer else (x * 0.37 + 1.25).astype(np.float32)


def test_gaussian_and_speckle_match_philox_oracle(gpu):
    from a_link_amd import noise as N
    from oracle import noise as ON
    x = _imgs(5, 13, 11, 0, integer=False)                 # 5*13*11*3 = 2145 elements: not a multiple of 4
    g = N.Gaussian(seed=42)
    s0 = g._seed
    got = g.addNoise(x, None)
    np.testing.assert_allclose(got, ON.philox_gaussian(x, s0, 10.0, 10 ** 0.5), rtol=0, atol=2e-4)
    got2 = g.addNoise(x, None)                             # next call, next stream
    assert np.abs(got2 - got).max() > 1.0
    sp = N.Speckle(seed=7)
    got = sp.addNoise(x, None)
    np.testing.assert_allclose(got, ON.philox_speckle(x, sp._seed, 15.0), rtol=0, atol=2e-4)
    # chunked calls with `offset` continue one stream
    lib = gpu.load()
    xd = torch.from_numpy(x.ravel()).cuda()
    whole, parts = torch.empty_like(xd), torch.empty_like(xd)
    gpu.check(lib.alink_noise_gaussian(gpu.ptr(xd), gpu.ptr(whole), xd.numel(), 10.0, 3.0, 5, 0, None))
    cut = 1000
    gpu.check(lib.alink_noise_gaussian(gpu.ptr(xd), gpu.ptr(parts), cut, 10.0, 3.0, 5, 0, None))
    gpu.check(lib.alink_noise_gaussian(gpu.ptr(xd[cut:]), gpu.ptr(parts[cut:]), xd.numel() - cut, 10.0, 3.0, 5, cut, None))
    torch.cuda.synchronize()
    assert torch.equal(whole, parts)


def test_gaussian_speckle_distribution(gpu):
    """np.random.normal(10, sqrt(10)) / randn()/15 (code/noise.py:41,85): moments and tails at scale."""
    from scipy import stats
    from a_link_amd import noise as N
    x = np.full((64, 112, 112, 3), 100.0, np.float32)
    d = (N.Gaussian(seed=1).addNoise(x, None) - x).ravel().astype(np.float64)
    assert abs(d.mean() - 10.0) < 0.01 and abs(d.var() - 10.0) < 0.03
    assert stats.kstest((d[:200000] - 10) / np.sqrt(10), "norm").pvalue > 1e-3
    r = ((N.Speckle(seed=2).addNoise(x, None) - x) / x).ravel().astype(np.float64) * 15
    assert abs(r.mean()) < 0.01 and abs(r.var() - 1.0) < 0.01
    assert stats.kstest(r[:200000], "norm").pvalue > 1e-3


def test_saltpepper_matches_philox_oracle_and_reference_semantics(gpu):
    from a_link_amd import noise as N
    from oracle import noise as ON
    x = _imgs(6, 112, 112, 3) + 2.0                          # no pixel is 0 or 1 beforehand
    sp = N.SaltPepper(seed=9)
    got = sp.addNoise(x, None)
    assert np.array_equal(got, ON.philox_saltpepper(x, sp._seed))
    n_salt, n_pepper = sp.counts(x.shape[1:])
    assert (n_salt, n_pepper) == (76, 76) == ON.salt_pepper_counts(x.shape[1:])
    for i in range(len(x)):
        ch = np.argwhere(got[i] != x[i])
        assert 0 < len(ch) <= 152
        assert ch[:, 0].max() <= 110 and ch[:, 1].max() <= 110 and ch[:, 2].max() <= 1     # randint(0, i - 1)
        assert set(np.unique(got[i][got[i] != x[i]])) <= {0.0, 1.0}
    with pytest.raises(ValueError):
        sp.addNoise(np.zeros((1, 1, 5, 3), np.float32), None)


def test_perlin_matches_reference_field_given_same_vectors(gpu):
    """Feed the unit vectors the reference drew (np.random.seed(4321)) and compare with its output."""
    from a_link_amd import noise as N
    from oracle import noise as ON
    with np.load(os.path.join(GOLD, "noise.npz")) as g:
        for size in (224, 150):
            np.random.seed(4321)
            vec = []
            for ns in ON.perlin_octaves(size):
                gs = int(size / ns + 1)
                phi = np.random.uniform(0, 2 * np.pi, (gs, gs)).ravel()
                vec.append(np.stack([np.cos(phi), np.sin(phi)], axis=1))
            vec = np.concatenate(vec)[None].astype(np.float32)
            p = N.Perlin(seed=0)
            x = torch.zeros((1, size, size, 3), device="cuda")
            z = p._apply(x, vectors=torch.from_numpy(vec).cuda()).cpu().numpy()[0]
            assert np.array_equal(z[..., 0], z[..., 1]) and np.array_equal(z[..., 0], z[..., 2])
            np.testing.assert_allclose(z[::7, ::5, 0], g["perlin%d_sub" % size], rtol=0, atol=2e-3)
            m = g["perlin%d_moments" % size]
            assert abs(z[..., 0].astype(np.float64).sum() - m[0]) < 1e-4 * size * size * 10
            assert abs(z[..., 0].min() - m[2]) < 2e-3 and abs(z[..., 0].max() - m[3]) < 2e-3


def test_perlin_random_vectors_and_errors(gpu):
    from a_link_amd import noise as N
    from oracle import noise as ON
    p = N.Perlin(seed=3)
    x = _imgs(3, 150, 150, 4)
    got = p.addNoise(x, None)
    nodes = sum((150 // ns + 1) ** 2 for ns in (50, 30, 15))
    vec = ON.philox_perlin_vectors(3, nodes, p._seed)
    np.testing.assert_allclose(np.linalg.norm(vec, axis=2), 1.0, atol=1e-6)
    want = ON.perlin_from_vectors(x, vec)
    np.testing.assert_allclose(got, want, rtol=0, atol=3e-3)
    with pytest.raises(ValueError):                         # the reference cannot run at 112 x 112 either
        p.addNoise(np.zeros((1, 112, 112, 3), np.float32), None)


def test_poisson_matches_philox_oracle(gpu):
    from a_link_amd import noise as N
    from oracle import noise as ON
    x = np.concatenate([_imgs(2, 12, 12, 5), _imgs(2, 12, 12, 6, integer=False)])
    x[0, :2] = 0.0
    x[1] = np.round(x[1] / 64)                              # few unique values: vals = 4, lam < 10 branch
    p = N.Poisson(seed=11)
    got = p.addNoise(x, None)
    vals = p.last_vals.cpu().numpy()
    assert np.array_equal(vals, [ON.poisson_vals(im) for im in x])
    want = ON.philox_poisson(x, p._seed)
    assert np.mean(got != want) < 1e-3                       # accept/reject flips on libm ulps are rare
    np.testing.assert_allclose(got, want, rtol=0, atol=0.5)
    with pytest.raises(ValueError):
        p.addNoise(-np.ones((1, 4, 4, 3), np.float32), None)


def test_poisson_unique_counts_on_chip(gpu):
    """vals = 2^ceil(log2(#unique)) (code/noise.py:73-74) from the on-chip counters: the 256-mark table for integer-valued
    images, the LDS hash set (2^p parts) for anything else — 112 x 112 x 3 images with every value distinct, with heavy
    duplication, with -0.0 / 0.0 and NaN patterns, a 224 x 224 image (16 parts), and counts that sit exactly on / next to a
    power of two."""
    from a_link_amd import noise as N
    from oracle import noise as ON
    rng = np.random.RandomState(3)
    per = 112 * 112 * 3
    imgs = [rng.randint(0, 256, per).astype(np.float32),                                   # integer path, 256 values
            rng.randint(0, 256, per).astype(np.float32) * 0.5 + 0.25,                        # 256 fractional values
            rng.permutation(per).astype(np.float32) * 0.00390625,                            # all distinct: 37632 -> 65536
            (rng.permutation(per) % 32768).astype(np.float32) + 0.5,                         # exactly 32768 distinct
            (rng.permutation(per) % 32769).astype(np.float32) + 0.5,                         # 32769 -> 65536
            np.where(rng.rand(per) < 0.5, 0.0, -0.0).astype(np.float32),                     # one value (0.0 == -0.0)
            rng.randint(0, 3, per).astype(np.float32)]                                       # 3 values -> 4
    x = np.stack(imgs).reshape(len(imgs), 112, 112, 3)
    p = N.Poisson(seed=1)
    p.addNoise(x, None)
    vals = p.last_vals.cpu().numpy()
    assert np.array_equal(vals, [ON.poisson_vals(im) for im in x]), (vals, [ON.poisson_vals(im) for im in x])
    big = (rng.permutation(224 * 224 * 3) % 70000).astype(np.float32).reshape(1, 224, 224, 3) * 0.001
    p.addNoise(big, None)
    assert p.last_vals.cpu().numpy()[0] == ON.poisson_vals(big[0]) == 131072.0


def test_poisson_matches_oracle_at_image_scale_lambdas(gpu):
    """The float32 PTRS sampler against its NumPy restatement at the lam the loop produces (x * 256 up to 65280, and
    x * 65536 up to 1.67e7 for a resized image): identical counts except where an acceptance test lands within ~1e-6 of
    equality."""
    from a_link_amd import noise as N
    from oracle import noise as ON
    rng = np.random.RandomState(4)
    a = rng.randint(0, 256, (1, 24, 24, 3)).astype(np.float32)                      # 256 values -> lam = x * 256
    b = (rng.permutation(24 * 24 * 3).astype(np.float32) * (255.0 / 1727)).reshape(1, 24, 24, 3)      # 1728 distinct -> vals 2048
    c = np.concatenate([a, b])
    p = N.Poisson(seed=21)
    got = p.addNoise(c, None)
    want = ON.philox_poisson(c, p._seed)
    assert np.mean(got != want) < 2e-4, np.mean(got != want)
    np.testing.assert_allclose(got, want, rtol=0, atol=1.0 / 256 + 1e-6)
    # rows of a larger batch (first_image) draw the same
    lib_rows = N.Poisson(seed=21).addNoise(c[1:], None, first_row=1)
    assert np.array_equal(lib_rows, got[1:])
    # the top of the float32 sampler's range: 33,075 distinct values -> vals = 65536, lam up to 255 x 65536 = 1.671e7 < 2^24
    d = (rng.permutation(105 * 105 * 3).astype(np.float32) * np.float32(255.0 / 33074)).reshape(1, 105, 105, 3)
    p2 = N.Poisson(seed=22)
    got2 = p2.addNoise(d, None)
    assert p2.last_vals.cpu().numpy()[0] == 65536.0
    want2 = ON.philox_poisson(d, p2._seed)
    assert np.mean(got2 != want2) < 2e-4, np.mean(got2 != want2)
    np.testing.assert_allclose(got2, want2, rtol=0, atol=1.0 / 65536 * 4 + 1e-6)
    rel = (got2.astype(np.float64) - d) / np.sqrt(np.maximum(d, 1e-3) / 65536.0)          # standardised: ~N(0, 1)
    sel = d > 1.0
    assert abs(rel[sel].mean()) < 0.03 and abs(rel[sel].std() - 1.0) < 0.03


def test_poisson_distribution_at_image_scale(gpu):
    """Poisson(x vals)/vals has mean x and variance x/vals (code/noise.py:75)."""
    from a_link_amd import noise as N
    x = np.tile(np.linspace(0, 255, 112 * 112 * 3, dtype=np.float32).reshape(1, 112, 112, 3), (32, 1, 1, 1))
    x = np.round(x * 4) / 4                                  # 1021 unique values -> vals = 1024
    p = N.Poisson(seed=5)
    y = p.addNoise(x, None).astype(np.float64)
    assert np.all(p.last_vals.cpu().numpy() == 1024.0)
    d = (y - x)
    assert abs(d.mean()) < 2e-3
    sel = x > 50
    assert abs((d[sel] ** 2 / (x[sel] / 1024.0)).mean() - 1.0) < 0.01
    assert np.all(y * 1024 == np.round(y * 1024))            # integer counts / vals


def test_perturb_image_matches_reference_golden(gpu):
    from a_link_amd import attack as A
    with np.load(os.path.join(GOLD, "noise.npz")) as g:
        got = A.perturb_image(g["perturb_xs"], g["img"])
        assert got.dtype == np.float32 and np.array_equal(got, g["perturb_out"])
        assert np.array_equal(A.perturb_image(g["perturb_xs"][0], g["img"]), g["perturb_one"])
    from oracle import noise as ON
    rng = np.random.RandomState(0)
    img = _imgs(1, 224, 112, 1)[0]
    xs = rng.rand(200, 200) * np.tile([224, 112, 256, 256, 256], 40)
    xs[:, 5:7] = xs[:, 0:2]                                   # same position twice: the later pixel wins
    assert np.array_equal(A.perturb_image(xs, img), ON.perturb_image(xs, img))
    halves = A._perturb_device(xs, torch.from_numpy(img).cuda(), True).cpu().numpy()
    full = ON.perturb_image(xs, img)
    assert np.array_equal(halves[0], full[:, :112]) and np.array_equal(halves[1], full[:, 112:])
    with pytest.raises(IndexError):
        A.perturb_image(np.array([[224.0, 0, 1, 2, 3]]), img)


def test_resize_bilinear(gpu):
    from a_link_amd import committee, noise as N
    from oracle import noise as ON
    x = _imgs(4, 150, 150, 2, integer=False)
    for size in ((32, 32), (48, 48), (224, 224), (150, 150), (40, 64)):
        got = N.resize_images(x, size)
        want = ON.resize_bilinear(x, size)
        assert got.shape == (4, size[1], size[0], 3)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-4)
    assert np.array_equal(N.resize_images(x, (150, 150)), x)
    c = np.full((2, 9, 7, 3), 3.5, np.float32)
    np.testing.assert_allclose(N.resize_images(c, (20, 31)), np.full((2, 31, 20, 3), 3.5, np.float32), rtol=2e-7)
    ramp = np.tile(np.arange(8, dtype=np.float32)[None, None, :, None], (1, 4, 1, 3))
    np.testing.assert_allclose(N.resize_images(ramp, (4, 4))[0, 0, :, 0], [0.5, 2.5, 4.5, 6.5])   # 2x: pixel-pair means
    bag = committee.Bagging([], [N.Gaussian(seed=1), N.Noise()])
    out = bag.attackModel([x[:2], x[2:]], (32, 32), None)
    assert np.asarray(out[0]).shape == (2, 2, 32, 32, 3)
    assert np.array_equal(out[1][1], N.resize_images(x[2:], (32, 32)))           # 'plain' noise = resize only


def test_pixel_attack_device_objective_equals_generic_route(gpu):
    """One DE generation's objective through the fused path (perturb -> two embed batches -> head)
    equals PredictionWrappedModel.predict on host-perturbed images; a short attack lowers 1 - P[0]."""
    from a_link_amd import attack as A, noise as N, siamese
    size = (32, 32)
    fm = siamese.ArcFace(size, "synthetic:r18:3")
    pm = siamese.SiameseNetwork((512,), "m2", 0.1, seed=4)
    wrapped = N.PredictionWrappedModel(pm, fm)
    rng = np.random.RandomState(0)
    img = _imgs(1, 64, 32, 8)[0]
    xs = rng.rand(50, 25) * np.tile([64, 32, 256, 256, 256], 5)
    sc = A._DevicePairScorer(wrapped, img)
    fused = sc.predict(xs)
    generic = wrapped.predict(A.perturb_image(xs, img))
    np.testing.assert_allclose(fused, generic, rtol=0, atol=1e-6)
    att = A.PixelAttacker(wrapped, seed=np.random.RandomState(1))
    before = wrapped.predict(img[None])[0]
    out = att.attack(img, 1, 0, pixel_count=5, dimensions=(64, 32), maxiter=4, popsize=50)
    after = wrapped.predict(out[None])[0]
    assert out.shape == img.shape and (out != img).any(axis=2).sum() <= 5
    assert after[0] >= before[0] - 1e-6
    assert abs((1 - after[0]) - att.last_result.fun) < 1e-5
    # AdversarialNoise end to end on two pairs
    adv = N.AdversarialNoise(pm, None, fm, seed=2, pixel_count=3, maxiter=2, popsize=15)
    pairs = [_imgs(2, 32, 32, 10), _imgs(2, 32, 32, 11)]
    l, r = adv.addPairNoise(pairs, np.array([[1], [0]]))
    assert np.asarray(l).shape == (2, 32, 32, 3) and np.asarray(r).shape == (2, 32, 32, 3)


def test_row_ranges_draw_what_the_whole_batch_draws(gpu):
    """One process per GPU (alink_loop with `group`): a rank perturbs rows lo : hi of the pair batch with rows=(lo, total) and
    must get, bit for bit, rows lo : hi of the whole-batch call — for every noise class, both sides of the pair, odd image
    sizes (element counts that are no multiple of 4) and an empty shard."""
    from a_link_amd import noise as N
    for size, classes in (((13, 11), ("gaussian", "speckle", "saltpepper", "poisson")), ((30, 30), ("perlin", "gaussian", "poisson"))):
        n = 7
        L = torch.from_numpy(_imgs(n, size[0], size[1], 1)).cuda()
        R = torch.from_numpy(_imgs(n, size[0], size[1], 2, integer=False)).cuda()
        for name in classes:
            whole_obj = N.get_relevant_noise(name)(seed=123)
            if name == "perlin":
                whole_obj.octaves = (lambda row: [15, 10, 5])          # 30 x 30 test images
            whole = whole_obj.addPairNoise([L, R], None)
            for cuts in ((0, 3, 7), (0, 0, 1, 7), (0, 7)):
                obj = N.get_relevant_noise(name)(seed=123)
                if name == "perlin":
                    obj.octaves = (lambda row: [15, 10, 5])
                parts = [[], []]
                for lo, hi in zip(cuts[:-1], cuts[1:]):
                    o = N.get_relevant_noise(name)(seed=123)                       # every "rank" holds the same stream state
                    if name == "perlin":
                        o.octaves = (lambda row: [15, 10, 5])
                    got = o.addPairNoise([L[lo:hi], R[lo:hi]], None, rows=(lo, n))
                    assert o.stream_state() == whole_obj.stream_state(), name      # an empty shard consumes its calls too
                    for s in (0, 1):
                        parts[s].append(got[s])
                for s in (0, 1):
                    assert torch.equal(torch.cat(parts[s]), whole[s]), (name, size, cuts, s)


def test_empty_shards_pass_through_the_perturbation_stage(gpu):
    """More ranks than pair rows: a rank's shard is empty — noise, resize and the committee's attackModel hand back empty
    containers (CUDA tensors stay CUDA tensors) and still consume their streams."""
    from a_link_amd import committee, noise as N
    e = torch.empty((0, 16, 16, 3), device="cuda")
    nz = [N.Gaussian(seed=1), N.SaltPepper(seed=2), N.Poisson(seed=3), N.Speckle(seed=4)]
    out = committee.Bagging([], nz).attackModel([e, e], (8, 8), np.zeros((0,), int), rows=(5, 5))
    assert len(out) == 2 and all(len(side) == 4 and all(len(t) == 0 for t in side) for side in out)
    assert all(z.stream_state()[1] == 2 for z in nz)
    assert len(N.resize_images(e, (8, 8))) == 0 and len(N.resize_images(np.zeros((0, 16, 16, 3), np.float32), (8, 8))) == 0


def test_pgd_step_matches_numpy_bit_for_bit(gpu):
    """alink_pgd_step (extension: FGSM / PGD) is an elementwise sign / clamp kernel: adv <- clip(clip(adv + step *
    sign(grad), clean - eps, clean + eps), lo, hi) — every float32 operation is exact or correctly rounded, so NumPy's
    float32 result is the device's, bit for bit; lengths that are no multiple of 4, zero and negative gradients, both
    step signs, no pixel clip."""
    lib = gpu.load()
    rng = np.random.RandomState(0)
    for n, step, eps, lo, hi in ((4096, 1.0, 4.0, 0.0, 255.0), (1003, -0.75, 2.5, 0.0, 255.0), (7, 2.0, 1.0, float("-inf"), float("inf")),
                                 (1, 1.0, 4.0, 0.0, 255.0)):
        clean = rng.randint(0, 256, n).astype(np.float32)
        adv = (clean + rng.uniform(-eps, eps, n)).astype(np.float32)
        grad = rng.randn(n).astype(np.float32)
        grad[::5] = 0.0
        grad[1::7] = -0.0
        want = adv + np.float32(step) * np.sign(grad).astype(np.float32)           # fmaf(step, s, a) with s in {-1, 0, 1}: exact product
        want = np.minimum(np.maximum(want, clean - np.float32(eps)), clean + np.float32(eps))
        want = np.minimum(np.maximum(want, np.float32(lo)), np.float32(hi)).astype(np.float32)
        a, c, g = torch.from_numpy(adv).cuda(), torch.from_numpy(clean).cuda(), torch.from_numpy(grad).cuda()
        gpu.check(lib.alink_pgd_step(gpu.ptr(a), gpu.ptr(c), gpu.ptr(g), n, step, eps, lo, hi, None))
        torch.cuda.synchronize()
        assert np.array_equal(a.cpu().numpy(), want), (n, step)
    # buffers that are not 16-byte aligned (a chunk of an odd-sized batch: ADVICE r4) take the scalar form
    base = torch.zeros(1003 + 3, device="cuda")
    for off in (1, 2, 3):
        n = 1000
        clean = rng.randint(0, 256, n).astype(np.float32)
        adv = (clean + rng.uniform(-3, 3, n)).astype(np.float32)
        grad = rng.randn(n).astype(np.float32)
        bufs = [torch.zeros(n + 4, device="cuda") for _ in range(3)]
        views = [b[off:off + n] for b in bufs]
        for v, h in zip(views, (adv, clean, grad)):
            v.copy_(torch.from_numpy(h))
        gpu.check(lib.alink_pgd_step(gpu.ptr(views[0]), gpu.ptr(views[1]), gpu.ptr(views[2]), n, 1.0, 3.0, 0.0, 255.0, None))
        torch.cuda.synchronize()
        want = np.minimum(np.maximum(np.minimum(np.maximum(adv + np.sign(grad).astype(np.float32), clean - np.float32(3)), clean + np.float32(3)),
                                     np.float32(0)), np.float32(255))
        assert np.array_equal(views[0].cpu().numpy(), want), off
        assert float(bufs[0][:off].abs().max()) == 0.0 and float(bufs[0][off + n:].abs().max()) == 0.0
    del base


@pytest.mark.parametrize("search", ["exact", "screen"])
def test_few_pixel_attack_rows_are_independent_of_the_batch(gpu, search):
    """noise.AdversarialNoise (code/noise.py:171-188 over code/attack.py:91-103): every pair's differential-evolution search
    has its own random stream keyed by the pair's global row, so a rank that attacks rows lo : hi finds what the
    whole-batch call finds for them (a short search here: 3 pixels, 3 generations)."""
    from a_link_amd import noise as N, siamese
    size = (32, 32)
    # search="screen": the candidates of the search are embedded by the feature model's 16-bit screening form (3x the rate)
    conv = siamese.ArcFace(size, "synthetic:r18:3", screen_dtype="f16") if search == "screen" else \
        siamese.ArcFace(size, "synthetic:r18:3", dtype="bf16", screen_dtype=None)
    student = siamese.SiameseNetwork((512,), "s", 0.1, seed=3)
    rng = np.random.RandomState(0)
    L = rng.randint(0, 256, (4,) + size + (3,)).astype(np.float32)
    R = rng.randint(0, 256, (4,) + size + (3,)).astype(np.float32)
    labels = np.array([0, 1, 1, 0])
    kw = dict(pixel_count=3, maxiter=3, popsize=30, search=search)
    whole_obj = N.AdversarialNoise(student, None, conv, seed=9, **kw)
    whole = whole_obj.addPairNoise([L, R], labels)
    if search == "screen":
        from a_link_amd import attack as A
        sc = A._DevicePairScorer(whole_obj.e2e_model, np.concatenate([L[0], R[0]]), search="screen")
        assert sc.bb is conv.screen.model and sc.bb.dtype == "f16"
    parts = [[], []]
    for lo, hi in ((0, 1), (1, 4)):
        got = N.AdversarialNoise(student, None, conv, seed=9, **kw).addPairNoise([L[lo:hi], R[lo:hi]], labels[lo:hi], rows=(lo, 4))
        for s in (0, 1):
            parts[s] += list(got[s])
    for s in (0, 1):
        assert np.array_equal(np.stack(parts[s]), np.stack(whole[s]))
    assert not np.array_equal(np.stack(whole[0]), L) or not np.array_equal(np.stack(whole[1]), R)      # pixels were written


def test_poisson_unique_count_survives_an_overfull_hash_part(gpu):
    """The general path of unique_count_kernel cuts the key space into 2^p parts by the top bits of a multiplicative hash so
    that a part fits its 32,768-slot LDS table.  An adversarial image — 37,632 distinct values whose hashes all fall into ONE
    of the four parts — overfills it: the kernel must neither hang nor miscount (it gives up on the table after a full
    round of probes and counts by brute force)."""
    from a_link_amd import noise as N
    rng = np.random.RandomState(7)
    per = 112 * 112 * 3
    vals = np.unique(rng.uniform(1.0, 200.0, 400000).astype(np.float32))
    h = (vals.view(np.uint32).astype(np.uint64) * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    bad = vals[(h >> np.uint64(30)) == 0][:per]
    assert len(bad) == per
    x = rng.permutation(bad).reshape(1, 112, 112, 3)
    p = N.Poisson(seed=2)
    p.addNoise(x, None)
    assert p.last_vals.cpu().numpy()[0] == 65536.0               # 37,632 distinct -> 2^16
    half = bad.copy()
    half[per // 2:] = half[:per - per // 2]                        # 18,816 distinct, all in one part: fits, counted by the table
    p.addNoise(rng.permutation(half).reshape(1, 112, 112, 3), None)
    assert p.last_vals.cpu().numpy()[0] == 32768.0


def test_poisson_unique_count_of_short_mantissa_images_is_exact_and_fast(gpu):
    """pixels with short mantissas — quarter steps, values rounded through bfloat16 — share the low bits of the
    multiplicative hash; a table slot taken from those bits put thousands of distinct values into a handful of home slots
    (linear probing then costs ~distinct / 2 contended atomicCAS per element: seconds per image).  The slot now comes from the
    high bits: the count stays exact and 32 such images take well under a second."""
    import time
    from a_link_amd import noise as N
    rng = np.random.RandomState(3)
    shape = (32, 112, 112, 3)
    quarter = (rng.randint(0, 4 * 255, shape) / 4.0).astype(np.float32)                       # ~1020 distinct quarter steps
    bf = torch.from_numpy(rng.uniform(0, 255, shape).astype(np.float32)).to(torch.bfloat16).float().numpy()      # bf16-rounded
    for x in (quarter, bf):
        want = np.array([2.0 ** np.ceil(np.log2(len(np.unique(im)))) for im in x])
        p = N.Poisson(seed=2)
        xd = torch.from_numpy(x).cuda()
        p.addNoise(xd[:2], None)                                                            # warm-up
        torch.cuda.synchronize()
        t = time.perf_counter()
        p.addNoise(xd, None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        assert np.array_equal(p.last_vals.cpu().numpy(), want)
        assert dt < 1.0, "32 short-mantissa images took %.2f s" % dt


def test_pool_scale_batches_take_several_launches(gpu):
    """The sampling kernels carry the image index in grid.y (at most 65,535): a batch of 70,000 (tiny) images — a pool-scale
    call — must go through, and equal the same images perturbed as two row ranges."""
    from a_link_amd import noise as N
    rng = np.random.RandomState(1)
    x = torch.from_numpy(rng.randint(0, 256, (70000, 2, 2, 3)).astype(np.float32)).cuda()
    whole = N.Poisson(seed=4).addNoise(x, None)
    a = N.Poisson(seed=4).addNoise(x[:40000], None, first_row=0)
    b = N.Poisson(seed=4).addNoise(x[40000:], None, first_row=40000)
    assert torch.equal(whole, torch.cat([a, b])) and torch.isfinite(whole).all()
    assert float((whole - x).abs().max()) > 0


@pytest.mark.parametrize("search", ["exact", "screen", "screen_f16x2"])
def test_lockstep_attack_equals_the_sequential_attack(gpu, search):
    """PixelAttacker.attack_all (code/attack.py:91-103) advances K pairs' differential-evolution searches together — one
    perturb + one backbone launch chain + one scoring launch for the candidates of all of them, the success test
    (code/attack.py:47-63) read off the generation's own scores instead of a batch-1 forward.  Attacked images, best
    parameters, energies and generation counts must equal, bit for bit, what the one-pair-after-another form (lockstep=0:
    solve() with a real callback forward per generation) finds, for K in {1, 4, 16}, with searches that stop after the first
    generation, in the middle and at maxiter."""
    from a_link_amd import attack as A, noise as N, siamese
    size = (32, 32)
    if search == "exact":
        conv = siamese.ArcFace(size, "synthetic:r18:3", dtype="bf16", screen_dtype=None)
    elif search == "screen":
        conv = siamese.ArcFace(size, "synthetic:r18:3", screen_dtype="f16")
    else:                                               # the one-product form of the split-precision handle itself
        conv = siamese.ArcFace(size, "synthetic:r18:3", screen_dtype="f16x2/1")
    student = siamese.SiameseNetwork((512,), "s", 0.1, seed=3)
    ws = student.siamese_net.get_weights()
    ws[4] = ws[4] * np.float32(6.0)                     # spread the scores: some pairs flip within a few generations
    student.siamese_net.set_weights(ws)
    wrapped = N.PredictionWrappedModel(student, conv)
    n = 7
    rng = np.random.RandomState(5)
    imgs = [rng.randint(0, 256, (64, 32, 3)).astype(np.float32) for _ in range(n)]
    clean = wrapped.predict(np.stack(imgs))
    # targets: pairs 0, 1 ask for the class the clean pair already has (success after generation 1), the rest for the other one
    tcs = [int(np.argmax(clean[i])) if i < 2 else 1 - int(np.argmax(clean[i])) for i in range(n)]
    targets = [[1 - t, t] for t in tcs]
    seeds = [100 + 7 * i for i in range(n)]
    kw = dict(dimensions=(64, 32), pixel_count=3, maxiter=6, popsize=30, seeds=seeds)
    mode = "screen" if search != "exact" else "exact"
    seq = A.PixelAttacker(wrapped, search=mode)
    want, want_res = [], []
    for i in range(n):
        want.append(seq.attack(imgs[i], 1 - tcs[i], tcs[i], 3, (64, 32), maxiter=6, popsize=30, seed=seeds[i]))
        want_res.append(seq.last_result)
    nits = [int(r.nit) for r in want_res]
    assert min(nits) == 1 and max(nits) == 6, nits                          # an early stop and a search that runs out
    assert np.array_equal(np.stack(seq.attack_all(imgs, targets, lockstep=0, **kw)), np.stack(want))
    for K in (1, 4, 16):
        att = A.PixelAttacker(wrapped, search=mode, lockstep=K)
        got = att.attack_all(imgs, targets, **kw)
        assert np.array_equal(np.stack(got), np.stack(want)), (search, K)
        for r, w in zip(att.last_results, want_res):
            assert np.array_equal(r.x, w.x) and r.fun == w.fun and r.nit == w.nit and r.nfev == w.nfev and r.message == w.message, (search, K)
    # early_stop=False (a timing aid): every search runs to maxiter
    att = A.PixelAttacker(wrapped, search=mode, lockstep=4)
    att.attack_all(imgs, targets, early_stop=False, **kw)
    assert [int(r.nit) for r in att.last_results] == [6] * n
    # stacked pairs handed over as ONE device tensor come back as one (no trip through the host), the same images
    dev_out = A.PixelAttacker(wrapped, search=mode, lockstep=4).attack_all(torch.from_numpy(np.stack(imgs)).cuda(), targets, **kw)
    assert dev_out.is_cuda and np.array_equal(dev_out.cpu().numpy(), np.stack(want))
    # through the noise class, with its per-row seeds: rows of a shard equal rows of the whole, whatever the lock-step width
    L = np.stack([im[:32] for im in imgs])
    R = np.stack([im[32:] for im in imgs])
    labels = np.array(tcs)
    nk = dict(pixel_count=3, maxiter=3, popsize=30, search=mode)
    whole = N.AdversarialNoise(student, None, conv, seed=9, lockstep=0, **nk).addPairNoise([L, R], labels)
    for K, cuts in ((32, (0, 7)), (3, (0, 2, 7))):
        parts = [[], []]
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            got = N.AdversarialNoise(student, None, conv, seed=9, lockstep=K, **nk).addPairNoise([L[lo:hi], R[lo:hi]], labels[lo:hi], rows=(lo, n))
            for s_ in (0, 1):
                parts[s_] += list(got[s_])
        for s_ in (0, 1):
            assert np.array_equal(np.stack(parts[s_]), np.stack(whole[s_])), (search, K)
    got_dev = N.AdversarialNoise(student, None, conv, seed=9, lockstep=32, **nk).addPairNoise([torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()], labels)
    for s_ in (0, 1):                                      # device tensors in -> device tensors out, the same rows
        assert got_dev[s_].is_cuda and np.array_equal(got_dev[s_].cpu().numpy(), np.stack(whole[s_])), search


def test_lockstep_attack_survives_a_range_exit_of_the_exact_mode(gpu):
    """The exact (split-precision) backbone re-calibrates itself when a batch leaves its calibrated range; under the lock-step
    engine its per-call range synchronisation is deferred to one flag read per step, and a raised flag makes the engine
    re-calibrate on the images in flight (scales only go down) and run the step again.  Given scales 2^7 too fine and
    attacked with full-range pairs: the searches must complete with finite energies, the scales must have moved, and a second
    run under the now-sufficient scales must equal the one-pair-at-a-time form bit for bit."""
    from a_link_amd import attack as A, noise as N, siamese
    size = (32, 32)
    conv = siamese.ArcFace(size, "synthetic:r18:3", screen_dtype=None)            # f16x2, the API default
    bb = conv.model.model
    assert bb.dtype == "f16x2"
    # scales 2^7 too fine for real images (as if calibrated on images with activations 128 x smaller: the headroom is 32 x)
    st0 = bb.state()
    bb.load_state(dict(st0, scale_exponents=[e + 7 for e in st0["scale_exponents"]]))
    before = bb.state()["scale_exponents"]
    student = siamese.SiameseNetwork((512,), "s", 0.1, seed=3)
    wrapped = N.PredictionWrappedModel(student, conv)
    rng = np.random.RandomState(8)
    imgs = [rng.randint(0, 256, (64, 32, 3)).astype(np.float32) for _ in range(5)]
    targets = [[0, 1]] * 5
    kw = dict(dimensions=(64, 32), pixel_count=3, maxiter=3, popsize=30, seeds=[1, 2, 3, 4, 5], early_stop=False)
    att = A.PixelAttacker(wrapped, lockstep=4)
    out = att.attack_all(imgs, targets, **kw)
    after = bb.state()["scale_exponents"]
    assert after != before and all(a <= b for a, b in zip(after, before)), "the engine should have re-calibrated (scales only go down)"
    assert all(np.isfinite(r.fun) for r in att.last_results) and np.isfinite(np.stack(out)).all()
    assert not bb.lazy_range_check                                                   # the deferred check was handed back
    again = A.PixelAttacker(wrapped, lockstep=4).attack_all(imgs, targets, **kw)
    seq = A.PixelAttacker(wrapped, lockstep=0)
    seq.attack_success = lambda *a_, **k_: None
    one = seq.attack_all(imgs, targets, **{k_: v for k_, v in kw.items() if k_ != "early_stop"})
    assert bb.state()["scale_exponents"] == after
    assert np.array_equal(np.stack(again), np.stack(one))

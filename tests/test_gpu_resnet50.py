"""GPU: the VGGFace2 ResNet-50 feature model (csrc/resnet50.hip, siamese.RESNET50) against the
unfused torch-CPU oracle with identical synthetic weights; preprocess fold, Keras .h5 weights,
chunking."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cos_dist(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return 1.0 - (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))


@pytest.fixture(scope="module")
def net(gpu):
    from a_link_amd import resnet50 as R, siamese
    params = R.synthetic_params(3)
    return siamese.RESNET50((224, 224), weights=params, max_batch=3, dtype="bf16"), params


def test_features_match_oracle(net):
    from oracle import vgg_resnet50 as O
    m, params = net
    x = np.random.default_rng(0).integers(0, 256, (5, 224, 224, 3)).astype(np.float32)
    x[4] = 255.0 - x[0]
    got = m.process(x)                                  # 5 images through chunks of 3 + 2
    want = O.process(params, x)
    assert got.shape == (5, 2048) and got.dtype == np.float32
    assert _cos_dist(got, want).max() < 1e-3            # bf16 storage, f32 accumulation vs f32 oracle
    rel = np.linalg.norm(got - want, axis=1) / np.linalg.norm(want, axis=1)
    assert rel.max() < 3e-2, rel
    # f16 storage is tighter on the same weights
    from a_link_amd import siamese
    m16 = siamese.RESNET50((224, 224), weights=params, dtype="f16")
    g16 = m16.process(x[:2] / 8.0 + 100.0)
    w16 = O.process(params, x[:2] / 8.0 + 100.0)
    assert _cos_dist(g16, w16).max() < 2e-5
    # split precision (the default of siamese.RESNET50): float32 accuracy — f16 pairs, three products on the f16 matrix
    # cores, scales calibrated per tensor (csrc/resnet50.hip: stem7_x2 / maxpool3s2_x2 / avgpool_x2 + the SP conv forms)
    mx = siamese.RESNET50((224, 224), weights=params, max_batch=3)
    assert mx.model.dtype == "f16x2"
    gx = mx.process(x)
    relx = np.abs(gx - want).max() / np.abs(want).max()
    print("VGGFace2 ResNet-50 f16x2 vs f32 oracle: max |d| / max |feature| = %.2e, 1 - cos %.1e" % (relx, _cos_dist(gx, want).max()))
    assert _cos_dist(gx, want).max() < 1e-9 and relx < 2e-5, (relx, _cos_dist(gx, want).max())
    assert np.array_equal(mx.process(x[1:2])[0], gx[1])            # batch-invariant bit for bit


def test_preprocess_fold_and_api(net):
    from oracle import vgg_resnet50 as O
    m, params = net
    x = np.random.default_rng(1).integers(0, 256, (2, 224, 224, 3)).astype(np.float32)
    pre = m.preprocess(x)
    assert np.array_equal(pre, O.preprocess_input_v2(x)) and pre is not x and x.max() > 200   # np.copy semantics
    a = m.process(x)                                    # flip + means inside the stem loader
    b = m.model.predict(pre, batch_size=128)            # the reference's call: predict(preprocess(X))
    assert np.array_equal(a, b)
    xd = torch.from_numpy(x).cuda()
    assert np.array_equal(m.model.embed_device(xd).cpu().numpy(), a)
    assert m.process(np.zeros((0, 224, 224, 3), np.float32)).shape == (0, 2048)
    with pytest.raises(ValueError):
        m.process(np.zeros((1, 112, 112, 3), np.float32))
    from a_link_amd import _abi, siamese
    with pytest.raises(_abi.AlinkError):
        siamese.RESNET50((112, 112))                    # avg_pool((7,7)) needs the 224 x 224 geometry


def test_keras_weight_file_roundtrip(net, tmp_path):
    from a_link_amd import resnet50 as R, siamese
    m, params = net
    path = str(tmp_path / "rcmalli_vggface_tf_notop_resnet50.h5")
    R.save_keras_h5(path, params)
    back = R.load_keras_h5(path)
    assert set(back) == set(params) and all(np.array_equal(back[k], params[k]) for k in params)
    m2 = siamese.RESNET50((224, 224), weights=path, dtype="bf16")
    x = np.random.default_rng(2).integers(0, 256, (2, 224, 224, 3)).astype(np.float32)
    assert np.array_equal(m2.process(x), m.process(x))
    bad = dict(params)
    del bad["conv3_1_1x1_proj/kernel"]
    with pytest.raises(KeyError):
        siamese.RESNET50((224, 224), weights=bad)


def test_alink_py_shape_pipeline(net):
    """ALINK.py's shapes: RESNET50 features (2048) -> SiameseNetwork((2048,)) -> Bagging mean."""
    from a_link_amd import committee, siamese
    m, _ = net
    x = np.random.default_rng(5).integers(0, 256, (4, 224, 224, 3)).astype(np.float32)
    f = m.process(x)
    f = f / np.linalg.norm(f, axis=1, keepdims=True)
    heads = [siamese.SiameseNetwork((2048,), "e%d" % i, 0.1, seed=i) for i in range(2)]
    p = committee.Bagging(heads, []).predict([f[:2], f[2:]])
    assert p.shape == (2, 2) and np.allclose(p.sum(1), 1.0, atol=1e-6)


@pytest.mark.parametrize("dtype", ["f16x2", "bf16", "f16x2+bf16"])
def test_alink_py_iteration_selection_through_resnet50_features(gpu, capsys, dtype):
    """One A-LINK iteration the way code/ALINK.py runs it — VGGFace2 ResNet-50 features (code/ALINK.py:67), an ensemble
    of two pair heads on 2048-d features plus the disguised-faces head on two noisy copies, the selection rule of
    code/ALINK.py:170-201 (column 1) and code/ALINK_arc.py:167-198 (column 0) — against the oracle's query sets on the
    f32 torch-CPU ResNet-50's features of the same pixels.  Heads are trained oracle-side on other identities so that
    probabilities spread over (0, 1).  Split precision (the default of siamese.RESNET50) must reproduce both query sets;
    bf16 storage is measured beside it (screening); "f16x2+bf16" = SCREEN-THEN-SETTLE (a-link_amd/settle.py): clean features
    exact, the noisy copies through the bf16 screening handle (RESNET50(screen_dtype="bf16")), only the images of pairs near a
    cut re-embedded in split precision — must reproduce both query sets too."""
    from a_link_amd import committee, pairs, resnet50 as R, selection, siamese
    from oracle import al_logic as OA
    from oracle import vgg_resnet50 as O
    import _synth
    size = (224, 224)
    params = R.synthetic_params(3)

    def feats_o(x):
        f = O.process(params, x)
        return (f / np.linalg.norm(f, axis=1, keepdims=True)).astype(np.float32)

    from oracle import siamese_head as OH

    class _Head(object):                 # an oracle head with fixed weights (predict / get_weights: what the test needs)
        def __init__(self, ws):
            self.ws = ws

        def predict(self, X):
            return OH.forward(self.ws, X[0], X[1])

        def get_weights(self):
            return self.ws

    def spread_head(seed, L, R):
        """glorot head whose last layer is rescaled so that probabilities run from ~0.12 to ~0.88 over these pairs (a
        fresh head on the features of a random-weight ResNet-50 puts every pair at 0.65 +- 0.01): the selection rule then
        has cuts inside the data, and the gain (~1e3) makes every probability a sensitive function of the features."""
        ws = OH.init_weights(2048, seed=seed)
        _, (d, z1, a1, z2, a2) = OH.forward(ws, L, R, cache=True)
        t = (a2 @ (ws[4][:, 1] - ws[4][:, 0])).astype(np.float64)
        lo, hi = np.percentile(t, [1, 99])
        gain = np.float32(4.0 / max(hi - lo, 1e-12))
        ws[4] = (ws[4] * gain).astype(np.float32)
        ws[5] = np.array([0, np.float32(-gain * 0.5 * (lo + hi))], np.float32)
        return _Head(ws)

    n_plain, n_dig = [2, 1, 2, 2, 1, 2, 2, 2], [2, 3, 2, 1, 2, 2, 3, 2]
    te = _synth.identities(8, [a + b for a, b in zip(n_plain, n_dig)], size, seed=7, var=40.0)
    uniq = _synth.unique_rows(te, n_plain)
    li, ri, y = pairs.createMiniBatchIndices(n_plain, n_dig)
    rng = np.random.default_rng(5)
    noises = [uniq + rng.normal(10, np.sqrt(10), uniq.shape).astype(np.float32),
              uniq + uniq * (rng.normal(0, 1, uniq.shape).astype(np.float32) / 15)]
    # oracle
    Eo = feats_o(uniq)
    m1 = [spread_head(1, Eo[li], Eo[ri]), spread_head(2, Eo[li], Eo[ri])]
    Eno = [feats_o(nz) for nz in noises]
    m2 = spread_head(3, Eno[0][li], Eno[0][ri])
    ens_o = OA.bagging_predict([m.predict([Eo[li], Eo[ri]]) for m in m1])
    dis_o = [m2.predict([En[li], En[ri]]) for En in Eno]
    # device
    settled_mode = dtype == "f16x2+bf16"
    fm = siamese.RESNET50(size, weights=params, dtype="f16x2" if settled_mode else dtype, max_batch=32,
                          screen_dtype="bf16" if settled_mode else None)

    def feats_g(x, screen=False):
        f = fm.process_screen(x) if screen else fm.process(x)
        return (f / np.linalg.norm(f, axis=1, keepdims=True)).astype(np.float32)

    heads = []
    for i, om in enumerate(m1 + [m2]):
        net = siamese.SiameseNetwork((2048,), "r%d" % i, 0.1, seed=i)
        net.siamese_net.set_weights(om.get_weights())
        heads.append(net)
    E = feats_g(uniq)
    ens = committee.Bagging(heads[:-1], []).predict_indexed(E, E, li, ri).cpu().numpy()
    En_g = [feats_g(nz, screen=settled_mode) for nz in noises]
    dis = [heads[-1].siamese_net.predict_device(En, En, li, ri).cpu().numpy() for En in En_g]
    if settled_mode:
        from a_link_amd import settle
        done = [np.zeros(len(uniq), bool) for _ in noises]

        def settle_fn(k, idx):                   # noisy copies are per unique IMAGE here: a pair's two images, exactly, once
            need = np.unique(np.concatenate([li[idx], ri[idx]]))
            need = need[~done[k][need]]
            if len(need):
                En_g[k][need] = feats_g(noises[k][need])
                done[k][need] = True
            return heads[-1].siamese_net.predict_device(En_g[k], En_g[k], li[idx], ri[idx]).cpu().numpy()
        screened = [np.array(d, copy=True) for d in dis]
        for col in (0, 1):
            for k in range(len(noises)):
                done[k][:] = False
            q, active, labels, _, _, info = settle.select_queries_settled(ens, screened, y, settle_fn, col=col, disparity_ratio=0.25, eps=0.05)
            qs, act_o = OA.select_queries(ens_o, dis_o, y, col, 0.25, 0.05)
            q_scr = selection.select_queries(ens, screened, y, col=col, disparity_ratio=0.25, eps=0.05)[0]
            with capsys.disabled():
                print("\n[ALINK.py shape, VGGFace2 ResNet-50 screen-then-settle (bf16 screening), column %d] oracle queries %d; screening alone "
                      "differs in %d; settled differs in %d; (pair, noise) rows settled %.1f %%" % (col, len(qs), len(set(q_scr) ^ qs),
                                                                                                  len(set(q) ^ qs), 100 * info["fraction_settled"]))
            assert set(q) == qs and active == act_o, (col, sorted(set(q) ^ qs))
        return
    d_ens = float(np.abs(ens - ens_o).max())
    d_dis = float(max(np.abs(a - b).max() for a, b in zip(dis, dis_o)))
    pct = np.percentile(ens_o[:, 0], [1, 99])
    assert pct[0] < 0.3 and pct[1] > 0.7, pct                      # spread, not 0.5 +- 0.02
    diffs = []
    for col in (0, 1):
        q, active, labels = selection.select_queries(ens, dis, y, col=col, disparity_ratio=0.25, eps=0.05)
        qs, act_o = OA.select_queries(ens_o, dis_o, y, col, 0.25, 0.05)
        diffs.append((len(qs), len(set(q) ^ qs)))
        if dtype == "f16x2":
            assert set(q) == qs and active == act_o, (col, sorted(set(q) ^ qs))
    with capsys.disabled():
        print("\n[ALINK.py shape, VGGFace2 ResNet-50 %s @224, P=%d] max|d ens|=%.2e max|d dis|=%.2e  query sets (size, differing): "
              "column 0 %s, column 1 %s" % (dtype, len(li), d_ens, d_dis, diffs[0], diffs[1]))
    assert max(d[0] for d in diffs) >= 10
    assert d_ens < (2e-5 if dtype == "f16x2" else 0.1), d_ens


def test_split_precision_overflow_is_reported_not_hidden(gpu):
    """16-bit float storage that leaves its range must never come back as plausible finite features: in split precision a
    value beyond 65504 stores as (inf, -inf), the next convolution makes NaN of it — and a ReLU written as fmaxf(v, 0) would
    turn that NaN into 0 (the IR backbone's PReLU is a select and keeps it).  Scales calibrated on images 200x darker than
    the batch (the headroom is 32x): the flag must go up (the ReLU / max-pool keep NaN), the network re-calibrates on the
    batch and re-runs, and the features equal those of a handle calibrated on such images from the start.  (Pixels
    themselves must stay within +-1023: the stem's loader keeps them as f16 x 2^6.)"""
    from a_link_amd.resnet50 import VGGResNet50
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (4, 224, 224, 3)).astype(np.float32) - 128.0          # already-preprocessed pixels (no mean to subtract)
    big = x * 7.0                                                                   # +-900
    m = VGGResNet50(dtype="f16x2", max_batch=4, seed=1)
    m.calibrate(x * 0.01, preprocessed=True)                                        # +-1.3: 700x below the batch
    e0 = list(m.state()["scale_exponents"])
    f = m.predict(big, preprocessed=True)
    assert np.isfinite(f).all()
    assert list(m.state()["scale_exponents"]) != e0, "the overflow went unnoticed: no re-calibration happened"
    m2 = VGGResNet50(dtype="f16x2", max_batch=4, seed=1)
    m2.calibrate(big, preprocessed=True)
    f2 = m2.predict(big, preprocessed=True)
    scale = np.abs(f2).max()
    assert np.abs(f - f2).max() < 2e-6 * scale, (np.abs(f - f2).max(), scale)
    # plain f16 cannot re-calibrate: it must raise, not return zeros where NaN was
    from a_link_amd import _abi
    m3 = VGGResNet50(dtype="f16", max_batch=4, seed=1)
    try:
        f3 = m3.predict(big, preprocessed=True)
        assert np.isfinite(f3).all() and np.abs(f3).max() > 0
    except _abi.AlinkError:
        pass

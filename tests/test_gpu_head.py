"""GPU parity of the siamese head (head.hip) against the NumPy oracle (oracle/siamese_head.py):
forward (f32 MFMA), pair gather, committee mean, train_on_batch / test_on_batch, fit()."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _data(n, d, seed):
    rng = np.random.RandomState(seed)
    L = rng.randn(n, d).astype(np.float32)
    R = rng.randn(n, d).astype(np.float32)
    L /= np.linalg.norm(L, axis=1, keepdims=True)
    R /= np.linalg.norm(R, axis=1, keepdims=True)
    return L, R


def _pair(d=512, h1=512, h2=64, seed=11, lr=0.1):
    from a_link_amd.head import DenseHead
    from oracle import siamese_head as O
    o = O.HeadModel(d, h1, h2, lr=lr, seed=seed)
    # make the biases non-zero so the bias path is exercised
    rng = np.random.RandomState(seed + 1)
    ws = o.get_weights()
    for i in (1, 3, 5):
        ws[i] = (rng.randn(*ws[i].shape) * 0.1).astype(np.float32)
    o.set_weights(ws)
    g = DenseHead(d, h1, h2, lr=lr, seed=seed)
    g.set_weights(ws)
    return g, o


@pytest.mark.parametrize("d,h1,h2", [(512, 512, 64), (2048, 512, 64), (2048, 128, 32), (64, 256, 64)])
def test_forward_matches_oracle(gpu, d, h1, h2):
    g, o = _pair(d, h1, h2)
    for n in (1, 31, 32, 33, 1000):
        L, R = _data(n, d, n)
        got = g.predict([L, R])
        ref = o.predict([L, R])
        assert got.shape == (n, 2)
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)
        np.testing.assert_allclose(got.sum(1), 1.0, atol=1e-6)


def test_get_set_weights_roundtrip(gpu):
    g, o = _pair()
    for a, b in zip(g.get_weights(), o.get_weights()):
        assert np.array_equal(a, b)


def test_indexed_gather_equals_materialised_pairs(gpu):
    g, o = _pair()
    E, _ = _data(300, 512, 5)
    rng = np.random.RandomState(0)
    li = rng.randint(0, 300, 5000).astype(np.int32)
    ri = rng.randint(0, 300, 5000).astype(np.int32)
    a = g.predict_device(E, E, li, ri).cpu().numpy()
    b = g.predict([E[li], E[ri]])
    assert np.array_equal(a, b)
    np.testing.assert_allclose(a, o.predict([E[li], E[ri]]), atol=2e-6)


def test_committee_mean(gpu):
    from a_link_amd import committee, siamese
    from oracle import siamese_head as O
    members = [siamese.SiameseNetwork((512,), "m%d" % i, 0.1, seed=i) for i in range(3)]
    L, R = _data(777, 512, 3)
    bag = committee.Bagging(members, [])
    got = bag.predict([L, R])
    each = [m.predict([L, R]) for m in members]
    ref = O.bagging_predict(each)           # the reference's own host arithmetic on member outputs
    np.testing.assert_allclose(got, ref, atol=1e-7)
    oracle_members = []
    for m in members:
        om = O.HeadModel(512)
        om.set_weights(m.siamese_net.get_weights())
        oracle_members.append(om.predict([L, R]))
    np.testing.assert_allclose(got, O.bagging_predict(oracle_members), atol=2e-6)


def test_train_on_batch_and_eval(gpu):
    from oracle import siamese_head as O
    g, o = _pair(lr=0.1)
    rng = np.random.RandomState(4)
    for step in range(5):
        n = [16, 13, 16, 7, 1][step]
        L, R = _data(n, 512, 100 + step)
        y = O.to_categorical(rng.randint(0, 2, n))
        cw = None if step % 2 == 0 else {0: 1.0 / 3, 1: 2.0 / 3}
        mg = g.train_on_batch([L, R], y, class_weight=cw)
        mo = o.train_on_batch([L, R], y, class_weight=cw)
        np.testing.assert_allclose(mg, mo, rtol=2e-5, atol=1e-6)
        for a, b in zip(g.get_weights(), o.get_weights()):
            np.testing.assert_allclose(a, b, rtol=0, atol=3e-6)
        tg = g.test_on_batch([L, R], y)
        to = o.test_on_batch([L, R], y)
        np.testing.assert_allclose(tg, to, rtol=2e-5, atol=1e-6)


def test_finetune_matches_keras_semantics(gpu):
    """SiameseNetwork.finetune (reference code/siamese.py:52-58): to_categorical, fit(bs=16, epochs=3,
    validation_split=.2): same np.random stream -> same shuffles -> same weights as the oracle."""
    from a_link_amd import siamese
    from oracle import siamese_head as O
    net = siamese.SiameseNetwork((512,), "ft", 0.1, seed=21)
    o = O.HeadModel(512, lr=0.1, seed=21)
    o.set_weights(net.siamese_net.get_weights())
    L, R = _data(150, 512, 9)
    Y = (np.random.RandomState(1).rand(150, 1) > 0.5).astype(int)
    np.random.seed(77)
    hg = net.finetune([L, R], Y, 3, 16, verbose=0)
    np.random.seed(77)
    ho = O.finetune(o, [L, R], Y, 3, 16)
    for k in ("loss", "acc", "val_loss", "val_acc"):
        np.testing.assert_allclose(hg[k], ho[k], rtol=1e-4, atol=1e-5)
    for a, b in zip(net.siamese_net.get_weights(), o.get_weights()):
        np.testing.assert_allclose(a, b, atol=2e-5)
    np.testing.assert_allclose(net.predict([L, R]), o.predict([L, R]), atol=1e-5)


def test_save_load_roundtrip(gpu, tmp_path):
    from a_link_amd import siamese
    a = siamese.SiameseNetwork((512,), str(tmp_path / "model_a"), 0.1, seed=1)
    b = siamese.SiameseNetwork((512,), str(tmp_path / "model_a"), 0.1, seed=2)
    assert not b.maybeLoadFromMemory()          # nothing saved yet -> False, like the reference
    a.save()
    assert b.maybeLoadFromMemory()
    L, R = _data(10, 512, 0)
    assert np.array_equal(a.predict([L, R]), b.predict([L, R]))
    # the file is Keras 2.1.2's HDF5 layout (code/siamese.py:121-125): readable by the real libhdf5,
    # and a file written by libhdf5 the way Keras/h5py would loads back (pretrained disguisedModel.h5)
    path = str(tmp_path / "model_a.h5")
    assert open(path, "rb").read(8) == b"\x89HDF\r\n\x1a\n"
    import h5ref
    if h5ref.lib() is not None:
        ws = a.siamese_net.get_weights()
        assert np.array_equal(h5ref.read_dataset(path, "/dense_1/dense_1/kernel:0"), ws[0])
        assert np.array_equal(h5ref.read_dataset(path, "/dense_3/dense_3/bias:0"), ws[5])
        rng = np.random.RandomState(3)
        layers = [("input_5", []), ("input_6", []), ("lambda_3", []),
                  ("dense_7", [("dense_7/kernel:0", rng.randn(512, 512).astype(np.float32)), ("dense_7/bias:0", rng.randn(512).astype(np.float32))]),
                  ("dense_8", [("dense_8/kernel:0", rng.randn(512, 64).astype(np.float32)), ("dense_8/bias:0", rng.randn(64).astype(np.float32))]),
                  ("dense_9", [("dense_9/kernel:0", rng.randn(64, 2).astype(np.float32)), ("dense_9/bias:0", rng.randn(2).astype(np.float32))]),
                  ("activation_3", [])]
        h5ref.write_keras_like(str(tmp_path / "pretrained.h5"), layers)
        c = siamese.SiameseNetwork((512,), str(tmp_path / "pretrained"), 0.1, seed=9)
        assert c.maybeLoadFromMemory()
        got = c.siamese_net.get_weights()
        want = [w for _, ws_ in layers for _, w in ws_]
        assert all(np.array_equal(g, w) for g, w in zip(got, want))
    d = siamese.SiameseNetwork((2048,), str(tmp_path / "model_a"), 0.1, seed=2)
    assert not d.maybeLoadFromMemory()          # shape mismatch -> any exception -> False (code/siamese.py:114-119)


def test_dp_train_step_single_rank_group(gpu):
    """distributed.dp_train_on_batch on a 1-rank process group (RCCL): gradient alias, global
    normaliser, all-reduce, update — must equal the plain train_on_batch / the oracle."""
    import os
    import torch.distributed as dist
    from a_link_amd import distributed as D
    from oracle import siamese_head as O
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        g, o = _pair(lr=0.1)
        L, R = _data(16, 512, 42)
        y = O.to_categorical(np.random.RandomState(3).randint(0, 2, 16))
        cw = {0: 0.25, 1: 0.75}
        mg = D.dp_train_on_batch(g, [L, R], y, class_weight=cw, mode="sharded")     # the all-reduce path
        mo = o.train_on_batch([L, R], y, class_weight=cw)
        np.testing.assert_allclose(mg, mo, rtol=2e-5, atol=1e-6)
        for a, b in zip(g.get_weights(), o.get_weights()):
            np.testing.assert_allclose(a, b, atol=3e-6)
        # "auto" at the reference's batch 16 is the replicated step: the plain train_on_batch, no collective
        assert len(y) < D.DP_SHARD_MIN_ROWS
        g2, o2 = _pair(lr=0.1)
        g3, _ = _pair(lr=0.1)
        m2 = D.dp_train_on_batch(g2, [L, R], y, class_weight=cw)
        m3 = g3.train_on_batch([L, R], y, class_weight=cw)
        assert m2 == m3
        for a, b in zip(g2.get_weights(), g3.get_weights()):
            assert np.array_equal(a, b)                       # replicas stay bit-identical
        gt = g.grads_tensor()
        assert gt.is_cuda and gt.numel() == 295618
    finally:
        dist.destroy_process_group()


def test_custom_train_model_matches_oracle(gpu):
    """SiameseNetwork.customTrainModel (reference code/siamese.py:81-112): generator batches, random
    80/20 split, Python-2 integer-division class weights, train_on_batch + test_on_batch."""
    from a_link_amd import siamese
    from oracle import siamese_head as O

    def gen(seed):
        rs = np.random.RandomState(seed)
        while True:
            n = 16
            L, R = rs.randn(n, 512).astype(np.float32), rs.randn(n, 512).astype(np.float32)
            y = (rs.rand(n, 1) > 0.3).astype(int)       # unbalanced: class weights differ
            yield [L, R], y

    net = siamese.SiameseNetwork((512,), "ctm", 0.1, seed=31)
    o = O.HeadModel(512, lr=0.1)
    o.set_weights(net.siamese_net.get_weights())
    np.random.seed(5)
    lg = net.customTrainModel(gen(1), 2, 16, 0.2, n_steps=16 * 6, verbose=0)
    np.random.seed(5)
    lo = O.custom_train_model(o, gen(1), 2, 16, 0.2, n_steps=16 * 6)
    np.testing.assert_allclose(np.array(lg), np.array(lo), rtol=1e-4, atol=1e-5)
    for a, b in zip(net.siamese_net.get_weights(), o.get_weights()):
        np.testing.assert_allclose(a, b, atol=2e-5)
    acc = net.testAccuracy(np.random.RandomState(0).randn(20, 512).astype(np.float32), np.arange(20) % 4)
    assert 0.0 <= acc <= 1.0


def test_head_error_paths(gpu):
    from a_link_amd.head import DenseHead
    with pytest.raises(gpu.AlinkError):
        DenseHead(510)                       # d_in not a multiple of 8
    with pytest.raises(gpu.AlinkError):
        DenseHead(512, h1=100)
    with pytest.raises(gpu.AlinkError):
        DenseHead(512, h2=48)
    h = DenseHead(64, 128, 32, seed=0)
    with pytest.raises(AssertionError):
        h.set_weights([np.zeros((64, 128))])
    L = np.zeros((5000, 64), np.float32)
    with pytest.raises(gpu.AlinkError):      # train batches are capped (scratch sized for 4096 rows)
        h.train_on_batch([L, L], np.zeros((5000, 2), np.float32))
    assert h.predict([L[:0], L[:0]]).shape == (0, 2)


@pytest.mark.parametrize("d,out_dim", [(512, 2), (2048, 2), (512, 1)])
def test_tiny_batch_step_equals_generic_chain(gpu, d, out_dim):
    """Batches of <= 32 rows take the three-launch train step (head.hip, tiny_*); it must leave the
    same metrics, gradients, parameters and Adadelta state as the generic chain — zero sample weights,
    ragged row groups (n % 4 != 0), gradient-only mode and the 32 / 33 boundary included."""
    from a_link_amd.head import DenseHead
    rng = np.random.RandomState(5)
    a = DenseHead(d, lr=0.1, seed=3, out_dim=out_dim)
    b = DenseHead(d, lr=0.1, seed=3, out_dim=out_dim)
    ws = a.get_weights()
    for i in (1, 3, 5):
        ws[i] = (rng.randn(*ws[i].shape) * 0.1).astype(np.float32)
    a.set_weights(ws)
    b.set_weights(ws)
    for step, n in enumerate([16, 5, 32, 1, 33, 16]):
        L, R = _data(n, d, 300 + step)
        lab = rng.randint(0, 2, n)
        y = np.eye(2, dtype=np.float32)[lab] if out_dim == 2 else lab.reshape(n, 1).astype(np.float32)
        sw = None
        if step % 2 == 1:
            sw = rng.rand(n).astype(np.float32)
            sw[::3] = 0.0
            if not sw.any():
                sw[0] = 1.0
        a.lib.alink_debug_set_tiny_step(1)
        ma = a.train_on_batch([L, R], y, sample_weight=sw)
        a.lib.alink_debug_set_tiny_step(0)
        mb = b.train_on_batch([L, R], y, sample_weight=sw)
        a.lib.alink_debug_set_tiny_step(1)
        np.testing.assert_allclose(ma, mb, rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(a.grads_tensor().cpu().numpy(), b.grads_tensor().cpu().numpy(), rtol=1e-5, atol=1e-8)
        for x, z in zip(a.get_weights(), b.get_weights()):
            np.testing.assert_allclose(x, z, rtol=0, atol=2e-6)
    # gradient-only mode (the data-parallel step) and the input gradient that follows it
    L, R = _data(12, d, 999)
    lab = rng.randint(0, 2, 12)
    y = np.eye(2, dtype=np.float32)[lab] if out_dim == 2 else lab.reshape(12, 1).astype(np.float32)
    ga = [t.cpu().numpy() for t in a.input_gradients(L, R, y)]
    a.lib.alink_debug_set_tiny_step(0)
    gb = [t.cpu().numpy() for t in b.input_gradients(L, R, y)]
    a.lib.alink_debug_set_tiny_step(1)
    for x, z in zip(ga, gb):
        np.testing.assert_allclose(x, z, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(a.grads_tensor().cpu().numpy(), b.grads_tensor().cpu().numpy(), rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("bs,n", [(16, 203), (64, 300), (16, 16)])
def test_device_resident_fit_equals_step_by_step_fit(gpu, bs, n):
    """DenseHead.fit keeps the training set on the device (index gathers, one metrics read-back per epoch); KerasFitMixin.fit
    is the step-by-step form (host gather, upload, train_on_batch, synchronise — Keras' own control flow, reference
    code/siamese.py:57).  Same kernels on the same batches in the same order: weights and the per-epoch logs are equal bit for
    bit, with a validation split, a ragged last batch, batches on both sides of the tiny-step switch (32 rows), and the
    finetune() callbacks attached."""
    from a_link_amd.head import DenseHead, KerasFitMixin, EarlyStopping, ReduceLROnPlateau
    rng = np.random.RandomState(0)
    L, R = rng.randn(n, 512).astype(np.float32), rng.randn(n, 512).astype(np.float32)
    y = np.zeros((n, 2), np.float32)
    y[np.arange(n), rng.randint(0, 2, n)] = 1
    out = []
    for fast in (True, False):
        hd = DenseHead(512, lr=0.1, seed=3)
        cbs = [EarlyStopping(monitor='val_loss', min_delta=0.1, patience=5), ReduceLROnPlateau(monitor='val_loss', factor=0.2, patience=5, min_lr=0.01)]
        np.random.seed(11)
        fit = hd.fit if fast else (lambda *a, **k: KerasFitMixin.fit(hd, *a, **k))
        hist = fit([L, R], y, batch_size=bs, epochs=3, validation_split=0.2, verbose=0, callbacks=cbs)
        out.append((hd.get_weights(), hist))
    (wa, ha), (wb, hb) = out
    for a, b in zip(wa, wb):
        assert np.array_equal(a, b)
    assert sorted(ha) == sorted(hb)
    for k in ha:
        assert ha[k] == hb[k], (k, ha[k], hb[k])


def test_custom_train_model_with_deferred_metrics_equals_the_step_by_step_form(gpu):
    """customTrainModel(verbose=0) on a DenseHead enqueues its steps and reads their {loss, accuracy} back in blocks; with
    `_defer_metrics = False` every step is synchronised and read on its own (the reference's shape, code/siamese.py:99-108).  Same
    kernels, same batches, same order of the float64 sums: logs and weights equal bit for bit — across a block boundary (300
    steps, blocks of 256), with held-out rows and without (valRatio 0)."""
    from a_link_amd import siamese

    def gen(seed):
        rs = np.random.RandomState(seed)
        while True:
            n = int(rs.randint(10, 17))
            L, R = rs.randn(n, 512).astype(np.float32), rs.randn(n, 512).astype(np.float32)
            yield [L, R], (rs.rand(n, 1) > 0.3).astype(int)

    for val_ratio in (0.2, 0.0):
        out = []
        for deferred in (True, False):
            net = siamese.SiameseNetwork((512,), "ctm", 0.1, seed=31)
            net._defer_metrics = deferred
            np.random.seed(5)
            logs = net.customTrainModel(gen(1), 2, 16, val_ratio, n_steps=16 * 300, verbose=0)
            out.append((logs, net.siamese_net.get_weights()))
        assert out[0][0] == out[1][0], (val_ratio, out[0][0], out[1][0])
        for a, b in zip(out[0][1], out[1][1]):
            assert np.array_equal(a, b)


def test_custom_train_model_over_index_batches_equals_the_step_by_step_form(gpu):
    """customTrainModel over this package's own balanced generator (pairs.getGenerator, code/readDFW.py:180-209) keeps the
    generator's feature table on the device and ships a step as row indices, `block` steps per call
    (alink_head_custom_train_steps); `_index_steps = False` gathers the rows on the host and uploads them step by step, and
    `_defer_metrics = False` also synchronises every step (the reference's shape, code/siamese.py:91-110).  The same batches,
    the same random stream, the same kernels: logs and weights equal bit for bit — across block boundaries (600 steps, blocks of
    256), batches above and below the 32-row limit of the three-launch step, with and without held-out rows, for the two-column
    scorer with class weights and the baseline scripts' one-column scorer without (code/siamese3.py:64-86); and a finite
    generator that ends mid-epoch raises StopIteration after the same steps."""
    from a_link_amd import pairs, siamese, siamese3
    rng = np.random.RandomState(0)
    feats = [rng.randn(rng.randint(2, 6), 512).astype(np.float32) for _ in range(40)]

    def make(infinite=True, bs=16):
        return pairs.getGenerator(pairs.getNormalGenerator(feats, bs, infinite=infinite), pairs.getNormalGenerator(feats, bs, infinite=infinite),
                                  pairs.getImposterGenerator(feats, feats, bs, infinite=infinite), 16)

    for cls, val_ratio, bs in ((siamese.SiameseNetwork, 0.2, 16), (siamese.SiameseNetwork, 0.0, 16), (siamese3.SiameseNetwork, 0.2, 16),
                               (siamese.SiameseNetwork, 0.2, 64)):
        out = []
        for indexed, deferred in ((True, True), (False, True), (False, False)):
            net = cls((512,), "ctm", 0.1, seed=31)
            net._index_steps, net._defer_metrics = indexed, deferred
            gen = make(bs=bs)
            assert gen.indexable
            np.random.seed(5)
            logs = net.customTrainModel(gen, 2, 16, val_ratio, n_steps=16 * 300, verbose=0)
            out.append((logs, net.siamese_net.get_weights(), np.random.get_state()[1][:8].tolist()))
        for other in out[1:]:
            assert out[0][0] == other[0], (cls, val_ratio, out[0][0], other[0])
            assert out[0][2] == other[2]                              # the random stream was consumed identically
            for a, b in zip(out[0][1], other[1]):
                assert np.array_equal(a, b)
    # a finite generator: StopIteration after the steps it had batches for
    ends = []
    for indexed in (True, False):
        net = siamese.SiameseNetwork((512,), "ctm", 0.1, seed=31)
        net._index_steps = indexed
        np.random.seed(6)
        with pytest.raises(StopIteration):
            net.customTrainModel(make(infinite=False), 1, 16, 0.2, n_steps=16 * 100000, verbose=0)
        ends.append(net.siamese_net.get_weights())
    for a, b in zip(*ends):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("d", [2048, 256])
def test_three_launch_step_of_smallres_head_matches_generic_chain_and_oracle(gpu, d):
    """alink_head_train_step_input_grads: gradients-only train step + input gradients in one call.  For SmallRes' head shape
    (128 / 32 / 2 on a multiple of 256 features, at most 32 pairs) it is three launches — Dense1 partial sums per 32 inputs, ONE
    workgroup for everything whose operands fit in LDS, dW1 + input gradients per 32 inputs — instead of the generic chain's
    seven (alink_debug_set_mini_step(0) brings those back).  Same sums in a different order for Dense1 / Dense2: parameters'
    gradients, input gradients (with and without the ReLU mask of SmallRes' feature layer) and metrics agree to rounding
    with the generic chain, and with the NumPy oracle's gradients; 33 pairs are beyond the path and take the generic one."""
    from oracle import siamese_head as O
    g, o = _pair(d, 128, 32, seed=5)
    lib = g.lib
    rng = np.random.RandomState(8)
    for n in (1, 5, 16, 17, 32, 33):
        L = np.abs(rng.randn(n, d)).astype(np.float32) * (rng.rand(n, d) > 0.4)      # post-ReLU features: zeros included
        R = np.abs(rng.randn(n, d)).astype(np.float32) * (rng.rand(n, d) > 0.4)
        L, R = L.astype(np.float32), R.astype(np.float32)
        R[:, :7] = L[:, :7]                                                          # l == r: the abs gradient is 0 there
        y = np.eye(2, dtype=np.float32)[rng.randint(0, 2, n)]
        sw = None
        if n in (5, 17, 32):
            sw = rng.rand(n).astype(np.float32)
            sw[::4] = 0.0
        out = {}
        for mini in (1, 0):
            lib.alink_debug_set_mini_step(mini)
            try:
                dL, dR = g.input_gradients(L, R, y, sample_weight=sw)
                met = g._metrics.cpu().numpy()[:2].copy()
                Ld, Rd, yd = g._dev(L), g._dev(R), g._dev(y)
                swd = None if sw is None else g._dev(sw)
                mL, mR = torch.empty_like(Ld), torch.empty_like(Rd)
                cs = torch.empty(d, dtype=torch.float32, device=g.device)
                gpu.check(lib.alink_head_train_step_input_grads(g.h, gpu.ptr(Ld), gpu.ptr(Rd), gpu.ptr(yd), gpu.ptr(swd), n, 0.0, 1,
                                                                gpu.ptr(mL), gpu.ptr(mR), gpu.ptr(cs), gpu.ptr(g._metrics),
                                                                gpu.current_stream(g.device)))
                out[mini] = (dL.cpu().numpy(), dR.cpu().numpy(), g.grads_tensor().cpu().numpy().copy(), met,
                             mL.cpu().numpy(), mR.cpu().numpy(), cs.cpu().numpy())
            finally:
                lib.alink_debug_set_mini_step(1)
        a, b = out[1], out[0]
        scale = max(np.abs(b[2]).max(), 1e-12)
        for x, z in zip(a[:3] + a[4:], b[:3] + b[4:]):
            np.testing.assert_allclose(x, z, rtol=2e-4, atol=2e-6 * max(np.abs(z).max(), 1e-12))
        np.testing.assert_allclose(a[3], b[3], rtol=2e-6, atol=1e-7)
        if n == 33:
            for x, z in zip(a, b):
                assert np.array_equal(x, z)
        # the masked form: the plain input gradients where the input is positive, zero elsewhere
        assert np.array_equal(a[4], np.where(L > 0, a[0], 0.0).astype(np.float32))
        assert np.array_equal(a[5], np.where(R > 0, a[1], 0.0).astype(np.float32))
        assert np.all(a[0][:, :7] == 0) and np.all(a[1][:, :7] == 0)
        # the column sums of [dL ; dR] (the bias gradient of the layer before the head), rows ascending: the same bits as a loop
        for got in (a, b):
            t = np.zeros(d, np.float32)
            for row in np.concatenate([got[4], got[5]]):
                t = t + row
            assert np.array_equal(got[6], t)
        # the oracle's parameter gradients of the same batch
        ref, loss, acc = O.gradients(o.get_weights(), L, R, y, sw, dtype=np.float64)
        flat = np.concatenate([np.asarray(t).ravel() for t in ref])
        np.testing.assert_allclose(a[2], flat, rtol=1e-3, atol=1e-5 * scale)
        np.testing.assert_allclose(a[3], [loss, acc], rtol=1e-5, atol=1e-6)

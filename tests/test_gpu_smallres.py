"""GPU parity of SmallRes (smallres.hip) against the torch-CPU autograd oracle (oracle/smallres.py):
predict, one and several train_on_batch steps with explicit dropout masks, test_on_batch,
preprocessing, and the reference-shaped API (siamese.SmallRes)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _imgs(n, s, seed):
    rng = np.random.RandomState(seed)
    return rng.randint(0, 256, (n, s, s, 3)).astype(np.float32)


@pytest.mark.parametrize("size,feat", [(32, 2048), (48, 256), (16, 64)])      # 16: rows too short for the weight gradient's pixel walk (its 4-byte form)
def test_forward_and_training_match_oracle(gpu, size, feat):
    from a_link_amd.smallres import SmallResNet
    from oracle import siamese_head as O
    from oracle import smallres as OS
    net = SmallResNet((size, size, 3), feat, lr=0.1, seed=3)
    ws = net.get_weights()
    rng = np.random.RandomState(7)
    for i in range(1, len(ws), 2):                       # non-zero biases
        ws[i] = (rng.randn(*ws[i].shape) * 0.05).astype(np.float32)
    net.set_weights(ws)
    om = OS.SmallResModel(ws, lr=0.1)
    n = 6
    L = (_imgs(n, size, 1) - 128.) / 128.
    R = (_imgs(n, size, 2) - 128.) / 128.
    np.testing.assert_allclose(net.predict([L, R]), om.predict([L, R]), atol=2e-5)
    p1, p2 = (size - 2) // 2, ((size - 2) // 2 - 2) // 2
    shapes = ((p1, p1, 32), (p2, p2, 64))
    assert net.mask_sizes == (p1 * p1 * 32, p2 * p2 * 64)
    y = O.to_categorical(rng.randint(0, 2, n))
    for step in range(3):
        masks = (rng.rand(2 * n * (net.mask_sizes[0] + net.mask_sizes[1])) >= 0.25).astype(np.uint8) if step < 2 else None
        sw = None if step != 1 else np.array([1, 0.5, 0, 2, 1, 1], np.float32)
        net.training_dropout = masks is not None
        mg = net.train_on_batch([L, R], y, sample_weight=sw, masks=masks)
        mo, _ = om.train_on_batch([L, R], y, sample_weight=sw, masks=masks, mask_shapes=shapes)
        np.testing.assert_allclose(mg, mo, rtol=1e-4, atol=1e-5)
        for a, b in zip(net.get_weights(), om.ws):
            np.testing.assert_allclose(a, b, atol=3e-5)
    np.testing.assert_allclose(net.test_on_batch([L, R], y), om.test_on_batch([L, R], y), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(net.predict([L, R]), om.predict([L, R]), atol=5e-5)


def test_reference_shaped_api(gpu, tmp_path):
    from a_link_amd import siamese
    from oracle import smallres as OS
    m = siamese.SmallRes((32, 32, 3), (2048,), str(tmp_path / "lowres"), 0.1, seed=1)
    om = OS.SmallResModel(m.siamese_net.get_weights())
    L, R = _imgs(5, 32, 3), _imgs(5, 32, 4)               # raw 0..255 pixels: predict applies preprocess
    got = m.predict([L, R])
    ref = om.predict([(L - 128.) / 128., (R - 128.) / 128.])
    np.testing.assert_allclose(got, ref, atol=2e-5)
    assert m.getDenseBarebones() == [(128, 'relu'), (32, 'relu'), (2, None)]
    # finetune runs (dropout on, random masks) and changes the weights; save/load round-trips
    np.random.seed(0)
    Y = np.array([[1], [0], [1], [0], [1]])
    hist = m.finetune([L, R], Y, 1, 2, verbose=0)
    assert "loss" in hist and np.isfinite(hist["loss"][0])
    m.save()
    m2 = siamese.SmallRes((32, 32, 3), (2048,), str(tmp_path / "lowres"), 0.1, seed=2)
    assert m2.maybeLoadFromMemory()
    assert np.array_equal(m2.predict([L, R]), m.predict([L, R]))


def test_device_drawn_dropout_masks(gpu):
    """alink_keep_masks (the Dropout(0.25) keep-masks of the SmallRes tower, code/siamese.py:146,153, drawn on the device since
    round 5): keep probability 0.75 to 3 sigma over 2 M elements, a different seed gives different masks, the same seed the same;
    and a train step that draws its own masks consumes exactly ONE np.random draw (what keeps the ranks of a multi-rank loop in
    step) and changes the weights."""
    import torch
    from a_link_amd.smallres import SmallResNet
    lib = gpu.load()
    n = 1 << 21
    a, b, c = (torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(3))
    gpu.check(lib.alink_keep_masks(gpu.ptr(a), n, 0.75, 11, None))
    gpu.check(lib.alink_keep_masks(gpu.ptr(b), n, 0.75, 11, None))
    gpu.check(lib.alink_keep_masks(gpu.ptr(c), n, 0.75, 12, None))
    torch.cuda.synchronize()
    assert torch.equal(a, b) and not torch.equal(a, c) and int(a.max()) == 1
    p = float(a.float().mean())
    assert abs(p - 0.75) < 3 * np.sqrt(0.75 * 0.25 / n)
    assert abs(float((a.float() * c.float()).mean()) - 0.75 * 0.75) < 4e-3          # independent streams
    net = SmallResNet((32, 32, 3), 64, lr=0.1, seed=3)
    rng = np.random.RandomState(0)
    L = ((rng.randint(0, 256, (4, 32, 32, 3)) - 128.) / 128.).astype(np.float32)
    R = ((rng.randint(0, 256, (4, 32, 32, 3)) - 128.) / 128.).astype(np.float32)
    y = np.eye(2, dtype=np.float32)[rng.randint(0, 2, 4)]
    w0 = net.get_weights()
    np.random.seed(3)
    net.train_on_batch([L, R], y)
    after = np.random.rand()
    np.random.seed(3)
    np.random.randint(0, 2 ** 31 - 1)
    assert after == np.random.rand()
    assert any(not np.array_equal(u, v) for u, v in zip(w0, net.get_weights()))


def test_train_step_replayed_as_a_graph_equals_plain_launches(gpu):
    """SmallResNet(use_graph=True) + alink_smallres_set_graph: the ~40 launches of a step on the model's own stream, captured the
    second time an operand set is seen and replayed from then on (optional, off by default: a replayed node costs more than a
    launch on this ROCm).  Same kernels in the same order: metrics and every weight after five steps — plain, plain, captured,
    replayed, replayed — equal the plain-launch model's bit for bit, and predict sees the updated weights."""
    from a_link_amd.smallres import SmallResNet
    rs = np.random.RandomState(2)
    L = ((rs.randint(0, 256, (8, 32, 32, 3)) - 128.0) / 128.0).astype(np.float32)
    R = ((rs.randint(0, 256, (8, 32, 32, 3)) - 128.0) / 128.0).astype(np.float32)
    y = np.eye(2, dtype=np.float32)[rs.randint(0, 2, 8)]
    out = []
    for graph in (False, True):
        net = SmallResNet((32, 32, 3), 256, lr=0.1, seed=4)
        net.use_graph = graph
        gpu.check(net.lib.alink_smallres_set_graph(net.h, 1 if graph else 0))
        np.random.seed(9)
        ms = [net.train_on_batch([L, R], y) for _ in range(5)]
        out.append((ms, net.get_weights(), net.predict([L, R])))
    assert out[0][0] == out[1][0]
    for a, b in zip(out[0][1], out[1][1]):
        assert np.array_equal(a, b)
    assert np.array_equal(out[0][2], out[1][2])


def test_masks_drawn_by_the_step_equal_a_mask_launch_then_the_step(gpu):
    """alink_smallres_train_step_drawn: the Dropout keep-masks come from extra workgroups of the step's first launch.  They are
    alink_keep_masks' bytes for the same seed, so three steps that draw their own (train_on_batch's default) leave the same
    metrics and weights, bit for bit, as three steps fed masks a separate alink_keep_masks launch drew from the same seeds —
    with and without the captured-graph form (which draws in a launch of its own: a captured launch's seed would be frozen)."""
    import torch
    from a_link_amd.smallres import SmallResNet
    rs = np.random.RandomState(7)
    L = rs.randint(0, 256, (12, 32, 32, 3)).astype(np.float32)
    R = rs.randint(0, 256, (12, 32, 32, 3)).astype(np.float32)
    y = np.eye(2, dtype=np.float32)[rs.randint(0, 2, 12)]
    lib = gpu.load()
    out = []
    for mode in ("drawn", "launch", "drawn_graph"):
        net = SmallResNet((32, 32, 3), 256, lr=0.1, seed=8)
        if mode == "drawn_graph":
            net.use_graph = True
            gpu.check(lib.alink_smallres_set_graph(net.h, 1))
        np.random.seed(21)
        ms = []
        for _ in range(3):
            if mode == "launch":
                e1, e2 = net.mask_sizes
                md = torch.empty(2 * 12 * (e1 + e2), dtype=torch.uint8, device="cuda")
                gpu.check(lib.alink_keep_masks(gpu.ptr(md), md.numel(), 0.75, int(np.random.randint(0, 2 ** 31 - 1)), None))
                torch.cuda.synchronize()
                ms.append(net.train_on_batch([L, R], y, masks=md.cpu().numpy()))
            else:
                ms.append(net.train_on_batch([L, R], y))
        out.append((ms, net.get_weights()))
    for other in out[1:]:
        assert out[0][0] == other[0]
        for a, b in zip(out[0][1], other[1]):
            assert np.array_equal(a, b)

"""GPU: the pair head's bf16 compute mode (alink_head_set_compute_dtype: BASELINE configs[4]'s "bf16 fine-tune" —
f32 master weights, gradients and Adadelta state; bf16 GEMM operands) against
  (a) the oracle restating exactly that arithmetic (oracle.siamese_head quant="bf16": same rounding points, products
      and sums in float32) — tight, and
  (b) the float32 oracle, i.e. what the reference's Keras model computes (code/siamese.py:27-35,52-58) — with the
      STATED tolerance of the mode: after 10 fine-tune steps at batch 16, |d loss| <= 1e-2, max |d weight| <= 2e-3,
      probabilities within 3e-2 (measured 3.8e-3, 4.9e-4 and 1.0e-2 on random 512-d inputs).
The two sides of (a) sum in different orders (fmaf chains / MFMA on the device, BLAS in NumPy), so a GEMM operand whose
float32 value lies on a bf16 rounding boundary can round the other way: one bf16 ulp (0.4 %) on one of ~10^5 operands,
seen as ~1e-3 on a probability or a loss now and then.  The tolerances of (a) are set for that, not for 1e-6."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _data(n, d, seed):
    rs = np.random.RandomState(seed)
    return rs.randn(n, d).astype(np.float32), rs.randn(n, d).astype(np.float32)


def _pair(d_in=512, lr=0.1, seed=3, out_dim=2):
    from a_link_amd.head import DenseHead
    from oracle import siamese_head as O
    g = DenseHead(d_in, lr=lr, seed=seed, out_dim=out_dim, compute_dtype="bf16")
    oq = O.HeadModel(d_in, lr=lr, out_dim=out_dim, quant="bf16")
    of = O.HeadModel(d_in, lr=lr, out_dim=out_dim)
    ws = g.get_weights()
    ws = [w + (0.05 * np.random.RandomState(9).randn(*w.shape).astype(np.float32) if w.ndim == 1 else 0) for w in ws]
    g.set_weights(ws)
    oq.set_weights(ws)
    of.set_weights(ws)
    return g, oq, of


@pytest.mark.parametrize("d_in,n", [(512, 16), (512, 32), (512, 64), (2048, 16), (512, 700)])
def test_bf16_train_steps_match_the_bf16_oracle(gpu, d_in, n):
    """batch <= 32: the three-launch step reading the 2-byte weights; larger: the generic chain on their widened
    image; both must be the arithmetic the oracle restates.  Checked STEP BY STEP from the device's own weights (ten
    steps of two free-running copies drift apart through the rounding-boundary flips described above): loss and
    gradients of each step against the oracle at the same weights, and the new master weights against the oracle's
    Adadelta applied to the device's own gradients (f32 masters, f32 state: exact to rounding)."""
    from oracle import siamese_head as O
    g, oq, of = _pair(d_in)
    L, R = _data(n, d_in, 1)
    rs = np.random.RandomState(2)
    opt = O.Adadelta([w.shape for w in g.get_weights()], lr=0.1)
    for step in range(10):
        y = O.to_categorical(rs.randint(0, 2, n))
        cw = {0: 0.3, 1: 0.7} if step % 2 else None
        sw = None if cw is None else np.asarray([cw[c] for c in y.argmax(1)], np.float32)
        ws0 = g.get_weights()
        mg = g.train_on_batch([L, R], y, class_weight=cw)
        go, loss_o, acc_o = O.gradients(ws0, L, R, y, sw, quant="bf16")
        assert abs(mg[0] - loss_o) < 2e-3 and abs(mg[1] - acc_o) <= 1.0 / n, (step, mg, loss_o, acc_o)
        flat = g.grads_tensor().cpu().numpy()
        gd, o = [], 0
        for w in ws0:
            gd.append(flat[o:o + w.size].reshape(w.shape))
            o += w.size
        for a, b in zip(gd, go):
            scale = np.abs(b).max() + 1e-12
            err = np.abs(a - b) / scale
            assert err.max() < 5e-2 and np.mean(err > 1e-3) < 0.05, (step, err.max(), np.mean(err > 1e-3))
        want = opt.step(ws0, gd)
        for a, b in zip(g.get_weights(), want):
            assert np.abs(a - b).max() < 2e-6, step
    pg, pq_ = g.predict([L, R]), O.forward(g.get_weights(), L, R, quant="bf16")
    assert np.abs(pg - pq_).max() < 5e-3 and np.mean(np.abs(pg - pq_) > 2e-4) < 0.1
    eg = g.test_on_batch([L, R], y)
    lo, ao = O.loss_and_metrics(y, pq_)
    assert abs(eg[0] - lo) < 3e-3 and abs(eg[1] - ao) <= 1.0 / n


def test_bf16_finetune_stated_tolerance_vs_f32_reference_arithmetic(gpu, capsys):
    from oracle import siamese_head as O
    g, oq, of = _pair(512)
    L, R = _data(16, 512, 4)
    rs = np.random.RandomState(5)
    dl = 0.0
    for step in range(10):
        y = O.to_categorical(rs.randint(0, 2, 16))
        mg = g.train_on_batch([L, R], y)
        mf = of.train_on_batch([L, R], y)
        dl = max(dl, abs(mg[0] - mf[0]))
    dw = max(np.abs(a - b).max() for a, b in zip(g.get_weights(), of.get_weights()))
    dp = np.abs(g.predict([L, R]) - of.predict([L, R])).max()
    with capsys.disabled():
        print("\n[bf16 fine-tune vs float32 oracle, 10 steps, batch 16] max |d loss| %.2e  max |d weight| %.2e  max |d p| %.2e" % (dl, dw, dp))
    assert dl <= 1e-2 and dw <= 2e-3 and dp <= 3e-2


def test_bf16_predict_index_gather_and_committee(gpu):
    from a_link_amd import committee, siamese
    from oracle import al_logic as OA
    from oracle import siamese_head as O
    rs = np.random.RandomState(0)
    E = rs.randn(300, 512).astype(np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    li = rs.randint(0, 300, 5000).astype(np.int32)
    ri = rs.randint(0, 300, 5000).astype(np.int32)
    nets = [siamese.SiameseNetwork((512,), "b%d" % i, 0.1, seed=i, compute_dtype="bf16") for i in range(3)]
    refs = []
    for nnet in nets:
        o = O.HeadModel(512, quant="bf16")
        o.set_weights(nnet.siamese_net.get_weights())
        refs.append(o.predict([E[li], E[ri]]))
    got = committee.Bagging(nets, []).predict_indexed(E, E, li, ri).cpu().numpy()
    ref = OA.bagging_predict(refs)
    assert np.abs(got - ref).max() < 3e-3 and np.mean(np.abs(got - ref) > 2e-4) < 0.03


def test_masters_stay_f32_and_mode_switch(gpu):
    from a_link_amd.head import DenseHead
    from oracle import siamese_head as O
    g = DenseHead(512, lr=0.1, seed=7, compute_dtype="bf16")
    ws = g.get_weights()
    assert any(not np.array_equal(w, O.bf16_round(w)) for w in ws)         # what comes back is the f32 master
    L, R = _data(16, 512, 8)
    y = O.to_categorical(np.random.RandomState(1).randint(0, 2, 16))
    f = DenseHead(512, lr=0.1, seed=7)
    f.set_weights(ws)
    g.set_compute_dtype("f32")
    assert np.array_equal(g.predict([L, R]), f.predict([L, R]))            # back in f32: bit-equal to an f32 head
    assert g.train_on_batch([L, R], y) == f.train_on_batch([L, R], y)
    g.set_compute_dtype("bf16")
    o = O.HeadModel(512, lr=0.1, quant="bf16")
    o.set_weights(g.get_weights())
    o.opt.a = [a.copy() for a in _adadelta_state(f)[0]]
    o.opt.d = [d.copy() for d in _adadelta_state(f)[1]]
    mg, mo = g.train_on_batch([L, R], y), o.train_on_batch([L, R], y)
    assert abs(mg[0] - mo[0]) < 1e-3
    gt = g.grads_tensor()
    assert gt.dtype == torch.float32 and gt.numel() == 295618               # the all-reduce buffer stays f32


def _adadelta_state(head):
    """Adadelta accumulators after ONE step from zero state, recomputed from the gradient buffer:
    a = (1 - rho) g^2, d = (1 - rho) u^2 with u = g sqrt(eps) / sqrt(a + eps)."""
    g = head.grads_tensor().cpu().numpy().astype(np.float32)
    rho, eps = np.float32(0.95), np.float32(1e-8)
    a = (np.float32(1) - rho) * g * g
    u = g * np.sqrt(eps) / np.sqrt(a + eps)
    d = (np.float32(1) - rho) * u * u
    out_a, out_d, o = [], [], 0
    for s in head._shapes():
        n = int(np.prod(s))
        out_a.append(a[o:o + n].reshape(s))
        out_d.append(d[o:o + n].reshape(s))
        o += n
    return out_a, out_d


@pytest.mark.parametrize("d_in,P", [(512, 1), (512, 63), (512, 5000), (2048, 777)])
def test_bf16_matrix_core_predict_equals_the_rounding_kernel(gpu, d_in, P):
    """bf16 mode scores pairs with head_fwd_bf16_kernel (v_mfma_f32_16x16x32_bf16) when the head has the reference's
    shape; the f32-input kernel with in-flight rounding computes the same thing (bf16 operands, exact products, f32 sums)
    and stays the fallback for other shapes: the two agree up to summation order / rounding-boundary flips, with and
    without index gather, alone and inside a committee mean, for pair counts that do not fill a 64-pair workgroup."""
    from a_link_amd import committee, siamese
    from oracle import siamese_head as O
    rs = np.random.RandomState(P)
    n = 300
    E = rs.randn(n, d_in).astype(np.float32)
    E /= np.linalg.norm(E, axis=1, keepdims=True)
    li = rs.randint(0, n, P).astype(np.int32)
    ri = rs.randint(0, n, P).astype(np.int32)
    nets = [siamese.SiameseNetwork((d_in,), "q%d" % i, 0.1, seed=i, compute_dtype="bf16") for i in range(2)]
    lib = nets[0].siamese_net.lib
    outs = {}
    try:
        for fast in (1, 0):
            lib.alink_debug_set_head_bf16_mfma(fast)
            single = nets[0].siamese_net.predict_device(E, E, li, ri).cpu().numpy()
            plain = nets[0].predict([E[li], E[ri]])
            bag = committee.Bagging(nets, []).predict_indexed(E, E, li, ri).cpu().numpy()
            outs[fast] = (single, plain, bag)
    finally:
        lib.alink_debug_set_head_bf16_mfma(1)
    for a, b in zip(outs[1], outs[0]):
        assert a.shape == (P, 2) and np.isfinite(a).all()
        assert np.abs(a - b).max() < 3e-3 and np.mean(np.abs(a - b) > 2e-4) < 0.06
        np.testing.assert_allclose(a.sum(1), 1.0, atol=1e-5)
    assert np.array_equal(outs[1][0], outs[1][1])                           # gather by index == materialised pairs
    o = O.HeadModel(d_in, quant="bf16")
    o.set_weights(nets[0].siamese_net.get_weights())
    assert np.abs(outs[1][0] - o.predict([E[li], E[ri]])).max() < 3e-3

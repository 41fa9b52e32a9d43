"""CPU: the hand-derived parts of oracle/siamese_head.py against independent machinery.

oracle.gradients and oracle.Adadelta were derived by hand by the same author as csrc/head.hip; these tests
re-derive them with torch autograd (float64) through a forward written from the Keras graph
(reference code/siamese.py:27-35: abs(L-R) -> Dense relu -> Dense relu -> Dense -> softmax, compiled with
binary_crossentropy + Adadelta) and with torch.optim.Adadelta(lr, rho=0.95, eps=1e-8).  This does not pin the
oracle to the reference (Keras 2.1.2 is not installable: parity stays "unpinned") — it removes
the common-author failure mode for the derivatives and the update rule."""
import numpy as np
import pytest
import torch

from oracle import siamese_head as O


def _keras_graph_loss(ws, L, R, y, sw):
    """Forward + Keras loss in torch float64, written from the layer list, not from oracle.forward."""
    W1, b1, W2, b2, W3, b3 = ws
    x = (L - R).abs()
    x = torch.relu(x @ W1 + b1)
    x = torch.relu(x @ W2 + b2)
    z = x @ W3 + b3
    p = torch.softmax(z, dim=1) if z.shape[1] > 1 else torch.sigmoid(z)
    # keras.backend.binary_crossentropy(from_logits=False): clip, back to logits, sigmoid CE with logits
    pc = p.clamp(1e-7, 1 - 1e-7)
    logit = torch.log(pc / (1 - pc))
    per_elem = torch.nn.functional.binary_cross_entropy_with_logits(logit, y, reduction="none")
    per_sample = per_elem.mean(dim=1)
    # keras/engine/training.py _weighted_masked_objective
    return (per_sample * sw).mean() / (sw != 0).to(per_sample.dtype).mean(), p


@pytest.mark.parametrize("d_in,out_dim,weighted", [(512, 2, False), (512, 2, True), (2048, 2, True), (64, 1, False)])
def test_gradients_match_autograd_f64(d_in, out_dim, weighted):
    rng = np.random.RandomState(d_in + out_dim)
    ws = [w.astype(np.float64) for w in O.init_weights(d_in, 48, 16, seed=3, out_dim=out_dim)]
    for i in (1, 3, 5):
        ws[i] = rng.randn(*ws[i].shape) * 0.1                       # non-zero biases
    n = 24
    L, R = rng.randn(n, d_in), rng.randn(n, d_in)
    lab = rng.randint(0, 2, n)
    y = O.to_categorical(lab).astype(np.float64) if out_dim == 2 else lab.reshape(-1, 1).astype(np.float64)
    sw = rng.choice([0.0, 0.3, 1.7], n) if weighted else None
    g_o, loss_o, acc_o = O.gradients(ws, L, R, y, sw, dtype=np.float64)
    tw = [torch.tensor(w, dtype=torch.float64, requires_grad=True) for w in ws]
    tsw = torch.ones(n, dtype=torch.float64) if sw is None else torch.tensor(sw)
    loss_t, p_t = _keras_graph_loss(tw, torch.tensor(L), torch.tensor(R), torch.tensor(y), tsw)
    loss_t.backward()
    assert abs(loss_t.item() - float(loss_o)) < 1e-12
    for a, b in zip(g_o, tw):
        scale = max(1e-30, float(b.grad.abs().max()))
        assert np.abs(a - b.grad.numpy()).max() / scale < 1e-10
    acc_t = float((p_t.detach().round() == torch.tensor(y)).double().mean())
    assert abs(acc_t - float(acc_o)) < 1e-12
    np.testing.assert_allclose(O.forward(ws, L, R, np.float64), p_t.detach().numpy(), atol=1e-14)


def test_saturated_probabilities_have_zero_gradient_like_the_clip():
    """p outside [1e-7, 1-1e-7] is clipped BEFORE the log, so d(loss)/dp = 0 there — in the oracle and in
    autograd through clamp."""
    rng = np.random.RandomState(0)
    ws = [w.astype(np.float64) for w in O.init_weights(32, 16, 8, seed=1)]
    ws[4] = ws[4] * 400.0                                           # saturate the softmax
    L, R = rng.randn(8, 32), rng.randn(8, 32)
    y = O.to_categorical(rng.randint(0, 2, 8)).astype(np.float64)
    g_o, _, _ = O.gradients(ws, L, R, y, None, dtype=np.float64)
    tw = [torch.tensor(w, dtype=torch.float64, requires_grad=True) for w in ws]
    loss_t, p_t = _keras_graph_loss(tw, torch.tensor(L), torch.tensor(R), torch.tensor(y), torch.ones(8, dtype=torch.float64))
    loss_t.backward()
    assert float(((p_t < 1e-7) | (p_t > 1 - 1e-7)).double().mean()) > 0.5
    for a, b in zip(g_o, tw):
        np.testing.assert_allclose(a, b.grad.numpy(), atol=1e-12 * max(1.0, float(b.grad.abs().max())))


@pytest.mark.parametrize("lr", [1.0, 0.1])
def test_adadelta_matches_torch_optim(lr):
    """Keras 2.1.2 Adadelta(lr, rho=.95, epsilon=1e-8) and torch.optim.Adadelta are the same recurrence
    (Zeiler 2012 with a learning-rate factor): 5 steps on the head's six tensors, f64 and f32."""
    rng = np.random.RandomState(7)
    shapes = [(40, 24), (24,), (24, 8), (8,), (8, 2), (2,)]
    for dtype, tdt, tol in ((np.float64, torch.float64, 1e-13), (np.float32, torch.float32, 2e-6)):
        ws = [rng.randn(*s).astype(dtype) for s in shapes]
        tw = [torch.tensor(w.copy(), dtype=tdt, requires_grad=True) for w in ws]
        opt_t = torch.optim.Adadelta(tw, lr=lr, rho=0.95, eps=1e-8, weight_decay=0)
        opt_o = O.Adadelta(shapes, lr=lr, rho=0.95, epsilon=1e-8, dtype=dtype)
        for step in range(5):
            gs = [(rng.randn(*s) * 10.0 ** rng.randint(-4, 1)).astype(dtype) for s in shapes]
            ws = opt_o.step(ws, gs)
            for t, g in zip(tw, gs):
                t.grad = torch.tensor(g, dtype=tdt)
            opt_t.step()
            for a, b in zip(ws, tw):
                assert np.abs(a - b.detach().numpy()).max() <= tol * max(1.0, np.abs(a).max()), (step, dtype)


def test_train_on_batch_equals_autograd_plus_torch_adadelta():
    """Five full oracle train_on_batch steps == autograd gradients fed to torch.optim.Adadelta (f64)."""
    rng = np.random.RandomState(11)
    m = O.HeadModel(96, 32, 16, lr=0.1, seed=4, dtype=np.float64)
    tw = [torch.tensor(w.copy(), dtype=torch.float64, requires_grad=True) for w in m.get_weights()]
    opt = torch.optim.Adadelta(tw, lr=0.1, rho=0.95, eps=1e-8)
    for step in range(5):
        L, R = rng.randn(16, 96), rng.randn(16, 96)
        lab = rng.randint(0, 2, 16)
        y = O.to_categorical(lab).astype(np.float64)
        cw = {0: 0.25, 1: 0.75}
        out = m.train_on_batch([L, R], y, class_weight=cw)
        opt.zero_grad()
        sw = torch.tensor([cw[int(c)] for c in lab], dtype=torch.float64)
        loss, _ = _keras_graph_loss(tw, torch.tensor(L), torch.tensor(R), torch.tensor(y), sw)
        loss.backward()
        opt.step()
        assert abs(out[0] - float(loss)) < 1e-12
    for a, b in zip(m.get_weights(), tw):
        np.testing.assert_allclose(a, b.detach().numpy(), atol=1e-12)

"""GPU: the two labelled EXTENSIONS of csrc/margin.hip (additive angular margin softmax, pairwise-L2 contrastive
loss — named by BASELINE.json's north_star, absent from the reference: SURVEY.md §0) against torch autograd in
float64 on the CPU.  There is no reference arithmetic to be on par with; the formulas are the published ones
(ArcFace, Deng et al. 2019 as implemented by insightface's margin softmax; Keras' mnist_siamese contrastive loss)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _arcface_ref(e, w, y, s, m, easy):
    eh = e / e.norm(dim=1, keepdim=True)
    wh = w / w.norm(dim=1, keepdim=True)
    cos = eh @ wh.t()
    ct = cos.gather(1, y.view(-1, 1)).squeeze(1)
    sin_t = torch.sqrt(torch.clamp(1 - ct * ct, min=1e-12))
    phi = ct * math.cos(m) - sin_t * math.sin(m)
    if easy:
        new = torch.where(ct > 0, phi, ct)
    else:
        new = torch.where(ct > math.cos(math.pi - m), phi, ct - math.sin(math.pi - m) * m)
    logits = s * cos.scatter(1, y.view(-1, 1), new.view(-1, 1))
    return torch.nn.functional.cross_entropy(logits, y)


@pytest.mark.parametrize("n,d,c,s,m,easy", [(64, 512, 1000, 64.0, 0.5, False), (37, 512, 85, 30.0, 0.35, False),
                                            (16, 128, 7, 16.0, 0.5, True), (5, 64, 2, 8.0, 1.2, False)])
def test_arcface_margin_loss_matches_autograd(gpu, n, d, c, s, m, easy):
    from a_link_amd import extensions as X
    rng = np.random.RandomState(n + c)
    e = rng.randn(n, d).astype(np.float32) * 3
    w = rng.randn(c, d).astype(np.float32) * 0.1
    y = rng.randint(0, c, n)
    # a few targets on the far side of the margin threshold / at negative cosine (the non-cond branches)
    e[0] = -w[y[0]] * 5 + 0.01 * rng.randn(d)
    e[1] = w[y[1]] * 2 + 0.01 * rng.randn(d)
    loss, de, dw = X.arcface_margin_loss(torch.from_numpy(e).cuda(), torch.from_numpy(w).cuda(), y, s, m, easy)
    te = torch.tensor(e, dtype=torch.float64, requires_grad=True)
    tw = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    ref = _arcface_ref(te, tw, torch.tensor(y, dtype=torch.long), s, m, easy)
    ref.backward()
    assert abs(float(loss) - ref.item()) < 2e-5 * max(1.0, abs(ref.item()))
    for got, want in ((de, te.grad), (dw, tw.grad)):
        scale = float(want.abs().max())
        assert float((got.cpu().double() - want).abs().max()) < 2e-5 * scale + 1e-9, scale
    # loss only
    l2, a, b = X.arcface_margin_loss(e, w, y, s, m, easy, need_grads=False)
    assert a is None and b is None and float(l2) == float(loss)


def _contrastive_ref(l, r, y, margin):
    d = torch.sqrt(torch.clamp(((l - r) ** 2).sum(1), min=1e-7))
    per = y * d * d + (1 - y) * torch.clamp(margin - d, min=0) ** 2
    return per.mean(), per


@pytest.mark.parametrize("p,d,margin", [(1000, 512, 1.0), (33, 2048, 0.5), (4, 64, 2.0), (100000, 512, 1.24)])
def test_contrastive_loss_matches_autograd(gpu, p, d, margin):
    from a_link_amd import extensions as X
    rng = np.random.RandomState(p)
    l = rng.randn(p, d).astype(np.float32)
    l /= np.linalg.norm(l, axis=1, keepdims=True)
    r = (l + rng.randn(p, d).astype(np.float32) * rng.uniform(0, 0.08, (p, 1)).astype(np.float32))
    y = (rng.rand(p) < 0.5).astype(np.float32)
    r[0] = l[0]                                        # identical pair: distance clamped at sqrt(1e-7), zero gradient
    loss, pair, dl, dr = X.contrastive_loss(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda(), y, margin)
    tl = torch.tensor(l, dtype=torch.float64, requires_grad=True)
    tr = torch.tensor(r, dtype=torch.float64, requires_grad=True)
    ref, per = _contrastive_ref(tl, tr, torch.tensor(y, dtype=torch.float64), margin)
    ref.backward()
    assert abs(float(loss) - ref.item()) < 1e-5 * max(1.0, ref.item())
    assert float((pair.cpu().double() - per.detach()).abs().max()) < 1e-5
    for got, want in ((dl, tl.grad), (dr, tr.grad)):
        assert float((got.cpu().double() - want).abs().max()) < 1e-5 * float(want.abs().max()) + 1e-10
    assert float(dl[0].abs().max()) == 0.0 and torch.equal(dl, -dr)


def test_extension_error_paths(gpu):
    from a_link_amd import extensions as X
    with pytest.raises(ValueError):
        X.arcface_margin_loss(np.zeros((4, 64), np.float32), np.zeros((3, 64), np.float32), [0, 1, 2, 3])
    with pytest.raises(gpu.AlinkError):
        X.contrastive_loss(np.zeros((4, 66), np.float32), np.zeros((4, 66), np.float32), np.zeros(4))

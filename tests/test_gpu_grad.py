"""GPU: the backbone's input-gradient pass (FGSM / PGD extension; csrc/backward.hip + the backward-mode
convolutions) against torch autograd through the unfused CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _autograd(params, x_nhwc, demb):
    from oracle import ir_resnet
    x = torch.from_numpy(np.transpose(x_nhwc, (0, 3, 1, 2)).copy()).double().requires_grad_(True)
    z = ir_resnet.forward_raw({k: np.asarray(v, np.float64) for k, v in params.items()}, x, dtype=torch.float64)
    e = z / z.norm(dim=1, keepdim=True)
    (e * torch.from_numpy(demb).double()).sum().backward()
    return e.detach().numpy(), np.transpose(x.grad.numpy(), (0, 2, 3, 1))


@pytest.mark.parametrize("units,size,dtype,tol", [((1, 1, 1, 1), (32, 32), "f16", 2e-3), ((2, 2, 2, 1), (48, 32), "bf16", 2e-2)])
def test_input_gradient_matches_autograd(gpu, units, size, dtype, tol):
    from a_link_amd import weights as W
    from a_link_amd.backbone import IRBackbone
    params = W.synthetic_ir_params(units, size=size, seed=4)
    bb = IRBackbone(params, image_size=size, dtype=dtype, max_batch=8, enable_grad=True)
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (5,) + size + (3,)).astype(np.float32)
    demb = rng.standard_normal((5, 512)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    emb = bb.embed_with_cache(xd)
    assert torch.equal(emb, bb.embed_device(xd))                     # the cached forward is the same forward
    g = bb.input_gradient(torch.from_numpy(demb).cuda()).cpu().numpy()
    e_ref, g_ref = _autograd(params, x, demb)
    assert g.shape == x.shape
    a, b = g.reshape(5, -1).astype(np.float64), g_ref.reshape(5, -1)
    cos = (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    rel = np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)
    assert cos.min() > 1 - tol and rel.max() < 4 * np.sqrt(tol), (cos, rel)
    # NCHW input -> NCHW gradient, same numbers
    xc = xd.permute(0, 3, 1, 2).contiguous()
    bb.embed_with_cache(xc)
    gc = bb.input_gradient(torch.from_numpy(demb).cuda()).permute(0, 2, 3, 1).cpu().numpy()
    assert np.array_equal(gc, g)


def test_grad_needs_enable_and_nonnegative_slopes(gpu):
    from a_link_amd import _abi, weights as W
    from a_link_amd.backbone import IRBackbone
    params = W.synthetic_ir_params((1, 1, 1, 1), size=(32, 32), seed=1)
    bb = IRBackbone(params, image_size=(32, 32), max_batch=2)
    with pytest.raises(_abi.AlinkError):
        bb.embed_with_cache(torch.zeros((1, 32, 32, 3), device="cuda"))
    bad = dict(params)
    bad["stage2_unit1_relu1_gamma"] = -np.abs(params["stage2_unit1_relu1_gamma"])
    with pytest.raises(_abi.AlinkError):
        IRBackbone(bad, image_size=(32, 32), max_batch=2, enable_grad=True)


def test_fgsm_and_pgd_move_the_pair_score(gpu):
    """The extension attacks: within the eps-ball, inside [0, 255], and the targeted step raises the
    scorer's probability of the target class on (almost) every pair."""
    from a_link_amd import noise as N, siamese
    size = (32, 32)
    fm = siamese.ArcFace(size, "synthetic:r18:3", enable_grad=True, max_batch=16)
    pm = siamese.SiameseNetwork((512,), "m2", 0.1, seed=4)
    rng = np.random.RandomState(0)
    L = rng.randint(0, 256, (24,) + size + (3,)).astype(np.float32)
    R = rng.randint(0, 256, (24,) + size + (3,)).astype(np.float32)
    target = rng.randint(0, 2, 24)
    before = pm.predict([fm.process(L), fm.process(R)])[np.arange(24), target]
    for cls, kw in ((N.FGSM, dict(eps=6.0)), (N.PGD, dict(eps=6.0, alpha=2.0, steps=4, seed=1))):
        att = N.get_relevant_noise(cls.__name__.lower())(model=pm, sess=None, feature_model=fm, **kw)
        al, ar = att.addPairNoise([L, R], target)
        assert al.shape == L.shape and ar.shape == R.shape
        assert np.abs(al - L).max() <= 6.0 + 1e-4 and np.abs(ar - R).max() <= 6.0 + 1e-4
        assert al.min() >= 0 and al.max() <= 255 and ar.min() >= 0 and ar.max() <= 255
        after = pm.predict([fm.process(al), fm.process(ar)])[np.arange(24), target]
        assert (after > before).mean() >= 0.9 and after.mean() > before.mean()
    # untargeted: the probability of the TRUE label goes down
    att = N.FGSM(model=pm, feature_model=fm, eps=6.0, targeted=False)
    al, ar = att.addPairNoise([L, R], target)
    after = pm.predict([fm.process(al), fm.process(ar)])[np.arange(24), target]
    assert (after < before).mean() >= 0.9
    with pytest.raises(TypeError):
        N.FGSM(model=pm, feature_model=siamese.ArcFace(size, "synthetic:r18:3")).addPairNoise([L, R], target)

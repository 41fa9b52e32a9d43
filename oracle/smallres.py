"""oracle/smallres.py — torch-CPU (autograd) restatement of SmallRes (reference code/siamese.py:134-170).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED by the reference (Keras 2.1.2 layers,
not vendored).  Keras semantics: Conv2D 'same'/'valid', MaxPooling2D(2,2) floor, Dropout(0.25) in
training = mask/0.75, Flatten of NHWC is (h,w,c)-major, Dense kernels (in,out); loss/metrics/Adadelta
as in oracle/siamese_head.py.  Weights are in Keras order and layout (conv kernels (3,3,in,out)).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import siamese_head as O


def _tower(ws, x_nhwc, masks=None):
    x = x_nhwc.permute(0, 3, 1, 2)
    w = [w.permute(3, 2, 0, 1) for w in (ws[0], ws[2], ws[4], ws[6])]      # (kh,kw,in,out) -> (out,in,kh,kw)
    x = F.relu(F.conv2d(x, w[0], ws[1], padding=1))
    x = F.relu(F.conv2d(x, w[1], ws[3], padding=0))
    x = F.max_pool2d(x, 2)
    if masks is not None:
        x = x * masks[0].permute(0, 3, 1, 2) / 0.75
    x = F.relu(F.conv2d(x, w[2], ws[5], padding=1))
    x = F.relu(F.conv2d(x, w[3], ws[7], padding=0))
    x = F.max_pool2d(x, 2)
    if masks is not None:
        x = x * masks[1].permute(0, 3, 1, 2) / 0.75
    x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)                       # Keras Flatten: (h,w,c)
    return F.relu(x @ ws[8] + ws[9])


def _split_masks(masks, n, shapes):
    """masks: flat u8 [2n*e1 | 2n*e2] in [L ; R] image order -> per-branch float tensors (n,h,w,c)."""
    (h1, w1, c1), (h2, w2, c2) = shapes
    e1, e2 = h1 * w1 * c1, h2 * w2 * c2
    m1 = torch.as_tensor(masks[:2 * n * e1].astype(np.float32)).reshape(2 * n, h1, w1, c1)
    m2 = torch.as_tensor(masks[2 * n * e1:].astype(np.float32)).reshape(2 * n, h2, w2, c2)
    return (m1[:n], m2[:n]), (m1[n:], m2[n:])


def forward(ws, L, R, masks=None, mask_shapes=None):
    t = [torch.as_tensor(w) if not isinstance(w, torch.Tensor) else w for w in ws]
    L = torch.as_tensor(np.asarray(L, np.float32))
    R = torch.as_tensor(np.asarray(R, np.float32))
    mL = mR = None
    if masks is not None:
        mL, mR = _split_masks(masks, len(L), mask_shapes)
    fl, fr = _tower(t, L, mL), _tower(t, R, mR)
    d = (fl - fr).abs()
    h = F.relu(d @ t[10] + t[11])
    h = F.relu(h @ t[12] + t[13])
    return F.softmax(h @ t[14] + t[15], dim=1)


def loss_fn(p, y, sw=None):
    """Keras binary_crossentropy on the softmax output + sample-weighted batch mean."""
    y = torch.as_tensor(np.asarray(y, np.float32))
    pc = torch.clamp(p, 1e-7, 1 - 1e-7)
    x = torch.log(pc / (1 - pc))
    l = (torch.clamp(x, min=0) - x * y + torch.log1p(torch.exp(-x.abs()))).mean(dim=1)
    w = torch.ones(len(y)) if sw is None else torch.as_tensor(np.asarray(sw, np.float32))
    return (l * w).mean() / (w != 0).float().mean()


class SmallResModel(object):
    def __init__(self, ws, lr=1.0, rho=0.95, epsilon=1e-8):
        self.ws = [np.asarray(w, np.float32).copy() for w in ws]
        self.opt = O.Adadelta([w.shape for w in self.ws], lr, rho, epsilon)

    def predict(self, X):
        with torch.no_grad():
            return forward(self.ws, X[0], X[1]).numpy()

    def train_on_batch(self, x, y, sample_weight=None, masks=None, mask_shapes=None):
        t = [torch.tensor(w, requires_grad=True) for w in self.ws]
        p = forward(t, x[0], x[1], masks, mask_shapes)
        loss = loss_fn(p, y, sample_weight)
        loss.backward()
        gs = [w.grad.numpy() for w in t]
        acc = float((torch.round(p.detach()) == torch.as_tensor(np.asarray(y, np.float32))).float().mean())
        self.ws = self.opt.step(self.ws, gs)
        return [float(loss.detach()), acc], gs

    def test_on_batch(self, x, y):
        with torch.no_grad():
            p = forward(self.ws, x[0], x[1])
            loss = loss_fn(p, y)
            acc = float((torch.round(p) == torch.as_tensor(np.asarray(y, np.float32))).float().mean())
        return [float(loss), acc]

"""extensions — two training losses BASELINE.json's north_star names and the REFERENCE DOES NOT HAVE:

    arcface_margin_loss   additive angular margin softmax over L2-normalised embeddings and class centres
                          (the reference cuts the ArcFace checkpoint at fc1_output and never builds this head:
                          reference code/face_model.py:35-36,53)
    contrastive_loss      pairwise-L2 contrastive loss (the reference's pair scorer is |l - r| -> Dense -> softmax with
                          binary cross-entropy: reference code/siamese.py:27-35)

Labelled extensions like noise.FGSM / noise.PGD: nothing in the drop-in path uses them; they exist for callers that
fine-tune the embedding space itself.  HIP kernels in csrc/margin.hip (exact-f32 MFMA GEMMs of csrc/sgemm.hip for the
cosine matrix and its two gradient products); checked against torch autograd in tests/test_gpu_extensions.py.
"""
import ctypes as C

from . import _abi


def _device_of(x):
    import torch
    if isinstance(x, torch.Tensor) and x.is_cuda:
        return x.device
    return torch.device("cuda", torch.cuda.current_device())


def _f32(t, dev):
    import torch
    return torch.as_tensor(t).to(dev, torch.float32).contiguous()


def arcface_margin_loss(emb, weight, labels, s=64.0, m=0.5, easy_margin=False, need_grads=True):
    """emb (N, D), weight (C, D) [raw, normalised inside], labels (N,) int.  Returns (loss, d_emb, d_weight) as CUDA
    tensors (gradients None when need_grads is False).  s, m: insightface's defaults."""
    import torch
    dev = _device_of(emb)
    lib = _abi.init(dev.index)
    e, w = _f32(emb, dev), _f32(weight, dev)
    y = torch.as_tensor(labels).to(dev, torch.int32).contiguous()
    n, d = e.shape
    c = w.shape[0]
    if w.shape[1] != d or y.numel() != n:
        raise ValueError("shapes: emb %s weight %s labels %s" % (tuple(e.shape), tuple(w.shape), tuple(y.shape)))
    if int(y.min()) < 0 or int(y.max()) >= c:
        raise ValueError("labels outside 0..%d" % (c - 1))
    nb = lib.alink_arcface_margin_workspace_bytes(n, d, c)
    ws = torch.empty(nb + 256, dtype=torch.uint8, device=dev)
    off = (-ws.data_ptr()) % 256
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    de = torch.empty_like(e) if need_grads else None
    dw = torch.empty_like(w) if need_grads else None
    _abi.check(lib.alink_arcface_margin_loss(_abi.ptr(e), _abi.ptr(w), _abi.ptr(y), n, d, c, float(s), float(m),
                                             1 if easy_margin else 0, _abi.ptr(loss), _abi.ptr(de), _abi.ptr(dw),
                                             C.c_void_p(ws.data_ptr() + off), nb, _abi.current_stream(dev)),
               "alink_arcface_margin_loss")
    return loss[0], de, dw


def contrastive_loss(left, right, y, margin=1.0, need_grads=True):
    """left, right (P, D), y (P,) or (P, 1) in {0, 1} (1 = same identity).  Returns (loss, per-pair terms, d_left,
    d_right) as CUDA tensors."""
    import torch
    dev = _device_of(left)
    lib = _abi.init(dev.index)
    l, r = _f32(left, dev), _f32(right, dev)
    yy = _f32(y, dev).reshape(-1)
    p, d = l.shape
    if r.shape != l.shape or yy.numel() != p:
        raise ValueError("shapes: left %s right %s y %s" % (tuple(l.shape), tuple(r.shape), tuple(yy.shape)))
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    pair = torch.empty(p, dtype=torch.float32, device=dev)
    dl = torch.empty_like(l) if need_grads else None
    dr = torch.empty_like(r) if need_grads else None
    _abi.check(lib.alink_contrastive_loss(_abi.ptr(l), _abi.ptr(r), _abi.ptr(yy), p, d, float(margin), _abi.ptr(loss),
                                          _abi.ptr(pair), _abi.ptr(dl), _abi.ptr(dr), _abi.current_stream(dev)),
               "alink_contrastive_loss")
    return loss[0], pair, dl, dr

"""noise — drop-in for reference code/noise.py: the A2-LINK perturbations, computed on the GPU.

    Noise / Gaussian / SaltPepper / Poisson / Speckle / Perlin      code/noise.py:10-150
    PredictionWrappedModel                                          code/noise.py:153-168
    AdversarialNoise                                                code/noise.py:171-188
    get_relevant_noise(name)                                        code/noise.py:191-205

Same class names, constructor keywords and methods (addIndividualNoise / addNoise / addPairNoise).
Differences, all deliberate:
  * a whole batch is one kernel launch (the reference loops images on the host, code/noise.py:20-24);
  * random numbers come from a counter-based Philox stream on the device (csrc/noise.hip); the
    reference's np.random global stream is unseeded, so only the distributions are
    contractual.  Every object takes an optional `seed`; successive calls use successive streams;
  * outputs are float32 (the reference's are float64 because np.random returns doubles; the models
    cast to float32 on entry anyway: code/face_model.py:88);
  * SaltPepper uses the tuple-index meaning the reference's list index had under its NumPy
    (SURVEY.md §0); Perlin raises the reference's ValueError for sizes its reshape cannot take
    (112 x 112: code/noise.py:96,130).
NumPy in -> NumPy out; CUDA tensor in -> CUDA tensor out.

Row ranges (one process per GPU, alink_loop with `group`): `addPairNoise(pairs, labels, rows=(lo, total))` /
`addNoise(images, labels, first_row=lo)` perturb rows lo : lo + len of a logical batch and return exactly what rows
lo : lo + len of the whole-batch call would have held — every draw is keyed by (stream seed, GLOBAL element / image
index), so the noise a pair receives does not depend on how many ranks share the batch.
"""
import ctypes as C

import numpy as np

from . import _abi


def _as_device(images, device):
    import torch
    if isinstance(images, torch.Tensor):
        return images.to("cuda:%d" % device, torch.float32).contiguous(), True
    a = np.ascontiguousarray(np.asarray(images), dtype=np.float32)
    return torch.from_numpy(a).to("cuda:%d" % device), False


def _ret(t, as_torch):
    return t if as_torch else t.cpu().numpy()


class Noise(object):
    def __init__(self, model=None, sess=None, feature_model=None, seed=None, device=None):
        self.model = model
        self.sess = sess
        self.feature_model = feature_model
        self.device = _abi.resolve_device(device)         # None: the current torch device
        self._seed = int(np.random.randint(0, 2 ** 31 - 1)) if seed is None else int(seed)
        self._calls = 0

    def _next_seed(self):
        s = (self._seed + 0x9E3779B97F4A7C15 * self._calls) & 0xFFFFFFFFFFFFFFFF
        self._calls += 1
        return s

    # what alink_loop's multi-rank iteration needs from a noise object: row-range calls (above) and a stream state
    # that can be made rank 0's on every rank
    supports_rows = True

    def stream_state(self):
        return (self._seed, self._calls)

    def set_stream_state(self, st):
        self._seed, self._calls = int(st[0]), int(st[1])

    def _stream(self):
        return _abi.current_stream(self.device)

    # batch hook: subclasses implement _apply(dev_images (n,H,W,C) f32, first = global index of image 0) -> dev tensor
    def _apply(self, x, first=0):
        return x.clone()

    def addIndividualNoise(self, image, target_labels=None):
        import torch
        if isinstance(image, torch.Tensor):
            return self.addNoise(image[None], None)[0]
        return self.addNoise(np.asarray(image)[None], None)[0]

    def addNoise(self, images, target_labels, first_row=0):
        if len(images) == 0:
            self._next_seed()                    # an empty shard still consumes the call's stream: ranks stay in step
            return images if hasattr(images, "detach") else np.array(images)
        x, as_torch = _as_device(images if not isinstance(images, (list, tuple)) else np.stack(images), self.device)
        if x.ndim != 4:
            raise ValueError("expected images of shape (n, H, W, C), got %s" % (tuple(x.shape),))
        return _ret(self._apply(x, int(first_row)), as_torch)

    def addPairNoise(self, image_pairs, target_labels, rows=None):
        first = 0 if rows is None else int(rows[0])
        left_half = self.addNoise(image_pairs[0], target_labels, first)
        right_half = self.addNoise(image_pairs[1], target_labels, first)
        return [left_half, right_half]


class Gaussian(Noise):
    def __init__(self, mean=10, var=10, model=None, sess=None, feature_model=None, seed=None, device=None):
        super(Gaussian, self).__init__(seed=seed, device=device)
        self.mean = mean
        self.var = var
        self.sigma = self.var ** 0.5

    def _apply(self, x, first=0):
        import torch
        lib = _abi.init(self.device)
        out = torch.empty_like(x)
        _abi.check(lib.alink_noise_gaussian(_abi.ptr(x), _abi.ptr(out), x.numel(), float(self.mean), float(self.sigma),
                                            self._next_seed(), first * x[0].numel(), self._stream()), "alink_noise_gaussian")
        return out


class Speckle(Noise):
    def __init__(self, model=None, sess=None, feature_model=None, seed=None, device=None):
        super(Speckle, self).__init__(seed=seed, device=device)

    def _apply(self, x, first=0):
        import torch
        lib = _abi.init(self.device)
        out = torch.empty_like(x)
        _abi.check(lib.alink_noise_speckle(_abi.ptr(x), _abi.ptr(out), x.numel(), 15.0, self._next_seed(),
                                           first * x[0].numel(), self._stream()), "alink_noise_speckle")
        return out


class SaltPepper(Noise):
    def __init__(self, s_vs_p=0.5, amount=0.004, model=None, sess=None, feature_model=None, seed=None, device=None):
        super(SaltPepper, self).__init__(seed=seed, device=device)
        self.s_vs_p = s_vs_p
        self.amount = amount

    def counts(self, shape):
        size = int(np.prod(shape))
        return (int(np.ceil(self.amount * size * self.s_vs_p)), int(np.ceil(self.amount * size * (1. - self.s_vs_p))))

    def _apply(self, x, first=0):
        import torch
        lib = _abi.init(self.device)
        n, H, W, Cc = x.shape
        if min(H, W, Cc) < 2:
            raise ValueError("low >= high")                   # np.random.randint(0, i - 1) (code/noise.py:59)
        n_salt, n_pepper = self.counts((H, W, Cc))
        out = torch.empty_like(x)
        _abi.check(lib.alink_noise_saltpepper(_abi.ptr(x), _abi.ptr(out), n, H, W, Cc, n_salt, n_pepper,
                                              self._next_seed(), first, self._stream()), "alink_noise_saltpepper")
        return out


class Poisson(Noise):
    def __init__(self, model=None, sess=None, feature_model=None, seed=None, device=None):
        super(Poisson, self).__init__(seed=seed, device=device)
        self.last_vals = None

    def _apply(self, x, first=0):
        import torch
        lib = _abi.init(self.device)
        n = x.shape[0]
        per = x[0].numel()
        nbytes = lib.alink_noise_poisson_scratch_bytes(n, per)
        scratch = torch.empty(nbytes // 4 + 1, dtype=torch.int32, device=x.device)
        vals = torch.empty(n, dtype=torch.float32, device=x.device)
        out = torch.empty_like(x)
        _abi.check(lib.alink_noise_poisson(_abi.ptr(x), _abi.ptr(out), n, per, self._next_seed(), first, _abi.ptr(scratch),
                                           nbytes, _abi.ptr(vals), self._stream()), "alink_noise_poisson")
        # np.random.poisson raises for a negative rate (code/noise.py:75): the sampling kernel raises a flag in the first
        # word of the scratch when it meets one — one 4-byte read-back instead of a reduction over the batch before the launch
        if int(scratch[0]) != 0:
            raise ValueError("lam < 0")
        self.last_vals = vals
        return out


class Perlin(Noise):
    def __init__(self, model=None, sess=None, feature_model=None, seed=None, device=None):
        super(Perlin, self).__init__(seed=seed, device=device)

    @staticmethod
    def octaves(row):
        return [56, 32, 16] if row % 56 == 0 else [50, 30, 15]  # code/noise.py:144-147

    def _apply(self, x, first=0, vectors=None):
        import torch
        lib = _abi.init(self.device)
        n, row, col, Cc = x.shape
        assert row == col                                         # code/noise.py:143
        ns = self.octaves(row)
        for s in ns:
            nc = int(row / s)
            if nc * s != row:                                     # m.reshape(nc, ns, nc, ns) (code/noise.py:130)
                raise ValueError("cannot reshape array of size %d into shape (%d,%d,%d,%d)" % (row * row, nc, s, nc, s))
        ns3 = (C.c_int * 3)(*ns)
        nodes = lib.alink_perlin_nodes(row, ns3)
        if vectors is None:
            vectors = torch.empty((n, nodes, 2), dtype=torch.float32, device=x.device)
            _abi.check(lib.alink_perlin_vectors(n, nodes, self._next_seed(), first, _abi.ptr(vectors), self._stream()),
                       "alink_perlin_vectors")
        else:
            vectors = vectors.to(x.device, torch.float32).contiguous()
            assert tuple(vectors.shape) == (n, nodes, 2)
        out = torch.empty_like(x)
        _abi.check(lib.alink_noise_perlin(_abi.ptr(x), _abi.ptr(out), n, row, Cc, ns3, _abi.ptr(vectors),
                                          self._stream()), "alink_noise_perlin")
        return out


def resize_images(images, new_size, device=None):
    """cv2.resize(image, new_size) per image (code/committee.py:22-26; readMTP.resizeImages,
    code/readMTP.py:116-119): new_size = (width, height), bilinear."""
    import torch
    if len(images) == 0:
        # an empty shard (more ranks than pairs): nothing to resize — but the result has the shape its peers' shards have,
        # (0, height, width, C), whatever noise produced the empty input (some returned [] -> shape (0,))
        Wo, Ho = int(new_size[0]), int(new_size[1])
        if hasattr(images, "detach"):
            ch = images.shape[-1] if images.dim() == 4 else 3
            return images.new_zeros((0, Ho, Wo, ch), dtype=torch.float32)
        a = np.asarray(images)
        return np.zeros((0, Ho, Wo, a.shape[-1] if a.ndim == 4 else 3), np.float32)
    device = _abi.resolve_device(device)
    x, as_torch = _as_device(images if not isinstance(images, (list, tuple)) else np.stack(images), device)
    n, H, W, Cc = x.shape
    Wo, Ho = int(new_size[0]), int(new_size[1])
    if (Ho, Wo) == (H, W):
        return _ret(x.clone(), as_torch)
    lib = _abi.init(device)
    out = torch.empty((n, Ho, Wo, Cc), dtype=torch.float32, device=x.device)
    _abi.check(lib.alink_resize_bilinear(_abi.ptr(x), _abi.ptr(out), n, H, W, Cc, Ho, Wo, _abi.current_stream()),
               "alink_resize_bilinear")
    return _ret(out, as_torch)


class PredictionWrappedModel:
    """code/noise.py:153-168: X = stacked pair images (2H, W, 3); split, embed both halves, score."""

    def __init__(self, model, feature_model):
        self.model = model
        self.feature_model = feature_model

    def predict(self, X):
        half = X[0].shape[0] // 2                                # Python-2 integer division (code/noise.py:160)
        left_half = [p[:half] for p in X]
        right_half = [p[half:] for p in X]
        if self.feature_model:
            left_features = self.feature_model.process(left_half)
            right_features = self.feature_model.process(right_half)
        else:
            left_features = left_half
            right_features = right_half
        return self.model.predict([left_features, right_features])


class AdversarialNoise(Noise):
    """code/noise.py:171-188.  pixel_count / maxiter / popsize: the reference's attack_all defaults (code/attack.py:91),
    exposed so that a test can run a short search.  Every pair's search has its own random stream, seeded by (this
    object's stream, the pair's GLOBAL row): the reference seeds nothing (only distributions are contractual), and a
    rank that attacks rows lo : hi of the batch (rows=(lo, total)) finds what the whole-batch call finds for them —
    the search is 0.5 s per pair at the defaults, the part of an A2-LINK iteration that most needs every GPU."""

    def __init__(self, model, sess, feature_model, seed=None, device=None, pixel_count=40, maxiter=50, popsize=250, search="exact",
                 lockstep=32):
        super(AdversarialNoise, self).__init__(model, sess, feature_model, seed=seed, device=device)
        from . import attack
        self.e2e_model = PredictionWrappedModel(model, feature_model)
        # search="screen": the search's candidates go through the feature model's 16-bit screening form (attack._DevicePairScorer)
        # lockstep: searches advanced together on the device (attack._LockstepEngine; 0 = one pair after another, the
        # reference's shape) — the attacked images do not depend on it
        self.attacker = attack.PixelAttacker(self.e2e_model, search=search, lockstep=lockstep)
        self.search = dict(pixel_count=pixel_count, maxiter=maxiter, popsize=popsize)

    def addPairNoise(self, image_pairs, target_labels, rows=None):
        first = 0 if rows is None else int(rows[0])
        base = self._next_seed()
        n = len(image_pairs[0])
        if n == 0:                                  # an empty shard: empty sides of the input's own shape (and container kind)
            return [image_pairs[0], image_pairs[1]]
        on_device = all(hasattr(p, "detach") for p in image_pairs) and self.attacker.lockstep > 0
        if on_device:
            import torch
            concat_data = torch.cat((image_pairs[0].float(), image_pairs[1].float()), dim=1)      # (n, 2H, W, 3): the stacked pairs, on the device
            img_shape = tuple(image_pairs[0].shape[1:])
        else:
            image_pairs = [p.detach().cpu().numpy() if hasattr(p, "detach") else p for p in image_pairs]
            concat_data = [np.concatenate((image_pairs[0][i], image_pairs[1][i]), axis=0) for i in range(n)]
            img_shape = image_pairs[0][0].shape
        # splitmix64 of (stream, global row) -> a 32-bit RandomState seed per pair
        seeds = []
        for i in range(n):
            z = (base + 0x9E3779B97F4A7C15 * (first + i + 1)) & 0xFFFFFFFFFFFFFFFF
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
            seeds.append(int((z ^ (z >> 31)) & 0xFFFFFFFF))
        perturbed = self.attacker.attack_all(concat_data, target_labels, dimensions=(2 * img_shape[0], img_shape[1]),
                                             seeds=seeds, **self.search)
        if hasattr(perturbed, "detach"):                     # device tensor out: the two halves as views
            h = perturbed.shape[1] // 2
            return [perturbed[:, :h].contiguous(), perturbed[:, h:].contiguous()]
        left_half = [p[:p.shape[0] // 2] for p in perturbed]
        right_half = [p[p.shape[0] // 2:] for p in perturbed]
        return [left_half, right_half]


class FGSM(Noise):
    """EXTENSION — not in the reference (its only attack is the black-box few-pixel search above);
    BASELINE.json's north_star and SURVEY.md §8f N1 name it.  One signed-gradient step on BOTH images of
    every pair through the frozen feature model and the pair scorer:
        x' = clip(x -/+ eps * sign(d BCE(scorer(f(xl), f(xr)), target) / dx), 0, 255)
    targeted (default, like PixelAttacker.attack_all: target class = argmax(target_labels[i])) steps
    DOWN the loss of the target class; untargeted steps UP the loss of the given labels.
    Needs an ArcFace-style feature model built with gradients (`ArcFace(..., enable_grad=True)`) and a
    DenseHead-backed pair model."""
    steps, alpha, random_start = 1, None, False

    def __init__(self, model=None, sess=None, feature_model=None, eps=4.0, targeted=True, clip=(0.0, 255.0),
                 seed=None, device=None):
        super(FGSM, self).__init__(model, sess, feature_model, seed=seed, device=device)
        self.eps, self.targeted, self.clip = float(eps), bool(targeted), clip

    def _parts(self):
        from .head import DenseHead
        bb = getattr(getattr(self.feature_model, "model", None), "model", None)
        head = getattr(self.model, "siamese_net", None)
        if bb is None or not getattr(bb, "grad_enabled", False) or not isinstance(head, DenseHead):
            raise TypeError("FGSM/PGD need ArcFace(..., enable_grad=True) and a DenseHead-backed pair model")
        return bb, head

    def _targets(self, target_labels, n, out_dim):
        t = np.asarray(target_labels)
        cls = t.argmax(axis=1) if (t.ndim == 2 and t.shape[1] > 1) else t.reshape(n).astype(int)
        if out_dim == 1:
            return cls.reshape(n, 1).astype(np.float32)
        y = np.zeros((n, out_dim), np.float32)
        y[np.arange(n), cls] = 1.0
        return y

    def addPairNoise(self, image_pairs, target_labels, rows=None):
        import torch
        first = 0 if rows is None else int(rows[0])
        if len(image_pairs[0]) == 0:                # an empty shard consumes the call's streams like any other
            if self.random_start:
                self._next_seed(), self._next_seed()
            return [image_pairs[0], image_pairs[1]]
        bb, head = self._parts()
        xl, as_torch = _as_device(image_pairs[0] if not isinstance(image_pairs[0], (list, tuple)) else np.stack(image_pairs[0]), self.device)
        xr, _ = _as_device(image_pairs[1] if not isinstance(image_pairs[1], (list, tuple)) else np.stack(image_pairs[1]), self.device)
        n = xl.shape[0]
        y = torch.from_numpy(self._targets(target_labels, n, head.out_dim)).to(xl.device)
        step = self.eps if self.alpha is None else float(self.alpha)
        sign = -1.0 if self.targeted else 1.0
        al, ar = xl.clone(), xr.clone()
        lib = _abi.init(self.device)
        al, ar = al.float().contiguous(), ar.float().contiguous()
        if self.random_start:
            # U(-eps, eps) keyed by (stream, GLOBAL element): a row's start does not depend on the batch it arrives in
            for t in (al, ar):
                _abi.check(lib.alink_noise_uniform(_abi.ptr(t), _abi.ptr(t), t.numel(), -self.eps, self.eps, self._next_seed(),
                                                   first * t[0].numel(), self._stream()), "alink_noise_uniform")
        mb = bb.max_batch
        for _ in range(self.steps):
            for s in range(0, n, mb):
                sl = slice(s, min(n, s + mb))
                er = bb.embed_device(ar[sl])
                el = bb.embed_with_cache(al[sl])                  # left side: forward with cache, backward
                dL, _ = head.input_gradients(el, er, y[sl])
                gl = bb.input_gradient(dL)
                er = bb.embed_with_cache(ar[sl])                  # right side against the un-stepped left
                _, dR = head.input_gradients(el, er, y[sl])
                gr = bb.input_gradient(dR)
                # the step, the projection into the eps-ball of the clean images and the pixel clip: one kernel, in place
                lo, hi = (self.clip if self.clip is not None else (float("-inf"), float("inf")))
                for adv, clean, grad in ((al, xl, gl), (ar, xr, gr)):
                    a_, c_, g_ = adv[sl], clean[sl].contiguous(), grad.contiguous()
                    _abi.check(lib.alink_pgd_step(_abi.ptr(a_), _abi.ptr(c_), _abi.ptr(g_), a_.numel(), sign * step, self.eps,
                                                  float(lo), float(hi), self._stream()), "alink_pgd_step")
        return [_ret(al, as_torch), _ret(ar, as_torch)]

    def addNoise(self, images, target_labels, first_row=0):
        raise TypeError("gradient attacks perturb pairs: use addPairNoise")


class PGD(FGSM):
    """EXTENSION — projected gradient descent: `steps` signed-gradient steps of size `alpha` inside the
    eps-ball (Madry et al. 2018), optional uniform random start."""

    def __init__(self, model=None, sess=None, feature_model=None, eps=4.0, alpha=1.0, steps=5, random_start=True,
                 targeted=True, clip=(0.0, 255.0), seed=None, device=None):
        super(PGD, self).__init__(model, sess, feature_model, eps=eps, targeted=targeted, clip=clip, seed=seed,
                                  device=device)
        self.alpha, self.steps, self.random_start = float(alpha), int(steps), bool(random_start)


def get_relevant_noise(noise_string):
    noise_mapping = {
        'gaussian': Gaussian,
        'saltpepper': SaltPepper,
        'poisson': Poisson,
        'speckle': Speckle,
        'plain': Noise,
        'perlin': Perlin,
        'adversarial': AdversarialNoise,
        'fgsm': FGSM,          # extensions, not in the reference's table (code/noise.py:192-200)
        'pgd': PGD,
    }
    if noise_string.lower() in noise_mapping:
        return noise_mapping[noise_string.lower()]
    else:
        raise NotImplementedError("%s noise is not implemented!" % (noise_string))

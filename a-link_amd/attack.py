"""attack — drop-in for reference code/attack.py: the black-box few-pixel attack of A2-LINK.

    perturb_image(xs, img)                                   code/attack.py:5-29
    PixelAttacker(model).predict_classes / attack_success / attack / attack_all   code/attack.py:32-103

The search is the reference's population-batched differential evolution (differential_evolution.py).
What the reference does per generation — write k pixels into `popsize` copies of the stacked pair
image on the host, slice every copy in two, embed 2 x popsize images ONE AT A TIME through MXNet and
score the pairs through Keras (code/noise.py:158-168, code/siamese.py:232-234) — is here one
perturb kernel that writes both halves as two contiguous device batches (alink_perturb_images,
split = 1), two batched backbone launch chains and one fused pair-scoring kernel; only the
population (popsize x 5k doubles) goes up and popsize energies come back.

The fast path needs `model` to be a noise.PredictionWrappedModel over an ArcFace-style feature model
(`.model.model.embed_device`) and a DenseHead-backed pair model; any other duck-typed model takes the
reference's generic route (perturb on device, hand host arrays to model.predict).
"""
import numpy as np

from . import _abi
from .differential_evolution import differential_evolution


def _perturb_device(xs, img_dev, split, device=None):
    """xs (n, 5k) float64 host, img_dev (Hc, W, 3) f32 device -> device tensor
    (n, Hc, W, 3) or, split, (2, n, Hc/2, W, 3)."""
    import torch
    device = _abi.resolve_device(device)
    lib = _abi.init(device)
    xs = np.ascontiguousarray(np.asarray(xs, dtype=np.float64))
    n, m = xs.shape
    if m % 5:
        raise ValueError("array split does not result in an equal division")       # np.split (code/attack.py:22)
    k = m // 5
    Hc, W, Cc = img_dev.shape
    assert Cc == 3
    xs_d = torch.from_numpy(xs).to(img_dev.device)
    shape = (2, n, Hc // 2, W, 3) if split else (n, Hc, W, 3)
    out = torch.empty(shape, dtype=torch.float32, device=img_dev.device)
    _abi.check(lib.alink_perturb_images(_abi.ptr(img_dev), _abi.ptr(xs_d), n, k, Hc, W, 1 if split else 0,
                                        _abi.ptr(out), _abi.current_stream()), "alink_perturb_images")
    return out


def perturb_image(xs, img, device=None):
    """n perturbation vectors [x, y, r, g, b] * k -> n perturbed copies of img (code/attack.py:5-29).
    NumPy in -> NumPy out (float32); CUDA tensor image in -> CUDA tensor out."""
    import torch
    xs = np.asarray(xs)
    if xs.ndim < 2:
        xs = np.array([xs])
    as_torch = isinstance(img, torch.Tensor)
    device = _abi.resolve_device(device)
    img_d = img.to("cuda:%d" % device, torch.float32).contiguous() if as_torch else \
        torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32)).to("cuda:%d" % device)
    r = xs.astype(int)[:, 0::5] if xs.shape[1] >= 5 else np.zeros((len(xs), 0), int)
    c = xs.astype(int)[:, 1::5] if xs.shape[1] >= 5 else np.zeros((len(xs), 0), int)
    if r.size and (r.min() < -img_d.shape[0] or r.max() >= img_d.shape[0] or c.min() < -img_d.shape[1]
                   or c.max() >= img_d.shape[1]):
        raise IndexError("perturbation pixel outside the %d x %d image" % (img_d.shape[0], img_d.shape[1]))
    out = _perturb_device(xs, img_d, False, device)
    return out if as_torch else out.cpu().numpy()


class _DevicePairScorer(object):
    """The fused objective: population -> P(class) of the perturbed pair, all on device.
    search="screen": the candidates of the search are embedded by the feature model's 16-bit SCREENING form where it has one
    (ArcFace.screen: 44 k embeddings/s against the exact mode's 15 k — the search is 20,400 backbone forwards per pair, 1.4 s in
    the exact mode).  The reference's search is a random one (differential evolution, unseeded: code/attack.py:81-83), so which
    arithmetic ranks its candidates is not contractual; the image it RETURNS is embedded like any other by whoever calls the
    attack.  Default "exact"."""

    def __init__(self, wrapped, image, device=None, search="exact"):
        import torch
        from .head import DenseHead
        fm = getattr(wrapped, "feature_model", None)
        bb = getattr(getattr(fm, "model", None), "model", None)           # ArcFace -> FaceModel -> IRBackbone
        if search == "screen" and getattr(fm, "screen", None) is not None:
            bb = fm.screen.model
        elif search not in ("exact", "screen"):
            raise ValueError("search must be exact or screen")
        head = getattr(getattr(wrapped, "model", None), "siamese_net", None)
        if bb is None or not hasattr(bb, "embed_device") or not isinstance(head, DenseHead):
            raise TypeError("no device fast path for this model")
        if not getattr(wrapped.model, "_identity_preprocess", False):
            raise TypeError("pair model preprocesses its inputs")
        self.bb, self.head = bb, head
        image = np.asarray(image, dtype=np.float32)
        Hc, W, _ = image.shape
        if Hc % 2 or (Hc // 2, W) != tuple(bb.image_size):
            raise TypeError("stacked pair image %s does not match the backbone input" % (image.shape,))
        device = _abi.resolve_device(device)
        self.img = torch.from_numpy(np.ascontiguousarray(image)).to("cuda:%d" % device)
        self.device = device

    def predict(self, xs):
        halves = _perturb_device(xs, self.img, True, self.device)
        n = halves.shape[1]
        emb = self.bb.embed_device(halves.reshape(2 * n, *halves.shape[2:]))
        return self.head.predict_device(emb[:n], emb[n:]).cpu().numpy()


class PixelAttacker:
    def __init__(self, model, rng_compat=False, seed=None, search="exact"):
        self.model = model
        self.rng_compat = rng_compat
        self.seed = seed
        self.search = search                      # "exact" | "screen": _DevicePairScorer

    def _scorer(self, img):
        try:
            return _DevicePairScorer(self.model, img, search=self.search)
        except TypeError:
            return None

    def predict_classes(self, xs, img, target_class, minimize=True, _scorer=None):
        if _scorer is not None:
            predictions = _scorer.predict(np.atleast_2d(xs))[:, target_class]
        else:
            imgs_perturbed = perturb_image(xs, img)
            predictions = self.model.predict(imgs_perturbed)[:, target_class]
        return predictions if minimize else 1 - predictions

    def attack_success(self, x, img, target_class, targeted_attack=False, verbose=False, _scorer=None):
        if _scorer is not None:
            confidence = _scorer.predict(np.atleast_2d(x))[0]
        else:
            attack_image = perturb_image(x, img)
            confidence = self.model.predict(attack_image)[0]
        predicted_class = np.argmax(confidence)
        if verbose:
            print('Confidence:', confidence[target_class])
        if ((targeted_attack and predicted_class == target_class) or
                (not targeted_attack and predicted_class != target_class)):
            return True

    def attack(self, image, actual_class, target, pixel_count, dimensions, maxiter=75, popsize=400, verbose=False,
               seed=None):
        """seed (not in the reference, which seeds nothing: code/attack.py:81-83): this search's own random stream,
        instead of the attacker's `self.seed` — what makes a pair's search independent of the pairs attacked before it"""
        seed = self.seed if seed is None else seed
        targeted_attack = target is not None
        target_class = target if targeted_attack else actual_class
        dim_x, dim_y = dimensions
        bounds = [(0, dim_x), (0, dim_y), (0, 256), (0, 256), (0, 256)] * pixel_count
        popmul = max(1, popsize // len(bounds))
        scorer = self._scorer(image)

        def predict_fn(xs):
            return self.predict_classes(xs, image, target_class, target is None, _scorer=scorer)

        def callback_fn(x, convergence):
            return self.attack_success(x, image, target_class, targeted_attack, verbose, _scorer=scorer)

        attack_result = differential_evolution(predict_fn, bounds, maxiter=maxiter, popsize=popmul, recombination=1,
                                               atol=-1, callback=callback_fn, polish=False, seed=seed,
                                               rng_compat=self.rng_compat)
        self.last_result = attack_result
        attack_image = perturb_image(attack_result.x, image)[0]
        return attack_image

    def attack_all(self, input_data, targets, dimensions, pixel_count=40, maxiter=50, popsize=250, verbose=False,
                   seeds=None):
        """seeds (optional, one per image): see attack() — a rank that attacks rows lo : hi of a pair batch with the
        seeds of those rows finds what the whole-batch call finds for them"""
        X = []
        for i, img in enumerate(input_data):
            target_class = np.argmax(targets[i])
            result = self.attack(img, 1 - target_class, target_class, pixel_count, dimensions, maxiter=maxiter,
                                 popsize=popsize, verbose=verbose, seed=None if seeds is None else seeds[i])
            X.append(result)
        return X

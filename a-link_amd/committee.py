"""committee — drop-in for reference code/committee.py.

Bagging.predict (code/committee.py:13-20) = sum of member predictions / number of members.  When every
member is a device-resident DenseHead the mean is fused on the GPU (alink_committee_forward: member
softmaxes accumulated in member order, one divide); any other duck-typed member falls back to the
reference's own host arithmetic on the members' outputs.
attackModel / resize (code/committee.py:22-37) drive the noise objects (noise.py) and resize with
the device bilinear kernel (cv2.resize INTER_LINEAR rule, alink_resize_bilinear).
"""
import numpy as np

from . import head as _head


class Bagging:
    def __init__(self, models, attacks):
        self.models = models
        self.attacks = []
        for attack in attacks:
            self.attacks.append(attack)

    def _device_heads(self):
        hs = [getattr(m, "siamese_net", None) for m in self.models]
        if all(isinstance(h, _head.DenseHead) for h in hs) and \
                all(getattr(m, "_identity_preprocess", False) for m in self.models):
            return hs
        return None

    def predict(self, predict_on):
        hs = self._device_heads()
        if hs is not None:
            out = _head.committee_predict_device(hs, predict_on[0], predict_on[1])
            return out if isinstance(predict_on[0], hs[0].torch.Tensor) else out.cpu().numpy()
        predictions = []
        for model in self.models:
            predictions.append(model.predict(predict_on))
        predicted = np.sum(np.array(predictions), axis=0) / len(self.models)
        return np.array(predicted)

    def predict_indexed(self, emb_left, emb_right, li, ri):
        """Extension: score pairs (li[p], ri[p]) gathered from embedding matrices on device.  emb_left /
        emb_right may be lists with one matrix per member (each member behind its own feature extractor)."""
        hs = self._device_heads()
        if hs is None:
            raise TypeError("predict_indexed needs DenseHead members")
        return _head.committee_predict_device(hs, emb_left, emb_right, li, ri)

    def resize(self, images, new_size):
        """cv2.resize(image, new_size) per image (code/committee.py:22-26); new_size = (width, height)."""
        from . import noise as _noise
        out = _noise.resize_images(images, new_size)
        return out if hasattr(out, "detach") else np.array(out)      # CUDA tensors stay on the device

    def attackModel(self, image_pairs, target_size, target_labels=None, rows=None):
        """code/committee.py:28-37: every noise perturbs the pair batch, both sides are resized to
        target_size; returns [[left per noise], [right per noise]].
        rows=(lo, total) (not in the reference): image_pairs / target_labels hold rows lo : lo + len of a batch of
        `total` pairs — one rank's shard; noise objects that take row ranges (noise.py: every one of this package)
        then draw for those rows what the whole-batch call would have drawn, any other duck-typed noise is called the
        reference's way on the rows it is given."""
        sides = ([], [])
        for attack in self.attacks:
            if rows is not None and getattr(attack, "supports_rows", False):
                noisy = attack.addPairNoise(image_pairs, target_labels, rows=rows)
            else:
                noisy = attack.addPairNoise(image_pairs, target_labels)
            for side, images in zip(sides, noisy):
                side.append(self.resize(images, target_size))
        return [sides[0], sides[1]]

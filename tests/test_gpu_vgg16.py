"""GPU: the VGGFace VGG-16 feature model (csrc/vgg16.hip, siamese.FaceVGG16) against the torch-CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_vgg16_features_match_oracle(gpu, tmp_path):
    from a_link_amd import siamese, vgg16 as V
    from oracle import vgg16 as O
    params = V.synthetic_params(2)
    m = siamese.FaceVGG16((224, 224), weights=params, max_batch=2)
    x = np.random.default_rng(0).integers(0, 256, (3, 224, 224, 3)).astype(np.float32)
    got = m.process(x)
    want = O.process(params, x)
    assert got.shape == (3, 25088) and got.dtype == np.float32
    a, b = got.astype(np.float64), want.astype(np.float64)
    cos = 1 - (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    assert cos.max() < 1e-3, cos
    assert (np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)).max() < 3e-2
    # the reference's call: predict(preprocess(X)) equals the fused raw-pixel path
    pre = m.preprocess(x)
    assert np.array_equal(pre, O.preprocess_input_v1(x))
    assert np.array_equal(m.model.predict(pre), got)
    # Keras weight-file round trip
    path = str(tmp_path / "rcmalli_vggface_tf_notop_vgg16.h5")
    V.save_keras_h5(path, params)
    m2 = siamese.FaceVGG16((224, 224), weights=path)
    assert np.array_equal(m2.process(x[:1]), got[:1])
    assert m.process(np.zeros((0, 224, 224, 3), np.float32)).shape == (0, 25088)

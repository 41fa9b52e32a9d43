"""oracle/ — CPU restatement of the reference's hot-path arithmetic.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product (a-link_amd/) never does — it fails loudly when the HIP library is missing.

Parity status:
  * pinned by golden vectors recorded from the reference's own code (tests/golden/make_golden*.py):
    uncertainty measures and samplers, Bagging.predict, createMiniBatch, splitDisguiseData (al_logic.py);
    Gaussian / Speckle / Poisson / Perlin noise, perturb_image (noise.py); the population-batched
    differential evolution and PixelAttacker (de.py); ROC_precompute and getStats (evaluation.py).
  * PARITY UNPINNED by the reference for everything that lives in un-vendored third-party code:
    the insightface LResNet-E-IR symbol executed by MXNet (reference code/face_model.py:34-40,90)
    (ir_resnet.py); the Keras 2.1.2 / TF 1.15 dense graph, binary_crossentropy, Adadelta and fit() loop
    (reference code/siamese.py:24-35,57,103,107) (siamese_head.py, smallres.py); keras-vggface 0.5's
    RESNET50 and preprocess_input (reference code/siamese.py:203-216) (vgg_resnet50.py); OpenCV's
    bilinear resize and the NumPy-version-dependent SaltPepper indexing (noise.py).  The reference holds
    no test, fixture or golden vector for those, the frameworks and the pretrained
    checkpoints cannot be obtained here (no network), so these modules restate the published
    algorithms and anchor on the reference's call sites.
"""

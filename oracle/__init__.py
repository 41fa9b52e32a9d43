"""oracle/ — CPU restatement of the reference's hot-path arithmetic.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product (a-link_amd/) never does — it fails loudly when the HIP library is missing.

Parity status (see DESIGN.md "Oracle"):
  * pinned by golden vectors generated from the reference's own pure-NumPy functions
    (tests/golden/make_golden.py): uncertainty measures, Bagging.predict, createMiniBatch,
    splitDisguiseData, perturb_image.
  * PARITY UNPINNED by the reference for everything that lives in un-vendored third-party code:
    the insightface LResNet-E-IR symbol executed by MXNet (reference code/face_model.py:34-40,90),
    and the Keras 2.1.2 / TF 1.15 dense graph, binary_crossentropy, Adadelta and fit() loop
    (reference code/siamese.py:24-35,57,103,107).  The reference holds no test, fixture or golden
    vector for those, the frameworks and the pretrained checkpoint cannot be
    obtained here (no network), so these modules restate the published algorithms and anchor on the
    reference's call sites.
"""

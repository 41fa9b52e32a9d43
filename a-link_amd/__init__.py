"""a-link_amd — MI355X-native (gfx950) implementation of the A-LINK face-recognition hot path.

Import as `a_link_amd` (the shim at the repository root maps the hyphenated directory).
Module names mirror the reference's (code/face_model.py, siamese.py, committee.py, uncertainty.py,
learners.py, base.py, helpers.py) so the active-learning drivers switch by changing imports only.
All compute goes through libalink_hip.so (include/alink_hip.h); there is no CPU fallback.
"""
__version__ = "0.1.0"

#!/usr/bin/env python3
"""Embedding parity at the headline depth: IR-ResNet-100 / -50 on the GPU (bf16 and f16 storage, f32
accumulation) against the unfused float32 CPU oracle on the same synthetic weights and images."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import a_link_amd  # noqa
from a_link_amd import weights as W
from a_link_amd.backbone import IRBackbone
from oracle import ir_resnet

for arch in ("r50", "r100"):
    params = W.synthetic_ir_params(W.ARCH_UNITS[arch], seed=1)
    x = np.random.default_rng(0).integers(0, 256, (8, 112, 112, 3)).astype(np.float32)
    ref = ir_resnet.embed(params, x, batch=8).astype(np.float64)
    for dtype in ("bf16",):    # f16 overflows on these synthetic weights (IRBackbone raises): see tests/test_gpu_backbone.py
        got = IRBackbone(params, dtype=dtype, max_batch=8).embed(x).astype(np.float64)
        cos = 1.0 - (got * ref).sum(1)
        print("%s %s: 1 - cos  max %.3e  mean %.3e ; |norm - 1| max %.1e" % (arch, dtype, cos.max(), cos.mean(),
                                                                          np.abs(np.linalg.norm(got, axis=1) - 1).max()))

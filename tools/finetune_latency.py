#!/usr/bin/env python3
"""Latency of one 16-pair fine-tune step (code/siamese.py:104 train_on_batch): plain launches vs hipGraph
replay, through DenseHead.train_on_batch and through the raw C-ABI call."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import a_link_amd
from a_link_amd.head import DenseHead
from a_link_amd import _abi
import ctypes as C
rng = np.random.RandomState(0)
L = torch.from_numpy(rng.randn(16, 512).astype(np.float32)).cuda(); R = torch.from_numpy(rng.randn(16, 512).astype(np.float32)).cuda()
y = np.zeros((16, 2), np.float32); y[np.arange(16), rng.randint(0, 2, 16)] = 1; yd = torch.from_numpy(y).cuda()
def bench(fn, reps=300):
    for _ in range(30): fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts))
for graph in (1, 0):
    hd = DenseHead(512, lr=0.1, seed=0)
    hd.lib.alink_head_set_graph(hd.h, graph)
    print("graph=%d train_on_batch (incl. metrics read-back): %.4f ms" % (graph, bench(lambda: hd.train_on_batch([L, R], yd))))
    m = hd._metrics
    s = torch.cuda.Stream()
    def raw(stream_ptr):
        _abi.check(hd.lib.alink_head_train_step(hd.h, _abi.ptr(L), _abi.ptr(R), _abi.ptr(yd), None, 16, 0.0, 1, _abi.ptr(m), stream_ptr))
    def on_side():
        raw(C.c_void_p(s.cuda_stream)); s.synchronize()
    def on_default():
        raw(C.c_void_p(0)); torch.cuda.synchronize()
    print("graph=%d raw call on side stream + stream sync (no readback): %.4f ms" % (graph, bench(on_side)))
    print("graph=%d raw call on default stream + sync (no readback): %.4f ms" % (graph, bench(on_default)))

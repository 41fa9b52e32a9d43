#!/usr/bin/env python3
"""Generates tests/golden/*.npz by IMPORTING the reference's own pure-NumPy functions from
/root/reference/code (this container only — the reference never travels to the GPU box) and
recording their outputs on seeded inputs.  Only data is stored: inputs and expected outputs.

Third-party modules the reference imports at module scope but which are absent here (cv2, modAL,
keras, tensorflow, ...) are replaced by EMPTY placeholder modules so that `import uncertainty`,
`import committee`, ... succeed.  The placeholders contain no arithmetic; the only function body
they provide is modAL.utils.selection.multi_argmax, which is third-party (modAL, version unpinned in
reference requirements.txt) and restated from modAL 0.3.x: argpartition(-values, n-1)[:n].
Golden vectors that depend on it are marked `order_free` (compared as sets).

Run:  python tests/golden/make_golden.py     (writes next to this file)
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference/code"
HERE = os.path.dirname(os.path.abspath(__file__))


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_placeholders():
    _placeholder("cv2")
    modal = _placeholder("modAL")
    utils = _placeholder("modAL.utils")
    data = _placeholder("modAL.utils.data", modALinput=object, data_vstack=None)
    val = _placeholder("modAL.utils.validation", check_class_labels=None, check_class_proba=None)

    def multi_argmax(values, n_instances=1):
        assert n_instances <= values.shape[0]
        return np.argpartition(-values, n_instances - 1, axis=0)[:n_instances]

    def shuffled_argmax(values, n_instances=1):
        raise NotImplementedError
    sel = _placeholder("modAL.utils.selection", multi_argmax=multi_argmax, shuffled_argmax=shuffled_argmax)
    modal.utils = utils
    utils.data, utils.validation, utils.selection = data, val, sel
    _placeholder("noise")


def main():
    install_placeholders()
    sys.path.insert(0, REF)
    import uncertainty as ref_unc            # reference code/uncertainty.py
    import committee as ref_committee        # reference code/committee.py
    import readDFW3 as ref_read              # reference code/readDFW3.py (py3 copy of readDFW.py)

    rng = np.random.RandomState(20261003)

    # ---- uncertainty measures (code/uncertainty.py:15-60) -------------------------------------
    out = {}
    logits = rng.randn(257, 2).astype(np.float32) * 3
    p2 = np.exp(logits) / np.exp(logits).sum(1, keepdims=True)
    p2 = p2.astype(np.float32)
    p2[:3] = np.array([[.9, .1], [.5, .5], [.3, .7]], np.float32)
    p2[3] = [1.0, 0.0]
    logits5 = rng.randn(64, 5)
    p5 = (np.exp(logits5) / np.exp(logits5).sum(1, keepdims=True)).astype(np.float32)
    for tag, p in (("p2", p2), ("p5", p5)):
        out[tag] = p
        out[tag + "_uncertainty"] = ref_unc._proba_uncertainty(p)
        out[tag + "_margin"] = ref_unc._proba_margin(p)
        out[tag + "_entropy"] = ref_unc._proba_entropy(p)

    class Clf(object):
        def __init__(self, p):
            self.p = p

        def predict_proba(self, X, **kw):
            return self.p
    X = [np.arange(257)[:, None].astype(np.float32), np.arange(257)[:, None].astype(np.float32) + 1000]
    for name in ("uncertainty_sampling", "margin_sampling", "entropy_sampling"):
        idx, inst = getattr(ref_unc, name)(Clf(p2), X, n_instances=20)
        out[name + "_idx20"] = np.asarray(idx)
        out[name + "_inst0"] = np.asarray(inst[0])
        out[name + "_inst1"] = np.asarray(inst[1])   # the reference returns X[0] twice (code/uncertainty.py:159)
    np.savez(os.path.join(HERE, "uncertainty.npz"), **out)

    # ---- Bagging.predict (code/committee.py:13-20) ----------------------------------------------
    class Member(object):
        def __init__(self, p):
            self.p = p

        def predict(self, X):
            return self.p
    mem = []
    for m in range(3):
        lg = rng.randn(100, 2).astype(np.float32)
        mem.append((np.exp(lg) / np.exp(lg).sum(1, keepdims=True)).astype(np.float32))
    bag = ref_committee.Bagging([Member(p) for p in mem], [])
    np.savez(os.path.join(HERE, "bagging.npz"), members=np.stack(mem), mean3=bag.predict(None),
             mean2=ref_committee.Bagging([Member(p) for p in mem[:2]], []).predict(None),
             mean1=ref_committee.Bagging([Member(mem[0])], []).predict(None))

    # ---- createMiniBatch / splitDisguiseData (code/readDFW3.py = readDFW.py:212-244) ------------
    n_plain = [2, 1, 3, 2]
    n_dig = [3, 2, 1, 4]
    code = 0
    plain, dig = [], []
    for k in n_plain:
        plain.append(np.arange(code, code + k, dtype=np.float32).reshape(k, 1, 1, 1)); code += k
    for k in n_dig:
        dig.append(np.arange(code, code + k, dtype=np.float32).reshape(k, 1, 1, 1)); code += k
    (xl, xr), y = ref_read.createMiniBatch(plain, dig)
    pre, post = ref_read.splitDisguiseData(dig, pre_ratio=0.5)
    np.savez(os.path.join(HERE, "minibatch.npz"), n_plain=n_plain, n_dig=n_dig, left_ids=xl.ravel().astype(np.int64),
             right_ids=xr.ravel().astype(np.int64), y=y, split_pre_len=[len(a) for a in pre],
             split_post_first=[a.ravel()[0] for a in post])
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()

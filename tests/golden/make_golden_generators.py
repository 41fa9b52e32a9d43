#!/usr/bin/env python3
"""Generates tests/golden/generators.npz by IMPORTING the reference's in-memory pair generators
(/root/reference/code/readDFW3.py — the Python-3 copy of readDFW.py — and readMTP.py; this container
only) and recording the batches they yield on seeded inputs.  Only data is stored.
cv2 is an empty placeholder module (the generators recorded here never call it).

Run:  python tests/golden/make_golden_generators.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference/code"
HERE = os.path.dirname(os.path.abspath(__file__))


def people(n, seed, lo=1, hi=3, d=2):
    """per-person feature arrays; every value unique so that batches can be compared as data"""
    rng = np.random.RandomState(seed)
    out, code = [], seed * 1000.0
    for _ in range(n):
        k = rng.randint(lo, hi + 1)
        out.append((code + np.arange(k * d, dtype=np.float32).reshape(k, d)))
        code += k * d
    return out


def main():
    sys.modules["cv2"] = types.ModuleType("cv2")
    sys.path.insert(0, REF)
    import readDFW3 as R                           # reference code/readDFW3.py
    plain, imp = people(5, 1), people(4, 2)
    out = {}
    g = R.getNormalGenerator(plain, 4, infinite=False)
    b = [next(g) for _ in range(6)]
    out["normal_left"] = np.stack([x[0][0] for x in b]); out["normal_right"] = np.stack([x[0][1] for x in b])
    out["normal_y"] = np.stack([x[1] for x in b])
    g = R.getImposterGenerator(plain, imp, 5, infinite=True)
    b = [next(g) for _ in range(4)]
    out["imp_left"] = np.stack([x[0][0] for x in b]); out["imp_right"] = np.stack([x[0][1] for x in b])
    out["imp_y"] = np.stack([x[1] for x in b])
    np.random.seed(123)
    g = R.getGenerator(R.getNormalGenerator(plain, 8), R.getNormalGenerator(imp, 8), R.getImposterGenerator(plain, imp, 8), 10)
    for i in range(3):
        (xl, xr), y = next(g)
        out["mix%d_left" % i], out["mix%d_right" % i], out["mix%d_y" % i] = xl, xr, y
    np.savez(os.path.join(HERE, "generators.npz"), **out)
    print("wrote generators.npz")


if __name__ == "__main__":
    main()

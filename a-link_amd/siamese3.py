"""siamese3 — drop-in for reference code/siamese3.py, the pair scorer of the baseline active-learning
scripts (imported as `siamese` by code/existing_al.py).  It is siamese.SiameseNetwork with three differences,
and is written as exactly that:

    * the dense graph ends in Dense(1, sigmoid) instead of Dense(2) + softmax   (code/siamese3.py:25)
    * labels stay (n, 1) — no to_categorical                                    (code/siamese3.py:39, 74-80)
    * customTrainModel passes no class weights                                  (code/siamese3.py:64-86)

finetune / customTrainModel / save / maybeLoadFromMemory / predict are the parent's, fed through the two hooks.
RESNET50 / SmallRes feature models are siamese.py's.
"""
import numpy as np

from . import siamese as _pairs
from .head import DenseHead
from .siamese import RESNET50  # noqa: F401  (code/siamese3.py:159-172)


class SiameseNetwork(_pairs.SiameseNetwork):
    @staticmethod
    def _targets(y):
        return y

    @staticmethod
    def _step_class_weight(y):
        return None

    def __init__(self, shape, modelName, learningRate=1.0, seed=None, adadelta_epsilon=1e-8):
        self.learningRate, self.modelName, self.shape = learningRate, modelName, shape
        self.siamese_net = DenseHead(shape[0], 512, 64, lr=learningRate, rho=0.95, eps=adadelta_epsilon, seed=seed,
                                     out_dim=1)

    def getDenseBarebones(self):
        return [(512, 'relu'), (64, 'relu'), (1, 'sigmoid')]

    def customTrainModel(self, dataGen, epochs, batch_size, valRatio=0.2, n_steps=320000, verbose=1):
        return _pairs.SiameseNetwork.customTrainModel(self, dataGen, epochs, batch_size, valRatio=valRatio, n_steps=n_steps,
                                                      preprocess=False, verbose=verbose)

    def testAccuracy(self, X, Y, batch_size=512):
        """code/siamese3.py:42-62 takes np.argmax(..., axis=1) of the (n, 1) sigmoid output — always 0 — so what it
        reports is the share of different-identity pairs; kept, and computed as that."""
        Y = np.asarray(Y).ravel()
        same = Y[:, None] == Y[None, :]
        return float(np.sum(~same)) / float(same.size)

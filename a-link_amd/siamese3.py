"""siamese3 — drop-in for reference code/siamese3.py, the pair scorer of the baseline active-learning
scripts (imported as `siamese` by code/existing_al.py): same dense graph as siamese.SiameseNetwork but
ending in Dense(1, sigmoid) (code/siamese3.py:25), labels (n, 1), no class weights in
customTrainModel (code/siamese3.py:64-86).  RESNET50 / SmallRes feature models are siamese.py's.
"""
import sys

import numpy as np

from .head import DenseHead, EarlyStopping, ReduceLROnPlateau
from .siamese import RESNET50  # noqa: F401  (code/siamese3.py:159-172)


class SiameseNetwork:
    _identity_preprocess = True

    def __init__(self, shape, modelName, learningRate=1.0, seed=None, adadelta_epsilon=1e-8):
        self.learningRate = learningRate
        self.modelName = modelName
        self.siamese_net = DenseHead(shape[0], 512, 64, lr=learningRate, rho=0.95, eps=adadelta_epsilon, seed=seed,
                                     out_dim=1)

    def finetune(self, X, Y, epochs, batch_size, verbose=1):
        early_stop = EarlyStopping(monitor='val_loss', min_delta=0.1, patience=5, verbose=1)
        reduce_lr = ReduceLROnPlateau(monitor='val_loss', factor=0.2, patience=5, min_lr=0.01, verbose=verbose)
        return self.siamese_net.fit(self.preprocess(X), Y, batch_size=batch_size, epochs=epochs, validation_split=0.2,
                                    verbose=verbose, callbacks=[early_stop, reduce_lr])

    def testAccuracy(self, X, Y, batch_size=512):
        """code/siamese3.py:42-62.  The reference takes np.argmax(..., axis=1) of the (n, 1) sigmoid
        output — always 0 — so it reports the share of different-identity pairs; kept."""
        X = np.asarray(X, dtype=np.float32)
        Y = np.asarray(Y).ravel()
        n = len(X)
        li = np.repeat(np.arange(n, dtype=np.int32), n)
        ri = np.tile(np.arange(n, dtype=np.int32), n)
        probs = self.siamese_net.predict_device(X, X, li, ri).cpu().numpy()
        pred = np.argmax(probs, axis=1)
        return np.sum(pred == 1 * (Y[li] == Y[ri])) / float(len(li))

    def customTrainModel(self, dataGen, epochs, batch_size, valRatio=0.2, n_steps=320000, verbose=1):
        steps_per_epoch = int(n_steps / batch_size)
        logs = []
        for _ in range(epochs):
            train_loss, val_loss = 0, 0
            train_acc, val_acc = 0, 0
            for i in range(steps_per_epoch):
                x, y = next(dataGen)
                indices = np.random.permutation(len(y))
                splitPoint = int(len(y) * valRatio)
                x_train, y_train = [pp[indices[splitPoint:]] for pp in x], y[indices[splitPoint:]]
                x_test, y_test = [pp[indices[:splitPoint]] for pp in x], y[indices[:splitPoint]]
                train_metrics = self.siamese_net.train_on_batch(x_train, y_train)
                train_loss += train_metrics[0]
                train_acc += train_metrics[1]
                if len(y_test) > 0:
                    val_metrics = self.siamese_net.test_on_batch(x_test, y_test)
                    val_loss += val_metrics[0]
                    val_acc += val_metrics[1]
                if verbose:
                    sys.stdout.write("%d / %d : Tr loss: %f, Tr acc: %f, Vl loss: %f, Vl acc: %f  \r" % (
                        i + 1, steps_per_epoch, train_loss / (i + 1), train_acc / (i + 1), val_loss / (i + 1),
                        val_acc / (i + 1)))
                    sys.stdout.flush()
            if verbose:
                print("\n")
            logs.append((train_loss / steps_per_epoch, train_acc / steps_per_epoch, val_loss / steps_per_epoch,
                         val_acc / steps_per_epoch))
        return logs

    def maybeLoadFromMemory(self):
        try:
            self.siamese_net.load_weights(self.modelName + ".h5")
            return True
        except Exception:
            return False

    def save(self, customName=None):
        if not customName:
            self.siamese_net.save_weights(self.modelName + ".h5")
        else:
            self.siamese_net.save_weights(customName + ".h5")

    def preprocess(self, X):
        return X

    def predict(self, X):
        return self.siamese_net.predict(self.preprocess(X), batch_size=1024)

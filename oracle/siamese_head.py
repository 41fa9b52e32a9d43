"""oracle/siamese_head.py — NumPy restatement of the siamese pair scorer and its Keras training loop.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED by the reference: the graph is built
from Keras layers at reference code/siamese.py:24-35 and driven by Keras `predict` / `fit` /
`train_on_batch` / `test_on_batch` (code/siamese.py:57,103,107,131); Keras==2.1.2 and
tensorflow-gpu==1.15.4 (reference requirements.txt:18,46) are not vendored and cannot be installed.
Restated from the published Keras 2.1.2 sources:
  keras/losses.py + backend/tensorflow_backend.py  binary_crossentropy: clip(p, 1e-7, 1-1e-7),
      logit = log(p/(1-p)), tf.nn.sigmoid_cross_entropy_with_logits, mean over the last axis
  keras/engine/training.py  _weighted_masked_objective: mean(l*w) / mean(w != 0);
      metrics=['accuracy'] + binary_crossentropy -> binary_accuracy = mean(round(p) == y) (unweighted)
      fit(): validation_split takes the LAST fraction before shuffling; np.random.shuffle(index_array)
      each epoch; epoch logs are batch-size-weighted means
  keras/optimizers.py  Adadelta(lr, rho=0.95, epsilon=1e-8, decay=0)
All arithmetic in float32 (Keras floatx), float64 optionally for a tighter reference.
"""
import numpy as np


def bf16_round(x):
    """float32 -> nearest bfloat16 (ties to even) -> float32: the value a 2-byte store of x holds."""
    x = np.ascontiguousarray(x, np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    out = (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32).reshape(x.shape)
    return np.where(np.isfinite(x), out, x).astype(np.float32)


def glorot_uniform(rng, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, (fan_in, fan_out)).astype(np.float32)


def init_weights(d_in, h1=512, h2=64, seed=0, out_dim=2):
    rng = np.random.RandomState(seed)
    return [glorot_uniform(rng, d_in, h1), np.zeros(h1, np.float32), glorot_uniform(rng, h1, h2),
            np.zeros(h2, np.float32), glorot_uniform(rng, h2, out_dim), np.zeros(out_dim, np.float32)]


def softmax(z):
    e = np.exp(z - z.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def forward(ws, L, R, dtype=np.float32, cache=False, quant=None):
    """code/siamese.py:27-32: abs(L-R) -> Dense relu -> Dense relu -> Dense -> softmax.
    quant="bf16" restates the build's mixed-precision mode (NOT the reference, which is float32): every GEMM
    operand — weights, |l - r|, activations — rounded to bfloat16, products and sums in float32, biases untouched."""
    q = bf16_round if quant == "bf16" else (lambda v: v)
    W1, b1, W2, b2, W3, b3 = [np.asarray(w, dtype) for w in ws]
    if quant:
        W1, W2, W3 = q(W1), q(W2), q(W3)
    d = q(np.abs(np.asarray(L, dtype) - np.asarray(R, dtype)))
    z1 = q(d @ W1 + b1)                 # stored rounded: relu(q(z)) == q(relu(z))
    a1 = np.maximum(z1, 0)
    z2 = q(a1 @ W2 + b2)
    a2 = np.maximum(z2, 0)
    z3 = a2 @ W3 + b3
    # Dense(2) + softmax (code/siamese.py:31-32) or Dense(1, sigmoid) (code/siamese3.py:25)
    p = softmax(z3) if z3.shape[1] > 1 else (1 / (1 + np.exp(-z3))).astype(z3.dtype)
    if cache:
        return p, (d, z1, a1, z2, a2)
    return p


def to_categorical(y, num_classes=2):
    y = np.asarray(y, dtype=int).ravel()
    out = np.zeros((len(y), num_classes), np.float32)
    out[np.arange(len(y)), y] = 1
    return out


def bce_per_sample(y, p):
    dt = p.dtype
    eps = dt.type(1e-7)
    pc = np.clip(p, eps, 1 - eps)
    x = np.log(pc / (1 - pc))
    l = np.maximum(x, 0) - x * y + np.log1p(np.exp(-np.abs(x)))
    return l.mean(axis=1)


def loss_and_metrics(y, p, sw=None):
    n = len(y)
    w = np.ones(n, p.dtype) if sw is None else np.asarray(sw, p.dtype)
    l = bce_per_sample(y.astype(p.dtype), p)
    loss = (l * w).mean() / (w != 0).astype(p.dtype).mean()
    acc = (np.rint(p) == y).astype(p.dtype).mean(axis=1).mean()
    return loss, acc


def gradients(ws, L, R, y, sw=None, dtype=np.float32, quant=None):
    q = bf16_round if quant == "bf16" else (lambda v: v)
    W1, b1, W2, b2, W3, b3 = [np.asarray(w, dtype) for w in ws]
    if quant:
        W1, W2, W3 = q(W1), q(W2), q(W3)
    y = np.asarray(y, dtype)
    p, (d, z1, a1, z2, a2) = forward(ws, L, R, dtype, cache=True, quant=quant)
    n = len(y)
    w = np.ones(n, dtype) if sw is None else np.asarray(sw, dtype)
    denom = (w != 0).sum()
    eps = dtype(1e-7)
    pc = np.clip(p, eps, 1 - eps)
    inside = (p >= eps) & (p <= 1 - eps)
    od = p.shape[1]
    dp = np.where(inside, (pc - y) / (pc * (1 - pc)) / dtype(od), 0) * (w / denom)[:, None]
    dz3 = p * (dp - (dp * p).sum(axis=1, keepdims=True)) if od > 1 else dp * p * (1 - p)
    dz3 = q(dz3.astype(dtype))          # activation gradients are GEMM operands too: rounded where produced
    gW3, gb3 = a2.T @ dz3, dz3.sum(axis=0)
    dz2 = q(((dz3 @ W3.T) * (z2 > 0)).astype(dtype))
    gW2, gb2 = a1.T @ dz2, dz2.sum(axis=0)
    dz1 = q(((dz2 @ W2.T) * (z1 > 0)).astype(dtype))
    gW1, gb1 = d.T @ dz1, dz1.sum(axis=0)
    loss, acc = loss_and_metrics(y, p, sw)
    return [gW1, gb1, gW2, gb2, gW3, gb3], loss, acc


class Adadelta(object):
    def __init__(self, shapes, lr=1.0, rho=0.95, epsilon=1e-8, dtype=np.float32):
        self.lr, self.rho, self.eps, self.dtype = dtype(lr), dtype(rho), dtype(epsilon), dtype
        self.a = [np.zeros(s, dtype) for s in shapes]
        self.d = [np.zeros(s, dtype) for s in shapes]

    def step(self, ws, gs):
        out = []
        one = self.dtype(1)
        for i, (p, g) in enumerate(zip(ws, gs)):
            g = g.astype(self.dtype)
            na = self.rho * self.a[i] + (one - self.rho) * g * g
            u = g * np.sqrt(self.d[i] + self.eps) / np.sqrt(na + self.eps)
            out.append((p - self.lr * u).astype(self.dtype))
            self.d[i] = self.rho * self.d[i] + (one - self.rho) * u * u
            self.a[i] = na
        return out


class HeadModel(object):
    """The slice of keras.models.Model used on `siamese_net`."""

    def __init__(self, d_in, h1=512, h2=64, lr=1.0, rho=0.95, epsilon=1e-8, seed=0, dtype=np.float32, out_dim=2,
                 quant=None):
        self.dtype = dtype
        self.quant = quant              # "bf16": the build's mixed-precision mode (master weights / Adadelta stay float32)
        self.ws = [w.astype(dtype) for w in init_weights(d_in, h1, h2, seed, out_dim)]
        self.opt = Adadelta([w.shape for w in self.ws], lr, rho, epsilon, dtype)
        self.stop_training = False

    def get_weights(self):
        return [w.copy() for w in self.ws]

    def set_weights(self, ws):
        self.ws = [np.asarray(w, self.dtype).copy() for w in ws]

    def get_lr(self):
        return float(self.opt.lr)

    def set_lr(self, lr):
        self.opt.lr = self.dtype(lr)

    def predict(self, X, batch_size=1024):
        return forward(self.ws, X[0], X[1], self.dtype, quant=self.quant)

    def train_on_batch(self, x, y, class_weight=None, sample_weight=None):
        sw = sample_weight
        if sw is None and class_weight is not None:
            sw = np.asarray([class_weight[c] for c in np.asarray(y).argmax(axis=1)], self.dtype)
        gs, loss, acc = gradients(self.ws, x[0], x[1], y, sw, self.dtype, quant=self.quant)
        self.ws = self.opt.step(self.ws, gs)
        return [float(loss), float(acc)]

    def test_on_batch(self, x, y):
        p = forward(self.ws, x[0], x[1], self.dtype, quant=self.quant)
        loss, acc = loss_and_metrics(np.asarray(y, self.dtype), p)
        return [float(loss), float(acc)]

    def fit(self, x, y, batch_size=32, epochs=1, validation_split=0.0, shuffle=True, callbacks=None):
        L, R, y = np.asarray(x[0]), np.asarray(x[1]), np.asarray(y)
        n_all = len(y)
        vL = None
        if 0.0 < validation_split < 1.0:
            split_at = int(n_all * (1.0 - validation_split))
            vL, vR, vy = L[split_at:], R[split_at:], y[split_at:]
            L, R, y = L[:split_at], R[:split_at], y[:split_at]
        n = len(y)
        index_array = np.arange(n)
        hist = {}
        for epoch in range(epochs):
            if shuffle:
                np.random.shuffle(index_array)
            tot, seen = np.zeros(2), 0
            for s in range(0, n, batch_size):
                ids = index_array[s:s + batch_size]
                out = self.train_on_batch([L[ids], R[ids]], y[ids])
                tot += np.asarray(out) * len(ids)
                seen += len(ids)
            logs = {"loss": tot[0] / seen, "acc": tot[1] / seen}
            if vL is not None and len(vy) > 0:
                vt, vs = np.zeros(2), 0
                for s in range(0, len(vy), batch_size):
                    k = len(vy[s:s + batch_size])
                    out = self.test_on_batch([vL[s:s + batch_size], vR[s:s + batch_size]], vy[s:s + batch_size])
                    vt += np.asarray(out) * k
                    vs += k
                logs["val_loss"], logs["val_acc"] = vt[0] / vs, vt[1] / vs
            for cb in (callbacks or []):
                cb.on_epoch_end(epoch, logs, self)
            for k, v in logs.items():
                hist.setdefault(k, []).append(v)
            if self.stop_training:
                break
        return hist


def finetune(model, X, Y, epochs, batch_size):
    """SiameseNetwork.finetune (code/siamese.py:52-58); the two callbacks are inert below 6 epochs."""
    return model.fit(X, to_categorical(Y, 2), batch_size=batch_size, epochs=epochs, validation_split=0.2)


def bagging_predict(member_predictions):
    """Bagging.predict arithmetic (code/committee.py:18): np.sum(np.array(preds), axis=0) / len(models)."""
    preds = np.array(member_predictions)
    return np.array(np.sum(preds, axis=0) / len(member_predictions))


def custom_train_model(model, data_gen, epochs, batch_size, val_ratio=0.2, n_steps=320000):
    """SiameseNetwork.customTrainModel (reference code/siamese.py:81-112), python loops kept.
    The reference file has no `from __future__ import division`, so under Python 2
    `len(y_train) / np.sum(y_train == 1)` is an INTEGER division (int / numpy.int64 floors)."""
    steps_per_epoch = int(n_steps / batch_size)
    logs = []
    for eno in range(epochs):
        train_loss = val_loss = train_acc = val_acc = 0
        for i in range(steps_per_epoch):
            x, y = next(data_gen)
            indices = np.random.permutation(len(y))
            split_point = int(len(y) * val_ratio)
            x_train, y_train = [pp[indices[split_point:]] for pp in x], y[indices[split_point:]]
            x_test, y_test = [pp[indices[:split_point]] for pp in x], y[indices[:split_point]]
            with np.errstate(divide='ignore'):
                class_1_weight = np.int64(len(y_train)) // np.sum(y_train == 1)
                class_0_weight = np.int64(len(y_train)) // np.sum(y_train == 0)
            scaling_factor = float(class_1_weight + class_0_weight)
            class_weight = {0: class_0_weight / scaling_factor, 1: class_1_weight / scaling_factor}
            tm = model.train_on_batch(x_train, to_categorical(y_train, 2), class_weight=class_weight)
            train_loss += tm[0]
            train_acc += tm[1]
            if len(y_test) > 0:
                vm = model.test_on_batch(x_test, to_categorical(y_test, 2))
                val_loss += vm[0]
                val_acc += vm[1]
        logs.append((train_loss / steps_per_epoch, train_acc / steps_per_epoch, val_loss / steps_per_epoch,
                     val_acc / steps_per_epoch))
    return logs

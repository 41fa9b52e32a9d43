"""keras_wrapper — drop-in for reference code/keras_wrapper.py (itself a copy of
keras.wrappers.scikit_learn): the scikit-learn classifier facade the baseline script wraps around
`model.siamese_net` (code/existing_al.py:88-92).

Same public surface — BaseWrapper(build_fn, **sk_params) with check_params / get_params / set_params /
filter_sk_params / fit, KerasClassifier with fit / predict / predict_proba / score, to_list — written
for the "Keras models" of this package (DenseHead, SmallResNet: anything with fit / predict / evaluate,
a `loss` attribute and `metrics_names`).  Parameter routing: a keyword is legal if the model factory
takes it or if it is one of Keras' fit / predict / evaluate arguments below.
"""
import copy
import inspect

import numpy as np

from .head import to_categorical

FIT_KEYS = frozenset(("batch_size", "epochs", "verbose", "callbacks", "validation_split", "shuffle"))
PREDICT_KEYS = frozenset(("batch_size", "verbose"))
EVALUATE_KEYS = frozenset(("batch_size", "verbose"))


def to_list(x, allow_tuple=False):
    if isinstance(x, list) or (allow_tuple and isinstance(x, tuple)):
        return list(x)
    return [x]


def _accepts(fn, name):
    try:
        return name in inspect.signature(fn).parameters
    except (TypeError, ValueError):
        return False


def _loss_name(model):
    loss = getattr(model, "loss", None)
    return getattr(loss, "__name__", loss)


class BaseWrapper(object):
    def __init__(self, build_fn=None, **sk_params):
        self.build_fn, self.sk_params = build_fn, sk_params
        self.check_params(sk_params)
        self.build_self()

    # the callable whose signature decides which sk_params are model parameters
    def _factory(self):
        if self.build_fn is None:
            return self.__call__
        plain = inspect.isfunction(self.build_fn) or inspect.ismethod(self.build_fn)
        return self.build_fn if plain else self.build_fn.__call__

    def check_params(self, params):
        factory = self._factory()
        legal = FIT_KEYS | PREDICT_KEYS | EVALUATE_KEYS | {"nb_epoch"}
        for key in params:
            if key not in legal and not _accepts(factory, key):
                raise ValueError("{} is not a legal parameter".format(key))

    def get_params(self, **params):
        return dict(copy.deepcopy(self.sk_params), build_fn=self.build_fn)

    def set_params(self, **params):
        self.check_params(params)
        self.sk_params.update(params)
        return self

    def build_self(self):
        factory = self._factory()
        self.model = factory(**{k: v for k, v in self.sk_params.items() if _accepts(factory, k)})

    def filter_sk_params(self, keys, override=None):
        """sk_params restricted to `keys` (the reference filters by a function's signature; here by
        Keras' argument names), then overridden."""
        chosen = {k: v for k, v in self.sk_params.items() if k in keys}
        chosen.update(override or {})
        return chosen

    def fit(self, x, y, **kwargs):
        if _loss_name(self.model) == "categorical_crossentropy" and np.ndim(y) != 2:
            y = to_categorical(y)
        return self.model.fit(x, y, **self.filter_sk_params(FIT_KEYS, kwargs))


class KerasClassifier(BaseWrapper):
    def fit(self, x, y, sample_weight=None, **kwargs):
        y = np.array(y)
        one_hot = y.ndim == 2 and y.shape[1] > 1
        if not one_hot and not (y.ndim == 1 or (y.ndim == 2 and y.shape[1] == 1)):
            raise ValueError("Invalid shape for y: " + str(y.shape))
        if one_hot:
            self.classes_ = np.arange(y.shape[1])
        else:                                   # class labels -> indices into the sorted label set
            self.classes_ = np.unique(y)
            y = np.searchsorted(self.classes_, y)
        self.n_classes_ = len(self.classes_)
        if sample_weight is not None:
            kwargs["sample_weight"] = sample_weight
        return super(KerasClassifier, self).fit(x, y, **kwargs)

    def _probabilities(self, x, kwargs):
        return np.asarray(self.model.predict(x, **self.filter_sk_params(PREDICT_KEYS, kwargs)))

    def predict(self, x, **kwargs):
        proba = self._probabilities(x, kwargs)
        idx = proba.argmax(axis=-1) if proba.shape[-1] > 1 else (proba > 0.5).astype("int32")
        return self.classes_[idx]

    def predict_proba(self, x, **kwargs):
        probs = self._probabilities(x, kwargs)
        # a single sigmoid output becomes the two-column form scikit-learn expects
        return np.hstack([1 - probs, probs]) if probs.shape[1] == 1 else probs

    def score(self, x, y, **kwargs):
        y = np.searchsorted(self.classes_, y)
        if _loss_name(self.model) == "categorical_crossentropy" and np.ndim(y) != 2:
            y = to_categorical(y)
        outputs = to_list(self.model.evaluate(x, y, **self.filter_sk_params(EVALUATE_KEYS, kwargs)))
        named = dict(zip(self.model.metrics_names, outputs))
        if "acc" not in named:
            raise ValueError('The model is not configured to compute accuracy. '
                             'You should pass `metrics=["accuracy"]` to the `model.compile()` method.')
        return named["acc"]

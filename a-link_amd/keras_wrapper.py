"""keras_wrapper — drop-in for reference code/keras_wrapper.py (a copy of keras.wrappers.scikit_learn):
the scikit-learn classifier facade the baseline script wraps around `model.siamese_net`
(code/existing_al.py:88-92).  The wrapped "Keras model" here is a DenseHead / SmallResNet — anything
with fit / predict / evaluate and a `loss` attribute.
"""
import copy
import inspect
import types

import numpy as np

from .head import to_categorical

# parameters a caller may route through sk_params (the arguments of Sequential.fit / predict / evaluate)
_FIT_ARGS = ("batch_size", "epochs", "verbose", "callbacks", "validation_split", "shuffle")
_PREDICT_ARGS = ("batch_size", "verbose")


def to_list(x, allow_tuple=False):
    if isinstance(x, list):
        return x
    if allow_tuple and isinstance(x, tuple):
        return list(x)
    return [x]


def _has_arg(fn, name):
    try:
        return name in inspect.signature(fn).parameters
    except (TypeError, ValueError):
        return False


class BaseWrapper(object):
    def __init__(self, build_fn=None, **sk_params):
        self.build_fn = build_fn
        self.sk_params = sk_params
        self.check_params(sk_params)
        self.build_self()

    def _build_callable(self):
        if self.build_fn is None:
            return self.__call__
        if not isinstance(self.build_fn, (types.FunctionType, types.MethodType)):
            return self.build_fn.__call__
        return self.build_fn

    def check_params(self, params):
        fn = self._build_callable()
        for name in params:
            if name in _FIT_ARGS or name in _PREDICT_ARGS or _has_arg(fn, name) or name == 'nb_epoch':
                continue
            raise ValueError('{} is not a legal parameter'.format(name))

    def get_params(self, **params):
        res = copy.deepcopy(self.sk_params)
        res.update({'build_fn': self.build_fn})
        return res

    def set_params(self, **params):
        self.check_params(params)
        self.sk_params.update(params)
        return self

    def build_self(self):
        fn = self._build_callable()
        self.model = fn(**{k: v for k, v in self.sk_params.items() if _has_arg(fn, k)})

    def filter_sk_params(self, names, override=None):
        res = {k: v for k, v in self.sk_params.items() if k in names}
        res.update(override or {})
        return res

    def fit(self, x, y, **kwargs):
        loss_name = getattr(self.model, "loss", None)
        if hasattr(loss_name, '__name__'):
            loss_name = loss_name.__name__
        if loss_name == 'categorical_crossentropy' and len(y.shape) != 2:
            y = to_categorical(y)
        fit_args = copy.deepcopy(self.filter_sk_params(_FIT_ARGS))
        fit_args.update(kwargs)
        return self.model.fit(x, y, **fit_args)


class KerasClassifier(BaseWrapper):
    def fit(self, x, y, sample_weight=None, **kwargs):
        y = np.array(y)
        if len(y.shape) == 2 and y.shape[1] > 1:
            self.classes_ = np.arange(y.shape[1])
        elif (len(y.shape) == 2 and y.shape[1] == 1) or len(y.shape) == 1:
            self.classes_ = np.unique(y)
            y = np.searchsorted(self.classes_, y)
        else:
            raise ValueError('Invalid shape for y: ' + str(y.shape))
        self.n_classes_ = len(self.classes_)
        if sample_weight is not None:
            kwargs['sample_weight'] = sample_weight
        return super(KerasClassifier, self).fit(x, y, **kwargs)

    def predict(self, x, **kwargs):
        proba = np.asarray(self.model.predict(x, **self.filter_sk_params(_PREDICT_ARGS, kwargs)))
        if proba.shape[-1] > 1:
            classes = proba.argmax(axis=-1)
        else:
            classes = (proba > 0.5).astype('int32')
        return self.classes_[classes]

    def predict_proba(self, x, **kwargs):
        probs = np.asarray(self.model.predict(x, **self.filter_sk_params(_PREDICT_ARGS, kwargs)))
        if probs.shape[1] == 1:
            probs = np.hstack([1 - probs, probs])
        return probs

    def score(self, x, y, **kwargs):
        y = np.searchsorted(self.classes_, y)
        loss_name = getattr(self.model, "loss", None)
        if hasattr(loss_name, '__name__'):
            loss_name = loss_name.__name__
        if loss_name == 'categorical_crossentropy' and len(y.shape) != 2:
            y = to_categorical(y)
        outputs = to_list(self.model.evaluate(x, y, **self.filter_sk_params(("batch_size", "verbose"), kwargs)))
        for name, output in zip(self.model.metrics_names, outputs):
            if name == 'acc':
                return output
        raise ValueError('The model is not configured to compute accuracy. '
                         'You should pass `metrics=["accuracy"]` to the `model.compile()` method.')

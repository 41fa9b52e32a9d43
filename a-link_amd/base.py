"""base — API-surface mirror of reference code/base.py (vendored modAL BaseLearner / BaseCommittee,
patched for pair input: X is a list [left, right], checks look at X[0] — code/base.py:74,120,149).

Glue only: the estimator underneath does the arithmetic (a DenseHead on the GPU in this package).
"""
import abc

import numpy as np


def _check_X_y(X, y):
    """stand-in for sklearn.utils.check_X_y(X[0], y, allow_nd=True, multi_output=True): lengths agree."""
    n = len(X[0]) if isinstance(X, (list, tuple)) else len(X)
    if n != len(y):
        raise ValueError("Found input variables with inconsistent numbers of samples: [%d, %d]" % (n, len(y)))


def data_vstack(blocks):
    """modAL.utils.data.data_vstack for ndarray / list-of-ndarray (pair) inputs."""
    first = blocks[0]
    if isinstance(first, (list, tuple)):
        return [np.concatenate([b[i] for b in blocks], axis=0) for i in range(len(first))]
    return np.concatenate(blocks, axis=0)


def _take(X, idx):
    if isinstance(X, (list, tuple)):
        return [x[idx] for x in X]
    return X[idx]


def _n(X):
    return len(X[0]) if isinstance(X, (list, tuple)) else len(X)


class BaseLearner(abc.ABC):
    """code/base.py:23-213"""

    def __init__(self, estimator, query_strategy, X_training=None, y_training=None, bootstrap_init=False,
                 **fit_kwargs):
        assert callable(query_strategy), 'query_strategy must be callable'
        self.estimator = estimator
        self.query_strategy = query_strategy
        self.X_training = X_training
        self.y_training = y_training
        if X_training is not None:
            self._fit_to_known(bootstrap=bootstrap_init, **fit_kwargs)

    def _add_training_data(self, X, y):
        _check_X_y(X, y)
        if self.X_training is None:
            self.X_training, self.y_training = X, y
        else:
            try:
                self.X_training = data_vstack((self.X_training, X))
                self.y_training = data_vstack((self.y_training, y))
            except ValueError:
                raise ValueError('the dimensions of the new training data and label must'
                                 'agree with the training data and labels provided so far')

    def _fit_to_known(self, bootstrap=False, **fit_kwargs):
        if not bootstrap:
            self.estimator.fit(self.X_training, self.y_training, **fit_kwargs)
        else:
            n = _n(self.X_training)
            idx = np.random.choice(range(n), n, replace=True)
            self.estimator.fit(_take(self.X_training, idx), self.y_training[idx], **fit_kwargs)
        return self

    def _fit_on_new(self, X, y, bootstrap=False, **fit_kwargs):
        _check_X_y(X, y)
        if not bootstrap:
            self.estimator.fit(X, y, **fit_kwargs)
        else:
            n = _n(X)
            idx = np.random.choice(range(n), n, replace=True)
            self.estimator.fit(_take(X, idx), y[idx])
        return self

    def fit(self, X, y, bootstrap=False, **fit_kwargs):
        _check_X_y(X, y)
        self.X_training, self.y_training = X, y
        return self._fit_to_known(bootstrap=bootstrap, **fit_kwargs)

    def predict(self, X, **predict_kwargs):
        return self.estimator.predict(X, **predict_kwargs)

    def predict_proba(self, X, **predict_proba_kwargs):
        return self.estimator.predict_proba(X, **predict_proba_kwargs)

    def query(self, *query_args, **query_kwargs):
        return self.query_strategy(self, *query_args, **query_kwargs)

    def score(self, X, y, **score_kwargs):
        return self.estimator.score(X, y, **score_kwargs)

    @abc.abstractmethod
    def teach(self, *args, **kwargs):
        pass


class BaseCommittee(abc.ABC):
    """code/base.py:216-349"""

    def __init__(self, learner_list, query_strategy):
        assert type(learner_list) == list, 'learners must be supplied in a list'
        self.learner_list = learner_list
        self.query_strategy = query_strategy

    def __iter__(self):
        for learner in self.learner_list:
            yield learner

    def __len__(self):
        return len(self.learner_list)

    def _add_training_data(self, X, y):
        for learner in self.learner_list:
            learner._add_training_data(X, y)

    def _fit_to_known(self, bootstrap=False, **fit_kwargs):
        for learner in self.learner_list:
            learner._fit_to_known(bootstrap=bootstrap, **fit_kwargs)

    def _fit_on_new(self, X, y, bootstrap=False, **fit_kwargs):
        for learner in self.learner_list:
            learner._fit_on_new(X, y, bootstrap=bootstrap, **fit_kwargs)

    def fit(self, X, y, **fit_kwargs):
        for learner in self.learner_list:
            learner.fit(X, y, **fit_kwargs)
        return self

    def query(self, *query_args, **query_kwargs):
        return self.query_strategy(self, *query_args, **query_kwargs)

    def rebag(self, **fit_kwargs):
        self._fit_to_known(bootstrap=True, **fit_kwargs)

    def teach(self, X, y, bootstrap=False, only_new=False, **fit_kwargs):
        self._add_training_data(X, y)
        if not only_new:
            self._fit_to_known(bootstrap=bootstrap, **fit_kwargs)
        else:
            self._fit_on_new(X, y, bootstrap=bootstrap, **fit_kwargs)

    @abc.abstractmethod
    def predict(self, X):
        pass

    @abc.abstractmethod
    def vote(self, X):
        pass

"""learners — API-surface mirror of reference code/learners.py: ActiveLearner (code/learners.py:21-99)
and Committee (code/learners.py:239-416).  BayesianOptimizer / CommitteeRegressor are regression
tools no reference driver uses and are not provided.

KerasClassifier (keras_wrapper.py) is the scikit-learn facade the baseline driver wraps around
`model.siamese_net` (reference code/existing_al.py:91-101, code/keras_wrapper.py:187-308).
"""
import numpy as np

from .base import BaseCommittee, BaseLearner, _n
from .uncertainty import uncertainty_sampling


class ActiveLearner(BaseLearner):
    def __init__(self, estimator, query_strategy=uncertainty_sampling, X_training=None, y_training=None,
                 bootstrap_init=False, **fit_kwargs):
        super().__init__(estimator, query_strategy, X_training, y_training, bootstrap_init, **fit_kwargs)

    def teach(self, X, y, bootstrap=False, only_new=False, **fit_kwargs):
        self._add_training_data(X, y)
        if not only_new:
            self._fit_to_known(bootstrap=bootstrap, **fit_kwargs)
        else:
            self._fit_on_new(X, y, bootstrap=bootstrap, **fit_kwargs)


def vote_entropy_sampling(committee, X, n_instances=1, **kw):
    """modAL.disagreement.vote_entropy_sampling (default strategy of Committee, code/learners.py:287)."""
    from scipy.stats import entropy
    from .uncertainty import multi_argmax
    votes = committee.vote(X, **kw)
    p_vote = np.zeros((votes.shape[0], len(committee.classes_)))
    for i, row in enumerate(votes):
        for j, c in enumerate(committee.classes_):
            p_vote[i, j] = np.sum(row == c) / float(len(committee))
    ent = entropy(p_vote.T)
    idx = multi_argmax(ent, n_instances=n_instances)
    return idx, [X[0][idx], X[0][idx]]


class Committee(BaseCommittee):
    def __init__(self, learner_list, query_strategy=vote_entropy_sampling):
        super().__init__(learner_list, query_strategy)
        self._set_classes()

    def _set_classes(self):
        try:
            known = tuple(learner.estimator.classes_ for learner in self.learner_list)
        except AttributeError:
            self.classes_ = None
            self.n_classes_ = 0
            return
        self.classes_ = np.unique(np.concatenate(known, axis=0), axis=0)
        self.n_classes_ = len(self.classes_)

    def _add_training_data(self, X, y):
        super()._add_training_data(X, y)
        self._set_classes()

    def vote(self, X, **predict_kwargs):
        prediction = np.zeros(shape=(_n(X), len(self.learner_list)))
        for i, learner in enumerate(self.learner_list):
            prediction[:, i] = learner.predict(X, **predict_kwargs)
        return prediction

    def vote_proba(self, X, **predict_proba_kwargs):
        proba = np.zeros(shape=(_n(X), len(self.learner_list), self.n_classes_))
        for i, learner in enumerate(self.learner_list):
            proba[:, i, :] = learner.predict_proba(X, **predict_proba_kwargs)
        return proba

    def predict_proba(self, X, **predict_proba_kwargs):
        return np.mean(self.vote_proba(X, **predict_proba_kwargs), axis=1)

    def predict(self, X, **predict_proba_kwargs):
        proba = self.predict_proba(X, **predict_proba_kwargs)
        return self.classes_[np.argmax(proba, axis=1)]

    def score(self, X, y, sample_weight=None):
        y_pred = self.predict(X)
        w = np.ones(len(y)) if sample_weight is None else np.asarray(sample_weight)
        return float(np.sum(w * (np.asarray(y).ravel() == y_pred.ravel())) / np.sum(w))


from .keras_wrapper import KerasClassifier  # noqa: E402,F401  (code/existing_al.py:5 imports it from keras_wrapper)

"""existing_al — the baseline active-learning loop of reference code/existing_al.py:95-121 as a function:
modAL-style ActiveLearner over a KerasClassifier-wrapped pair scorer, query by uncertainty / margin /
entropy, teach on the queried pairs only (epochs=2, validation_split=0.1)."""
import numpy as np

from . import uncertainty
from .keras_wrapper import KerasClassifier
from .learners import ActiveLearner


def get_strategy_object(query_strategy):
    """code/existing_al.py:43-49"""
    if query_strategy == 'uncertainty_sampling':
        return uncertainty.uncertainty_sampling
    elif query_strategy == 'margin_sampling':
        return uncertainty.margin_sampling
    elif query_strategy == 'entropy_sampling':
        return uncertainty.entropy_sampling


def run_baseline(model, dataGen, query_strategy='uncertainty_sampling', active_ratio=1.0, out_model=None, verbose=1,
                 max_queries=None):
    """model: siamese3.SiameseNetwork; dataGen: pairs.getGenerator over FINITE source generators (the
    loop ends when it is exhausted; the Python-3 copy of the reference detects a `None` batch).
    Returns (learner, n_queries)."""
    def dummy_fn():
        return model.siamese_net

    wrapped_model = KerasClassifier(dummy_fn)
    learner = ActiveLearner(estimator=wrapped_model, query_strategy=get_strategy_object(query_strategy), verbose=verbose)
    n_queries = 0
    while max_queries is None or n_queries < max_queries:
        try:
            (X_old_left, X_old_right), Y_old = next(dataGen)
        except StopIteration:
            break
        if X_old_left is None:
            break
        query_idx, query_instance = learner.query([X_old_left, X_old_right],
                                                  n_instances=int(len(X_old_left) * active_ratio), verbose=0)
        learner.teach(X=[X_old_left[query_idx], X_old_right[query_idx]], y=Y_old[query_idx], only_new=True,
                      verbose=verbose, epochs=2, validation_split=0.1)
        n_queries += 1
    if out_model:
        model.siamese_net.save_weights(out_model + ".h5")
    return learner, n_queries

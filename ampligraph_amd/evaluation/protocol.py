"""Host-side helpers around the evaluation protocol with the reference's names and semantics
(/root/reference/ampligraph/evaluation/protocol.py:27-233): dataset splitting without unseen entities and
filtering of triples whose entities the model has not seen, and `select_best_model_ranking` (:447-933), the grid / random
search that calls fit() and evaluate().  Plain numpy and Python; nothing here touches the device directly."""
import numpy as np


def train_test_split_no_unseen(X, test_size=100, seed=0, allow_duplication=False, filtered_test_predicates=None):
    """Split X into (X_train, X_test) such that every entity and relation of the test set also occurs in the training
    set (:27-198): candidates are visited in a seeded random order and moved to the test set while the counts of their
    subject, relation and object in the remaining data stay positive."""
    X = np.asarray(X)
    rng_state = np.random.get_state()
    np.random.seed(seed)   # the reference seeds numpy's global stream (:110); restored below
    try:
        if filtered_test_predicates:
            cand_mask = np.isin(X[:, 1], filtered_test_predicates)
            cand, fixed_train = X[cand_mask], X[~cand_mask]
        else:
            cand, fixed_train = X, None
        if isinstance(test_size, float):
            test_size = int(len(cand) * test_size)
        ents, ent_cnt = np.unique(np.concatenate([cand[:, 0], cand[:, 2]]), return_counts=True)
        rels, rel_cnt = np.unique(cand[:, 1], return_counts=True)
        e_left = dict(zip(ents.tolist(), ent_cnt.tolist()))
        r_left = dict(zip(rels.tolist(), rel_cnt.tolist()))
        order = np.random.permutation(np.arange(cand.shape[0]))
        idx_test, idx_train = [], []
        for i, idx in enumerate(order):
            s, p, o = cand[idx][:3].tolist()   # rows may carry numeric edge columns behind s, p, o (FocusE): only the triple counts
            e_left[s] -= 1
            r_left[p] -= 1
            e_left[o] -= 1
            if e_left[s] > 0 and r_left[p] > 0 and e_left[o] > 0:
                idx_test.append(idx)
                if len(idx_test) == test_size:
                    idx_train.extend(order[i + 1:].tolist())
                    break
            else:   # taking it out would leave an entity / relation unseen: it stays in training
                e_left[s] += 1
                r_left[p] += 1
                e_left[o] += 1
                idx_train.append(idx)
        if len(idx_test) != test_size:
            if not allow_duplication:
                raise Exception("Cannot create a test split of the desired size. Some entities will not occur in both "
                                "training and test set. Set allow_duplication=True,remove filter on test predicates or "
                                "set test_size to a smaller value.")
            idx_test.extend(np.random.choice(idx_test, size=test_size - len(idx_test)).tolist())
        X_train = cand[idx_train] if fixed_train is None else np.concatenate([fixed_train, cand[idx_train]])
        X_test = cand[idx_test]
        return np.random.permutation(X_train), np.random.permutation(X_test)
    finally:
        np.random.set_state(rng_state)


def filter_unseen_entities(X, model, verbose=False):
    """Drop the triples of X whose subject or object the model was not trained on (:201-233)."""
    X = np.asarray(X)
    ix = model.data_indexer
    seen_s = np.isin(X[:, 0], ix._ent_raw)
    seen_o = np.isin(X[:, 2], ix._ent_raw)
    keep = seen_s & seen_o
    removed = int((~keep).sum())
    if removed > 0:
        if verbose:
            print("Removing {} triples containing unseen entities. ".format(removed))
        return X[keep]
    return X


# ----------------------------------------------------------------------------------------------------------------------
# Model selection: the caller of fit()/evaluate() that /root/reference/ampligraph/evaluation/protocol.py:447-933
# (select_best_model_ranking) is -- grid search (all combinations) or random search (max_combinations draws) over a
# parameter grid whose values are scalars, lists, callables (random search) or nested dicts (e.g. "loss_params").
# ----------------------------------------------------------------------------------------------------------------------
def _expand_grid(grid):
    """All combinations of a (possibly nested) grid, in the order of itertools.product over its keys."""
    import itertools

    keys, choices = [], []
    for key, val in grid.items():
        keys.append(key)
        if isinstance(val, dict):
            choices.append(list(_expand_grid(val)))
        elif isinstance(val, (list, tuple, np.ndarray)):
            choices.append(list(val))
        else:
            choices.append([val])
    seen = []
    for combo in itertools.product(*choices):
        params = dict(zip(keys, combo))
        if params not in seen:   # duplicates in the lists are tried once
            seen.append(params)
            yield params


def _sample_grid(grid):
    """One random draw: callables are called, lists sampled (np.random, seeded by the caller), dicts recursed."""
    out = {}
    for key, val in grid.items():
        if callable(val):
            out[key] = val()
        elif isinstance(val, dict):
            out[key] = _sample_grid(val)
        elif isinstance(val, (list, tuple, np.ndarray)):
            pick = val[int(np.random.randint(len(val)))]
            out[key] = int(pick) if isinstance(pick, (np.integer,)) else pick
        else:
            out[key] = val
    return out


def _count_grid(grid):
    n = 1
    for val in grid.values():
        if isinstance(val, dict):
            n *= _count_grid(val)
        elif isinstance(val, (list, tuple, np.ndarray)):
            n *= max(1, len(val))
    return n


def _metrics(ranks):
    from .metrics import hits_at_n_score, mr_score, mrr_score

    return {"mrr": mrr_score(ranks), "mr": mr_score(ranks), "hits_1": hits_at_n_score(ranks, n=1),
            "hits_3": hits_at_n_score(ranks, n=3), "hits_10": hits_at_n_score(ranks, n=10)}


def select_best_model_ranking(model_class, X_train, X_valid, X_test, param_grid, max_combinations=None,
                              param_grid_random_seed=0, use_filter=True, early_stopping=True, early_stopping_params=None,
                              use_test_for_selection=False, entities_subset=None, corrupt_side="s,o", focusE=False,
                              focusE_params={}, retrain_best_model=False, verbose=False, _model_factory=None):
    """Train one model per parameter combination, keep the one with the best filtered MRR on the selection set (the odd
    rows of X_valid -- the even rows drive early stopping -- or X_test with use_test_for_selection), optionally retrain it
    on train + validation for the number of epochs early stopping found, and evaluate it on X_test.

    model_class: scoring type name ("TransE", "DistMult", "ComplEx", "HolE", "RotatE").  Grid keys: batch_size, seed, epochs,
    k, eta, loss, loss_params, regularizer, regularizer_params, optimizer, optimizer_params, initializer, focusE_params.
    Returns (best_model, best_params, best_mrr_train, ranks_test, test_evaluation, experimental_history); a combination
    that raises is recorded in the history ({"exception": ...}) and skipped, as in the reference."""
    from ..callbacks import EarlyStopping
    from ..latent_features import loss_functions, optimizers, regularizers

    if _model_factory is None:
        from ..latent_features import ScoringBasedEmbeddingModel as _model_factory
    esp = dict(early_stopping_params or {})
    grid = dict(param_grid)
    total = _count_grid(grid)
    if max_combinations is not None:
        np.random.seed(param_grid_random_seed)

        def random_search():
            tried = []
            attempts = 0
            while len(tried) < min(total, max_combinations) and attempts < 100 * max_combinations:
                attempts += 1
                p = _sample_grid(grid)
                if p not in tried:
                    tried.append(p)
                    yield p

        combos = random_search()
    else:
        combos = _expand_grid(grid)
    if focusE:
        assert isinstance(X_train, np.ndarray) and X_train.shape[1] > 3, \
            "Weights are missing! Concatenate them to X_train in order to use FocusE!"
    X_filter = {"train": X_train, "valid": X_valid, "test": X_test} if use_filter else False
    if use_test_for_selection:
        selection = X_test
    else:
        selection, X_valid = X_valid[1::2], X_valid[::2]

    def build(params):
        model = _model_factory(eta=int(params.get("eta", 1)), k=int(params.get("k", 100)), scoring_type=model_class,
                               seed=int(params.get("seed", 0)))
        reg = params.get("regularizer")
        model.compile(loss=loss_functions.get(params.get("loss", "multiclass_nll"), params.get("loss_params", {})),
                      optimizer=optimizers.get(params.get("optimizer", "adam"), params.get("optimizer_params", {})),
                      entity_relation_regularizer=regularizers.get(reg, params.get("regularizer_params", {})) if reg else None,
                      entity_relation_initializer=params.get("initializer", "glorot_uniform"))
        return model

    best_model, best_params, best_mrr = None, None, 0
    history = []
    for params in combos:
        record = {"model_name": model_class, "model_params": dict(params)}
        try:
            model = build(params)
            callbacks = []
            if early_stopping:
                callbacks.append(EarlyStopping(monitor="val_{}".format(esp.get("criteria", "mrr")), min_delta=0,
                                               patience=esp.get("stop_interval", 5), verbose=1 if verbose else 0, mode="max",
                                               restore_best_weights=True))
            hist = model.fit(X_train, batch_size=params.get("batch_size", 1000), epochs=params.get("epochs", 100),
                             validation_data=X_valid, validation_freq=esp.get("check_interval", 10),
                             validation_batch_size=esp.get("validation_batch_size", 10), validation_burn_in=esp.get("burn_in", 0),
                             validation_corrupt_side="s,o", validation_filter=X_filter, callbacks=callbacks, focusE=focusE,
                             focusE_params=params.get("focusE_params", focusE_params), verbose=verbose)
            ranks = model.evaluate(selection, use_filter=X_filter, entities_subset=entities_subset, corrupt_side=corrupt_side,
                                   verbose=verbose)
            record["results"] = _metrics(ranks)
            if record["results"]["mrr"] > best_mrr:
                best_mrr, best_model = record["results"]["mrr"], model
                best_params = dict(params, early_stopping_epoch=len(hist.history["loss"]))
        except Exception as e:   # noqa: BLE001  (a failing combination must not end the search)
            record["results"] = {"exception": str(e)}
        history.append(record)
    if best_model is None:
        nan = float("nan")
        return None, None, best_mrr, [], {"mrr": nan, "mr": nan, "hits_1": nan, "hits_3": nan, "hits_10": nan}, history
    if retrain_best_model:
        if focusE:
            assert isinstance(X_valid, np.ndarray) and X_valid.shape[1] > 3, \
                "Validation set is used as training data for retraining the best model, but weights are missing."
        best_model = build(best_params)
        best_model.fit(np.concatenate((X_train, X_valid)), batch_size=best_params.get("batch_size", 1000),
                       epochs=best_params["early_stopping_epoch"], validation_data=None, focusE=focusE,
                       focusE_params=best_params.get("focusE_params", focusE_params), verbose=verbose)
    ranks_test = best_model.evaluate(X_test, use_filter=X_filter, verbose=verbose, entities_subset=entities_subset,
                                     corrupt_side=corrupt_side)
    return best_model, best_params, best_mrr, ranks_test, _metrics(ranks_test), history

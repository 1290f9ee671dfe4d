"""Host-side helpers around the evaluation protocol with the reference's names and semantics
(/root/reference/ampligraph/evaluation/protocol.py:27-233): dataset splitting without unseen entities and
filtering of triples whose entities the model has not seen.  Plain numpy; nothing here touches the device.
(`select_best_model_ranking`, the grid search, is a caller of fit()/evaluate() and is out of scope)"""
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
            s, p, o = cand[idx].tolist()
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

"""Triple-file loading either side of the hot path: what feeds fit()/evaluate() when the data is not already an array.
Behaviour of /root/reference/ampligraph/datasets/datasets.py:173-240 (load_from_csv) and :142-170 (reciprocal relations):
every column is read as a string, duplicate rows are dropped (first occurrence kept, order preserved), and with
add_reciprocal_rels every (s, p, o) gains (o, p + "_reciprocal", s), appended behind the originals.  The reference's
benchmark-dataset downloaders are out of scope (no network; bench.py uses datasets/synthetic.py)."""
import os

import numpy as np


def _drop_duplicate_rows(X):
    _, first = np.unique(X, axis=0, return_index=True)
    return X[np.sort(first)]


def add_reciprocal_relations(X):
    X = np.asarray(X).astype(str)
    rec = np.stack([X[:, 2], np.char.add(X[:, 1], "_reciprocal"), X[:, 0]], 1)   # numpy widens the string dtype
    return np.concatenate([X.astype(rec.dtype), rec], 0)


def load_from_csv(directory_path, file_name, sep="\t", header=None, add_reciprocal_rels=False):
    import pandas as pd

    df = pd.read_csv(os.path.join(directory_path, file_name), sep=sep, header=header, names=None, dtype=str)
    X = _drop_duplicate_rows(df.values.astype(str))
    return add_reciprocal_relations(X) if add_reciprocal_rels else X

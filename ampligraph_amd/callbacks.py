"""Keras-callback look-alikes for fit() (the reference passes tf.keras.callbacks objects through
ScoringBasedEmbeddingModel.fit(callbacks=...), :786-799,873-876; TensorFlow is not a dependency here).
fit() calls set_model / on_train_begin / on_epoch_end(epoch, logs) / on_train_end on anything duck-typed like these."""
import numpy as np


class EarlyStopping:
    """tf.keras.callbacks.EarlyStopping semantics on the fit() logs (`loss`, `val_mrr`, `val_hits@10`, ...):
    stop when `monitor` has not improved by more than `min_delta` for `patience` checks; optionally restore the
    best tables.  Validation metrics only exist on validation epochs (validation_freq): other epochs are skipped."""

    def __init__(self, monitor="val_mrr", min_delta=0.0, patience=0, verbose=0, mode="auto", baseline=None,
                 restore_best_weights=False, start_from_epoch=0):
        if mode not in ("auto", "min", "max"):
            mode = "auto"
        if mode == "auto":
            mode = "min" if ("loss" in monitor or monitor.endswith("mr")) else "max"
        self.monitor, self.min_delta, self.patience, self.verbose = monitor, abs(float(min_delta)), int(patience), verbose
        self.mode, self.baseline, self.restore_best_weights = mode, baseline, restore_best_weights
        self.start_from_epoch = int(start_from_epoch)
        self.model = None

    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None):
        self.wait, self.stopped_epoch, self.best_epoch = 0, 0, 0
        self.best = (np.inf if self.mode == "min" else -np.inf) if self.baseline is None else self.baseline
        self.best_weights = None

    def _better(self, cur, ref):
        return cur < ref - self.min_delta if self.mode == "min" else cur > ref + self.min_delta

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None or epoch < self.start_from_epoch:
            return
        self.wait += 1
        if self._better(cur, self.best):
            self.best, self.best_epoch, self.wait = cur, epoch, 0
            if self.restore_best_weights:
                eng = self.model._engine
                self.best_weights = (eng.ent.clone(), eng.rel.clone())
            return
        if self.wait >= self.patience and epoch > 0:
            self.stopped_epoch = epoch
            self.model.stop_training = True
            if self.restore_best_weights and self.best_weights is not None:
                eng = self.model._engine
                eng.ent.copy_(self.best_weights[0])
                eng.rel.copy_(self.best_weights[1])
                self.model._full_ent = None
                if self.verbose:
                    print(f"Restoring model weights from the end of the best epoch: {self.best_epoch + 1}.")

    def on_train_end(self, logs=None):
        if self.stopped_epoch > 0 and self.verbose:
            print(f"Epoch {self.stopped_epoch + 1}: early stopping")

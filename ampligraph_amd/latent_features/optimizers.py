"""Optimizer surface (/root/reference/ampligraph/latent_features/optimizers.py:255-291).  The reference
wraps a Keras *legacy* optimizer; here an Optimizer carries the hyper-parameters of the same update
rule and the dense sweep runs in HIP (ampligraph_amd/csrc/kge_opt.hip).  Supported by name, like the reference's lookup in
the Keras legacy namespace (:57-67): adam, adagrad, sgd (+ momentum / nesterov), rmsprop (+ momentum), adadelta, adamax, with
Keras legacy defaults (epsilon 1e-7, Adagrad accumulator 0.1, rho 0.9 / 0.95).  Not supported (three state tensors per table
or scalar schedules): amsgrad, centered RMSprop, Nadam, Ftrl -- rejected with ValueError."""
import numpy as np

from .. import _ffi


class OptimizerWrapper:
    """name: the Keras name ("sgd", "rmsprop", ...).  `kind` is the update rule the kernels run ("momentum" for SGD with
    momentum, "rmsprop_mom" for RMSprop with momentum): it also names the optimizer-state layout of engine / checkpoints."""

    def __init__(self, name, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, momentum=0.0, nesterov=False,
                 rho=None, amsgrad=False, centered=False, lazy=False, **unused):
        name = name.lower()
        if name not in ("sgd", "adagrad", "adam", "rmsprop", "adadelta", "adamax") or name.endswith("_mom"):
            raise ValueError("Could not interpret optimizer identifier: ", name)
        if amsgrad or centered:
            raise ValueError("amsgrad / centered RMSprop keep three state tensors per table: not supported")
        self.keras_name = name
        self.learning_rate = float(learning_rate)
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)
        self.momentum, self.nesterov = float(momentum), bool(nesterov)
        self.rho = float(rho) if rho is not None else (0.95 if name == "adadelta" else 0.9)
        self.name = name   # update rule ("kind"): what engine.prepare_training / the kernels are given
        if name == "sgd" and self.momentum != 0.0:
            self.name = "momentum"
        elif name == "rmsprop" and self.momentum != 0.0:
            self.name = "rmsprop_mom"
        self.iterations = 0
        # touched-rows mode: an opt-in that deviates from the reference's dense optimizer (optimizers.py:136-168), see
        # amdkge_opt.lazy in include/amdkge.h; set by compile(optimizer_mode="lazy") or OptimizerWrapper(lazy=True)
        self.lazy = bool(lazy)
        self.num_optimized_vars = 2          # entity and relation tables (optimizers.py:112)
        self.is_partitioned_training = False
        self._engine = None                   # bound by the step loop: where the state tensors live (HBM)

    # ---- state access with the reference's names and ordering (optimizers.py:133,177-239): Keras' optimizer.get_weights() is
    # [iterations, slot0(ent), slot0(rel), slot1(ent), slot1(rel)]; here the state tensors are the engine's optimizer slots
    def set_partitioned_training(self, value=True):
        self.is_partitioned_training = value

    def bind(self, engine):
        self._engine = engine

    def get_hyperparam_count(self):
        """Number of state tensors per optimized variable (Adam: m and v -> 2; optimizers.py:177-182)."""
        return len(_ffi.OPT_SLOTS[self.name])

    def get_iterations(self):
        return int(self.iterations)

    def get_weights(self):
        """[iterations, then per state tensor: entity part, relation part] as dense numpy arrays."""
        out = [np.int64(self.iterations)]
        eng = self._engine
        if eng is None or not getattr(eng, "slots", None):
            return out
        for nme in _ffi.OPT_SLOTS[self.name]:
            for tab in ("e", "r"):
                t = eng.slots[f"{nme}_{tab}"]
                out.append(eng.unpack(t).cpu().numpy() if hasattr(eng, "unpack") else np.array(t))
        return out

    def set_weights(self, weights):
        weights = list(weights)
        names = _ffi.OPT_SLOTS[self.name]
        if len(weights) != 1 + 2 * len(names):
            raise ValueError(f"expected {1 + 2 * len(names)} arrays (iterations + {len(names)} state tensors x entity / relation)")
        self.iterations = int(weights[0])
        eng = self._engine
        i = 1
        for nme in names:
            for tab in ("e", "r"):
                if eng is None:
                    raise RuntimeError("optimizer is not bound to an engine yet (no training step has been set up)")
                eng.pack(np.asarray(weights[i], dtype=np.float32), out=eng.slots[f"{nme}_{tab}"])
                i += 1

    def get_entity_relation_hyperparams(self):
        """(entity-table state tensors, relation-table state tensors), optimizers.py:184-201."""
        w = self.get_weights()
        return [w[i] for i in range(1, len(w), 2)], [w[i + 1] for i in range(1, len(w), 2)]

    def set_entity_relation_hyperparams(self, ent_hyperparams, rel_hyperparams):
        w = self.get_weights()
        for j, i in enumerate(range(1, len(w), 2)):
            w[i], w[i + 1] = ent_hyperparams[j], rel_hyperparams[j]
        self.set_weights(w)

    def to_ffi(self, iteration, reg_p=2):
        b1, b2 = self.beta_1, self.beta_2
        if self.name == "momentum":
            b1, b2 = self.momentum, (1.0 if self.nesterov else 0.0)
        elif self.name in ("rmsprop", "rmsprop_mom"):
            b1, b2 = self.rho, self.momentum
        elif self.name == "adadelta":
            b1, b2 = self.rho, 0.0
        return _ffi.Opt(_ffi.OPTIMIZERS[self.name], int(reg_p), self.learning_rate, b1, b2, self.epsilon, 0.0, int(iteration),
                        1 if self.lazy else 0, 0)   # row_floats is filled in by the engine (it knows the stored row layout)

    def get_config(self):
        return {"name": self.keras_name, "learning_rate": self.learning_rate, "beta_1": self.beta_1,
                "beta_2": self.beta_2, "epsilon": self.epsilon, "momentum": self.momentum, "nesterov": self.nesterov,
                "rho": self.rho, "lazy": self.lazy}


def get(identifier, hyperparams=None):
    hyperparams = dict(hyperparams or {})
    if isinstance(identifier, OptimizerWrapper):
        return identifier
    if isinstance(identifier, str):
        lr = hyperparams.pop("learning_rate", 0.001)  # optimizers.py:284
        return OptimizerWrapper(identifier, learning_rate=lr, **hyperparams)
    # duck-typed Keras-like optimizer object: class name + learning_rate attribute
    name = type(identifier).__name__.lower()
    if name in ("sgd", "adagrad", "adam", "rmsprop", "adadelta", "adamax") and hasattr(identifier, "learning_rate"):
        cfg = {k: getattr(identifier, k) for k in ("beta_1", "beta_2", "epsilon", "momentum", "nesterov", "rho", "amsgrad",
                                                   "centered") if hasattr(identifier, k)}
        return OptimizerWrapper(name, learning_rate=float(identifier.learning_rate), **cfg)
    raise ValueError("Could not interpret optimizer identifier: ", identifier)

"""save_model / restore_model with the reference's signatures and directory semantics
(/root/reference/ampligraph/utils/model_utils.py:29-129: `tf.keras.models.save_model` into a directory + `save_metadata`).
The format is this engine's own: <dir>/model.npz (tables, optimizer state, id maps; row-sharded runs add one shard file per
rank) and <dir>/model.json (constructor arguments, optimizer / loss configuration, iteration count, calibration) -- what
ScoringBasedEmbeddingModel.save_weights writes, plus what is needed to rebuild and re-compile the object."""
import json
import os
import shutil
from time import gmtime, strftime


def save_model(model, model_name_path=None, protocol=None):
    """Save a trained model to the DIRECTORY `model_name_path` (default: a timestamp in the working directory, :64-65).
    An existing path is overwritten (:66-73).  Collective under torch.distributed (every rank calls it)."""
    if model_name_path is None:
        model_name_path = "{0}".format(strftime("%Y_%m_%d-%H_%M_%S", gmtime()))
    model = getattr(model, "model", model) if getattr(model, "is_backward", False) else model   # 1.x wrappers hold the model (:74-75)
    d = model._dist() if hasattr(model, "_dist") else None
    rank0 = d is None or d.get_rank() == 0
    if rank0:
        if os.path.exists(model_name_path):
            print("The path {} already exists. This save operation will overwrite the model at the specified path.".format(model_name_path))
            shutil.rmtree(model_name_path) if os.path.isdir(model_name_path) else os.remove(model_name_path)
        os.makedirs(model_name_path)
    if d is not None:
        d.barrier()
    base = os.path.join(model_name_path, "model")
    model.save_weights(base)
    if rank0:
        meta = json.load(open(base + ".json"))
        meta["config"] = model.get_config()
        meta["focusE"] = getattr(model, "focusE_params", None) if getattr(model, "use_focusE", False) else None
        with open(base + ".json.tmp", "w") as f:
            json.dump(meta, f)
        os.replace(base + ".json.tmp", base + ".json")
    if d is not None:
        d.barrier()


def restore_model(model_name_path=None):
    """Restore a model saved by save_model: same class, tables, id maps, calibration; compiled with the saved optimizer
    (state and iteration count restored) and loss, so predict / evaluate / continued fit work at once."""
    from ..latent_features import ScoringBasedEmbeddingModel, loss_functions, optimizers

    if model_name_path is None:
        raise Exception("No default model found. Please specify model_name_path...")   # (:97-110: no default lookup here)
    base = os.path.join(model_name_path, "model")
    if not (os.path.isdir(model_name_path) and os.path.exists(base + ".json") and os.path.exists(base + ".npz")):
        raise FileNotFoundError("No model found: {}.".format(model_name_path))
    meta = json.load(open(base + ".json"))
    cfg = meta.get("config") or {"eta": meta["eta"], "k": meta["k"], "scoring_type": meta["scoring_type"], "seed": meta["seed"]}
    model = ScoringBasedEmbeddingModel.from_config(cfg)
    if meta.get("optimizer") and meta.get("loss"):
        oc = dict(meta["optimizer"])
        name = oc.pop("name")
        prm = dict(meta["loss"].get("params") or {})
        from ..latent_features import regularizers

        regs = [None if r is None else regularizers.LPRegularizer(r.get("p", 2), r.get("lambda", 1e-5), r.get("p2"), r.get("lambda2", 0.0))
                for r in (meta.get("regularizer") or [None, None])]
        model.compile(optimizer=optimizers.get(name, oc), loss=loss_functions.get(meta["loss"]["name"], prm),
                      entity_relation_regularizer=regs)
    model.load_weights(base)
    if meta.get("focusE"):
        model.use_focusE, model.focusE_params = True, meta["focusE"]
    return model

"""CPU stand-in for KgeEngine built on the oracle -- TEST INFRASTRUCTURE: lets the host-side
data-parallel logic of ampligraph_amd.trainer.StepLoop (batch sharding, global RNG rows, gradient
all-reduce, loss aggregation) run under gloo on a GPU-less box."""
import numpy as np
import torch

from oracle import kge_oracle as O

LOSS_BY_ID = {0: "pairwise", 1: "nll", 2: "absolute_margin", 3: "self_adversarial", 4: "multiclass_nll"}
OPT_BY_ID = {0: "sgd", 1: "adagrad", 2: "adam"}


class OracleEngine:
    def __init__(self, model, k, ent, rel, tiled=False):
        self.model, self.k = model, k
        if tiled:   # expose the owner-computes entry points of KgeEngine (same calling convention)
            self.train_step_tiled = self._train_step_tiled
            self.tiled_supported = lambda B, eta: True
        self.ent0, self.rel0 = ent.copy(), rel.copy()
        self.n_ents, self.n_rels = ent.shape[0], rel.shape[0]
        self.loss_acc = torch.zeros(2, dtype=torch.float64)
        self.state = None
        self.calls = []

    def prepare_training(self, optimizer):
        self.state = O.TrainState(self.ent0, self.rel0, optimizer, 0.0)
        self.g_ent = torch.zeros(self.ent0.shape, dtype=torch.float32)
        self.g_rel = torch.zeros(self.rel0.shape, dtype=torch.float32)

    def grad_tensors(self):
        return [self.g_ent, self.g_rel]

    def train_fwdbwd(self, triples, eta, loss, seed, step, row_offset=0, b_global=0, **kw):
        X = triples.numpy()
        self.calls.append((int(X.shape[0]), int(row_offset), int(b_global), int(step)))
        negs = O.generate_corruptions(X, self.n_ents, eta, seed, step, row_offset, b_global or X.shape[0])
        total, Ge, Gr, _ = O.dense_gradients(self.model, self.state.ent, self.state.rel, X, negs, eta,
                                             LOSS_BY_ID[loss.kind], {"margin": loss.margin, "alpha": loss.alpha},
                                             "mean" if loss.reduction_mean else "sum", self.n_rels)
        self.g_ent += torch.as_tensor(Ge.astype(np.float32))
        self.g_rel += torch.as_tensor(Gr.astype(np.float32))
        self.loss_acc[0] += float(total)

    def _train_step_tiled(self, triples, eta, loss, opt, seed, step, reg_e=0.0, reg_r=0.0, row_offset=0,
                          b_global=0, grad_only=False, **kw):
        """amdkge_train_step_tiled: grad_only stores the entity gradient (overwrite) and adds the relation
        gradient; otherwise it is the complete step (both tables updated, gradients left zero)."""
        self.g_ent.zero_()
        self.train_fwdbwd(triples, eta, loss, seed, step, row_offset=row_offset, b_global=b_global)
        if not grad_only:
            self.opt_step(opt, reg_e, reg_r)

    def opt_step(self, opt, lam_e=0.0, lam_r=0.0):
        self.state.lr = opt.lr
        self.state.iterations = opt.iteration - 1
        Ge, Gr = self.g_ent.numpy().astype(np.float64), self.g_rel.numpy().astype(np.float64)
        for x, G, lam in ((self.state.ent, Ge, lam_e), (self.state.rel, Gr, lam_r)):
            if lam:
                xx = x.astype(np.float64)
                self.loss_acc[1] += lam * float((np.abs(xx) ** opt.reg_p).sum())
                G += lam * opt.reg_p * np.abs(xx) ** (opt.reg_p - 1) * np.sign(xx)
        O.apply_optimizer(self.state, Ge, Gr, opt.beta1, opt.beta2, opt.epsilon)
        self.g_ent.zero_()
        self.g_rel.zero_()

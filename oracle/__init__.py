"""CPU oracle for the AmpliGraph hot path (fit / predict / evaluate of
ScoringBasedEmbeddingModel).

THIS PACKAGE IS TEST INFRASTRUCTURE.  Only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import it.  Nothing under
`ampligraph_amd/` imports, links or executes anything from here; the product
path fails loudly when the HIP library is missing.

Parity status
-------------
The reference (TensorFlow 2.15 / Keras 2, 100 % Python) cannot be imported in
this container (no tensorflow wheel, no network), so the oracle is a numpy
restatement of the reference's arithmetic, function by function, each citing
the reference file:line it follows.  It is pinned by every known-answer vector
the reference's own tests hold for this path (tests/test_oracle_kat.py
transcribes them from /root/reference/tests/ampligraph/latent_features/**):
scoring (5 models x 3 tests), ranks, five losses x two reductions, lookup,
filter sets, id assignment.

"parity unpinned" (no reference test pins them; documented in DESIGN.md):
  * optimizer update arithmetic (Keras *legacy* Adam/Adagrad/SGD live in the
    third-party tensorflow==2.15 wheel, not in /root/reference; restated from the
    published update rule),
  * gradients (the reference relies on TF autodiff; ours are hand-derived and
    cross-checked against torch.autograd in fp64 in tests/test_oracle_grads.py),
  * the negative-sampling stream (TF Philox stream; we share a counter-based
    Philox4x32-10 between oracle and kernel instead).
"""

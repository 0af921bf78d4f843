"""CPU oracle for the Interactron adaptive-detection hot path.

TEST INFRASTRUCTURE ONLY.  This package is a from-scratch, functional
(state_dict-in, tensors-out) pure-PyTorch CPU restatement of the reference
algorithm (allenai/interactron: models/interactron.py, models/detr_models/*,
models/gpt.py, models/transformer.py, models/new_transformer.py,
utils/meta_utils.py, utils/storage_utils.py).  Each function cites the
reference file:line it follows.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it -- as the checker, never as the product.  The
product package ``interactron_amd`` never imports it and has no CPU fallback.

Parity pin: the reference ships no tests or golden vectors, so
the oracle is pinned against outputs of the reference itself, imported in the
build container behind a torchvision stand-in and dumped by
``tests/golden/make_golden.py`` into ``tests/golden/*.pt``;
``tests/test_oracle_golden.py`` replays them.  Third-party arithmetic the
reference depends on and that is not under /root/reference: torch (pinned
1.9.0 there, 2.10.0 here), torchvision 0.10.0 (ResNet-50 v1.5 topology,
restated in ``oracle/detector.py``), scipy 1.8.0 ``linear_sum_assignment``
(1.15.3 here; called directly, as the reference does).
"""

"""CPU oracle for the VirTex bicaptioning pretraining step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``virtex_amd/`` may import, call or link
anything in this package: only ``tests/``, ``__graft_entry__.smoke()`` and the
baseline legs of ``bench.py`` do (``cpu_baseline``: this port on the host cores; and,
only on request with ``--stock-pytorch-baseline``, the same port on the GPU through stock
PyTorch-ROCm = what the reference's own code reaches on the part), and only as the
checker / the reported baseline -- never as the thing that is measured or shipped.

What it is
----------
A torch-CPU fp32 *restatement* ("port") of the reference hot path
(SURVEY.md section 8a): ``oracle/bicaptioning.py`` rebuilds

* torchvision's ResNet-v1.5 graph (third-party, not vendored by the reference; pinned
  only as ``torchvision>=0.10`` in ``/root/reference/requirements.txt:11``; call site
  ``virtex/modules/visual_backbones.py:43-47``),
* ``WordAndPositionalEmbedding`` (``virtex/modules/embedding.py:24-74``),
* ``TransformerDecoderTextualHead`` (``virtex/modules/textual_heads.py:146-278``),
* ``BidirectionalCaptioningModel`` / ``VirTexModel``
  (``virtex/models/captioning.py:40-138,258-283``),
* the training-step body (``scripts/pretrain_virtex.py:145-163``) with the optimizer
  grouping of ``virtex/factories.py:529-545``, ``Lookahead``
  (``virtex/optim/lookahead.py:82-102``) and the cosine/warm-up schedule
  (``virtex/optim/lr_scheduler.py:174-183``)

out of the same ``torch.nn`` primitives the reference itself dispatches to, with the
reference's parameter names, so a state dict moves between the two unchanged.

How it is pinned
----------------
The reference ships **no** tests, golden vectors or fixtures for this path
(SURVEY.md section 4, 8c).  The pin is therefore the reference itself, executed in the
build container: ``oracle/make_goldens.py`` imports the reference's own classes
verbatim from ``/root/reference`` (``oracle/reference_import.py`` stubs the missing
``torchvision`` / ``albumentations`` / ``cv2`` imports), runs them on seeded synthetic
inputs and commits loss values, logit samples and per-parameter gradient summaries to
``tests/golden/``.  ``tests/test_oracle.py`` checks this restatement against those
fixtures everywhere, and bit-for-bit against the live reference wherever
``/root/reference`` exists.  ``/root/reference`` is never read on the GPU box.
"""

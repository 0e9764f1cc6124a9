"""CPU oracle for the text-variant-consistency (TVC) hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and there only as the checker (or as the
timed CPU baseline), never as the thing shipped.  The product path in
``multimodal-detection-consistency_amd/`` never imports this package and fails
loudly when its HIP extension is missing.

Pinning status
--------------
* ``tvc_oracle`` (detector / consistency-checker / retrieval / ref-bank
  arithmetic) is PINNED: ``oracle/make_golden.py`` path-loads the reference's
  importable files (``src/ref_bank.py``, ``src/utils/metrics.py``,
  ``experiments/defenses/consistency_checker.py``) in the build container and
  writes their outputs to ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
  checks the restatement against them, and against the reference's own 20x512
  ``cache/ref_bank/references.json`` data fixture.
* ``clip_oracle`` (CLIP ViT / text towers) is **parity unpinned** against the
  reference: the reference's ``src/models/clip_model.py`` wrapper and its
  weights are absent from the snapshot and it ships no tests or golden vectors
  at that boundary.  The restatement follows the
  published OpenAI-CLIP architecture and is cross-checked against
  ``transformers.CLIPModel`` built from a local config (no download).
* ``sd_oracle`` (UNet2DConditionModel / AutoencoderKL.decode / PNDM loop, SD-1.5 and SD-2.x geometry) is **parity
  unpinned and cross-checked against nothing** (``diffusers`` is not importable, weights and vectors are absent); its
  parameter inventories reproduce the published parameter counts of both UNets.
* ``defence_check`` drives the PRODUCT's public API and the oracles above side by side (the three-method defence on PGD
  inputs: ``tests/test_gpu_auroc.py``, and ``sd_reference.auroc_delta`` in the ``cpu_baseline`` leg of ``bench.py``).
"""

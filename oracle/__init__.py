"""CPU oracle for the CLIP-on-COCO training step (TEST INFRASTRUCTURE ONLY).

This package is a CPU restatement (torch-CPU / numpy, fp32 with an fp64 option)
of the reference's algorithm for the hot path named in BASELINE.json.  It is the
checker, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The shipped package
(``sparsify_clip_amd``) must never import anything from here and fails loudly
when its HIP extension is missing.

Pinning status
--------------
* loss head, schedules, loss_type dispatch, retrieval ranks, uniformity metrics:
  PINNED.  ``oracle/make_golden.py`` imports the reference's own functions from
  ``/root/reference`` (in the build container only) and writes the fixtures in
  ``tests/golden/``; ``tests/test_oracle_golden.py`` replays them against this
  restatement without the reference.
* encoders (ViT image tower, text transformer): PARITY UNPINNED.  Their
  arithmetic lives in the un-vendored dependency ``open-clip-torch==2.29.0``
  (reference ``environment.yml:191``), which is not installed here and has no
  fixtures in the reference.  ``oracle/clip_model.py`` restates its published
  architecture; nothing in the reference can confirm it.
"""

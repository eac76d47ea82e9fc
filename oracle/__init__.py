"""CPU oracle for the Faster-RCNN hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU (fp32, PyTorch-CPU + plain C) restatement of the algorithm of the
reference hot path (antoineBarbez/2D_object_detection: models/faster_rcnn.py,
models/feature_extractor.py, models/detectors/*.py, utils/post_processing.py,
utils/boxes.py, utils/training.py, utils/losses.py, utils/metrics.py:136-208).

It is the *checker*, never the product:
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
    import it;
  * nothing under ``2d_object_detection_amd/`` imports it, and the product path raises if the
    HIP library is missing instead of falling back to anything here.

PARITY PIN STATUS
-----------------
The reference has no tests, fixtures or golden vectors and its arithmetic
lives inside TensorFlow 2.x (unpinned version; not installed in the build container, ordinary
``ModuleNotFoundError``; no network).  Therefore:

  * pure reference logic (boxes / IoU / target assignment / anchors / loss formulas) is pinned
    by *hand-derived known answers* computed from the reference source text
    (``tests/golden/known_answers.json``; derivations in ``tests/golden/make_golden.py``);
  * everything whose arithmetic lives inside TensorFlow (Keras ResNet50 graph, crop_and_resize,
    combined_non_max_suppression, Keras CCE/Huber, SGD) is restated from the public API
    contract and is **parity unpinned** against TensorFlow itself.
    PyTorch-CPU's own conv/BN/pool/autograd serve as an independent cross-check for the
    dense layers.
"""

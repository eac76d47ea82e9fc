"""MI355X-native (gfx950) Faster-RCNN hot path behind the call surface of
antoineBarbez/2D_object_detection (models/faster_rcnn.py, models/feature_extractor.py,
models/detectors/*, utils/post_processing.py).  See DESIGN.md / INTEGRATION.md.

Import as ``importlib.import_module("2d_object_detection_amd")`` (the directory name is fixed by
the project and starts with a digit).
"""
__version__ = "0.1.0"

"""Input side of the hot path: the reference's data/ package (input_pipeline.py, kitti_classes.py, build_tf_records.py)
re-built without TensorFlow, plus the synthetic batches bench.py and the tests use."""
from .synthetic import synthetic_batch  # noqa: F401

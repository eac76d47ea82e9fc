"""Create training and validation TFRecord files from the raw KITTI detection data set: the reference's
data/build_tf_records.py (same flags, same record layout, same first-N-files validation split) without TensorFlow.

usage: python -m 2d_object_detection_amd.data.build_records --images-dir ... --labels-dir ... [--output-dir ./tf_records]"""
import argparse
import os

from . import tfrecord
from .input_pipeline import example_from_files


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--images-dir", required=True, type=str, help="Path to images")
    parser.add_argument("--labels-dir", required=True, type=str, help="Path to label files")
    parser.add_argument("--output-dir", default="./tf_records", type=str,
                        help="Path to output TFRecord files: <output-dir>/train.tfrecord and <output-dir>/valid.tfrecord")
    parser.add_argument("--validation-set-size", default=500, type=int, help="Number of images to be used as a validation set")
    parser.add_argument("--image-format", default="keep", choices=("keep", "bmp"),
                        help="keep: store the image files' bytes as they are (the reference tool); bmp: re-encode the frames without "
                             "compression -- same pixels, read back at 1 ms instead of 17-19 ms a frame (tf.io.decode_image reads both)")
    return parser.parse_args(argv)


def write_tf_records(data_files, output_file, image_format="keep"):
    os.makedirs(os.path.dirname(os.path.abspath(output_file)), exist_ok=True)
    return tfrecord.write_records(output_file, (tfrecord.serialize_example(example_from_files(i, l, image_format)) for i, l in data_files))


def main(argv=None):
    args = parse_args(argv)
    train, valid = [], []
    for i, name in enumerate(sorted(os.listdir(args.images_dir))):       # (the reference takes listdir order; sorted here)
        pair = (os.path.join(args.images_dir, name), os.path.join(args.labels_dir, os.path.splitext(name)[0] + ".txt"))
        (valid if i < args.validation_set_size else train).append(pair)
    n_train = write_tf_records(train, os.path.join(args.output_dir, "train.tfrecord"), args.image_format)
    n_valid = write_tf_records(valid, os.path.join(args.output_dir, "valid.tfrecord"), args.image_format)
    print("wrote %d training and %d validation records to %s" % (n_train, n_valid, args.output_dir))


if __name__ == "__main__":
    main()

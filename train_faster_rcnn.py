"""Training driver with the command line, step loop and artefacts of the reference's train_faster_rcnn.py, on the
MI355X-native FasterRCNN.

Same flags (:19-69), piecewise-constant SGD schedule (:109-112), per-step metric updates (:130-143), per-epoch validation
pass + epoch summary (:145-225), `{step, optimizer, model}` checkpoint every 5 epochs keeping the latest one (:114-125,
:236-239) with restore at start, and the final weight file (:242-244).  Differences, all additive: the data paths may be the
reference's TFRecord files, files written by `python -m 2d_object_detection_amd.data.build_records`, or a raw KITTI
directory; `--synthetic N` trains on N generated records instead (no data set in the container); scalars go to
`<logs-dir>/<time>/faster-rcnn/{train,valid}/` as a TensorBoard event file (events.out.tfevents.*, data/tfevents.py) and as
scalars.jsonl, both with the reference's tag names; launched under `python -m torch.distributed.run` every rank reads its own shard of the records and
gradients are all-reduced over RCCL (one process per GPU).  The validation pass's image summaries (utils/images.py: the drawings of the reference's train_faster_rcnn.py:146-239) go to the same event file."""
import argparse
import datetime
import glob
import importlib
import json
import os
import sys

import torch

PKG = "2d_object_detection_amd"


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--train-data-path", type=str, help="Path to the training TFRecord file (or KITTI directory)")
    parser.add_argument("--valid-data-path", type=str, help="Path to the validation TFRecord file (or KITTI directory)")
    parser.add_argument("--logs-dir", default="logs", type=str, help="Path to the directory where to write training logs")
    parser.add_argument("--save-dir", default="saved_models", type=str,
                        help="Path to the directory where to store weights of the final model")
    parser.add_argument("--checkpoints-dir", default="checkpoints", type=str, help="Path to the directory where to store checkpoints")
    parser.add_argument("--config-file", default=None, type=str,
                        help="Path to the configuration file (reference config.json schema; default: built-in 375x1242 KITTI config)")
    parser.add_argument("--num-steps", default=100000, type=int, help="Number of parameters update")
    parser.add_argument("--num-steps-per-epoch", default=500, type=int, help="Number of steps to complete an epoch")
    parser.add_argument("--batch-size", default=2, type=int, help="Size of the batches used to update parameters (per GPU)")
    parser.add_argument("--learning-rates", nargs="+", default=[0.001, 0.0001, 0.00001], type=float, help="List of learning rate values")
    parser.add_argument("--decay-steps", nargs="*", default=[40000, 80000], type=int,
                        help="List of steps at which we decay the learning rate")
    # additions
    parser.add_argument("--synthetic", default=0, type=int, metavar="N", help="train / validate on N synthetic records (no data paths needed)")
    parser.add_argument("--depth", default=50, type=int, choices=(50, 101), help="ResNet depth of the backbone")
    parser.add_argument("--metrics-every", default=1, type=int, help="update the AP / mAP metrics every k-th training step (reference: 1)")
    parser.add_argument("--init-weights", default=None, type=str, help="weight file (FasterRCNN.save_weights format) to start from")
    parser.add_argument("--seed", default=0, type=int)
    parser.add_argument("--precision", default="bf16", choices=("bf16", "fp8"),
                        help="fp8: e4m3 / e5m2 operands in the training step's convolutions (BASELINE.json configs[4]'s precision)")
    parser.add_argument("--topology", default="c4", choices=("c4", "fpn"),
                        help="c4: the reference's single conv4 feature map; fpn: feature pyramid over C2..C4 (BASELINE.json configs[4]; "
                             "config['anchors']['scales'] then holds one scale per level P2..P5)")
    parser.add_argument("--collective-timeout-minutes", default=120.0, type=float,
                        help="time-out of the process group's collectives (several ranks): must exceed the chief's validation pass, "
                             "during which the other ranks wait in a barrier")
    args = parser.parse_args(argv)
    if not args.synthetic and not (args.train_data_path and args.valid_data_path):
        parser.error("--train-data-path and --valid-data-path are required (or use --synthetic N)")
    if len(args.learning_rates) != len(args.decay_steps) + 1:
        parser.error("need one more learning rate than decay steps")
    return args


class Mean:
    """tf.keras.metrics.Mean on device tensors (no host sync until result())."""

    def __init__(self, name=None):
        self.name, self.total, self.count = name, None, 0

    def update_state(self, value):
        v = value.detach().to(torch.float32)
        self.total = v.clone() if self.total is None else self.total + v
        self.count += 1

    def result(self):
        return float(self.total) / self.count if self.count else 0.0

    def reset_states(self):
        self.total, self.count = None, 0


class MetricStream:
    """The per-step metric updates (reference train_faster_rcnn.py:137-143) on a stream of their own: some sixty small launches
    that would otherwise sit between two replays of the step's graph (0.5 ms of a 4.4 ms loop, tools/driver_rate.py) run under the
    next step instead.  The inputs are cloned on the training stream first -- the predictions live in the step's static buffers,
    which the next step overwrites -- and everything the metrics allocate belongs to this stream: read the results inside `reading()`."""

    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device)

    def update(self, fn, *tensors):
        copies = [t.clone() for t in tensors]
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            fn(*copies)
        for c in copies:
            c.record_stream(self.stream)

    def reading(self):
        """Context in which result() / reset_states() of the metrics updated here are called."""
        return torch.cuda.stream(self.stream)

    def wrap(self, metric):
        """The metric behind its own call surface (update_state / result / reset_states), living on this stream."""
        return _OnStream(metric, self)


class _OnStream:
    def __init__(self, metric, owner):
        self.metric, self.owner, self.name = metric, owner, getattr(metric, "name", None)

    def update_state(self, *tensors):
        self.owner.update(self.metric.update_state, *tensors)

    def result(self):
        with self.owner.reading():
            return self.metric.result()

    def reset_states(self):
        with self.owner.reading():
            self.metric.reset_states()


class ScalarWriter:
    """tf.summary.create_file_writer(directory) + tf.summary.scalar(tag, value, step) (reference train_faster_rcnn.py:102-106,
    146-154): a TensorBoard event file (events.out.tfevents.*: data/tfevents.py writes the TFRecord-framed Event protos itself)
    and, beside it, the same scalars as scalars.jsonl for tools without TensorBoard."""

    def __init__(self, directory, enabled=True):
        self.fh = self.events = None
        if enabled:
            os.makedirs(directory, exist_ok=True)
            self.fh = open(os.path.join(directory, "scalars.jsonl"), "a")
            self.events = importlib.import_module(PKG + ".data.tfevents").EventFileWriter(directory)

    def scalar(self, tag, value, step):
        if self.fh:
            self.fh.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")
            self.fh.flush()
            self.events.scalar(tag, value, step)

    def image(self, tag, pil_image, step):
        """tf.summary.image(tag, to_tensor(pil_image), step) (reference train_faster_rcnn.py:180,194): the PNG goes into the event file."""
        if self.events:
            images = importlib.import_module(PKG + ".utils.images")
            self.events.image(tag, images.to_png(pil_image), pil_image.size[1], pil_image.size[0], step)


def _image_summaries(writer, epoch, step, images, gt_classes, gt_boxes, preds):
    """The first validation image with its ground-truth boxes (once, at epoch 1) and with the detections of score > 0.5 (every
    5 epochs), as the reference logs them (train_faster_rcnn.py:169-195)."""
    if not writer.events or not (epoch == 1 or epoch % 5 == 0):
        return
    from PIL import Image
    draw = importlib.import_module(PKG + ".utils.images")
    names = importlib.import_module(PKG + ".data.kitti_classes").class_names
    frame = images[0].detach().to("cpu", torch.uint8).numpy()
    if epoch == 1:
        img = Image.fromarray(frame)
        cls = gt_classes[0].detach().float().cpu()
        real = cls.sum(-1) > 0                    # (padding rows are all-zero one-hot vectors: data/input_pipeline.py)
        draw.draw_predictions_on_image(img, gt_boxes[0].detach().float().cpu()[real].tolist(), class_indices=cls[:, 1:].argmax(-1)[real].tolist(),
                                       class_names=names, relative=True)
        writer.image("Ground-truth", img, 0)
        img.close()
    if epoch % 5 == 0:
        img = Image.fromarray(frame)
        sc = preds["rcnn_scores"][0].detach().float().cpu()
        keep = sc > 0.5
        draw.draw_predictions_on_image(img, preds["rcnn_boxes"][0].detach().float().cpu()[keep].tolist(), scores=sc[keep].tolist(),
                                       class_indices=preds["rcnn_classes"][0].detach().cpu().long()[keep].tolist(), class_names=names, relative=True)
        writer.image("Predictions/pred@score=.50", img, step)
        img.close()


class CheckpointManager:
    """tf.train.CheckpointManager(max_to_keep=1) over {step, optimizer, model} (train_faster_rcnn.py:114-125)."""

    def __init__(self, directory):
        self.directory = directory

    @property
    def latest_checkpoint(self):
        files = glob.glob(os.path.join(self.directory, "ckpt-*.pt"))
        return max(files, key=lambda f: int(os.path.basename(f)[5:-3])) if files else None

    def save(self, step, model, optimizer):
        os.makedirs(self.directory, exist_ok=True)
        previous = glob.glob(os.path.join(self.directory, "ckpt-*.pt"))
        path = os.path.join(self.directory, "ckpt-%d.pt" % step)
        tmp = path + ".tmp"
        ck = {"step": step, "model": model.get_weights(), "optimizer": optimizer.state_dict()}
        fp8 = model.get_fp8_state() if hasattr(model, "get_fp8_state") else None      # delayed-scaling state of the fp8 twins (None in bf16)
        if fp8 is not None:
            ck["fp8_scales"] = fp8.cpu()
        torch.save(ck, tmp)
        os.replace(tmp, path)
        for f in previous:
            if f != path:
                os.remove(f)
        return path

    def restore(self, model, optimizer):
        path = self.latest_checkpoint
        if path is None:
            return 0
        ck = torch.load(path, map_location="cpu")
        model.set_weights(ck["model"])
        optimizer.load_state_dict(ck["optimizer"])
        if ck.get("fp8_scales") is not None and hasattr(model, "set_fp8_state"):
            model.set_fp8_state(ck["fp8_scales"])
        return int(ck["step"])


def _synthetic_records(n, cfg, seed):
    DATA = importlib.import_module(PKG + ".data")
    return [DATA.synthetic_batch(1, cfg["image_shape"], cfg["num_classes"], seed=seed + i) for i in range(n)]


def _synthetic_dataset(records, batch_size, training, rank, world, device):
    mine = records[rank::world] or records

    def gen():
        i = 0
        while True:
            if not training and i >= len(mine):
                return
            chunk = [mine[(i + k) % len(mine)] for k in range(batch_size if training else min(batch_size, len(mine) - i))]
            i += len(chunk)
            yield tuple(torch.cat([c[j] for c in chunk]).to(device, non_blocking=True) for j in range(3))
    return gen


def main(argv=None):
    args = parse_args(argv)
    D = importlib.import_module(PKG + ".distributed")
    M = importlib.import_module(PKG + ".models.faster_rcnn")
    OPT = importlib.import_module(PKG + ".optimizers")
    C = importlib.import_module(PKG + ".config")
    MET = importlib.import_module(PKG + ".utils.metrics")
    IP = importlib.import_module(PKG + ".data.input_pipeline")

    rank, world, local_rank = D.init_from_env(timeout_s=60.0 * args.collective_timeout_minutes)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    chief = rank == 0
    config = C.load_config(args.config_file) if args.config_file else C.default_config()

    if args.synthetic:
        records = _synthetic_records(args.synthetic, config, 1234)
        dataset_train = _synthetic_dataset(records, args.batch_size, True, rank, world, dev)
        dataset_valid = _synthetic_dataset(records[:max(1, args.synthetic // 4)], 1, False, 0, 1, dev)
    else:
        creator = IP.InputPipelineCreator(num_classes=config["num_classes"], image_shape=config["image_shape"])
        train_pipe = creator.create_input_pipeline(args.train_data_path, batch_size=args.batch_size, training=True, rank=rank,
                                                   world_size=world, seed=args.seed, device=dev)
        valid_pipe = creator.create_input_pipeline(args.valid_data_path, device=dev)
        dataset_train, dataset_valid = train_pipe.__iter__, valid_pipe.__iter__

    nc = config["num_classes"]
    train_cls, train_reg, train_map = Mean("train_classification_loss"), Mean("train_regression_loss"), MET.MeanAveragePrecision(nc, 0.5, name="train_mAP@IoU=.50")
    rpn_train_cls, rpn_train_reg, rpn_train_ap = Mean(), Mean(), MET.AveragePrecision(0.5, name="rpn_train_AP@IoU=.50")
    valid_cls, valid_reg, valid_map = Mean(), Mean(), MET.MeanAveragePrecision(nc, 0.5, name="valid_mAP@IoU=.50")
    rpn_valid_cls, rpn_valid_reg, rpn_valid_ap = Mean(), Mean(), MET.AveragePrecision(0.5, name="valid_AP@IoU=.50")
    metric_stream = MetricStream(dev)                    # the AP / mAP updates of every step run under the next step
    train_map, rpn_train_ap, valid_map, rpn_valid_ap = (metric_stream.wrap(m) for m in (train_map, rpn_train_ap, valid_map, rpn_valid_ap))
    every = (train_cls, train_reg, train_map, rpn_train_cls, rpn_train_reg, rpn_train_ap,
             valid_cls, valid_reg, valid_map, rpn_valid_cls, rpn_valid_reg, rpn_valid_ap)

    current_time = datetime.datetime.now().strftime("%Y%m%d-%H%M%S")
    train_writer = ScalarWriter(os.path.join(args.logs_dir, current_time, "faster-rcnn", "train"), chief)
    valid_writer = ScalarWriter(os.path.join(args.logs_dir, current_time, "faster-rcnn", "valid"), chief)

    learning_rate = OPT.PiecewiseConstantDecay(boundaries=args.decay_steps, values=args.learning_rates)
    optimizer = OPT.SGD(learning_rate=learning_rate, momentum=0.9)
    # every rank draws its own fg/bg sample positions (the Philox key differs per rank); the weights' seed is shared
    model = M.FasterRCNN(config, depth=args.depth, device=dev, seed=args.seed, sampling_seed=args.seed + rank, world_size=world,
                         precision=args.precision, topology=args.topology)
    if args.init_weights:
        model.load_weights(args.init_weights)
    optimizer.bind(model.store)

    manager = CheckpointManager(os.path.join(args.checkpoints_dir, "faster-rcnn"))
    step = manager.restore(model, optimizer)
    if chief:
        print("Restored from {}".format(manager.latest_checkpoint) if step else "Initializing from scratch.", flush=True)
    sync = D.GradientSynchronizer(model.store.g, model.store.buckets) if world > 1 else None
    hook = sync.after_segment if sync is not None else None
    for images, gt_classes, gt_boxes in dataset_train():
        step += 1
        losses, preds = model.train_step(images, gt_classes, gt_boxes, optimizer, sync_fn=hook)

        train_cls.update_state(losses["rcnn_cls"])
        train_reg.update_state(losses["rcnn_reg"])
        rpn_train_cls.update_state(losses["rpn_cls"])
        rpn_train_reg.update_state(losses["rpn_reg"])
        if step % args.metrics_every == 0:
            train_map.update_state(gt_boxes, gt_classes, preds["rcnn_boxes"], preds["rcnn_scores"], preds["rcnn_classes"])
            rpn_train_ap.update_state(gt_boxes, preds["rpn_boxes"], preds["rpn_scores"])

        if step % args.num_steps_per_epoch == 0:
            epoch = step // args.num_steps_per_epoch
            for tag, m in (("Losses/Faster-RCNN/classification_loss", train_cls), ("Losses/Faster-RCNN/regression_loss", train_reg),
                           ("Metrics/Faster-RCNN/mAP@IoU=.50", train_map), ("Losses/RPN/classification_loss", rpn_train_cls),
                           ("Losses/RPN/regression_loss", rpn_train_reg), ("Metrics/RPN/AP@IoU=.50", rpn_train_ap)):
                train_writer.scalar(tag, m.result(), step)

            # validation: one ordered pass on rank 0 (the reference is single-device).  The other ranks wait in the barrier
            # behind it; a barrier is a collective under the process group's watchdog like the all-reduce it stands in front
            # of, so the group is created with a time-out sized for this pass (--collective-timeout-minutes, default 2 h; the
            # backend default of 10 minutes would abort the waiting ranks on a long validation set)
            if world > 1:
                torch.distributed.barrier()
            if chief:
                for test_step, (vimages, vclasses, vboxes) in enumerate(dataset_valid()):
                    vlosses, vpreds = model.test_step(vimages, vclasses, vboxes)
                    if test_step == 0:
                        _image_summaries(valid_writer, epoch, step, vimages, vclasses, vboxes, vpreds)
                    valid_cls.update_state(vlosses["rcnn_cls"])
                    valid_reg.update_state(vlosses["rcnn_reg"])
                    valid_map.update_state(vboxes, vclasses, vpreds["rcnn_boxes"], vpreds["rcnn_scores"], vpreds["rcnn_classes"])
                    rpn_valid_cls.update_state(vlosses["rpn_cls"])
                    rpn_valid_reg.update_state(vlosses["rpn_reg"])
                    rpn_valid_ap.update_state(vboxes, vpreds["rpn_boxes"], vpreds["rpn_scores"])
                for tag, m in (("Losses/Faster-RCNN/classification_loss", valid_cls), ("Losses/Faster-RCNN/regression_loss", valid_reg),
                               ("Metrics/Faster-RCNN/mAP@IoU=.50", valid_map), ("Losses/RPN/classification_loss", rpn_valid_cls),
                               ("Losses/RPN/regression_loss", rpn_valid_reg), ("Metrics/RPN/AP@IoU=.50", rpn_valid_ap)):
                    valid_writer.scalar(tag, m.result(), step)

                s = f"Epoch {epoch}/{args.num_steps // args.num_steps_per_epoch}: \n"
                s += "\tFaster-RCNN: \n"
                s += f"\t\tCls Loss       --> Train: {train_cls.result():.2f}, Valid: {valid_cls.result():.2f}\n"
                s += f"\t\tReg Loss       --> Train: {train_reg.result():.2f}, Valid: {valid_reg.result():.2f}\n"
                s += f"\t\tmAP at IoU=.50 --> Train: {train_map.result():.2f}, Valid: {valid_map.result():.2f}\n"
                s += "\tRPN: \n"
                s += f"\t\tCls Loss       --> Train: {rpn_train_cls.result():.2f}, Valid: {rpn_valid_cls.result():.2f}\n"
                s += f"\t\tReg Loss       --> Train: {rpn_train_reg.result():.2f}, Valid: {rpn_valid_reg.result():.2f}\n"
                s += f"\t\tAP at IoU=.50  --> Train: {rpn_train_ap.result():.2f}, Valid: {rpn_valid_ap.result():.2f}\n"
                print(s, flush=True)
            if world > 1:
                torch.distributed.barrier()
            for m in every:
                m.reset_states()

            if epoch % 5 == 0 and chief:
                print("Saved checkpoint for epoch {}: {}".format(epoch, manager.save(step, model, optimizer)), flush=True)

        if step >= args.num_steps:
            if chief:
                out = os.path.join(args.save_dir, "faster-rcnn")
                os.makedirs(out, exist_ok=True)
                model.save_weights(os.path.join(out, "weights"))
            break
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())

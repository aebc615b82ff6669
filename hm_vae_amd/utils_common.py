"""Config / logging helpers of the training entrypoint (utils_common.py:123-147, 502-511 of the reference).
Visualisation (matplotlib/cv2 code, ~1000 lines there) is out of scope."""
import os
import time

import yaml


def get_config(config):
    with open(config, 'r') as stream:
        return yaml.load(stream, Loader=yaml.SafeLoader)


def make_result_folders(output_directory):
    image_directory = os.path.join(output_directory, 'images')
    checkpoint_directory = os.path.join(output_directory, 'checkpoints')
    for d in (image_directory, checkpoint_directory):
        if not os.path.exists(d):
            print("Creating directory: {}".format(d))
            os.makedirs(d)
    return checkpoint_directory, image_directory


def loss_members(trainer):
    """Attributes the reference reflects on for logging: non-callables whose name mentions loss/grad/nwd/accuracy."""
    keys = ('loss', 'grad', 'nwd', 'accuracy')
    out = []
    for attr in dir(trainer):
        if attr.startswith("__") or not any(k in attr for k in keys):
            continue
        try:
            v = getattr(trainer, attr)
        except Exception:
            continue
        if not callable(v):
            out.append(attr)
    return out


def write_loss(iterations, trainer, train_writer):
    for m in loss_members(trainer):
        train_writer.add_scalar(m, getattr(trainer, m), iterations + 1)


class ScalarLog:
    """Stand-in for tensorboard's SummaryWriter when tensorboard is not installed: appends `tag step value` lines."""

    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "scalars.tsv")

    def add_scalar(self, tag, value, step):
        with open(self.path, "a") as f:
            f.write("%s\t%d\t%.9g\n" % (tag, step, float(value)))


def make_writer(log_dir):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir)
    except Exception:
        return ScalarLog(log_dir)


class Timer:
    def __init__(self, msg):
        self.msg = msg
        self.start_time = None

    def __enter__(self):
        self.start_time = time.time()

    def __exit__(self, exc_type, exc_value, exc_tb):
        print(self.msg % (time.time() - self.start_time))

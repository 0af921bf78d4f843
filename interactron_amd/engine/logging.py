"""Scalar logger with the reference's TBLogger interface (utils/logging_utils.py:6-41): values are buffered per name and
their mean is written once per ``log_values()``.  Writes TensorBoard events when tensorboard is importable, and always a
``scalars.jsonl`` next to them (tensorboard is not installed on the MI355X image)."""
import json
import os

import numpy as np
import torch


class TBLogger:
    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.log_dir = log_dir
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.writer = SummaryWriter(log_dir=log_dir)
        except Exception:   # tensorboard missing
            self.writer = None
        self.scalar_buffer = {}
        self.img_buffer = {}
        self.iter_counter = 0

    def add_value(self, name, value):
        assert any(isinstance(value, t) for t in [int, float, np.ndarray, np.floating, torch.Tensor]), \
            "Invalid type {}. Only int, float, np.ndarray and torch.Tensor are accepted".format(type(value))
        if isinstance(value, torch.Tensor):
            assert len(value.shape) == 0, "Got tensor of shape {}. Only single value tensors are valid.".format(value.shape)
            value = value.item()
        self.scalar_buffer.setdefault(name, []).append(value)

    def add_image(self, name, img):
        assert isinstance(img, torch.Tensor), "Invalid type {}. Only torch.Tensor are accepted".format(type(img))
        self.img_buffer[name] = img

    def log_values(self):
        means = {name: float(np.mean(values)) for name, values in self.scalar_buffer.items()}
        with open(os.path.join(self.log_dir, "scalars.jsonl"), "a") as f:
            f.write(json.dumps({"iter": self.iter_counter, **means}) + "\n")
        if self.writer is not None:
            for name, v in means.items():
                self.writer.add_scalar(name, v, self.iter_counter)
            for name, value in self.img_buffer.items():
                self.writer.add_image(name, value, self.iter_counter, dataformats="HWC")
        self.scalar_buffer = {}
        self.img_buffer = {}
        self.iter_counter += 1

"""The scalar log the trainers write (`engine/trainers.py`): `add_value(name, x)` accumulates a running sum per name,
`log_values()` appends one JSON record of the means to `<log_dir>/scalars.jsonl` and starts the next record.

The reference's trainers hand their numbers to a TensorBoard writer (SURVEY §2 "Logging"; image dumps are out of scope and
tensorboard is not installed on the MI355X image): the class keeps that logger's name so `set_logger` / `build_trainer`
callers read the same, writes the jsonl record always and, when `torch.utils.tensorboard` is importable, the same means as
scalars of an event file in `log_dir` (`writer`)."""
import json
import os

try:   # optional: when tensorboard is importable the means also go to an event file, as the reference's logger writes them
    from torch.utils.tensorboard import SummaryWriter
except Exception:   # noqa: BLE001  (not installed on the MI355X image)
    SummaryWriter = None


class TBLogger:
    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.log_dir = log_dir
        self.path = os.path.join(log_dir, "scalars.jsonl")
        self.record = 0
        self._sums = {}     # name -> [sum, count]
        self.writer = SummaryWriter(log_dir) if SummaryWriter is not None else None

    def add_value(self, name, x):
        assert getattr(x, "ndim", 0) == 0, "add_value takes scalars (python / numpy numbers, 0-d tensors), got shape %s" % (getattr(x, "shape", None),)
        x = float(x)        # (one D2H copy for a device tensor)
        acc = self._sums.setdefault(name, [0.0, 0])
        acc[0] += x
        acc[1] += 1

    def log_values(self):
        line = {"iter": self.record}
        line.update((name, s / n) for name, (s, n) in self._sums.items())
        with open(self.path, "a") as f:
            f.write(json.dumps(line) + "\n")
        if self.writer is not None:
            for name, v in line.items():
                if name != "iter":
                    self.writer.add_scalar(name, v, self.record)
            self.writer.flush()
        self._sums.clear()
        self.record += 1

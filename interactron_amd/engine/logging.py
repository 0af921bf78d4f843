"""The scalar log the trainers write (`engine/trainers.py`): `add_value(name, x)` accumulates a running sum per name,
`log_values()` appends one JSON record of the means to `<log_dir>/scalars.jsonl` and starts the next record.

The reference's trainers hand their numbers to a TensorBoard writer (SURVEY §2 "Logging", out of scope: tensorboard is not
installed on the MI355X image and image dumps are not part of the path); the class keeps that logger's name so
`set_logger` / `build_trainer` callers read the same, nothing else of it."""
import json
import os


class TBLogger:
    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "scalars.jsonl")
        self.record = 0
        self._sums = {}     # name -> [sum, count]

    def add_value(self, name, x):
        x = float(x)        # python / numpy scalars and 0-d tensors (one D2H copy for a device tensor); anything else raises
        acc = self._sums.setdefault(name, [0.0, 0])
        acc[0] += x
        acc[1] += 1

    def log_values(self):
        line = {"iter": self.record}
        line.update((name, s / n) for name, (s, n) in self._sums.items())
        with open(self.path, "a") as f:
            f.write(json.dumps(line) + "\n")
        self._sums.clear()
        self.record += 1

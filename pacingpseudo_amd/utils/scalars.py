"""Scalar log with the tags of the reference's TensorBoard writer (train_chaos.py:183-185, :362-367, :416-423;
upper_bound_chaos.py:180-182, :216-222).  tensorboard is not part of this image, so ``tb_writer.add_scalar(tag, value,
step)`` becomes one JSON line ``{"tag": ..., "value": ..., "step": ...}`` in ``<run>/tb_summary/scalars.jsonl``."""
import json
import os


class ScalarLog:
    def __init__(self, path: str):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        self.path = path
        self._f = open(path, 'a')

    def add(self, tag: str, value, step: int):
        self._f.write(json.dumps(dict(tag=tag, value=float(value), step=int(step))) + '\n')
        self._f.flush()

    add_scalar = add          # the SummaryWriter spelling

    def close(self):
        self._f.close()

"""Tabular stdout / text-file logger (mirror of the reference's util/logger.py surface: log, print_log, write_log,
set_step_key, configure_output_file; values are averaged over ranks like util/logger.py:163-174)."""
import collections

import torch

from . import mp_util


class Logger:
    @staticmethod
    def print(*args, **kwargs):
        if mp_util.is_root_proc():
            print(*args, **kwargs)

    def __init__(self):
        self._entries = collections.OrderedDict()
        self._file = None
        self._step_key = None
        self._wrote_header = False

    def set_step_key(self, key):
        self._step_key = key

    def configure_output_file(self, filename=None):
        if filename and mp_util.is_root_proc():
            import os
            os.makedirs(os.path.dirname(filename) or ".", exist_ok=True)
            self._file = open(filename, "w")

    def log(self, key, val, collection=None, quiet=False):
        if torch.is_tensor(val):
            val = val.item()
        self._entries[key] = (float(val), collection, quiet)

    def _aggregate(self):
        if mp_util.enable_mp() and len(self._entries) > 0:
            keys = list(self._entries.keys())
            buf = torch.tensor([self._entries[k][0] for k in keys], dtype=torch.float64, device=mp_util.get_device())
            mp_util.reduce_inplace_mean(buf)
            for k, v in zip(keys, buf.tolist()):
                self._entries[k] = (v, self._entries[k][1], self._entries[k][2])

    def print_log(self):
        self._aggregate()
        if not mp_util.is_root_proc():
            return
        print("-" * 46)
        for k, (v, _, quiet) in self._entries.items():
            if not quiet:
                print("| {:<26s} | {:>13.6g} |".format(k[:26], v))
        print("-" * 46)

    def write_log(self):
        if self._file is None:
            return
        keys = list(self._entries.keys())
        if not self._wrote_header:
            self._file.write("\t".join(keys) + "\n")
            self._wrote_header = True
        self._file.write("\t".join("{:.8g}".format(self._entries[k][0]) for k in keys) + "\n")
        self._file.flush()

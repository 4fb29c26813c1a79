"""``--key value ...`` argument table (mirror of the reference's util/arg_parser.py:3-140: same methods, first
definition of a key wins, '#' starts a comment, files hold the same tokens)."""
import re


class ArgParser:
    def __init__(self):
        self._table = dict()

    def clear(self):
        self._table.clear()

    def load_args(self, arg_strs):
        key, vals = "", []
        for tok in list(arg_strs) + ["--"]:
            if len(tok) > 0 and tok[0] == "#":
                continue
            if tok.startswith("--") and (len(tok) >= 3 or tok == "--"):
                if key != "" and key not in self._table:
                    self._table[key] = vals
                key, vals = tok[2:], []
            else:
                vals.append(tok)
        return True

    def load_file(self, filename):
        with open(filename, "r") as f:
            toks = []
            for line in re.split(r"[\n\r]+", f.read()):
                if len(line) > 0 and line[0] != "#":
                    toks += line.split()
        return self.load_args(toks)

    def has_key(self, key):
        return key in self._table

    def parse_string(self, key, default=""):
        return self._table[key][0] if self.has_key(key) else default

    def parse_strings(self, key, default=[]):
        return self._table[key] if self.has_key(key) else default

    def parse_int(self, key, default=0):
        return int(self._table[key][0]) if self.has_key(key) else default

    def parse_ints(self, key, default=[]):
        return [int(s) for s in self._table[key]] if self.has_key(key) else default

    def parse_float(self, key, default=0.0):
        return float(self._table[key][0]) if self.has_key(key) else default

    def parse_floats(self, key, default=[]):
        return [float(s) for s in self._table[key]] if self.has_key(key) else default

    def parse_bool(self, key, default=False):
        return self._table[key][0] in ("true", "True", "1", "T", "t") if self.has_key(key) else default

"""Shim: the subset of configargparse the reference encoder CLI uses.

Config files hold `key = value` lines (`;`/`#` start comments); their values act
as lowest-priority command-line arguments.
"""
import argparse
import sys


class ArgumentParser(argparse.ArgumentParser):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._cfg_dests = []
        self._seen = ""

    def add(self, *names, is_config_file=False, **kw):
        act = self.add_argument(*names, **kw)
        if is_config_file:
            self._cfg_dests.append(act.dest)
        return act

    def parse_args(self, args=None, namespace=None):
        args = list(sys.argv[1:] if args is None else args)
        pre, _ = super().parse_known_args(args)
        extra = []
        for dest in self._cfg_dests:
            path = getattr(pre, dest, None)
            if not path:
                continue
            for line in open(path):
                line = line.split(";")[0].split("#")[0].strip()
                if not line or "=" not in line:
                    continue
                k, v = [x.strip() for x in line.split("=", 1)]
                extra.append(f"--{k}={v}")
        self._seen = " ".join(extra + args)
        return super().parse_args(extra + args, namespace)

    def format_values(self):
        return self._seen

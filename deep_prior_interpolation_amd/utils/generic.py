"""Small host helpers (drop-in for reference utils/generic.py): time strings, args.txt JSON I/O."""
import json
import random
import string
from argparse import Namespace
from math import ceil, floor, log2, log10

__all__ = ["nextpow2", "random_code", "ten_digit", "sec2time", "time2sec", "read_args", "write_args"]


def nextpow2(x):
    return ceil(log2(abs(x)))


def random_code(n=6):
    return "".join(random.choice(string.ascii_letters + string.digits) for _ in range(int(n)))


def ten_digit(number):
    """Number of decimal digits of a positive number (zero-padding width of patch / iteration names)."""
    return int(floor(log10(number)) + 1)


def sec2time(seconds):
    return "%dh:%dm:%ds" % (seconds // 3600, (seconds // 60) % 60, seconds % 60)


def time2sec(timestamp):
    h, m, s = timestamp.split(":")
    return int(h.replace("h", "")) * 3600 + int(m.replace("m", "")) * 60 + int(s.replace("s", ""))


def read_args(filename):
    args = Namespace()
    with open(filename, "r") as fp:
        args.__dict__.update(json.load(fp))
    return args


def write_args(filename, args, indent=2):
    with open(filename, "w") as fp:
        json.dump(args.__dict__, fp, indent=indent)

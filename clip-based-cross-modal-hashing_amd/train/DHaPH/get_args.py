"""DHaPH flags (reference train/DHaPH/get_args.py:7-20).  Upstream declares margin / alpha / tau with type=int and float defaults; the
types below accept what the defaults are."""
from argsbase import method_args

FLAGS = [("--HM", int, 500), ("--margin", float, 0.1), ("--topk", int, 15), ("--alpha", float, 1), ("--tau", float, 0.3)]


def get_args(main_args):
    return method_args(main_args, FLAGS)

"""DCHMT flags (reference train/DCHMT/get_args.py:11-16)."""
from argsbase import method_args

FLAGS = [("--hash-layer", str, "select", "select | linear"), ("--similarity-function", str, "euclidean", "cosine | euclidean"),
         ("--loss-type", str, "l2", "l1 | l2"), ("--vartheta", float, 0.5, "tolerated share of wrong code bits"),
         ("--sim-threshold", float, 0.1)]


def get_args(main_args):
    return method_args(main_args, FLAGS)

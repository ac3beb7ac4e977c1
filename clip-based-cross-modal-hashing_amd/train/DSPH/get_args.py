"""DSPH flags (reference train/DSPH/get_args.py:11-13)."""
from argsbase import method_args

FLAGS = [("--numclass", int, 24), ("--hypseed", int, 0), ("--alpha", float, 0.8)]


def get_args(main_args):
    return method_args(main_args, FLAGS)

"""DMsH-LN flags (reference train/DMsH_LN/get_args.py:7-19: numclass, hypseed, alpha on top of the base flags)."""
from argsbase import method_args

FLAGS = [("--numclass", int, 24), ("--hypseed", int, 0), ("--alpha", float, 0.8)]


def get_args(main_args):
    return method_args(main_args, FLAGS)

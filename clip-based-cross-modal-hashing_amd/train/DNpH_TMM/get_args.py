"""DNpH flags (reference train/DNpH_TMM/get_args.py:7-14: the base flags only)."""
from argsbase import method_args


def get_args(main_args):
    return method_args(main_args, [])

"""MITH flags (reference train/MITH/get_args.py:11-22)."""
from argsbase import method_args

FLAGS = [("--dropout", float, 0), ("--transformer-layers", int, 2), ("--activation", str, "gelu"), ("--top-k-label", int, 8),
         ("--res-mlp-layers", int, 2), ("--hyper-lambda", float, 0.99), ("--hyper-tokens-intra", float, 1),
         ("--hyper-cls-inter", float, 10), ("--hyper-quan", float, 8), ("--hyper-info-nce", float, 50),
         ("--hyper-alpha", float, 0.01), ("--hyper-distill", float, 1)]


def get_args(main_args):
    return method_args(main_args, FLAGS)

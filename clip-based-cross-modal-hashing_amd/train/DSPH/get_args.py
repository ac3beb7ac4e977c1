"""DSPH flags (reference train/DSPH/get_args.py:11-13)."""
import os

from argsbase import get_baseargs, merge


def get_args(main_args):
    parser = get_baseargs()
    parser.add_argument("--numclass", type=int, default=24)
    parser.add_argument("--hypseed", type=int, default=0)
    parser.add_argument("--alpha", type=float, default=0.8)
    args = merge(parser, main_args)
    args.save_dir = os.path.join(args.save_dir, args.method, args.dataset, str(args.output_dim))
    return args

"""DCHMT flags (reference train/DCHMT/get_args.py:11-16)."""
import os

from argsbase import get_baseargs, merge


def get_args(main_args):
    parser = get_baseargs()
    parser.add_argument("--hash-layer", type=str, default="select", help="[select, linear]")
    parser.add_argument("--similarity-function", type=str, default="euclidean", help="[cosine, euclidean]")
    parser.add_argument("--loss-type", type=str, default="l2", help="[l1, l2]")
    parser.add_argument("--vartheta", type=float, default=0.5, help="the rate of error code.")
    parser.add_argument("--sim-threshold", type=float, default=0.1)
    args = merge(parser, main_args)
    args.save_dir = os.path.join(args.save_dir, args.method, args.dataset, str(args.output_dim))
    return args

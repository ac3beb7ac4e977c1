"""DNPH flags (reference train/DNPH_TOMM/get_args.py)."""
import os

from argsbase import get_baseargs, merge


def get_args(main_args):
    parser = get_baseargs()
    args = merge(parser, main_args)
    args.save_dir = os.path.join(args.save_dir, args.method, args.dataset, str(args.output_dim))
    return args

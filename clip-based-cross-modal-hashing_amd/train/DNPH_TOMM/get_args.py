"""DNPH flags (reference train/DNPH_TOMM/get_args.py): the base flags only."""
from argsbase import method_args


def get_args(main_args):
    return method_args(main_args, [])

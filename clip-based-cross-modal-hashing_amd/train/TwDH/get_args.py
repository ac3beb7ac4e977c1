"""TwDH flags (reference train/TwDH/get_args.py:11-15)."""
from argsbase import method_args

FLAGS = [("--long_center", str, "./train/TwDH/center/mirflickr/long"), ("--short_center", str, "./train/TwDH/center/mirflickr/short"),
         ("--trans_matrix", str, "./train/TwDH/center/mirflickr/trans"), ("--quan_alpha", float, 0.5), ("--low_rate", float, 0),
         ("--synthetic-centers", int, 1,
          "this build: draw +-1 centres / random transition matrices instead of loading the reference's .pkl assets")]


def get_args(main_args):
    return method_args(main_args, FLAGS)

"""Shared flags of every method — the names and defaults of the reference's argsbase.py:8-35, kept in one table."""
import argparse
import os


def str2bool(v):
    if isinstance(v, bool):
        return v
    return str(v).lower() in ("1", "true", "yes", "y", "t")


# (flag, type, default[, help]) — upstream's flags first, this build's additions last
BASE_FLAGS = [
    ("--save-dir", str, "./result/"), ("--save-mat", str2bool, True), ("--save-model", str2bool, False),
    ("--save_csv", str2bool, True), ("--valid", None, True), ("-vit-use", str2bool, True), ("-clip-path", str, "./ViT-B-32.pt"),
    ("--pretrained", str, ""), ("--epochs", int, 200), ("--max-words", int, 32), ("--resolution", int, 224),
    ("--batch-size", int, 300), ("--num-workers", int, 8), ("--query-num", int, 5000), ("--train-num", int, 10000),
    ("--lr-decay-freq", int, 5), ("--display-step", int, 50), ("--seed", int, 1814), ("--lr", float, 0.001),
    ("--lr-decay", float, 0.9), ("--clip-lr", float, 0.00001), ("--weight-decay", float, 0.2),
    ("--warmup-proportion", float, 0.1, "share of the training steps spent on the linear warm-up of the learning rate"),
    ("--gemm-dtype", str, "f32", "encoder GEMM arithmetic: f32 = reference parity, bf16 = throughput, fp8 = e4m3 block GEMMs, inference only (this build)"),
    ("--data-dir", str, "", "directory with index.mat / caption.mat|txt / label.mat (this build; upstream hard-codes it)"),
    ("--synthetic-size", int, 2000, "items of the synthetic dataset (this build)"),
]


def add_flags(parser, table):
    for flag, typ, default, *doc in table:
        kw = {"default": default}
        if typ is not None:
            kw["type"] = typ
        if doc:
            kw["help"] = doc[0]
        if flag == "--gemm-dtype":
            kw["choices"] = ["f32", "bf16", "fp8"]      # fp8: inference only (--is-train false)
        parser.add_argument(flag, **kw)
    return parser


def get_baseargs():
    return add_flags(argparse.ArgumentParser(), BASE_FLAGS)


def merge(parser, main_args):
    """The reference parses sys.argv twice with two strict parsers, which makes every CLI
    flag fatal; here both parsers ignore what they do not know."""
    args, _ = parser.parse_known_args()
    merged = dict(vars(args))
    merged.update(vars(main_args))
    return argparse.Namespace(**merged)


def method_args(main_args, table):
    """Base flags + the method's own table, merged with main.py's arguments; save_dir = <save-dir>/<method>/<dataset>/<bits>
    (what every upstream get_args.py ends with)."""
    args = merge(add_flags(get_baseargs(), table), main_args)
    args.save_dir = os.path.join(args.save_dir, args.method, args.dataset, str(args.output_dim))
    return args

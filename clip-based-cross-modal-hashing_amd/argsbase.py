"""Shared flags of every method — same names and defaults as the reference's argsbase.py:8-35."""
import argparse


def str2bool(v):
    if isinstance(v, bool):
        return v
    return str(v).lower() in ("1", "true", "yes", "y", "t")


def get_baseargs():
    parser = argparse.ArgumentParser()
    parser.add_argument("--save-dir", type=str, default="./result/")
    parser.add_argument("--save-mat", type=str2bool, default=True)
    parser.add_argument("--save-model", type=str2bool, default=False)
    parser.add_argument("--save_csv", type=str2bool, default=True)
    parser.add_argument("--valid", default=True)
    parser.add_argument("-vit-use", type=str2bool, default=True)
    parser.add_argument("-clip-path", type=str, default="./ViT-B-32.pt")
    parser.add_argument("--pretrained", type=str, default="")
    parser.add_argument("--epochs", type=int, default=200)
    parser.add_argument("--max-words", type=int, default=32)
    parser.add_argument("--resolution", type=int, default=224)
    parser.add_argument("--batch-size", type=int, default=300)
    parser.add_argument("--num-workers", type=int, default=8)
    parser.add_argument("--query-num", type=int, default=5000)
    parser.add_argument("--train-num", type=int, default=10000)
    parser.add_argument("--lr-decay-freq", type=int, default=5)
    parser.add_argument("--display-step", type=int, default=50)
    parser.add_argument("--seed", type=int, default=1814)
    parser.add_argument("--lr", type=float, default=0.001)
    parser.add_argument("--lr-decay", type=float, default=0.9)
    parser.add_argument("--clip-lr", type=float, default=0.00001)
    parser.add_argument("--weight-decay", type=float, default=0.2)
    parser.add_argument("--warmup-proportion", type=float, default=0.1,
                        help="Proportion of training to perform linear learning rate warmup for.")
    # additions of this build (absent upstream)
    parser.add_argument("--gemm-dtype", type=str, default="f32", choices=["f32", "bf16"],
                        help="encoder GEMM arithmetic: f32 = reference parity, bf16 = throughput")
    parser.add_argument("--data-dir", type=str, default="", help="directory with index/caption/label .mat files")
    parser.add_argument("--synthetic-size", type=int, default=2000, help="items of the synthetic dataset")
    return parser


def merge(parser, main_args):
    """The reference parses sys.argv twice with two strict parsers, which makes every CLI
    flag fatal; here both parsers ignore what they do not know."""
    args, _ = parser.parse_known_args()
    merged = dict(vars(args))
    merged.update(vars(main_args))
    return argparse.Namespace(**merged)

"""DSPH proxy loss — HyP (reference train/DSPH/loss.py:10-72) as ONE fused native call.

`proxies` [numclass, K] is seeded exactly like upstream (torch.manual_seed(hypseed); randn;
kaiming_normal_ fan_out, :14-17).  `threshold` is read from the code-length table (:19-20) that
tests/golden/make_golden.py extracted from the reference's codetable.xlsx into codetable.json
(numbers only): rows[str(output_dim)][ceil(log2(numclass))]."""
import json
import math
import os

import torch
import torch.nn as nn

import cmh_native as N
from model.base.model import no_backward

_TABLE = None


def code_threshold(output_dim: int, numclass: int) -> float:
    global _TABLE
    if _TABLE is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "codetable.json")) as f:
            _TABLE = json.load(f)["rows"]
    return float(_TABLE[str(output_dim)][math.ceil(math.log(numclass, 2))])


class HyP(torch.nn.Module):
    def __init__(self, args=None, numclass=24, output_dim=64, hypseed=0, alpha=0.8):
        torch.nn.Module.__init__(self)
        if args is not None:
            numclass, output_dim = args.numclass, args.output_dim
            hypseed, alpha = args.hypseed, args.alpha
        self.alpha = alpha
        torch.manual_seed(hypseed)
        self.proxies = torch.nn.Parameter(torch.randn(numclass, output_dim))
        nn.init.kaiming_normal_(self.proxies, mode='fan_out')
        self.threshold = code_threshold(output_dim, numclass)

    def forward(self, x=None, y=None, label=None):
        label = label.to(x.device)
        if torch.is_grad_enabled() and (x.requires_grad or y.requires_grad or self.proxies.requires_grad):
            from backward_ops import HypLoss
            return HypLoss.apply(x, y, label, self.proxies, self.threshold, self.alpha)
        return N.dsph_hyp_loss(x, y, label, self.proxies, self.threshold, self.alpha)

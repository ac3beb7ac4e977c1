"""DNPH_out (reference train/DNPH_TOMM/loss.py:6-32) as one native call."""
import torch

import cmh_native as N
from model.base.model import no_backward


class DNPH_out(torch.nn.Module):
    def __init__(self, args):
        torch.nn.Module.__init__(self)
        self.args = args
        self.proxies = torch.nn.Parameter((torch.randn(self.args.nclass, self.args.output_dim) / 8))
        self.mrg = 1.0

    def forward(self, feature_1, feature_2, predict_1, predict_2, label_1, label_2, noise_1=None, noise_2=None,
                noise_weight=0.1):
        """label_1 must equal label_2 (the trainer passes the same tensor twice).  With noise rows the returned value
        is the full step loss `loss1 - 0.1*noise_loss` (hash_train.py:76-81), otherwise DNPH_out's loss1."""
        if torch.is_grad_enabled() and any(t.requires_grad for t in (feature_1, feature_2, predict_1, predict_2, self.proxies)):
            from backward_ops import DnphLoss
            return DnphLoss.apply(feature_1, feature_2, predict_1, predict_2, label_1, self.proxies, noise_1, noise_2, self.mrg,
                                  noise_weight)
        total, loss1, _ = N.dnph_loss(feature_1, feature_2, predict_1, predict_2, label_1, self.proxies, noise_1, noise_2,
                                      self.mrg, noise_weight)
        return no_backward(loss1 if noise_1 is None else total, self.proxies)

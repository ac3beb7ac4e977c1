"""MultiSimilarityLoss of DMsH-LN (reference train/DMsH_LN/MSLOSS.py:4-55) as ONE native forward and one native backward
(csrc/msl.hip): the row-normalised B x B similarity matrix, the label-code similarity bits, the per-row mining and the two
log-sum-exp terms never leave the GPU and no Python loop runs over the batch.

Only the branch the reference's trainer takes is built (train/DMsH_LN/hash_train.py:58-60 passes no `dataset`): the
"cifar10-1" branch raises."""
import torch
import torch.nn as nn

import cmh_native as N


class _MslFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, labels, feat2):
        feats, labels = N.f32c(feats), N.f32c(labels)
        feat2 = None if feat2 is None else N.f32c(feat2)
        ctx.save_for_backward(feats, labels, *([] if feat2 is None else [feat2]))
        return N.msl_loss(feats, labels, feat2)

    @staticmethod
    def backward(ctx, dloss):
        feats, labels, *rest = ctx.saved_tensors
        feat2 = rest[0] if rest else None
        dfeats, dfeat2 = N.msl_loss_backward(feats, labels, feat2, dloss)
        return dfeats, None, dfeat2


class MultiSimilarityLoss(nn.Module):
    def __init__(self):
        super(MultiSimilarityLoss, self).__init__()
        self.thresh, self.margin, self.scale_pos, self.scale_neg = 0.5, 0.1, 2.0, 40.0      # fixed in libcmh as in the reference (:7-11)

    def forward(self, feats, labels, dataset="MSLOSS", feat2=None):
        assert feats.size(0) == labels.size(0), \
            f"feats.size(0): {feats.size(0)} is not equal to labels.size(0): {labels.size(0)}"
        if dataset == "cifar10-1":
            raise NotImplementedError("MultiSimilarityLoss: the single-label ('cifar10-1') branch is not built")
        labels = labels.detach()                                      # `labels @ labels.t() > 0` (:26) carries no gradient
        needs_grad = torch.is_grad_enabled() and (feats.requires_grad or (feat2 is not None and feat2.requires_grad))
        loss = _MslFn.apply(feats, labels, feat2) if needs_grad else N.msl_loss(feats, labels, feat2)
        if needs_grad and float(loss.detach()) == 0.0:
            # every row was skipped (a row's loss is > 0 whenever it counts): the reference then returns a fresh constant (:53-54), so
            # backward() reaches no parameter and the optimiser skips them all - e.g. the whole run of a freshly initialised LabelNet,
            # whose codes mark every pair as similar
            return torch.zeros([], device=feats.device, requires_grad=True)
        return loss

"""Self-paced contrastive loss of DHaPH (reference train/DHaPH/MSLoss.py:6-33) as one native forward and one native backward
(csrc/msl.hip `spl_*`): the cosine matrix, the label-sharing mask, the detached self-paced weights and the per-row log-ratio stay on
the GPU.  Passing the same tensor for both features (the image-image / text-text calls, train/DHaPH/hash_train.py:68-69) gives the
sum of both roles' gradients, as autograd does."""
import torch
import torch.nn as nn

import cmh_native as N


class _SplFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, labels, temperature, delta):
        a, labels = N.f32c(a), N.f32c(labels)
        b = None if b is None else N.f32c(b)
        ctx.save_for_backward(a, labels, *([] if b is None else [b]))
        ctx.temperature, ctx.delta = temperature, delta
        return N.spl_loss(a, b, labels, temperature, delta)

    @staticmethod
    def backward(ctx, dloss):
        a, labels, *rest = ctx.saved_tensors
        b = rest[0] if rest else None
        da, db = N.spl_loss_backward(a, b, labels, ctx.temperature, ctx.delta, dloss)
        return da, db, None, None, None


class MSLoss(nn.Module):
    def __init__(self, temperature=0.3, totalepoch=100, self_paced=True):
        super(MSLoss, self).__init__()
        self.temperature = temperature
        self.totalepoch = totalepoch
        self.self_paced = self_paced

    def delta(self, epoch):
        """:23-27 (0 switches the weights off: exp(.)^0 = 1)"""
        if not self.self_paced:
            return 0.0
        third = int(self.totalepoch / 3)
        return epoch / third if epoch <= third else 1.0

    def forward(self, image_feature, text_feature, labels=None, epoch=0):
        same = text_feature is image_feature
        labels = labels.detach().float()
        delta = self.delta(epoch)
        b = None if same else text_feature
        if torch.is_grad_enabled() and (image_feature.requires_grad or text_feature.requires_grad):
            return _SplFn.apply(image_feature, b, labels, float(self.temperature), float(delta))
        return N.spl_loss(image_feature, b, labels, self.temperature, delta)

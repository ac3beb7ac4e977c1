"""qmi_loss of DNpH (reference train/DNpH_TMM/loss.py:5-72) as ONE native forward (three B x B cosine matrices, the label
indicator and the clamped quadratic-mutual-information sum, nothing B x B materialised) and one native backward.

Only the reference's default configuration is built - use_cosine=True, use_square_clamp=True, M = B^2 / sum(D) (the trainer,
train/DNpH_TMM/hash_train.py:60, passes nothing else); other arguments raise."""
import torch

import cmh_native as N


class _QmiLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images, texts, targets, eps):
        images, texts = N.f32c(images), N.f32c(texts)
        loss, sum_d, packed = N.qmi_loss(images, texts, targets, eps)
        ctx.save_for_backward(images, texts, packed, sum_d)
        ctx.classes, ctx.eps = targets.shape[1], eps
        return loss

    @staticmethod
    def backward(ctx, dloss):
        images, texts, packed, sum_d = ctx.saved_tensors
        dimg, dtxt = N.qmi_loss_backward(images, texts, packed, ctx.classes, sum_d, dloss, ctx.eps)
        return dimg, dtxt, None, None


def qmi_loss(images, texts, targets, sigma=3, M=0, eps=1e-8, use_cosine=True, use_square_clamp=True):
    if not use_cosine or not use_square_clamp or M != 0:
        raise NotImplementedError("qmi_loss: only the reference trainer's configuration (use_cosine, use_square_clamp, M=0) is built")
    targets = targets.to(images.device).float()
    if torch.is_grad_enabled() and (images.requires_grad or texts.requires_grad):
        return _QmiLoss.apply(images, texts, targets, float(eps))
    return N.qmi_loss(images, texts, targets, eps)[0]

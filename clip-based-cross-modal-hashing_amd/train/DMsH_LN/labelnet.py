"""LabelNet of DMsH-LN (reference train/DMsH_LN/labelnet.py:6-21): labels -> fc1 -> ReLU -> fc2 -> tanh(alpha * .), alpha = sqrt(epoch
+ 1).  The trainer only uses its codes to decide which pairs count as similar (`codes . codes^T > 0`, MSLOSS.py:26), a comparison
that passes no gradient back - so the two layers run forward-only on libcmh's small-linear kernel."""
import math

import torch
import torch.nn as nn

import cmh_native as N


def _linear(x, w, b, act, **kw):
    """cmh_linear_act wants K % 4 == 0 (float4 lanes): odd label / hidden widths (21 classes, (21 + 128) // 2 = 74) get zero columns"""
    pad = (-x.shape[1]) % 4
    if pad:
        x, w = torch.nn.functional.pad(x, (0, pad)), torch.nn.functional.pad(w, (0, pad))
    return N.linear_act(x, w, b, act, **kw)


class LabelNet(nn.Module):
    def __init__(self, label_dim, code_len):
        super(LabelNet, self).__init__()
        self.fc1 = nn.Linear(label_dim, (label_dim + code_len) // 2)
        self.fc2 = nn.Linear((label_dim + code_len) // 2, code_len)
        self.alpha = 1.0

    def forward(self, x, device=None):
        x = x.to(device if device is not None else self.fc1.weight.device).to(torch.float32)
        with torch.no_grad():
            feat = _linear(x, self.fc1.weight, self.fc1.bias, N.ACT_RELU)
            hid = _linear(feat, self.fc2.weight, self.fc2.bias, N.ACT_NONE)
            # tanh(alpha * hid): the kernel's (x W^T + b) * mask * scale form with an all-ones mask and scale = alpha
            code = _linear(feat, self.fc2.weight, self.fc2.bias, N.ACT_TANH, drop_mask=torch.ones_like(hid), p=1.0 - 1.0 / self.alpha)
        return feat, hid, code

    def set_alpha(self, epoch):
        self.alpha = math.pow((1.0 * epoch + 1.0), 0.5)

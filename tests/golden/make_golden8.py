#!/usr/bin/env python3
"""Eighth golden generator — image side of the input pipeline (dataset/base.py:35-44, :55-64).  torchvision is not in this
image, so the transform chain is run on its third-party parts directly: Pillow's own Image.resize(BICUBIC) / Image.crop
(the calls torchvision's Resize / CenterCrop make on a PIL image, with torchvision's published size and origin rules) and
ATen float32 ops for ToTensor / Normalize.  Records the uint8 image that reaches ToTensor for the small cases, SHA-256 of it
for the 224-pixel ones, and the float tensor of two cases."""
import hashlib
import os
import sys

import numpy as np
import PIL
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import preputil as pu  # noqa: E402
from make_golden import save  # noqa: E402

MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)


def tv_chain(arr, R, train):
    img = Image.fromarray(arr).convert("RGB")
    w, h = img.size
    if train:                                        # Resize(R): shorter edge -> R, longer -> int(R * long / short)
        short, long_ = (w, h) if w <= h else (h, w)
        ns, nl = R, int(R * long_ / short)
        nw, nh = (ns, nl) if w <= h else (nl, ns)
        img = img.resize((nw, nh), Image.BICUBIC)
        top, left = int(round((nh - R) / 2.0)), int(round((nw - R) / 2.0))      # CenterCrop(R)
        img = img.crop((left, top, left + R, top + R))
    else:                                            # Resize((R, R))
        img = img.resize((R, R), Image.BICUBIC)
    u8 = np.asarray(img).copy()
    t = torch.from_numpy(u8).permute(2, 0, 1).contiguous().to(torch.float32).div(255)          # ToTensor
    t.sub_(torch.tensor(MEAN).view(-1, 1, 1)).div_(torch.tensor(STD).view(-1, 1, 1))            # Normalize
    return u8, t.numpy()


def gen():
    out = {"pillow": np.array(PIL.__version__)}
    for (h, w, R) in pu.CASES:
        arr = pu.image(h, w)
        for train in (True, False):
            tag = f"H{h}_W{w}_R{R}_{'train' if train else 'eval'}"
            u8, f = tv_chain(arr, R, train)
            out[f"{tag}_sha"] = np.array(hashlib.sha256(u8.tobytes()).hexdigest())
            out[f"{tag}_fsha"] = np.array(hashlib.sha256(f.tobytes()).hexdigest())
            if R < 224:
                out[f"{tag}_u8"] = u8
            if (h, w) in ((37, 53), (100, 75)):
                out[f"{tag}_f32"] = f
    save("preprocess.npz", **out)


if __name__ == "__main__":
    gen()

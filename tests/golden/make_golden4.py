#!/usr/bin/env python3
"""Fourth golden generator — SURVEY §8f "next" #1: runs the REFERENCE's BertAdam (model/base/optimization.py) for a few
steps on seeded tensors with the hyper-parameters its trainers use (train/DSPH/hash_train.py:35-41: two learning rates,
warmup_cosine, b2 = 0.98, weight_decay 0.2, max_grad_norm 1.0) plus a no-clip / no-decay / no-schedule group."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_stubs, ref_import, save  # noqa: E402

SHAPES = [(1,), (7,), (64, 3), (1000,), (129, 127), (16385,), (40001,)]   # 16383 / 16385: around one 16 Ki chunk
SLICE = 5      # tensors above 2000 elements are recorded every 5th element (the oracle is compared there, the GPU against the oracle everywhere)
STEPS = 4


def tensors(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(*s, generator=g) * (0.02 if i % 2 else 1.0) for i, s in enumerate(SHAPES)]


def grads(seed, step):
    g = torch.Generator().manual_seed(seed * 100 + step)
    # a mix of norms: some tensors get clipped (norm > 1), some do not
    return [torch.randn(*s, generator=g) * (1e-3 if i % 3 == 0 else 0.5) for i, s in enumerate(SHAPES)]


def gen():
    opt = ref_import("model.base.optimization")
    out = {"steps": np.int64(STEPS), "nshapes": np.int64(len(SHAPES))}
    for tag, kw in (("trainer", dict(lr=1e-3, warmup=0.1, schedule="warmup_cosine", b1=0.9, b2=0.98, e=1e-6, t_total=10,
                                     weight_decay=0.2, max_grad_norm=1.0)),
                    ("plain", dict(lr=5e-4, warmup=-1, schedule="warmup_linear", b1=0.9, b2=0.999, e=1e-6, t_total=-1,
                                   weight_decay=0.0, max_grad_norm=-1))):
        ps = [torch.nn.Parameter(t.clone()) for t in tensors(3)]
        half = len(ps) // 2
        o = opt.BertAdam([{"params": ps[:half], "lr": 1e-5}, {"params": ps[half:]}], **kw)
        for s in range(STEPS):
            for p, g in zip(ps, grads(3, s)):
                p.grad = g.clone()
            o.step()
            out[f"{tag}_lr_step{s}"] = np.array(o.get_lr(), np.float64)
        for i, p in enumerate(ps):
            st = o.state[p]
            cut = (lambda a: a.reshape(-1)[::SLICE].copy()) if p.numel() > 2000 else (lambda a: a.reshape(-1).copy())
            out[f"{tag}_p{i}"] = cut(p.detach().numpy())
            out[f"{tag}_m{i}"] = cut(st["next_m"].numpy())
            out[f"{tag}_v{i}"] = cut(st["next_v"].numpy())
            out[f"{tag}_g{i}"] = cut(p.grad.numpy())             # the clipped gradient of the last step
            assert st["step"] == STEPS
    save("adam.npz", **out)


if __name__ == "__main__":
    install_stubs()
    gen()

"""Shared by the BertAdam tests: the seeded tensors / gradients of tests/golden/make_golden4.py and an oracle driver."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden4 as mg  # noqa: E402  (only its seeded-input helpers; nothing of the reference is imported here)

CONFIGS = {
    "trainer": dict(lr=1e-3, warmup=0.1, schedule="warmup_cosine", b1=0.9, b2=0.98, e=1e-6, t_total=10, weight_decay=0.2,
                    max_grad_norm=1.0),
    "plain": dict(lr=5e-4, warmup=-1, schedule="warmup_linear", b1=0.9, b2=0.999, e=1e-6, t_total=-1, weight_decay=0.0,
                  max_grad_norm=-1),
}
GROUP0_LR = 1e-5        # the first half of the tensors sits in a group with its own lr (like clip_lr)


def run_oracle(tag):
    """-> lists p, g, m, v (float32 arrays) after mg.STEPS steps of oracle.adam_oracle on the seeded problem."""
    import oracle.adam_oracle as ao
    kw = CONFIGS[tag]
    ps = [t.numpy().copy() for t in mg.tensors(3)]
    ms = [np.zeros_like(p) for p in ps]
    vs = [np.zeros_like(p) for p in ps]
    gs = [None] * len(ps)
    half = len(ps) // 2
    for s in range(mg.STEPS):
        for i, g in enumerate(mg.grads(3, s)):
            lr = GROUP0_LR if i < half else kw["lr"]
            ps[i], gs[i], ms[i], vs[i] = ao.step(ps[i], g.numpy(), ms[i], vs[i], s, lr, kw["b1"], kw["b2"], kw["e"],
                                                 kw["weight_decay"], kw["max_grad_norm"], kw["t_total"], kw["warmup"],
                                                 kw["schedule"])
    return ps, gs, ms, vs


def cut(a):
    a = np.asarray(a).reshape(-1)
    return a[::mg.SLICE] if a.size > 2000 else a


def atol(ref):
    """a few ulps of the array's largest magnitude (see test_oracle_adam.py)"""
    return 4e-7 * float(np.abs(np.asarray(ref)).max())

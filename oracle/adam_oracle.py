"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy, fp32 arithmetic in the reference's operation order) of one step of the
reference's BertAdam for ONE parameter tensor (model/base/optimization.py:103-168).  Pinned by tests/golden/adam.npz, which
was produced by running the reference's own BertAdam (tests/golden/make_golden4.py)."""
import math

import numpy as np

f32 = np.float32


def schedule(name, x, warmup):
    """optimization.py:26-43"""
    if x < warmup:
        return x / warmup
    if name == "warmup_cosine":
        return 0.5 * (1.0 + math.cos(math.pi * x))
    if name == "warmup_constant":
        return 1.0
    return max((x - 1.) / (warmup - 1.), 0)


def step(p, g, m, v, step_no, lr, b1, b2, e, weight_decay, max_grad_norm, t_total=-1, warmup=-1, sched="warmup_linear"):
    """-> (p, g, m, v) after one step; inputs are float32 arrays (not modified)."""
    p, g, m, v = (np.array(a, f32) for a in (p, g, m, v))
    if max_grad_norm > 0:                                               # :135-136 clip_grad_norm_(p, max_grad_norm)
        total = f32(np.sqrt(np.sum(g.astype(np.float64) ** 2)))
        coef = f32(max_grad_norm) / (total + f32(1e-6))
        g = g * min(coef, f32(1.0))
    # :141 next_m.mul_(b1).add_(grad, alpha=1-b1): ATen's add(alpha) is a fused multiply-add (vec::fmadd), i.e. ONE rounding of
    # m*b1 (already rounded) + alpha*g; emulated with an exact f64 product (24x24 bits) and one f64 add
    m = ((m * f32(b1)).astype(np.float64) + g.astype(np.float64) * np.float64(f32(1.0 - b1))).astype(f32)
    v = v * f32(b2) + (f32(1.0 - b2) * g) * g                           # :143
    upd = m / (np.sqrt(v) + f32(e))                                     # :144
    if weight_decay > 0.0:
        upd = upd + f32(weight_decay) * p                               # :153-154
    lr_s = lr * schedule(sched, step_no / t_total, warmup) if t_total != -1 else lr   # :156-161 (python double)
    p = p + (-(f32(lr_s) * upd))                                        # :163-164
    return p, g, m, v

"""BertAdam — host mirror of the reference's model/base/optimization.py (same constructor, schedules, `get_lr`, and the
same per-parameter state keys `step` / `next_m` / `next_v`, so optimizer state dicts are interchangeable), with the whole
step executed by ONE fused multi-tensor HIP launch pair (cmh_bert_adam_step) instead of ~12 small kernels per parameter.
There is no CPU path: parameters must live on the GPU."""
import math

import torch
from torch.optim import Optimizer

import cmh_native as N

try:                      # torch.optim.optimizer.required went away in newer torch; the reference only uses it as a sentinel
    from torch.optim.optimizer import required
except ImportError:       # pragma: no cover
    required = object()


def warmup_cosine(x, warmup=0.002):          # optimization.py:26-29
    if x < warmup:
        return x / warmup
    return 0.5 * (1.0 + math.cos(math.pi * x))


def warmup_constant(x, warmup=0.002):        # :31-36
    if x < warmup:
        return x / warmup
    return 1.0


def warmup_linear(x, warmup=0.002):          # :38-43
    if x < warmup:
        return x / warmup
    return max((x - 1.) / (warmup - 1.), 0)


SCHEDULES = {"warmup_cosine": warmup_cosine, "warmup_constant": warmup_constant, "warmup_linear": warmup_linear}


class BertAdam(Optimizer):
    """BERT's Adam with decoupled weight decay and per-tensor gradient clipping (optimization.py:52-168)."""

    def __init__(self, params, lr=required, warmup=-1, t_total=-1, schedule="warmup_linear", b1=0.9, b2=0.999, e=1e-6,
                 weight_decay=0.01, max_grad_norm=1.0):
        if lr is not required and lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if schedule not in SCHEDULES:
            raise ValueError("Invalid schedule parameter: {}".format(schedule))
        if not 0.0 <= warmup < 1.0 and not warmup == -1:
            raise ValueError("Invalid warmup: {} - should be in [0.0, 1.0[ or -1".format(warmup))
        if not 0.0 <= b1 < 1.0:
            raise ValueError("Invalid b1 parameter: {} - should be in [0.0, 1.0[".format(b1))
        if not 0.0 <= b2 < 1.0:
            raise ValueError("Invalid b2 parameter: {} - should be in [0.0, 1.0[".format(b2))
        if not e >= 0.0:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(e))
        defaults = dict(lr=lr, schedule=schedule, warmup=warmup, t_total=t_total, b1=b1, b2=b2, e=e,
                        weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        super().__init__(params, defaults)

    @staticmethod
    def _scheduled(group, step):
        if group["t_total"] != -1:
            return group["lr"] * SCHEDULES[group["schedule"]](step / group["t_total"], group["warmup"])
        return group["lr"]

    def get_lr(self):
        lr = []
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                state = self.state[p]
                if len(state) == 0:
                    return [0]
                lr.append(self._scheduled(group, state["step"]))
        return lr

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # one fused launch per distinct (b1, b2, e); the reference's trainers use a single triple for every group
        batches = {}
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise N.NativeError("BertAdam: parameters and gradients must be contiguous float32 (model.float())")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = 0
                    state["next_m"] = torch.zeros_like(p)
                    state["next_v"] = torch.zeros_like(p)
                entry = (p, p.grad, state["next_m"], state["next_v"], self._scheduled(group, state["step"]),
                         group["weight_decay"], group["max_grad_norm"])
                batches.setdefault((group["b1"], group["b2"], group["e"]), []).append(entry)
                state["step"] += 1
        for (b1, b2, e), entries in batches.items():
            N.bert_adam_step(entries, b1, b2, e)
            # the kernel wrote through raw pointers: tell autograd / the weight caches (model/base/model.py::_key)
            torch._C._increment_version([t for p, g, *_ in entries for t in (p, g)])
            for p, *_ in entries:       # bf16 GEMM copies rewritten by the kernel are current for the new version
                if getattr(p, "_cmh_bf16", None) is not None:
                    p._cmh_bf16_version = (p.data_ptr(), p._version)
        return loss

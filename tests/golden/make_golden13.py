#!/usr/bin/env python3
"""Thirteenth golden generator — gradients of the MITH step loss: the REFERENCE's MITHTrainer.compute_loss
(train/MITH/hash_train.py:149-201, all five groups summed as train_epoch does) differentiated by torch autograd on the CPU
w.r.t. the eight HashingModel outputs.  Inputs: tests/mithutil.py::loss_inputs (the cases of make_golden3.py)."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import mithutil as mu  # noqa: E402
from make_golden import install_stubs, ref_import, save, t  # noqa: E402


def gen():
    tr = ref_import("train.MITH.hash_train").MITHTrainer
    out = {}
    for (Nb, K, C, Mb) in [(8, 16, 24, 50), (16, 64, 80, 200)]:
        tag = f"loss_N{Nb}_K{K}"
        od, banks, label, train_labels = mu.loss_inputs(Nb, K, C, Mb)
        self = SimpleNamespace(args=SimpleNamespace(**mu.HP), rank="cpu", k_bits=K, train_labels=t(train_labels),
                               img_buffer_tokens=t(banks["img_tokens"]), img_buffer_cls=t(banks["img_cls"]),
                               txt_buffer_tokens=t(banks["txt_tokens"]), txt_buffer_cls=t(banks["txt_cls"]))
        for name in ("bayesian_loss", "info_nce_loss", "info_nce_loss_bmm", "quantization_loss_2"):
            setattr(self, name, (lambda n: (lambda *a, **k: getattr(tr, n)(self, *a, **k)))(name))
        tod = {k: t(v).requires_grad_() for k, v in od.items()}
        lam = mu.HP["hyper_lambda"]
        with torch.no_grad():
            Bc = torch.sign((tod["img_cls_hash"] * lam + tod["img_tokens_hash"] * (1 - lam)) +
                            (tod["txt_cls_hash"] * lam + tod["txt_tokens_hash"] * (1 - lam)))
        LD = tr.compute_loss(self, tod, t(label), Bc)
        total = sum(LD.values())
        total.backward()
        out[f"{tag}_total"] = total.detach().numpy()
        for k, v in tod.items():
            gk = v.grad.numpy()
            out[f"{tag}_d_{k}"] = gk[::4, :, ::4] if k.startswith("trans_tokens") else gk       # keep the fixture small
    save("mith_loss_grads.npz", **out)


if __name__ == "__main__":
    install_stubs()
    gen()

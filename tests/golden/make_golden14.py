#!/usr/bin/env python3
"""Fourteenth golden generator — the loss kernels at the REAL sizes of BASELINE configs[2] / configs[3] (round-1 goldens stopped at
miniature sizes): the REFERENCE's own code on the CPU,
  * MITHTrainer.compute_loss (train/MITH/hash_train.py:149-201) at batch 256 x 64 bit x 80 classes against the 10 000-row memory
    bank: the summed step loss and eight sampled gradient rows of every HashingModel output;
  * DNPH_out + the trainer's noise term (train/DNPH_TOMM/loss.py:14-32, hash_train.py:65-81) at 2B = 512 rows, 128 bit, 21 classes:
    the step loss, eight sampled gradient rows per input and the full proxy gradient.
Inputs come from the seeded recipes (tests/mithutil.py, tests/heads2util.py); the fixture keeps outputs only (~100 KiB)."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import heads2util as hu  # noqa: E402
import mithutil as mu  # noqa: E402
import recipe  # noqa: E402
from make_golden import install_stubs, ref_import, save, t  # noqa: E402

ROWS = [0, 37, 64, 101, 128, 190, 222, 255]


def gen():
    out = {}
    tr = ref_import("train.MITH.hash_train").MITHTrainer
    Nb, K, C, Mb = 256, 64, 80, 10000
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
    out["mith_total"] = total.detach().numpy()
    for k, v in LD.items():
        out[f"mith_group_{k}"] = v.detach().numpy()
    for k, v in tod.items():
        g = v.grad.numpy()
        out[f"mith_d_{k}"] = g[::8, ROWS, ::16] if k.startswith("trans_tokens") else g[ROWS]

    loss_mod = ref_import("train.DNPH_TOMM.loss")
    B, K, C = 256, 128, 21
    c = hu.dnph_case(B, K, C)
    m = loss_mod.DNPH_out(SimpleNamespace(nclass=C, output_dim=K))
    m.proxies.data.copy_(t(c["prox"]))
    hi, ht, pi, pt = (t(c[k]).requires_grad_() for k in ("hi", "ht", "pi", "pt"))
    lab = t(c["lab"])
    ni, nt = (t(a) for a in hu.dnph_noise(B, K))
    noise_loss = hi.mul(ni).sum(dim=-1).mean() + ht.mul(nt).sum(dim=-1).mean()
    loss = m(hi, ht, pi, pt, lab, lab) - 0.1 * noise_loss
    loss.backward()
    out["dnph_step_loss"] = loss.detach().numpy()
    for name, v in (("ghi", hi), ("ght", ht), ("gpi", pi), ("gpt", pt)):
        out[f"dnph_{name}"] = v.grad.numpy()[ROWS]
    out["dnph_gprox"] = m.proxies.grad.numpy()
    out["rows"] = np.array(ROWS)
    save("real_size.npz", **out)


if __name__ == "__main__":
    import tempfile
    install_stubs()
    os.chdir(tempfile.mkdtemp())
    torch.manual_seed(0)
    gen()

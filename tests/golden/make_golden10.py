#!/usr/bin/env python3
"""Tenth golden generator — gradients of the TwDH step: the REFERENCE's two ModalityHash heads (model/TwDH.py, image head with
BatchNorm1d in training mode, text head with LayerNorm), softmax_hash short codes and TwDHTrainer.compute_loss
(train/TwDH/hash_train.py:124-163, quan_alpha 0.5, low_rate 0.3) differentiated by torch autograd on the CPU w.r.t. the CLIP
features and every head parameter.  Inputs and the random centres are those of make_golden2.py (heads2util.twdh_case)."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import heads2util as hu  # noqa: E402
from make_golden import install_stubs, ref_import, save, t  # noqa: E402


def gen():
    mod = ref_import("model.TwDH")
    tr = ref_import("train.TwDH.hash_train").TwDHTrainer
    out = {}
    for (B, K, S, C) in hu.TWDH_CASES:
        c = hu.twdh_case(B, K, S, C)
        tag = c["tag"]
        heads, feats = {}, {}
        for side, ln in (("img", False), ("txt", True)):
            h = mod.ModalityHash(inputDim=512, outputDim=K, layernorm=ln, num_heads=8, batch_first=True, hash_func="softmax")
            p = c["p_" + side]
            h.atten.in_proj_weight.data.copy_(t(p["in_w"])); h.atten.in_proj_bias.data.copy_(t(p["in_b"]))
            h.atten.out_proj.weight.data.copy_(t(p["out_w"])); h.atten.out_proj.bias.data.copy_(t(p["out_b"]))
            h.norm.weight.data.copy_(t(p["norm_w"])); h.norm.bias.data.copy_(t(p["norm_b"]))
            h.fc2.weight.data.copy_(t(p["fc2_w"])); h.fc2.bias.data.copy_(t(p["fc2_b"]))
            h.train()
            heads[side] = h
            feats[side] = t(c["feat_i" if side == "img" else "feat_t"]).requires_grad_()
        long_i, long_t = heads["img"](feats["img"]), heads["txt"](feats["txt"])
        trans = t(c["trans"])
        short_i, short_t = mod.softmax_hash(long_i.matmul(trans)), mod.softmax_hash(long_t.matmul(trans))
        self = SimpleNamespace(args=SimpleNamespace(quan_alpha=0.5, low_rate=0.3), criterion=torch.nn.BCELoss())
        self.hash_convert = lambda x: tr.hash_convert(self, x)
        self.soft_argmax_hash_loss = lambda code: tr.soft_argmax_hash_loss(self, code)

        def hcm(lab, center, _key=[0]):
            torch.manual_seed(1000 + _key[0])          # the random centres of make_golden2.py (twdh.npz: rc_long / rc_short)
            _key[0] += 1
            return tr.hash_center_multilables(self, lab, center)
        self.hash_center_multilables = hcm
        loss = tr.compute_loss(self, long_i, long_t, {str(S): short_i}, {str(S): short_t}, t(c["labels"]), None, t(c["lc"]),
                               {str(S): t(c["sc"])})
        loss.backward()
        out[f"{tag}_loss"] = loss.detach().numpy()
        for side in ("img", "txt"):
            h = heads[side]
            out[f"{tag}_{side}_gfeat"] = feats[side].grad.numpy()
            for name, prm in (("in_w", h.atten.in_proj_weight), ("in_b", h.atten.in_proj_bias), ("out_w", h.atten.out_proj.weight),
                              ("out_b", h.atten.out_proj.bias), ("norm_w", h.norm.weight), ("norm_b", h.norm.bias),
                              ("fc2_w", h.fc2.weight), ("fc2_b", h.fc2.bias)):
                g = prm.grad.numpy()
                if name == "in_w":                      # [3d, d]: keep the V block and a proof that the Q / K blocks are zero
                    out[f"{tag}_{side}_g_in_w_qk_absmax"] = np.abs(g[:1024]).max()
                    g = g[1024:]
                if g.size > 20000:                      # keep the fixture small: every 32nd row of the big matrices
                    g = g[::32]
                out[f"{tag}_{side}_g_{name}"] = g
            if side == "img":
                out[f"{tag}_img_running_mean"] = h.norm.running_mean.numpy().copy()
                out[f"{tag}_img_running_var"] = h.norm.running_var.numpy().copy()
    save("twdh_grads.npz", **out)


if __name__ == "__main__":
    import tempfile
    install_stubs()
    os.chdir(tempfile.mkdtemp())
    torch.manual_seed(0)
    gen()

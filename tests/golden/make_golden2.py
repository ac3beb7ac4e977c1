#!/usr/bin/env python3
"""Second golden generator (TwDH + DNPH rows of SURVEY §8a): runs the REFERENCE's own modules
(model/TwDH.py ModalityHash/softmax_hash, train/TwDH/hash_train.py trainer methods driven unbound,
train/DNPH_TOMM/loss.py DNPH_out, train/DNPH_TOMM/b_reg.py gene_noise) on the seeded recipes.
Same rules as make_golden.py: build container only, outputs only."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import recipe  # noqa: E402
from make_golden import install_stubs, ref_import, save, t  # noqa: E402


def twdh_head_params(K, seed, side):
    d = 512
    p = {}
    p["in_w"] = (recipe.features(3 * d, d, seed, f"twdh_{side}_inw_{K}") * 0.1).astype(np.float32)
    p["in_b"] = (recipe.features(1, 3 * d, seed, f"twdh_{side}_inb_{K}")[0] * 0.1).astype(np.float32)
    p["out_w"] = (recipe.features(d, d, seed, f"twdh_{side}_ow_{K}") * 0.1).astype(np.float32)
    p["out_b"] = (recipe.features(1, d, seed, f"twdh_{side}_ob_{K}")[0] * 0.1).astype(np.float32)
    p["norm_w"] = (1 + 0.2 * recipe.features(1, d, seed, f"twdh_{side}_nw_{K}")[0]).astype(np.float32)
    p["norm_b"] = (0.1 * recipe.features(1, d, seed, f"twdh_{side}_nb_{K}")[0]).astype(np.float32)
    p["fc2_w"], p["fc2_b"] = recipe.head_linear(d, 2 * K, seed, f"twdh_{side}_fc2_{K}")
    return p


def pm1(shape, seed, tag):
    return recipe.sign_codes(shape[0], shape[1], seed, tag)


def gen_twdh():
    mod = ref_import("model.TwDH")
    tr = ref_import("train.TwDH.hash_train").TwDHTrainer
    out, seed = {}, 41
    for (B, K, S, C) in [(12, 16, 8, 24), (32, 128, 16, 21)]:
        tag = f"B{B}_K{K}"
        feat_i = recipe.features(B, 512, seed, f"twdh_fi_{tag}")
        feat_t = recipe.features(B, 512, seed, f"twdh_ft_{tag}")
        hashes = {}
        for side, ln, feat in (("img", False, feat_i), ("txt", True, feat_t)):
            h = mod.ModalityHash(inputDim=512, outputDim=K, layernorm=ln, num_heads=8, batch_first=True, hash_func="softmax")
            p = twdh_head_params(K, seed, side)
            h.atten.in_proj_weight.data.copy_(t(p["in_w"])); h.atten.in_proj_bias.data.copy_(t(p["in_b"]))
            h.atten.out_proj.weight.data.copy_(t(p["out_w"])); h.atten.out_proj.bias.data.copy_(t(p["out_b"]))
            h.norm.weight.data.copy_(t(p["norm_w"])); h.norm.bias.data.copy_(t(p["norm_b"]))
            h.fc2.weight.data.copy_(t(p["fc2_w"])); h.fc2.bias.data.copy_(t(p["fc2_b"]))
            h.train()                       # BatchNorm1d uses batch statistics (the head is never put in eval upstream)
            with torch.no_grad():
                hashes[side] = h(t(feat))
            out[f"{tag}_{side}_long"] = hashes[side].numpy()
        trans = (recipe.features(2 * K, 2 * S, seed, f"twdh_trans_{tag}") * (2 * K) ** -0.5 * 4).astype(np.float32)
        with torch.no_grad():
            short_i = mod.softmax_hash(hashes["img"].matmul(t(trans)))
            short_t = mod.softmax_hash(hashes["txt"].matmul(t(trans)))
        out[f"{tag}_img_short"], out[f"{tag}_txt_short"] = short_i.numpy(), short_t.numpy()
        # targets + loss through the trainer's own methods (unbound, SimpleNamespace self)
        labels = recipe.labels(B, C, seed, p=0.12, tag=f"twdh_lab_{tag}")
        labels[0] = 0                                             # a row without labels
        lc, sc = pm1((C, K), seed, f"twdh_lc_{tag}"), pm1((C, S), seed, f"twdh_sc_{tag}")
        self = SimpleNamespace(args=SimpleNamespace(quan_alpha=0.5, low_rate=0.3), criterion=torch.nn.BCELoss())
        self.hash_convert = lambda x: tr.hash_convert(self, x)
        self.soft_argmax_hash_loss = lambda c: tr.soft_argmax_hash_loss(self, c)
        rcs = {}

        def hcm(lab, center, _key=[0]):
            torch.manual_seed(1000 + _key[0])
            rc = torch.randint_like(center[0], 2)
            rc[rc == 0] = -1
            rcs[_key[0]] = rc.numpy().copy()
            torch.manual_seed(1000 + _key[0])
            _key[0] += 1
            return tr.hash_center_multilables(self, lab, center)
        self.hash_center_multilables = hcm
        # rows without labels make the reference index with an empty mean -> NaN; keep them (they are part of the contract)
        loss = tr.compute_loss(self, hashes["img"], hashes["txt"], {str(S): short_i}, {str(S): short_t}, t(labels), None,
                               t(lc), {str(S): t(sc)})
        out[f"{tag}_loss"] = loss.numpy()
        out[f"{tag}_rc_long"], out[f"{tag}_rc_short"] = rcs[0], rcs[1]
        torch.manual_seed(1000)
        out[f"{tag}_target_long"] = np.nan_to_num(tr.hash_center_multilables(self_plain(tr), t(labels), t(lc)).numpy(), nan=-1.0)
        code = tr.make_hash_code(None, hashes["img"])
        out[f"{tag}_img_code"] = code.numpy()
        out[f"{tag}_meta"] = np.array([B, K, S, C])
    save("twdh.npz", **out)


def self_plain(tr):
    return SimpleNamespace()


def gen_dnph():
    loss_mod = ref_import("train.DNPH_TOMM.loss")
    breg = ref_import("train.DNPH_TOMM.b_reg")
    out, seed = {}, 51
    for (B, K, C) in [(16, 16, 21), (40, 128, 24)]:
        tag = f"B{B}_K{K}_C{C}"
        m = loss_mod.DNPH_out(SimpleNamespace(nclass=C, output_dim=K))
        prox = (recipe.features(C, K, seed, f"dnph_prox_{tag}") / 4).astype(np.float32)
        m.proxies.data.copy_(t(prox))
        hi = np.tanh(recipe.features(B, K, seed, f"dnph_hi_{tag}")); ht = np.tanh(recipe.features(B, K, seed, f"dnph_ht_{tag}"))
        pi = recipe.features(B, C, seed, f"dnph_pi_{tag}"); pt = recipe.features(B, C, seed, f"dnph_pt_{tag}")
        lab = recipe.labels(B, C, seed, p=0.15, tag=f"dnph_lab_{tag}")
        with torch.no_grad():
            loss1 = m(t(hi), t(ht), t(pi), t(pt), t(lab), t(lab))
        np.random.seed(7)
        s_vec = breg.rand_unit_rect(B, K)
        ni, nt = breg.gene_noise(hi, s_vec), breg.gene_noise(ht, s_vec)
        noise = (t(hi).mul(t(ni.astype(np.float32))).sum(-1).mean() + t(ht).mul(t(nt.astype(np.float32))).sum(-1).mean())
        out[f"{tag}_loss1"] = loss1.numpy()
        out[f"{tag}_step_loss"] = (loss1 - 0.1 * noise).numpy()
        out[f"{tag}_s_vec"] = s_vec.astype(np.int8)
        out[f"{tag}_noise_i"], out[f"{tag}_noise_t"] = ni.astype(np.int8), nt.astype(np.int8)
    save("dnph.npz", **out)


if __name__ == "__main__":
    import tempfile
    install_stubs()
    os.chdir(tempfile.mkdtemp())
    torch.manual_seed(0)
    gen_twdh()
    gen_dnph()

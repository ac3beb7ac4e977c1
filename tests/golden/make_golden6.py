#!/usr/bin/env python3
"""Sixth golden generator — BASELINE configs[0] in miniature: the REFERENCE's DCHMT training loop body
(train/DCHMT/hash_train.py:44-68: model forward, torch.cat of the select head's pairs, our_loss, BertAdam step) run for a few
steps on the CPU with the reference's own CLIP, HashLayer, similarity_loss / our_loss and BertAdam.  Records the loss of every
step and the parameters after the last one."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import dchmtutil as du  # noqa: E402
from make_golden import build_ref_clip, install_stubs, ref_import, save  # noqa: E402


def gen():
    clip = build_ref_clip(du.CFG, du.SEED)
    HashLayer = ref_import("model.DCHMT").HashLayer
    tr = ref_import("train.DCHMT.hash_train").DCHMTTrainer
    BertAdam = ref_import("model.base.optimization").BertAdam
    ih, th = HashLayer(du.CFG["embed_dim"], du.K), HashLayer(du.CFG["embed_dim"], du.K)
    du.fill_head(ih, 1)
    du.fill_head(th, 2)
    clip_params = [p for n, p in clip.named_parameters()]
    opt = BertAdam([{"params": clip_params, "lr": du.CLIP_LR}, {"params": ih.parameters(), "lr": du.OPT["lr"]},
                    {"params": th.parameters(), "lr": du.OPT["lr"]}], **du.OPT)
    args = SimpleNamespace(output_dim=du.K, hash_layer="select", display_step=10 ** 9, epochs=1, **du.LOSS)
    self = SimpleNamespace(args=args, rank="cpu", global_step=1, logger=None)
    self.similarity_loss = lambda a, b, s, self=self: tr.similarity_loss(self, a, b, s)
    out = {}
    losses = []
    for step in range(du.STEPS):
        img, txt, lab = du.batch(step)
        hi = torch.cat(ih(clip.encode_image(img)), dim=-1)
        ht = torch.cat(th(clip.encode_text(txt)), dim=-1)
        loss = tr.our_loss(self, hi, ht, lab, 0, step + 1)
        losses.append(float(loss))
        opt.zero_grad()
        loss.backward()
        opt.step()
    out["losses"] = np.array(losses, np.float64)
    for name, p in list(clip.named_parameters()) + [("image_hash." + n, p) for n, p in ih.named_parameters()] + \
            [("text_hash." + n, p) for n, p in th.named_parameters()]:
        if name.startswith("image_hash.hash_list") or name.startswith("text_hash.hash_list"):
            if int(name.split(".")[2]) > 1:
                continue                      # two of the K two-way Linears are enough
        out["p_" + name] = du.cut(p.detach().numpy())
    save("dchmt_traj.npz", **out)
    print("losses", losses)


if __name__ == "__main__":
    install_stubs()
    gen()

#!/usr/bin/env python3
"""Third golden generator — MITH rows of SURVEY §8a (a5, a8, a13): runs the REFERENCE's model/MITH.py (CLIP1 trunk via its
own build_model, HashingModel) and train/MITH/hash_train.py::compute_loss (unbound) on seeded inputs."""
import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import recipe  # noqa: E402
import mithutil as mu  # noqa: E402
from make_golden import install_stubs, ref_import, save, t  # noqa: E402


def gen():
    mm = ref_import("model.MITH")
    out = {}
    # ---- trunk with tokens (a5): tiny CLIP, embed 512, through the reference's own build_model (fp16 round trip)
    cfg, seed = mu.CLIP_TINY512, 7
    sd = {k: t(v) for k, v in recipe.clip_state_dict(cfg, seed).items()}
    clip = mm.build_model(sd).float()
    image = t(recipe.images(3, cfg["image_resolution"], seed))
    text = recipe.captions(3, 16, cfg["vocab_size"], seed)
    text[text == cfg["vocab_size"] - 1] = 49407 % cfg["vocab_size"] if False else cfg["vocab_size"] - 1
    kpm = (text == 0)
    with torch.no_grad():
        seq_i, _, cls_i = clip.encode_image(image)
        seq_t, _, new_kpm, eos_t = clip.encode_text(t(text), t(kpm))
    out.update(trunk_seq_i=seq_i.numpy(), trunk_cls_i=cls_i.numpy(), trunk_seq_t=seq_t.numpy(), trunk_eos_t=eos_t.numpy())

    # ---- HashingModel (a8)
    for (Nb, L, K) in [(3, 12, 16), (4, 32, 64)]:
        tag = f"N{Nb}_L{L}_K{K}"
        args = SimpleNamespace(output_dim=K, **mu.ARGS)
        hm = mm.HashingModel(clip_embed_dim=512, args=args)
        shapes = {k: tuple(v.shape) for k, v in hm.state_dict().items()}
        st = mu.fill_state(shapes, 100 + K)
        full = {k: (t(st[k]) if k in st else v) for k, v in hm.state_dict().items()}
        hm.load_state_dict(full)
        hm.eval()
        c = mu.hash_inputs(Nb, L, K)
        with torch.no_grad():
            od = hm(t(c["img_tokens"]), t(c["txt_tokens"]), t(c["img_cls"]), t(c["txt_eos"]), t(c["kpm"]))
        for k, v in od.items():
            v = v.numpy()
            out[f"{tag}_{k}"] = v[::4] if (k.startswith("trans_tokens") and K == 64) else v   # keep the fixture small
        out[f"{tag}_keys"] = np.array(sorted(shapes))

    # ---- losses (a13)
    tr = ref_import("train.MITH.hash_train").MITHTrainer
    for (Nb, K, C, Mb) in [(8, 16, 24, 50), (16, 64, 80, 200)]:
        tag = f"loss_N{Nb}_K{K}"
        od, banks, label, train_labels = mu.loss_inputs(Nb, K, C, Mb)
        self = SimpleNamespace(args=SimpleNamespace(**mu.HP), rank="cpu", k_bits=K, train_labels=t(train_labels),
                               img_buffer_tokens=t(banks["img_tokens"]), img_buffer_cls=t(banks["img_cls"]),
                               txt_buffer_tokens=t(banks["txt_tokens"]), txt_buffer_cls=t(banks["txt_cls"]))
        for name in ("bayesian_loss", "info_nce_loss", "info_nce_loss_bmm", "quantization_loss_2"):
            setattr(self, name, (lambda n: (lambda *a, **k: getattr(tr, n)(self, *a, **k)))(name))
        tod = {k: t(v) for k, v in od.items()}
        lam = mu.HP["hyper_lambda"]
        Bc = torch.sign((tod["img_cls_hash"] * lam + tod["img_tokens_hash"] * (1 - lam)) +
                        (tod["txt_cls_hash"] * lam + tod["txt_tokens_hash"] * (1 - lam)))
        with torch.no_grad():
            LD = tr.compute_loss(self, tod, t(label), Bc)
        for k, v in LD.items():
            out[f"{tag}_{k}"] = np.float32(v)
        out[f"{tag}_B"] = Bc.numpy()
    save("mith.npz", **out)


if __name__ == "__main__":
    install_stubs()
    os.chdir(tempfile.mkdtemp())
    torch.manual_seed(0)
    gen()

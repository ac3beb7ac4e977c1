"""Host-side mirror of the reference's plugin surface (no GPU needed)."""
import argparse
import sys

import numpy as np
import pytest
import torch

import recipe


def _tiny_state():
    return {k: torch.from_numpy(v) for k, v in recipe.clip_state_dict(recipe.CLIP_TINY, 7).items()}


def test_build_model_state_dict_contract():
    from model.base.model import build_model
    m = build_model(_tiny_state())
    keys = set(m.state_dict().keys())
    assert keys == set(recipe.clip_state_shapes(recipe.CLIP_TINY).keys())
    for k, shape in recipe.clip_state_shapes(recipe.CLIP_TINY).items():
        assert tuple(m.state_dict()[k].shape) == tuple(shape), k
    # convert_weights semantics: GEMM weights fp16 until .float(), LN/embeddings fp32 (reference :391-412)
    assert m.visual.conv1.weight.dtype == torch.float16
    assert m.transformer.resblocks[0].attn.in_proj_weight.dtype == torch.float16
    assert m.ln_final.weight.dtype == torch.float32 and m.token_embedding.weight.dtype == torch.float32
    m.float()
    ref = recipe.clip_state_dict(recipe.CLIP_TINY, 7, fp16_roundtrip=True)
    for k in ("visual.conv1.weight", "transformer.resblocks.1.mlp.c_fc.bias", "ln_final.weight", "text_projection"):
        np.testing.assert_array_equal(m.state_dict()[k].numpy(), ref[k])


def test_vitb32_has_302_tensors():
    assert len(recipe.clip_state_shapes(recipe.CLIP_VITB32)) == 302


def test_baseclip_api_and_state_dict_keys(tmp_path):
    from model.DSPH import MDSPH
    from model.DCHMT import MDCMHT
    m = MDSPH(outputDim=16, clipPath=_tiny_state(), saveDir=str(tmp_path))
    assert m.embedDim == 64
    for k in ("image_hash.fc.weight", "image_hash.fc.bias", "text_hash.fc.weight", "text_hash.fc.bias",
              "clip.visual.proj", "clip.token_embedding.weight"):
        assert k in m.state_dict()
    assert m.eval() is None and m.train() is None              # reference quirk
    m.eval()
    assert not m.image_hash.training and m.clip.training
    d = MDCMHT(outputDim=8, clipPath=_tiny_state(), saveDir=str(tmp_path))
    assert len(d.image_hash.hash_list) == 8 and "image_hash.hash_list.7.weight" in d.state_dict()
    # checkpoints round-trip through torch.save / load_clip(path)
    p = tmp_path / "clip.pt"
    torch.save(_tiny_state(), p)
    m2 = MDSPH(outputDim=16, clipPath=str(p), saveDir=str(tmp_path))
    assert m2.clip.visual.proj.shape == (128, 64)


def test_registry_is_lazy_and_cli_tolerant(monkeypatch):
    import main
    assert main.trainers.get("nope") is None
    t = main.trainers.get("DSPH")
    assert t.__name__ == "DSPHTrainer"
    with pytest.raises(NotImplementedError):
        main.trainers["DScPH"]
    for m in ("DCHMT", "TwDH", "DNPH", "MITH", "DNpH", "DMsH_LN", "DHaPH"):
        assert main.trainers[m].__name__.endswith("Trainer")
    # both parsers tolerate each other's flags
    monkeypatch.setattr(sys, "argv", ["main.py", "--method", "DSPH", "--dataset", "synthetic", "--output-dim", "64",
                                      "--batch-size", "8", "--alpha", "0.5"])
    from train.DSPH.get_args import get_args
    a = get_args(argparse.Namespace(method="DSPH", dataset="synthetic", output_dim=64, is_train=True))
    assert a.batch_size == 8 and a.alpha == 0.5 and a.max_words == 32 and a.query_num == 5000
    assert a.save_dir.endswith("DSPH/synthetic/64")


def test_hyp_proxies_seeded_like_reference(golden):
    import hashlib
    from train.DSPH.loss import HyP
    g = golden("loss_dsph.npz")
    for (B, K, C) in [(32, 64, 24), (48, 16, 80)]:
        h = HyP(numclass=C, output_dim=K, hypseed=0, alpha=0.8)
        tag = f"B{B}_K{K}_C{C}"
        assert hashlib.sha256(h.proxies.detach().numpy().tobytes()).hexdigest() == str(g[f"{tag}_default_proxies_sha"])
        assert h.threshold == float(g[f"{tag}_threshold"])


def test_synthetic_split_rule():
    from dataset.synthetic import dataloader
    tr, q, r = dataloader(total=300, nclass=24, maxWords=32, imageResolution=64, query_num=50, train_num=100, seed=3)
    assert len(q) == 50 and len(tr) == 100 and len(r) == 250
    img, cap, lab, idx = r[0]
    assert img.shape == (3, 64, 64) and cap.shape == (32,) and cap[0] == 49406 and cap.max() == 49407
    assert lab.shape == (24,) and r.get_all_label().shape == (250, 24)


def test_dataset_files_and_caption_layouts(tmp_path):
    """dataset/dataloader.py on the three caption layouts (string matrix, caption.txt, MATLAB cell array), the reference's split
    rule and the label / index plumbing — host logic only, no GPU."""
    import random
    import scipy.io as scio
    from dataset.base import BaseDataset
    from dataset.dataloader import load_files, split_data
    n = 12
    paths = np.array([f"/data/im{i:02d}.jpg" for i in range(n)] + ["/data/long_name_image.jpg"])[:n]
    scio.savemat(tmp_path / "index.mat", {"index": paths})
    lab = (np.arange(n * 5).reshape(n, 5) % 3 == 0).astype(np.int8)
    scio.savemat(tmp_path / "label.mat", {"category": lab})
    mat = np.array([[f"caption {i} a", f"second caption of {i}"] for i in range(n)])
    scio.savemat(tmp_path / "caption.mat", {"caption": mat})
    caps, idx, labels = load_files(str(tmp_path / "caption.mat"), str(tmp_path / "index.mat"), str(tmp_path / "label.mat"))
    assert caps.shape == (n, 2) and idx.shape == (n,) and labels.shape == (n, 5)
    assert idx[3].strip() == "/data/im03.jpg"
    ds = BaseDataset(caps, idx, labels, tokenizer=object())
    random.seed(0)
    assert {ds._choose_caption(4).strip() for _ in range(40)} == {"caption 4 a", "second caption of 4"}
    assert torch.equal(ds.get_all_label(), torch.from_numpy(lab).float())
    (tmp_path / "caption.txt").write_text("".join("only caption %d\n" % i for i in range(n)))
    caps_t, _, _ = load_files(str(tmp_path / "caption.txt"), str(tmp_path / "index.mat"), str(tmp_path / "label.mat"))
    assert caps_t.shape == (n, 1) and BaseDataset(caps_t, idx, labels, tokenizer=object())._choose_caption(7) == "only caption 7"
    cell = np.empty((1, n), dtype=object)
    for i in range(n):
        cell[0, i] = np.array([f"cell {i} x", f"cell {i} y", f"cell {i} z"][: 1 + i % 3], dtype=object)
    scio.savemat(tmp_path / "cell.mat", {"caption": cell})
    caps_c, _, _ = load_files(str(tmp_path / "cell.mat"), str(tmp_path / "index.mat"), str(tmp_path / "label.mat"))
    dc = BaseDataset(caps_c, idx, labels, tokenizer=object())
    assert dc._caption_list(caps_c[5]) == ["cell 5 x", "cell 5 y", "cell 5 z"] and dc._caption_list(caps_c[3]) == ["cell 3 x"]
    (qi, ti, ri), (qc, tc, rc), (ql, tl, rl) = split_data(caps, idx, labels, query_num=3, train_num=4, seed=9)
    np.random.seed(9)
    order = np.random.permutation(range(n))
    assert list(qi) == list(idx[order[:3]]) and list(ti) == list(idx[order[3:7]]) and list(ri) == list(idx[order[3:]])
    with pytest.raises(ValueError):
        load_files(str(tmp_path / "caption.json"), str(tmp_path / "index.mat"), str(tmp_path / "label.mat"))

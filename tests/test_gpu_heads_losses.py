"""Hash heads, code generation, Baseclip API and the fused losses on the GPU vs reference goldens / oracle."""
import numpy as np
import pytest
import torch

import recipe
from oracle import clip_oracle as co

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = dict(rtol=1e-4, atol=1e-4)


def _state(seed=7):
    return {k: torch.from_numpy(v) for k, v in recipe.clip_state_dict(recipe.CLIP_TINY, seed).items()}


@pytest.mark.parametrize("K", [16, 64])
def test_baseclip_dsph_and_dchmt_match_reference(golden, tmp_path, K):
    from model.DCHMT import MDCMHT
    from model.DSPH import MDSPH
    from train.base import TrainBase
    g = golden("baseclip_tiny.npz")
    cfg, seed = recipe.CLIP_TINY, int(g["seed"])
    image = torch.from_numpy(recipe.images(3, cfg["image_resolution"], seed)).to(DEV)
    text = torch.from_numpy(recipe.captions(3, 16, cfg["vocab_size"], seed)).to(DEV)
    import cmh_native as N

    m = MDSPH(outputDim=K, clipPath=_state(seed), saveDir=str(tmp_path)).to(DEV)
    m.float()
    for side in ("image", "text"):
        w, b = recipe.head_linear(cfg["embed_dim"], K, seed, f"dsph_{side}_{K}")
        getattr(m, f"{side}_hash").fc.weight.data.copy_(torch.from_numpy(w))
        getattr(m, f"{side}_hash").fc.bias.data.copy_(torch.from_numpy(b))
    m.eval()
    with torch.no_grad():
        hi, ht = m(image, text)
        np.testing.assert_allclose(m.clip.encode_image(image).cpu().numpy(), g["feat_img_fp16w"], **TOL)
    for h, s in ((hi, "img"), (ht, "txt")):
        ref = g[f"dsph_{s}_K{K}"]
        np.testing.assert_allclose(h.cpu().numpy(), ref, **TOL)
        safe = np.abs(ref) > 1e-3                      # sign() is only pinned away from 0
        assert np.array_equal(N.sign_codes(h).cpu().numpy()[safe], g[f"dsph_{s}_code_K{K}"][safe])

    d = MDCMHT(outputDim=K, clipPath=_state(seed), saveDir=str(tmp_path)).to(DEV)
    d.float()
    for side in ("image", "text"):
        hl = getattr(d, f"{side}_hash")
        w, b = recipe.head_linear(cfg["embed_dim"], 128, seed, f"dchmt_{side}_fc_{K}")
        hl.fc.weight.data.copy_(torch.from_numpy(w)); hl.fc.bias.data.copy_(torch.from_numpy(b))
        w2, b2 = recipe.head_linear(128, 2 * K, seed, f"dchmt_{side}_bits_{K}")
        for j, lin in enumerate(hl.hash_list):
            lin.weight.data.copy_(torch.from_numpy(w2[2 * j:2 * j + 2])); lin.bias.data.copy_(torch.from_numpy(b2[2 * j:2 * j + 2]))
    d.eval()
    with torch.no_grad():
        li, lt = d(image, text)
    assert isinstance(li, list) and len(li) == K and li[0].shape == (3, 2)
    for l, s in ((li, "img"), (lt, "txt")):
        ref = g[f"dchmt_{s}_K{K}"]
        np.testing.assert_allclose(torch.stack(l, 1).cpu().numpy(), ref, **TOL)
        codes = TrainBase.make_hash_code_DCHMT(None, l).cpu().numpy()
        safe = np.abs(ref[..., 0] - ref[..., 1]) > 1e-3
        assert np.array_equal(codes[safe], g[f"dchmt_{s}_code_K{K}"][safe])


def test_dnph_heads_match_reference(golden, tmp_path):
    from model.DNPH_TOMM import MDNPH
    g = golden("baseclip_tiny.npz")
    cfg, seed = recipe.CLIP_TINY, int(g["seed"])
    m = MDNPH(outputDim=16, num_classes=21, clipPath=_state(seed), saveDir=str(tmp_path)).to(DEV)
    m.float()
    for side in ("image", "text"):
        w, b = recipe.head_linear(cfg["embed_dim"], 16, seed, f"dnph_{side}_hash")
        getattr(m, f"{side}_hash").fc.weight.data.copy_(torch.from_numpy(w)); getattr(m, f"{side}_hash").fc.bias.data.copy_(torch.from_numpy(b))
        w, b = recipe.head_linear(cfg["embed_dim"], 21, seed, f"dnph_{side}_pre")
        getattr(m, f"{side}_pre").fc.weight.data.copy_(torch.from_numpy(w)); getattr(m, f"{side}_pre").fc.bias.data.copy_(torch.from_numpy(b))
    m.eval()
    with torch.no_grad():
        hi, pi, ht, pt = m(torch.from_numpy(recipe.images(3, cfg["image_resolution"], seed)).to(DEV),
                           torch.from_numpy(recipe.captions(3, 16, cfg["vocab_size"], seed)).to(DEV))
    for a, k in ((hi, "dnph_img"), (pi, "dnph_img_pre"), (ht, "dnph_txt"), (pt, "dnph_txt_pre")):
        np.testing.assert_allclose(a.cpu().numpy(), g[k], **TOL)


def test_linear_hash_dropout_mask_and_code_edge_cases():
    """Training-mode LinearHash with an injected keep-mask (GPU RNG != CPU RNG) vs the oracle; sign(0)=0;
    argmax tie -> -1 (train/base.py:150-158)."""
    import cmh_native as N
    from model.modelbase import LinearHash
    f = recipe.features(9, 512, 3, "lh")
    w, b = recipe.head_linear(512, 64, 3, "lhw")
    lh = LinearHash(512, 64).to(DEV)
    lh.fc.weight.data.copy_(torch.from_numpy(w)); lh.fc.bias.data.copy_(torch.from_numpy(b))
    mask = (np.random.default_rng(0).random((9, 64)) >= 0.2).astype(np.float32)
    lh.train()
    out = lh(torch.from_numpy(f).to(DEV), drop_mask=torch.from_numpy(mask).to(DEV)).detach().cpu().numpy()
    np.testing.assert_allclose(out, co.linear_hash(f, w, b, drop_mask=mask), **TOL)
    assert (out[mask == 0] == 0).all()
    rnd = lh(torch.from_numpy(f).to(DEV)).detach().cpu().numpy()       # device-drawn mask: ~20 % exact zeros
    assert 0.05 < (rnd == 0).mean() < 0.4
    h = torch.tensor([[0.0, -0.0, 1e-30, -3.0, float("nan")]], device=DEV)
    c = N.sign_codes(h).cpu().numpy()[0]
    assert list(c[:4]) == [0.0, 0.0, 1.0, -1.0] and np.isnan(c[4])
    p = torch.tensor([[0.5, 0.5, 0.2, 0.8, 0.9, 0.1]], device=DEV)
    assert N.pair_argmax_codes(p).cpu().numpy().tolist() == [[-1.0, 1.0, -1.0]]


@pytest.mark.parametrize("B,K,C", [(32, 64, 24), (48, 16, 80), (16, 128, 21), (8, 32, 24)])
def test_dsph_hyp_loss_matches_reference(golden, B, K, C):
    from train.DSPH.loss import HyP
    g = golden("loss_dsph.npz")
    tag, seed = f"B{B}_K{K}_C{C}", 21
    hyp = HyP(numclass=C, output_dim=K, hypseed=0, alpha=float(g[f"{tag}_alpha"])).to(DEV)
    hyp.proxies.data.copy_(torch.from_numpy(recipe.features(C, K, seed, f"dsph_prox_{tag}")))
    x = torch.tanh(torch.from_numpy(recipe.features(B, K, seed, f"dsph_x_{tag}"))).to(DEV)
    y = torch.tanh(torch.from_numpy(recipe.features(B, K, seed, f"dsph_y_{tag}"))).to(DEV)
    lab = torch.from_numpy(recipe.labels(B, C, seed, p=float(g[f"{tag}_p"]), tag=f"dsph_lab_{tag}"))
    loss = hyp(x, y, lab)
    assert abs(float(loss) - float(g[f"{tag}_loss"])) < 1e-4          # north_star tolerance for float losses


@pytest.mark.parametrize("B,K,C,fn,lt", [(32, 16, 24, "euclidean", "l2"), (32, 16, 24, "cosine", "l2"),
                                          (24, 64, 24, "euclidean", "l1"), (24, 64, 80, "cosine", "l1")])
def test_dchmt_loss_matches_reference(golden, B, K, C, fn, lt):
    import cmh_native as N
    g = golden("loss_dchmt.npz")
    tag, seed = f"B{B}_K{K}_C{C}_{fn}_{lt}", 31
    zi = recipe.features(B, 2 * K, seed, f"dchmt_zi_{tag}")
    zt = recipe.features(B, 2 * K, seed, f"dchmt_zt_{tag}")
    hi = N.pair_softmax(torch.from_numpy(2 * zi).to(DEV))
    ht = N.pair_softmax(torch.from_numpy(2 * zt).to(DEV))
    lab = torch.from_numpy(recipe.labels(B, C, seed, tag=f"dchmt_lab_{tag}")).to(DEV)
    loss = N.dchmt_loss(hi, ht, lab, K, fn, lt)
    ref = float(g[f"{tag}_loss"])
    assert abs(float(loss) - ref) < 1e-4 * max(1.0, abs(ref))


def test_trainer_valid_end_to_end(tmp_path, monkeypatch):
    """DSPH trainer on the synthetic dataset: get_code -> 4x calc_map_k (train/base.py:242-275) runs natively
    and its i->t mAP equals the oracle on the very codes it produced."""
    import argparse
    import sys
    import oracle
    import main
    from model.DSPH import MDSPH
    ck = tmp_path / "clip.pt"
    torch.save(_state(), ck)
    monkeypatch.setattr(sys, "argv", ["main.py", "-clip-path", str(ck), "--save-dir", str(tmp_path), "--batch-size", "16",
                                      "--num-workers", "0", "--resolution", "64", "--max-words", "16",
                                      "--query-num", "40", "--train-num", "60", "--synthetic-size", "200",
                                      "--epochs", "0"])
    args = argparse.Namespace(method="DSPH", dataset="synthetic", output_dim=16, is_train=True)
    import dataset.synthetic as ds
    monkeypatch.setattr(ds, "SOT", 510); monkeypatch.setattr(ds, "EOT", 511)   # tiny vocabulary
    tr = main.trainers["DSPH"](args, 0)
    tr.change_state(mode="valid")                    # heads to eval: no dropout in the codes
    q_img, q_txt, r_img, r_txt, _, _ = tr._codes_for_eval()
    assert set(np.unique(q_img.cpu().numpy())) <= {-1.0, 0.0, 1.0} and r_img.shape == (160, 16)
    maps = tr.valid(0)
    m_ref, _, _ = oracle.map_k(q_img.cpu().numpy(), r_txt.cpu().numpy(), tr.query_labels.numpy(), tr.retrieval_labels.numpy())
    assert abs(float(maps[0]) - float(m_ref)) < 2e-6
    # two real epochs: forward (tape) -> HyP loss -> backward through heads and both towers -> fused BertAdam + SGD on the proxies
    for grp in tr.optimizer.param_groups:
        grp["t_total"] = 8                       # the trainer was built with --epochs 0
    before = {n: p.detach().clone() for n, p in tr.model.named_parameters()}
    prox0 = tr.hyp.proxies.detach().clone()
    tr.train_epoch(0)
    tr.train_epoch(1)
    changed = [n for n, p in tr.model.named_parameters() if not torch.equal(p.detach(), before[n])]
    assert set(before) - set(changed) == {"clip.logit_scale"}, sorted(set(before) - set(changed))[:5]   # unused by DSPH, as upstream
    assert all(torch.isfinite(p).all() for p in tr.model.parameters()) and not torch.equal(prox0, tr.hyp.proxies.detach())


def test_cli_train_then_test_roundtrip(tmp_path, monkeypatch):
    """The reference's whole user flow (main.py -> trainer.run()): 2 epochs of train_epoch -> valid -> save_model, the PR-curve
    .mat files, then `--is-train false --pretrained model-1.pth` reproduces the last validation's mAPs from the checkpoint."""
    import argparse
    import sys
    import scipy.io as scio
    import main
    import dataset.synthetic as ds
    ck = tmp_path / "clip.pt"
    torch.save(_state(), ck)
    monkeypatch.setattr(ds, "SOT", 510); monkeypatch.setattr(ds, "EOT", 511)
    out = tmp_path / "run"
    common = ["main.py", "-clip-path", str(ck), "--save-dir", str(out), "--batch-size", "16", "--num-workers", "0", "--resolution", "64",
              "--max-words", "16", "--query-num", "24", "--train-num", "32", "--synthetic-size", "120", "--gemm-dtype", "f32"]
    monkeypatch.setattr(sys, "argv", common + ["--epochs", "2"])
    tr = main.trainers["DSPH"](argparse.Namespace(method="DSPH", dataset="synthetic", output_dim=16, is_train=True), 0)
    out = out / "DSPH" / "synthetic" / "16"                       # get_args: <save-dir>/<method>/<dataset>/<bits>, as upstream
    assert (out / "model-0.pth").exists() and (out / "model-1.pth").exists()
    mats = sorted(p.name for p in (out / "PR_cruve").glob("*.mat"))
    assert "16-ours-synthetic-i2t.mat" in mats
    m = scio.loadmat(out / "PR_cruve" / "16-ours-synthetic-i2t.mat")
    assert m["q_img"].shape == (24, 16) and m["r_txt"].shape == (96, 16) and m["q_l"].shape == (24, 24) and m["r_l"].shape == (96, 24)
    assert set(np.unique(m["q_img"])) <= {-1.0, 0.0, 1.0}
    tr.change_state(mode="valid")
    want = [float(v) for v in tr._four_maps(*tr._codes_for_eval()[:4])]
    common[common.index("--save-dir") + 1] = str(tmp_path / "run2")
    monkeypatch.setattr(sys, "argv", common + ["--epochs", "2", "--pretrained", str(out / "model-1.pth")])
    te = main.trainers["DSPH"](argparse.Namespace(method="DSPH", dataset="synthetic", output_dim=16, is_train=False), 0)
    # upstream quirk kept: the checkpoint is loaded while the CLIP weights are still fp16 (build_model's convert_weights), i.e.
    # BEFORE model.float() (train/base.py / hash_train.py _init_model), so those tensors come back rounded to fp16
    rounded = 0
    for (ka, a), (kb, b) in zip(tr.model.state_dict().items(), te.model.state_dict().items()):
        assert ka == kb
        if not torch.equal(a, b):
            assert torch.equal(a.half().float(), b), ka
            rounded += 1
    assert rounded > 0
    ca, cb = tr._codes_for_eval()[:4], te._codes_for_eval()[:4]
    for a, b in zip(ca, cb):
        assert float((a != b).float().mean()) < 0.02
    got = [float(v) for v in te._four_maps(*cb)]
    assert all(abs(g - w) < 0.03 for g, w in zip(got, want)), (got, want)


def test_training_learns(tmp_path, monkeypatch):
    """The whole stack as a learner: on a synthetic set whose images and captions carry their labels (class patterns / class
    tokens under noise) eight short DSPH epochs — tape forward, HyP, backward through heads and towers, fused BertAdam — must
    lift the mAPs well above their random-weights starting point (measured: i->t 0.42 -> 0.55, i->i 0.48 -> 0.70)."""
    import argparse
    import sys
    import main
    import dataset.synthetic as ds
    ck = tmp_path / "clip.pt"
    torch.save(_state(), ck)
    monkeypatch.setattr(ds, "SOT", 510); monkeypatch.setattr(ds, "EOT", 511)
    monkeypatch.setattr(ds.SyntheticPairs, "signal", 2.0)
    monkeypatch.setattr(sys, "argv", ["main.py", "-clip-path", str(ck), "--save-dir", str(tmp_path / "run"), "--batch-size", "32",
                                      "--num-workers", "0", "--resolution", "64", "--max-words", "16", "--query-num", "100",
                                      "--train-num", "400", "--synthetic-size", "600", "--epochs", "0", "--gemm-dtype", "f32",
                                      "--lr", "0.001", "--clip-lr", "0.0003"])
    tr = main.trainers["DSPH"](argparse.Namespace(method="DSPH", dataset="synthetic", output_dim=32, is_train=True), 0)
    epochs = 8
    for grp in tr.optimizer.param_groups:
        grp["t_total"] = epochs * len(tr.train_loader)
    before = [float(v) for v in tr.valid(0)]
    for epoch in range(epochs):
        tr.train_epoch(epoch)
    after = [float(v) for v in tr.valid(epochs)]
    assert after[0] > before[0] + 0.06 and after[1] > before[1] + 0.04 and after[2] > before[2] + 0.12, (before, after)


@pytest.mark.parametrize("method,K", [("DCHMT", 32), ("DNPH", 32), ("TwDH", 32), ("MITH", 16), ("DNpH", 32), ("DHaPH", 32)])
def test_every_method_learns(tmp_path, monkeypatch, method, K):
    """The other four trainers on the same learnable synthetic set (eight short epochs each: their own heads, losses, backward
    kernels and the fused BertAdam): the image-to-text mAP and the sum of the four mAPs must rise (measured i->i: DCHMT 0.49 -> 0.59,
    DNPH 0.48 -> 0.68, TwDH 0.58 -> 0.72, MITH 0.48 -> 0.55)."""
    import argparse
    import sys
    import main
    import dataset.synthetic as ds
    ck = tmp_path / "clip.pt"
    sd = recipe.clip_state_dict(dict(recipe.CLIP_TINY, embed_dim=512), 7)            # MITH's HashingModel is built for 512-d features
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, ck)
    monkeypatch.setattr(ds, "SOT", 510); monkeypatch.setattr(ds, "EOT", 511)
    monkeypatch.setattr(ds.SyntheticPairs, "signal", 2.0)
    monkeypatch.setattr(sys, "argv", ["main.py", "-clip-path", str(ck), "--save-dir", str(tmp_path / "run"), "--batch-size", "32",
                                      "--num-workers", "0", "--resolution", "64", "--max-words", "16", "--query-num", "100",
                                      "--train-num", "400", "--synthetic-size", "600", "--epochs", "0", "--gemm-dtype", "f32",
                                      "--lr", "0.001", "--clip-lr", "0.0003"])
    torch.manual_seed(0)
    tr = main.trainers[method](argparse.Namespace(method=method, dataset="synthetic", output_dim=K, is_train=True), 0)
    epochs = 8
    for grp in tr.optimizer.param_groups:
        grp["t_total"] = epochs * len(tr.train_loader)

    def maps():
        r = tr.valid(0)
        return [float(v) for v in (r["long"] if isinstance(r, dict) else r)[:4]]
    before = maps()
    for epoch in range(epochs):
        tr.train_epoch(epoch)
    after = maps()
    assert after[0] > before[0] + 0.015 and sum(after) > sum(before) + 0.08, (method, before, after)


def test_dmsh_ln_trainer_steps(tmp_path, monkeypatch):
    """DMsH-LN end to end (train/DMsH_LN/hash_train.py): (1) as upstream ships it - a freshly initialised LabelNet whose codes make
    every pair 'similar' - every row of the multi-similarity loss is skipped, the loss is the reference's constant zero
    (MSLOSS.py:53-54) and an optimiser step moves nothing; (2) with label codes that do split the pairs (fc2's bias removes the
    batch mean, as a trained LabelNet would) the loss is live and falls on a fixed batch through the native forward, loss,
    backward and fused BertAdam."""
    import argparse
    import sys
    import main
    import dataset.synthetic as ds
    ck = tmp_path / "clip.pt"
    sd = recipe.clip_state_dict(dict(recipe.CLIP_TINY, embed_dim=512), 7)
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, ck)
    monkeypatch.setattr(ds, "SOT", 510); monkeypatch.setattr(ds, "EOT", 511)
    monkeypatch.setattr(sys, "argv", ["main.py", "-clip-path", str(ck), "--save-dir", str(tmp_path / "run"), "--batch-size", "32",
                                      "--num-workers", "0", "--resolution", "64", "--max-words", "16", "--query-num", "100",
                                      "--train-num", "400", "--synthetic-size", "600", "--epochs", "0", "--gemm-dtype", "f32",
                                      "--lr", "0.001", "--clip-lr", "0.0003"])
    torch.manual_seed(0)
    tr = main.trainers["DMsH_LN"](argparse.Namespace(method="DMsH_LN", dataset="synthetic", output_dim=32, is_train=True), 0)
    for grp in tr.optimizer.param_groups:
        grp["t_total"] = 100
    tr.change_state(mode="train")
    image, text, label, _ = next(iter(tr.train_loader))
    watch = [tr.model.image_hash.fc.weight, tr.model.clip.visual.proj, tr.L_net.fc1.weight]
    before = [w.detach().clone() for w in watch]
    loss = tr._step(image, text, label)
    assert float(loss) == 0.0
    assert all(torch.equal(a, b.detach()) for a, b in zip(before, watch))
    with torch.no_grad():
        all_lab = tr.train_loader.dataset.get_all_label().to(DEV).float()
        feat = torch.relu(all_lab @ tr.L_net.fc1.weight.t() + tr.L_net.fc1.bias)
        tr.L_net.fc2.bias.copy_(-(feat.mean(0) @ tr.L_net.fc2.weight.t()))
    losses = [float(tr._step(image, text, label)) for _ in range(12)]
    assert losses[0] > 0.5 and losses[-1] < losses[0] - 0.15, losses          # measured 5.45 -> 5.17 (ten of the twelve steps are warm-up)
    assert not torch.equal(before[0], watch[0].detach()) and torch.equal(before[2], watch[2].detach())     # LabelNet never moves

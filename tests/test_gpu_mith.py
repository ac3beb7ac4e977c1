"""MITH on the GPU: token-returning trunk (a5), HashingModel (a8) and the loss groups (a13) vs goldens produced by the
reference's model/MITH.py and train/MITH/hash_train.py."""
import argparse
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import mithutil as mu
import recipe

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_trunk_returns_all_tokens(golden):
    from model.MITH import build_model
    g = golden("mith.npz")
    cfg, seed = mu.CLIP_TINY512, 7
    clip = build_model({k: torch.from_numpy(v) for k, v in recipe.clip_state_dict(cfg, seed).items()}).to(DEV).float()
    image = tt(recipe.images(3, cfg["image_resolution"], seed))
    text = recipe.captions(3, 16, cfg["vocab_size"], seed)
    with torch.no_grad():
        seq_i, aw, cls_i = clip.encode_image(image)
        seq_t, _, new_kpm, eos = clip.encode_text(tt(text), tt(text == 0))
    assert aw is None and seq_i.shape == (4, 3, 512) and seq_t.shape == (16, 3, 512) and new_kpm.dtype == torch.bool
    tol = dict(rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(seq_i.cpu().numpy(), g["trunk_seq_i"], **tol)
    np.testing.assert_allclose(cls_i.cpu().numpy(), g["trunk_cls_i"], **tol)
    # padded positions attend to nothing meaningful upstream either; compare the rows a caption really has + EOS
    valid = (text != 0).T
    np.testing.assert_allclose(seq_t.cpu().numpy()[valid], g["trunk_seq_t"][valid], **tol)
    np.testing.assert_allclose(eos.cpu().numpy(), g["trunk_eos_t"], **tol)


_PACK_CFGS = {"w128": (dict(mu.CLIP_TINY512), 16),                  # width 128: the GEMMs take their row count from the host (one read-back)
              "w256": (dict(mu.CLIP_TINY512, context_length=24, transformer_width=256, transformer_heads=4), 24)}   # device-side row count, fp16 stream in bf16 mode


def test_token_packing_edge_captions():
    """an all-zero caption (every position padded: the EOT's argmax is position 0, one row is kept), a caption without any padding, and
    a batch with no padded position at all (the packed plan is the dense one): kept positions equal the dense call's bits"""
    import mith_ops as M
    from model.MITH import build_model
    cfg, L = _PACK_CFGS["w256"]
    seed, B = 3, 4
    clip = build_model({k: torch.from_numpy(v) for k, v in recipe.clip_state_dict(cfg, seed).items()}).to(DEV).float().set_gemm_dtype("bf16")
    text_np = recipe.captions(B, L, cfg["vocab_size"], seed).copy()
    text_np[1, :] = 0                                         # nothing but padding
    text_np[2, :] = np.arange(5, 5 + L); text_np[2, -1] = cfg["vocab_size"] - 1      # no padding at all, EOT last
    full = np.tile(text_np[2], (B, 1))
    for t_np in (text_np, full):
        text, kpm = tt(t_np), tt(t_np == 0)
        with torch.no_grad():
            dense, rows_d = M.text_encode_tokens(clip, text, kpm)
            packed, rows_p = M.text_encode_tokens(clip, text, kpm, padded_unused=True)
        assert torch.equal(rows_d, rows_p)
        last = np.array([np.flatnonzero(r != 0).max() if (r != 0).any() else -1 for r in t_np])
        keep = torch.from_numpy(np.arange(L)[None, :] <= np.maximum(last, t_np.argmax(1))[:, None]).to(DEV)
        # (the all-padded caption has every key masked: its one kept row is NaN in the reference, in the dense call and here)
        nn = lambda t: torch.nan_to_num(t, nan=12345.0)
        assert torch.equal(nn(packed[keep]), nn(dense[keep])) and (bool(keep.all()) or float(packed[~keep].abs().max()) == 0.0)
        ok_rows = torch.from_numpy((t_np != 0).any(1)).to(DEV)
        assert bool(torch.isfinite(packed[ok_rows]).all())


def _ragged_text(B, L, vocab, seed):
    """captions with their EOT anywhere, zeros behind it, and - the reference's quirk - one '!' (token id 0, which `text == 0` masks
    like padding) INSIDE a caption's prefix"""
    text = recipe.captions(B, L, vocab, seed).copy()
    eot = text.argmax(1)
    b = int(np.argmax(eot))            # the longest caption gets the inner zero
    if eot[b] >= 3:
        text[b, 2] = 0
    return text


@pytest.mark.parametrize("cfgname", ["w128", "w256"])
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_token_packing_keeps_the_bits_of_every_position_somebody_reads(mode, cfgname):
    """cmh_text_encode_tokens_packed (round 5; what MITH.forward runs): rows behind a caption's last unpadded token are not computed.
    Against the dense call: every position up to there carries the same bits - the key-padding mask still applies to the keys inside
    the kept prefix -, the positions behind are zeros, the EOT rows and features are the same, and HashingModel (the only reader,
    reference model/MITH.py:349-376, 424-453) returns the same bits from either."""
    import cmh_native as Nn
    import mith_ops as M
    from model.MITH import HashingModel, build_model
    (cfg, L), seed, B = _PACK_CFGS[cfgname], 7, 6
    clip = build_model({k: torch.from_numpy(v) for k, v in recipe.clip_state_dict(cfg, seed).items()}).to(DEV).float().set_gemm_dtype(mode)
    text_np = _ragged_text(B, L, cfg["vocab_size"], seed)
    text, kpm = tt(text_np), tt(text_np == 0)
    with torch.no_grad():
        dense, rows_d = M.text_encode_tokens(clip, text, kpm)
        packed, rows_p = M.text_encode_tokens(clip, text, kpm, padded_unused=True)
        try:
            Nn.set_text_token_packing(0)
            off, rows_o = M.text_encode_tokens(clip, text, kpm, padded_unused=True)      # the switch: exactly the dense call
        finally:
            Nn.set_text_token_packing(-1)
    assert torch.equal(rows_d, rows_p) and torch.equal(off, dense) and torch.equal(rows_o, rows_d)
    last = np.array([np.flatnonzero(r != 0).max() for r in text_np])
    keep = torch.from_numpy(np.arange(L)[None, :] <= np.maximum(last, text_np.argmax(1))[:, None]).to(DEV)
    assert 0 < int((~keep).sum()) < B * L                                     # something was skipped, something kept
    assert torch.equal(packed[keep], dense[keep])
    # (the packed projection lands in the tower's MLP scratch: 4 d e bytes per row must hold embed_dim f32 - every real tower's do, the
    # 128-wide test tower's bf16 scratch does not, and that call stays dense)
    packs = 512 * 4 <= 4 * cfg["transformer_width"] * (2 if mode == "bf16" else 4)
    assert (float(packed[~keep].abs().max()) == 0.0) == packs
    hm = HashingModel(clip_embed_dim=512, args=SimpleNamespace(output_dim=16, **mu.ARGS)).to(DEV).eval().set_gemm_dtype(mode)
    img_tokens, img_cls = torch.randn(9, B, 512, device=DEV), torch.randn(B, 512, device=DEV)
    new_kpm = kpm + (text == 49407)
    with torch.no_grad():
        outs = []
        for tok, rows in ((dense, rows_d), (packed, rows_p)):
            eos = tok.reshape(B * L, -1)[rows.long()]
            outs.append(hm(img_tokens, tok.permute(1, 0, 2), img_cls, eos, new_kpm))
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("cfgname", ["w128", "w256"])
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_token_packing_under_training_matches_the_dense_tape(mode, cfgname):
    """cmh_text_forward_train_tokens_packed + cmh_text_backward_tokens on its tape against the dense pair, with a loss that - like
    MITH's - puts no weight on padded positions: same tokens at the kept positions, every parameter gradient equal up to the
    summation order of the weight gradients (fewer rows in the contraction)."""
    from model.MITH import build_model
    (cfg, L), seed, B = _PACK_CFGS[cfgname], 7, 6
    text_np = _ragged_text(B, L, cfg["vocab_size"], seed)
    text, kpm = tt(text_np), tt(text_np == 0)
    last = np.array([np.flatnonzero(r != 0).max() for r in text_np])
    keep = torch.from_numpy(np.arange(L)[None, :] <= np.maximum(last, text_np.argmax(1))[:, None]).to(DEV)
    gen = torch.Generator().manual_seed(5)
    G = torch.randn(L, B, 512, generator=gen).to(DEV) * keep.T[:, :, None]          # no weight on the padded positions
    Ge = torch.randn(B, 512, generator=gen).to(DEV)
    res = {}
    for packed in (False, True):
        clip = build_model({k: torch.from_numpy(v) for k, v in recipe.clip_state_dict(cfg, seed).items()}).to(DEV).float().set_gemm_dtype(mode)
        clip.padded_tokens_unused = packed
        seq_t, _, _, eos = clip.encode_text(text, kpm)
        ((seq_t * G).sum() + (eos * Ge).sum()).backward()
        res[packed] = (seq_t.detach().clone(), eos.detach().clone(), {n: p.grad.detach().clone() for n, p in clip.named_parameters() if p.grad is not None})
    assert torch.equal(res[True][1], res[False][1])
    assert torch.equal(res[True][0][keep.T], res[False][0][keep.T]) and float(res[True][0][~keep.T].abs().max()) == 0.0      # (the tape has its own [M, E] f32 region)
    assert res[True][2].keys() == res[False][2].keys() and len(res[True][2]) > 20
    tol = 2e-5 if mode == "f32" else 4e-3
    for name, ref in res[False][2].items():
        got = res[True][2][name]
        err = float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-6)
        assert err < tol, (name, err)


@pytest.mark.parametrize("Nb,L,K", [(3, 12, 16), (4, 32, 64)])
def test_hashing_model_matches_reference(golden, Nb, L, K):
    from model.MITH import HashingModel
    g = golden("mith.npz")
    tag = f"N{Nb}_L{L}_K{K}"
    hm = HashingModel(clip_embed_dim=512, args=SimpleNamespace(output_dim=K, **mu.ARGS))
    ref_keys = sorted(str(k) for k in g[f"{tag}_keys"])
    assert sorted(hm.state_dict().keys()) == ref_keys            # state_dict contract incl. shared gcl_i/gcl_t and pe buffers
    st = mu.fill_state({k: tuple(v.shape) for k, v in hm.state_dict().items()}, 100 + K)
    hm.load_state_dict({k: (torch.from_numpy(st[k]) if k in st else v) for k, v in hm.state_dict().items()})
    hm = hm.to(DEV).eval()
    c = mu.hash_inputs(Nb, L, K)
    with torch.no_grad():
        od = hm(tt(c["img_tokens"]), tt(c["txt_tokens"]), tt(c["img_cls"]), tt(c["txt_eos"]), tt(c["kpm"]))
    assert od["trans_tokens_i"].shape == (K, Nb, 512) and od["img_tokens_hash"].shape == (Nb, K)
    for k, v in od.items():
        v = v.cpu().numpy()
        if k.startswith("trans_tokens") and K == 64:
            v = v[::4]
        np.testing.assert_allclose(v, g[f"{tag}_{k}"], rtol=2e-3, atol=2e-4, err_msg=k)


@pytest.mark.parametrize("Nb,K,C,Mb", [(8, 16, 24, 50), (16, 64, 80, 200)])
def test_losses_match_reference(golden, Nb, K, C, Mb):
    from train.MITH.hash_train import MITHTrainer
    g = golden("mith.npz")
    tag = f"loss_N{Nb}_K{K}"
    od, banks, label, train_labels = mu.loss_inputs(Nb, K, C, Mb)
    self = SimpleNamespace(args=SimpleNamespace(**mu.HP), rank=0, k_bits=K, train_labels=tt(train_labels),
                           img_buffer_tokens=tt(banks["img_tokens"]), img_buffer_cls=tt(banks["img_cls"]),
                           txt_buffer_tokens=tt(banks["txt_tokens"]), txt_buffer_cls=tt(banks["txt_cls"]),
                           model=SimpleNamespace(hash=SimpleNamespace(img_concept_proj=SimpleNamespace(weight=torch.zeros(1)))))
    for name in ("bayesian_loss", "info_nce_loss", "info_nce_loss_bmm", "quantization_loss_2", "make_B", "sq_diff"):
        setattr(self, name, (lambda n: (lambda *a, **k: getattr(MITHTrainer, n)(self, *a, **k)))(name))
    self._grad = MITHTrainer._grad
    tod = {k: tt(v) for k, v in od.items()}
    with torch.no_grad():
        Bc, _, _ = self.make_B(tod)
        L = MITHTrainer.compute_loss(self, tod, tt(label))
    np.testing.assert_array_equal(Bc.cpu().numpy(), g[f"{tag}_B"])
    for k, v in L.items():
        ref = float(g[f"{tag}_{k}"])
        assert abs(float(v) - ref) < 1e-4 * max(1.0, abs(ref)), (k, float(v), ref)


def test_mith_trainer_end_to_end(tmp_path, monkeypatch):
    import dataset.synthetic as ds
    import main
    ck = tmp_path / "clip.pt"
    torch.save({k: torch.from_numpy(v) for k, v in recipe.clip_state_dict(mu.CLIP_TINY512, 7).items()}, ck)
    monkeypatch.setattr(ds, "SOT", 510); monkeypatch.setattr(ds, "EOT", 511)
    monkeypatch.setattr(sys, "argv", ["main.py", "-clip-path", str(ck), "--save-dir", str(tmp_path), "--batch-size", "8",
                                      "--num-workers", "0", "--resolution", "64", "--max-words", "16", "--query-num", "16",
                                      "--train-num", "24", "--synthetic-size", "64", "--epochs", "0"])
    tr = main.trainers["MITH"](argparse.Namespace(method="MITH", dataset="synthetic", output_dim=16, is_train=True), 0)
    maps = tr.valid(0)
    assert all(0.0 <= float(m) <= 1.0 for m in maps)
    image, text, kpm, label, index = next(iter(tr.train_loader))
    with torch.no_grad():
        od = tr.model(image.to(DEV), text.to(DEV), kpm.to(DEV))
        losses = tr.compute_loss(od, label)
    assert set(losses) == {"tokens_intra_likelihood", "cls_inter_likelihood", "quantization", "infoNCE", "distillation"}
    assert all(torch.isfinite(v).item() for v in losses.values())
    # two real epochs: token-returning trunk with tape -> HashingModel -> memory banks -> five loss groups -> backward through the
    # HashingModel and both towers -> fused BertAdam
    for grp in tr.optimizer.param_groups:
        grp["t_total"] = 8
    before = {n: p.detach().clone() for n, p in tr.model.named_parameters()}
    banks0 = tr.img_buffer_cls.clone()
    tr.train_epoch(0)
    tr.train_epoch(1)
    same = {n for n, p in tr.model.named_parameters() if torch.equal(p.detach(), before[n])}
    assert same <= {"clip.logit_scale"}, sorted(same)[:8]
    assert all(torch.isfinite(p).all() for p in tr.model.parameters())
    assert not torch.equal(banks0, tr.img_buffer_cls)                 # the memory banks took this epoch's codes
    maps = tr.valid(1)
    assert all(0.0 <= float(m) <= 1.0 for m in maps)


def test_hashing_model_bf16_gemms_track_f32():
    """HashingModel.set_gemm_dtype("bf16"): bf16 operands in the ResidualMLPs / concept transformer / concept projections; the
    outputs stay within bf16 noise of the f32 (parity) mode and the sign codes rarely flip."""
    from model.MITH import HashingModel
    Nb, L, K = 16, 32, 64
    hm = HashingModel(clip_embed_dim=512, args=SimpleNamespace(output_dim=K, **mu.ARGS))
    st = mu.fill_state({k: tuple(v.shape) for k, v in hm.state_dict().items()}, 164)
    hm.load_state_dict({k: (torch.from_numpy(st[k]) if k in st else v) for k, v in hm.state_dict().items()})
    hm = hm.to(DEV).eval()
    c = mu.hash_inputs(Nb, L, K)
    args = [tt(c[k]) for k in ("img_tokens", "txt_tokens", "img_cls", "txt_eos", "kpm")]
    with torch.no_grad():
        ref = {k: v.clone() for k, v in hm.set_gemm_dtype("f32")(*args).items()}
        out = hm.set_gemm_dtype("bf16")(*args)
    report = {}
    for k, v in out.items():
        a, r = v.double().flatten(), ref[k].double().flatten()
        report[k] = (float(a @ r / (a.norm() * r.norm())),
                     float((torch.sign(v) != torch.sign(ref[k])).float().mean()) if k.endswith("_hash") else 0.0)
    print("bf16 vs f32 HashingModel (cosine, sign-flip rate):", {k: (round(c, 4), round(f, 4)) for k, (c, f) in report.items()})
    for k, (cos, flips) in report.items():
        # everything downstream of the top-k token selection (LTA, model/MITH.py:140-160) can change discretely when a
        # similarity moves by bf16 noise on these random weights, so those outputs get the looser bound
        local = "tokens" in k
        assert cos > (0.97 if local else 0.999), (k, cos)
        assert flips < (0.10 if local else 0.03), (k, flips)


def test_trunk_token_gradients_match_reference_autograd(golden):
    """The MITH trunk under training (every token projected, key_padding_mask in the text blocks): all parameter gradients of
    L = sum(seq_i G1) + sum(cls_i G2) + sum(seq_t G3) + sum(eos_t G4) against torch autograd on the REFERENCE's CLIP1
    (tests/golden/make_golden11.py), f32 mode."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_golden5 import cut
    import recipe
    from model.MITH import build_model
    g = golden("mith_trunk_grads.npz")
    cfg, seed = mu.CLIP_TINY512, 7
    sd = {k: torch.from_numpy(v) for k, v in recipe.clip_state_dict(cfg, seed).items()}
    clip = build_model(sd).to(DEV).float().set_gemm_dtype("f32")
    image = torch.from_numpy(recipe.images(3, cfg["image_resolution"], seed)).to(DEV)
    text_np = recipe.captions(3, 16, cfg["vocab_size"], seed)
    text, kpm = torch.from_numpy(text_np).to(DEV), torch.from_numpy(text_np == 0).to(DEV)
    seq_i, _, cls_i = clip.encode_image(image)
    seq_t, _, new_kpm, eos_t = clip.encode_text(text, kpm)
    for a, k in ((seq_i, "seq_i"), (cls_i, "cls_i"), (eos_t, "eos_t")):
        np.testing.assert_allclose(a.detach().cpu().numpy(), g[k], rtol=1e-3, atol=2e-4)
    gen = torch.Generator().manual_seed(31)
    G = [torch.randn(x.shape, generator=gen).to(DEV) for x in (seq_i, cls_i, seq_t, eos_t)]
    ((seq_i * G[0]).sum() + (cls_i * G[1]).sum() + (seq_t * G[2]).sum() + (eos_t * G[3]).sum()).backward()
    params = dict(clip.named_parameters())
    worst = 0.0
    for name in [str(n) for n in g["names"]]:
        got = params[name].grad
        assert got is not None, name
        ref, norm = g["g_" + name], float(g["n_" + name])
        assert abs(float(got.double().norm()) - norm) <= 5e-4 * max(norm, 1e-3), (name, float(got.double().norm()), norm)
        err = np.abs(cut(got.cpu().numpy()) - ref).max() / max(np.abs(ref).max(), 1e-6)
        worst = max(worst, err)
        assert err < 1e-3, (name, err)
    print(f"MITH trunk gradients: worst relative-to-max error {worst:.2e} over {len(g['names'])} tensors")


def test_hashing_model_gradients_match_reference_autograd(golden):
    """HashingModel under training (ResidualMLPs with exact GELU, concept logits, F.normalize, token aggregation with detached
    similarities, positional encoding, the 2-layer concept transformer, bitwise hashing, concept projections): gradients w.r.t.
    the four inputs and all 129 parameters of L = sum <output, cotangent> against torch autograd on the REFERENCE's module
    (tests/golden/make_golden12.py), f32 mode."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden12 as mg
    from model.MITH import HashingModel
    g = golden("mith_hash_grads.npz")
    Nb, L, K = 3, 12, 16
    tag = f"N{Nb}_L{L}_K{K}"
    hm = HashingModel(clip_embed_dim=512, args=SimpleNamespace(output_dim=K, **mu.ARGS))
    st = mu.fill_state({k: tuple(v.shape) for k, v in hm.state_dict().items()}, 100 + K)
    hm.load_state_dict({k: (torch.from_numpy(st[k]) if k in st else v) for k, v in hm.state_dict().items()})
    hm = hm.to(DEV).train()
    c = mu.hash_inputs(Nb, L, K)
    ins = {k: tt(c[k]).requires_grad_() for k in ("img_tokens", "txt_tokens", "img_cls", "txt_eos")}
    od = hm(ins["img_tokens"], ins["txt_tokens"], ins["img_cls"], ins["txt_eos"], tt(c["kpm"]))
    G = mg.cotangents({k: v.detach().cpu() for k, v in od.items()})
    sum((od[k] * G[k].to(DEV)).sum() for k in mg.KEYS).backward()
    worst = 0.0

    def check(name, got, key_g, key_n):
        nonlocal worst
        assert got is not None, name
        ref, norm = g[key_g], float(g[key_n])
        assert abs(float(got.double().norm()) - norm) <= 1e-3 * max(norm, 1e-3), (name, float(got.double().norm()), norm)
        err = np.abs(mg.cut(got.cpu().numpy()) - ref).max() / max(np.abs(ref).max(), 1e-6)
        worst = max(worst, err)
        assert err < 2e-3, (name, err)
    for k, v in ins.items():
        check(k, v.grad, f"{tag}_d_{k}", f"{tag}_n_{k}")
    params = dict(hm.named_parameters())
    names = [str(n) for n in g[f"{tag}_names"]]
    assert len(names) == 129
    for name in names:
        check(name, params[name].grad, f"{tag}_g_{name}", f"{tag}_n_{name}")
    print(f"HashingModel gradients: worst relative-to-max error {worst:.2e} over {len(names) + 4} tensors")


@pytest.mark.parametrize("Nb,K,C,Mb", [(8, 16, 24, 50), (16, 64, 80, 200)])
def test_loss_gradients_match_reference_autograd(golden, Nb, K, C, Mb):
    """The summed MITH step loss (bayesian x4 against the memory banks, quantisation, InfoNCE on cls and on concept tokens,
    distillation with its detach pattern) differentiated on the GPU w.r.t. the eight HashingModel outputs, against torch
    autograd on the REFERENCE's compute_loss (tests/golden/make_golden13.py)."""
    from train.MITH.hash_train import MITHTrainer
    g = golden("mith_loss_grads.npz")
    tag = f"loss_N{Nb}_K{K}"
    od, banks, label, train_labels = mu.loss_inputs(Nb, K, C, Mb)
    self = SimpleNamespace(args=SimpleNamespace(**mu.HP), rank=0, k_bits=K, train_labels=tt(train_labels),
                           img_buffer_tokens=tt(banks["img_tokens"]), img_buffer_cls=tt(banks["img_cls"]),
                           txt_buffer_tokens=tt(banks["txt_tokens"]), txt_buffer_cls=tt(banks["txt_cls"]))
    for name in ("bayesian_loss", "info_nce_loss", "info_nce_loss_bmm", "quantization_loss_2", "make_B", "sq_diff", "_grad"):
        setattr(self, name, (lambda n: (lambda *a, **k: getattr(MITHTrainer, n)(self, *a, **k)))(name) if name != "_grad" else MITHTrainer._grad)
    tod = {k: tt(v).requires_grad_() for k, v in od.items()}
    L = MITHTrainer.compute_loss(self, tod, tt(label))
    total = sum(L.values())
    assert abs(float(total.detach()) - float(g[f"{tag}_total"])) < 1e-4 * max(1.0, abs(float(g[f"{tag}_total"])))
    total.backward()
    for k, v in tod.items():
        want, got = g[f"{tag}_d_{k}"], v.grad.cpu().numpy()
        if k.startswith("trans_tokens"):
            got = got[::4, :, ::4]
        np.testing.assert_allclose(got, want, rtol=1e-3, atol=2e-5 * max(np.abs(want).max(), 1e-30), err_msg=k)


def test_mismatched_feature_width_fails_loudly():
    """A HashingModel built for 512-d features refuses narrower ones before any kernel runs, and the GEMM / LayerNorm wrappers
    refuse operands whose shapes do not fit (a residual of the wrong width would otherwise be read out of bounds)."""
    import cmh_native as N
    import mith_ops as M
    from model.MITH import HashingModel
    hm = HashingModel(clip_embed_dim=512, args=SimpleNamespace(output_dim=16, **mu.ARGS)).to(DEV).eval()
    z = lambda *s: torch.zeros(*s, device=DEV)
    with pytest.raises(ValueError):
        hm(z(49, 2, 64), z(8, 2, 64), z(2, 64), z(2, 64), torch.zeros(2, 8, dtype=torch.bool, device=DEV))
    with pytest.raises(N.NativeError):
        M.gemm(z(4, 64), z(128, 512), None)
    with pytest.raises(N.NativeError):
        M.gemm(z(4, 512), z(128, 512), None, residual=z(4, 64))
    with pytest.raises(N.NativeError):
        N.layernorm(z(4, 64), z(512), z(512))


@pytest.mark.parametrize("R,G,D", [(256, 256, 512), (1024, 64, 512), (120, 40, 64), (200, 200, 128), (96, 8, 64)])
def test_info_nce_on_the_matrix_pipe_equals_the_per_row_kernels(R, G, D):
    """cmh_info_nce / cmh_info_nce_backward with the workspace of cmh_info_nce_workspace_bytes form the scores once as 16 x 16 MFMA
    tiles (train/MITH/hash_train.py:103-136 upstream: info_nce_loss / info_nce_loss_bmm); with a small workspace they keep the
    per-row kernels.  Both against torch autograd in float64, and against each other (f32 summation order apart); group sizes that are
    not multiples of the tile included."""
    import cmh_native as N
    g = torch.Generator().manual_seed(R + G + D)
    a = torch.nn.functional.normalize(torch.randn(R, D, generator=g), dim=-1).to(DEV)
    b = torch.nn.functional.normalize(torch.randn(R, D, generator=g), dim=-1).to(DEV)
    dl = torch.tensor([0.7], device=DEV)
    res = {}
    for name, nbytes in (("tiles", N.lib().cmh_info_nce_workspace_bytes(R, G)), ("rows", 2 * R * 4 + 512)):
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
        out = torch.zeros(1, device=DEV)
        da, db = torch.full_like(a, float("nan")), torch.full_like(b, float("nan"))
        N.check(N.lib().cmh_info_nce(N.ptr(a), N.ptr(b), R, G, D, 0.07, N.ptr(out), N.ptr(ws), ws.numel(), N.stream_ptr(a.device)), "nce")
        N.check(N.lib().cmh_info_nce_backward(N.ptr(a), N.ptr(b), R, G, D, 0.07, N.ptr(dl), N.ptr(da), N.ptr(db), N.ptr(ws), ws.numel(),
                                              N.stream_ptr(a.device)), "nce_bwd")
        res[name] = (out.cpu().double(), da.cpu().double(), db.cpu().double())
    a64, b64 = a.cpu().double().requires_grad_(True), b.cpu().double().requires_grad_(True)
    sc = torch.einsum("gid,gjd->gij", a64.view(-1, G, D), b64.view(-1, G, D)) / 0.07
    tgt = torch.arange(G).repeat(R // G)
    loss = 0.5 * (torch.nn.functional.cross_entropy(sc.reshape(R, G), tgt) + torch.nn.functional.cross_entropy(sc.transpose(1, 2).reshape(R, G), tgt))
    (0.7 * loss).backward()
    for name, (lo, da, db) in res.items():
        torch.testing.assert_close(lo[0], loss.detach(), rtol=2e-6, atol=2e-6, msg=lambda m: f"{name}: {m}")
        for got, ref in ((da, a64.grad), (db, b64.grad)):
            assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), name
    torch.testing.assert_close(res["tiles"][1], res["rows"][1], rtol=1e-4, atol=1e-6 * float(res["rows"][1].abs().max()))


@pytest.mark.parametrize("Mb,B,K,C", [(10000, 256, 64, 80), (1500, 37, 32, 21), (700, 130, 128, 24), (300, 16, 16, 5)])
def test_bayesian_loss_on_the_matrix_pipe(Mb, B, K, C):
    """cmh_mith_bayesian_loss(+_backward) as 16 x 16 MFMA tiles (K, C <= 128) against torch autograd in float64 (train/MITH/hash_train.py:
    138-144 upstream), ragged bank slices and batch tiles included, and one pair beyond the +-64 clamp (no gradient there)."""
    import mith_train_ops as T
    g = torch.Generator().manual_seed(Mb + B + K + C)
    bank = torch.randn(Mb, K, generator=g).tanh()
    bank[3] = 4.0                                              # bank_3 . batch_5 = 4 * 4 * K > 64: clamped
    batch = torch.randn(B, K, generator=g).tanh()
    batch[5] = 4.0
    bl, lab = (torch.rand(Mb, C, generator=g) < 0.15).float(), (torch.rand(B, C, generator=g) < 0.15).float()
    x = batch.to(DEV).requires_grad_(True)
    loss = T.BayesianLossFn.apply(bank.to(DEV), x, bl.to(DEV), lab.to(DEV))
    (0.3 * loss).backward()
    x64 = batch.double().requires_grad_(True)
    s = 0.5 * (bank.double() @ x64.t()).clamp(-64, 64)
    ref = -(((bl.double() @ lab.double().t()) > 0).double() * s - torch.log(1 + torch.exp(s))).mean()
    (0.3 * ref).backward()
    torch.testing.assert_close(loss.detach().cpu().double(), ref.detach(), rtol=2e-6, atol=2e-6)
    assert float((x.grad.cpu().double() - x64.grad).abs().max()) <= 2e-5 * float(x64.grad.abs().max())

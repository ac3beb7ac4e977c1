"""TwDH and DNPH heads / losses on the GPU (C ABI) vs goldens produced by the reference."""
import numpy as np
import pytest
import torch

from heads2util import DNPH_CASES, TWDH_CASES, dnph_case, twdh_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = dict(rtol=1e-4, atol=1e-4)
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _head(K, p, layernorm):
    from model.TwDH import ModalityHash
    h = ModalityHash(inputDim=512, outputDim=K, layernorm=layernorm, num_heads=8, batch_first=True, hash_func="softmax").to(DEV)
    h.atten.in_proj_weight.data.copy_(tt(p["in_w"])); h.atten.in_proj_bias.data.copy_(tt(p["in_b"]))
    h.atten.out_proj.weight.data.copy_(tt(p["out_w"])); h.atten.out_proj.bias.data.copy_(tt(p["out_b"]))
    h.norm.weight.data.copy_(tt(p["norm_w"])); h.norm.bias.data.copy_(tt(p["norm_b"]))
    h.fc2.weight.data.copy_(tt(p["fc2_w"])); h.fc2.bias.data.copy_(tt(p["fc2_b"]))
    return h


@pytest.mark.parametrize("B,K,S,C", TWDH_CASES)
def test_twdh_heads_short_codes_targets_loss(golden, B, K, S, C):
    import cmh_native as N
    from model.TwDH import softmax_hash
    g = golden("twdh.npz")
    c = twdh_case(B, K, S, C)
    tag = c["tag"]
    with torch.no_grad():
        li = _head(K, c["p_img"], False)(tt(c["feat_i"]))
        lt = _head(K, c["p_txt"], True)(tt(c["feat_t"]))
    np.testing.assert_allclose(li.cpu().numpy(), g[f"{tag}_img_long"], **TOL)
    np.testing.assert_allclose(lt.cpu().numpy(), g[f"{tag}_txt_long"], **TOL)
    trans_t = tt(c["trans"]).t().contiguous()
    si = softmax_hash(N.linear_act(li, trans_t, None))
    st = softmax_hash(N.linear_act(lt, trans_t, None))
    np.testing.assert_allclose(si.cpu().numpy(), g[f"{tag}_img_short"], **TOL)
    np.testing.assert_allclose(st.cpu().numpy(), g[f"{tag}_txt_short"], **TOL)
    tl = N.twdh_targets(tt(c["labels"]), tt(c["lc"]), tt(g[f"{tag}_rc_long"]))
    np.testing.assert_array_equal(tl.cpu().numpy(), g[f"{tag}_target_long"])
    ts = N.twdh_targets(tt(c["labels"]), tt(c["sc"]), tt(g[f"{tag}_rc_short"]))
    nce, quan = N.twdh_loss(li, lt, tl)
    nce_s, quan_s = N.twdh_loss(si, st, ts)
    loss = float(nce) + 0.5 * float(quan) + 0.3 * float(nce_s) + 0.3 * float(quan_s)
    assert abs(loss - float(g[f"{tag}_loss"])) < 1e-4 * max(1.0, abs(loss))
    code = N.pair_argmax_codes(li)
    ref = g[f"{tag}_img_long"].reshape(B, K, 2)
    safe = np.abs(ref[..., 0] - ref[..., 1]) > 1e-3
    assert np.array_equal(code.cpu().numpy()[safe], g[f"{tag}_img_code"][safe])


@pytest.mark.parametrize("B,K,C", DNPH_CASES)
def test_dnph_loss(golden, B, K, C):
    import cmh_native as N
    g = golden("dnph.npz")
    c = dnph_case(B, K, C)
    tag = c["tag"]
    args = [tt(c[k]) for k in ("hi", "ht", "pi", "pt", "lab", "prox")]
    _, loss1, _ = N.dnph_loss(*args)
    assert abs(float(loss1) - float(g[f"{tag}_loss1"])) < 1e-4 * max(1.0, abs(float(loss1)))
    total, _, _ = N.dnph_loss(*args, noise_img=tt(g[f"{tag}_noise_i"].astype(np.float32)),
                              noise_txt=tt(g[f"{tag}_noise_t"].astype(np.float32)))
    assert abs(float(total) - float(g[f"{tag}_step_loss"])) < 1e-4 * max(1.0, abs(float(total)))


def test_twdh_and_dnph_trainers_end_to_end(tmp_path, monkeypatch):
    """Registry -> trainer -> model -> native loss/valid for the two methods, on the synthetic dataset."""
    import argparse
    import sys
    import recipe
    import main
    import dataset.synthetic as ds
    ck = tmp_path / "clip.pt"
    sd = recipe.clip_state_dict(dict(recipe.CLIP_TINY, embed_dim=512), 7)      # heads are written for embedDim 512? no: any
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, ck)
    monkeypatch.setattr(ds, "SOT", 510); monkeypatch.setattr(ds, "EOT", 511)
    common = ["-clip-path", str(ck), "--save-dir", str(tmp_path), "--batch-size", "16", "--num-workers", "0",
              "--resolution", "64", "--max-words", "16", "--query-num", "24", "--train-num", "32",
              "--synthetic-size", "120", "--epochs", "0"]
    for method, K in (("TwDH", 32), ("DNPH", 16)):
        monkeypatch.setattr(sys, "argv", ["main.py"] + common)
        args = argparse.Namespace(method=method, dataset="synthetic", output_dim=K, is_train=True)
        tr = main.trainers[method](args, 0)
        tr.change_state(mode="valid")
        res = tr.valid(0)
        image, text, label, index = next(iter(tr.train_loader))
        with torch.no_grad():
            outs = tr.model(image.to(DEV), text.to(DEV))
            if method == "TwDH":
                il, ish, tl, tsh, lc, sc = outs
                assert il.shape == (16, 2 * K) and set(ish) == {"16"} and ish["16"].shape == (16, 32)
                loss = tr.compute_loss(il, tl, ish, tsh, label, None, lc, sc)
                assert isinstance(res, dict) and "long" in res and "16" in res
            else:
                hi, pi, ht, pt = outs
                loss = tr.compute_loss(hi, pi, ht, pt, label.to(DEV).float())
        assert torch.isfinite(loss).item()
        if method == "DNPH":
            # one real epoch: tape forward -> DNPH_out + noise term -> backward through heads, classifiers and both towers -> BertAdam
            for grp in tr.optimizer.param_groups:
                grp["t_total"] = 4
            before = {n: p.detach().clone() for n, p in tr.model.named_parameters()}
            prox0 = tr.DNPH.proxies.detach().clone()
            tr.train_epoch(0)
            same = {n for n, p in tr.model.named_parameters() if torch.equal(p.detach(), before[n])}
            assert same == {"clip.logit_scale"}, sorted(same)[:5]
            assert all(torch.isfinite(p).all() for p in tr.model.parameters())
            assert torch.equal(prox0, tr.DNPH.proxies.detach())        # upstream never steps its proxy SGD

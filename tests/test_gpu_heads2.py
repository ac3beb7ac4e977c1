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
        if method == "TwDH":
            for grp in tr.optimizer.param_groups:
                grp["t_total"] = 4
            before = {n: p.detach().clone() for n, p in tr.model.named_parameters()}
            tr.train_epoch(0)
            same = {n for n, p in tr.model.named_parameters() if torch.equal(p.detach(), before[n])}
            assert same <= {"clip.logit_scale", "image_hash.fc.weight", "image_hash.fc.bias", "text_hash.fc.weight", "text_hash.fc.bias"}, sorted(same)[:5]
            assert all(torch.isfinite(p).all() for p in tr.model.parameters())
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


@pytest.mark.parametrize("B,K,S,C", TWDH_CASES)
def test_twdh_step_gradients_match_reference_goldens(golden, B, K, S, C):
    """Both ModalityHash heads (BatchNorm1d with batch statistics / LayerNorm), short codes through `trans` and the trainer's
    compute_loss (quan_alpha 0.5, low_rate 0.3) differentiated on the GPU against the gradients torch autograd produced for the
    REFERENCE's modules (tests/golden/make_golden10.py); the BatchNorm running statistics move like upstream's."""
    from types import SimpleNamespace
    from train.TwDH.hash_train import TwDHTrainer
    from backward_ops import LinearAct, PairSoftmax
    import cmh_native as N
    g, g2 = golden("twdh_grads.npz"), golden("twdh.npz")
    c = twdh_case(B, K, S, C)
    tag = c["tag"]
    hi, ht = _head(K, c["p_img"], False), _head(K, c["p_txt"], True)
    fi, ft = tt(c["feat_i"]).requires_grad_(), tt(c["feat_t"]).requires_grad_()
    li, lt = hi(fi), ht(ft)
    trans_t = tt(c["trans"]).t().contiguous()
    zb = torch.zeros(2 * S, device=DEV)
    si = PairSoftmax.apply(LinearAct.apply(li, trans_t, zb, N.ACT_NONE, None, 0.0))
    st = PairSoftmax.apply(LinearAct.apply(lt, trans_t, zb, N.ACT_NONE, None, 0.0))
    me = SimpleNamespace(args=SimpleNamespace(quan_alpha=0.5, low_rate=0.3), rank=0)
    me.hash_center_multilables = lambda l, cen, rc=None: TwDHTrainer.hash_center_multilables(me, l, cen, rc)
    loss = TwDHTrainer.compute_loss(me, li, lt, {str(S): si}, {str(S): st}, tt(c["labels"]), None, tt(c["lc"]), {str(S): tt(c["sc"])},
                                    random_centers={"long": tt(g2[f"{tag}_rc_long"]), str(S): tt(g2[f"{tag}_rc_short"])})
    assert abs(float(loss.detach()) - float(g[f"{tag}_loss"])) < 1e-4 * max(1.0, abs(float(loss.detach())))
    loss.backward()
    sub = lambda a: a[::32] if a.size > 20000 else a
    # B12_K16 has saturated pairs (p within 1e-3 of 0 / 1): BCE's (p - t) / (p (1 - p)) followed by the softmax backward cancels
    # in float32, and the reference's own float32 gradients sit 1.7 % (of the tensor's maximum) from the float64 truth there;
    # B32_K128 is well conditioned and must agree tightly.
    frac = 0.08 if K == 16 else 2e-5
    for side, h, f in (("img", hi, fi), ("txt", ht, ft)):
        got = {"gfeat": f.grad, "g_in_b": h.atten.in_proj_bias.grad, "g_out_w": h.atten.out_proj.weight.grad,
               "g_out_b": h.atten.out_proj.bias.grad, "g_norm_w": h.norm.weight.grad, "g_norm_b": h.norm.bias.grad,
               "g_fc2_w": h.fc2.weight.grad, "g_fc2_b": h.fc2.bias.grad}
        giw = h.atten.in_proj_weight.grad.cpu().numpy()
        assert np.abs(giw[:1024]).max() <= 1e-7                      # Q / K rows: zero gradient (upstream: 1e-9 of rounding noise)
        want = g[f"{tag}_{side}_g_in_w"]
        np.testing.assert_allclose(sub(giw[1024:]), want, rtol=2e-3, atol=frac * np.abs(want).max(), err_msg=f"{side} in_w")
        for name, v in got.items():
            a, want = v.cpu().numpy(), g[f"{tag}_{side}_{name}"]
            if name != "gfeat":
                a = sub(a)
            np.testing.assert_allclose(a, want, rtol=2e-3, atol=max(frac * np.abs(want).max(), 1e-7), err_msg=f"{side} {name}")
    np.testing.assert_allclose(hi.norm.running_mean.cpu().numpy(), g[f"{tag}_img_running_mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(hi.norm.running_var.cpu().numpy(), g[f"{tag}_img_running_var"], rtol=1e-5, atol=1e-6)
    assert int(hi.norm.num_batches_tracked) == 1


@pytest.mark.parametrize("M,N,K,act", [(12544, 64, 512, "tanh"), (2051, 30, 512, "none"), (4096, 128, 1024, "relu"), (3000, 16, 260, "tanh")])
def test_linear_act_many_rows_has_the_one_row_kernels_bits(M, N, K, act):
    """cmh_linear_act on >= 2048 rows takes the kernel that keeps eight rows' inputs in registers per block (MITH's token-level
    concept similarities, model/MITH.py:441-442 upstream); every output is formed by the same lane partition, FMA chain and
    butterfly as in the one-row-per-block kernel, so slices of fewer rows (which take that kernel) give the same bits."""
    import cmh_native as N_
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    a = {"tanh": N_.ACT_TANH, "none": N_.ACT_NONE, "relu": N_.ACT_RELU}[act]
    full = N_.linear_act(x, w, b, a)
    parts = torch.cat([N_.linear_act(x[i:i + 1000].contiguous(), w, b, a) for i in range(0, M, 1000)])
    assert torch.equal(full, parts)
    ref = x.double() @ w.double().t() + b.double()
    ref = torch.tanh(ref) if act == "tanh" else (ref.clamp_min(0) if act == "relu" else ref)
    torch.testing.assert_close(full.double(), ref, rtol=1e-4, atol=1e-5)

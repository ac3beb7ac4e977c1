"""Backward building blocks (csrc/backward.hip, attention_bwd.hip, GEMM epilogues) against fp64 torch-CPU autograd of the
same op."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("R,C", [(1, 1), (63, 65), (130, 768), (1000, 3072)])
@pytest.mark.parametrize("sdt,ddt", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16), (torch.float32, torch.bfloat16)])
def test_transpose(R, C, sdt, ddt):
    import backward_ops as B
    x = torch.randn(R, C).to(sdt)
    out = B.transpose(x.to(DEV), ddt)
    assert out.shape == (C, R) and out.dtype == ddt
    assert torch.equal(out.cpu(), x.t().contiguous().to(ddt))


@pytest.mark.parametrize("R,C,dt", [(1, 5, torch.float32), (129, 768, torch.float32), (12800, 2304, torch.bfloat16)])
def test_colsum(R, C, dt):
    import backward_ops as B
    x = torch.randn(R, C).to(dt)
    out = B.colsum(x.to(DEV)).cpu().double()
    ref = x.double().sum(0)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-4 * R ** 0.5)


@pytest.mark.parametrize("M,d", [(3, 128), (100, 512), (1000, 768), (33, 1024)])
@pytest.mark.parametrize("xdt,gdt", [(torch.float32, torch.float32), (torch.float16, torch.bfloat16)])
def test_layernorm_backward(M, d, xdt, gdt):
    import backward_ops as B
    g = torch.Generator().manual_seed(M + d)
    x = (torch.randn(M, d, generator=g) * 2 + 0.5).to(xdt)
    dy = torch.randn(M, d, generator=g).to(gdt)
    gamma = torch.randn(d, generator=g)
    beta = torch.randn(d, generator=g)
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (d,), gr, br, 1e-5).backward(dy.double())
    base = torch.randn(M, d, generator=g)
    dx, dg, db = B.layernorm_backward(x.to(DEV), dy.to(DEV), gamma.to(DEV), dx=base.clone().to(DEV))
    torch.testing.assert_close(dx.cpu().double(), base.double() + xr.grad, rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(dg.cpu().double(), gr.grad, rtol=2e-4, atol=2e-4 * M ** 0.5)
    torch.testing.assert_close(db.cpu().double(), br.grad, rtol=2e-4, atol=2e-4 * M ** 0.5)
    dx2, _, _ = B.layernorm_backward(x.to(DEV), dy.to(DEV), gamma.to(DEV))
    torch.testing.assert_close(dx2.cpu().double(), xr.grad, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_quick_gelu_forward(dt):
    import backward_ops as B
    x = (torch.randn(10007) * 3).to(dt)
    out = B.quick_gelu(x.to(DEV)).cpu().double()
    ref = x.double() * torch.sigmoid(1.702 * x.double())
    torch.testing.assert_close(out, ref, rtol=1e-2 if dt == torch.bfloat16 else 1e-5, atol=1e-2 if dt == torch.bfloat16 else 1e-6)


@pytest.mark.parametrize("B,T,d,causal,kpm", [(2, 5, 128, 0, False), (3, 50, 192, 0, False), (2, 77, 128, 1, False),
                                              (2, 16, 128, 1, True), (1, 128, 64, 0, True)])
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_attention_backward(B, T, d, causal, kpm, mode):
    import backward_ops as Bo
    import cmh_native as N
    g = torch.Generator().manual_seed(B * 100 + T + d)
    qkv = torch.randn(B * T, 3 * d, generator=g)
    dout = torch.randn(B * T, d, generator=g)
    mask = None
    if kpm:
        mask = torch.zeros(B, T, dtype=torch.bool)
        mask[:, T - 3:] = True                       # trailing pads; position 0 always visible (EOT/causal safe)
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    qd, dd = qkv.to(dt), dout.to(dt)
    # fp64 reference on the same (rounded) operands
    H = d // 64
    x = qd.double().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4).requires_grad_(True)     # [3, B, H, T, 64]
    q, k, v = x[0], x[1], x[2]
    s = q @ k.transpose(-1, -2) / 8.0
    if causal:
        s = s + torch.full((T, T), float("-inf"), dtype=torch.float64).triu(1)
    if mask is not None:
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    o = torch.softmax(s, -1) @ v                                                          # [B, H, T, 64]
    o2 = o.permute(0, 2, 1, 3).reshape(B * T, d)
    o2.backward(dd.double())
    ref = x.grad.permute(1, 3, 0, 2, 4).reshape(B * T, 3 * d)
    o_gpu = N.attention(qd.to(DEV), B, T, causal, None if mask is None else mask.to(DEV))
    dqkv = Bo.attention_backward(qd.to(DEV), o_gpu, dd.to(DEV), B, T, causal, None if mask is None else mask.to(DEV))
    tol = dict(rtol=3e-2, atol=3e-2) if mode == "bf16" else dict(rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dqkv.cpu().double(), ref, **tol)


@pytest.mark.parametrize("M,N,K", [(160, 256, 64), (333, 512, 128), (1000, 3072, 768)])
@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_gemm_epilogue_mul_dquickgelu(M, N, K, mode):
    """dgrad through c_fc's QuickGELU: out = (dY . W^T) * d/dv[v sigmoid(1.702 v)] at the saved pre-activation."""
    import ctypes as C
    import cmh_native as Nn
    g = torch.Generator().manual_seed(M + N + K)
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    x = torch.randn(M, K, generator=g).to(dt)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dt)
    pre = (torch.randn(M, N, generator=g) * 2).to(dt)
    v = pre.double()
    sg = torch.sigmoid(1.702 * v)
    ref = (x.double() @ w.double().t()) * (sg * (1 + 1.702 * v * (1 - sg)))
    xd, wd, pd = x.to(DEV), w.to(DEV), pre.to(DEV)
    out = torch.empty(M, N, dtype=dt, device=DEV)
    epi = 1024 | (8 if mode == "bf16" else 0)
    Nn.check(Nn.lib().cmh_linear_gemm(Nn.BF16 if mode == "bf16" else Nn.F32, Nn.ptr(xd), Nn.ptr(wd), None, Nn.ptr(pd), Nn.ptr(out),
                                      M, N, K, epi, Nn.stream_ptr(xd.device)), "gemm")
    tol = dict(rtol=1e-2, atol=1e-2) if mode == "bf16" else dict(rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(out.cpu().double(), ref, **tol)


def test_tower_gradients_match_reference_autograd(golden):
    """L = sum(encode_image * Gi) + sum(encode_text * Gt): every parameter gradient of both towers against the gradients
    torch autograd produced on the REFERENCE's CLIP (tests/golden/make_golden5.py), f32 mode."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden5 as mg
    import recipe
    from test_gpu_clip import _clip
    g = golden("clip_tiny_grads.npz")
    cfg, seed, B, L = recipe.CLIP_TINY, 7, 3, 16
    clip = _clip(cfg, seed, "f32")          # same weights as the reference model of make_golden5.py (no fp16 round trip)
    image = torch.from_numpy(recipe.images(B, cfg["image_resolution"], seed)).to(DEV)
    text = torch.from_numpy(recipe.captions(B, L, cfg["vocab_size"], seed)).to(DEV)
    gi, gt = mg.cotangents(B, cfg["embed_dim"], 23)
    fi = clip.encode_image(image)
    ft = clip.encode_text(text)
    np.testing.assert_allclose(fi.detach().cpu().numpy(), g["img_feat"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(ft.detach().cpu().numpy(), g["txt_feat"], rtol=1e-4, atol=1e-4)
    ((fi * gi.to(DEV)).sum() + (ft * gt.to(DEV)).sum()).backward()
    params = dict(clip.named_parameters())
    worst = 0.0
    for name in [str(n) for n in g["names"]]:
        got = params[name].grad
        assert got is not None, name
        ref, norm = g["g_" + name], float(g["n_" + name])
        a = mg.cut(got.cpu().numpy())
        assert abs(float(got.double().norm()) - norm) <= 2e-4 * max(norm, 1e-3), (name, float(got.double().norm()), norm)
        err = np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-6)
        worst = max(worst, err)
        assert err < 5e-4, (name, err)
    print(f"tower gradients: worst relative-to-max error {worst:.2e} over {len(g['names'])} tensors")


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_pooled_tail_of_the_training_towers_matches_the_full_path(mode):
    """The training forward / backward carry only the pooled rows through the last block's row-wise tail
    (csrc/encoders_bwd.hip: block_forward_train / block_backward with pooled_rows).  Against the full-size path
    (cmh_set_pooled_tail(0)): identical features, every parameter gradient equal up to the summation order of the weight
    gradients (B rows instead of B*T rows of which all but B are zero).  Ragged captions: the text tower is packed."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import recipe
    import cmh_native as Nn
    from test_gpu_clip import _clip
    cfg, seed, B, L = recipe.CLIP_TINY, 7, 5, 16
    image = torch.from_numpy(recipe.images(B, cfg["image_resolution"], seed)).to(DEV)
    text = torch.from_numpy(recipe.captions(B, L, cfg["vocab_size"], seed)).to(DEV)
    g = torch.Generator().manual_seed(3)
    gi, gt = torch.randn(B, cfg["embed_dim"], generator=g).to(DEV), torch.randn(B, cfg["embed_dim"], generator=g).to(DEV)
    res = {}
    try:
        for on in (False, True):
            Nn.set_pooled_tail(on)
            clip = _clip(cfg, seed, mode)
            fi, ft = clip.encode_image(image), clip.encode_text(text)
            ((fi * gi).sum() + (ft * gt).sum()).backward()
            res[on] = (fi.detach().clone(), ft.detach().clone(), {n: p.grad.detach().clone() for n, p in clip.named_parameters() if p.grad is not None})
    finally:
        Nn.set_pooled_tail(True)
    assert torch.equal(res[False][0], res[True][0]) and torch.equal(res[False][1], res[True][1])
    assert res[False][2].keys() == res[True][2].keys() and len(res[True][2]) > 50
    tol = 2e-5 if mode == "f32" else 4e-3
    for name, ref in res[False][2].items():
        got = res[True][2][name]
        err = float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-6)
        assert err < tol, (name, err)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_tower_backward_in_parts_equals_one_call(mode):
    """cmh_vit_backward_part / cmh_text_backward_part (model/base/train_ops.py runs each tower's backward as PARTS calls so that a
    data-parallel trainer can send every bucket of gradients while the next part runs): the same gradients as ONE call, bit for
    bit, all of them views of one flat buffer, and the bucket sink sees every parameter exactly once, last layers first."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import recipe
    from model.base import train_ops as T
    from test_gpu_clip import _clip
    cfg, seed, B, L = recipe.CLIP_TINY, 7, 5, 16
    image = torch.from_numpy(recipe.images(B, cfg["image_resolution"], seed)).to(DEV)
    text = torch.from_numpy(recipe.captions(B, L, cfg["vocab_size"], seed)).to(DEV)
    g = torch.Generator().manual_seed(3)
    gi, gt = torch.randn(B, cfg["embed_dim"], generator=g).to(DEV), torch.randn(B, cfg["embed_dim"], generator=g).to(DEV)
    res, sunk = {}, []
    old_parts, old_sink = T.PARTS, T.BUCKET_SINK
    try:
        for parts in (1, 2):
            T.PARTS = parts
            T.BUCKET_SINK = (lambda flat, params, views: sunk.append((flat.numel(), [id(p) for p in params],
                                                                       sum(v.numel() for v in views)))) if parts == 2 else None
            clip = _clip(cfg, seed, mode)
            ((clip.encode_image(image) * gi).sum() + (clip.encode_text(text) * gt).sum()).backward()
            res[parts] = {n: p.grad for n, p in clip.named_parameters() if p.grad is not None}
            if parts == 2:
                names = {id(p): n for n, p in clip.named_parameters()}
    finally:
        T.PARTS, T.BUCKET_SINK = old_parts, old_sink
    assert res[1].keys() == res[2].keys() and len(res[1]) > 50
    for n in res[1]:
        if n == "token_embedding.weight":      # the one atomic scatter of the backward pass: sums in arrival order
            torch.testing.assert_close(res[1][n], res[2][n], rtol=1e-4, atol=1e-5 * float(res[1][n].abs().max()))
        else:
            assert torch.equal(res[1][n], res[2][n]), n
        assert res[2][n].untyped_storage().nbytes() > 4 * res[2][n].numel(), n      # lives inside its tower's flat buffer
    assert len(sunk) == 4                                         # 2 towers x 2 parts (the tiny towers have 2 blocks)
    covered = [i for _, ids, _ in sunk for i in ids]
    assert len(covered) == len(set(covered)) == len(res[2])       # every parameter in exactly one bucket
    for numel, ids, vsum in sunk:
        assert numel == vsum                                      # the slice holds exactly its parameters' gradients
    first = [names[i] for i in sunk[0][1]] + [names[i] for i in sunk[2][1]]
    assert any("resblocks.1." in n for n in first) and not any("resblocks.0." in n for n in first)     # last block first
    assert any("ln_post" in n or "ln_final" in n for n in first)                                       # with the head
    last = [names[i] for i in sunk[1][1]] + [names[i] for i in sunk[3][1]]
    assert any("positional_embedding" in n for n in last)


@pytest.mark.parametrize("B,K,C,alpha", [(8, 16, 5, 0.8), (64, 64, 24, 0.8), (256, 64, 80, 0.8), (32, 128, 10, 0.0)])
def test_hyp_loss_backward(B, K, C, alpha):
    """d HyP / d(x, y, proxies) against torch autograd (fp64) of the reference formula (train/DSPH/loss.py:22-72)."""
    import torch.nn.functional as F
    from backward_ops import HypLoss
    g = torch.Generator().manual_seed(B + K + C)
    x = torch.tanh(torch.randn(B, K, generator=g))
    y = torch.tanh(torch.randn(B, K, generator=g))
    label = (torch.rand(B, C, generator=g) < 0.15).float()
    label[0] = 0
    label[1, :3] = 1
    prox = torch.randn(C, K, generator=g)
    thr = 0.05

    def ref(x, y, p):
        cos = F.normalize(x, dim=1) @ F.normalize(p, dim=1).T
        cos_t = F.normalize(y, dim=1) @ F.normalize(p, dim=1).T
        lab = label.double()
        P, Nn = (lab != 0).sum(), (lab == 0).sum()
        tot = ((1 - cos)[lab == 1].sum() + (1 - cos_t)[lab == 1].sum()) / P + (F.relu(cos - thr)[lab == 0].sum() + F.relu(cos_t - thr)[lab == 0].sum()) / Nn
        if alpha > 0:
            idx = lab.sum(1) > 1
            l_ = lab[idx]
            cs = l_ @ l_.T
            if (cs == 0).sum() > 0:
                xn, tn = F.normalize(x[idx], dim=1), F.normalize(y[idx], dim=1)
                Z = (cs == 0).sum()
                for s in (xn @ xn.T, tn @ tn.T, xn @ tn.T):
                    tot = tot + (alpha * F.relu(s - thr))[cs == 0].sum() / Z
        return tot
    xr, yr, pr = (t.double().requires_grad_(True) for t in (x, y, prox))
    (ref(xr, yr, pr) * 1.7).backward()
    xd, yd, pd = (t.to(DEV).requires_grad_(True) for t in (x, y, prox))
    loss = HypLoss.apply(xd, yd, label.to(DEV), pd, thr, alpha)
    (loss * 1.7).backward()
    assert abs(float(loss) - float(ref(x.double(), y.double(), prox.double()))) < 1e-5
    for a, r in ((xd.grad, xr.grad), (yd.grad, yr.grad), (pd.grad, pr.grad)):
        torch.testing.assert_close(a.cpu().double(), r, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("use_mask", [False, True])
def test_linear_act_backward(act, use_mask):
    from backward_ops import LinearAct
    g = torch.Generator().manual_seed(act * 2 + use_mask)
    M, Nn, K = 37, 64, 512
    x, w, b = torch.randn(M, K, generator=g), torch.randn(Nn, K, generator=g) * 0.05, torch.randn(Nn, generator=g)
    mask = (torch.rand(M, Nn, generator=g) >= 0.2).float() if use_mask else None
    dy = torch.randn(M, Nn, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    z = xr @ wr.T + br
    if mask is not None:
        z = z * mask.double() / 0.8
    yref = torch.tanh(z) if act == 1 else (torch.relu(z) if act == 2 else z)
    yref.backward(dy.double())
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    out = LinearAct.apply(xd, wd, bd, act, None if mask is None else mask.to(DEV), 0.2)
    out.backward(dy.to(DEV))
    torch.testing.assert_close(out.detach().cpu().double(), yref.detach(), rtol=1e-4, atol=1e-5)
    for a, r in ((xd.grad, xr.grad), (wd.grad, wr.grad), (bd.grad, br.grad)):
        torch.testing.assert_close(a.cpu().double(), r, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,K,C,alpha,p", [(32, 64, 24, 0.8, 0.15), (48, 16, 80, 0.8, 0.05), (16, 128, 21, 0.0, 0.2), (8, 32, 24, 0.8, 0.04)])
def test_hyp_loss_gradients_match_reference_goldens(golden, B, K, C, alpha, p):
    """The gradients the REFERENCE's HyP produced (tests/golden/make_golden.py::gen_loss_dsph)."""
    import recipe
    from backward_ops import HypLoss
    g = golden("loss_dsph.npz")
    tag, seed = f"B{B}_K{K}_C{C}", 21
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    x = torch.tanh(t(recipe.features(B, K, seed, f"dsph_x_{tag}"))).requires_grad_()
    y = torch.tanh(t(recipe.features(B, K, seed, f"dsph_y_{tag}"))).requires_grad_()
    prox = t(recipe.features(C, K, seed, f"dsph_prox_{tag}")).requires_grad_()
    lab = t(recipe.labels(B, C, seed, p=p, tag=f"dsph_lab_{tag}"))
    loss = HypLoss.apply(x, y, lab, prox, float(g[f"{tag}_threshold"]), alpha)
    loss.backward()
    assert abs(float(loss.detach()) - float(g[f"{tag}_loss"])) < 1e-4
    for a, name in ((x.grad, "gx"), (y.grad, "gy"), (prox.grad, "gprox")):
        np.testing.assert_allclose(a.cpu().numpy(), g[f"{tag}_{name}"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("M,O,I", [(15, 128, 128), (1000, 256, 128), (4096, 768, 768), (12800, 3072, 768), (19712, 512, 2048),
                                   (10499, 1536, 512), (777, 128, 256), (64, 128, 256)])
@pytest.mark.parametrize("mode", ["f32", "bf16", "bf16_tn"])
def test_linear_wgrad(M, O, I, mode):
    """dW = dY^T X, db = sum dY: padded K (M not a multiple of 64), plain and split-K launches.  "bf16_tn": both operands bf16
    and row-major as the backward pass holds them -> the TN GEMM (transposing LDS reads, K tail zero-filled in the kernel) when
    I % 256 == 0 and O % 128 == 0, the transposing path otherwise; products of bf16 values accumulate in f32 either way."""
    import backward_ops as B
    g = torch.Generator().manual_seed(M + O + I)
    dy = torch.randn(M, O, generator=g)
    x = torch.randn(M, I, generator=g)
    if mode == "bf16_tn":
        dyd, xd = dy.bfloat16().to(DEV), x.bfloat16().to(DEV)
        ref_w = dy.bfloat16().double().t() @ x.bfloat16().double()
        tol = dict(rtol=1e-3, atol=2e-3 * M ** 0.5)            # only the f32 summation order differs from the fp64 statement
        dw, db = B.linear_wgrad(dyd, xd, "bf16")
        torch.testing.assert_close(dw.cpu().double(), ref_w, **tol)
        torch.testing.assert_close(db.cpu().double(), dy.bfloat16().double().sum(0), rtol=1e-5, atol=1e-4 * M ** 0.5)
        return
    if mode == "bf16":
        dyd, xd = dy.to(DEV), x.bfloat16().to(DEV)             # f32 gradient stream x bf16 activation, as the towers call it
        ref_w = dy.bfloat16().double().t() @ x.bfloat16().double()
        tol = dict(rtol=2e-2, atol=2e-2 * M ** 0.5)
    else:
        dyd, xd = dy.to(DEV), x.to(DEV)
        ref_w = dy.double().t() @ x.double()
        tol = dict(rtol=1e-4, atol=1e-4 * M ** 0.5)
    dw, db = B.linear_wgrad(dyd, xd, mode)
    torch.testing.assert_close(dw.cpu().double(), ref_w, **tol)
    torch.testing.assert_close(db.cpu().double(), dy.double().sum(0), rtol=1e-5, atol=1e-4 * M ** 0.5)


def test_bf16_mode_gradients_track_f32_mode():
    """The bf16 training path (bf16 GEMM operands, fp16 residual stream, MFMA attention backward, split-K wgrad) against the
    f32 path on the same weights: per-tensor cosine > 0.99 and norms within 5 % (width 256 so that every bf16-only code path
    is taken; 64 samples x 50 / 16 tokens so that the wgrad contraction is long enough to be split)."""
    from model.base.model import CLIP
    cfg = dict(embed_dim=128, image_resolution=224, vision_layers=2, vision_width=256, vision_patch_size=32, context_length=16,
               vocab_size=512, transformer_width=256, transformer_heads=4, transformer_layers=2)
    torch.manual_seed(3)
    m = CLIP(**cfg).to(DEV).float()
    B = 64
    img = torch.randn(B, 3, 224, 224, device=DEV)
    txt = torch.randint(1, 500, (B, 16), device=DEV)
    txt[:, -1] = 511
    gi, gt = torch.randn(B, 128, device=DEV), torch.randn(B, 128, device=DEV)
    grads = {}
    for mode in ("f32", "bf16"):
        m.set_gemm_dtype(mode)
        m.zero_grad(set_to_none=True)
        ((m.encode_image(img) * gi).sum() + (m.encode_text(txt) * gt).sum()).backward()
        grads[mode] = {n: p.grad.detach().double().flatten().cpu() for n, p in m.named_parameters() if p.grad is not None}
    worst = (1.0, "")
    for n, a in grads["f32"].items():
        b = grads["bf16"][n]
        cos = float(a @ b / (a.norm() * b.norm() + 1e-30))
        worst = min(worst, (cos, n))
        assert cos > 0.99, (n, cos)
        assert abs(float(b.norm() / (a.norm() + 1e-30)) - 1.0) < 0.05, (n, float(a.norm()), float(b.norm()))
    print(f"bf16 vs f32 gradients: worst cosine {worst[0]:.5f} ({worst[1]})")


def test_block_wgrads_in_one_launch_match_the_four_launches(monkeypatch):
    """csrc/encoders_bwd.hip block_backward: the four weight gradients of a block as ONE multi-problem TN launch (the default where
    the shapes allow it) against the four launches of rounds 1-3 (CMH_WGRAD_MULTI=0): the same products summed in another split
    of the batch rows - equal to f32 rounding of the partial sums - and both track the f32 mode.  3 layers, width 256, 128 samples so
    that two blocks per tower take the path (the last block runs its pooled tail)."""
    from model.base.model import CLIP
    cfg = dict(embed_dim=128, image_resolution=224, vision_layers=3, vision_width=256, vision_patch_size=32, context_length=40,
               vocab_size=512, transformer_width=256, transformer_heads=4, transformer_layers=3)
    torch.manual_seed(5)
    m = CLIP(**cfg).to(DEV).float()
    B = 128
    img = torch.randn(B, 3, 224, 224, device=DEV)
    txt = torch.randint(1, 500, (B, 40), device=DEV)
    txt[:, -1] = 511
    gi, gt = torch.randn(B, 128, device=DEV), torch.randn(B, 128, device=DEV)

    def grads_of(mode):
        m.set_gemm_dtype(mode)
        m.zero_grad(set_to_none=True)
        ((m.encode_image(img) * gi).sum() + (m.encode_text(txt) * gt).sum()).backward()
        return {n: p.grad.detach().double().flatten().cpu() for n, p in m.named_parameters() if p.grad is not None}

    import cmh_native as Nn
    Nn.set_grad_stream16(0)       # (round 5: the multi-launch blocks also carry the gradient stream as bf16 - its own test below; here the
    try:                          # two wgrad forms are compared on the SAME f32 stream, as before)
        monkeypatch.setenv("CMH_WGRAD_MULTI", "1")
        one = grads_of("bf16")
        monkeypatch.setenv("CMH_WGRAD_MULTI", "0")
        four = grads_of("bf16")
    finally:
        Nn.set_grad_stream16(-1)
    ref = grads_of("f32")
    moved = 0
    for n, a in four.items():
        b = one[n]
        scale = float(a.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 2e-4 * scale, (n, float((a - b).abs().max()), scale)
        moved += int(not torch.equal(a, b))
        cos = float(ref[n] @ b / (ref[n].norm() * b.norm() + 1e-30))
        assert cos > 0.99, (n, cos)
    assert moved > 0          # the path was taken: some weight gradient is summed in another order


def test_16bit_gradient_stream_tracks_the_f32_stream():
    """csrc/encoders_bwd.hip block_backward, round 5: in the bf16 mode's multi-launch blocks the residual-gradient stream between the
    LayerNorm backward kernels is the bf16 copy they write anyway (8 instead of 14 bytes per element and launch), the sum formed in f32
    and rounded once per LayerNorm.  Against the f32 stream (cmh_set_grad_stream16(0)) on the same weights, 6 layers per tower (5
    blocks each take the path, 10 roundings): every parameter gradient's cosine > 0.9995, norms within 1 %, and both versions sit at
    the same distance from the f32 MODE (the bf16 operands, not the stream, set that distance).  Reference: autograd through
    model/base/model.py:167-207."""
    from model.base.model import CLIP
    import cmh_native as Nn
    cfg = dict(embed_dim=128, image_resolution=224, vision_layers=6, vision_width=256, vision_patch_size=32, context_length=40,
               vocab_size=512, transformer_width=256, transformer_heads=4, transformer_layers=6)
    torch.manual_seed(11)
    m = CLIP(**cfg).to(DEV).float()
    B = 128
    img = torch.randn(B, 3, 224, 224, device=DEV)
    txt = torch.randint(1, 500, (B, 40), device=DEV)
    txt[:, -1] = 511
    gi, gt = torch.randn(B, 128, device=DEV), torch.randn(B, 128, device=DEV)

    def grads_of(mode):
        m.set_gemm_dtype(mode)
        m.zero_grad(set_to_none=True)
        ((m.encode_image(img) * gi).sum() + (m.encode_text(txt) * gt).sum()).backward()
        return {n: p.grad.detach().double().flatten().cpu() for n, p in m.named_parameters() if p.grad is not None}

    try:
        Nn.set_grad_stream16(1)
        s16 = grads_of("bf16")
        Nn.set_grad_stream16(0)
        s32 = grads_of("bf16")
    finally:
        Nn.set_grad_stream16(-1)
    ref = grads_of("f32")
    cosf = lambda a, b: float(a @ b / (a.norm() * b.norm() + 1e-30))
    worst, moved, gap = (1.0, ""), 0, 0.0
    for n, a in s32.items():
        b = s16[n]
        moved += int(not torch.equal(a, b))
        c = cosf(a, b)
        worst = min(worst, (c, n))
        assert c > 0.9995, (n, c)
        assert abs(float(b.norm() / (a.norm() + 1e-30)) - 1.0) < 0.01, (n, float(a.norm()), float(b.norm()))
        c16, c32 = cosf(ref[n], b), cosf(ref[n], a)
        assert c16 > 0.99, (n, c16)
        gap = max(gap, c32 - c16)
    assert moved > 0          # the path was taken
    assert gap < 2e-3, gap    # the 16-bit stream costs the gradients' agreement with the f32 mode next to nothing
    print(f"16-bit gradient stream vs f32 stream: worst cosine {worst[0]:.6f} ({worst[1]}); largest loss of cosine against the f32 mode {gap:.2e}")


@pytest.mark.parametrize("B,K,C,fn,lt", [(32, 16, 24, "euclidean", "l2"), (32, 16, 24, "cosine", "l2"), (24, 64, 24, "euclidean", "l1"),
                                         (24, 64, 80, "cosine", "l1")])
def test_dchmt_loss_gradients_match_reference_goldens(golden, B, K, C, fn, lt):
    """d our_loss / d(pair probabilities) as the REFERENCE's autograd produced them (make_golden.py::gen_loss_dchmt), through
    the pair softmax as well (chain: z -> softmax pairs -> loss)."""
    import recipe
    from backward_ops import DchmtLoss, PairSoftmax
    g = golden("loss_dchmt.npz")
    tag, seed = f"B{B}_K{K}_C{C}_{fn}_{lt}", 31
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    zi = (2 * t(recipe.features(B, 2 * K, seed, f"dchmt_zi_{tag}"))).requires_grad_()
    zt = (2 * t(recipe.features(B, 2 * K, seed, f"dchmt_zt_{tag}"))).requires_grad_()
    lab = t(recipe.labels(B, C, seed, tag=f"dchmt_lab_{tag}"))
    hi, ht = PairSoftmax.apply(zi), PairSoftmax.apply(zt)
    hi.retain_grad(); ht.retain_grad()
    loss = DchmtLoss.apply(hi, ht, lab, K, fn, lt, 0.5, 0.1)
    loss.backward()
    ref = float(g[f"{tag}_loss"])
    assert abs(float(loss.detach()) - ref) < 1e-4 * max(1.0, abs(ref))
    np.testing.assert_allclose(hi.grad.cpu().numpy(), g[f"{tag}_gi"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(ht.grad.cpu().numpy(), g[f"{tag}_gt"], rtol=1e-4, atol=1e-7)
    # pair softmax backward against fp64 autograd
    zr = zi.detach().cpu().double().requires_grad_(True)
    torch.softmax(zr.view(B, K, 2), -1).reshape(B, 2 * K).backward(hi.grad.cpu().double())
    torch.testing.assert_close(zi.grad.cpu().double(), zr.grad, rtol=1e-5, atol=1e-9)


def test_dchmt_training_trajectory_matches_reference(golden):
    """BASELINE configs[0] in miniature: 4 steps of the DCHMT loop (tape forward -> select heads -> our_loss -> backward through
    heads and towers -> fused BertAdam), loss of every step and the final parameters against the REFERENCE's own run on the CPU
    (tests/golden/make_golden6.py)."""
    import dchmtutil as du
    from backward_ops import DchmtLoss
    from model.DCHMT import HashLayer
    from model.base.optimization import BertAdam
    from test_gpu_clip import _clip
    g = golden("dchmt_traj.npz")
    clip = _clip(du.CFG, du.SEED, "f32")
    ih, th = HashLayer(du.CFG["embed_dim"], du.K), HashLayer(du.CFG["embed_dim"], du.K)
    du.fill_head(ih, 1)
    du.fill_head(th, 2)
    ih, th = ih.to(DEV), th.to(DEV)
    opt = BertAdam([{"params": [p for _, p in clip.named_parameters()], "lr": du.CLIP_LR}, {"params": ih.parameters(), "lr": du.OPT["lr"]},
                    {"params": th.parameters(), "lr": du.OPT["lr"]}], **du.OPT)
    losses = []
    for step in range(du.STEPS):
        img, txt, lab = du.batch(step)
        hi = ih.pair_probs(clip.encode_image(img.to(DEV)))
        ht = th.pair_probs(clip.encode_text(txt.to(DEV)))
        loss = DchmtLoss.apply(hi, ht, lab.to(DEV), du.K, du.LOSS["similarity_function"], du.LOSS["loss_type"], du.LOSS["vartheta"],
                               du.LOSS["sim_threshold"])
        losses.append(float(loss.detach()))
        opt.zero_grad()
        loss.backward()
        opt.step()
    np.testing.assert_allclose(np.array(losses), g["losses"], rtol=1e-4)
    named = dict(clip.named_parameters())
    named.update({"image_hash." + n: p for n, p in ih.named_parameters()})
    named.update({"text_hash." + n: p for n, p in th.named_parameters()})
    worst = 0.0
    for key in g.files:
        if not key.startswith("p_"):
            continue
        ref = g[key]
        a = du.cut(named[key[2:]].detach().cpu().numpy())
        err = np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-6)
        worst = max(worst, err)
        assert err < 1e-4, (key, err)
    print(f"DCHMT trajectory: losses {losses}, worst parameter error {worst:.2e} of max")


@pytest.mark.parametrize("B,K,C", [(16, 16, 21), (40, 128, 24)])
@pytest.mark.parametrize("with_noise", [True, False])
def test_dnph_loss_gradients_match_reference_goldens(golden, B, K, C, with_noise):
    """The gradients the REFERENCE's DNPH_out (+ the trainer's noise term) produced (tests/golden/make_golden7.py); without the
    noise rows, fp64 autograd of the same formula (train/DNPH_TOMM/loss.py:14-32)."""
    import torch.nn.functional as F
    from heads2util import dnph_case
    from backward_ops import DnphLoss
    c = dnph_case(B, K, C)
    tag = c["tag"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    hi, ht, pi, pt, prox = (t(c[k]).requires_grad_() for k in ("hi", "ht", "pi", "pt", "prox"))
    lab = t(c["lab"])
    if with_noise:
        g, gn = golden("dnph_grads.npz"), golden("dnph.npz")
        loss = DnphLoss.apply(hi, ht, pi, pt, lab, prox, t(gn[f"{tag}_noise_i"]), t(gn[f"{tag}_noise_t"]), 1.0, 0.1)
        (3.0 * loss).backward()                                  # exercises the upstream-gradient scaling too
        assert abs(float(loss.detach()) - float(g[f"{tag}_step_loss"])) < 1e-4 * max(1.0, abs(float(loss.detach())))
        want = {n: 3.0 * g[f"{tag}_{n}"] for n in ("ghi", "ght", "gpi", "gpt", "gprox")}
    else:
        loss = DnphLoss.apply(hi, ht, pi, pt, lab, prox, None, None, 1.0, 0.1)
        loss.backward()
        d = [torch.from_numpy(np.asarray(c[k], dtype=np.float64)).requires_grad_() for k in ("hi", "ht", "pi", "pt", "prox")]
        labd = torch.from_numpy(c["lab"]).double()
        la = torch.cat((labd, labd))
        D = torch.cdist(F.normalize(torch.cat((d[0], d[1])), p=2, dim=-1), F.normalize(d[4], p=2, dim=-1)) ** 2
        pl = torch.sum(-la * F.log_softmax(-(D + (la == 1).double()), 1), -1).mean()
        ce = F.cross_entropy(d[2], labd.argmax(-1)) + F.cross_entropy(d[3], labd.argmax(-1))
        (pl + ce).backward()
        assert abs(float(loss.detach()) - float((pl + ce).detach())) < 1e-4
        want = dict(zip(("ghi", "ght", "gpi", "gpt", "gprox"), (x.grad.numpy() for x in d)))
    for a, name in ((hi, "ghi"), (ht, "ght"), (pi, "gpi"), (pt, "gpt"), (prox, "gprox")):
        np.testing.assert_allclose(a.grad.cpu().numpy(), want[name], rtol=2e-4, atol=2e-6, err_msg=name)

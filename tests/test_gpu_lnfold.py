"""LayerNorm folded into the GEMMs around it (bf16 mode; csrc/gemm_wide.hip template parameter LN, csrc/encoders.hip run_block):
the residual GEMM before a LayerNorm leaves per-row sums of its fp16 output, the Linear after it reads the raw fp16 stream against
W * gamma and normalises in its epilogue.  Replaces `self.ln_1(x)` / `self.ln_2(x)` of model/base/model.py:191-196 as launches.
Checked here: the folded operands, the statistics, the product against a float64 LayerNorm + Linear, bit-identity across tile
heights / packed captions / the pooled tail, and the towers against the f32 parity mode (no worse than with LayerNorm launches)."""
import numpy as np
import pytest
import torch

import recipe

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_ln_fold_weight_matches_its_definition():
    import cmh_native as N
    g = torch.Generator().manual_seed(5)
    w = torch.randn(512, 768, generator=g) * 0.03
    gamma = 1.0 + 0.3 * torch.randn(768, generator=g)
    beta = 0.2 * torch.randn(768, generator=g)
    bias = torch.randn(512, generator=g)
    wf, bf, cf = N.ln_fold_weight(w.to(DEV), gamma.to(DEV), beta.to(DEV), bias.to(DEV))
    ref_w = (w * gamma).half()
    assert torch.equal(wf.cpu(), ref_w)                                     # one RNE rounding of the f32 product
    torch.testing.assert_close(bf.cpu().double(), bias.double() + w.double() @ beta.double(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(cf.cpu().double(), ref_w.double().sum(1), rtol=1e-5, atol=1e-5)
    wf2, bf2, cf2 = N.ln_fold_weight(w.to(DEV), gamma.to(DEV), beta.to(DEV), None)
    torch.testing.assert_close(bf2.cpu().double(), w.double() @ beta.double(), rtol=1e-5, atol=1e-5)
    assert torch.equal(wf2, wf) and torch.equal(cf2, cf)


@pytest.mark.parametrize("M,d,No", [(2500, 768, 2304), (333, 512, 2048), (2100, 768, 3072), (160, 256, 256)])
def test_folded_layernorm_gemm_pair_against_float64(M, d, No):
    """producer (x = fp16(a @ wo.T + bo + r) + row sums) -> consumer (LayerNorm(x) @ w.T + b, QuickGELU) against float64 on the
    producer's own fp16 output; the statistics against float64 sums of those fp16 values; partial tiles, 1-3 tiles per workgroup."""
    import cmh_native as N
    g = torch.Generator().manual_seed(M + d + No)
    a = torch.randn(M, d, generator=g).bfloat16()
    wo = (torch.randn(d, d, generator=g) * d ** -0.5).bfloat16()
    bo = torch.randn(d, generator=g)
    r = (2.0 * torch.randn(M, d, generator=g) + 0.7).half()
    r[:, 5] += 30.0                                                         # an outlier channel, as CLIP's residual stream has them
    x16, part = N.linear_gemm_ln_producer(a.to(DEV), wo.to(DEV), bo.to(DEV), r.to(DEV))
    plain = N.linear_gemm(a.to(DEV), wo.to(DEV), bias=bo.to(DEV), residual=r.to(DEV), out_f16=True)
    assert torch.equal(x16, plain)                                          # the statistics ride along: same output bits
    xd = x16.cpu().double()
    s1 = xd.view(M, d // 256, 256).sum(2).t()
    s2 = (xd * xd).view(M, d // 256, 256).sum(2).t()
    torch.testing.assert_close(part[:, :, 0].cpu().double(), s1, rtol=2e-6, atol=2e-4)
    torch.testing.assert_close(part[:, :, 1].cpu().double(), s2, rtol=2e-6, atol=2e-3)

    w = torch.randn(No, d, generator=g) * d ** -0.5
    b = torch.randn(No, generator=g)
    gamma = 1.0 + 0.3 * torch.randn(d, generator=g)
    beta = 0.2 * torch.randn(d, generator=g)
    wf, bf, cf = N.ln_fold_weight(w.to(DEV), gamma.to(DEV), beta.to(DEV), b.to(DEV))
    ln = torch.nn.functional.layer_norm(xd, (d,), gamma.double(), beta.double(), 1e-5)
    for act in (False, True):
        out = N.linear_gemm_ln_consumer(x16, part, wf, bf, cf, quickgelu=act)
        ref = ln @ w.double().t() + b.double()
        if act:
            ref = ref * torch.sigmoid(1.702 * ref)
        # bf16 output (2^-9 relative) on top of fp16 weights; tighter than the LayerNorm-launch path's bf16 operands allow
        torch.testing.assert_close(out.cpu().double(), ref, rtol=6e-3, atol=6e-3)
    # against the path it replaces: LayerNorm launch (bf16 out) + bf16 GEMM - the fold must not be the less accurate of the two
    h = N.layernorm(x16.float(), gamma.to(DEV), beta.to(DEV), out_bf16=True)
    old = N.linear_gemm(h, w.to(DEV).bfloat16(), bias=b.to(DEV), out_bf16=True)
    ref = ln @ w.double().t() + b.double()
    new = N.linear_gemm_ln_consumer(x16, part, wf, bf, cf)
    err_new = (new.cpu().double() - ref).abs().mean()
    err_old = (old.cpu().double() - ref).abs().mean()
    assert err_new <= 1.05 * err_old, (float(err_new), float(err_old))


@pytest.mark.parametrize("tower", ["image", "text"])
def test_towers_with_folded_layernorms(tower):
    """ViT-B/32 towers, batch large enough for the wide kernel (> 2048 rows): features with the fold against the f32 parity mode
    (same gates as the bf16 mode's own test) and against the LayerNorm-launch path; identical bits for every tile height, with and
    without the pooled tail, packed and dense captions."""
    import cmh_native as N
    from model.base.model import CLIP
    cfg = recipe.CLIP_VITB32
    torch.manual_seed(21)
    m = CLIP(**cfg).to(DEV).float()
    m.assume_frozen = True
    if tower == "image":
        x = torch.from_numpy(recipe.images(44, cfg["image_resolution"], 3)).to(DEV)       # 44 * 50 = 2200 rows
        enc = m.encode_image
    else:
        x = torch.from_numpy(recipe.captions(30, 77, cfg["vocab_size"], 4)).to(DEV)        # 30 * 77 = 2310 rows
        enc = m.encode_text
    with torch.no_grad():
        m.set_gemm_dtype("f32")
        ref = enc(x).clone()
        m.set_gemm_dtype("bf16")
        try:
            N.set_ln_fold(0)
            plain = enc(x).clone()
            N.set_ln_fold(1)
            fold = enc(x).clone()
            assert not torch.equal(fold, plain)                              # the fold really ran
            for rows in (96, 128, 160):
                N.lib().cmh_gemm_tuning(rows, -1)
                assert torch.equal(enc(x), fold), rows
            N.lib().cmh_gemm_tuning(-1, -1)
            N.set_pooled_tail(False)
            assert torch.equal(enc(x), fold)
            N.set_pooled_tail(True)
            if tower == "text":
                m.pack_text = not m.pack_text
                assert torch.equal(enc(x), fold)
                m.pack_text = not m.pack_text
        finally:
            N.set_ln_fold(-1)
            N.set_pooled_tail(True)
            N.lib().cmh_gemm_tuning(-1, -1)
    cos = torch.nn.functional.cosine_similarity
    c_fold, c_plain = cos(fold, ref).min().item(), cos(plain, ref).min().item()
    assert c_fold > 0.9995, c_fold
    assert (1 - c_fold) <= 1.5 * (1 - c_plain) + 1e-6, (c_fold, c_plain)
    flips = ((fold > 0) != (ref > 0)).float().mean().item()
    assert flips < 0.01, flips

"""fp8 (OCP e4m3) encoder mode, BASELINE configs[4]: quantisation kernels, the e4m3 GEMM with its fused dequantisation, and the
encoders run through it.  The reference has no fp8 path (its precision hook is convert_weights, model/base/model.py:391-412), so
the checker is the fp64 statement of the same quantised operands for the GEMM, and the f32 parity mode for the towers."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _e4m3_values():
    """all 256 byte patterns of OCP e4m3fn as f32 (NaN for 0x7f / 0xff)"""
    out = np.zeros(256, np.float32)
    for b in range(256):
        s, e, m = b >> 7, (b >> 3) & 15, b & 7
        if e == 15 and m == 7:
            v = np.nan
        elif e == 0:
            v = m / 8.0 * 2.0 ** -6
        else:
            v = (1 + m / 8.0) * 2.0 ** (e - 7)
        out[b] = -v if s else v
    return out


def test_fp8_quantise_roundtrip_is_ocp_e4m3():
    import cmh_native as N
    table = _e4m3_values()
    q = torch.arange(256, dtype=torch.uint8, device=DEV)
    back = N.fp8_dequantize(q).cpu().numpy()
    ok = ~np.isnan(table)
    assert np.array_equal(back[ok], table[ok]) and np.isnan(back[~ok]).all()
    assert np.nanmax(table) == 448.0                                  # e4m3fn (OCP), not e4m3fnuz (240)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4096, generator=g) * 3
    x[:4] = torch.tensor([1000.0, -1000.0, 0.0, 449.0])               # saturates instead of becoming NaN
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        xs = x.to(dt)
        q = N.fp8_quantize(xs.to(DEV), 0.5)
        y = N.fp8_dequantize(q, 0.5).cpu()
        ref = (xs.float() / 0.5).clamp(-448, 448).numpy()
        # round-to-nearest representable value
        finite = table[ok]
        nearest = finite[np.abs(ref[:, None] - finite[None, :]).argmin(1)]
        err = np.abs(y.numpy() / 0.5 - ref)
        assert (err <= np.abs(nearest - ref) + 1e-6).all()
    w = torch.randn(300, 512, generator=g) * torch.rand(300, 1, generator=g)
    w[7] = 0
    q, cs = N.fp8_quantize_weight(w.to(DEV))
    amax = w.abs().amax(1)
    torch.testing.assert_close(cs.cpu(), torch.where(amax > 0, amax / 448, torch.ones_like(amax)))
    deq = N.fp8_dequantize(q).cpu() * cs.cpu()[:, None]
    assert float((deq - w).abs().max()) <= float(amax.max()) / 16 + 1e-6     # 3 mantissa bits: half an ulp <= 2^-4 of the row maximum
    assert float(N.amax(w.to(DEV))) == float(w.abs().max())
    assert float(N.amax(w.bfloat16().to(DEV))) == float(w.bfloat16().float().abs().max())


@pytest.mark.parametrize("M,N,K", [(160, 256, 128), (1000, 2304, 768), (333, 512, 512), (10499, 512, 512), (777, 768, 3072), (3000, 1536, 512)])
def test_fp8_gemm_matches_fp64_of_the_quantised_operands(M, N, K):
    """Against fp64 of the SAME quantised operands.  Products of e4m3 values are exact in f32, but v_mfma_scale_f32_16x16x128_f8f6f4
    does NOT add them as an f32 chain: measured (tools/fp8_mfma_probe.py, profiles/r03_b_fp8_mfma_accumulation.txt) it sums the
    products of 8 consecutive k aligned to the largest of the eight and cut ~13 bits below it (a product 2^-17 of its group's
    maximum is dropped entirely), and only the group sums accumulate like f32.  The error of an output element is therefore bounded
    by ~2^-13 of (largest |x| of its row) x (largest |w| of its column) per group of 8 - which is what is asserted - not by f32
    rounding.  Every output type, every tile height; the asymmetric operands catch a transposed or permuted fragment layout."""
    import cmh_native as Nn
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * (torch.rand(N, 1, generator=g) + 0.1) * K ** -0.5
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g).half()
    ax = float(x.abs().max()) / 448
    x8 = Nn.fp8_quantize(x.to(DEV), ax)
    w8, cs = Nn.fp8_quantize_weight(w.to(DEV))
    xq = Nn.fp8_dequantize(x8, ax).cpu().double()
    wq = (Nn.fp8_dequantize(w8).cpu() * cs.cpu()[:, None]).double()
    base = xq @ wq.t()
    try:
        first = None
        for rows in (-1, 96, 128, 160):                 # (160 rows: only the e4m3-output launches take it, the others ignore the request)
            Nn.gemm_tuning(rows, -1)
            plain = Nn.linear_gemm_fp8(x8, w8, cs, ax)
            qg = Nn.linear_gemm_fp8(x8, w8, cs, ax, bias=b.to(DEV), quickgelu=True, out="bf16")
            rs = Nn.linear_gemm_fp8(x8, w8, cs, ax, bias=b.to(DEV), residual=r.to(DEV), out="f16")
            o8 = Nn.linear_gemm_fp8(x8, w8, cs, ax, bias=b.to(DEV), quickgelu=True, out="fp8", out_scale=0.05)
            if first is None:
                first = (plain, qg, rs, o8)
                bound = xq.abs().max(1).values[:, None] * wq.abs().max(1).values[None, :] * (K / 8) * 2.0 ** -13
                assert bool(((plain.cpu().double() - base).abs() <= bound + 1e-6 * base.abs()).all())
                torch.testing.assert_close(plain.cpu().double(), base, rtol=1e-4, atol=1e-4 * float(base.abs().max()))
                v = base + b.double()
                ref_q = v * torch.sigmoid(1.702 * v)
                torch.testing.assert_close(qg.cpu().double(), ref_q, rtol=1e-2, atol=1e-2)
                torch.testing.assert_close(rs.cpu().double(), base + b.double() + r.double(), rtol=1.5e-3, atol=2e-3)
                got8 = Nn.fp8_dequantize(o8, 0.05).cpu().double()
                ref8 = ref_q.clamp(-448 * 0.05, 448 * 0.05)
                assert float((got8 - ref8).abs().max()) <= float(ref8.abs().max()) / 16 + 1e-3        # one e4m3 rounding
                assert float(((got8 - ref8).abs() / (ref8.abs() + 0.02)).mean()) < 0.04
            else:
                for a_, b_ in zip((plain, qg, rs, o8), first):
                    assert torch.equal(a_, b_), rows
    finally:
        Nn.gemm_tuning(-1, -1)


def test_fp8_gemm_exact_integers():
    """x = one-hot rows, w = small integers: every output is a single exactly representable product."""
    import cmh_native as Nn
    M = N = 256
    K = 128
    x = torch.zeros(M, K)
    x[torch.arange(M), torch.arange(M) % K] = 1.0
    w = ((torch.arange(N * K) * 7 % 31) - 15).float().reshape(N, K)        # asymmetric, |w| <= 15: exact in e4m3
    x8 = Nn.fp8_quantize(x.to(DEV), 1.0)
    w8 = Nn.fp8_quantize(w.to(DEV), 1.0)
    out = Nn.linear_gemm_fp8(x8, w8, torch.ones(N, device=DEV), 1.0)
    ref = x @ w.t()
    assert torch.equal(out.cpu(), ref)


def _vitb32(seed, mode):
    import recipe
    from model.base.model import CLIP
    cfg = recipe.CLIP_VITB32
    m = CLIP(cfg["embed_dim"], cfg["image_resolution"], cfg["vision_layers"], cfg["vision_width"], cfg["vision_patch_size"],
             cfg["context_length"], cfg["vocab_size"], cfg["transformer_width"], cfg["transformer_heads"], cfg["transformer_layers"])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in recipe.clip_state_dict(cfg, seed).items()}, strict=True)
    return m.to(DEV).float().set_gemm_dtype(mode)


def test_fp8_towers_against_the_f32_parity_mode(golden):
    """ViT-B/32 with the blocks' GEMMs on e4m3 operands against the f32 mode (= the reference's model.float() arithmetic; the
    two golden rows were produced by the reference itself): feature cosine and the sign-flip rate of 64-bit codes, both stated.
    Measured on MI355X with the seeded random-init weights: see DESIGN 2 (fp8 row)."""
    import recipe
    g = golden("clip_vitb32.npz")
    seed = int(g["seed"])
    m = _vitb32(seed, "fp8")
    B = 32
    img = torch.from_numpy(recipe.images(B, 224, seed)).to(DEV)
    txt = torch.from_numpy(recipe.captions(B, 77, 49408, seed)).to(DEV)
    with torch.no_grad():
        cal = m.calibrate_fp8(image=img[:8], text=txt[:8])                # explicit calibration on a quarter of the batch
        assert len(cal) == 2 and cal[0].shape == (8, 512)
        f8 = (m.encode_image(img).cpu().numpy(), m.encode_text(txt).cpu().numpy())
        assert m.gemm_dtype == "fp8"
        m.set_gemm_dtype("f32")
        f32 = (m.encode_image(img).cpu().numpy(), m.encode_text(txt).cpu().numpy())
    # the first two rows of the f32 mode are the reference's goldens
    np.testing.assert_allclose(f32[0][:2], g["img_feat"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(f32[1][:2], g["txt_feat_L77"], rtol=1e-3, atol=1e-4)
    w, b = recipe.head_linear(512, 64, seed, "flip")
    for name, a, r in (("image", f8[0], f32[0]), ("text", f8[1], f32[1])):
        cos = (a * r).sum(-1) / np.linalg.norm(a, axis=-1) / np.linalg.norm(r, axis=-1)
        flips = float(np.mean(np.sign(a @ w.T + b) != np.sign(r @ w.T + b)))
        print(f"fp8 {name}: feature cosine min {cos.min():.5f} mean {cos.mean():.5f}; sign flips of 64-bit codes {flips:.4f} ({B} samples)")
        assert cos.min() > 0.99, cos.min()
        assert flips < 0.06, flips
    # inference only, and loud about it
    m.set_gemm_dtype("fp8")
    with pytest.raises(Exception):
        m.encode_text(txt[:2]).sum().backward()


def test_fp8_lazy_calibration_and_packed_text_equals_dense():
    """Without an explicit calibrate_fp8 the first batch calibrates; the packed text path (tokens after the EOT skipped) and the
    dense one share scales and arithmetic row by row, so the pooled features are identical bits."""
    import recipe
    m = _vitb32(5, "fp8")
    txt = torch.from_numpy(recipe.captions(6, 77, 49408, 5)).to(DEV)
    with torch.no_grad():
        a = m.encode_text(txt)
        assert m._fp8_amax["text"] is not None and m._fp8_amax["vit"] is None
        m.pack_text = False
        b = m.encode_text(txt)
    assert torch.equal(a, b)


def test_twdh_codes_through_fp8_encoders():
    """BASELINE configs[4]: TwDH long (128 bit) + short (16 bit) codes with fp8 CLIP encoders; against the same model in the f32
    parity mode the codes differ only where a pair probability sits near 0.5."""
    import recipe
    from model.TwDH import MTwDH
    from train.TwDH.hash_train import synthetic_assets
    seed, K, C = 3, 128, 21
    sd = {k: torch.from_numpy(v) for k, v in recipe.clip_state_dict(recipe.CLIP_VITB32, seed).items()}
    lc, sc, tr = synthetic_assets(C, K, short_dims=(16,), seed=seed)
    torch.manual_seed(seed)
    model = MTwDH(outputDim=K, clipPath=sd, saveDir="/tmp/cmh_fp8_twdh", long_center=lc, short_center=sc, trans=tr).to(DEV)
    model.float()
    model.eval()
    import cmh_native as Nn
    B = 16
    img = torch.from_numpy(recipe.images(B, 224, seed)).to(DEV)
    txt = torch.from_numpy(recipe.captions(B, 32, 49408, seed)).to(DEV)
    codes = {}
    with torch.no_grad():
        for mode in ("fp8", "f32"):
            model.clip.set_gemm_dtype(mode)
            li, si = model.encode_image(img)
            lt, st = model.encode_text(txt)
            codes[mode] = [Nn.pair_argmax_codes(v.reshape(B, -1)) for v in (li, si["16"], lt, st["16"])]
    for name, a, r in zip(("long image", "short image", "long text", "short text"), codes["fp8"], codes["f32"]):
        assert a.shape == r.shape and a.shape[1] in (128, 16)
        assert set(a.unique().tolist()) <= {-1.0, 1.0}
        rate = float((a != r).float().mean())
        print(f"TwDH {name} codes, fp8 vs f32 encoders: {rate:.4f} of the bits differ")
        assert rate < 0.10, (name, rate)


def test_amax_reports_nan_as_inf_and_stale_scales_recalibrate():
    """cmh_amax must not let a NaN hide (fmaxf drops NaN operands): it reports +inf, which calibrate_fp8 refuses.  And activation
    scales are tied to the weights they were measured with: a changed block parameter re-calibrates on the next fp8 batch."""
    import ctypes as C
    import warnings
    import cmh_native as Nn
    x = torch.randn(4096, device=DEV)
    x[1234] = float("nan")
    out = torch.zeros(1, device=DEV)
    Nn.check(Nn.lib().cmh_amax(Nn.ptr(x), 0, x.numel(), Nn.ptr(out), Nn.stream_ptr(x.device)), "cmh_amax")
    assert float(out) == float("inf")
    w = torch.randn(8, 256, device=DEV)
    w[3, 77] = float("nan")
    _, cs = Nn.fp8_quantize_weight(w)
    assert bool(torch.isnan(cs[3])) and bool(torch.isfinite(cs[[0, 1, 2, 4, 5, 6, 7]]).all())
    import recipe
    from test_gpu_clip import _clip
    cfg = dict(recipe.CLIP_TINY, vision_width=256, transformer_width=256, transformer_heads=4)
    clip = _clip(cfg, 5, "fp8")
    img = torch.from_numpy(recipe.images(4, cfg["image_resolution"], 5)).to(DEV)
    with torch.no_grad():
        clip.encode_image(img)
        first = clip._fp8_amax["vit"]
        assert first is not None
        clip.visual.transformer.resblocks[0].mlp.c_fc.weight.mul_(3.0)          # a "training step"
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            clip.encode_image(img)
        assert any("re-calibrating" in str(r.message) for r in rec)
        assert clip._fp8_amax["vit"] != first


@pytest.mark.parametrize("M,N,K", [(256, 768, 768), (256, 3072, 768), (256, 768, 3072), (200, 512, 2048), (1600, 1536, 512)])
def test_fp8_gemm_rows_kernel_gives_the_wide_kernels_bits(M, N, K):
    """The few-row kernel's e4m3 instantiation (csrc/gemm_rows.hip, the fp8 mode's pooled-row tail) against the wide kernel on the same
    operands: identical bits for plain, bias + QuickGELU -> bf16, bias + fp16 residual -> fp16 and bias + QuickGELU -> e4m3 outputs."""
    import cmh_native as Nn
    g = torch.Generator().manual_seed(M + 2 * N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * (torch.rand(N, 1, generator=g) + 0.1) * K ** -0.5
    b = torch.randn(N, generator=g).to(DEV)
    r = torch.randn(M, N, generator=g).half().to(DEV)
    ax = float(x.abs().max()) / 448
    x8 = Nn.fp8_quantize(x.to(DEV), ax)
    w8, cs = Nn.fp8_quantize_weight(w.to(DEV))
    outs = {}
    try:
        for on in (1, 0):
            Nn.set_gemm_rows(on)
            outs[on] = (Nn.linear_gemm_fp8(x8, w8, cs, ax), Nn.linear_gemm_fp8(x8, w8, cs, ax, bias=b, quickgelu=True, out="bf16"),
                        Nn.linear_gemm_fp8(x8, w8, cs, ax, bias=b, residual=r, out="f16"),
                        Nn.linear_gemm_fp8(x8, w8, cs, ax, bias=b, quickgelu=True, out="fp8", out_scale=0.05))
    finally:
        Nn.set_gemm_rows(-1)
    for i, (a_, c_) in enumerate(zip(outs[1], outs[0])):
        assert torch.equal(a_.view(torch.uint8) if a_.dtype not in (torch.float32, torch.float16, torch.bfloat16) else a_,
                           c_.view(torch.uint8) if c_.dtype not in (torch.float32, torch.float16, torch.bfloat16) else c_), i
    base = Nn.fp8_dequantize(x8, ax).cpu().double() @ (Nn.fp8_dequantize(w8).cpu() * cs.cpu()[:, None]).double().t()
    torch.testing.assert_close(outs[1][0].cpu().double(), base, rtol=1e-4, atol=1e-4 * float(base.abs().max()))

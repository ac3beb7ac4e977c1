"""Oracle (numpy restatement) vs goldens produced by the reference itself."""
import numpy as np
import pytest

import recipe
from oracle import clip_oracle as co

TOL = dict(rtol=2e-4, atol=2e-5)   # fp32 CPU (torch ATen) vs fp32 numpy/OpenBLAS summation order


def test_clip_tiny_matches_reference(golden):
    g = golden("clip_tiny.npz")
    cfg, seed = recipe.CLIP_TINY, int(g["seed"])
    sd = recipe.clip_state_dict(cfg, seed)
    taps = {}
    img = co.encode_image(sd, recipe.images(3, cfg["image_resolution"], seed), taps)
    np.testing.assert_allclose(taps["v_ln_pre"], g["v_ln_pre"], **TOL)
    for i in range(cfg["vision_layers"]):
        np.testing.assert_allclose(taps[f"v_block{i}"], g[f"v_block{i}"], **TOL)
    np.testing.assert_allclose(img, g["img_feat"], **TOL)
    for L, s in ((16, seed), (9, seed + 1)):
        taps = {}
        txt = co.encode_text(sd, recipe.captions(3, L, cfg["vocab_size"], s), taps)
        for i in range(cfg["transformer_layers"]):
            np.testing.assert_allclose(taps[f"t_block{i}"], g[f"t_block{i}_L{L}"], **TOL)
        np.testing.assert_allclose(txt, g[f"txt_feat_L{L}"], **TOL)


def test_clip_vitb32_matches_reference(golden):
    g = golden("clip_vitb32.npz")
    cfg, seed = recipe.CLIP_VITB32, int(g["seed"])
    sd = recipe.clip_state_dict(cfg, seed)
    rows = g["v_rows"]
    taps = {}
    img = co.encode_image(sd, recipe.images(2, 224, seed), taps)
    np.testing.assert_allclose(taps["v_ln_pre"][:, rows], g["v_ln_pre_rows"], **TOL)
    for i in (0, 5, 11):
        np.testing.assert_allclose(taps[f"v_block{i}"][:, rows], g[f"v_block{i}_rows"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(img, g["img_feat"], rtol=1e-3, atol=1e-4)
    taps = {}
    txt = co.encode_text(sd, recipe.captions(2, 77, cfg["vocab_size"], seed), taps)
    for i in (0, 5, 11):
        np.testing.assert_allclose(taps[f"t_block{i}"][:, rows], g[f"t_block{i}_L77_rows"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(txt, g["txt_feat_L77"], rtol=1e-3, atol=1e-4)
    txt32 = co.encode_text(sd, recipe.captions(2, 32, cfg["vocab_size"], seed + 1))
    np.testing.assert_allclose(txt32, g["txt_feat_L32"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("K", [16, 64])
def test_baseclip_heads_match_reference(golden, K):
    g = golden("baseclip_tiny.npz")
    cfg, seed = recipe.CLIP_TINY, int(g["seed"])
    sd = recipe.clip_state_dict(cfg, seed, fp16_roundtrip=True)   # load_clip -> build_model path
    fi = co.encode_image(sd, recipe.images(3, cfg["image_resolution"], seed))
    ft = co.encode_text(sd, recipe.captions(3, 16, cfg["vocab_size"], seed))
    np.testing.assert_allclose(fi, g["feat_img_fp16w"], **TOL)
    np.testing.assert_allclose(ft, g["feat_txt_fp16w"], **TOL)
    for side, f in (("image", fi), ("text", ft)):
        s = "img" if side == "image" else "txt"
        w, b = recipe.head_linear(cfg["embed_dim"], K, seed, f"dsph_{side}_{K}")
        h = co.linear_hash(f, w, b)
        np.testing.assert_allclose(h, g[f"dsph_{s}_K{K}"], **TOL)
        safe = np.abs(g[f"dsph_{s}_K{K}"]) > 1e-4
        assert np.array_equal(co.sign_codes(h)[safe], g[f"dsph_{s}_code_K{K}"][safe])
        w1, b1 = recipe.head_linear(cfg["embed_dim"], 128, seed, f"dchmt_{side}_fc_{K}")
        w2, b2 = recipe.head_linear(128, 2 * K, seed, f"dchmt_{side}_bits_{K}")
        pr = co.dchmt_hash_layer(f, w1, b1, w2, b2)
        np.testing.assert_allclose(pr, g[f"dchmt_{s}_K{K}"], **TOL)
        ref = g[f"dchmt_{s}_K{K}"]
        safe = np.abs(ref[..., 0] - ref[..., 1]) > 1e-4
        assert np.array_equal(co.dchmt_codes(pr)[safe], g[f"dchmt_{s}_code_K{K}"][safe])


def test_dnph_heads_match_reference(golden):
    g = golden("baseclip_tiny.npz")
    cfg, seed = recipe.CLIP_TINY, int(g["seed"])
    sd = recipe.clip_state_dict(cfg, seed, fp16_roundtrip=True)
    fi = co.encode_image(sd, recipe.images(3, cfg["image_resolution"], seed))
    w, b = recipe.head_linear(cfg["embed_dim"], 16, seed, "dnph_image_hash")
    np.testing.assert_allclose(co.linear_hash(fi, w, b), g["dnph_img"], **TOL)
    w, b = recipe.head_linear(cfg["embed_dim"], 21, seed, "dnph_image_pre")
    np.testing.assert_allclose(co.pre_layer(fi, w, b), g["dnph_img_pre"], **TOL)


def test_torch_cpu_port_matches_the_numpy_oracle():
    """oracle/torch_cpu.py (the timed CPU baseline's second form) against clip_oracle on the tiny configuration."""
    import recipe
    from oracle import clip_oracle as co
    from oracle.torch_cpu import TorchClip
    cfg = recipe.CLIP_TINY
    sd = recipe.clip_state_dict(cfg, 5)
    img = recipe.images(3, cfg["image_resolution"], 2)
    txt = recipe.captions(3, cfg["context_length"], cfg["vocab_size"], 2)
    tc = TorchClip(sd)
    np.testing.assert_allclose(tc.encode_image(img).numpy(), co.encode_image(sd, img), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(tc.encode_text(txt).numpy(), co.encode_text(sd, txt), rtol=2e-4, atol=2e-4)

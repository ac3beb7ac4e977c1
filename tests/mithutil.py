"""Seeded MITH fixtures shared by tests/golden/make_golden3.py and the tests."""
import zlib

import numpy as np

import recipe

HP = dict(hyper_lambda=0.99, hyper_tokens_intra=1.0, hyper_cls_inter=10.0, hyper_quan=8.0, hyper_info_nce=50.0,
          hyper_alpha=0.01, hyper_distill=1.0)
ARGS = dict(dropout=0.0, transformer_layers=2, activation="gelu", top_k_label=8, res_mlp_layers=2)
CLIP_TINY512 = dict(recipe.CLIP_TINY, embed_dim=512)


def fill_state(shapes: dict, seed: int) -> dict:
    """{key: shape} -> {key: f32 array}; LayerNorm-like weights near 1, biases small, matrices ~ N(0, 1/fan_in)."""
    out = {}
    for key in sorted(shapes):
        shape = tuple(shapes[key])
        alias = key.replace("gcl_t.", "gcl_i.")          # gcl_i and gcl_t are ONE shared module (model/MITH.py:412)
        rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(alias.encode())]))
        n = rng.standard_normal(shape)
        if key.endswith("pe"):
            continue
        if (".lns." in key or ".ln_" in key) and key.endswith("weight"):
            a = 1 + 0.1 * n
        elif key.endswith("bias"):
            a = 0.02 * n
        else:
            a = n * (1.5 / np.sqrt(shape[-1]))
        out[key] = a.astype(np.float32)
    return out


def hash_inputs(Nb, L, K, seed=61):
    r = lambda *s, tag: (recipe._rng(seed, tag).standard_normal(s) * 0.6).astype(np.float32)
    c = dict(img_tokens=r(49, Nb, 512, tag="mith_it"), txt_tokens=r(L, Nb, 512, tag="mith_tt"),
             img_cls=r(Nb, 512, tag="mith_ic"), txt_eos=r(Nb, 512, tag="mith_te"))
    kpm = np.zeros((Nb, L), bool)
    lens = recipe._rng(seed, "mith_len").integers(3, L, size=Nb)
    for i, n in enumerate(lens):
        kpm[i, n:] = True          # EOT position n and the padding behind it (new_key_padding_mask)
    kpm[0, :] = False
    kpm[0, L - 1] = True
    c["kpm"] = kpm
    return c


def loss_inputs(Nb, K, C, Mb, seed=71):
    f = lambda *s, tag: recipe._rng(seed, tag).standard_normal(s).astype(np.float32)
    out = {k: np.tanh(f(Nb, K, tag=f"ml_{k}")) for k in ("img_cls_hash", "txt_cls_hash", "img_tokens_hash", "txt_tokens_hash")}
    nz = lambda v: (v / np.sqrt((v * v).sum(-1, keepdims=True))).astype(np.float32)
    out["res_img_cls"], out["res_txt_cls"] = nz(f(Nb, 512, tag="ml_ri")), nz(f(Nb, 512, tag="ml_rt"))
    out["trans_tokens_i"], out["trans_tokens_t"] = nz(f(K, Nb, 512, tag="ml_ti")), nz(f(K, Nb, 512, tag="ml_tt"))
    banks = {k: f(Mb, K, tag=f"ml_bank_{k}") for k in ("img_tokens", "img_cls", "txt_tokens", "txt_cls")}
    label = recipe.labels(Nb, C, seed, tag="ml_label")
    train_labels = recipe.labels(Mb, C, seed, tag="ml_train_labels")
    return out, banks, label, train_labels

"""Rebuild the TwDH / DNPH fixture inputs from the recipe (mirrors tests/golden/make_golden2.py)."""
import numpy as np

import recipe


def twdh_head_params(K, seed, side):
    d = 512
    p = {}
    p["in_w"] = (recipe.features(3 * d, d, seed, f"twdh_{side}_inw_{K}") * 0.1).astype(np.float32)
    p["in_b"] = (recipe.features(1, 3 * d, seed, f"twdh_{side}_inb_{K}")[0] * 0.1).astype(np.float32)
    p["out_w"] = (recipe.features(d, d, seed, f"twdh_{side}_ow_{K}") * 0.1).astype(np.float32)
    p["out_b"] = (recipe.features(1, d, seed, f"twdh_{side}_ob_{K}")[0] * 0.1).astype(np.float32)
    p["norm_w"] = (1 + 0.2 * recipe.features(1, d, seed, f"twdh_{side}_nw_{K}")[0]).astype(np.float32)
    p["norm_b"] = (0.1 * recipe.features(1, d, seed, f"twdh_{side}_nb_{K}")[0]).astype(np.float32)
    p["fc2_w"], p["fc2_b"] = recipe.head_linear(d, 2 * K, seed, f"twdh_{side}_fc2_{K}")
    return p


def twdh_case(B, K, S, C, seed=41):
    tag = f"B{B}_K{K}"
    c = dict(tag=tag)
    c["feat_i"] = recipe.features(B, 512, seed, f"twdh_fi_{tag}")
    c["feat_t"] = recipe.features(B, 512, seed, f"twdh_ft_{tag}")
    c["p_img"], c["p_txt"] = twdh_head_params(K, seed, "img"), twdh_head_params(K, seed, "txt")
    c["trans"] = (recipe.features(2 * K, 2 * S, seed, f"twdh_trans_{tag}") * (2 * K) ** -0.5 * 4).astype(np.float32)
    lab = recipe.labels(B, C, seed, p=0.12, tag=f"twdh_lab_{tag}")
    lab[0] = 0
    c["labels"] = lab
    c["lc"] = recipe.sign_codes(C, K, seed, f"twdh_lc_{tag}")
    c["sc"] = recipe.sign_codes(C, S, seed, f"twdh_sc_{tag}")
    return c


TWDH_CASES = [(12, 16, 8, 24), (32, 128, 16, 21)]
DNPH_CASES = [(16, 16, 21), (40, 128, 24)]


def dnph_case(B, K, C, seed=51):
    tag = f"B{B}_K{K}_C{C}"
    return dict(tag=tag, prox=(recipe.features(C, K, seed, f"dnph_prox_{tag}") / 4).astype(np.float32),
                hi=np.tanh(recipe.features(B, K, seed, f"dnph_hi_{tag}")), ht=np.tanh(recipe.features(B, K, seed, f"dnph_ht_{tag}")),
                pi=recipe.features(B, C, seed, f"dnph_pi_{tag}"), pt=recipe.features(B, C, seed, f"dnph_pt_{tag}"),
                lab=recipe.labels(B, C, seed, p=0.15, tag=f"dnph_lab_{tag}"))


def dnph_noise(B, K, seed=51):
    """+-1 noise rows for the real-size DNPH case (what b_reg.gene_noise hands out; which row goes to which sample is the host's
    Hungarian step, tested apart)"""
    return tuple(np.where(recipe.features(B, K, seed, f"dnph_real_noise_{s}") >= 0, 1.0, -1.0).astype(np.float32) for s in "it")

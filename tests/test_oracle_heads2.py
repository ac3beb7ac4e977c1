"""TwDH / DNPH numpy restatements vs goldens produced by the reference (tests/golden/make_golden2.py)."""
import numpy as np
import pytest

from heads2util import DNPH_CASES, TWDH_CASES, dnph_case, twdh_case
from oracle import heads2_oracle as h2

TOL = dict(rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("B,K,S,C", TWDH_CASES)
def test_twdh_heads_targets_loss(golden, B, K, S, C):
    g = golden("twdh.npz")
    c = twdh_case(B, K, S, C)
    tag = c["tag"]
    li = h2.twdh_modality_hash(c["feat_i"], **c["p_img"], layernorm=False)
    lt = h2.twdh_modality_hash(c["feat_t"], **c["p_txt"], layernorm=True)
    np.testing.assert_allclose(li, g[f"{tag}_img_long"], **TOL)
    np.testing.assert_allclose(lt, g[f"{tag}_txt_long"], **TOL)
    si, st = h2.twdh_short_hash(li, c["trans"]), h2.twdh_short_hash(lt, c["trans"])
    np.testing.assert_allclose(si, g[f"{tag}_img_short"], **TOL)
    np.testing.assert_allclose(st, g[f"{tag}_txt_short"], **TOL)
    tl = h2.twdh_targets(c["labels"], c["lc"], g[f"{tag}_rc_long"])
    np.testing.assert_array_equal(tl, g[f"{tag}_target_long"])
    ts = h2.twdh_targets(c["labels"], c["sc"], g[f"{tag}_rc_short"])
    nce, quan = h2.twdh_loss_terms(g[f"{tag}_img_long"], g[f"{tag}_txt_long"], tl)
    nce_s, quan_s = h2.twdh_loss_terms(g[f"{tag}_img_short"], g[f"{tag}_txt_short"], ts)
    loss = nce + 0.5 * quan + 0.3 * nce_s + 0.3 * quan_s
    assert abs(loss - float(g[f"{tag}_loss"])) < 1e-5 * max(1, abs(loss))


@pytest.mark.parametrize("B,K,C", DNPH_CASES)
def test_dnph_loss(golden, B, K, C):
    g = golden("dnph.npz")
    c = dnph_case(B, K, C)
    tag = c["tag"]
    l1 = h2.dnph_out_loss(c["hi"], c["ht"], c["pi"], c["pt"], c["lab"], c["prox"])
    assert abs(l1 - float(g[f"{tag}_loss1"])) < 1e-5 * max(1, abs(l1))
    ls = h2.dnph_step_loss(c["hi"], c["ht"], c["pi"], c["pt"], c["lab"], c["prox"], g[f"{tag}_noise_i"].astype(np.float64),
                           g[f"{tag}_noise_t"].astype(np.float64))
    assert abs(ls - float(g[f"{tag}_step_loss"])) < 1e-5 * max(1, abs(ls))


@pytest.mark.parametrize("B,K,C", DNPH_CASES)
def test_dnph_hungarian_host_side(golden, B, K, C):
    """The product's host-side gene_noise (scipy Hungarian, like upstream) reproduces the reference's assignment."""
    from train.DNPH_TOMM.b_reg import gene_noise
    g = golden("dnph.npz")
    c = dnph_case(B, K, C)
    s_vec = g[f"{c['tag']}_s_vec"].astype(np.int64)
    np.testing.assert_array_equal(gene_noise(c["hi"], s_vec), g[f"{c['tag']}_noise_i"].astype(np.float64))
    np.testing.assert_array_equal(gene_noise(c["ht"], s_vec), g[f"{c['tag']}_noise_t"].astype(np.float64))

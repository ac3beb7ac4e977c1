"""numpy loss restatements vs goldens from the reference's loss code."""
import json
import math
import os

import numpy as np
import pytest

import recipe
from conftest import PKG
from oracle import clip_oracle as co

DSPH = [(32, 64, 24), (48, 16, 80), (16, 128, 21), (8, 32, 24)]


@pytest.mark.parametrize("B,K,C", DSPH)
def test_dsph_hyp_loss(golden, B, K, C):
    g = golden("loss_dsph.npz")
    tag, seed = f"B{B}_K{K}_C{C}", 21
    prox = recipe.features(C, K, seed, f"dsph_prox_{tag}")
    x = np.tanh(recipe.features(B, K, seed, f"dsph_x_{tag}"))
    y = np.tanh(recipe.features(B, K, seed, f"dsph_y_{tag}"))
    lab = recipe.labels(B, C, seed, p=float(g[f"{tag}_p"]), tag=f"dsph_lab_{tag}")
    loss = co.dsph_hyp_loss(x, y, lab, prox, float(g[f"{tag}_threshold"]), float(g[f"{tag}_alpha"]))
    assert abs(float(loss) - float(g[f"{tag}_loss"])) < 1e-5 * max(1, abs(float(loss)))


@pytest.mark.parametrize("B,K,C", DSPH)
def test_dsph_threshold_table(golden, B, K, C):
    g = golden("loss_dsph.npz")
    rows = json.load(open(os.path.join(PKG, "train", "DSPH", "codetable.json")))["rows"]
    assert rows[str(K)][math.ceil(math.log(C, 2))] == float(g[f"B{B}_K{K}_C{C}_threshold"])


@pytest.mark.parametrize("B,K,C,fn,lt", [(32, 16, 24, "euclidean", "l2"), (32, 16, 24, "cosine", "l2"),
                                          (24, 64, 24, "euclidean", "l1"), (24, 64, 80, "cosine", "l1")])
def test_dchmt_our_loss(golden, B, K, C, fn, lt):
    g = golden("loss_dchmt.npz")
    tag, seed = f"B{B}_K{K}_C{C}_{fn}_{lt}", 31
    zi = recipe.features(B, 2 * K, seed, f"dchmt_zi_{tag}").reshape(B, K, 2)
    zt = recipe.features(B, 2 * K, seed, f"dchmt_zt_{tag}").reshape(B, K, 2)
    hi = co.softmax(2 * zi, -1).reshape(B, 2 * K).astype(np.float32)
    ht = co.softmax(2 * zt, -1).reshape(B, 2 * K).astype(np.float32)
    lab = recipe.labels(B, C, seed, tag=f"dchmt_lab_{tag}")
    loss = co.dchmt_our_loss(hi, ht, lab, K, fn, lt)
    assert abs(float(loss) - float(g[f"{tag}_loss"])) < 1e-4 * max(1, abs(float(loss)))


@pytest.mark.parametrize("B,K,C,p", [(8, 16, 24, 0.15), (48, 32, 80, 0.05), (256, 64, 24, 0.15), (16, 128, 21, 0.0)])
def test_dnph_tmm_qmi_loss(golden, B, K, C, p):
    """numpy restatement of train/DNpH_TMM/loss.py:5-72 against the reference's own values (tests/golden/make_golden15.py)."""
    from qmiutil import qmi_case
    from oracle.qmi_oracle import qmi_loss
    g = golden("qmi.npz")
    c = qmi_case(B, K, C, p)
    loss = qmi_loss(c["x"], c["y"], c["lab"])
    want = float(g[f"{c['tag']}_loss"])
    assert abs(float(loss) - want) < 2e-5 * max(1.0, abs(want)), (float(loss), want)


@pytest.mark.parametrize("B,K,C,p,epoch", [(8, 16, 24, 0.3, 0), (48, 32, 80, 0.05, 3), (256, 64, 24, 0.15, 7), (6, 16, 4, 1.0, 1)])
def test_dmsh_ln_label_net_and_multi_similarity_loss(golden, B, K, C, p, epoch):
    """numpy restatement of train/DMsH_LN/labelnet.py:13-21 and MSLOSS.py:13-55 against the reference's own values
    (tests/golden/make_golden16.py): the label codes, then the three losses of a training step."""
    from mslutil import msl_case
    from oracle.msl_oracle import label_net, msl_loss
    g = golden("msl.npz")
    c = msl_case(B, K, C, p, epoch)
    tag = c["tag"]
    feat, hid, code = label_net(c["lab"], c["w1"], c["b1"], c["w2"], c["b2"], epoch)
    np.testing.assert_allclose(hid, g[f"{tag}_ln_hid"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(code, g[f"{tag}_ln_code"], rtol=1e-4, atol=1e-5)
    code = g[f"{tag}_ln_code"]
    for name, (a, b) in (("ii", (c["x"], None)), ("tt", (c["y"], None)), ("it", (c["x"], c["y"]))):
        want = float(g[f"{tag}_loss_{name}"])
        got = float(msl_loss(a, code, b))
        assert abs(got - want) < 2e-5 * max(1.0, abs(want)), (name, got, want)


@pytest.mark.parametrize("B,K,C,p,epoch,total", [(8, 16, 24, 0.3, 1, 100), (48, 32, 80, 0.08, 20, 100), (256, 64, 24, 0.15, 50, 100), (32, 128, 21, 0.2, 3, 2)])
def test_dhaph_self_paced_loss(golden, B, K, C, p, epoch, total):
    """numpy restatement of train/DHaPH/MSLoss.py:13-33 against the reference's own values (tests/golden/make_golden17.py)."""
    from mslutil import spl_case
    from oracle.msl_oracle import spl_loss
    g = golden("spl.npz")
    c = spl_case(B, K, C, p, epoch, total)
    tag = c["tag"]
    for name, (a, b) in (("ii", (c["x"], c["x"])), ("tt", (c["y"], c["y"])), ("it", (c["x"], c["y"]))):
        want = float(g[f"{tag}_loss_{name}"])
        got = float(spl_loss(a, b, c["lab"], epoch, totalepoch=total))
        assert abs(got - want) < 2e-5 * max(1.0, abs(want)), (name, got, want)
    want = float(g[f"{tag}_loss_it_plain"])
    assert abs(float(spl_loss(c["x"], c["y"], c["lab"], epoch, totalepoch=total, self_paced=False)) - want) < 2e-5 * max(1.0, abs(want))

"""DMsH-LN on the GPU: LabelNet (cmh_linear_act) and the multi-similarity loss (cmh_msl_loss / cmh_msl_loss_backward) against the
REFERENCE's own values and autograd gradients (train/DMsH_LN/labelnet.py:6-21, MSLOSS.py:4-55; tests/golden/make_golden16.py), the
three calls of a training step summed as the trainer sums them (train/DMsH_LN/hash_train.py:58-61)."""
import numpy as np
import pytest
import torch

from mslutil import CASES, SPL_CASES, msl_case, spl_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B,K,C,p,epoch", CASES)
def test_label_net_matches_reference(golden, B, K, C, p, epoch):
    from train.DMsH_LN.labelnet import LabelNet
    g = golden("msl.npz")
    c = msl_case(B, K, C, p, epoch)
    net = LabelNet(label_dim=C, code_len=K).to(DEV)
    with torch.no_grad():
        for prm, name in ((net.fc1.weight, "w1"), (net.fc1.bias, "b1"), (net.fc2.weight, "w2"), (net.fc2.bias, "b2")):
            prm.copy_(torch.from_numpy(c[name]))
    net.set_alpha(epoch)
    feat, hid, code = net(torch.from_numpy(c["lab"]), device=DEV)
    tag = c["tag"]
    np.testing.assert_allclose(feat.cpu().numpy(), g[f"{tag}_ln_feat"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(hid.cpu().numpy(), g[f"{tag}_ln_hid"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(code.cpu().numpy(), g[f"{tag}_ln_code"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,K,C,p,epoch", CASES)
def test_msl_losses_and_gradients_match_reference(golden, B, K, C, p, epoch):
    from train.DMsH_LN.MSLOSS import MultiSimilarityLoss
    g = golden("msl.npz")
    c = msl_case(B, K, C, p, epoch)
    tag = c["tag"]
    msl = MultiSimilarityLoss()
    x = torch.from_numpy(c["x"]).to(DEV).requires_grad_()
    y = torch.from_numpy(c["y"]).to(DEV).requires_grad_()
    code = torch.from_numpy(g[f"{tag}_ln_code"]).to(DEV)
    losses = dict(ii=msl(x, code), tt=msl(y, code), it=msl(x, code, feat2=y))
    for name, l in losses.items():
        want = float(g[f"{tag}_loss_{name}"])
        assert abs(float(l.detach()) - want) < 1e-4 * max(1.0, abs(want)), (name, float(l.detach()), want)
        with torch.no_grad():
            plain = msl(*((x, code) if name == "ii" else (y, code) if name == "tt" else (x, code)), **({"feat2": y} if name == "it" else {}))
        assert float(plain) == float(l.detach())
    (2.0 * (losses["ii"] + losses["tt"] + losses["it"])).backward()              # exercises the upstream-gradient scaling too
    for got, name in ((x.grad, "gx"), (y.grad, "gy")):
        ref = 2.0 * g[f"{tag}_{name}"]
        if got is None:                      # every row skipped: the loss is the reference's constant zero, no gradient reaches the inputs
            assert not ref.any() and all(float(l.detach()) == 0.0 for l in losses.values())
            continue
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-4, atol=2e-5 * max(np.abs(ref).max(), 1e-30), err_msg=name)


def test_msl_refuses_what_is_not_built():
    from train.DMsH_LN.MSLOSS import MultiSimilarityLoss
    x = torch.zeros(4, 8, device=DEV)
    with pytest.raises(NotImplementedError):
        MultiSimilarityLoss()(x, x, dataset="cifar10-1")


@pytest.mark.parametrize("B,K,C,p,epoch,total", SPL_CASES)
def test_dhaph_self_paced_losses_and_gradients_match_reference(golden, B, K, C, p, epoch, total):
    """cmh_spl_loss / cmh_spl_loss_backward against the reference's MSLoss (train/DHaPH/MSLoss.py:6-33) and its autograd gradients,
    the three calls of a training step summed as the trainer sums them (train/DHaPH/hash_train.py:68-70, 76)."""
    from train.DHaPH.MSLoss import MSLoss
    g = golden("spl.npz")
    c = spl_case(B, K, C, p, epoch, total)
    tag = c["tag"]
    crit = MSLoss(temperature=0.3, totalepoch=total, self_paced=True)
    x = torch.from_numpy(c["x"]).to(DEV).requires_grad_()
    y = torch.from_numpy(c["y"]).to(DEV).requires_grad_()
    lab = torch.from_numpy(c["lab"]).to(DEV)
    losses = dict(ii=crit(x, x, lab, epoch), tt=crit(y, y, lab, epoch), it=crit(x, y, lab, epoch))
    for name, l in losses.items():
        want = float(g[f"{tag}_loss_{name}"])
        assert abs(float(l.detach()) - want) < 1e-4 * max(1.0, abs(want)), (name, float(l.detach()), want)
    with torch.no_grad():
        plain = MSLoss(temperature=0.3, totalepoch=total, self_paced=False)(x, y, lab, epoch)
    want = float(g[f"{tag}_loss_it_plain"])
    assert abs(float(plain) - want) < 1e-4 * max(1.0, abs(want))
    (2.0 * (losses["ii"] + losses["tt"] + losses["it"])).backward()
    for got, name in ((x.grad, "gx"), (y.grad, "gy")):
        ref = 2.0 * g[f"{tag}_{name}"]
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-4, atol=2e-5 * np.abs(ref).max(), err_msg=name)

"""DNpH's qmi_loss on the GPU (cmh_qmi_loss / cmh_qmi_loss_backward) against the REFERENCE's own values and autograd gradients
(train/DNpH_TMM/loss.py:5-72; tests/golden/make_golden15.py)."""
import numpy as np
import pytest
import torch

from qmiutil import CASES, qmi_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B,K,C,p", CASES)
def test_qmi_loss_and_gradients_match_reference(golden, B, K, C, p):
    from train.DNpH_TMM.loss import qmi_loss
    g = golden("qmi.npz")
    c = qmi_case(B, K, C, p)
    tag = c["tag"]
    x = torch.from_numpy(c["x"]).to(DEV).requires_grad_()
    y = torch.from_numpy(c["y"]).to(DEV).requires_grad_()
    lab = torch.from_numpy(c["lab"]).to(DEV)
    with torch.no_grad():
        plain = qmi_loss(images=x, texts=y, targets=lab)
    loss = qmi_loss(images=x, texts=y, targets=lab)
    want = float(g[f"{tag}_loss"])
    assert abs(float(plain) - want) < 1e-4 * max(1.0, abs(want)) and float(plain) == float(loss.detach())
    (2.0 * loss).backward()                                       # exercises the upstream-gradient scaling too
    for got, name in ((x.grad, "gx"), (y.grad, "gy")):
        ref = 2.0 * g[f"{tag}_{name}"]
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-4, atol=2e-5 * np.abs(ref).max(), err_msg=name)


def test_qmi_loss_refuses_what_is_not_built():
    from train.DNpH_TMM.loss import qmi_loss
    x = torch.zeros(4, 8, device=DEV)
    with pytest.raises(NotImplementedError):
        qmi_loss(x, x, torch.ones(4, 3, device=DEV), use_cosine=False)

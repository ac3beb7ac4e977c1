"""Loss kernels at the REAL sizes of BASELINE configs[2] / configs[3] against the reference's own outputs
(tests/golden/make_golden14.py): MITH step loss + gradients at batch 256 x 64 bit x 80 classes with the 10 000-row memory bank,
DNPH step loss + gradients at 2B = 512 rows x 128 bit x 21 classes."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def tt(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def test_mith_step_loss_and_gradients_at_config_size(golden):
    import mithutil as mu
    from train.MITH.hash_train import MITHTrainer
    g = golden("real_size.npz")
    rows = g["rows"].tolist()
    Nb, K, C, Mb = 256, 64, 80, 10000
    od, banks, label, train_labels = mu.loss_inputs(Nb, K, C, Mb)
    self = SimpleNamespace(args=SimpleNamespace(**mu.HP), rank=0, k_bits=K, train_labels=tt(train_labels),
                           img_buffer_tokens=tt(banks["img_tokens"]), img_buffer_cls=tt(banks["img_cls"]),
                           txt_buffer_tokens=tt(banks["txt_tokens"]), txt_buffer_cls=tt(banks["txt_cls"]))
    for name in ("bayesian_loss", "info_nce_loss", "info_nce_loss_bmm", "quantization_loss_2", "make_B", "sq_diff", "_grad"):
        setattr(self, name, (lambda n: (lambda *a, **k: getattr(MITHTrainer, n)(self, *a, **k)))(name) if name != "_grad" else MITHTrainer._grad)
    tod = {k: tt(v).requires_grad_() for k, v in od.items()}
    L = MITHTrainer.compute_loss(self, tod, tt(label))
    for k, v in L.items():                                                        # every loss group, then their sum
        want = float(g[f"mith_group_{k}"])
        assert abs(float(v.detach()) - want) < 1e-4 * max(1.0, abs(want)), (k, float(v.detach()), want)
    total = sum(L.values())
    assert abs(float(total.detach()) - float(g["mith_total"])) < 1e-4 * max(1.0, abs(float(g["mith_total"])))
    total.backward()
    for k, v in tod.items():
        want, got = g[f"mith_d_{k}"], v.grad.cpu().numpy()
        got = got[::8][:, rows][:, :, ::16] if k.startswith("trans_tokens") else got[rows]
        np.testing.assert_allclose(got, want, rtol=2e-3, atol=5e-5 * max(np.abs(want).max(), 1e-30), err_msg=k)


def test_dnph_step_loss_and_gradients_at_config_size(golden):
    from heads2util import dnph_case, dnph_noise
    from backward_ops import DnphLoss
    g = golden("real_size.npz")
    rows = g["rows"].tolist()
    B, K, C = 256, 128, 21
    c = dnph_case(B, K, C)
    hi, ht, pi, pt, prox = (tt(c[k]).requires_grad_() for k in ("hi", "ht", "pi", "pt", "prox"))
    ni, nt = (tt(a) for a in dnph_noise(B, K))
    loss = DnphLoss.apply(hi, ht, pi, pt, tt(c["lab"]), prox, ni, nt, 1.0, 0.1)
    assert abs(float(loss.detach()) - float(g["dnph_step_loss"])) < 1e-4 * max(1.0, abs(float(g["dnph_step_loss"])))
    loss.backward()
    for a, name in ((hi, "ghi"), (ht, "ght"), (pi, "gpi"), (pt, "gpt")):
        np.testing.assert_allclose(a.grad.cpu().numpy()[rows], g[f"dnph_{name}"], rtol=2e-4, atol=2e-6, err_msg=name)
    np.testing.assert_allclose(prox.grad.cpu().numpy(), g["dnph_gprox"], rtol=2e-4, atol=2e-6, err_msg="gprox")

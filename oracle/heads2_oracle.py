"""ORACLE — TEST INFRASTRUCTURE ONLY.  numpy restatements for the DNPH (TOMM) and TwDH methods.
Pinned by tests/golden/{twdh,dnph}.npz (tests/golden/make_golden2.py runs the reference)."""
import numpy as np

from .clip_oracle import F32, l2_normalize, layer_norm, linear, softmax


def batchnorm1d_train(x, w, b, eps=1e-5):
    """nn.BatchNorm1d in training mode: batch mean, BIASED batch variance (model/TwDH.py:61,78)."""
    mu = x.mean(0, keepdims=True, dtype=np.float64)
    var = ((x - mu) ** 2).mean(0, keepdims=True, dtype=np.float64)
    return ((x - mu) / np.sqrt(var + eps) * w + b).astype(F32)


def twdh_modality_hash(feat, in_w, in_b, out_w, out_b, norm_w, norm_b, fc2_w, fc2_b, layernorm):
    """model/TwDH.py:74-85: MHA over a length-1 sequence (softmax over one key == 1 -> out_proj(v_proj(x))),
    BatchNorm1d(train) / LayerNorm, fc2, ReLU, pair softmax -> [B, 2K]."""
    d = feat.shape[1]
    v = linear(feat, in_w[2 * d:], in_b[2 * d:])
    e = linear(v, out_w, out_b)
    e = layer_norm(e, norm_w, norm_b) if layernorm else batchnorm1d_train(e, norm_w, norm_b)
    z = np.maximum(linear(e, fc2_w, fc2_b), 0).astype(F32)
    return softmax(z.reshape(z.shape[0], -1, 2), -1).reshape(z.shape[0], -1).astype(F32)


def twdh_short_hash(long_hash, trans):
    """model/TwDH.py:148-155: softmax_hash(long_hash @ trans)."""
    z = (long_hash @ trans).astype(F32)
    return softmax(z.reshape(z.shape[0], -1, 2), -1).reshape(z.shape[0], -1).astype(F32)


def twdh_targets(labels, centers, random_center):
    """train/TwDH/hash_train.py:93-115 hash_center_multilables with the random +-1 filler injected."""
    out = np.empty((labels.shape[0], centers.shape[1]), F32)
    for i, lab in enumerate(labels):
        ones = np.nonzero(lab == 1)[0]
        if len(ones) == 0:
            out[i] = -1          # NaN mean upstream -> hash_convert's (x > 0) is False -> bit 0 == -1
            continue
        m = centers[ones].astype(np.float64).mean(0)
        c = np.where(m > 0, 1.0, np.where(m < 0, -1.0, random_center))
        out[i] = c
    return out


def twdh_loss_terms(p_img, p_txt, target):
    """train/TwDH/hash_train.py:77-91,117-139: BCELoss against hash_convert(target) (log clamped at -100) and
    the soft-argmax quantisation term.  -> (nce, quan)"""
    y = np.stack([(target <= 0), (target > 0)], -1).reshape(target.shape[0], -1).astype(np.float64)

    def bce(p):
        p = p.astype(np.float64)
        return float(np.mean(-(y * np.maximum(np.log(p), -100) + (1 - y) * np.maximum(np.log(1 - p), -100))))

    def quan(p):
        return float(1 - np.mean((2 * p.astype(np.float64) - 1) ** 2))
    return (bce(p_img) + bce(p_txt)) / 2, (quan(p_img) + quan(p_txt)) / 2


def dnph_out_loss(f1, f2, pre1, pre2, label, proxies, mrg=1.0):
    """train/DNPH_TOMM/loss.py:14-32 DNPH_out.forward (label_1 == label_2)."""
    fa = l2_normalize(np.concatenate([f1, f2], 0)).astype(np.float64)
    la = np.concatenate([label, label], 0).astype(np.float64)
    pn = l2_normalize(proxies).astype(np.float64)
    D = ((fa[:, None, :] - pn[None, :, :]) ** 2).sum(-1) + mrg * (la == 1)
    z = -D
    lse = np.log(np.exp(z - z.max(1, keepdims=True)).sum(1, keepdims=True)) + z.max(1, keepdims=True)
    p_loss = np.mean(np.sum(-la * (z - lse), -1))

    def ce(pre):
        pre = pre.astype(np.float64)
        t = label.argmax(-1)
        l = np.log(np.exp(pre - pre.max(1, keepdims=True)).sum(1)) + pre.max(1)
        return float(np.mean(l - pre[np.arange(len(t)), t]))
    return float(p_loss + ce(pre1) + ce(pre2))


def dnph_step_loss(f1, f2, pre1, pre2, label, proxies, noise1, noise2):
    """train/DNPH_TOMM/hash_train.py:70-81: loss1 - 0.1 * (mean <h_img, n_img> + mean <h_txt, n_txt>)."""
    noise = float((f1.astype(np.float64) * noise1).sum(-1).mean() + (f2.astype(np.float64) * noise2).sum(-1).mean())
    return dnph_out_loss(f1, f2, pre1, pre2, label, proxies) - 0.1 * noise
